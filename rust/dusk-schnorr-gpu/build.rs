// Link against the engine built by `python -m schnorr_amd.build` (schnorr_amd/libdsv.so).
// DSV_NO_LINK=1: skip it — for the two CPU-only tools (`golden_gen`, `bench_ref`), which use the real
// dusk-schnorr crate and none of this crate's FFI, on a machine without the engine.
fn main() {
    println!("cargo:rerun-if-env-changed=DSV_LIB_DIR");
    println!("cargo:rerun-if-env-changed=DSV_NO_LINK");
    if std::env::var("DSV_NO_LINK").map(|v| v == "1").unwrap_or(false) {
        return;
    }
    let dir = std::env::var("DSV_LIB_DIR").unwrap_or_else(|_| "../../schnorr_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dsv");
}
