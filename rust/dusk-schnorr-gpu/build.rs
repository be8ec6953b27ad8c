// Link against the engine built by `python -m schnorr_amd.build` (schnorr_amd/libdsv.so).
fn main() {
    let dir = std::env::var("DSV_LIB_DIR").unwrap_or_else(|_| "../../schnorr_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dsv");
    println!("cargo:rerun-if-env-changed=DSV_LIB_DIR");
}
