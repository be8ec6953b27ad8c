//! CPU baseline from the REAL crate: loops `pk.verify(&sig, m)` of dusk-schnorr 0.18 over the very
//! batches `bench.py` times on the GPU, on one thread and on all cores, and prints one JSON object that
//! `bench.py --cpu-baseline-file` ingests as `cpu_baseline.kind = "crate"` ("Preferred:
//! the real crate").  SOURCE ONLY like the rest of this crate: the authoring image has no Rust toolchain;
//! run it on a machine with cargo + crates.io:
//!
//!     python bench.py --dump-inputs /tmp/dsv_inputs            # on the GPU box: writes the files below
//!     DSV_NO_LINK=1 cargo run --release --bin bench_ref -- /tmp/dsv_inputs > /tmp/cpu_baseline.json
//!     python bench.py --cpu-baseline-file /tmp/cpu_baseline.json
//!
//! What is timed: `PublicKey::verify` (/root/reference/src/keys/public.rs:121-130),
//! `PublicKeyDouble::verify` (:222-244), `PublicKeyVarGen::verify` (:401-415) in the shape of the
//! reference's own bench loop (/root/reference/benches/signature.rs:48-60: one key, one signature, one
//! message per item).  Deserialisation is NOT timed (reported beside it): the objects are built once from
//! the wire records — `Signature::new` is `pub(crate)`, so `from_bytes` is the only public constructor.
//!
//! Input directory (everything little-endian, array-of-records, as `to_bytes()` lays it out):
//!   meta.json                      {"single": n, "double": n, "vargen": n}   (informational)
//!   single_sig.bin   n x 64 B      Signature::to_bytes()        = u || compressed R
//!   single_pk.bin    n x 32 B      PublicKey::to_bytes()
//!   single_m.bin     n x 32 B      BlsScalar::to_bytes()
//!   single_expected.bin  n x 1 B   the verdicts the GPU engine returned (1 = true)
//!   double_sig.bin   n x 96 B      SignatureDouble::to_bytes()  = u || R || R'
//!   double_pk.bin    n x 64 B      PublicKeyDouble::to_bytes()  = pk || pk'
//!   double_m.bin, double_expected.bin
//!   vargen_sig.bin   n x 64 B      SignatureVarGen::to_bytes()
//!   vargen_pk.bin    n x 64 B      PublicKeyVarGen::to_bytes()  = pk || generator
//!   vargen_m.bin, vargen_expected.bin
//! An item whose records do not deserialise (`from_bytes` = Err: the harness's tamper classes include
//! such encodings) counts as verdict `false`, which is what the engine's wire entry points return too.
//!
//! Output: {"kind": "crate", "crate": "dusk-schnorr 0.18", "cores": C, "cpu": "...",
//!          "single": {"items", "threads_1": {"items", "seconds", "value"}, "threads_all": {...},
//!                     "decode_seconds", "undecodable", "mismatches_vs_gpu"}, "double": {...}, "vargen": {...}}
//! `mismatches_vs_gpu` must be 0: it is the bit-exactness check of the GPU verdicts against the
//! reference itself on the full batch.
use std::time::Instant;

use dusk_bls12_381::BlsScalar;
use dusk_bytes::Serializable;
use dusk_schnorr::{
    PublicKey, PublicKeyDouble, PublicKeyVarGen, Signature, SignatureDouble, SignatureVarGen,
};

fn read(dir: &str, name: &str) -> Option<Vec<u8>> {
    std::fs::read(format!("{dir}/{name}")).ok()
}

/// `count` items on `threads` threads (contiguous shards, like the engine's multi-GPU split);
/// returns (seconds, verdicts).
fn timed<F: Fn(usize) -> bool + Sync>(count: usize, threads: usize, f: F) -> (f64, Vec<u8>) {
    let mut out = vec![0u8; count];
    let chunk = (count + threads - 1) / threads.max(1);
    let t0 = Instant::now();
    std::thread::scope(|sc| {
        for (t, part) in out.chunks_mut(chunk.max(1)).enumerate() {
            let f = &f;
            sc.spawn(move || {
                for (j, o) in part.iter_mut().enumerate() {
                    *o = f(t * chunk + j) as u8;
                }
            });
        }
    });
    (t0.elapsed().as_secs_f64(), out)
}

struct Scheme {
    name: &'static str,
    items: usize,
    undecodable: usize,
    decode_seconds: f64,
    one: (usize, f64),
    all: (usize, f64),
    mismatches: usize,
}

/// One scheme: `decode(i)` builds item i's objects (None = some `from_bytes` failed), `verify` is the
/// reference's verify on them.
fn run<T: Send + Sync>(name: &'static str, n: usize, expected: &[u8], cores: usize, one_thread_items: usize,
                       decode: impl Fn(usize) -> Option<T> + Sync, verify: impl Fn(&T) -> bool + Sync) -> Scheme {
    let t0 = Instant::now();
    let mut objs: Vec<Option<T>> = Vec::with_capacity(n);
    // (decoding on all cores: two or three square roots per item)
    let chunk = (n + cores - 1) / cores.max(1);
    let parts: Vec<Vec<Option<T>>> = std::thread::scope(|sc| {
        let hs: Vec<_> = (0..cores)
            .map(|t| {
                let decode = &decode;
                sc.spawn(move || (t * chunk..((t + 1) * chunk).min(n)).map(|i| decode(i)).collect::<Vec<_>>())
            })
            .collect();
        hs.into_iter().map(|h| h.join().unwrap()).collect()
    });
    for p in parts {
        objs.extend(p);
    }
    let decode_seconds = t0.elapsed().as_secs_f64();
    let undecodable = objs.iter().filter(|o| o.is_none()).count();
    let f = |i: usize| objs[i].as_ref().map(|o| verify(o)).unwrap_or(false);
    let k1 = one_thread_items.min(n);
    let (s1, _) = timed(k1, 1, &f);
    let (sa, got) = timed(n, cores, &f);
    let mismatches = got.iter().zip(expected.iter()).filter(|(a, b)| a != b).count();
    Scheme { name, items: n, undecodable, decode_seconds, one: (k1, s1), all: (n, sa), mismatches }
}

fn arr<const N: usize>(buf: &[u8], i: usize) -> [u8; N] {
    buf[N * i..N * (i + 1)].try_into().unwrap()
}

fn main() {
    let dir = std::env::args().nth(1).expect("usage: bench_ref <input dir> [one-thread items]");
    let one_thread_items: usize = std::env::args().nth(2).and_then(|s| s.parse().ok()).unwrap_or(1 << 14);
    let cores = std::thread::available_parallelism().map(|x| x.get()).unwrap_or(1);
    let mut done: Vec<Scheme> = Vec::new();

    if let (Some(sig), Some(pk), Some(m), Some(exp)) = (read(&dir, "single_sig.bin"), read(&dir, "single_pk.bin"),
                                                        read(&dir, "single_m.bin"), read(&dir, "single_expected.bin")) {
        let n = exp.len();
        assert!(sig.len() == 64 * n && pk.len() == 32 * n && m.len() == 32 * n, "single: file sizes");
        done.push(run("single", n, &exp, cores, one_thread_items,
            |i| Some((Signature::from_bytes(&arr::<64>(&sig, i)).ok()?, PublicKey::from_bytes(&arr::<32>(&pk, i)).ok()?,
                      BlsScalar::from_bytes(&arr::<32>(&m, i)).ok()?)),
            |(s, p, m)| p.verify(s, *m)));
    }
    if let (Some(sig), Some(pk), Some(m), Some(exp)) = (read(&dir, "double_sig.bin"), read(&dir, "double_pk.bin"),
                                                        read(&dir, "double_m.bin"), read(&dir, "double_expected.bin")) {
        let n = exp.len();
        assert!(sig.len() == 96 * n && pk.len() == 64 * n && m.len() == 32 * n, "double: file sizes");
        done.push(run("double", n, &exp, cores, one_thread_items,
            |i| Some((SignatureDouble::from_bytes(&arr::<96>(&sig, i)).ok()?,
                      PublicKeyDouble::from_bytes(&arr::<64>(&pk, i)).ok()?, BlsScalar::from_bytes(&arr::<32>(&m, i)).ok()?)),
            |(s, p, m)| p.verify(s, *m)));
    }
    if let (Some(sig), Some(pk), Some(m), Some(exp)) = (read(&dir, "vargen_sig.bin"), read(&dir, "vargen_pk.bin"),
                                                        read(&dir, "vargen_m.bin"), read(&dir, "vargen_expected.bin")) {
        let n = exp.len();
        assert!(sig.len() == 64 * n && pk.len() == 64 * n && m.len() == 32 * n, "vargen: file sizes");
        done.push(run("vargen", n, &exp, cores, one_thread_items,
            |i| Some((SignatureVarGen::from_bytes(&arr::<64>(&sig, i)).ok()?,
                      PublicKeyVarGen::from_bytes(&arr::<64>(&pk, i)).ok()?, BlsScalar::from_bytes(&arr::<32>(&m, i)).ok()?)),
            |(s, p, m)| p.verify(s, *m)));
    }
    assert!(!done.is_empty(), "no input files found in {dir}");

    let cpu = std::fs::read_to_string("/proc/cpuinfo").ok()
        .and_then(|t| t.lines().find(|l| l.starts_with("model name")).map(|l| l.split(':').nth(1).unwrap_or("").trim().to_string()))
        .unwrap_or_default();
    let mut s = format!("{{\"kind\": \"crate\", \"crate\": \"dusk-schnorr 0.18\", \"cores\": {cores}, \"cpu\": \"{}\"",
                        cpu.replace('"', "'"));
    for d in &done {
        s += &format!(
            ", \"{}\": {{\"items\": {}, \"undecodable\": {}, \"decode_seconds\": {:.6}, \"mismatches_vs_gpu\": {}, \
             \"threads_1\": {{\"items\": {}, \"seconds\": {:.6}, \"value\": {:.3}}}, \
             \"threads_all\": {{\"threads\": {cores}, \"items\": {}, \"seconds\": {:.6}, \"value\": {:.3}}}}}",
            d.name, d.items, d.undecodable, d.decode_seconds, d.mismatches,
            d.one.0, d.one.1, d.one.0 as f64 / d.one.1, d.all.0, d.all.1, d.all.0 as f64 / d.all.1);
    }
    s += "}";
    println!("{s}");
    if done.iter().any(|d| d.mismatches != 0) {
        eprintln!("bench_ref: GPU verdicts differ from the reference's");
        std::process::exit(1);
    }
}
