//! GPU batch verification for `dusk-schnorr` (never compiled in the authoring image: no Rust
//! toolchain there).  The reference's `SecretKey` / `PublicKey` / `Signature` types and their
//! `verify()` stay as they are; this crate adds
//! `verify_batch(&[Signature], &[PublicKey], &[BlsScalar]) -> Vec<bool>` (+ `_double`, `_var_gen`)
//! with `out[i] == pks[i].verify(&sigs[i], msgs[i])`, computed by libdsv.so on an MI355X.
//!
//! Replaces, batch-wise: `PublicKey::verify` (dusk-schnorr src/keys/public.rs:121-130),
//! `PublicKeyDouble::verify` (:222-244), `PublicKeyVarGen::verify` (:401-415).
use core::ffi::{c_char, c_int, CStr};

use dusk_bls12_381::BlsScalar;
use dusk_bytes::Serializable;
use dusk_jubjub::JubJubExtended;
use dusk_schnorr::{
    PublicKey, PublicKeyDouble, PublicKeyVarGen, Signature, SignatureDouble, SignatureVarGen,
};

#[allow(non_snake_case)]
extern "C" {
    fn dsv_init(device: c_int) -> c_int;
    fn dsv_init_visible() -> c_int;
    fn dsv_last_error() -> *const c_char;
    // Points go over as (u, v, z), 96 bytes: the coordinates the JubJubExtended values hold, with
    // NO normalisation on the host — the device performs `to_hash_inputs` (one field inversion per
    // signature at most).  The *_multi entry points shard one host batch over EVERY initialised
    // device (contiguous shards, one host thread per device, no collective).
    fn dsv_verify_single_ext_multi(u: *const u8, R_uvz: *const u8, PK_uvz: *const u8, m: *const u8,
                                   n: usize, ok: *mut u8) -> c_int;
    fn dsv_verify_double_ext_multi(u: *const u8, R_uvz: *const u8, Rp_uvz: *const u8,
                                   PK_uvz: *const u8, PKp_uvz: *const u8, m: *const u8, n: usize,
                                   ok: *mut u8) -> c_int;
    fn dsv_verify_vargen_ext_multi(u: *const u8, R_uvz: *const u8, PK_uvz: *const u8,
                                   Gen_uvz: *const u8, m: *const u8, n: usize, ok: *mut u8) -> c_int;
}

/// Engine failure (no GPU, HIP error).  Never a verdict.
#[derive(Debug)]
pub struct EngineError(pub i32, pub String);

fn check(rc: c_int) -> Result<(), EngineError> {
    if rc == 0 {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(dsv_last_error()) }.to_string_lossy().into_owned();
    Err(EngineError(rc, msg))
}

/// Create the context of one GPU: fixed-base tables, streams (idempotent).
pub fn init(device: i32) -> Result<(), EngineError> {
    check(unsafe { dsv_init(device) })
}

/// Initialise the GPUs this process should use: the ordinals in the environment variable
/// `DSV_DEVICES` ("0,2,3"), else every visible one (8 on an MI355X node); `verify_batch*` then
/// shard over all of them.  Idempotent and cheap after the first call; the first call builds two
/// 75.5 MB window tables per device (~50 ms of device time each).
pub fn init_all() -> Result<usize, EngineError> {
    let n = unsafe { dsv_init_visible() };
    if n < 0 {
        check(n)?;
    }
    Ok(n as usize)
}

fn push_point(dst: &mut Vec<u8>, p: &JubJubExtended) {
    // three `to_bytes()` (a Montgomery reduction each, ~50 ns): no inversion, no multiplication.
    // The reference's verify would start with `p.to_hash_inputs()` here (src/signatures.rs:131,
    // :280-281) — that step now runs on the device.
    dst.extend_from_slice(&p.get_u().to_bytes());
    dst.extend_from_slice(&p.get_v().to_bytes());
    dst.extend_from_slice(&p.get_z().to_bytes());
}

fn verdicts(ok: Vec<u8>) -> Vec<bool> {
    ok.into_iter().map(|b| b == 1).collect()
}

pub fn verify_batch(sigs: &[Signature], pks: &[PublicKey], msgs: &[BlsScalar])
    -> Result<Vec<bool>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    let (mut u, mut r, mut pk, mut m) =
        (Vec::with_capacity(32 * n), Vec::with_capacity(96 * n), Vec::with_capacity(96 * n),
         Vec::with_capacity(32 * n));
    for i in 0..n {
        u.extend_from_slice(&sigs[i].u().to_bytes());
        push_point(&mut r, sigs[i].R());
        push_point(&mut pk, pks[i].as_ref());
        m.extend_from_slice(&msgs[i].to_bytes());
    }
    let mut ok = vec![0u8; n];
    init_all()?;
    check(unsafe {
        dsv_verify_single_ext_multi(u.as_ptr(), r.as_ptr(), pk.as_ptr(), m.as_ptr(), n, ok.as_mut_ptr())
    })?;
    Ok(verdicts(ok))
}

pub fn verify_batch_double(sigs: &[SignatureDouble], pks: &[PublicKeyDouble], msgs: &[BlsScalar])
    -> Result<Vec<bool>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    let (mut u, mut r, mut rp, mut pk, mut pkp, mut m) =
        (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for i in 0..n {
        u.extend_from_slice(&sigs[i].u().to_bytes());
        push_point(&mut r, sigs[i].R());
        push_point(&mut rp, sigs[i].R_prime());
        push_point(&mut pk, pks[i].pk());
        push_point(&mut pkp, pks[i].pk_prime());
        m.extend_from_slice(&msgs[i].to_bytes());
    }
    let mut ok = vec![0u8; n];
    init_all()?;
    check(unsafe {
        dsv_verify_double_ext_multi(u.as_ptr(), r.as_ptr(), rp.as_ptr(), pk.as_ptr(), pkp.as_ptr(),
                                m.as_ptr(), n, ok.as_mut_ptr())
    })?;
    Ok(verdicts(ok))
}

pub fn verify_batch_var_gen(sigs: &[SignatureVarGen], pks: &[PublicKeyVarGen], msgs: &[BlsScalar])
    -> Result<Vec<bool>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    let (mut u, mut r, mut pk, mut g, mut m) =
        (Vec::new(), Vec::new(), Vec::new(), Vec::new(), Vec::new());
    for i in 0..n {
        u.extend_from_slice(&sigs[i].u().to_bytes());
        push_point(&mut r, sigs[i].R());
        push_point(&mut pk, pks[i].public_key());
        push_point(&mut g, pks[i].generator());
        m.extend_from_slice(&msgs[i].to_bytes());
    }
    let mut ok = vec![0u8; n];
    init_all()?;
    check(unsafe {
        dsv_verify_vargen_ext_multi(u.as_ptr(), r.as_ptr(), pk.as_ptr(), g.as_ptr(), m.as_ptr(), n,
                                ok.as_mut_ptr())
    })?;
    Ok(verdicts(ok))
}

#[cfg(test)]
mod tests {
    use super::*;
    use dusk_schnorr::SecretKey;
    use ff::Field;
    use rand::{rngs::StdRng, SeedableRng};

    /// tests/schnorr.rs:14-40 of the reference, batch-wise, CPU and GPU side by side
    #[test]
    fn batch_agrees_with_cpu_verify() {
        let mut rng = StdRng::seed_from_u64(2321);
        let (mut sigs, mut pks, mut msgs) = (vec![], vec![], vec![]);
        for _ in 0..256 {
            let sk = SecretKey::random(&mut rng);
            let m = BlsScalar::random(&mut rng);
            sigs.push(sk.sign(&mut rng, m));
            pks.push(PublicKey::from(&sk));
            msgs.push(m);
        }
        pks.swap(3, 4); // two wrong keys
        let gpu = verify_batch(&sigs, &pks, &msgs).expect("engine");
        for i in 0..sigs.len() {
            assert_eq!(gpu[i], pks[i].verify(&sigs[i], msgs[i]), "item {i}");
        }
        assert!(!gpu[3] && !gpu[4] && gpu[5]);
    }
}
