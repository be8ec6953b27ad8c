//! GPU batch verification for `dusk-schnorr` (never compiled in the authoring image: no Rust
//! toolchain there).  The reference's `SecretKey` / `PublicKey` / `Signature` types and their
//! `verify()` stay as they are; this crate adds
//! `verify_batch(&[Signature], &[PublicKey], &[BlsScalar]) -> Vec<bool>` (+ `_double`, `_var_gen`)
//! with `out[i] == pks[i].verify(&sigs[i], msgs[i])`, computed by libdsv.so on an MI355X.
//! `verify` is infallible in the reference, so `verify_batch*` return the verdicts directly and PANIC
//! if the engine itself fails (no GPU, HIP error) — `try_verify_batch*` return that as
//! `Err(EngineError)` instead.  `verify_batch*_submit` start a batch and return a `BatchJob`; with
//! two batches in flight per GPU the second one's ramp runs under the first one's tail
//! (include/dsv.h: `dsv_verify_*_mont_cols_submit`; bench.py `verify_batch_e2e.streamed`).
//!
//! Replaces, batch-wise: `PublicKey::verify` (dusk-schnorr src/keys/public.rs:121-130),
//! `PublicKeyDouble::verify` (:222-244), `PublicKeyVarGen::verify` (:401-415).
//!
//! **No field arithmetic on the host.**  The reference's types hold every field element as
//! `[u64; 4]` Montgomery limbs (R = 2^256): `BlsScalar(pub [u64; 4])`, `JubJubScalar`, the five
//! coordinates of `JubJubExtended`.  A `to_bytes()` is a Montgomery reduction (8 per single
//! signature, 14 per double one — ~0.4 us per signature on one thread, ~30x below the engine), so
//! this crate never calls it: `dsv_verify_*_mont_cols` take the limbs where they lie, one strided
//! column per field — base = the address the accessor of item 0 returns, stride =
//! `size_of::<Signature>()` etc. — and the engine's copy threads gather them into pinned staging
//! while the GPU works on the previous chunk.  The device performs the two scalar reductions and
//! `to_hash_inputs`.  Measured on the C++ mirror of these types (include/dusk_schnorr.hpp,
//! bench.py `verify_batch_e2e`): the typed-object call runs at the rate of the byte-array host
//! path.
//!
//! The only thing this relies on beyond the public API is the in-memory layout of two upstream
//! types, checked once at RUN time by `layout_ok()` (sizes and field order, through a byte view)
//! against values computed through the public API:
//!   * `JubJubScalar` is its four limbs (a one-field tuple struct),
//!   * `JubJubExtended` starts with the limbs of u, v, z in this order (`{u, v, z, t1, t2}`).
//! If the probe fails (a future dusk-jubjub reorders fields) the crate falls back to copying the
//! limbs through the accessors (`get_u().0` ...: field reads, no arithmetic) into dense arrays on
//! several threads and passes those as columns; the limbs of `u`, which has no public limb
//! accessor, are then obtained as `(u * R).to_bytes()` — two host multiplications per signature,
//! against the 8-14 reductions a byte-oriented binding pays.
use core::ffi::{c_char, c_int, c_void, CStr};
use std::sync::OnceLock;

use dusk_bls12_381::BlsScalar;
use dusk_jubjub::{JubJubExtended, JubJubScalar, GENERATOR_EXTENDED};
use dusk_schnorr::{
    PublicKey, PublicKeyDouble, PublicKeyVarGen, Signature, SignatureDouble, SignatureVarGen,
};

/// `dsv_column` of include/dsv.h: one field of n typed objects, item i at `base + i * stride`.
#[repr(C)]
#[derive(Clone, Copy)]
struct Column {
    base: *const c_void,
    stride: usize,
}

#[allow(non_snake_case)]
extern "C" {
    fn dsv_init(device: c_int) -> c_int;
    fn dsv_init_visible() -> c_int;
    fn dsv_last_error() -> *const c_char;
    // columns: single u, R, PK, m | double u, R, R', PK, PK', m | vargen u, R, PK, Gen, m
    // (u, m: 32 B of limbs; points: 96 B = limbs of u || v || z).  Sharded over every initialised
    // device (contiguous shards, one host thread per device, no collective).
    fn dsv_verify_single_mont_cols(cols: *const Column, n: usize, ok: *mut u8) -> c_int;
    fn dsv_verify_double_mont_cols(cols: *const Column, n: usize, ok: *mut u8) -> c_int;
    fn dsv_verify_vargen_mont_cols(cols: *const Column, n: usize, ok: *mut u8) -> c_int;
    // asynchronous forms: `submit` copies cols[], starts the batch on a library-owned driver thread and
    // returns a job; `dsv_job_wait` blocks, returns the batch's status and releases the job
    fn dsv_verify_single_mont_cols_submit(cols: *const Column, n: usize, ok: *mut u8, job: *mut *mut c_void) -> c_int;
    fn dsv_verify_double_mont_cols_submit(cols: *const Column, n: usize, ok: *mut u8, job: *mut *mut c_void) -> c_int;
    fn dsv_verify_vargen_mont_cols_submit(cols: *const Column, n: usize, ok: *mut u8, job: *mut *mut c_void) -> c_int;
    fn dsv_job_wait(job: *mut c_void) -> c_int;
    // batch fast accept (include/dsv.h): same verdicts, one aggregate decides "all true"
    fn dsv_verify_single_mont_cols_rlc(cols: *const Column, n: usize, ok: *mut u8, accepted: *mut c_int) -> c_int;
    fn dsv_verify_double_mont_cols_rlc(cols: *const Column, n: usize, ok: *mut u8, accepted: *mut c_int) -> c_int;
    fn dsv_verify_vargen_mont_cols_rlc(cols: *const Column, n: usize, ok: *mut u8, accepted: *mut c_int) -> c_int;
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

/// The layout facts the zero-copy path rests on, checked AT RUN TIME against values computed through
/// the public API: a dusk-jubjub that changes a size or reorders fields makes this return false and
/// selects the accessor-copy path below — it never breaks the build and never reads past an object
/// (the former compile-time size asserts + `transmute` turned a layout change into a hard
/// build failure, not a fallback).
fn layout_ok() -> bool {
    static OK: OnceLock<bool> = OnceLock::new();
    *OK.get_or_init(|| {
        use core::mem::size_of;
        if size_of::<JubJubScalar>() != 32 || size_of::<BlsScalar>() != 32 || size_of::<JubJubExtended>() != 160 {
            return false;
        }
        // limb k of a value, read through a byte view (no alignment or size assumption beyond the checks above)
        fn limbs<T, const K: usize>(v: &T) -> [u64; K] {
            assert!(size_of::<T>() >= 8 * K);
            let p = v as *const T as *const u8;
            core::array::from_fn(|k| unsafe { core::ptr::read_unaligned(p.add(8 * k) as *const u64) })
        }
        // JubJubScalar::one() must read as R mod r
        let r_mod_r = [0x25f8_0bb3_b996_07d9u64, 0xf315_d62f_66b6_e750, 0x9325_14ee_eb88_14f4,
                       0x09a6_fc6f_4791_55c6];
        let one: [u64; 4] = limbs(&JubJubScalar::one());
        // a point with three distinct, non-trivial coordinates
        let p = GENERATOR_EXTENDED * JubJubScalar::from(7u64);
        let raw: [u64; 12] = limbs(&p);
        // the three schemes' structs: the accessors must point INTO the objects handed to the engine
        one == r_mod_r
            && raw[0..4] == p.get_u().0
            && raw[4..8] == p.get_v().0
            && raw[8..12] == p.get_z().0
    })
}

fn col<T, F>(first: &F, _outer: &[T]) -> Column {
    Column { base: first as *const F as *const c_void, stride: core::mem::size_of::<T>() }
}

fn verdicts(ok: Vec<u8>) -> Vec<bool> {
    ok.into_iter().map(|b| b == 1).collect()
}

/// `out[i] == pks[i].verify(&sigs[i], msgs[i])` (dusk-schnorr src/keys/public.rs:121-130) — the entry
/// point `north_star` names.  Panics if the slices differ in length or the engine fails.
pub fn verify_batch(sigs: &[Signature], pks: &[PublicKey], msgs: &[BlsScalar]) -> Vec<bool> {
    try_verify_batch(sigs, pks, msgs).expect("dusk-schnorr-gpu: engine failure")
}
/// ... `PublicKeyDouble::verify` (src/keys/public.rs:222-244)
pub fn verify_batch_double(sigs: &[SignatureDouble], pks: &[PublicKeyDouble], msgs: &[BlsScalar]) -> Vec<bool> {
    try_verify_batch_double(sigs, pks, msgs).expect("dusk-schnorr-gpu: engine failure")
}
/// ... `PublicKeyVarGen::verify` (src/keys/public.rs:401-415)
pub fn verify_batch_var_gen(sigs: &[SignatureVarGen], pks: &[PublicKeyVarGen], msgs: &[BlsScalar]) -> Vec<bool> {
    try_verify_batch_var_gen(sigs, pks, msgs).expect("dusk-schnorr-gpu: engine failure")
}

pub fn try_verify_batch(sigs: &[Signature], pks: &[PublicKey], msgs: &[BlsScalar])
    -> Result<Vec<bool>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return Ok(Vec::new());
    }
    init_all()?;
    let mut ok = vec![0u8; n];
    if layout_ok() {
        // the objects are read in place: no copy, no arithmetic, nothing allocated but `ok`
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].as_ref(), pks),
                    col(&msgs[0], msgs)];
        check(unsafe { dsv_verify_single_mont_cols(cols.as_ptr(), n, ok.as_mut_ptr()) })?;
    } else {
        let soa = fallback::Soa::gather(n, |i| sigs[i].u(), |i| [sigs[i].R(), pks[i].as_ref()], |i| &msgs[i]);
        check(unsafe { dsv_verify_single_mont_cols(soa.cols().as_ptr(), n, ok.as_mut_ptr()) })?;
    }
    Ok(verdicts(ok))
}

pub fn try_verify_batch_double(sigs: &[SignatureDouble], pks: &[PublicKeyDouble], msgs: &[BlsScalar])
    -> Result<Vec<bool>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return Ok(Vec::new());
    }
    init_all()?;
    let mut ok = vec![0u8; n];
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(sigs[0].R_prime(), sigs),
                    col(pks[0].pk(), pks), col(pks[0].pk_prime(), pks), col(&msgs[0], msgs)];
        check(unsafe { dsv_verify_double_mont_cols(cols.as_ptr(), n, ok.as_mut_ptr()) })?;
    } else {
        let soa = fallback::Soa::gather(n, |i| sigs[i].u(),
            |i| [sigs[i].R(), sigs[i].R_prime(), pks[i].pk(), pks[i].pk_prime()], |i| &msgs[i]);
        check(unsafe { dsv_verify_double_mont_cols(soa.cols().as_ptr(), n, ok.as_mut_ptr()) })?;
    }
    Ok(verdicts(ok))
}

pub fn try_verify_batch_var_gen(sigs: &[SignatureVarGen], pks: &[PublicKeyVarGen], msgs: &[BlsScalar])
    -> Result<Vec<bool>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return Ok(Vec::new());
    }
    init_all()?;
    let mut ok = vec![0u8; n];
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].public_key(), pks),
                    col(pks[0].generator(), pks), col(&msgs[0], msgs)];
        check(unsafe { dsv_verify_vargen_mont_cols(cols.as_ptr(), n, ok.as_mut_ptr()) })?;
    } else {
        let soa = fallback::Soa::gather(n, |i| sigs[i].u(),
            |i| [sigs[i].R(), pks[i].public_key(), pks[i].generator()], |i| &msgs[i]);
        check(unsafe { dsv_verify_vargen_mont_cols(soa.cols().as_ptr(), n, ok.as_mut_ptr()) })?;
    }
    Ok(verdicts(ok))
}

/// For batches expected to be ENTIRELY valid (a block's signatures): the same verdicts as
/// [`verify_batch`], through one random-linear-combination aggregate over the whole batch (<= 2^22
/// items) when every item is valid — about twice as fast — and through the per-signature kernels
/// otherwise (a batch that fails has paid for both).  The second value says whether the aggregate
/// decided.  The aggregate also proves every key and nonce point to be of prime order, so it is exact
/// on the reference's cofactorless equation; its error probability (accepting a batch that holds a
/// wrong signature) is <= 2^-112, the weights come from `getrandom` per call.
pub fn verify_batch_fast(sigs: &[Signature], pks: &[PublicKey], msgs: &[BlsScalar]) -> (Vec<bool>, bool) {
    try_verify_batch_fast(sigs, pks, msgs).expect("dusk-schnorr-gpu: engine failure")
}
pub fn try_verify_batch_fast(sigs: &[Signature], pks: &[PublicKey], msgs: &[BlsScalar])
    -> Result<(Vec<bool>, bool), EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return Ok((Vec::new(), false));
    }
    init_all()?;
    let mut ok = vec![0u8; n];
    let mut accepted: c_int = 0;
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].as_ref(), pks),
                    col(&msgs[0], msgs)];
        check(unsafe { dsv_verify_single_mont_cols_rlc(cols.as_ptr(), n, ok.as_mut_ptr(), &mut accepted) })?;
    } else {
        let soa = fallback::Soa::gather(n, |i| sigs[i].u(), |i| [sigs[i].R(), pks[i].as_ref()], |i| &msgs[i]);
        check(unsafe { dsv_verify_single_mont_cols_rlc(soa.cols().as_ptr(), n, ok.as_mut_ptr(), &mut accepted) })?;
    }
    Ok((verdicts(ok), accepted == 1))
}
/// ... `PublicKeyDouble::verify`
pub fn try_verify_batch_double_fast(sigs: &[SignatureDouble], pks: &[PublicKeyDouble], msgs: &[BlsScalar])
    -> Result<(Vec<bool>, bool), EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return Ok((Vec::new(), false));
    }
    init_all()?;
    let mut ok = vec![0u8; n];
    let mut accepted: c_int = 0;
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(sigs[0].R_prime(), sigs),
                    col(pks[0].pk(), pks), col(pks[0].pk_prime(), pks), col(&msgs[0], msgs)];
        check(unsafe { dsv_verify_double_mont_cols_rlc(cols.as_ptr(), n, ok.as_mut_ptr(), &mut accepted) })?;
    } else {
        let soa = fallback::Soa::gather(n, |i| sigs[i].u(),
            |i| [sigs[i].R(), sigs[i].R_prime(), pks[i].pk(), pks[i].pk_prime()], |i| &msgs[i]);
        check(unsafe { dsv_verify_double_mont_cols_rlc(soa.cols().as_ptr(), n, ok.as_mut_ptr(), &mut accepted) })?;
    }
    Ok((verdicts(ok), accepted == 1))
}
/// ... `PublicKeyVarGen::verify`
pub fn try_verify_batch_var_gen_fast(sigs: &[SignatureVarGen], pks: &[PublicKeyVarGen], msgs: &[BlsScalar])
    -> Result<(Vec<bool>, bool), EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return Ok((Vec::new(), false));
    }
    init_all()?;
    let mut ok = vec![0u8; n];
    let mut accepted: c_int = 0;
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].public_key(), pks),
                    col(pks[0].generator(), pks), col(&msgs[0], msgs)];
        check(unsafe { dsv_verify_vargen_mont_cols_rlc(cols.as_ptr(), n, ok.as_mut_ptr(), &mut accepted) })?;
    } else {
        let soa = fallback::Soa::gather(n, |i| sigs[i].u(),
            |i| [sigs[i].R(), pks[i].public_key(), pks[i].generator()], |i| &msgs[i]);
        check(unsafe { dsv_verify_vargen_mont_cols_rlc(soa.cols().as_ptr(), n, ok.as_mut_ptr(), &mut accepted) })?;
    }
    Ok((verdicts(ok), accepted == 1))
}

/// A batch in flight.  Borrows the slices it was started from (the engine reads the objects in place
/// until `wait` returns); dropping it unwaited waits.
///
/// SOUNDNESS.  The engine's driver thread reads the borrowed objects until the job has been waited for,
/// and the only thing that makes it wait is `wait()` or `Drop`.  `core::mem::forget(job)` (or a leak
/// through an `Rc` cycle) ends the borrow without either — safe code could then free the slices under
/// the engine: the pre-1.0 `thread::scoped` hole.  The functions that hand out a `BatchJob` are therefore
/// `unsafe fn` (contract: the job is waited for or dropped, never leaked); the safe form of the same
/// thing is [`batch_scope`], which joins every job before it returns, as `std::thread::scope` does.
pub struct BatchJob<'a> {
    job: *mut c_void,
    ok: Vec<u8>,
    _soa: Option<Box<dyn core::any::Any>>, // the dense copies of the layout-agnostic path, if taken (owned arrays)
    _borrow: core::marker::PhantomData<&'a ()>,
}
impl<'a> BatchJob<'a> {
    /// Blocks until the verdicts are there: `out[i] == pks[i].verify(&sigs[i], msgs[i])`.
    pub fn wait(mut self) -> Result<Vec<bool>, EngineError> {
        let job = core::mem::replace(&mut self.job, core::ptr::null_mut());
        if !job.is_null() {
            check(unsafe { dsv_job_wait(job) })?;
        }
        Ok(verdicts(core::mem::take(&mut self.ok)))
    }
}
impl<'a> Drop for BatchJob<'a> {
    fn drop(&mut self) {
        if !self.job.is_null() {
            unsafe { dsv_job_wait(self.job) };
        }
    }
}
type SubmitFn = unsafe extern "C" fn(*const Column, usize, *mut u8, *mut *mut c_void) -> c_int;
fn submit<'a>(f: SubmitFn, cols: &[Column], n: usize, soa: Option<Box<dyn core::any::Any>>)
    -> Result<BatchJob<'a>, EngineError> {
    let mut j = BatchJob { job: core::ptr::null_mut(), ok: vec![0u8; n], _soa: soa, _borrow: core::marker::PhantomData };
    if n != 0 {
        init_all()?;
        check(unsafe { f(cols.as_ptr(), n, j.ok.as_mut_ptr(), &mut j.job) })?;
    }
    Ok(j)
}
/// Start `verify_batch` and return at once.  Keep two in flight per GPU:
/// `let a = submit(b0); let b = submit(b1); a.wait(); let c = submit(b2); b.wait(); ...`
///
/// # Safety
/// The returned job must be waited for or dropped before `sigs`, `pks` and `msgs` go away; it must never
/// be leaked (`mem::forget`, an `Rc` cycle): the engine reads the slices in place until then.  Safe
/// alternative: [`batch_scope`].  (The same holds for the `_double` and `_var_gen` forms.)
pub unsafe fn verify_batch_submit<'a>(sigs: &'a [Signature], pks: &'a [PublicKey], msgs: &'a [BlsScalar])
    -> Result<BatchJob<'a>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return submit(dsv_verify_single_mont_cols_submit, &[], 0, None);
    }
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].as_ref(), pks), col(&msgs[0], msgs)];
        submit(dsv_verify_single_mont_cols_submit, &cols, n, None)
    } else {
        let soa = Box::new(fallback::Soa::gather(n, |i| sigs[i].u(), |i| [sigs[i].R(), pks[i].as_ref()], |i| &msgs[i]));
        let cols = soa.cols(); // (pointers into the boxed arrays: they do not move with the box)
        submit(dsv_verify_single_mont_cols_submit, &cols, n, Some(soa as Box<dyn core::any::Any>))
    }
}
pub unsafe fn verify_batch_double_submit<'a>(sigs: &'a [SignatureDouble], pks: &'a [PublicKeyDouble], msgs: &'a [BlsScalar])
    -> Result<BatchJob<'a>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return submit(dsv_verify_double_mont_cols_submit, &[], 0, None);
    }
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(sigs[0].R_prime(), sigs),
                    col(pks[0].pk(), pks), col(pks[0].pk_prime(), pks), col(&msgs[0], msgs)];
        submit(dsv_verify_double_mont_cols_submit, &cols, n, None)
    } else {
        let soa = Box::new(fallback::Soa::gather(n, |i| sigs[i].u(),
            |i| [sigs[i].R(), sigs[i].R_prime(), pks[i].pk(), pks[i].pk_prime()], |i| &msgs[i]));
        let cols = soa.cols();
        submit(dsv_verify_double_mont_cols_submit, &cols, n, Some(soa as Box<dyn core::any::Any>))
    }
}
pub unsafe fn verify_batch_var_gen_submit<'a>(sigs: &'a [SignatureVarGen], pks: &'a [PublicKeyVarGen], msgs: &'a [BlsScalar])
    -> Result<BatchJob<'a>, EngineError> {
    assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
    let n = sigs.len();
    if n == 0 {
        return submit(dsv_verify_vargen_mont_cols_submit, &[], 0, None);
    }
    if layout_ok() {
        let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].public_key(), pks),
                    col(pks[0].generator(), pks), col(&msgs[0], msgs)];
        submit(dsv_verify_vargen_mont_cols_submit, &cols, n, None)
    } else {
        let soa = Box::new(fallback::Soa::gather(n, |i| sigs[i].u(),
            |i| [sigs[i].R(), pks[i].public_key(), pks[i].generator()], |i| &msgs[i]));
        let cols = soa.cols();
        submit(dsv_verify_vargen_mont_cols_submit, &cols, n, Some(soa as Box<dyn core::any::Any>))
    }
}


/// Batches in flight behind a SAFE interface: every job started through the scope is joined before
/// `batch_scope` returns (also when the closure panics), so leaking a ticket leaks nothing the engine
/// still uses — the verdict buffers and, on the layout-agnostic path, the dense copies belong to the scope.
/// ```ignore
/// let verdicts = batch_scope(|s| {
///     let a = s.verify_batch(&sigs0, &pks0, &msgs0)?;      // two in flight per GPU
///     let b = s.verify_batch(&sigs1, &pks1, &msgs1)?;
///     Ok::<_, EngineError>((s.wait(a)?, s.wait(b)?))
/// })?;
/// ```
pub struct BatchScope<'env> {
    slots: core::cell::RefCell<Vec<ScopedSlot>>,
    _env: core::marker::PhantomData<&'env ()>,
}
struct ScopedSlot {
    job: *mut c_void,
    ok: Box<[u8]>,                          // (its heap block does not move when `slots` grows)
    _soa: Option<Box<dyn core::any::Any>>,  // the dense copies of the layout-agnostic path, if taken
}
/// A ticket for one job of a [`BatchScope`]; redeem it with [`BatchScope::wait`].
pub struct ScopedJob(usize);
impl<'env> BatchScope<'env> {
    fn start(&self, f: SubmitFn, cols: &[Column], n: usize, soa: Option<Box<dyn core::any::Any>>)
        -> Result<ScopedJob, EngineError> {
        let mut slot = ScopedSlot { job: core::ptr::null_mut(), ok: vec![0u8; n].into_boxed_slice(), _soa: soa };
        if n != 0 {
            init_all()?;
            check(unsafe { f(cols.as_ptr(), n, slot.ok.as_mut_ptr(), &mut slot.job) })?;
        }
        let mut slots = self.slots.borrow_mut();
        slots.push(slot);
        Ok(ScopedJob(slots.len() - 1))
    }
    /// `verify_batch`, started now; the slices must outlive the scope (`'env`)
    pub fn verify_batch(&self, sigs: &'env [Signature], pks: &'env [PublicKey], msgs: &'env [BlsScalar])
        -> Result<ScopedJob, EngineError> {
        assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
        let n = sigs.len();
        if n == 0 {
            return self.start(dsv_verify_single_mont_cols_submit, &[], 0, None);
        }
        if layout_ok() {
            let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].as_ref(), pks), col(&msgs[0], msgs)];
            self.start(dsv_verify_single_mont_cols_submit, &cols, n, None)
        } else {
            let soa = Box::new(fallback::Soa::gather(n, |i| sigs[i].u(), |i| [sigs[i].R(), pks[i].as_ref()], |i| &msgs[i]));
            let cols = soa.cols();
            self.start(dsv_verify_single_mont_cols_submit, &cols, n, Some(soa as Box<dyn core::any::Any>))
        }
    }
    /// `verify_batch_double`, started now
    pub fn verify_batch_double(&self, sigs: &'env [SignatureDouble], pks: &'env [PublicKeyDouble], msgs: &'env [BlsScalar])
        -> Result<ScopedJob, EngineError> {
        assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
        let n = sigs.len();
        if n == 0 {
            return self.start(dsv_verify_double_mont_cols_submit, &[], 0, None);
        }
        if layout_ok() {
            let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(sigs[0].R_prime(), sigs),
                        col(pks[0].pk(), pks), col(pks[0].pk_prime(), pks), col(&msgs[0], msgs)];
            self.start(dsv_verify_double_mont_cols_submit, &cols, n, None)
        } else {
            let soa = Box::new(fallback::Soa::gather(n, |i| sigs[i].u(),
                |i| [sigs[i].R(), sigs[i].R_prime(), pks[i].pk(), pks[i].pk_prime()], |i| &msgs[i]));
            let cols = soa.cols();
            self.start(dsv_verify_double_mont_cols_submit, &cols, n, Some(soa as Box<dyn core::any::Any>))
        }
    }
    /// `verify_batch_var_gen`, started now
    pub fn verify_batch_var_gen(&self, sigs: &'env [SignatureVarGen], pks: &'env [PublicKeyVarGen], msgs: &'env [BlsScalar])
        -> Result<ScopedJob, EngineError> {
        assert!(sigs.len() == pks.len() && sigs.len() == msgs.len());
        let n = sigs.len();
        if n == 0 {
            return self.start(dsv_verify_vargen_mont_cols_submit, &[], 0, None);
        }
        if layout_ok() {
            let cols = [col(sigs[0].u(), sigs), col(sigs[0].R(), sigs), col(pks[0].public_key(), pks),
                        col(pks[0].generator(), pks), col(&msgs[0], msgs)];
            self.start(dsv_verify_vargen_mont_cols_submit, &cols, n, None)
        } else {
            let soa = Box::new(fallback::Soa::gather(n, |i| sigs[i].u(),
                |i| [sigs[i].R(), pks[i].public_key(), pks[i].generator()], |i| &msgs[i]));
            let cols = soa.cols();
            self.start(dsv_verify_vargen_mont_cols_submit, &cols, n, Some(soa as Box<dyn core::any::Any>))
        }
    }
    /// Blocks until the job's verdicts are there.
    pub fn wait(&self, job: ScopedJob) -> Result<Vec<bool>, EngineError> {
        let handle = {
            let mut slots = self.slots.borrow_mut();
            core::mem::replace(&mut slots[job.0].job, core::ptr::null_mut())
        };
        if !handle.is_null() {
            check(unsafe { dsv_job_wait(handle) })?;
        }
        // (a null handle: an empty batch — tickets are neither Copy nor Clone, so none is redeemed twice)
        let slots = self.slots.borrow();
        Ok(verdicts(slots[job.0].ok.to_vec()))
    }
}
impl<'env> Drop for BatchScope<'env> {
    fn drop(&mut self) {
        for slot in self.slots.get_mut().iter_mut() {
            if !slot.job.is_null() {
                unsafe { dsv_job_wait(slot.job) };  // joins the engine's driver thread: nothing reads 'env data after this
                slot.job = core::ptr::null_mut();
            }
        }
    }
}
/// Runs `f` with a scope for batches in flight and joins every job it started before returning — on the
/// normal path and on unwind (the scope is a local of this function: `f` only ever sees a reference to it).
pub fn batch_scope<'env, T>(f: impl FnOnce(&BatchScope<'env>) -> T) -> T {
    let scope = BatchScope { slots: core::cell::RefCell::new(Vec::new()), _env: core::marker::PhantomData };
    f(&scope)
    // (`scope` is dropped here, or during unwinding: Drop waits for whatever is still in flight)
}

/// Layout-agnostic path: limb COPIES through the public accessors (`get_u().0`: a field read, no
/// arithmetic) into dense arrays, in parallel chunks, then the same entry points with dense columns.
mod fallback {
    use super::*;

    pub struct Soa<const NP: usize> {
        u: Vec<[u64; 4]>,
        pts: [Vec<[u64; 12]>; NP],
        m: Vec<[u64; 4]>,
    }

    /// Montgomery limbs of `u` through the public API only: the canonical bytes of `u * R` (R = 2^256
    /// mod r as a scalar) ARE the limbs of `u`.  One multiplication + one reduction on the host.
    fn scalar_limbs(u: &JubJubScalar) -> [u64; 4] {
        let r_mod_r = JubJubScalar::from_raw([0x25f8_0bb3_b996_07d9, 0xf315_d62f_66b6_e750,
                                              0x9325_14ee_eb88_14f4, 0x09a6_fc6f_4791_55c6]);
        let b = (u * r_mod_r).to_bytes();
        let mut l = [0u64; 4];
        for k in 0..4 {
            l[k] = u64::from_le_bytes(b[8 * k..8 * k + 8].try_into().unwrap());
        }
        l
    }

    impl<const NP: usize> Soa<NP> {
        pub fn gather<'a>(n: usize, u: impl Fn(usize) -> &'a JubJubScalar + Sync,
                          pts: impl Fn(usize) -> [&'a JubJubExtended; NP] + Sync,
                          m: impl Fn(usize) -> &'a BlsScalar + Sync) -> Self {
            let mut s = Soa { u: vec![[0; 4]; n], pts: core::array::from_fn(|_| vec![[0; 12]; n]),
                              m: vec![[0; 4]; n] };
            let threads = std::thread::available_parallelism().map(|x| x.get()).unwrap_or(4).min(16);
            let chunk = (n + threads - 1) / threads;
            std::thread::scope(|sc| {
                let mut us = s.u.chunks_mut(chunk);
                let mut ms = s.m.chunks_mut(chunk);
                let mut ps: Vec<_> = s.pts.iter_mut().map(|v| v.chunks_mut(chunk)).collect();
                for t in 0..threads {
                    let (Some(uc), Some(mc)) = (us.next(), ms.next()) else { break };
                    let mut pc: Vec<_> = ps.iter_mut().map(|it| it.next().unwrap()).collect();
                    let (u, pts, m) = (&u, &pts, &m);
                    sc.spawn(move || {
                        for (j, i) in (t * chunk..(t * chunk + uc.len())).enumerate() {
                            uc[j] = scalar_limbs(u(i));
                            mc[j] = m(i).0;
                            for (k, p) in pts(i).iter().enumerate() {
                                pc[k][j][0..4].copy_from_slice(&p.get_u().0);
                                pc[k][j][4..8].copy_from_slice(&p.get_v().0);
                                pc[k][j][8..12].copy_from_slice(&p.get_z().0);
                            }
                        }
                    });
                }
            });
            s
        }

        pub fn cols(&self) -> Vec<Column> {
            let mut c = vec![Column { base: self.u.as_ptr() as *const c_void, stride: 32 }];
            for p in &self.pts {
                c.push(Column { base: p.as_ptr() as *const c_void, stride: 96 });
            }
            c.push(Column { base: self.m.as_ptr() as *const c_void, stride: 32 });
            c
        }
    }
}

#[cfg(test)]
mod tests {
    use super::*;
    use dusk_schnorr::SecretKey;
    use ff::Field;
    use rand::{rngs::StdRng, SeedableRng};

    #[test]
    fn upstream_layout_is_what_the_zero_copy_path_assumes() {
        assert!(layout_ok());
    }

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
        let gpu = verify_batch(&sigs, &pks, &msgs);
        assert_eq!(try_verify_batch(&sigs, &pks, &msgs).expect("engine"), gpu);
        // two batches in flight give the same verdicts: through the scope ...
        let (va, vb) = batch_scope(|s| {
            let a = s.verify_batch(&sigs, &pks, &msgs).expect("engine");
            let b = s.verify_batch(&sigs, &pks, &msgs).expect("engine");
            (s.wait(a).expect("engine"), s.wait(b).expect("engine"))
        });
        assert!(va == gpu && vb == gpu);
        // ... a ticket that is never redeemed is still joined when the scope ends
        batch_scope(|s| {
            let t = s.verify_batch(&sigs, &pks, &msgs).expect("engine");
            core::mem::forget(t);
        });
        // ... and through the unscoped jobs (unsafe: the caller promises to wait for them)
        let (a, b) = unsafe {
            (verify_batch_submit(&sigs, &pks, &msgs).expect("engine"),
             verify_batch_submit(&sigs, &pks, &msgs).expect("engine"))
        };
        assert_eq!(a.wait().expect("engine"), gpu);
        assert_eq!(b.wait().expect("engine"), gpu);
        for i in 0..sigs.len() {
            assert_eq!(gpu[i], pks[i].verify(&sigs[i], msgs[i]), "item {i}");
        }
        assert!(!gpu[3] && !gpu[4] && gpu[5]);
        // the fast accept: same verdicts, decided item by item (two wrong keys) ...
        let (fast, accepted) = verify_batch_fast(&sigs, &pks, &msgs);
        assert!(fast == gpu && !accepted);
        // ... also once the keys are back in place: 256 items are below the size from which an aggregate pays
        // (schnorr_amd/csrc/rlc.h: kRlcMinAuto = 2^17; smaller batches take the per-signature kernels and
        // report accepted = false) — the verdicts are what counts
        pks.swap(3, 4);
        let (fast, accepted) = verify_batch_fast(&sigs, &pks, &msgs);
        assert!(!accepted && fast.iter().all(|&b| b));
        pks.swap(3, 4);
        // the layout-agnostic path gives the same verdicts
        let soa = fallback::Soa::gather(sigs.len(), |i| sigs[i].u(),
                                        |i| [sigs[i].R(), pks[i].as_ref()], |i| &msgs[i]);
        let mut ok = vec![0u8; sigs.len()];
        check(unsafe { dsv_verify_single_mont_cols(soa.cols().as_ptr(), sigs.len(), ok.as_mut_ptr()) })
            .expect("engine");
        assert_eq!(verdicts(ok), gpu);
    }
}
