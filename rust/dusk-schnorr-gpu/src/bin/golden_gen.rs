//! Dumps golden vectors from the REAL dusk-schnorr 0.18 so the repository's restatement of the
//! hash (Hades constants, sponge padding, 250-bit truncation), of the three schemes, of the wire
//! formats and of the test harness's RNG can be pinned — nothing in the reference tree pins any of
//! them.  Run on a machine with cargo + crates.io:
//!     cargo run --features golden --bin golden_gen > ../../tests/golden/reference_vectors.txt
//! and commit that file: tests/reference_fixtures.py parses exactly this line format, and
//! tests/test_oracle.py::test_reference_fixtures_pin_the_oracle (CPU) and
//! tests/test_gpu_parity.py::test_reference_fixtures_on_gpu (GPU) then check the oracle and the
//! HIP engine against it — no code change needed.  tests/golden/predicted_reference.json holds
//! what this repository PREDICTS every line to be (tests/golden/make_predicted.py).
//!
//! Record kinds (one per line; hex = little-endian `to_bytes()`; points as affine u || v):
//!   sponge_hash n <k> le <32>        dusk_poseidon::sponge::hash(&[1, .., k])          k = 3, 4, 5, 8
//!   truncated_hash n <k> le <32>     sponge::truncated::hash of the same inputs        k = 3, 5
//!   sig  <i> sk m u R PK c sig_bytes pk_bytes verdict          tests/schnorr.rs:14-25, seed 2321
//!   sigd <i> sk m u R Rp PK PKp c sig_bytes pk_bytes verdict   tests/schnorr_double.rs:14-25
//!   sigv <i> sk_bytes m u R PK Gen c sig_bytes pk_bytes verdict  tests/schnorr_var_generator.rs:14-25
//!   stdrng <seed> first <n> <bytes>  the first n bytes StdRng::seed_from_u64(seed) hands out
//!   wide fr|fq <64 bytes> <32>       JubJubScalar::random / BlsScalar::random over an "rng" that
//!                                    returns exactly those 64 bytes
//!   from_bytes <i> <32> ok u <32> v <32> | err                  JubJubAffine::from_bytes edge cases
use dusk_bls12_381::BlsScalar;
use dusk_bytes::Serializable;
use dusk_jubjub::JubJubScalar;
use dusk_poseidon::sponge;
use dusk_schnorr::{PublicKey, PublicKeyDouble, PublicKeyVarGen, SecretKey, SecretKeyVarGen};
use ff::Field;
use rand::{rngs::StdRng, CryptoRng, RngCore, SeedableRng};

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}

fn uv(p: &dusk_jubjub::JubJubExtended) -> String {
    let [u, v] = p.to_hash_inputs();
    format!("{}{}", hex(&u.to_bytes()), hex(&v.to_bytes()))
}

/// An "rng" that hands out one fixed buffer: pins `Field::random` = `from_bytes_wide` of 64 bytes.
struct FixedRng(Vec<u8>, usize);
impl RngCore for FixedRng {
    fn next_u32(&mut self) -> u32 {
        let mut b = [0u8; 4];
        self.fill_bytes(&mut b);
        u32::from_le_bytes(b)
    }
    fn next_u64(&mut self) -> u64 {
        let mut b = [0u8; 8];
        self.fill_bytes(&mut b);
        u64::from_le_bytes(b)
    }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        for d in dest.iter_mut() {
            *d = self.0[self.1 % self.0.len()];
            self.1 += 1;
        }
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), rand::Error> {
        self.fill_bytes(dest);
        Ok(())
    }
}
impl CryptoRng for FixedRng {}

fn main() {
    // (a) the sponge at every chunking the schemes use, and beyond: 3 inputs (one short chunk, the
    // single challenge), 4 (exactly the rate), 5 (full chunk + one element: the double challenge,
    // src/signatures.rs:275-290), 8 (two full chunks)
    let ins: Vec<BlsScalar> = (1u64..=8).map(BlsScalar::from).collect();
    for k in [3usize, 4, 5, 8] {
        println!("sponge_hash n {k} le {}", hex(&sponge::hash(&ins[..k]).to_bytes()));
    }
    for k in [3usize, 5] {
        println!("truncated_hash n {k} le {}", hex(&sponge::truncated::hash(&ins[..k]).to_bytes()));
    }
    // (the r03 names of the 3-input records, still parsed)
    println!("sponge_hash_1_2_3_le {}", hex(&sponge::hash(&ins[..3]).to_bytes()));
    println!("truncated_hash_1_2_3_le {}", hex(&sponge::truncated::hash(&ins[..3]).to_bytes()));

    // (b) single signatures under the reference's own test seed (tests/schnorr.rs:16)
    let mut rng = StdRng::seed_from_u64(2321);
    for i in 0..8 {
        let sk = SecretKey::random(&mut rng);
        let m = BlsScalar::random(&mut rng);
        let sig = sk.sign(&mut rng, m);
        let pk = PublicKey::from(&sk);
        let [ru, rv] = sig.R().to_hash_inputs();
        let c = sponge::truncated::hash(&[ru, rv, m]); // = challenge_hash, src/signatures.rs:127-134
        println!(
            "sig {i} sk {} m {} u {} R {} PK {} c {} sig_bytes {} pk_bytes {} verdict {}",
            hex(&sk.to_bytes()), hex(&m.to_bytes()), hex(&sig.u().to_bytes()), uv(sig.R()),
            uv(pk.as_ref()), hex(&c.to_bytes()), hex(&sig.to_bytes()), hex(&pk.to_bytes()),
            pk.verify(&sig, m)
        );
    }
    // (c) double signatures (tests/schnorr_double.rs:14-25; src/keys/secret.rs:217-240)
    let mut rng = StdRng::seed_from_u64(2321);
    for i in 0..8 {
        let sk = SecretKey::random(&mut rng);
        let m = BlsScalar::random(&mut rng);
        let pk = PublicKeyDouble::from(&sk);
        let sig = sk.sign_double(&mut rng, m);
        let [ru, rv] = sig.R().to_hash_inputs();
        let [pu, pv] = sig.R_prime().to_hash_inputs();
        let c = sponge::truncated::hash(&[ru, rv, pu, pv, m]); // challenge_hash_double, :275-290
        println!(
            "sigd {i} sk {} m {} u {} R {} Rp {} PK {} PKp {} c {} sig_bytes {} pk_bytes {} verdict {}",
            hex(&sk.to_bytes()), hex(&m.to_bytes()), hex(&sig.u().to_bytes()), uv(sig.R()),
            uv(sig.R_prime()), uv(pk.pk()), uv(pk.pk_prime()), hex(&c.to_bytes()),
            hex(&sig.to_bytes()), hex(&pk.to_bytes()), pk.verify(&sig, m)
        );
    }
    // (d) var-generator signatures (tests/schnorr_var_generator.rs:14-25): SecretKeyVarGen::random
    // draws sk, then the generator scalar (src/keys/secret.rs:367-376); then message, then the nonce
    let mut rng = StdRng::seed_from_u64(2321);
    for i in 0..8 {
        let sk = SecretKeyVarGen::random(&mut rng);
        let m = BlsScalar::random(&mut rng);
        let pk = PublicKeyVarGen::from(&sk);
        let sig = sk.sign(&mut rng, m);
        let [ru, rv] = sig.R().to_hash_inputs();
        let c = sponge::truncated::hash(&[ru, rv, m]);
        println!(
            "sigv {i} sk_bytes {} m {} u {} R {} PK {} Gen {} c {} sig_bytes {} pk_bytes {} verdict {}",
            hex(&sk.to_bytes()), hex(&m.to_bytes()), hex(&sig.u().to_bytes()), uv(sig.R()),
            uv(pk.public_key()), uv(pk.generator()), hex(&c.to_bytes()), hex(&sig.to_bytes()),
            hex(&pk.to_bytes()), pk.verify(&sig, m)
        );
    }
    // (e) the harness's RNG, raw (rand 0.8 StdRng = ChaCha12, seed_from_u64 = PCG32 expansion)
    let mut rng = StdRng::seed_from_u64(2321);
    let mut raw = [0u8; 256];
    rng.fill_bytes(&mut raw);
    println!("stdrng 2321 first 256 {}", hex(&raw));
    // (f) `Field::random` = from_bytes_wide of 64 rng bytes, for both scalar types
    let wides: [Vec<u8>; 3] = [(0u8..64).collect(), vec![0xff; 64], (0u8..64).map(|x| x.wrapping_mul(37) ^ 0x5a).collect()];
    for w in wides.iter() {
        let fr = JubJubScalar::random(&mut FixedRng(w.clone(), 0));
        let fq = BlsScalar::random(&mut FixedRng(w.clone(), 0));
        println!("wide fr {} {}", hex(w), hex(&fr.to_bytes()));
        println!("wide fq {} {}", hex(w), hex(&fq.to_bytes()));
    }
    // (g) wire-decoding edge semantics: u = 0 with and without the sign bit,
    // the order-2 point, v = 0 — compare with tests/test_gpu_parity.py::
    // test_decompress_special_encodings (this build accepts rows 1 and 3, "negative zero")
    let enc = |v: BlsScalar, sign: u8| {
        let mut b = v.to_bytes();
        b[31] |= sign << 7;
        b
    };
    let cases = [enc(BlsScalar::one(), 0), enc(BlsScalar::one(), 1), enc(-BlsScalar::one(), 0),
                 enc(-BlsScalar::one(), 1), enc(BlsScalar::zero(), 0), enc(BlsScalar::zero(), 1)];
    for (i, b) in cases.iter().enumerate() {
        match dusk_jubjub::JubJubAffine::from_bytes(b) {
            Ok(p) => println!("from_bytes {i} {} ok u {} v {}", hex(b), hex(&p.get_u().to_bytes()),
                              hex(&p.get_v().to_bytes())),
            Err(_) => println!("from_bytes {i} {} err", hex(b)),
        }
    }
}
