//! Dumps golden vectors from the REAL dusk-schnorr 0.18 so the repository's restatement of the
//! hash (Hades constants, sponge padding, 250-bit truncation) can be pinned — the one thing
//! nothing in the reference tree pins.  Run on a machine with cargo + crates.io:
//!     cargo run --bin golden_gen > ../../tests/golden/reference_vectors.txt
//! and commit that file: tests/reference_fixtures.py parses exactly this line format, and
//! tests/test_oracle.py::test_reference_fixtures_pin_the_oracle (CPU) and
//! tests/test_gpu_parity.py::test_reference_fixtures_on_gpu (GPU) then check the oracle and the
//! HIP engine against it — no code change needed.  (tests/golden/predicted_reference.json is what
//! this repository PREDICTS the `sig` lines to be.)
use dusk_bls12_381::BlsScalar;
use dusk_bytes::Serializable;
use dusk_schnorr::{PublicKey, SecretKey};
use ff::Field;
use rand::{rngs::StdRng, SeedableRng};

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}

fn main() {
    // (a) raw sponge: compare with vectors.json["hash"][1]["sponge"] (big-endian hex there)
    let h = dusk_poseidon::sponge::hash(&[BlsScalar::from(1u64), BlsScalar::from(2u64),
                                          BlsScalar::from(3u64)]);
    println!("sponge_hash_1_2_3_le {}", hex(&h.to_bytes()));
    let t = dusk_poseidon::sponge::truncated::hash(&[BlsScalar::from(1u64), BlsScalar::from(2u64),
                                                     BlsScalar::from(3u64)]);
    println!("truncated_hash_1_2_3_le {}", hex(&t.to_bytes()));
    // (b) signatures under the reference's own test seed (tests/schnorr.rs:16)
    let mut rng = StdRng::seed_from_u64(2321);
    for i in 0..8 {
        let sk = SecretKey::random(&mut rng);
        let m = BlsScalar::random(&mut rng);
        let sig = sk.sign(&mut rng, m);
        let pk = PublicKey::from(&sk);
        let [ru, rv] = sig.R().to_hash_inputs();
        let [pu, pv] = pk.as_ref().to_hash_inputs();
        println!(
            "sig {i} sk {} m {} u {} R {}{} PK {}{} sig_bytes {} pk_bytes {} verdict {}",
            hex(&sk.to_bytes()), hex(&m.to_bytes()), hex(&sig.u().to_bytes()),
            hex(&ru.to_bytes()), hex(&rv.to_bytes()), hex(&pu.to_bytes()), hex(&pv.to_bytes()),
            hex(&sig.to_bytes()), hex(&pk.to_bytes()), pk.verify(&sig, m)
        );
    }
    // (c) wire-decoding edge semantics: u = 0 with and without the sign bit,
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
