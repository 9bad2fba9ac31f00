//! capture_golden -- dumps, with the REFERENCE's own crates, the values that would pin this repository's parity
//! (SURVEY.md section 8c "what would pin parity"; INTEGRATION.md section 4 documents the JSON schema).
//!
//! SOURCE ONLY: there is no Rust toolchain in the build image, so this file has never been compiled.  The calls into
//! `sipp::*` are written against /root/reference/src (prover_native.rs:26-80, verifier_native.rs:14-85,
//! transcript_native.rs:14-77, statements.rs); the calls into plonky2 / starky / starky-bn254 are written from memory of
//! the pinned revisions (Cargo.toml:21-27) and are marked `// [recalled]` -- adjust the paths if rustc disagrees.
//!
//!   cargo run --release --example capture_golden -- 128 > tests/golden/reference/capture_n128.json
//!
//! Inputs: A_i = [s_i] G1, B_i = [t_i] G2, scalars = four successive SplitMix64 outputs (seed 0x51515050 + 1 for n = 128, the
//! table SEEDS of tools/gen_golden.py) read big-endian and reduced mod r -- oracle/py/sipp_native.py::synthetic_inputs.
//! tests/test_reference_capture.py compares the file section by section with the CPU oracle and, on a GPU box, with the HIP path
//! (plain kinds, fs_rule = lookup_rule = pow_rule = 1: the configuration closest to upstream), and names the first section
//! that differs.
use ark_bn254::{Fq, Fq12, Fq2, Fr, G1Affine, G2Affine};
use ark_ec::{AffineRepr, CurveGroup};
use ark_ff::{BigInteger, PrimeField};
use num_bigint::BigUint;
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::polynomial::PolynomialValues; // [recalled]
use plonky2::field::types::{Field, PrimeField64};
use plonky2::fri::oracle::PolynomialBatch; // [recalled]
use plonky2::hash::hash_types::HashOut;
use plonky2::hash::hashing::hash_n_to_hash_no_pad;
use plonky2::hash::poseidon::{PoseidonHash, PoseidonPermutation};
use plonky2::plonk::config::{GenericConfig, Hasher, PoseidonGoldilocksConfig};
use plonky2::util::timing::TimingTree;
use serde_json::{json, Value};
use sipp::prover_native::sipp_prove_native;
use sipp::transcript_native::Transcript;
use sipp::verifier_native::sipp_verify_native;

type F = GoldilocksField;
type C = PoseidonGoldilocksConfig;
const D: usize = 2;

struct SplitMix64(u64);
impl SplitMix64 {
    fn next(&mut self) -> u64 {
        self.0 = self.0.wrapping_add(0x9E3779B97F4A7C15);
        let mut z = self.0;
        z = (z ^ (z >> 30)).wrapping_mul(0xBF58476D1CE4E5B9);
        z = (z ^ (z >> 27)).wrapping_mul(0x94D049BB133111EB);
        z ^ (z >> 31)
    }
    fn scalar(&mut self) -> Fr {
        // v = (((w0 << 64 | w1) << 64 | w2) << 64 | w3) mod r, 1 if that is 0
        let mut v = BigUint::from(0u32);
        for _ in 0..4 {
            v = (v << 64) | BigUint::from(self.next());
        }
        let s = Fr::from(v);
        if s == Fr::from(0u64) { Fr::from(1u64) } else { s }
    }
}

fn fq_limbs(x: Fq) -> Vec<u32> {
    let mut l = BigUint::from_bytes_le(&x.into_bigint().to_bytes_le()).to_u32_digits();
    l.resize(8, 0);
    l
}
fn fr_limbs(x: Fr) -> Vec<u32> {
    let mut l = BigUint::from_bytes_le(&x.into_bigint().to_bytes_le()).to_u32_digits();
    l.resize(8, 0);
    l
}
fn g1_limbs(p: &G1Affine) -> Vec<u32> {
    [fq_limbs(p.x), fq_limbs(p.y)].concat()
}
fn fq2_limbs(x: Fq2) -> Vec<u32> {
    [fq_limbs(x.c0), fq_limbs(x.c1)].concat()
}
fn g2_limbs(p: &G2Affine) -> Vec<u32> {
    [fq2_limbs(p.x), fq2_limbs(p.y)].concat() // x.c0 x.c1 y.c0 y.c1: transcript_native.rs:48-54
}
fn fq12_limbs(x: Fq12) -> Vec<u32> {
    // the 12 MyFq12 coefficients, the order transcript_native.rs:32-40 hashes them in
    let m: plonky2_bn254::fields::native::MyFq12 = x.into();
    m.coeffs.iter().flat_map(|&c| fq_limbs(c)).collect()
}
fn hash_json(h: &HashOut<F>) -> Value {
    json!(h.elements.iter().map(|e| e.to_canonical_u64()).collect::<Vec<_>>())
}

fn main() {
    let n: usize = std::env::args().nth(1).map(|s| s.parse().unwrap()).unwrap_or(128);
    let seed: u64 = match n { 4 => 7, 8 => 0x51515050, 128 => 0x51515050 + 1, 1024 => 0x51515050 + 2, _ => n as u64 };
    let mut rng = SplitMix64(seed);
    // all A scalars first, then all B scalars (synthetic_inputs draws them in that order)
    let a: Vec<G1Affine> = (0..n).map(|_| (G1Affine::generator() * rng.scalar()).into_affine()).collect();
    let b: Vec<G2Affine> = (0..n).map(|_| (G2Affine::generator() * rng.scalar()).into_affine()).collect();

    // ---- native chain (prover_native.rs:26-80) and the challenges the transcript produces (transcript_native.rs:56-65) ----
    let proof = sipp_prove_native(&a, &b);
    let statement = sipp_verify_native(&a, &b, &proof).expect("native verification");
    let mut t = Transcript::<F>::new();
    a.iter().zip(b.iter()).for_each(|(x, y)| {
        t.append_g1(*x);
        t.append_g2(*y);
    });
    let mut msgs = proof.clone();
    t.append_fq12(msgs.pop().unwrap());
    let mut challenges = vec![];
    while !msgs.is_empty() {
        t.append_fq12(msgs.pop().unwrap());
        t.append_fq12(msgs.pop().unwrap());
        challenges.push(fr_limbs(t.get_challenge()));
    }
    let statement_limbs: Vec<u32> = [
        statement.A.iter().flat_map(g1_limbs).collect::<Vec<_>>(),
        statement.B.iter().flat_map(g2_limbs).collect::<Vec<_>>(),
        fq12_limbs(statement.Z),
        g1_limbs(&statement.final_A),
        g2_limbs(&statement.final_B),
        fq12_limbs(statement.final_Z),
    ]
    .concat(); // statements.rs:24-39

    // ---- Poseidon: the sponge and the compression on fixed inputs ----
    let seq = |k: usize| (0..k).map(|i| F::from_canonical_u64(0x0123456789abcdef_u64.wrapping_mul(i as u64 + 1) % F::ORDER)).collect::<Vec<_>>();
    let hash_cases: Vec<Value> = [0usize, 1, 4, 7, 8, 9, 16, 21]
        .iter()
        .map(|&k| {
            let inp = seq(k);
            let out = hash_n_to_hash_no_pad::<F, PoseidonPermutation<F>>(&inp);
            json!({"input": inp.iter().map(|e| e.to_canonical_u64()).collect::<Vec<_>>(), "output": hash_json(&out)})
        })
        .collect();
    let l = hash_n_to_hash_no_pad::<F, PoseidonPermutation<F>>(&seq(5));
    let r = hash_n_to_hash_no_pad::<F, PoseidonPermutation<F>>(&seq(6));
    let two = <PoseidonHash as Hasher<F>>::two_to_one(l, r);

    // ---- PolynomialBatch::from_values: 4 columns of 2^10 values from SplitMix64(0xba7c4) mod p, blowup 2, cap height 4 ----
    let mut prng = SplitMix64(0xba7c4);
    let cols: Vec<PolynomialValues<F>> = (0..4).map(|_| PolynomialValues::new((0..1024).map(|_| F::from_noncanonical_u64(prng.next())).collect())).collect();
    let batch = PolynomialBatch::<F, C, D>::from_values(cols, 1, false, 4, &mut TimingTree::default(), None); // [recalled]
    let cap: Vec<Value> = batch.merkle_tree.cap.0.iter().map(hash_json).collect();

    // ---- the three starky proofs over the obligation lists of verifier_circuit.rs:68-131 ----
    // [recalled] starky-bn254 @ 2d46f9e: the three generators expose their proof through `*_exp_circuit`'s witness generators; the
    // capture needs the StarkProofWithPublicInputs each of them produces.  Fill `stark_section` from
    //   starky_bn254::curves::g1::exp::{G1ExpStark, ..}, starky_bn254::curves::g2::exp::{G2ExpStark, ..},
    //   starky_bn254::fields::fq12::exp::{Fq12ExpStark, ..} with starky::prover::prove::<F, C, _, D>(stark, &StarkConfig::standard_fast_config(), trace, pi, &mut timing)
    // and serialise: degree_bits, trace_cap, permutation_zs_cap, quotient_polys_cap, openings.{local_values, next_values,
    // permutation_zs, permutation_zs_next, quotient_polys} as [c0, c1] pairs, opening_proof.{commit_phase_merkle_caps, final_poly,
    // pow_witness}, public_inputs -- the field names of starky::proof::StarkProofWithPublicInputs.
    let stark_section = json!({"todo": "serialise the three StarkProofWithPublicInputs here (see the comment above)"});

    let out = json!({
        "schema": "sipp-capture-1",
        "n": n, "seed": seed,
        "inputs": {"A": a.iter().map(g1_limbs).collect::<Vec<_>>(), "B": b.iter().map(g2_limbs).collect::<Vec<_>>()},
        "native": {"proof": proof.iter().map(|x| fq12_limbs(*x)).collect::<Vec<_>>(), "challenges": challenges, "statement": statement_limbs},
        "poseidon": {"hash_no_pad": hash_cases, "two_to_one": [{"left": hash_json(&l), "right": hash_json(&r), "output": hash_json(&two)}]},
        "polynomial_batch": {"log_n": 10, "ncols": 4, "rate_bits": 1, "cap_height": 4, "seed": 0xba7c4, "cap": cap},
        "stark": stark_section,
    });
    println!("{}", serde_json::to_string(&out).unwrap());
}
