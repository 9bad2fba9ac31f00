//! Safe wrapper around libsipp_hip.so for the three starky sub-provers of qope/SIPP
//! (reference src/verifier_circuit.rs:133-135).  Source only; see README.md.
pub mod ffi;

use anyhow::{anyhow, Result};
use std::ffi::CStr;

#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Kind {
    G1Exp = 0,   // 56 u32 per IO:  x.x x.y offset.x offset.y exp_val out.x out.y          (8 LE limbs each)
    G2Exp = 1,   // 104 u32 per IO: Fq2 coordinates as (c0, c1)
    Fq12Exp = 2, // 296 u32 per IO: 12 MyFq12 coefficients for x, offset; exp_val; out
    MapG2 = 3,   // 48 u32 per IO:  u, x, y in Fq2 (batch_map_to_g2_circuit, src/bin/bls_aggregation.rs:65)
    G1ExpHardened = 4, // the records of G1Exp / G2Exp proved with the hardened AIR (canonical x3 + x-inequality witness: no free slope
    G2ExpHardened = 5, // where the accumulator meets the running power)
    Pairing = 6, // 144 u32 per IO: P (G1), Q (G2), e(P, Q) as 12 MyFq12 coefficients (pairing_circuit, src/bin/bls_aggregation.rs:76)
}

impl Kind {
    pub fn io_words(self) -> usize {
        match self {
            Kind::G1Exp => 56,
            Kind::G2Exp => 104,
            Kind::Fq12Exp => 296,
            Kind::MapG2 => 48,
            Kind::G1ExpHardened => 56,
            Kind::G2ExpHardened => 104,
            Kind::Pairing => 144,
        }
    }
}

/// One GPU, one HIP stream, one workspace arena.  Not Sync: use one per thread (three of them give the
/// three-stream overlap bench.py uses).
pub struct SippCtx {
    raw: *mut ffi::SippCtxOpaque,
}

unsafe impl Send for SippCtx {}

impl SippCtx {
    pub fn new(device: i32, kind: Kind, max_num_io: usize) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        let ws = unsafe { ffi::sipp_workspace_bytes(kind as i32, max_num_io) };
        // the checked form: a library built from another header version (another sipp_stark_config) is refused, not misread
        let rc = unsafe {
            ffi::sipp_ctx_create_checked(&mut raw, device, std::ptr::null(), std::mem::size_of::<ffi::SippStarkConfig>(), ffi::SIPP_ABI_VERSION, ws)
        };
        if rc != 0 {
            return Err(anyhow!("sipp_ctx_create_checked failed: status {rc} (library ABI version {})", unsafe { ffi::sipp_abi_version() }));
        }
        Ok(Self { raw })
    }

    fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(ffi::sipp_last_error(self.raw)) }.to_string_lossy().into_owned()
    }

    /// `ios`: `num_io * kind.io_words()` little-endian u32 limbs, outputs included.
    /// Returns the flat StarkProofWithPublicInputs buffer (layout: INTEGRATION.md section 2).
    pub fn prove(&mut self, kind: Kind, ios: &[u32]) -> Result<Vec<u64>> {
        let num_io = ios.len() / kind.io_words();
        anyhow::ensure!(num_io * kind.io_words() == ios.len() && num_io > 0, "ragged IO list");
        let cap = unsafe { ffi::sipp_proof_size(self.raw, kind as i32, num_io) };
        let mut buf = vec![0u64; cap];
        let mut len = 0usize;
        let rc = unsafe {
            match kind {
                Kind::G1Exp => ffi::sipp_g1_exp_prove(self.raw, ios.as_ptr(), num_io, buf.as_mut_ptr(), cap, &mut len),
                Kind::G2Exp => ffi::sipp_g2_exp_prove(self.raw, ios.as_ptr(), num_io, buf.as_mut_ptr(), cap, &mut len),
                Kind::Fq12Exp => ffi::sipp_fq12_exp_prove(self.raw, ios.as_ptr(), num_io, buf.as_mut_ptr(), cap, &mut len),
                Kind::MapG2 => ffi::sipp_map_to_g2_prove(self.raw, ios.as_ptr(), num_io, buf.as_mut_ptr(), cap, &mut len),
                Kind::Pairing => ffi::sipp_pairing_prove(self.raw, ios.as_ptr(), num_io, buf.as_mut_ptr(), cap, &mut len),
                Kind::G1ExpHardened | Kind::G2ExpHardened => ffi::sipp_prove(self.raw, kind as i32, ios.as_ptr(), num_io, buf.as_mut_ptr(), cap, &mut len),
            }
        };
        if rc != 0 {
            // the reference's call sites `.unwrap()` (src/verifier_circuit.rs:253): surface the same way
            return Err(anyhow!("sipp prove failed: status {rc}: {}", self.last_error()));
        }
        buf.truncate(len);
        Ok(buf)
    }
}

/// Level L-D: the slice of an obligation list that GPU `rank` of `world` proves as a STARK
/// of its own.  On the circuit side `g1_exp_circuit` / `g2_exp_circuit` / `fq12_exp_circuit` (reference
/// src/verifier_circuit.rs:133-135) are then called once per range, each generator proving its slice on its own GPU.
/// `verify_stark_proof` of the generators (right after `prove`) and the checks behind `data.verify(proof)` (reference
/// src/verifier_circuit.rs:254): the library's own host-side verifier over the flat proof; Err carries the refusing stage.
pub fn verify_stark_proof(proof: &[u64]) -> Result<()> {
    let mut reason: std::os::raw::c_int = 0;
    let rc = unsafe { ffi::sipp_stark_verify(proof.as_ptr(), proof.len(), std::ptr::null(), &mut reason) };
    if rc == 0 { Ok(()) } else { Err(anyhow::anyhow!("sipp_stark_verify: status {} at stage {}", rc, reason)) }
}

/// `data.verify(proof)` for the outer proof (reference src/verifier_circuit.rs:254): plonk/verifier.rs over a proof of sipp_plonk_prove_gates,
/// with the circuit's verifier data (constants_sigmas cap, digest) and the gate set the prover interpreted.
pub fn verify_plonk_proof(proof: &[u64], constants_sigmas_cap: &[u64], params: &ffi::SippPlonkParams, fri: &ffi::SippFriParams,
                          circuit: &ffi::SippPlonkCircuit, circuit_digest: &[u64; 4]) -> Result<()> {
    let mut reason: std::os::raw::c_int = 0;
    let rc = unsafe { ffi::sipp_plonk_verify_gates(proof.as_ptr(), proof.len(), constants_sigmas_cap.as_ptr(), params, fri, circuit, circuit_digest.as_ptr(), &mut reason) };
    if rc == 0 { Ok(()) } else { Err(anyhow::anyhow!("sipp_plonk_verify_gates: status {} at stage {}", rc, reason)) }
}

/// plonky2's `CircuitData` for the outer proof over `sipp_circuit_build` / `_prove` / `_verify` (reference src/verifier_circuit.rs:225
/// `builder.build::<C>()`, :253 `data.prove(pw)`, :254 `data.verify(proof)`): host slices only; the ctx must outlive the data.
pub struct CircuitData {
    raw: *mut ffi::SippCircuitDataOpaque,
}

impl CircuitData {
    /// constants_sigmas: [num_constants + num_routed_wires][N] values; sched / digest may be None (row-local generators / derived digest)
    pub fn build(ctx: *mut ffi::SippCtxOpaque, degree_bits: u32, params: &ffi::SippPlonkParams, fri: &ffi::SippFriParams,
                 circuit: &ffi::SippPlonkCircuit, constants_sigmas: &[u64], generators: &[ffi::SippPlonkGenerator],
                 sched: Option<&ffi::SippPlonkScheduleHost>, digest: Option<&[u64; 4]>) -> Result<Self> {
        let mut raw: *mut ffi::SippCircuitDataOpaque = std::ptr::null_mut();
        let rc = unsafe {
            ffi::sipp_circuit_build(ctx, degree_bits, params, fri, circuit, constants_sigmas.as_ptr(), generators.as_ptr(), generators.len(),
                                    sched.map_or(std::ptr::null(), |s| s as *const _), digest.map_or(std::ptr::null(), |d| d.as_ptr()), &mut raw)
        };
        if rc == 0 { Ok(CircuitData { raw }) } else { Err(anyhow::anyhow!("sipp_circuit_build: status {}", rc)) }
    }

    /// `data.prove(pw)`: pw = the wire table [num_wires][N] with the input cells set; returns the flat proof, public inputs at its end
    pub fn prove(&mut self, partial_witness: &[u64], public_inputs: &[u64]) -> Result<Vec<u64>> {
        let cap = unsafe { ffi::sipp_circuit_proof_size(self.raw, public_inputs.len() as u32) };
        let mut out = vec![0u64; cap];
        let mut len = 0usize;
        let rc = unsafe {
            ffi::sipp_circuit_prove(self.raw, partial_witness.as_ptr(), public_inputs.as_ptr(), public_inputs.len() as u32, out.as_mut_ptr(), cap, &mut len)
        };
        if rc != 0 { return Err(anyhow::anyhow!("sipp_circuit_prove: status {}", rc)); }
        out.truncate(len);
        Ok(out)
    }

    /// `data.verify(proof)`
    pub fn verify(&self, proof: &[u64]) -> Result<()> {
        let mut reason: std::os::raw::c_int = 0;
        let rc = unsafe { ffi::sipp_circuit_verify(self.raw, proof.as_ptr(), proof.len(), &mut reason) };
        if rc == 0 { Ok(()) } else { Err(anyhow::anyhow!("sipp_circuit_verify: status {} at stage {}", rc, reason)) }
    }
}

impl Drop for CircuitData {
    fn drop(&mut self) {
        unsafe { ffi::sipp_circuit_destroy(self.raw) }
    }
}

pub fn io_shard(num_io: usize, world: u32, rank: u32) -> Result<std::ops::Range<usize>> {
    let (mut first, mut count) = (0usize, 0usize);
    let rc = unsafe { ffi::sipp_io_shard(num_io, world, rank, &mut first, &mut count) };
    anyhow::ensure!(rc == 0, "sipp_io_shard failed: status {rc}");
    Ok(first..first + count)
}

/// The three sub-proofs of ONE rank's slices, concurrently on three ctxs of that rank's GPU (`sipp_instance_prove`): what the
/// patched generators of rank `rank` run.  `ios[k]` is the WHOLE list of kind k; an empty slice yields an empty proof.
pub fn prove_rank_shards(ctxs: &mut [SippCtx; 3], ios: [&[u32]; 3], world: u32, rank: u32) -> Result<[Vec<u64>; 3]> {
    let kinds = [Kind::G1Exp, Kind::G2Exp, Kind::Fq12Exp];
    let mut ptr = [std::ptr::null::<u32>(); 3];
    let mut num = [0usize; 3];
    let mut out: [Vec<u64>; 3] = [Vec::new(), Vec::new(), Vec::new()];
    for k in 0..3 {
        let w = kinds[k].io_words();
        anyhow::ensure!(ios[k].len() % w == 0, "ragged IO list");
        let r = io_shard(ios[k].len() / w, world, rank)?;
        num[k] = r.len();
        ptr[k] = ios[k][r.start * w..].as_ptr();
        let cap = if num[k] > 0 { unsafe { ffi::sipp_proof_size(ctxs[k].raw, k as i32, num[k]) } } else { 0 };
        out[k] = vec![0u64; cap.max(1)];
    }
    let raw = [ctxs[0].raw, ctxs[1].raw, ctxs[2].raw];
    let bufs = [out[0].as_mut_ptr(), out[1].as_mut_ptr(), out[2].as_mut_ptr()];
    let caps = [out[0].len(), out[1].len(), out[2].len()];
    let mut lens = [0usize; 3];
    let rc = unsafe { ffi::sipp_instance_prove(raw.as_ptr(), ptr.as_ptr(), num.as_ptr(), bufs.as_ptr(), caps.as_ptr(), lens.as_mut_ptr()) };
    if rc != 0 {
        return Err(anyhow!("sipp_instance_prove failed: status {rc}: {}", ctxs.iter().map(|c| c.last_error()).collect::<Vec<_>>().join("; ")));
    }
    for k in 0..3 {
        out[k].truncate(lens[k]);
    }
    Ok(out)
}

impl Drop for SippCtx {
    fn drop(&mut self) {
        unsafe { ffi::sipp_ctx_destroy(self.raw) }
    }
}

/// Section offsets of the flat proof (u64 words), mirroring StarkProofWithPublicInputs.
pub struct FlatProof<'a> {
    pub words: &'a [u64],
}

impl<'a> FlatProof<'a> {
    pub fn header(&self) -> &[u64] { &self.words[..16] }
    pub fn log_n(&self) -> usize { self.words[2] as usize }
    pub fn widths(&self) -> (usize, usize, usize) { (self.words[4] as usize, self.words[5] as usize, self.words[6] as usize) }
    pub fn cap_words(&self) -> usize { 4usize << self.words[7] }
    pub fn trace_cap(&self) -> &[u64] { &self.words[16..16 + self.cap_words()] }
    pub fn permutation_zs_cap(&self) -> &[u64] { let c = self.cap_words(); &self.words[16 + c..16 + 2 * c] }
    pub fn quotient_polys_cap(&self) -> &[u64] { let c = self.cap_words(); &self.words[16 + 2 * c..16 + 3 * c] }
    /// local_values | next_values | permutation_zs | permutation_zs_next | quotient_polys, 2 words per element
    pub fn openings(&self) -> &[u64] {
        let (w, p, q) = self.widths();
        let start = 16 + 3 * self.cap_words();
        &self.words[start..start + 2 * (2 * w + 2 * p + q)]
    }
    pub fn public_inputs(&self) -> &[u64] {
        let n = (self.words[3] * self.words[11]) as usize;
        &self.words[self.words.len() - n..]
    }
}

// In starky-bn254 (G1ExpStarkyProofGenerator::run_once and its G2 / Fq12 twins) the body
//     let trace = stark.generate_trace(&ios);
//     let pi = stark.generate_public_inputs(&ios);
//     let proof = starky::prover::prove::<F, C, _, D>(stark, &config, trace, pi, &mut TimingTree::default())?;
// becomes
//     let flat = ctx.prove(Kind::G1Exp, &ios_as_u32_limbs)?;
//     let proof = StarkProofWithPublicInputs::<F, C, D>::from_flat(&FlatProof { words: &flat });
// followed, unchanged, by set_stark_proof_with_pis_target(out_buffer, &self.proof_target, &proof).
