//! Raw bindings of include/sipp_hip.h -- every function and every struct of the header, in the header's order.
//! Never compiled here (no Rust toolchain in the image); tests/test_abi.py::test_rust_bindings_match_the_header parses this
//! file and the header and compares every function (name, argument types, return type) and every struct field.
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct SippCtxOpaque {
    _p: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippStarkConfig {
    pub rate_bits: u32,
    pub cap_height: u32,
    pub pow_bits: u32,
    pub arity_bits: u32,
    pub final_poly_bits: u32,
    pub num_queries: u32,
    pub num_challenges: u32,
    /// 0 = duplex grind (plonky2 fri/prover.rs of 2023), 1 = hash grind (the earlier rule)
    pub pow_rule: u32,
    /// 0 = the statement is observed before the trace cap (this library's format), 1 = starky's recalled order (starts at the trace cap)
    pub fs_rule: u32,
    /// 0 = independent (beta, gamma) per lookup factor, 1 = both factors under gamma (single-column permutation pairs as recalled)
    pub lookup_rule: u32,
}

pub const SIPP_G1_EXP: c_int = 0;
pub const SIPP_G2_EXP: c_int = 1;
pub const SIPP_FQ12_EXP: c_int = 2;
pub const SIPP_MAP_G2: c_int = 3;
pub const SIPP_G1_EXP_HARDENED: c_int = 4;
pub const SIPP_G2_EXP_HARDENED: c_int = 5;
pub const SIPP_PAIRING: c_int = 6;
/// the version of include/sipp_hip.h these bindings were written against (sipp_ctx_create_checked refuses a library of another version)
pub const SIPP_ABI_VERSION: u32 = 3;
pub const SIPP_SALT_SIZE: usize = 4;
pub const SIPP_FRI_MAX_ROUNDS: usize = 32;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippFriParams {
    pub rate_bits: u32,
    pub cap_height: u32,
    pub pow_bits: u32,
    pub num_queries: u32,
    pub pow_rule: u32,
    pub hiding: u32,
    pub n_rounds: u32,
    pub arity_bits: [u32; SIPP_FRI_MAX_ROUNDS],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippOracle {
    pub d_coeffs: *const u64,
    pub d_lde: *const u64,
    pub d_tree: *const u64,
    pub n_polys: u32,
    pub n_salt: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPolyRange {
    pub oracle: u32,
    pub col_begin: u32,
    pub col_end: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippFriBatch {
    pub point: [u64; 2],
    pub n_ranges: u32,
    pub ranges: *const SippPolyRange,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippChallenger {
    pub state: [u64; 12],
    pub in_buf: [u64; 8],
    pub n_in: u64,
    pub out_buf: [u64; 8],
    pub n_out: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPlonkParams {
    pub num_routed_wires: u32,
    pub max_degree: u32,
    pub num_challenges: u32,
}

/// one gate type of a circuit given as data (sipp_hip.h, "gates as data"): its selector and where its constraint programs start
#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPlonkGate {
    pub selector_index: u32,
    pub row: u32,
    pub group_lo: u32,
    pub group_hi: u32,
    pub prog_offset: u32,
    pub num_constraints: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPlonkCircuit {
    pub num_wires: u32,
    pub num_constants: u32,
    pub num_selectors: u32,
    pub num_gates: u32,
    pub gates: *const SippPlonkGate,
    /// per constraint: n_mono, then per monomial coef, n_factors, (kind, index) pairs; kind 0 = wire, 1 = constant column, 2 = public_inputs_hash word
    pub programs: *const i64,
    pub program_words: u32,
}

/// one gate family's witness generator with its layout (include/sipp_hip.h, "WITNESS GENERATORS"): kind = SIPP_GEN_*
#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPlonkGenerator {
    pub kind: u32,
    pub selector_index: u32,
    pub row: u32,
    pub p: [u32; 5],
}

/// the order copy constraints force on the generators, fixed at circuit-build time: rows sorted by level (device), per level the cells its
/// outputs feed (device; cell = wire * N + row); the offsets are host arrays of n_levels + 1 entries
#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPlonkSchedule {
    pub n_levels: u32,
    pub d_rows: *const u32,
    pub level_offsets: *const u32,
    pub d_copy_src: *const u64,
    pub d_copy_dst: *const u64,
    pub copy_offsets: *const u32,
}

/// opaque: a built circuit (gate set, generators, schedule, constants_sigmas and its commitment on the device)
#[repr(C)]
pub struct SippCircuitDataOpaque {
    _private: [u8; 0],
}

/// the level schedule as host arrays (sipp_circuit_build copies them)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct SippPlonkScheduleHost {
    pub n_levels: u32,
    pub rows: *const u32,
    pub level_offsets: *const u32,
    pub copy_src: *const u64,
    pub copy_dst: *const u64,
    pub copy_offsets: *const u32,
}

#[link(name = "sipp_hip")]
extern "C" {
    pub fn sipp_default_config(cfg: *mut SippStarkConfig);
    pub fn sipp_ctx_create(out: *mut *mut SippCtxOpaque, device: c_int, cfg: *const SippStarkConfig, workspace_bytes: usize) -> c_int;
    pub fn sipp_abi_version() -> u32;
    pub fn sipp_ctx_create_checked(out: *mut *mut SippCtxOpaque, device: c_int, cfg: *const SippStarkConfig, cfg_size: usize, abi_version: u32, workspace_bytes: usize) -> c_int;
    pub fn sipp_ctx_set_kernel_routes(ctx: *mut SippCtxOpaque, routes: u32) -> c_int;
    pub fn sipp_ctx_destroy(ctx: *mut SippCtxOpaque);
    /// kinds 0 / 1 on this ctx mean the hardened G1 / G2 AIRs (kinds 4 / 5)
    pub fn sipp_ctx_set_hardened(ctx: *mut SippCtxOpaque, on: c_int) -> c_int;
    pub fn sipp_ctx_set_stream_priority(ctx: *mut SippCtxOpaque, level: c_int) -> c_int;
    pub fn sipp_last_error(ctx: *const SippCtxOpaque) -> *const c_char;
    pub fn sipp_sync(ctx: *mut SippCtxOpaque) -> c_int;
    pub fn sipp_stream(ctx: *mut SippCtxOpaque) -> *mut c_void;
    pub fn sipp_g1_exp_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_g2_exp_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_fq12_exp_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    /// kind 3: the STARK behind batch_map_to_g2_circuit (reference src/bin/bls_aggregation.rs:65); records (u, x, y) of 48 u32
    pub fn sipp_map_to_g2_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_pairing_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    /// messages -> MapToG2 records, the G2ExpStark records of the cofactor clearing (2n, may be null), the cleared points (may be null)
    pub fn sipp_map_to_g2(ctx: *mut SippCtxOpaque, msgs: *const u32, n: usize, map_ios: *mut u32, g2_ios: *mut u32, cleared: *mut u32) -> c_int;
    /// any kind, synchronously (kinds 4 / 5: G1 / G2 exponentiation with the hardened AIR)
    pub fn sipp_prove(ctx: *mut SippCtxOpaque, kind: c_int, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_prove_async(ctx: *mut SippCtxOpaque, kind: c_int, ios: *const u32, num_io: usize, proof_out: *mut u64, proof_cap: usize) -> c_int;
    pub fn sipp_wait(ctx: *mut SippCtxOpaque, proof_len: *mut usize) -> c_int;
    /// arrays of three, indexed by kind
    pub fn sipp_instance_prove(ctxs: *const *mut SippCtxOpaque, ios: *const *const u32, num_io: *const usize, proof_out: *const *mut u64,
                               proof_cap: *const usize, proof_len: *mut usize) -> c_int;
    /// a queue of `count` instances through `in_flight` slots of three ctxs (arrays indexed 3 * i + kind)
    pub fn sipp_instances_prove(ctxs: *const *mut SippCtxOpaque, in_flight: usize, count: usize, ios: *const *const u32,
                                num_io: *const usize, proof_out: *const *mut u64, proof_cap: *const usize, proof_len: *mut usize,
                                status: *mut c_int) -> c_int;
    /// range of an obligation list that GPU `rank` of `world` proves as a STARK of its own (level L-D)
    pub fn sipp_io_shard(num_io: usize, world: u32, rank: u32, first: *mut usize, count: *mut usize) -> c_int;
    pub fn sipp_exp_outputs(ctx: *mut SippCtxOpaque, kind: c_int, ios: *mut u32, num_io: usize) -> c_int;
    /// the library's verifier of a flat proof (host code): starky's native verify_stark_proof in the generators / data.verify
    /// (reference src/verifier_circuit.rs:254); cfg null = default; *reason receives the refusing stage (include/sipp_hip.h)
    pub fn sipp_stark_verify(proof: *const u64, len: usize, cfg: *const SippStarkConfig, reason: *mut c_int) -> c_int;
    /// the verifiers of the generic proofs (host code): PolynomialBatch::verify_openings, and plonk/verifier.rs for the outer proof
    pub fn sipp_fri_verify_openings(proof: *const u64, len: usize, caps: *const *const u64, ncols: *const u32, n_salt: *const u32, n_oracles: usize,
                                    batches: *const SippFriBatch, n_batches: usize, log_n: u32, p: *const SippFriParams, ch: *mut SippChallenger,
                                    reason: *mut c_int) -> c_int;
    pub fn sipp_plonk_generate_witness(ctx: *mut SippCtxOpaque, d_wires: *mut u64, d_constants: *const u64, log_n: u32, num_wires: u32,
                                       num_constants: u32, gens: *const SippPlonkGenerator, n_gens: usize, public_inputs_hash: *const u64) -> c_int;
    pub fn sipp_plonk_generate_witness_levels(ctx: *mut SippCtxOpaque, d_wires: *mut u64, d_constants: *const u64, log_n: u32, num_wires: u32,
                                              num_constants: u32, gens: *const SippPlonkGenerator, n_gens: usize, public_inputs_hash: *const u64,
                                              sched: *const SippPlonkSchedule) -> c_int;
    pub fn sipp_circuit_workspace_bytes(log_n: u32, p: *const SippPlonkParams, fp: *const SippFriParams, c: *const SippPlonkCircuit) -> usize;
    pub fn sipp_circuit_build(ctx: *mut SippCtxOpaque, log_n: u32, p: *const SippPlonkParams, fp: *const SippFriParams, c: *const SippPlonkCircuit,
                              constants_sigmas: *const u64, gens: *const SippPlonkGenerator, n_gens: usize, sched: *const SippPlonkScheduleHost,
                              circuit_digest: *const u64, out: *mut *mut SippCircuitDataOpaque) -> c_int;
    pub fn sipp_circuit_destroy(cd: *mut SippCircuitDataOpaque);
    pub fn sipp_circuit_verifier_data(cd: *const SippCircuitDataOpaque, cap_out: *mut u64, digest_out: *mut u64) -> c_int;
    pub fn sipp_circuit_proof_size(cd: *const SippCircuitDataOpaque, n_public_inputs: u32) -> usize;
    pub fn sipp_circuit_prove(cd: *mut SippCircuitDataOpaque, wires: *const u64, public_inputs: *const u64, n_public_inputs: u32,
                              proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_circuit_verify(cd: *const SippCircuitDataOpaque, proof: *const u64, len: usize, reason: *mut c_int) -> c_int;
    pub fn sipp_plonk_verify_gates(proof: *const u64, len: usize, constants_sigmas_cap: *const u64, p: *const SippPlonkParams, fp: *const SippFriParams,
                                   c: *const SippPlonkCircuit, circuit_digest: *const u64, reason: *mut c_int) -> c_int;
    pub fn sipp_proof_size(ctx: *const SippCtxOpaque, kind: c_int, num_io: usize) -> usize;
    pub fn sipp_workspace_bytes(kind: c_int, num_io: usize) -> usize;
    pub fn sipp_workspace_bytes_cfg(kind: c_int, num_io: usize, cfg: *const SippStarkConfig) -> usize;
    pub fn sipp_device_memory(device: c_int, free_bytes: *mut usize, total_bytes: *mut usize) -> c_int;
    pub fn sipp_stark_shape(ctx: *const SippCtxOpaque, kind: c_int, num_io: usize, log_rows: *mut u32, main_cols: *mut u32,
                            perm_cols: *mut u32, quotient_cols: *mut u32) -> c_int;
    pub fn sipp_inner_product(ctx: *mut SippCtxOpaque, g1: *const u32, g2: *const u32, n: usize, out: *mut u32) -> c_int;
    pub fn sipp_inner_products(ctx: *mut SippCtxOpaque, g1: *const u32, g2: *const u32, n: usize, count: usize, out: *mut u32) -> c_int;
    pub fn sipp_native_proof_words(n: usize) -> usize;
    pub fn sipp_prove_native(ctx: *mut SippCtxOpaque, A: *const u32, B: *const u32, n: usize, proof: *mut u32) -> c_int;
    pub fn sipp_verify_native(ctx: *mut SippCtxOpaque, A: *const u32, B: *const u32, n: usize, proof: *const u32, statement: *mut u32,
                              g1_ios: *mut u32, g2_ios: *mut u32, fq12_ios: *mut u32, accepted: *mut c_int) -> c_int;
    pub fn sipp_fri_const_arity(p: *mut SippFriParams, arity_bits: u32, final_poly_bits: u32, degree_bits: u32);
    pub fn sipp_commit_batch_ex(ctx: *mut SippCtxOpaque, d_in: *const u64, from_coeffs: c_int, d_coeffs: *mut u64, d_lde: *mut u64,
                                d_tree: *mut u64, ncols: usize, log_n: u32, rate_bits: u32, cap_height: u32, d_salt: *const u64,
                                n_salt: u32, cap_out: *mut u64) -> c_int;
    pub fn sipp_fri_proof_size(oracles: *const SippOracle, n_oracles: usize, batches: *const SippFriBatch, n_batches: usize, log_n: u32,
                               p: *const SippFriParams) -> usize;
    pub fn sipp_fri_prove_openings(ctx: *mut SippCtxOpaque, oracles: *const SippOracle, n_oracles: usize, batches: *const SippFriBatch,
                                   n_batches: usize, log_n: u32, p: *const SippFriParams, ch: *mut SippChallenger, proof_out: *mut u64,
                                   proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_plonk_num_partial_products(p: *const SippPlonkParams) -> u32;
    pub fn sipp_plonk_zs_partial_products(ctx: *mut SippCtxOpaque, d_wires: *const u64, d_sigmas: *const u64, log_n: u32, p: *const SippPlonkParams,
                                          betas: *const u64, gammas: *const u64, d_out: *mut u64) -> c_int;
    pub fn sipp_plonk_quotient_chunks(ctx: *mut SippCtxOpaque, d_wires_lde: *const u64, d_sigmas_lde: *const u64, d_zs_lde: *const u64, log_n: u32,
                                      rate_bits: u32, p: *const SippPlonkParams, betas: *const u64, gammas: *const u64, alphas: *const u64,
                                      d_chunks: *mut u64) -> c_int;
    pub fn sipp_plonk_quotient_chunks_ex(ctx: *mut SippCtxOpaque, d_wires_lde: *const u64, d_sigmas_lde: *const u64, d_zs_lde: *const u64, log_n: u32,
                                         rate_bits: u32, p: *const SippPlonkParams, betas: *const u64, gammas: *const u64, alphas: *const u64,
                                         d_gate_terms: *const u64, num_gate_terms: u32, d_chunks: *mut u64) -> c_int;
    pub fn sipp_plonk_perm_proof_size(log_n: u32, p: *const SippPlonkParams, fp: *const SippFriParams) -> usize;
    pub fn sipp_plonk_perm_prove(ctx: *mut SippCtxOpaque, d_wires: *const u64, d_sigmas: *const u64, log_n: u32, p: *const SippPlonkParams,
                                 fp: *const SippFriParams, circuit_digest: *const u64, public_inputs_hash: *const u64, proof_out: *mut u64,
                                 proof_cap: usize, proof_len: *mut usize) -> c_int;
    /// prove() except gate evaluation and witness generation: caller-supplied gate-constraint terms, public inputs, pre-committed oracles
    pub fn sipp_plonk_prove_ex(ctx: *mut SippCtxOpaque, d_wires: *const u64, d_sigmas: *const u64, wires_oracle: *const SippOracle,
                               wires_cap: *const u64, sigmas_oracle: *const SippOracle, log_n: u32, p: *const SippPlonkParams,
                               fp: *const SippFriParams, circuit_digest: *const u64, public_inputs: *const u64, n_public_inputs: u32,
                               d_gate_terms: *const u64, num_gate_terms: u32, proof_out: *mut u64, proof_cap: usize,
                               proof_len: *mut usize) -> c_int;
    /// prove() except witness generation: the gate constraints are interpreted on the device from the circuit description
    pub fn sipp_plonk_gates_proof_size(log_n: u32, p: *const SippPlonkParams, fp: *const SippFriParams, c: *const SippPlonkCircuit,
                                       n_public_inputs: u32) -> usize;
    pub fn sipp_plonk_prove_gates(ctx: *mut SippCtxOpaque, d_wires: *const u64, d_constants_sigmas: *const u64, wires_oracle: *const SippOracle,
                                  wires_cap: *const u64, constants_sigmas_oracle: *const SippOracle, log_n: u32, p: *const SippPlonkParams,
                                  fp: *const SippFriParams, c: *const SippPlonkCircuit, circuit_digest: *const u64, public_inputs: *const u64,
                                  n_public_inputs: u32, proof_out: *mut u64, proof_cap: usize, proof_len: *mut usize) -> c_int;
    pub fn sipp_ntt_batch(ctx: *mut SippCtxOpaque, d_cols: *mut u64, col_stride: usize, ncols: usize, log_n: u32, inverse: c_int) -> c_int;
    pub fn sipp_lde_batch(ctx: *mut SippCtxOpaque, d_values: *const u64, d_coeffs: *mut u64, d_lde: *mut u64, ncols: usize, log_n: u32) -> c_int;
    pub fn sipp_poseidon_leaves(ctx: *mut SippCtxOpaque, d_lde: *const u64, ncols: usize, log_leaves: u32, d_digests: *mut u64) -> c_int;
    pub fn sipp_merkle_cap(ctx: *mut SippCtxOpaque, d_tree: *mut u64, log_leaves: u32, cap_out: *mut u64) -> c_int;
    pub fn sipp_commit_batch(ctx: *mut SippCtxOpaque, d_values: *const u64, d_coeffs: *mut u64, d_lde: *mut u64, d_tree: *mut u64,
                             ncols: usize, log_n: u32, cap_out: *mut u64) -> c_int;
    pub fn sipp_trace_build(ctx: *mut SippCtxOpaque, kind: c_int, ios: *const u32, num_io: usize, d_trace: *mut u64) -> c_int;
    pub fn sipp_poseidon_permute(ctx: *mut SippCtxOpaque, d_states: *mut u64, n: usize) -> c_int;
    pub fn sipp_host_poseidon_permute(states: *mut u64, n: usize, r#impl: c_int) -> c_int;
    pub fn sipp_profile_enable(ctx: *mut SippCtxOpaque, enable: c_int) -> c_int;
    pub fn sipp_profile_reset(ctx: *mut SippCtxOpaque) -> c_int;
    pub fn sipp_profile_report(ctx: *mut SippCtxOpaque, buf: *mut c_char, cap: usize) -> c_int;
    pub fn sipp_timer_start(ctx: *mut SippCtxOpaque) -> c_int;
    pub fn sipp_timer_stop(ctx: *mut SippCtxOpaque, ms: *mut f32) -> c_int;
}
