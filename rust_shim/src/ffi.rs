//! Raw bindings of include/sipp_hip.h (never compiled here: no Rust toolchain in the image).
use std::os::raw::{c_char, c_int};

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
}

pub const SIPP_G1_EXP: c_int = 0;
pub const SIPP_G2_EXP: c_int = 1;
pub const SIPP_FQ12_EXP: c_int = 2;

#[link(name = "sipp_hip")]
extern "C" {
    pub fn sipp_default_config(cfg: *mut SippStarkConfig);
    pub fn sipp_workspace_bytes(kind: c_int, num_io: usize) -> usize;
    pub fn sipp_ctx_create(out: *mut *mut SippCtxOpaque, device: c_int, cfg: *const SippStarkConfig, workspace_bytes: usize) -> c_int;
    pub fn sipp_ctx_destroy(ctx: *mut SippCtxOpaque);
    pub fn sipp_ctx_set_stream_priority(ctx: *mut SippCtxOpaque, level: c_int) -> c_int;
    pub fn sipp_last_error(ctx: *const SippCtxOpaque) -> *const c_char;
    pub fn sipp_proof_size(ctx: *const SippCtxOpaque, kind: c_int, num_io: usize) -> usize;
    pub fn sipp_g1_exp_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, out: *mut u64, cap: usize, len: *mut usize) -> c_int;
    pub fn sipp_g2_exp_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, out: *mut u64, cap: usize, len: *mut usize) -> c_int;
    pub fn sipp_fq12_exp_prove(ctx: *mut SippCtxOpaque, ios: *const u32, num_io: usize, out: *mut u64, cap: usize, len: *mut usize) -> c_int;
    pub fn sipp_prove_async(ctx: *mut SippCtxOpaque, kind: c_int, ios: *const u32, num_io: usize, out: *mut u64, cap: usize) -> c_int;
    pub fn sipp_wait(ctx: *mut SippCtxOpaque, len: *mut usize) -> c_int;
    pub fn sipp_instance_prove(ctxs: *const *mut SippCtxOpaque, ios: *const *const u32, num_io: *const usize, out: *const *mut u64,
                               cap: *const usize, len: *mut usize) -> c_int;
    pub fn sipp_inner_products(ctx: *mut SippCtxOpaque, g1: *const u32, g2: *const u32, n: usize, count: usize, out: *mut u32) -> c_int;
    pub fn sipp_native_proof_words(n: usize) -> usize;
    pub fn sipp_prove_native(ctx: *mut SippCtxOpaque, a: *const u32, b: *const u32, n: usize, proof: *mut u32) -> c_int;
    pub fn sipp_verify_native(ctx: *mut SippCtxOpaque, a: *const u32, b: *const u32, n: usize, proof: *const u32, statement: *mut u32,
                              g1_ios: *mut u32, g2_ios: *mut u32, fq12_ios: *mut u32, accepted: *mut c_int) -> c_int;
    pub fn sipp_exp_outputs(ctx: *mut SippCtxOpaque, kind: c_int, ios: *mut u32, num_io: usize) -> c_int;
    /// a queue of `count` instances through `in_flight` slots of three ctxs (arrays indexed 3 * i + kind)
    pub fn sipp_instances_prove(ctxs: *const *mut SippCtxOpaque, in_flight: usize, count: usize, ios: *const *const u32,
                                num_io: *const usize, out: *const *mut u64, cap: *const usize, len: *mut usize, status: *mut c_int) -> c_int;
    /// range of an obligation list that GPU `rank` of `world` proves as a STARK of its own (level L-D)
    pub fn sipp_io_shard(num_io: usize, world: u32, rank: u32, first: *mut usize, count: *mut usize) -> c_int;
}
