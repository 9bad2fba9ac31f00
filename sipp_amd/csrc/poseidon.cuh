// sipp_amd/csrc/poseidon.cuh -- Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds),
// one state per lane, for gfx950.
//
// Replaces plonky2's PoseidonPermutation / PoseidonHash (hash/poseidon.rs @ 541e127; selected by the
// reference at src/verifier_circuit.rs:196 `PoseidonGoldilocksConfig` and used natively at
// src/transcript_native.rs:27,57).
//
// Mapping to the hardware (DESIGN.md "Poseidon"):
//  * 12 x u64 state lives in 24 VGPRs of ONE lane; a wave hashes 64 independent leaves/nodes, so there
//    is no cross-lane traffic at all and every global access is lane-contiguous.
//  * round constants / sparse-matrix tables are wave-uniform: they sit in __constant__ memory and are
//    fetched with scalar loads (s_load), costing no VGPRs and no VALU issue slots.
//  * full-round MDS (circulant, entries <= 41): state split into 32-bit halves, each output is two
//    chains of 12 v_mad_u64_u32 (32 x 32 + 64) and ONE 96-bit reduction.
//  * the 22 partial rounds use the equivalent sparse factorisation (tables derived and verified in
//    tools/gen_poseidon_header.py): 1 dense 11x11 pre-multiplication, then per round a dot product and
//    11 multiply-adds instead of a dense MDS -- evaluated LAZILY in two blocks of 11 rounds: lanes 1..11 are
//    linear in the block's lane-0 values x_t, so inside a block nothing but lane 0 is reduced mod p and every
//    term is (per-lane u32 half) x (22-bit limb of a wave-uniform constant) accumulated by ONE v_mad_u64_u32
//    into a 64-bit accumulator that cannot overflow (Acc6).  594 + 132 such six-instruction MACs and 55
//    reductions replace 242 mulmods, 242 160-bit MACs and 264 reductions; the dense pre-multiplication is merged
//    with the linear layer of the full round before it (full_round3_combined).
//  * the round loops are NOT unrolled (code stays inside the instruction cache); the per-lane loops are.
#pragma once
#include "gl.hpp"
#include "gl_lazy.cuh"
#include "poseidon_constants.h"

namespace poseidon {

__constant__ uint64_t c_rc[360 + 12];  // 30 rounds x 12, then 12 zeros: the "next round" constants of round 29
__constant__ uint64_t c_fast_first[12];
__constant__ uint64_t c_fast_scalar[22];
__constant__ uint64_t c_fast_mi[121];
__constant__ uint64_t c_fast_vs[22 * 11];
__constant__ uint64_t c_fast_what[22 * 11];
__constant__ uint32_t c_blk3[2 * SIPP_POSEIDON_BLK_WORDS];
__constant__ uint32_t c_comb3[396];
__constant__ uint64_t c_comb_c[12];

// S-box products: the hand-scheduled block of gl_lazy.cuh when the translation unit reserves its temporaries (GLL_T), the compiler's
// sequence otherwise (-DSIPP_POSEIDON_C_MUL keeps the latter for A/B runs)
#if defined(GLL_T) && !defined(SIPP_POSEIDON_C_MUL)
#define SIPP_PMUL gll::mul_nc
#define SIPP_PRED96 gll::reduce96_nc
#else
#define SIPP_PMUL gl::mul_nc
#define SIPP_PRED96 gl::reduce96_nc
#endif
// the two- and four-lanes-per-state kernels (two waves per SIMD, latency-bound): the compiler's products interleave across the
// state elements, the fixed-register blocks cannot -- measured slower there (pair kernel 35.5 against 32 ms per n = 128 instance)
#ifdef SIPP_POSEIDON_THIN_ASM_MUL
#define SIPP_PMUL_THIN SIPP_PMUL
#else
#define SIPP_PMUL_THIN gl::mul_nc
#endif
__device__ __forceinline__ uint64_t sbox_thin(uint64_t x) {
    uint64_t x2 = SIPP_PMUL_THIN(x, x);
    uint64_t x3 = SIPP_PMUL_THIN(x2, x);
    uint64_t x4 = SIPP_PMUL_THIN(x2, x2);
    return SIPP_PMUL_THIN(x3, x4);
}
__device__ __forceinline__ uint64_t sbox(uint64_t x) {
    // lazy reduction: every intermediate is any u64 congruent to the true value (canonicalised once at the end)
    uint64_t x2 = SIPP_PMUL(x, x);
    uint64_t x3 = SIPP_PMUL(x2, x);
    uint64_t x4 = SIPP_PMUL(x2, x2);
    return SIPP_PMUL(x3, x4);
}

// out[r] = sum_i s[(i + r) % 12] * CIRC[i] + s[r] * DIAG[r] (+ add[r]),  CIRC = 17 15 41 16 2 28 13 13 39 18 34 20, DIAG[0] = 8.
// `add` (wave-uniform, may be null at compile time) = the NEXT round's constants: they start the two multiply-add chains
// instead of costing a modular addition per element in front of the next S-box layer.
template <bool ADD>
__device__ __forceinline__ void mds_full(uint64_t s[12], const uint64_t* __restrict__ add) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        if (ADD) {
            al = (uint32_t)add[r];
            ah = add[r] >> 32;
        }
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)lo[(i + r) % 12] * CIRC[i];
            ah += (uint64_t)hi[(i + r) % 12] * CIRC[i];
        }
        if (r == 0) {
            al += (uint64_t)lo[0] * 8u;
            ah += (uint64_t)hi[0] * 8u;
        }
        // value = al + ah * 2^32, al, ah < 2^43
        uint64_t l = al + (ah << 32);
        uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        s[r] = SIPP_PRED96(h, l);
    }
}

using gl::Acc160;

// S-box layer + MDS of round `rnd`; the state already carries the round's constants, and leaves with those of round rnd + 1
// folded into the MDS sums (zeros after round 29: c_rc carries 12 trailing zeros)
__device__ __forceinline__ void full_round(uint64_t s[12], int rnd, uint32_t z) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
    mds_full<true>(s, c_rc + 12 * (rnd + 1) + z);
}

using gl::Acc6;

// full round 3 without its own MDS: its linear layer, the FIRST constants of the sparse form and the dense 11 x 11
// pre-multiplication are ONE affine map s -> C s + c (tools/gen_poseidon_header.py combined_layer).  Row 0 of C is row 0 of
// the MDS (small constants); rows 1..11 are 12 lazy MACs each, with c as the accumulators' start value.
__device__ __forceinline__ void full_round3_combined(uint64_t s[12], uint32_t z) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const uint64_t v = sbox(s[i]);  // the constants of round 3 came with round 2's MDS
        lo[i] = (uint32_t)v;
        hi[i] = (uint32_t)(v >> 32);
    }
    {
        constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
        uint64_t al = (uint64_t)lo[0] * 8u, ah = (uint64_t)hi[0] * 8u;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)lo[i] * CIRC[i];
            ah += (uint64_t)hi[i] * CIRC[i];
        }
        const uint64_t l = al + (ah << 32);
        const uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        s[0] = gl::add_nc(SIPP_PRED96(h, l), c_comb_c[z]);
    }
#pragma unroll
    for (int i = 1; i < 12; i++) {
        Acc6 acc;
        acc.set((uint32_t)c_comb_c[i + z], (uint32_t)(c_comb_c[i + z] >> 32));
#pragma unroll
        for (int j = 0; j < 12; j++) acc.mac(lo[j], hi[j], c_comb3 + z + 3 * ((i - 1) * 12 + j));
        s[i] = acc.reduce();
    }
}

__device__ __forceinline__ void partial_rounds_blocked(uint64_t s[12], uint32_t z) {
    constexpr int B = SIPP_POSEIDON_BLK_ROUNDS;
#pragma unroll 1
    for (int b = 0; b < 22 / B; b++) {
        const uint32_t* __restrict__ T = c_blk3 + z + SIPP_POSEIDON_BLK_WORDS * b;
        uint32_t sl[11], sh[11], xl[B], xh[B];
#pragma unroll
        for (int j = 0; j < 11; j++) {
            sl[j] = (uint32_t)s[j + 1];
            sh[j] = (uint32_t)(s[j + 1] >> 32);
        }
        uint64_t s0 = s[0];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const uint64_t x = gl::add_nc(sbox(s0), c_fast_scalar[B * b + k + z]);
            xl[k] = (uint32_t)x;
            xh[k] = (uint32_t)(x >> 32);
            const uint32_t* __restrict__ Wt = T + 33 * k + 3 * (k * (k - 1) / 2);
            Acc6 acc;
            acc.zero();
            acc.a[0] = (uint64_t)xl[k] * 25u;  // M[0][0] = CIRC[0] + DIAG[0]
            acc.a[3] = (uint64_t)xh[k] * 25u;
#pragma unroll
            for (int i = 0; i < 11; i++) acc.mac(sl[i], sh[i], Wt + 3 * i);
#pragma unroll
            for (int j = 0; j < k; j++) acc.mac(xl[j], xh[j], Wt + 33 + 3 * j);
            s0 = acc.reduce();
        }
        s[0] = s0;
        const uint32_t* __restrict__ V = T + 33 * B + 3 * (B * (B - 1) / 2);
#pragma unroll
        for (int i = 0; i < 11; i++) {
            Acc6 acc;
            acc.set(sl[i], sh[i]);
#pragma unroll
            for (int k = 0; k < B; k++) acc.mac(xl[k], xh[k], V + 3 * (i * B + k));
            s[i + 1] = acc.reduce();
        }
    }
}

__device__ __forceinline__ void permute(uint64_t s[12]) {
    // an opaque zero added to every table index: the tables are wave-uniform and loop-invariant, and without this the
    // compiler hoists ~650 scalar loads out of the caller's column loop, runs out of SGPRs and parks the constants in
    // VGPR lanes (v_writelane once, then a v_readlane + wait states per constant per permutation)
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_nc(s[i], c_rc[i + z]);
#pragma unroll 1
    for (int r = 0; r < 3; r++) full_round(s, r, z);
    full_round3_combined(s, z);
    partial_rounds_blocked(s, z);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_nc(s[i], c_rc[12 * 26 + i + z]);
#pragma unroll 1
    for (int r = 26; r < 30; r++) full_round(s, r, z);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}

}  // namespace poseidon
