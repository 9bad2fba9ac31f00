// sipp_amd/csrc/poseidon.hpp -- Poseidon-Goldilocks permutation (width 12, x^7, 4 + 22 + 4 rounds),
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
#include "gl_lazy.hpp"
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
// dense constant products on the matrix pipe (dense_mfma below): per-lane A fragments in global memory (16 B per lane and digit,
// read with one coalesced load), chain start constants wave-uniform
__device__ uint32_t d_dense_a[SIPP_POSEIDON_DENSE_MATS * 8 * 64 * 4];
__constant__ uint64_t c_dense_start[SIPP_POSEIDON_DENSE_MATS * 12 * 2];

// S-box products and 96-bit reductions: the hand-scheduled blocks of gl_lazy.hpp (the translation unit reserves their temporaries: GLL_T)
#ifndef GLL_T
#error "poseidon.hpp: define GLL_T (first of the ten VGPRs of gl_lazy.hpp's product block) before including it"
#endif
#define SIPP_PMUL gll::mul_nc
#define SIPP_PRED96 gll::reduce96_nc
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

// ---- the same layer on the MATRIX pipe (round 3; scripts/ubench/mfma_mds.hip: 1.33x the VALU form, the pipe is otherwise idle) ----
// The 12 state words are cut into 8 byte planes (six 4 x 4 byte transposes = 48 v_perm_b32; the bytes are made signed by ^ 0x80: i8
// operands are signed); per plane ONE v_mfma_i32_32x32x32_i8 with a block-diagonal A: lane l = (n = l & 31, h = l >> 5) supplies
// B[k = 16 h + e][n] = byte of its element e, and A[(reg & 3) + 8 (reg >> 2) + 4 h][16 h + e] = M[reg][e], so the two wave halves are
// two independent 12 x 12 products and every lane finds ITS twelve sums in accumulator registers 0 .. 11 (the C/D map of the 32 x 32
// shapes) -- no lane exchange.  Recombination: sum_b D_b 2^(8 b) as two chains of four v_mad_i64_i32 (2^8, 2^16, 2^24 in SGPRs); the
// +128 of every byte is repaid by the constant 128 rowsum(M) 0x01010101 that starts each chain together with the next round's
// constant; then the same 96-bit reduction.  96 multiply-adds + 48 + 24 + 8 instead of 288 multiply-adds per layer.
#if 1
typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));

// this lane's fragment of A (constant): row = lane & 31 of the block-diagonal matrix, columns k = 16 (lane >> 5) + j
__device__ __forceinline__ mfma_v4i mds_a_fragment() {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t lane = threadIdx.x & 63, row = lane & 31, h = lane >> 5;
    const bool mine = row < 24 && ((row >> 2) & 1) == h;
    const uint32_t reg = (row & 3) + 4 * (row >> 3);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t e = 0; e < 12; e++) {
        uint32_t v = 0;
#pragma unroll
        for (uint32_t i = 0; i < 12; i++)            // M[reg][e] = CIRC[(e - reg) mod 12] (+ 8 on the diagonal of row 0)
            if ((i + reg) % 12 == e) v = CIRC[i];
        if (reg == 0 && e == 0) v += 8;
        w[e >> 2] |= (mine ? v : 0u) << (8 * (e & 3));
    }
    return mfma_v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
}

__device__ __forceinline__ void mds_transpose4(const uint32_t in[4], uint32_t out[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u), t1 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u), t3 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// d + start as ONE multiply-add (one = an opaque wave-uniform 1; the compiler's sign extension + 64-bit addition are three instructions).
// Not inline asm: D registers are read here, and the wait states between a matrix-pipe write and a VALU read come from the compiler's
// hazard recogniser, which does not look into asm blocks
__device__ __forceinline__ int64_t chain_start(int32_t d, uint64_t start, int32_t one) { return (int64_t)d * (int64_t)one + (int64_t)start; }
// a + 2^32 hh (both below 2^63) -> any u64 congruent to it: the carry of the middle words goes into the 96-bit reduction's high word (two
// carry instructions instead of a 64-bit addition, a 64-bit compare and a select)
__device__ __forceinline__ uint64_t fold_chains(uint64_t a, uint64_t hh) {
    uint32_t c = 0;
    const uint32_t mid = __builtin_addc((uint32_t)(a >> 32), (uint32_t)hh, 0u, &c);
    return SIPP_PRED96((uint32_t)(hh >> 32) + c, ((uint64_t)mid << 32) | (uint32_t)a);
}

template <bool ADD>
__device__ __forceinline__ void mds_full_mfma(uint64_t s[12], const uint64_t* __restrict__ add, mfma_v4i afrag, uint32_t z) {
    uint32_t lo[12], hi[12], plane[8][3];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
        uint32_t o[4];
        mds_transpose4(lo + 4 * g, o);
#pragma unroll
        for (int b = 0; b < 4; b++) plane[b][g] = o[b] ^ 0x80808080u;
        mds_transpose4(hi + 4 * g, o);
#pragma unroll
        for (int b = 0; b < 4; b++) plane[4 + b][g] = o[b] ^ 0x80808080u;
    }
    // 2^8, 2^16, 2^24 as wave-uniform values the compiler cannot fold into literals (VOP3 has no literal operand on gfx9: SGPRs)
    const int32_t p8 = (int32_t)(256u + z), p16 = (int32_t)(65536u + z), p24 = (int32_t)(16777216u + z), p1 = (int32_t)(1u + z);
    int64_t al[12], ah[12];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const mfma_v4i bfrag = {(int)plane[b][0], (int)plane[b][1], (int)plane[b][2], 0};
        const mfma_v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, bfrag, mfma_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 12; r++) {
            int64_t& acc = b < 4 ? al[r] : ah[r];
            if ((b & 3) == 0) {
                // chain start: the bias of the four planes of this chain, and the next round's constant
                const uint64_t bias = (uint64_t)(128u * (256u + (r == 0 ? 8u : 0u))) * 0x01010101ull;
                const uint64_t c = ADD ? (b < 4 ? (uint64_t)(uint32_t)add[r] : (add[r] >> 32)) : 0;
                acc = chain_start(d[r], bias + c, p1);
            } else {
                acc = (int64_t)d[r] * (int64_t)((b & 3) == 1 ? p8 : (b & 3) == 2 ? p16 : p24) + acc;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 12; r++) s[r] = fold_chains((uint64_t)al[r], (uint64_t)ah[r]);
}

// ---- DENSE products with full 64-bit constants on the matrix pipe (round 4) ----------------------------------------------------------
// out[r] = sum_e M[r][e] x_e (+ addend_r) + const_r for one of the five constant matrices the lazy partial rounds are built from (the
// merged affine layer of full round 3; W and V of each block of eleven rounds; tables and the algebra: tools/gen_poseidon_header.py
// dense_tables / dense_model).  x_e: ANY u64, as 8 byte planes (the transposes of mds_full_mfma); the constants: 8 signed base-256 digits;
// 64 products of (plane a) x (digit b), chained in the accumulator by t = a + b (<= 8 x 12 x 2^14 < 2^21 each), and the fifteen sums D_t
// folded into two signed 64-bit chains L, H with 2^64 = 2^32 - 1, 2^96 = -1 (mod p): 19 multiply-adds per output instead of 66 (72) and
// ONE 96-bit reduction (4 instructions) instead of an Acc6 reduction (~50).  The A fragments have the block-diagonal layout of
// mds_a_fragment, so every lane finds its own outputs in accumulator registers 0 .. 11.
template <bool ADDEND>
__device__ __forceinline__ void dense_mfma(const uint32_t lo[12], const uint32_t hi[12], uint32_t mat, uint64_t out[12], const uint64_t* __restrict__ addend,
                                           uint32_t addend_stride, uint32_t z) {
    uint32_t plane[8][3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        uint32_t o[4];
        mds_transpose4(lo + 4 * g, o);
#pragma unroll
        for (int b = 0; b < 4; b++) plane[b][g] = o[b] ^ 0x80808080u;
        mds_transpose4(hi + 4 * g, o);
#pragma unroll
        for (int b = 0; b < 4; b++) plane[4 + b][g] = o[b] ^ 0x80808080u;
    }
    // this lane's A fragments of the eight digits: 12 of the 16 bytes are columns of the matrix (k = 16 h + e, e < 12), the rest is zero
    // and never loaded -- 24 VGPRs instead of 32
    typedef int v3i __attribute__((ext_vector_type(3)));
    v3i afr[8];
    {
        // wave-uniform base + one 32-bit lane offset (no per-lane 64-bit pointers hoisted out of the caller's loops)
        const uint32_t* __restrict__ A = d_dense_a + mat * (8 * 64 * 4);
        const uint32_t lane4 = (threadIdx.x & 63) * 4;
#pragma unroll
        for (int b = 0; b < 8; b++) afr[b] = *reinterpret_cast<const v3i*>(A + (lane4 + b * 256));
    }
    const int32_t p8 = (int32_t)(256u + z), p16 = (int32_t)(65536u + z), p24 = (int32_t)(16777216u + z);
    const int32_t n1 = (int32_t)(0xffffffffu + z), n8 = -p8, n16 = -p16, n24 = -p24, p1 = (int32_t)(1u + z);
    const uint64_t* __restrict__ K = c_dense_start + mat * 24 + z;
    int64_t L[12], H[12];
#pragma unroll
    for (int t = 0; t < 15; t++) {
        __builtin_amdgcn_sched_barrier(0);      // one accumulator chain at a time: keeps a single D (16 VGPRs) live
        mfma_v16i d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int a = (t > 7 ? t - 7 : 0); a <= (t < 7 ? t : 7); a++) {
            const mfma_v4i bfrag = {(int)plane[a][0], (int)plane[a][1], (int)plane[a][2], 0};
            const mfma_v4i afrag = {afr[t - a][0], afr[t - a][1], afr[t - a][2], 0};
            d = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, bfrag, d, 0, 0, 0);
        }
        const int32_t pw = (t & 3) == 0 ? p1 : (t & 3) == 1 ? p8 : (t & 3) == 2 ? p16 : p24;
        const int32_t nw = (t & 3) == 0 ? n1 : (t & 3) == 1 ? n8 : (t & 3) == 2 ? n16 : n24;
#pragma unroll
        for (int r = 0; r < 11; r++) {
            if (t == 0) {
                uint64_t l0 = K[2 * r], h0 = K[2 * r + 1];          // chain starts (positive: + 2^50), plus the per-lane addend's halves
                if (ADDEND) {                                       // per-lane addend (LDS stash, see partial_rounds_blocked)
                    const uint64_t av = addend[r * addend_stride];
                    l0 += (uint32_t)av;
                    h0 += av >> 32;
                }
                L[r] = chain_start(d[r], l0, p1);
                H[r] = (int64_t)h0;
            } else if (t < 4) {
                L[r] = (int64_t)d[r] * (int64_t)pw + L[r];
                asm volatile("" : "+v"(L[r]));
                continue;
            } else if (t < 8) {
                H[r] = (int64_t)d[r] * (int64_t)pw + H[r];
            } else if (t < 12) {
                H[r] = (int64_t)d[r] * (int64_t)pw + H[r];
                L[r] = (int64_t)d[r] * (int64_t)nw + L[r];
            } else {
                L[r] = (int64_t)d[r] * (int64_t)nw + L[r];
            }
            // the chains are folded HERE, one D at a time: without the opaque use the compiler reassociates the sums and keeps all
            // fifteen accumulators (240 VGPRs) until the end
            asm volatile("" : "+v"(L[r]), "+v"(H[r]));
        }
    }
#pragma unroll
    for (int r = 0; r < 11; r++) out[r] = fold_chains((uint64_t)L[r], (uint64_t)H[r]);   // both chains in (0, 2^52)
}
#endif

using gl::Acc160;

// S-box layer + MDS of round `rnd`; the state already carries the round's constants, and leaves with those of round rnd + 1
// folded into the MDS sums (zeros after round 29: c_rc carries 12 trailing zeros)
template <bool MFMA>
__device__ __forceinline__ void full_round(uint64_t s[12], int rnd, uint32_t z, mfma_v4i afrag) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
    if (MFMA)
        mds_full_mfma<true>(s, c_rc + 12 * (rnd + 1) + z, afrag, z);
    else
        mds_full<true>(s, c_rc + 12 * (rnd + 1) + z);
}

using gl::Acc6;

// full round 3 without its own MDS: its linear layer, the FIRST constants of the sparse form and the dense 11 x 11
// pre-multiplication are ONE affine map s -> C s + c (tools/gen_poseidon_header.py combined_layer).  Row 0 of C is row 0 of
// the MDS (small constants); rows 1..11 are 12 lazy MACs each, with c as the accumulators' start value.
template <bool MFMA>
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
    if (MFMA) {
        uint64_t o[12];
        dense_mfma<false>(lo, hi, 0, o, nullptr, 0, z);        // rows 1 .. 11 of C, their constants in the chain starts
#pragma unroll
        for (int i = 1; i < 12; i++) s[i] = o[i - 1];
        return;
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

// stash (MFMA form only): LDS, 11 words per lane at stash[i * stash_stride] -- the block-start state S is an INPUT of the first dense
// product and the ADDEND of the second, eleven rounds later; in between it waits in LDS instead of 22 VGPRs (the compiler spilled
// half of it to scratch: ~35 scratch accesses per permutation with three waves per SIMD to hide them)
template <bool MFMA>
__device__ __forceinline__ void partial_rounds_blocked(uint64_t s[12], uint32_t z, uint64_t* __restrict__ stash = nullptr, uint32_t stash_stride = 0) {
    constexpr int B = SIPP_POSEIDON_BLK_ROUNDS;
    static_assert(!MFMA || B == 11, "the matrix-pipe form is laid out for blocks of eleven rounds");
#pragma unroll 1
    for (int b = 0; b < 22 / B; b++) {
        const uint32_t* __restrict__ T = c_blk3 + z + SIPP_POSEIDON_BLK_WORDS * b;
        uint32_t sl[12], sh[12], xl[12], xh[12];
#pragma unroll
        for (int j = 0; j < 11; j++) {
            sl[j] = (uint32_t)s[j + 1];
            sh[j] = (uint32_t)(s[j + 1] >> 32);
        }
        sl[11] = sh[11] = xl[11] = xh[11] = 0;
        uint64_t s0 = s[0];
        // the block-start state's share of every round's lane-0 value, sum_i W[k][i] S_i, is ONE dense product: on the matrix pipe
        uint64_t pre[12];
        if (MFMA) {
#pragma unroll
            for (int j = 0; j < 11; j++) stash[j * stash_stride] = s[j + 1];
            dense_mfma<false>(sl, sh, 1 + 2 * (uint32_t)b, pre, nullptr, 0, z);
        }
#pragma unroll
        for (int k = 0; k < B; k++) {
            const uint64_t x = gl::add_nc(sbox(s0), c_fast_scalar[B * b + k + z]);
            xl[k] = (uint32_t)x;
            xh[k] = (uint32_t)(x >> 32);
            const uint32_t* __restrict__ Wt = T + 33 * k + 3 * (k * (k - 1) / 2);
            Acc6 acc;
            acc.zero();
            if (MFMA) {
                // 25 x_k + pre_k: x_k < 2^64, so the halves times 25 stay below 2^37; pre_k's halves ride along
                acc.a[0] = (uint64_t)xl[k] * 25u + (uint32_t)pre[k];
                acc.a[3] = (uint64_t)xh[k] * 25u + (uint32_t)(pre[k] >> 32);
            } else {
                acc.a[0] = (uint64_t)xl[k] * 25u;  // M[0][0] = CIRC[0] + DIAG[0]
                acc.a[3] = (uint64_t)xh[k] * 25u;
#pragma unroll
                for (int i = 0; i < 11; i++) acc.mac(sl[i], sh[i], Wt + 3 * i);
            }
#pragma unroll
            for (int j = 0; j < k; j++) acc.mac(xl[j], xh[j], Wt + 33 + 3 * j);
            s0 = acc.reduce();
        }
        s[0] = s0;
        if (MFMA) {
            // S_i + sum_k V[i][k] x_k: the second dense product of the block, the block-start state as the per-lane addend
            uint64_t o[12];
            dense_mfma<true>(xl, xh, 2 + 2 * (uint32_t)b, o, stash, stash_stride, z);
#pragma unroll
            for (int i = 0; i < 11; i++) s[i + 1] = o[i];
            continue;
        }
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

// MFMA = the full-round linear layers on the matrix pipe (mds_full_mfma).  The matrix A of that product is spread over ALL 64 lanes
// of the wave and MFMA ignores EXEC, so it may only be used where every lane of the wave is active and runs this function: the
// leaf-hash kernel, whose launches are whole waves; every other user (Merkle levels above 2^16 parents, the proof-of-work grind, single
// permutations) runs the VALU form permute<false>.
template <bool MFMA = false>
__device__ __forceinline__ void permute(uint64_t s[12], uint64_t* __restrict__ stash = nullptr, uint32_t stash_stride = 0) {
    // an opaque zero added to every table index: the tables are wave-uniform and loop-invariant, and without this the
    // compiler hoists ~650 scalar loads out of the caller's column loop, runs out of SGPRs and parks the constants in
    // VGPR lanes (v_writelane once, then a v_readlane + wait states per constant per permutation)
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_nc(s[i], c_rc[i + z]);
    const mfma_v4i afrag = MFMA ? mds_a_fragment() : mfma_v4i{0, 0, 0, 0};
#pragma unroll 1
    for (int r = 0; r < 3; r++) full_round<MFMA>(s, r, z, afrag);
    full_round3_combined<MFMA>(s, z);
    partial_rounds_blocked<MFMA>(s, z, stash, stash_stride);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::add_nc(s[i], c_rc[12 * 26 + i + z]);
#pragma unroll 1
    for (int r = 26; r < 30; r++) full_round<MFMA>(s, r, z, afrag);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}

}  // namespace poseidon
