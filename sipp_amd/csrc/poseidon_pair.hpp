// sipp_amd/csrc/poseidon_pair.hpp -- Poseidon-Goldilocks with TWO lanes per state (32 states per wave).
//
// Between one state per lane (21.3 k lane-instructions per permutation, but a 2^14-leaf tree is only 256 waves and a sponge
// over ~1000 permutations per leaf then takes 50 us per permutation) and four lanes per state (poseidon_quad.hpp: 4x the waves,
// 39.6 k lane-instructions per permutation): lane h of a pair holds elements 6h .. 6h + 5, the partner's six arrive through a DPP
// quad_perm swap.  Rotating by the pair offset gives elements (6h + d) mod 12, d = 0 .. 11, in BOTH lanes, so the circulant MDS
// uses the lane-uniform coefficient C[(d - j) mod 12] (only the DIAG[0] term is lane dependent).  In the partial rounds the
// S-box of element 0 is evaluated by both lanes (one result is redundant), the sparse row dot product is split 6 + 6 and
// summed across the pair.  Per-element constants come from the LDS copy of the tables (poseidon_quad::load_tables).
// Bit-exact with the other two layouts (tests/test_gpu_generic.py).
#pragma once
#include "poseidon_quad.hpp"

namespace poseidon_pair {

using namespace poseidon_quad;   // table layout T_*

// value held by the other lane of the pair
__device__ __forceinline__ uint32_t pair_swap32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
}
__device__ __forceinline__ uint64_t pair_swap(uint64_t v) {
    return ((uint64_t)pair_swap32((uint32_t)(v >> 32)) << 32) | pair_swap32((uint32_t)v);
}
// value held by the even lane of the pair
__device__ __forceinline__ uint64_t pair_bcast0(uint64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)v, 0xA0 /* quad_perm [0, 0, 2, 2] */, 0xf, 0xf, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(v >> 32), 0xA0, 0xf, 0xf, true);
    return ((uint64_t)hi << 32) | lo;
}

struct Gathered {
    uint32_t lo[12], hi[12];  // index d: element (6 h + d) mod 12
};
__device__ __forceinline__ void gather(const uint64_t s[6], Gathered& g) {
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const uint32_t l = (uint32_t)s[j], h = (uint32_t)(s[j] >> 32);
        g.lo[j] = l;
        g.hi[j] = h;
        g.lo[6 + j] = pair_swap32(l);
        g.hi[6 + j] = pair_swap32(h);
    }
}

// out[6 h + j] = sum_d C[(d - j) mod 12] * in[(6 h + d) mod 12]  (+ 8 in[0] for element 0)
__device__ __forceinline__ void mds_full(uint64_t s[6], uint32_t diag0 /* 8 in the even lane, else 0 */) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    Gathered g;
    gather(s, g);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int d = 0; d < 12; d++) {
            al += (uint64_t)g.lo[d] * CIRC[(d - j + 12) % 12];
            ah += (uint64_t)g.hi[d] * CIRC[(d - j + 12) % 12];
        }
        if (j == 0) {
            al += (uint64_t)g.lo[0] * diag0;
            ah += (uint64_t)g.hi[0] * diag0;
        }
        const uint64_t l = al + (ah << 32);
        const uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        s[j] = gl::reduce96_nc(h, l);
    }
}

// LDS copy of the lazy-block tables of the one-lane kernel (poseidon.hpp::partial_rounds_blocked), plus the limbs of 25 = M[0][0]
// then the limbs of the merged affine layer of round 3 (poseidon.hpp::full_round3_combined): C3[11][12][3] and its constants
constexpr int B_C25 = 2 * SIPP_POSEIDON_BLK_WORDS, B_COMB3 = B_C25 + 3, B_COMBC = B_COMB3 + 396, B_WORDS = B_COMBC + 24;
__device__ __forceinline__ void load_block_tables(uint32_t* blk) {
    for (int i = threadIdx.x; i < B_WORDS; i += blockDim.x) {
        uint32_t v;
        if (i < B_C25) v = poseidon::c_blk3[i];
        else if (i < B_COMB3) v = i == B_C25 ? 25u : 0u;
        else if (i < B_COMBC) v = poseidon::c_comb3[i - B_COMB3];
        else v = (uint32_t)(poseidon::c_comb_c[(i - B_COMBC) >> 1] >> (32 * ((i - B_COMBC) & 1)));
        blk[i] = v;
    }
    __syncthreads();
}

// Full round 3 without its own MDS: its linear layer, the first constants of the sparse form and the dense 11 x 11
// pre-multiplication are ONE affine map s -> C s + c (tools/gen_poseidon_header.py combined_layer): row 0 is the MDS row with
// its small constants, rows 1..11 are twelve lazy multiply-accumulates each, started from c.
__device__ __forceinline__ void full_round3_combined(uint64_t s[6], const uint32_t h, const uint64_t* tab, const uint32_t* blk) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    const uint32_t e0 = 6 * h;
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = poseidon::sbox_thin(gl::add_nc(s[j], tab[T_RC + 12 * 3 + e0 + j]));
    Gathered g;
    gather(s, g);
    const uint32_t* C3 = blk + B_COMB3;
    const uint32_t* CC = blk + B_COMBC;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const uint32_t e = e0 + j;
        // rows 1..11 (for the even lane's j = 0 the row index is clamped and the result replaced below)
        const uint32_t row = e ? e - 1 : 0;
        gl::Acc6 acc;
        acc.set(CC[2 * e], CC[2 * e + 1]);
#pragma unroll
        for (int d = 0; d < 12; d++) {
            const uint32_t ie = d < 6 ? e0 + d : e0 + d - 12 + (h ? 0 : 12);   // (6 h + d) mod 12
            acc.mac(g.lo[d], g.hi[d], C3 + 3 * (row * 12 + ie));
        }
        uint64_t r = acc.reduce();
        if (j == 0) {
            // element 0: row 0 of the MDS (+ DIAG[0] = 8) on the even lane, with gathered index d = element d there
            uint64_t al = (uint64_t)g.lo[0] * 8u, ah = (uint64_t)g.hi[0] * 8u;
#pragma unroll
            for (int d = 0; d < 12; d++) {
                al += (uint64_t)g.lo[d] * CIRC[d];
                ah += (uint64_t)g.hi[d] * CIRC[d];
            }
            const uint64_t l = al + (ah << 32);
            const uint32_t hh = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
            const uint64_t c0 = ((uint64_t)CC[1] << 32) | CC[0];
            const uint64_t m0 = gl::add_nc(gl::reduce96_nc(hh, l), c0);
            r = h ? r : m0;
        }
        s[j] = r;
        asm volatile("" ::: "memory");   // keep the next row's table loads behind this one (register pressure)
    }
}

// The 22 partial rounds, lazily in two blocks of 11 like the one-lane kernel: inside a block only element 0 is reduced mod p,
// everything else is (u32 half of a start-of-block value or of an x_k) x (22-bit limb of a constant) accumulated into 64-bit sums
// (gl::Acc6).  Split over the pair: a lane multiplies its OWN six start values and the x_j of its parity, the two partial sums
// meet through one DPP swap; both lanes then hold element 0 and evaluate x^7 together -- x^2 on both, x^3 on the even and x^4
// on the odd lane, swapped, x^3 x^4 on both: three products deep instead of four.  Constants are per lane (they depend on
// which six elements the lane owns), hence from LDS.
__device__ __forceinline__ void partial_rounds_blocked(uint64_t s[6], const uint32_t h, const uint64_t* tab, const uint32_t* blk) {
    constexpr int B = SIPP_POSEIDON_BLK_ROUNDS;
    const uint32_t* C25 = blk + 2 * SIPP_POSEIDON_BLK_WORDS;
#pragma unroll 1
    for (int b = 0; b < 22 / B; b++) {
        const uint32_t* T = blk + SIPP_POSEIDON_BLK_WORDS * b;
        uint32_t sl[6], sh[6], xl[B], xh[B];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            sl[j] = (uint32_t)s[j];
            sh[j] = (uint32_t)(s[j] >> 32);
        }
        uint64_t s0 = pair_bcast0(s[0]);
#pragma unroll
        for (int k = 0; k < B; k++) {
            const uint64_t x2 = SIPP_PMUL_THIN(s0, s0);
            const uint64_t y = SIPP_PMUL_THIN(x2, h ? x2 : s0);
            const uint64_t x = gl::add_nc(SIPP_PMUL_THIN(y, pair_swap(y)), tab[T_SCALAR + B * b + k]);
            xl[k] = (uint32_t)x;
            xh[k] = (uint32_t)(x >> 32);
            const uint32_t* Wt = T + 33 * k + 3 * (k * (k - 1) / 2);
            gl::Acc6 acc;
            acc.zero();
#pragma unroll
            for (int j = 0; j < 6; j++) {
                // element e = 6 h + j carries weight e - 1; the even lane's j = 0 IS element 0: it contributes 25 x_k
                const bool is0 = j == 0 && h == 0;
                const uint32_t* w = is0 ? C25 : Wt + 3 * (6 * h + j - 1);
                acc.mac(is0 ? xl[k] : sl[j], is0 ? xh[k] : sh[j], w);
            }
#pragma unroll
            for (int m = 0; 2 * m < k; m++) {
                // x_j, j < k: j = 2 m on the even lane, 2 m + 1 (if it exists) on the odd one
                const bool odd_ok = 2 * m + 1 < k;
                const uint32_t lo = h ? (odd_ok ? xl[odd_ok ? 2 * m + 1 : 0] : 0u) : xl[2 * m];
                const uint32_t hi = h ? (odd_ok ? xh[odd_ok ? 2 * m + 1 : 0] : 0u) : xh[2 * m];
                acc.mac(lo, hi, Wt + 33 + 3 * (2 * m + (odd_ok ? h : 0)));
            }
            const uint64_t part = gl::canon(acc.reduce());
            s0 = gl::add(part, pair_swap(part));
        }
        const uint32_t* V = T + 33 * B + 3 * (B * (B - 1) / 2);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const bool is0 = j == 0 && h == 0;
            gl::Acc6 acc;
            acc.set(sl[j], sh[j]);
            const uint32_t* v = V + 3 * ((is0 ? 0 : 6 * h + j - 1) * B);
#pragma unroll
            for (int k = 0; k < B; k++) acc.mac(xl[k], xh[k], v + 3 * k);
            const uint64_t r = acc.reduce();
            s[j] = is0 ? s0 : r;
        }
    }
}

// s: this lane's six elements (6 h + j); h = lane & 1; tab = LDS tables
__device__ __forceinline__ void permute(uint64_t s[6], const uint32_t h, const uint64_t* tab, const uint32_t* blk) {
    const uint32_t diag0 = h == 0 ? 8u : 0u;
    const uint32_t e0 = 6 * h;
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int j = 0; j < 6; j++) s[j] = poseidon::sbox_thin(gl::add_nc(s[j], tab[T_RC + 12 * r + e0 + j]));
        mds_full(s, diag0);
    }
    full_round3_combined(s, h, tab, blk);
    partial_rounds_blocked(s, h, tab, blk);
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int j = 0; j < 6; j++) s[j] = poseidon::sbox_thin(gl::add_nc(s[j], tab[T_RC + 12 * r + e0 + j]));
        mds_full(s, diag0);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = gl::canon(s[j]);
}

}  // namespace poseidon_pair
