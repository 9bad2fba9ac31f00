// sipp_amd/csrc/poseidon_pair.cuh -- Poseidon-Goldilocks with TWO lanes per state (32 states per wave).
//
// Between one state per lane (21.3 k lane-instructions per permutation, but a 2^14-leaf tree is only 256 waves and a sponge
// over ~1000 permutations per leaf then takes 50 us per permutation) and four lanes per state (poseidon_quad.cuh: 4x the waves,
// 39.6 k lane-instructions per permutation): lane h of a pair holds elements 6h .. 6h + 5, the partner's six arrive through a DPP
// quad_perm swap.  Rotating by the pair offset gives elements (6h + d) mod 12, d = 0 .. 11, in BOTH lanes, so the circulant MDS
// uses the lane-uniform coefficient C[(d - j) mod 12] (only the DIAG[0] term is lane dependent).  In the partial rounds the
// S-box of element 0 is evaluated by both lanes (one result is redundant), the sparse row dot product is split 6 + 6 and
// summed across the pair.  Per-element constants come from the LDS copy of the tables (poseidon_quad::load_tables).
// Bit-exact with the other two layouts (tests/test_gpu_generic.py).
#pragma once
#include "poseidon_quad.cuh"

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

// LDS copy of the lazy-block tables of the one-lane kernel (poseidon.cuh::partial_rounds_blocked), plus the limbs of 25 = M[0][0]
constexpr int B_WORDS = 2 * SIPP_POSEIDON_BLK_WORDS + 3;
__device__ __forceinline__ void load_block_tables(uint32_t* blk) {
    for (int i = threadIdx.x; i < B_WORDS; i += blockDim.x)
        blk[i] = i < 2 * SIPP_POSEIDON_BLK_WORDS ? poseidon::c_blk3[i] : (i == 2 * SIPP_POSEIDON_BLK_WORDS ? 25u : 0u);
    __syncthreads();
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
            const uint64_t x2 = gl::mul_nc(s0, s0);
            const uint64_t y = gl::mul_nc(x2, h ? x2 : s0);
            const uint64_t x = gl::add_nc(gl::mul_nc(y, pair_swap(y)), tab[T_SCALAR + B * b + k]);
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
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int j = 0; j < 6; j++) s[j] = poseidon::sbox(gl::add_nc(s[j], tab[T_RC + 12 * r + e0 + j]));
        mds_full(s, diag0);
    }
    // ---- partial rounds, sparse form ----
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = gl::add_nc(s[j], tab[T_FIRST + e0 + j]);
    {
        // dense pre-multiplication of elements 1..11 (element 0 passes through)
        Gathered g;
        gather(s, g);
        uint64_t t[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t e = e0 + j;                // output element
            const uint32_t row = e == 0 ? 0 : e - 1;  // row of the 11 x 11 block (the even lane's j = 0 is discarded)
            poseidon::Acc160 acc;
#pragma unroll
            for (int d = 0; d < 12; d++) {
                const uint32_t ie = (e0 + d) % 12;    // input element; element 0 has coefficient 0
                const uint64_t cv = tab[T_MI + row * 11 + (ie ? ie - 1 : 0)];
                acc.mac(((uint64_t)g.hi[d] << 32) | g.lo[d], ie ? cv : 0);
            }
            t[j] = acc.reduce();
            asm volatile("" ::: "memory");   // keep the next output's table loads behind this one (register pressure)
        }
        if (h != 0) s[0] = t[0];
#pragma unroll
        for (int j = 1; j < 6; j++) s[j] = t[j];
    }
    partial_rounds_blocked(s, h, tab, blk);
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
#pragma unroll
        for (int j = 0; j < 6; j++) s[j] = poseidon::sbox(gl::add_nc(s[j], tab[T_RC + 12 * r + e0 + j]));
        mds_full(s, diag0);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = gl::canon(s[j]);
}

}  // namespace poseidon_pair
