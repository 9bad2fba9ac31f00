// sipp_amd/csrc/poseidon_pair.hpp -- Poseidon-Goldilocks with TWO lanes per state (32 states per wave), linear layers on the matrix pipe.
//
// For every THIN launch (the Fq12 trees, Merkle levels of at most 2^16 parents, FRI layer leaves): with one state per lane a 2^14-leaf tree
// is only 256 waves and a sponge over ~1000 permutations per leaf takes 50 us per permutation.  A lone wave per SIMD issues one
// instruction per four cycles at best, so what a thin launch pays for is the instruction count PER LANE.  (Rounds 1 - 3 also had four
// lanes per state over DPP quads: 8.1 k instructions per lane, 23 us per dependent permutation -- what this layout reaches with half the
// lanes, 8.7 k and 22 us; removed in round 4.)
//
// Layout (round 4): lanes l and l + 32 share state n = l & 31; lane (n, h) holds the six elements 6 h + j.  That is the operand layout
// of v_mfma_i32_32x32x32_i8 -- B[k][n] comes from lane (n, k >> 4), D[i][n] goes to lane (n, (i >> 2) & 1) -- so ONE instruction sees the
// whole state of 32 leaves in its K dimension, the halves' sums meet inside the matrix pipe, and every lane finds the six outputs it
// owns in its own accumulator registers: no lane exchange for any linear layer.  The unused rows / columns of the 32 x 32 x 32 shape
// carry a second product each (tools/gen_poseidon_header.py pair_tables): a full-round MDS layer is 4 instructions, a dense product
// with full 64-bit constants (byte planes x signed base-256 digits, poseidon.hpp::dense_mfma) 20 instead of 64.
//   * full rounds: S-boxes as blocks of three interleaved hand-scheduled products (gl_lazy.hpp mul3_nc), MDS on the matrix pipe with the
//     next round's constants in the chain starts;
//   * round 3's linear layer, the first constants of the sparse form and the dense pre-multiplication: ONE dense product (12 rows);
//   * the 22 partial rounds lazily in two blocks of 11 (poseidon.hpp::partial_rounds_blocked): W S and S + V x as dense products, the
//     triangular rest (x_j, j < k, times CC[k][j]) split over the pair by parity and summed through v_permlane32_swap; both lanes hold
//     element 0 and evaluate x^7 together (x^2 on both, x^3 on the even and x^4 on the odd lane, exchanged, x^3 x^4 on both).
// MFMA ignores EXEC: the kernel must run whole waves (n_leaves is a multiple of 32).  Bit-exact with the other layouts
// (tests/test_gpu_generic.py).  Round 3's form (lanes 2 i / 2 i + 1, DPP, everything on the VALU): 12.9 k instructions per lane and
// permutation; this one: see DESIGN.md section 4.
#pragma once
#include "poseidon.hpp"

namespace poseidon_pair {

using poseidon::mfma_v16i;
using poseidon::mfma_v4i;

// constant tables of the layout (generated: tools/gen_poseidon_header.py pair_tables; uploaded by sipp_poseidon_init_constants)
constexpr int PA_WORDS = 5 * SIPP_POSEIDON_PAIR_FRAGS * 64 * 4, CC_WORDS = 792, PS_WORDS = 120;
__device__ __attribute__((aligned(16))) uint32_t d_pair_a[PA_WORDS];       // dense A fragments [matrix][b / 2][lane][4]
__device__ __attribute__((aligned(16))) uint32_t d_pair_mds_a[256];        // diag(MDS, MDS) [lane][4]
__device__ uint32_t d_pair_cc[CC_WORDS];                                   // CC limbs [block][k][j 0..11][3]
__device__ uint64_t d_pair_start[PS_WORDS];                                // dense chain starts [matrix][element][L, H]

// their copy in LDS (per-lane indexed: not for scalar loads), 39 KB per block
struct Tables {
    uint64_t rc[360];         // round constants [round][element]
    uint64_t scalar[22];      // lane-0 constants of the sparse partial rounds
    // chain starts of the seven MDS layers [layer][element][low chain, high chain]: the halves of the NEXT round's constant (rounds 1, 2, 3,
    // 27, 28, 29; none behind round 29) plus 128 rowsum(MDS) 0x01010101, the repayment of the byte planes' sign flip
    uint64_t ms[7 * 12 * 2];
    uint64_t ps[PS_WORDS];
    __attribute__((aligned(16))) uint32_t pa[PA_WORDS];
    uint32_t cc[CC_WORDS];
};
// every thread of the block; ends with a barrier
__device__ __forceinline__ void load_tables(Tables& T) {
    for (int i = threadIdx.x; i < 360; i += blockDim.x) T.rc[i] = poseidon::c_rc[i];
    for (int i = threadIdx.x; i < 22; i += blockDim.x) T.scalar[i] = poseidon::c_fast_scalar[i];
    for (int i = threadIdx.x; i < 7 * 12 * 2; i += blockDim.x) {
        const int layer = i / 24, e = (i >> 1) % 12, half = i & 1;
        const uint64_t c = layer < 6 ? poseidon::c_rc[12 * (layer < 3 ? layer + 1 : layer + 24) + e] : 0;     // rounds 1..3, 27..29
        T.ms[i] = (uint64_t)(128u * (256u + (e ? 0u : 8u))) * 0x01010101ull + (half ? c >> 32 : (uint64_t)(uint32_t)c);
    }
    for (int i = threadIdx.x; i < PS_WORDS; i += blockDim.x) T.ps[i] = d_pair_start[i];
    for (int i = threadIdx.x; i < PA_WORDS; i += blockDim.x) T.pa[i] = d_pair_a[i];
    for (int i = threadIdx.x; i < CC_WORDS; i += blockDim.x) T.cc[i] = d_pair_cc[i];
    __syncthreads();
}
// this lane's words of diag(MDS, MDS)
__device__ __forceinline__ mfma_v4i mds_fragment(uint32_t lane) { return *reinterpret_cast<const mfma_v4i*>(d_pair_mds_a + 4 * lane); }

// v (this lane's) -> the even lane's and the odd lane's value, in both lanes of the pair
__device__ __forceinline__ void both32(uint32_t v, uint32_t& even, uint32_t& odd) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    even = r[0];
    odd = r[1];
}
__device__ __forceinline__ void both(uint64_t v, uint64_t& even, uint64_t& odd) {
    uint32_t el, ol, eh, oh;
    both32((uint32_t)v, el, ol);
    both32((uint32_t)(v >> 32), eh, oh);
    even = ((uint64_t)eh << 32) | el;
    odd = ((uint64_t)oh << 32) | ol;
}

// S-box layer of a full round on the lane's six elements (they carry the round's constants already): 24 products as eight blocks of
// three interleaved hand-scheduled chains (16 instructions per product against the compiler's 23; with 19 it measured 18.8 -> 18.1 ms for 2^14
// leaves x 4942 columns in round 3's layout; the one-chain block in the partial rounds: 22.0 ms, its latency is exposed there)
__device__ __forceinline__ void sbox6(uint64_t s[6]) {
#pragma unroll
    for (int g = 0; g < 2; g++) {
        uint64_t x[3], x2[3], x3[3], x4[3];
#pragma unroll
        for (int j = 0; j < 3; j++) x[j] = s[3 * g + j];
        gll::mul3_nc(x2, x, x);
        gll::mul3_nc(x3, x2, x);
        gll::mul3_nc(x4, x2, x2);
        gll::mul3_nc(x, x3, x4);
#pragma unroll
        for (int j = 0; j < 3; j++) s[3 * g + j] = x[j];
    }
}

// the eight byte planes of the lane's six words as B words: P[a][0] = bytes a of words 0 .. 3, P[a][1] = bytes a of words 4, 5 (+ two
// bytes that meet zero columns of A), made signed by ^ 0x80.  Three 4 x 4 byte transposes.
__device__ __forceinline__ void planes6(const uint32_t lo[6], const uint32_t hi[6], uint32_t P[8][2]) {
    uint32_t o[4];
    poseidon::mds_transpose4(lo, o);
#pragma unroll
    for (int b = 0; b < 4; b++) P[b][0] = o[b] ^ 0x80808080u;
    poseidon::mds_transpose4(hi, o);
#pragma unroll
    for (int b = 0; b < 4; b++) P[4 + b][0] = o[b] ^ 0x80808080u;
    const uint32_t in[4] = {lo[4], lo[5], hi[4], hi[5]};
    poseidon::mds_transpose4(in, o);
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const uint32_t x = o[b] ^ 0x80808080u;
        P[b][1] = x;
        P[4 + b][1] = x >> 16;
    }
}
__device__ __forceinline__ mfma_v4i plane_pair(const uint32_t P[8][2], int a) {
    return mfma_v4i{(int)P[a][0], (int)P[a][1], (int)P[a + 1][0], (int)P[a + 1][1]};
}
#define SIPP_PAIR_ZERO16 mfma_v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}

using poseidon::chain_start;   // d + start as one multiply-add
using poseidon::fold_chains;   // (L, H) chains -> u64

// out = MDS s + the next round's constants: register j of the product with planes (a, a + 1) is M x plane a, register 6 + j is
// M x plane a + 1 (A = diag(M, M)).  K: this lane's six pairs of chain starts (Tables::ms)
__device__ __forceinline__ void mds_pair(uint64_t s[6], const uint64_t* __restrict__ K, mfma_v4i afrag, uint32_t z) {
    uint32_t lo[6], hi[6], P[8][2];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        lo[j] = (uint32_t)s[j];
        hi[j] = (uint32_t)(s[j] >> 32);
    }
    planes6(lo, hi, P);
    const int32_t p8 = (int32_t)(256u + z), p16 = (int32_t)(65536u + z), p24 = (int32_t)(16777216u + z), p1 = (int32_t)(1u + z);
    int64_t al[6], ah[6];
    // the wave issues in order and the matrix pipe takes one product per 32 cycles: the folds of product q sit between the issue of
    // products q + 1 and q + 2 (a lone wave per SIMD has nobody else to fill the wait)
    mfma_v16i d[4];
    d[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, plane_pair(P, 0), SIPP_PAIR_ZERO16, 0, 0, 0);
    d[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, plane_pair(P, 2), SIPP_PAIR_ZERO16, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) {
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < 4) d[q + 2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, plane_pair(P, 2 * q + 4), SIPP_PAIR_ZERO16, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            int64_t& acc = q < 2 ? al[j] : ah[j];
            if ((q & 1) == 0) {
                acc = chain_start(d[q][j], K[2 * j + (q >> 1)], p1);
                acc = (int64_t)d[q][6 + j] * (int64_t)p8 + acc;
            } else {
                acc = (int64_t)d[q][j] * (int64_t)p16 + acc;
                acc = (int64_t)d[q][6 + j] * (int64_t)p24 + acc;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = fold_chains((uint64_t)al[j], (uint64_t)ah[j]);
}

// out[j] = sum_e M[6 h + j][e] x_e + const (+ addend_j): one of the five dense constant products (tools/gen_poseidon_header.py
// pair_matrices).  Digit sums D_t, t = a + b, as in poseidon.hpp::dense_mfma; here the chain of planes (a, a + 1), a = 0, 2, 4, 6, against
// fragment b = t - a leaves D_t in registers j and D_(t+1) in registers 6 + j: 20 instructions, 8 chains.
template <bool ADDEND>
__device__ __forceinline__ void dense_pair(const uint32_t lo[6], const uint32_t hi[6], uint32_t mat, uint64_t out[6], const Tables& T, uint32_t lane,
                                           uint32_t e0, const uint32_t* addlo, const uint32_t* addhi, uint32_t z) {
    uint32_t P[8][2];
    planes6(lo, hi, P);
    mfma_v4i afr[SIPP_POSEIDON_PAIR_FRAGS];
    {
        const uint32_t* __restrict__ A = T.pa + mat * (SIPP_POSEIDON_PAIR_FRAGS * 256) + lane * 4;
#pragma unroll
        for (int f = 0; f < SIPP_POSEIDON_PAIR_FRAGS; f++) afr[f] = *reinterpret_cast<const mfma_v4i*>(A + f * 256);
    }
    const int32_t p8 = (int32_t)(256u + z), p16 = (int32_t)(65536u + z), p24 = (int32_t)(16777216u + z);
    const int32_t n1 = (int32_t)(0xffffffffu + z), n8 = -p8, n16 = -p16, n24 = -p24, p1 = (int32_t)(1u + z);
    const uint64_t* __restrict__ K = T.ps + (mat * 12 + e0) * 2;
    int64_t L[6], H[6];
    // chain t0 (digit sums t0 and t0 + 1) = the products of planes (a, a + 1), a = 0, 2, 4, 6, with fragment b = t0 - a in 0 .. 8.  The wave
    // issues in order and the matrix pipe takes a product per 32 cycles: chain t0 + 2 is issued BETWEEN the folds of chain t0, one product
    // per group of folds (a lone wave per SIMD has nobody else to fill the wait)
    auto product = [&](mfma_v16i& d, int t0, int i, bool first) {     // the i-th product of chain t0
        int n = 0;
#pragma unroll
        for (int a = 0; a < 8; a += 2) {
            const int b = t0 - a;
            if (b < 0 || b > 8) continue;
            if (n == i) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[b / 2], plane_pair(P, a), first ? SIPP_PAIR_ZERO16 : d, 0, 0, 0);
            n++;
        }
    };
    auto products_of = [](int t0) { return t0 > 14 ? 0 : t0 <= 6 ? t0 / 2 + 1 : t0 == 8 ? 4 : (16 - t0) / 2; };
    auto fold = [&](const mfma_v16i& d, int t, int j) {
        const int32_t pw = (t & 3) == 0 ? p1 : (t & 3) == 1 ? p8 : (t & 3) == 2 ? p16 : p24;
        const int32_t nw = (t & 3) == 0 ? n1 : (t & 3) == 1 ? n8 : (t & 3) == 2 ? n16 : n24;
        const int32_t dv = d[6 * (t & 1) + j];
        if (t == 0) {
            uint64_t l0 = K[2 * j], h0 = K[2 * j + 1];      // chain starts (positive: + 2^50)
            if (ADDEND) {
                l0 += addlo[j];
                h0 += addhi[j];
            }
            L[j] = chain_start(dv, l0, p1);
            H[j] = (int64_t)h0;
        } else if (t < 4) {
            L[j] = (int64_t)dv * (int64_t)pw + L[j];
        } else if (t < 8) {
            H[j] = (int64_t)dv * (int64_t)pw + H[j];
        } else if (t < 12) {
            H[j] = (int64_t)dv * (int64_t)pw + H[j];
            L[j] = (int64_t)dv * (int64_t)nw + L[j];
        } else {
            L[j] = (int64_t)dv * (int64_t)nw + L[j];
        }
        // fold here, one D at a time (the compiler would otherwise reassociate and keep every accumulator alive)
        asm volatile("" : "+v"(L[j]), "+v"(H[j]));
    };
    mfma_v16i dd[2];
    product(dd[0], 0, 0, true);
#pragma unroll
    for (int t0 = 0; t0 < 16; t0 += 2) {
        mfma_v16i& cur = dd[(t0 >> 1) & 1];
        mfma_v16i& nxt = dd[((t0 >> 1) + 1) & 1];
        const int n = products_of(t0 + 2);
        const int items = t0 == 14 ? 6 : 12;                 // (t, j) folds of this chain: t0 first, then t0 + 1
        int done = 0;
#pragma unroll
        for (int i = 0; i < (n ? n : 1); i++) {
            __builtin_amdgcn_sched_barrier(0);
            if (n) product(nxt, t0 + 2, i, i == 0);
            __builtin_amdgcn_sched_barrier(0);
            const int upto = items * (i + 1) / (n ? n : 1);
#pragma unroll
            for (int it = done; it < upto; it++) fold(cur, t0 + it / 6, it % 6);
            done = upto;
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) out[j] = fold_chains((uint64_t)L[j], (uint64_t)H[j]);   // both chains in (0, 2^52)
}

// the partial rounds' chain is one dependent product after the other: the compiler's four multiply-adds (they overlap) with the
// hand-scheduled reduction behind them
__device__ __forceinline__ uint64_t mul_chain(uint64_t a, uint64_t b) {
    uint64_t hi, lo;
    gl::mul_wide(a, b, hi, lo);
    return gll::reduce128_nc(hi, lo);
}
// gl::Acc6::reduce with that reduction
__device__ __forceinline__ uint64_t acc6_reduce(const gl::Acc6& acc) {
    uint32_t l[4], h[4], v[5];
    gl::Acc6::fold3(l, acc.a[0], acc.a[1], acc.a[2]);
    gl::Acc6::fold3(h, acc.a[3], acc.a[4], acc.a[5]);
    uint32_t c = 0;
    v[0] = l[0];
    v[1] = __builtin_addc(l[1], h[0], c, &c);
    v[2] = __builtin_addc(l[2], h[1], c, &c);
    v[3] = __builtin_addc(l[3], h[2], c, &c);
    v[4] = h[3] + c;
    const uint64_t r = gll::reduce128_nc(((uint64_t)v[3] << 32) | v[2], ((uint64_t)v[1] << 32) | v[0]);
    const uint64_t t = (uint64_t)v[4] << 32;  // 2^128 = -2^32 (mod p)
    const uint64_t d = r - t;
    return r < t ? d - gl::EPS : d;
}

// The 22 partial rounds, lazily in two blocks of 11 (the algebra: poseidon.hpp::partial_rounds_blocked).  s0: element 0, held by both
// lanes.  A lane keeps the x_k of its parity (k = 2 m + h): they are its inputs of the V product and its share of the triangular sums.
__device__ __forceinline__ void partial_rounds_blocked(uint64_t s[6], const uint32_t h, const uint32_t lane, const Tables& T, uint32_t z) {
    constexpr int B = SIPP_POSEIDON_BLK_ROUNDS;
    static_assert(B == 11, "the matrix-pipe form is laid out for blocks of eleven rounds");
    const uint32_t e0 = 6 * h;
    const uint32_t c25 = h ? 0u : 25u;           // M[0][0] x_k enters the even lane's sum
#pragma unroll 1
    for (int b = 0; b < 22 / B; b++) {
        uint32_t sl[6], sh[6], xl[6], xh[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            sl[j] = (uint32_t)s[j];
            sh[j] = (uint32_t)(s[j] >> 32);
            xl[j] = xh[j] = 0;
        }
        uint64_t s0, unused;
        both(s[0], s0, unused);
        // pre_k = sum_i W[k][i] S_i, k = 6 h + j (the column of element 0 is zero: the even lane's s[0] does not enter)
        uint64_t pre[6];
        dense_pair<false>(sl, sh, 1 + 2 * (uint32_t)b, pre, T, lane, e0, nullptr, nullptr, z);
        const uint32_t* __restrict__ CC = T.cc + (b * B) * 36 + 3 * h;
#pragma unroll
        for (int k = 0; k < B; k++) {
            const uint64_t x2 = mul_chain(s0, s0);
            const uint64_t y = mul_chain(x2, h ? x2 : s0);
            uint64_t y3, y4;
            both(y, y3, y4);
            const uint64_t x = gl::add_nc(mul_chain(y3, y4), T.scalar[B * b + k]);
            const uint32_t xlo = (uint32_t)x, xhi = (uint32_t)(x >> 32);
            gl::Acc6 acc;
            acc.zero();
            // 25 x_k on the even lane, pre_k on the lane that owns it
            const bool mine = (k < 6) == (h == 0);
            const uint64_t pk = pre[k % 6];
            acc.a[0] = (uint64_t)xlo * c25 + (mine ? (uint32_t)pk : 0u);
            acc.a[3] = (uint64_t)xhi * c25 + (mine ? (uint32_t)(pk >> 32) : 0u);
#pragma unroll
            for (int m = 0; 2 * m < k; m++) acc.mac(xl[m], xh[m], CC + (k * 12 + 2 * m) * 3);     // x_(2 m + h); CC[k][j >= k] = 0
            // keep x_k: slot k >> 1 of the lane of parity k & 1 (an even k may overwrite the odd lane's slot: x_(k+1) follows)
            if ((k & 1) == 0) {
                xl[k >> 1] = xlo;
                xh[k >> 1] = xhi;
            } else {
                xl[k >> 1] = h ? xlo : xl[k >> 1];
                xh[k >> 1] = h ? xhi : xh[k >> 1];
            }
            const uint64_t part = gll::canon(acc6_reduce(acc));
            uint64_t pe, po;
            both(part, pe, po);
            s0 = gll::add_nc(pe, po);
        }
        // S_e + sum_k V[e][k] x_k (+ the constants of full round 26 behind the last block), the block-start state as the addend
        uint64_t o[6];
        dense_pair<true>(xl, xh, 2 + 2 * (uint32_t)b, o, T, lane, e0, sl, sh, z);
#pragma unroll
        for (int j = 0; j < 6; j++) s[j] = o[j];
        s[0] = h ? o[0] : s0;
    }
}

// s: this lane's six elements (6 h + j); lane = lane of the wave; h = lane >> 5; afrag = this lane's words of SIPP_POSEIDON_PAIR_MDS_A
__device__ __forceinline__ void permute(uint64_t s[6], const uint32_t lane, const Tables& T, mfma_v4i afrag) {
    uint32_t z = 0;
    asm volatile("" : "+s"(z));
    const uint32_t h = lane >> 5, e0 = 6 * h;
    const uint64_t* __restrict__ rc = T.rc + e0;
    const uint64_t* __restrict__ ms = T.ms + 2 * e0;
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = gl::add_nc(s[j], rc[j]);
#pragma unroll 1
    for (int r = 0; r < 3; r++) {
        sbox6(s);
        mds_pair(s, ms + 24 * r, afrag, z);
    }
    {   // full round 3: its MDS, the first constants of the sparse form and the dense pre-multiplication are ONE affine map
        sbox6(s);
        uint32_t lo[6], hi[6];
#pragma unroll
        for (int j = 0; j < 6; j++) {
            lo[j] = (uint32_t)s[j];
            hi[j] = (uint32_t)(s[j] >> 32);
        }
        dense_pair<false>(lo, hi, 0, s, T, lane, e0, nullptr, nullptr, z);
    }
    partial_rounds_blocked(s, h, lane, T, z);
    s[0] = h ? s[0] : gl::add_nc(s[0], rc[12 * 26]);      // element 0 left the blocks without round 26's constant
#pragma unroll 1
    for (int r = 26; r < 30; r++) {
        sbox6(s);
        mds_pair(s, ms + 24 * (r - 23), afrag, z);      // layers 3 .. 6
    }
#pragma unroll
    for (int j = 0; j < 6; j++) s[j] = gl::canon(s[j]);
}

}  // namespace poseidon_pair
