// sipp_amd/csrc/host_poseidon.cpp -- the host-side Poseidon-Goldilocks permutation of the Fiat-Shamir challenger.
//
// Replaces (host side) plonky2's PoseidonHash permutation under iop/challenger.rs @ 541e127 (reference Cargo.toml:21;
// `C = PoseidonGoldilocksConfig` at reference src/verifier_circuit.rs:196).  The challenger is a SEQUENTIAL sponge: observing
// the ~32 k opening words of the widest STARK is ~4 k dependent permutations on the proof's critical path, so what counts
// here is the latency of ONE permutation on ONE core (a GPU wave needs 18-50 us per dependent permutation).
//
// Implementations, bit-identical (tests/test_abi.py holds them against each other and against the CPU oracle):
//  * permute_scalar: portable C++ (u128 products), plonky2's evaluation order;
//  * look-ahead partial rounds: the sparse form with a one-round look-ahead, so that the chain s0 -> x^7 -> next s0 carries
//    one small product per round and the eleven-term dot products / state updates run beside it;
//  * AVX-512: the 8 full rounds on two 512-bit registers (vpmuludq products; S-box and circulant MDS for the 12 words at
//    once), optionally the state update of the partial rounds too.  Chosen at load time when the CPU has AVX-512F/DQ/VL
//    (the build box and the GPU box both have).
#include "host_poseidon.hpp"

#include <immintrin.h>
#include <stdlib.h>

#include "gl.hpp"
#include "poseidon_constants.h"

namespace host {
namespace {

typedef unsigned __int128 u128;

inline uint64_t red128(u128 v) {  // -> [0, 2^64), congruent, not canonical
    const uint64_t lo = (uint64_t)v, hi = (uint64_t)(v >> 64);
    const uint64_t hh = hi >> 32, hl = hi & gl::EPS;
    uint64_t t0 = lo - hh;
    if (lo < hh) t0 -= gl::EPS;
    const uint64_t t1 = (hl << 32) - hl;
    uint64_t r = t0 + t1;
    if (r < t1) r += gl::EPS;
    return r;
}
// v - c * 2^32 for any u64 v (2^128 = -2^32 mod p: c lost carries of a u128 accumulator)
inline uint64_t sub_carries(uint64_t v, uint32_t c) {
    const uint64_t k = (uint64_t)c << 32, d = v - k;
    return v < k ? d - gl::EPS : d;
}
inline uint64_t sbox7(uint64_t x) {
    const uint64_t x2 = red128((u128)x * x), x3 = red128((u128)x2 * x), x4 = red128((u128)x2 * x2);
    return red128((u128)x3 * x4);
}
const uint64_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

inline void full_round_scalar(uint64_t s[12], int rnd) {
    uint64_t lo[24], hi[24];
    for (int i = 0; i < 12; i++) {
        const uint64_t t = sbox7(gl::add_nc(s[i], SIPP_POSEIDON_RC[12 * rnd + i]));
        lo[i] = lo[i + 12] = t & gl::EPS;
        hi[i] = hi[i + 12] = t >> 32;
    }
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
        for (int i = 0; i < 12; i++) {
            al += lo[i + r] * CIRC[i];
            ah += hi[i + r] * CIRC[i];
        }
        if (r == 0) {
            al += lo[0] * 8;
            ah += hi[0] * 8;
        }
        s[r] = red128((u128)al + ((u128)ah << 32));
    }
}

// the sparse form of the 22 partial rounds exactly as plonky2 evaluates it (reference implementation of this file)
inline void partial_rounds_scalar(uint64_t s[12]) {
    for (int i = 0; i < 12; i++) s[i] = gl::add_nc(s[i], SIPP_POSEIDON_FAST_FIRST[i]);
    {
        uint64_t t[11];
        for (int i = 0; i < 11; i++) {
            u128 acc = 0;
            uint32_t c = 0;
            for (int j = 0; j < 11; j++) {
                const u128 pr = (u128)s[j + 1] * SIPP_POSEIDON_FAST_MI[i * 11 + j];
                acc += pr;
                c += acc < pr;
            }
            t[i] = sub_carries(red128(acc), c);
        }
        for (int i = 0; i < 11; i++) s[i + 1] = t[i];
    }
    for (int r = 0; r < 22; r++) {
        const uint64_t x = gl::add_nc(sbox7(s[0]), SIPP_POSEIDON_FAST_SCALAR[r]);
        u128 acc = (u128)x * 25;
        uint32_t c = 0;
        for (int i = 0; i < 11; i++) {
            const u128 pr = (u128)s[i + 1] * SIPP_POSEIDON_FAST_WHAT[r * 11 + i];
            acc += pr;
            c += acc < pr;
        }
        for (int i = 0; i < 11; i++) s[i + 1] = red128((u128)x * SIPP_POSEIDON_FAST_VS[r * 11 + i] + s[i + 1]);
        s[0] = sub_carries(red128(acc), c);
    }
}

void permute_scalar(uint64_t s[12]) {
    for (int r = 0; r < 4; r++) full_round_scalar(s, r);
    partial_rounds_scalar(s);
    for (int r = 26; r < 30; r++) full_round_scalar(s, r);
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}

// ---- partial rounds with a one-round look-ahead -------------------------------------------------------------------------
// Round r of the sparse form: x_r = (s0)^7 + c_r;  s0' = 25 x_r + W_r . v;  v' = v + x_r VS_r   (v = words 1..11).
// W_r . v_r = W_r . v_{r-1} + x_{r-1} (W_r . VS_{r-1}):  D_r = W_r . v_{r-1} needs only x_{r-2}, so the chain
// s0 -> x_r -> s0' carries ONE product (25 x_r) and one reduction; D_r, the K-term and the update of v run beside it.
struct Lookahead {
    uint64_t K[22];  // K[r] = W_r . VS_{r-1} (r >= 1), canonical
    Lookahead() {
        K[0] = 0;
        for (int r = 1; r < 22; r++) {
            uint64_t acc = 0;
            for (int i = 0; i < 11; i++)
                acc = gl::add(acc, gl::mul(SIPP_POSEIDON_FAST_WHAT[r * 11 + i], SIPP_POSEIDON_FAST_VS[(r - 1) * 11 + i]));
            K[r] = acc;
        }
    }
};
const Lookahead LA;

// 192-bit accumulator of u128 products (no carry is ever lost for fewer than 2^64 terms)
struct Acc192 {
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    inline void mac(uint64_t a, uint64_t b) {
        const u128 p = (u128)a * b;
        const u128 t = (u128)w0 + (uint64_t)p;
        w0 = (uint64_t)t;
        const u128 t1 = (u128)w1 + (uint64_t)(p >> 64) + (uint64_t)(t >> 64);
        w1 = (uint64_t)t1;
        w2 += (uint64_t)(t1 >> 64);
    }
    inline void add64(uint64_t a) {
        const u128 t = (u128)w0 + a;
        w0 = (uint64_t)t;
        const u128 t1 = (u128)w1 + (uint64_t)(t >> 64);
        w1 = (uint64_t)t1;
        w2 += (uint64_t)(t1 >> 64);
    }
    // w2 < 2^32 here: value = w0 + 2^64 w1 + 2^128 w2, 2^128 = -2^32 mod p
    inline uint64_t reduce() const { return sub_carries(red128(((u128)w1 << 64) | w0), (uint32_t)w2); }
};

inline void partial_rounds_lookahead(uint64_t s[12]) {
    for (int i = 0; i < 12; i++) s[i] = gl::add_nc(s[i], SIPP_POSEIDON_FAST_FIRST[i]);
    uint64_t v[11];
    for (int i = 0; i < 11; i++) {
        Acc192 a;
        for (int j = 0; j < 11; j++) a.mac(s[j + 1], SIPP_POSEIDON_FAST_MI[i * 11 + j]);
        v[i] = a.reduce();
    }
    // v holds v_{r-1} while round r runs its chain; x_prev = x_{r-1} is folded into v after D_r has been taken
    uint64_t a0 = s[0];
    uint64_t x_prev = 0;
    for (int r = 0; r < 22; r++) {
        // off the chain: D_r + x_{r-1} K_r as one 192-bit sum (v = v_{r-1}, not yet updated with x_{r-1})
        Acc192 e;
        for (int i = 0; i < 11; i++) e.mac(v[i], SIPP_POSEIDON_FAST_WHAT[r * 11 + i]);
        e.mac(x_prev, LA.K[r]);
        // off the chain: v_r = v_{r-1} + x_{r-1} VS_{r-1}
        if (r > 0)
            for (int i = 0; i < 11; i++) v[i] = red128((u128)x_prev * SIPP_POSEIDON_FAST_VS[(r - 1) * 11 + i] + v[i]);
        // the chain
        const uint64_t x = gl::add_nc(sbox7(a0), SIPP_POSEIDON_FAST_SCALAR[r]);
        e.mac(x, 25);
        a0 = e.reduce();
        x_prev = x;
    }
    for (int i = 0; i < 11; i++) s[i + 1] = red128((u128)x_prev * SIPP_POSEIDON_FAST_VS[21 * 11 + i] + v[i]);
    s[0] = a0;
}

// ---- 512-bit registers: the full rounds, and the state update of the partial rounds -------------------------------------
// (section title of the full rounds follows)
// ---- full rounds on two 512-bit registers ------------------------------------------------------------------------------
#define SIPP_AVX512 __attribute__((target("avx512f,avx512dq,avx512vl")))

struct alignas(64) MdsTables {
    // contribution of input word j to the output words: CR[j][r] = CIRC[(j - r) mod 12] (+ 8 for j = r = 0); words 0..7 in
    // [j][0], words 8..11 in [j][1] (upper four lanes zero)
    uint64_t CR[12][2][8];
    uint64_t RC[30][2][8];  // round constants in the same two-register layout
    MdsTables() {
        for (int j = 0; j < 12; j++)
            for (int r = 0; r < 16; r++) {
                uint64_t c = r < 12 ? CIRC[((j - r) % 12 + 12) % 12] : 0;
                if (j == 0 && r == 0) c += 8;
                CR[j][r >> 3][r & 7] = c;
            }
        for (int rnd = 0; rnd < 30; rnd++)
            for (int r = 0; r < 16; r++) RC[rnd][r >> 3][r & 7] = r < 12 ? SIPP_POSEIDON_RC[12 * rnd + r] : 0;
    }
};
const MdsTables MT;

SIPP_AVX512 inline __m512i v_add_nc(__m512i s, __m512i c) {  // s any, c canonical -> [0, 2^64)
    const __m512i r = _mm512_add_epi64(s, c);
    const __mmask8 m = _mm512_cmplt_epu64_mask(r, c);
    return _mm512_mask_add_epi64(r, m, r, _mm512_set1_epi64((long long)gl::EPS));
}
SIPP_AVX512 inline __m512i v_red128(__m512i hi, __m512i lo) {
    const __m512i eps = _mm512_set1_epi64((long long)gl::EPS);
    const __m512i hh = _mm512_srli_epi64(hi, 32);
    __m512i t0 = _mm512_sub_epi64(lo, hh);
    const __mmask8 b = _mm512_cmplt_epu64_mask(lo, hh);
    t0 = _mm512_mask_sub_epi64(t0, b, t0, eps);
    const __m512i t1 = _mm512_sub_epi64(_mm512_slli_epi64(hi, 32), _mm512_and_si512(hi, eps));  // (low half of hi) (2^32 - 1)
    const __m512i r = _mm512_add_epi64(t0, t1);
    const __mmask8 c = _mm512_cmplt_epu64_mask(r, t1);
    return _mm512_mask_add_epi64(r, c, r, eps);
}
SIPP_AVX512 inline void v_mul_wide(__m512i a, __m512i b, __m512i& hi_out, __m512i& lo_out) {
    const __m512i m32 = _mm512_set1_epi64((long long)gl::EPS);
    const __m512i ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
    const __m512i ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b),
                  hh = _mm512_mul_epu32(ah, bh);
    const __m512i t = _mm512_add_epi64(hl, _mm512_srli_epi64(ll, 32));
    const __m512i u = _mm512_add_epi64(lh, _mm512_and_si512(t, m32));
    const __m512i hi = _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(t, 32)), _mm512_srli_epi64(u, 32));
    hi_out = hi;
    lo_out = _mm512_add_epi64(ll, _mm512_slli_epi64(_mm512_add_epi64(lh, hl), 32));  // a b mod 2^64
}
SIPP_AVX512 inline __m512i v_mul(__m512i a, __m512i b) {
    __m512i hi, lo;
    v_mul_wide(a, b, hi, lo);
    return v_red128(hi, lo);
}
// a b + c, all words any u64
SIPP_AVX512 inline __m512i v_mad(__m512i a, __m512i b, __m512i c) {
    __m512i hi, lo;
    v_mul_wide(a, b, hi, lo);
    const __m512i lo2 = _mm512_add_epi64(lo, c);
    const __mmask8 k = _mm512_cmplt_epu64_mask(lo2, c);
    hi = _mm512_mask_add_epi64(hi, k, hi, _mm512_set1_epi64(1));  // a b <= (2^64 - 1)^2: hi + 1 does not wrap
    return v_red128(hi, lo2);
}
SIPP_AVX512 inline __m512i v_sbox7(__m512i x) {
    const __m512i x2 = v_mul(x, x), x3 = v_mul(x2, x), x4 = v_mul(x2, x2);
    return v_mul(x3, x4);
}
// sum of lo products al (< 2^42) and of hi products ah (< 2^42): al + 2^32 ah mod p
SIPP_AVX512 inline __m512i v_mds_combine(__m512i al, __m512i ah) {
    const __m512i eps = _mm512_set1_epi64((long long)gl::EPS);
    const __m512i t = _mm512_slli_epi64(ah, 32);  // drops ah >> 32 (< 2^10), added back times 2^64 = eps
    __m512i r = _mm512_add_epi64(al, t);
    const __mmask8 c = _mm512_cmplt_epu64_mask(r, t);
    r = _mm512_mask_add_epi64(r, c, r, eps);
    const __m512i top = _mm512_mul_epu32(_mm512_srli_epi64(ah, 32), eps);  // < 2^42
    const __m512i r2 = _mm512_add_epi64(r, top);
    const __mmask8 c2 = _mm512_cmplt_epu64_mask(r2, top);
    return _mm512_mask_add_epi64(r2, c2, r2, eps);
}
SIPP_AVX512 inline void full_round_avx512(__m512i& A, __m512i& B, int rnd) {
    A = v_sbox7(v_add_nc(A, _mm512_load_si512((const void*)MT.RC[rnd][0])));
    B = v_sbox7(v_add_nc(B, _mm512_load_si512((const void*)MT.RC[rnd][1])));
    const __m512i z = _mm512_setzero_si512();
    __m512i alA[3] = {z, z, z}, ahA[3] = {z, z, z}, alB[3] = {z, z, z}, ahB[3] = {z, z, z};  // three chains of four adds
#pragma GCC unroll 12
    for (int j = 0; j < 12; j++) {
        const __m512i src = j < 8 ? A : B;
        const __m512i bj = _mm512_permutexvar_epi64(_mm512_set1_epi64(j & 7), src);  // word j in every lane
        const __m512i bh = _mm512_srli_epi64(bj, 32);
        const __m512i cA = _mm512_load_si512((const void*)MT.CR[j][0]), cB = _mm512_load_si512((const void*)MT.CR[j][1]);
        alA[j % 3] = _mm512_add_epi64(alA[j % 3], _mm512_mul_epu32(bj, cA));
        ahA[j % 3] = _mm512_add_epi64(ahA[j % 3], _mm512_mul_epu32(bh, cA));
        alB[j % 3] = _mm512_add_epi64(alB[j % 3], _mm512_mul_epu32(bj, cB));
        ahB[j % 3] = _mm512_add_epi64(ahB[j % 3], _mm512_mul_epu32(bh, cB));
    }
    A = v_mds_combine(_mm512_add_epi64(_mm512_add_epi64(alA[0], alA[1]), alA[2]), _mm512_add_epi64(_mm512_add_epi64(ahA[0], ahA[1]), ahA[2]));
    B = v_mds_combine(_mm512_add_epi64(_mm512_add_epi64(alB[0], alB[1]), alB[2]), _mm512_add_epi64(_mm512_add_epi64(ahB[0], ahB[1]), ahB[2]));
}

struct alignas(64) PartialTables {
    uint64_t VS[22][2][8];  // VS_r in the two-register layout of words 1..11 (word 1 + i in lane i; lanes 11..15 zero)
    PartialTables() {
        for (int r = 0; r < 22; r++)
            for (int i = 0; i < 16; i++) VS[r][i >> 3][i & 7] = i < 11 ? SIPP_POSEIDON_FAST_VS[r * 11 + i] : 0;
    }
};
const PartialTables PT;

// partial_rounds_lookahead with the update v += x VS_r (eleven multiply-adds mod p) on two 512-bit registers: the scalar
// pipes keep the chain and the dot product, the vector pipes the state.  buf: 16 words, 64-byte aligned.
SIPP_AVX512 inline void partial_rounds_avx512(uint64_t* buf) {
    for (int i = 0; i < 12; i++) buf[i] = gl::add_nc(buf[i], SIPP_POSEIDON_FAST_FIRST[i]);
    alignas(64) uint64_t v[16];
    for (int i = 0; i < 11; i++) {
        Acc192 a;
        for (int j = 0; j < 11; j++) a.mac(buf[j + 1], SIPP_POSEIDON_FAST_MI[i * 11 + j]);
        v[i] = a.reduce();
    }
    for (int i = 11; i < 16; i++) v[i] = 0;
    __m512i VA = _mm512_load_si512((const void*)v), VB = _mm512_load_si512((const void*)(v + 8));
    uint64_t a0 = buf[0], x_prev = 0;
    for (int r = 0; r < 22; r++) {
        Acc192 e;
        for (int i = 0; i < 11; i++) e.mac(v[i], SIPP_POSEIDON_FAST_WHAT[r * 11 + i]);
        e.mac(x_prev, LA.K[r]);
        if (r > 0) {
            const __m512i xb = _mm512_set1_epi64((long long)x_prev);
            VA = v_mad(xb, _mm512_load_si512((const void*)PT.VS[r - 1][0]), VA);
            VB = v_mad(xb, _mm512_load_si512((const void*)PT.VS[r - 1][1]), VB);
            _mm512_store_si512((void*)v, VA);
            _mm512_store_si512((void*)(v + 8), VB);
        }
        const uint64_t x = gl::add_nc(sbox7(a0), SIPP_POSEIDON_FAST_SCALAR[r]);
        e.mac(x, 25);
        a0 = e.reduce();
        x_prev = x;
    }
    const __m512i xb = _mm512_set1_epi64((long long)x_prev);
    VA = v_mad(xb, _mm512_load_si512((const void*)PT.VS[21][0]), VA);
    VB = v_mad(xb, _mm512_load_si512((const void*)PT.VS[21][1]), VB);
    buf[0] = a0;
    _mm512_storeu_si512((void*)(buf + 1), VA);
    _mm512_mask_storeu_epi64((void*)(buf + 9), 0x07, VB);
}

template <bool VEC_PARTIAL>
SIPP_AVX512 inline void permute_avx512_t(uint64_t s[12]) {
    alignas(64) uint64_t buf[16];
    __m512i A = _mm512_loadu_si512((const void*)s);
    __m512i B = _mm512_maskz_loadu_epi64(0x0f, (const void*)(s + 8));
    for (int r = 0; r < 4; r++) full_round_avx512(A, B, r);
    _mm512_store_si512((void*)buf, A);
    _mm512_store_si512((void*)(buf + 8), B);
    if (VEC_PARTIAL)
        partial_rounds_avx512(buf);
    else
        partial_rounds_lookahead(buf);
    A = _mm512_load_si512((const void*)buf);
    B = _mm512_maskz_load_epi64(0x0f, (const void*)(buf + 8));
    for (int r = 26; r < 30; r++) full_round_avx512(A, B, r);
    // canonical words out
    const __m512i p = _mm512_set1_epi64((long long)gl::P);
    A = _mm512_mask_sub_epi64(A, _mm512_cmpge_epu64_mask(A, p), A, p);
    B = _mm512_mask_sub_epi64(B, _mm512_cmpge_epu64_mask(B, p), B, p);
    _mm512_storeu_si512((void*)s, A);
    _mm512_mask_storeu_epi64((void*)(s + 8), 0x0f, B);
}

SIPP_AVX512 void permute_avx512(uint64_t s[12]) { permute_avx512_t<true>(s); }
SIPP_AVX512 void permute_avx512_mixed(uint64_t s[12]) { permute_avx512_t<false>(s); }

void permute_scalar_lookahead(uint64_t s[12]) {
    for (int r = 0; r < 4; r++) full_round_scalar(s, r);
    partial_rounds_lookahead(s);
    for (int r = 26; r < 30; r++) full_round_scalar(s, r);
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}

SIPP_AVX512 void piece(uint64_t s[12], int impl) {
    alignas(64) uint64_t buf[16];
    for (int i = 0; i < 12; i++) buf[i] = s[i];
    for (int i = 12; i < 16; i++) buf[i] = 0;
    if (impl == 10) {  // the eight full rounds, AVX-512
        __m512i A = _mm512_load_si512((const void*)buf), B = _mm512_load_si512((const void*)(buf + 8));
        for (int r = 0; r < 4; r++) full_round_avx512(A, B, r);
        for (int r = 26; r < 30; r++) full_round_avx512(A, B, r);
        _mm512_store_si512((void*)buf, A);
        _mm512_store_si512((void*)(buf + 8), B);
    } else if (impl == 11) {
        for (int r = 0; r < 4; r++) full_round_scalar(buf, r);
        for (int r = 26; r < 30; r++) full_round_scalar(buf, r);
    } else if (impl == 12) {
        partial_rounds_scalar(buf);
    } else if (impl == 13) {
        partial_rounds_lookahead(buf);
    } else {
        partial_rounds_avx512(buf);
    }
    for (int i = 0; i < 12; i++) s[i] = buf[i];
}

typedef void (*permute_fn)(uint64_t*);
permute_fn pick() {
    // scalar | scalar with look-ahead | AVX-512 (state update of the partial rounds on the vector pipes too) | mixed (AVX-512 full
    // rounds + scalar look-ahead partial rounds): one dependent permutation on an EPYC 9575F 1.17 / 1.08 / 0.77 / 0.76 us
    // (scripts/ubench/host_poseidon_bench.cpp).  Mixed where AVX-512 exists, else the look-ahead form; every form is reachable
    // through sipp_host_poseidon_permute(.., impl) and checked against the oracle there (tests/test_abi.py).
    __builtin_cpu_init();  // this runs from a static initialiser of the shared library
    const bool have512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
    return have512 ? permute_avx512_mixed : permute_scalar_lookahead;
}
const permute_fn CHOSEN = pick();

}  // namespace

void poseidon_permute(uint64_t s[12]) { CHOSEN(s); }

int poseidon_permute_impl(uint64_t s[12], int impl) {
    switch (impl) {
        case 0: permute_scalar(s); return 0;
        case 1: permute_scalar_lookahead(s); return 0;
        case 2:
        case 3:
            if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl"))) return -1;
            if (impl == 2) permute_avx512(s); else permute_avx512_mixed(s);
            return 0;
        default: break;
    }
    // pieces, for scripts/ubench/host_poseidon_bench.cpp only (not permutations)
    if (impl >= 10 && impl <= 14) {
        if (!(__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl"))) return -1;
        piece(s, impl);
        return 0;
    }
    return -2;
}

}  // namespace host
