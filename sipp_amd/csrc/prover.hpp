// sipp_amd/csrc/prover.hpp -- host entry points of the prover kernels (prover.hip, poseidon.hip) and the
// host-side Fiat-Shamir challenger.
#pragma once
#include <functional>

#include "ctx.hpp"
#include "poseidon_constants.h"

int sipp_k_z_columns(sipp_ctx* ctx, const sipp_air_t* a, const uint64_t* d_trace, uint32_t log_n, const uint64_t beta[2],
                     const uint64_t gamma[2], uint64_t* d_zv);
// quotient on the coset 7 <w_2N> (the first 2N leaves of the LDEs, whose columns are lde_stride apart); d_aux [n_aux][2N],
// d_out [2][2N] in leaf order
int sipp_k_quotient(sipp_ctx* ctx, const sipp_air_t* a, uint32_t log_n, const uint64_t* d_lde, const uint64_t* d_zlde,
                    size_t lde_stride, const uint64_t* d_aux, const uint64_t alpha[2], const uint64_t beta[2],
                    const uint64_t gamma[2], uint64_t* d_out);
int sipp_k_pow_table(sipp_ctx* ctx, gl::E2 base, size_t n, uint64_t* d_tab);
int sipp_k_openings(sipp_ctx* ctx, const uint64_t* d_coeffs, size_t ncols, size_t n, const uint64_t* d_t0,
                    const uint64_t* d_t1, uint64_t* d_out);
int sipp_k_fri_final(sipp_ctx* ctx, const uint64_t* const src[3], const int cnt[3], size_t n, const uint32_t* d_apow3,
                     int n1, gl::E2 shift1, const uint64_t* d_zp[2], const uint64_t* d_zip[2], uint64_t* d_final);
int sipp_k_fri_batch_quotient(sipp_ctx* ctx, const uint64_t* const* d_cols, int total, size_t n, const uint32_t* d_apow3,
                              const uint64_t* d_zp, const uint64_t* d_zip, gl::E2 shift, bool first, uint64_t* d_acc);
int sipp_k_fri_mulx(sipp_ctx* ctx, const uint64_t* d_acc, size_t n, uint64_t* d_final);
int sipp_k_fri_fold(sipp_ctx* ctx, const uint64_t* d_in, size_t len_in, uint32_t arity_bits, gl::E2 beta, uint64_t* d_out);
int sipp_k_gather_rows(sipp_ctx* ctx, const uint64_t* d_lde, size_t m, uint32_t ncols, const uint32_t* d_idx, uint32_t nq,
                       uint64_t* d_out);
int sipp_k_gather_siblings(sipp_ctx* ctx, const uint64_t* d_tree, uint32_t log_leaves, uint32_t nsib, uint32_t shift,
                           const uint32_t* d_idx, uint32_t nq, uint64_t* d_out);
int sipp_k_gather_fri_leaf(sipp_ctx* ctx, const uint64_t* d_vals, size_t len, uint32_t shift, uint32_t arity_bits,
                           const uint32_t* d_idx, uint32_t nq, uint64_t* d_out);
// poseidon.hip
// leaf k = the 2^arity_bits consecutive (leaf-order) extension values [k 2^ab, (k + 1) 2^ab), flattened (c0, c1); hash_or_noop
int sipp_k_fri_leaves(sipp_ctx* ctx, const uint64_t* d_vals, size_t len, uint32_t arity_bits, uint64_t* d_digests);
// smallest w whose response has pow_bits leading zeros; response = word `resp_word` of permute(state with in_buf[0..n_in)
// and w at position n_in overwritten)
int sipp_k_pow_search(sipp_ctx* ctx, const uint64_t state[12], const uint64_t* in_buf, uint32_t n_in, uint32_t resp_word,
                      uint32_t pow_bits, uint64_t* witness);

// ---- host Poseidon + duplex challenger (plonky2 iop/challenger.rs) ----
namespace host {

// Same algorithm as the device permutation (poseidon.cuh): dense circulant MDS in the 8 full rounds, the sparse
// "fast" form in the 22 partial rounds, written for a 64-bit host core: u128 products, a three-instruction-deep
// reduction, values kept as any u64 congruent to the state word until the end, MDS on the 32-bit halves so its sums
// stay in u64.  Observing the ~27 k opening words of the widest STARK is ~3.4 k sequential permutations on the
// proof's critical path.  Checked against the device kernel by every proof parity test.
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
inline void full_round(uint64_t s[12], int rnd) {
    static const uint64_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
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
inline void poseidon_permute(uint64_t s[12]) {
    for (int r = 0; r < 4; r++) full_round(s, r);
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
    for (int r = 26; r < 30; r++) full_round(s, r);
    for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}

struct Challenger {
    uint64_t state[12] = {0};
    uint64_t in_buf[8];
    uint32_t n_in = 0;
    uint64_t out_buf[8];
    uint32_t n_out = 0;
    void duplex() {
        for (uint32_t i = 0; i < n_in; i++) state[i] = in_buf[i];
        n_in = 0;
        poseidon_permute(state);
        for (int i = 0; i < 8; i++) out_buf[i] = state[i];
        n_out = 8;
    }
    void observe(uint64_t e) {
        n_out = 0;
        in_buf[n_in++] = e;
        if (n_in == 8) duplex();
    }
    void observe_many(const uint64_t* e, size_t n) {
        for (size_t i = 0; i < n; i++) observe(e[i]);
    }
    uint64_t get() {
        if (n_in != 0 || n_out == 0) duplex();
        return out_buf[--n_out];
    }
    // hash_n_to_hash_no_pad (overwrite-mode sponge, rate 8) and two_to_one, for the statement binding of stark.hip
    static void hash_no_pad(const uint64_t* in, size_t n, uint64_t out[4]) {
        uint64_t s[12] = {0};
        for (size_t i = 0; i < n; i += 8) {
            for (size_t k = 0; k < 8 && i + k < n; k++) s[k] = in[i + k];
            poseidon_permute(s);
        }
        for (int k = 0; k < 4; k++) out[k] = s[k];
    }
    static void two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
        uint64_t s[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
        poseidon_permute(s);
        for (int k = 0; k < 4; k++) out[k] = s[k];
    }
    gl::E2 get_ext() {
        gl::E2 r;
        r.c0 = get();
        r.c1 = get();
        return r;
    }
};

}  // namespace host

// ---- the FRI core (fri.hip) ------------------------------------------------------------------------------------
struct FriOracleDev {
    const uint64_t* lde;    // [ncols][stride] leaf order (salt columns, if any, are the last ones)
    size_t stride;          // n << rate_bits
    uint32_t ncols;         // words per leaf
    const uint64_t* tree;   // levels back to back
};
struct FriParamsDev {
    uint32_t rate_bits = 1, cap_height = 4, pow_bits = 16, num_queries = 84, pow_rule = 0;
    std::vector<uint32_t> arity_bits;   // FriParams::reduction_arity_bits
};
// u64 words of the section sipp_fri_prove_core appends (caps, final polynomial, witness, query rounds)
size_t sipp_fri_core_words(const FriParamsDev& p, uint32_t log_n, const uint32_t* leaf_words, int n_oracles);
// d_final: [2][n] extension coefficients (SoA) of the final polynomial, already multiplied by X
int sipp_fri_prove_core(sipp_ctx* ctx, const FriOracleDev* ors, int n_oracles, uint32_t log_n, const FriParamsDev& p,
                        uint64_t* d_final, host::Challenger& ch, uint64_t* pf, size_t& pos, size_t cap_total, size_t* final_len,
                        const std::function<void(const char*)>& tick);
