// sipp_amd/csrc/prover.hpp -- host entry points of the prover kernels (prover.hip, poseidon.hip) and the
// host-side Fiat-Shamir challenger.
#pragma once
#include "ctx.hpp"
#include "poseidon_constants.h"

int sipp_k_z_columns(sipp_ctx* ctx, const sipp_air_t* a, const uint64_t* d_trace, uint32_t log_n, const uint64_t gamma[2],
                     uint64_t* d_zv);
int sipp_k_quotient(sipp_ctx* ctx, const sipp_air_t* a, uint32_t log_n, const uint64_t* d_lde, const uint64_t* d_zlde,
                    const uint64_t* d_aux, const uint64_t alpha[2], const uint64_t gamma[2], uint64_t* d_out);
int sipp_k_pow_table(sipp_ctx* ctx, gl::E2 base, size_t n, uint64_t* d_tab);
int sipp_k_openings(sipp_ctx* ctx, const uint64_t* d_coeffs, size_t ncols, size_t n, const uint64_t* d_t0,
                    const uint64_t* d_t1, uint64_t* d_out);
int sipp_k_fri_final(sipp_ctx* ctx, const uint64_t* const src[3], const int cnt[3], size_t n, const uint64_t* d_apow,
                     int n1, gl::E2 shift1, const uint64_t* d_zp[2], const uint64_t* d_zip[2], uint64_t* d_final);
int sipp_k_fri_fold(sipp_ctx* ctx, const uint64_t* d_in, size_t len_in, gl::E2 beta, uint64_t* d_out);
int sipp_k_gather_rows(sipp_ctx* ctx, const uint64_t* d_lde, size_t m, uint32_t ncols, const uint32_t* d_idx, uint32_t nq,
                       uint64_t* d_out);
int sipp_k_gather_siblings(sipp_ctx* ctx, const uint64_t* d_tree, uint32_t log_leaves, uint32_t nsib, uint32_t shift,
                           const uint32_t* d_idx, uint32_t nq, uint64_t* d_out);
int sipp_k_gather_fri_leaf(sipp_ctx* ctx, const uint64_t* d_vals, size_t len, uint32_t shift, const uint32_t* d_idx,
                           uint32_t nq, uint64_t* d_out);
// poseidon.hip
int sipp_k_fri_leaves(sipp_ctx* ctx, const uint64_t* d_vals, size_t len, uint64_t* d_digests);
int sipp_k_pow_search(sipp_ctx* ctx, const uint64_t state[12], const uint64_t* in_buf, uint32_t n_in, uint32_t pow_bits,
                      uint64_t* witness);

// ---- host Poseidon + duplex challenger (plonky2 iop/challenger.rs) ----
namespace host {

inline void poseidon_permute(uint64_t s[12]) {
    static const uint64_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    auto sbox = [](uint64_t x) {
        uint64_t x2 = gl::sqr(x), x3 = gl::mul(x2, x), x4 = gl::sqr(x2);
        return gl::mul(x3, x4);
    };
    auto mds = [&](uint64_t* st) {
        uint64_t out[12];
        for (int r = 0; r < 12; r++) {
            unsigned __int128 acc = 0;
            for (int i = 0; i < 12; i++) acc += (unsigned __int128)st[(i + r) % 12] * CIRC[i];
            if (r == 0) acc += (unsigned __int128)st[0] * 8;
            out[r] = gl::reduce128((uint64_t)(acc >> 64), (uint64_t)acc);
        }
        for (int r = 0; r < 12; r++) st[r] = out[r];
    };
    for (int rnd = 0; rnd < 30; rnd++) {
        const bool full = rnd < 4 || rnd >= 26;
        for (int i = 0; i < 12; i++) s[i] = gl::add(s[i], SIPP_POSEIDON_RC[12 * rnd + i]);
        if (full)
            for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        else
            s[0] = sbox(s[0]);
        mds(s);
    }
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
    gl::E2 get_ext() {
        gl::E2 r;
        r.c0 = get();
        r.c1 = get();
        return r;
    }
};

}  // namespace host
