// sipp_amd/csrc/prover.hpp -- host entry points of the prover kernels (prover.hip, poseidon.hip) and the
// host-side Fiat-Shamir challenger.
#pragma once
#include <functional>

#include "ctx.hpp"
#include "host_challenger.hpp"
#include "host_poseidon.hpp"
#include "poseidon_constants.h"

// quotient_rest over a thin quotient domain (2N <= 2^15 points) walks its checked columns in up to sixteen ranges of at least 32 and
// needs chunks x 6 sums per point of scratch; ONE rule for the kernel launch (prover.hip) and the arena size (stark.hip)
inline uint32_t sipp_quotient_rest_chunks(uint32_t log_n, int n_checked) {
    if (log_n + 1 > 15) return 1;
    const int c = n_checked / 32 < 16 ? n_checked / 32 : 16;
    return c > 1 ? (uint32_t)c : 1;
}
int sipp_k_z_columns(sipp_ctx* ctx, const air_spec_t* a, const uint64_t* d_trace, uint32_t log_n, const uint64_t beta[2],
                     const uint64_t gamma[2], uint64_t* d_zv);
// quotient on the coset 7 <w_2N> (the first 2N leaves of the LDEs, whose columns are lde_stride apart); d_aux [n_aux][2N],
// d_out [2][2N] in leaf order
int sipp_k_quotient(sipp_ctx* ctx, const air_spec_t* a, uint32_t log_n, const uint64_t* d_lde, const uint64_t* d_zlde,
                    size_t lde_stride, const uint64_t* d_aux, const uint64_t alpha[2], const uint64_t beta[2],
                    const uint64_t gamma[2], uint64_t* d_out);
int sipp_k_pow_table(sipp_ctx* ctx, gl::E2 base, size_t n, uint64_t* d_tab);
int sipp_k_pow_table4(sipp_ctx* ctx, const gl::E2 base[4], size_t n, uint64_t* const d_tab[4]);
int sipp_k_openings(sipp_ctx* ctx, const uint64_t* d_coeffs, size_t ncols, size_t n, const uint64_t* d_t0,
                    const uint64_t* d_t1, uint64_t* d_out);
// (scratch of the grouped form: at most SIPP_OPENINGS_MAX_SEGS x 4 words per column -- sipp_workspace_bytes_cfg counts them)
constexpr size_t SIPP_OPENINGS_MAX_SEGS = 32;
// a gadget of more than 64 products is evaluated by up to this many lanes per quotient point (prover.hip: slices of its product list)
constexpr int SIPP_QUOTIENT_MAX_SLICES = 8;
int sipp_k_openings3(sipp_ctx* ctx, const uint64_t* const d_coeffs[3], const uint32_t ncols[3], size_t n, const uint64_t* d_t0,
                     const uint64_t* d_t1, uint64_t* d_out);
int sipp_k_fri_final(sipp_ctx* ctx, const uint64_t* const src[3], const int cnt[3], size_t n, const uint32_t* d_apow3,
                     int n1, gl::E2 shift1, const uint64_t* d_zp[2], const uint64_t* d_zip[2], uint64_t* d_final);
int sipp_k_fri_batch_quotient(sipp_ctx* ctx, const uint64_t* const* d_cols, int total, size_t n, const uint32_t* d_apow3,
                              const uint64_t* d_zp, const uint64_t* d_zip, gl::E2 shift, bool first, uint64_t* d_acc);
int sipp_k_fri_mulx(sipp_ctx* ctx, const uint64_t* d_acc, size_t n, uint64_t* d_final);
int sipp_k_fri_fold(sipp_ctx* ctx, const uint64_t* d_in, size_t len_in, uint32_t arity_bits, gl::E2 beta, uint64_t* d_out);
// one entry of the fused query-phase gather (sipp_k_gather_tasks): type 0 = oracle row (a = column stride, b = columns),
// 1 = Merkle siblings (b = log2 leaves, c = siblings, d = index shift), 2 = FRI leaf (a = values per component, b = shift, c = arity bits)
struct QueryGatherTask {
    const uint64_t* src;
    uint64_t* out;
    uint64_t a;
    uint32_t type, b, c, d;
};
int sipp_k_gather_tasks(sipp_ctx* ctx, const QueryGatherTask* d_tasks, uint32_t n_tasks, const uint32_t* d_idx, uint32_t nq);
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

// ---- host Poseidon + duplex challenger: host_challenger.hpp (shared with the host-only verifier, verify.cpp) ----

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
