// sipp_amd/csrc/fri.hip -- the FRI core shared by the three STARK provers and the generic opening proofs:
// commit phase (coset NTT of the folded polynomial, Merkle commit, fold by any arity 2 .. 16), final polynomial,
// proof of work (both rules), query rounds over any number of initial oracles.
//
// Replaces plonky2's fri_proof (fri/prover.rs: fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds
// @ InternetMaximalism/plonky2 541e127, not vendored); the reference reaches it through the STARK sub-proofs behind
// src/verifier_circuit.rs:133-135 and through the outer proof of src/verifier_circuit.rs:253.  Same section layout as
// oracle/fri.c::orc_fri_prove_core.
#include <algorithm>

#include "prover.hpp"

static size_t tree_words(uint32_t log_leaves) { return ((size_t)8 << log_leaves); }  // 2 * leaves * 4

static int read_cap(sipp_ctx* ctx, const uint64_t* d_tree, uint32_t log_leaves, uint32_t cap_height, uint64_t* cap_host) {
    const uint32_t ch = std::min(cap_height, log_leaves);
    uint64_t off = 0;
    for (uint32_t l = 0; l < log_leaves - ch; l++) off += (uint64_t)1 << (log_leaves - l);
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(cap_host, d_tree + 4 * off, ((size_t)4 << ch) * 8, hipMemcpyDeviceToHost, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SIPP_OK;
}

size_t sipp_fri_core_words(const FriParamsDev& p, uint32_t log_n, const uint32_t* leaf_words, int n_oracles) {
    const uint32_t log_m = log_n + p.rate_bits;
    const size_t cap = (size_t)4 << p.cap_height;
    uint32_t sum = 0;
    for (uint32_t ab : p.arity_bits) sum += ab;
    size_t w = p.arity_bits.size() * cap + 2 * ((size_t)1 << (log_n - sum)) + 1;
    size_t per_q = 0;
    for (int o = 0; o < n_oracles; o++) per_q += leaf_words[o] + (size_t)(log_m - p.cap_height) * 4;
    uint32_t lt = log_m;
    for (uint32_t ab : p.arity_bits) {
        lt -= ab;
        per_q += ((size_t)2 << ab) + (size_t)(lt > p.cap_height ? lt - p.cap_height : 0) * 4;
    }
    return w + p.num_queries * per_q;
}

int sipp_fri_prove_core(sipp_ctx* ctx, const FriOracleDev* ors, int n_oracles, uint32_t log_n, const FriParamsDev& p,
                        uint64_t* d_final, host::Challenger& ch, uint64_t* pf, size_t& pos, size_t cap_total, size_t* final_len,
                        const std::function<void(const char*)>& tick) {
    const uint32_t log_m = log_n + p.rate_bits, R = (uint32_t)p.arity_bits.size(), nq = p.num_queries;
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m;
    const size_t cap_words = (size_t)4 << p.cap_height;
    if (p.cap_height > 8 || n_oracles < 1 || n_oracles > 8) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri: unsupported shape");
    auto push = [&](const uint64_t* v, size_t cnt) -> int {
        if (pos + cnt > cap_total) return SIPP_E_BUFSZ;
        memcpy(pf + pos, v, cnt * 8);
        pos += cnt;
        return SIPP_OK;
    };
    uint64_t cap_host[4 << 8];
    // commit phase
    std::vector<uint64_t*> r_vals(R), r_tree(R);
    std::vector<uint32_t> r_loglen(R);
    uint64_t* cur = d_final;   // [2][len_c] coefficients with len_c non-zero entries
    size_t len_c = n;
    uint32_t log_len = log_m;
    uint64_t shift = gl::GEN;
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t ab = p.arity_bits[r];
        const size_t len = (size_t)1 << log_len;      // values this round
        if (log_len < ab + 0u || (len_c >> ab) == 0) return sipp_fail(ctx, SIPP_E_BADARG, "fri: reduction arities exceed the degree");
        r_loglen[r] = log_len;
        r_vals[r] = arena_alloc_t<uint64_t>(ctx, 2 * len);
        r_tree[r] = arena_alloc_t<uint64_t>(ctx, tree_words(log_len - ab));
        uint64_t* nxt = arena_alloc_t<uint64_t>(ctx, 2 * (len_c >> ab) + 2);
        if (!r_vals[r] || !r_tree[r] || !nxt) return SIPP_E_NOMEM;
        // coset NTT of the current polynomial (len_c = len >> rate_bits coefficients, zero padded to len)
        SIPP_TRY(sipp_ntt_dif(ctx, cur, len_c, log_len - p.rate_bits, r_vals[r], len, log_len, 2, false, NttDiag{shift, 0}));
        SIPP_TRY(sipp_k_fri_leaves(ctx, r_vals[r], len, ab, r_tree[r]));
        SIPP_TRY(sipp_k_merkle_levels(ctx, r_tree[r], log_len - ab, p.cap_height));
        SIPP_TRY(read_cap(ctx, r_tree[r], log_len - ab, p.cap_height, cap_host));
        const size_t cw = (size_t)4 << std::min(p.cap_height, log_len - ab);
        ch.observe_many(cap_host, cw);
        SIPP_TRY(push(cap_host, cw));
        const gl::E2 fb = ch.get_ext();
        SIPP_TRY(sipp_k_fri_fold(ctx, cur, len_c, ab, fb, nxt));
        cur = nxt;
        len_c >>= ab;
        log_len -= ab;
        shift = gl::pow(shift, (uint64_t)1 << ab);
    }
    (void)cap_words;
    if (final_len) *final_len = len_c;
    {
        std::vector<uint64_t> fpv(2 * len_c);
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(fpv.data(), cur, fpv.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < len_c; i++) {
            uint64_t e[2] = {fpv[i], fpv[len_c + i]};
            ch.observe_many(e, 2);
            SIPP_TRY(push(e, 2));
        }
    }
    if (tick) tick("fri commit phase");
    // proof of work: smallest valid nonce (deterministic; upstream's rayon find_any may return another one)
    uint64_t pow_witness = 0;
    if (p.pow_rule == SIPP_POW_HASH) {
        // response = hash_no_pad(challenger.get_hash() || w)[0]: one permutation of (h0..h3, w, 0, ...)
        uint64_t zero[12] = {0}, cur_h[4];
        for (int i = 0; i < 4; i++) cur_h[i] = ch.get();
        SIPP_TRY(sipp_k_pow_search(ctx, zero, cur_h, 4, 0, p.pow_bits, &pow_witness));
    } else {
        // observe w, response = next challenge = word 7 of the duplexed state
        SIPP_TRY(sipp_k_pow_search(ctx, ch.state, ch.in_buf, ch.n_in, 7, p.pow_bits, &pow_witness));
        ch.observe(pow_witness);
        (void)ch.get();
    }
    SIPP_TRY(push(&pow_witness, 1));
    if (tick) tick("pow");
    // ---- queries ----
    std::vector<uint32_t> qidx(nq);
    for (uint32_t i = 0; i < nq; i++) qidx[i] = (uint32_t)(ch.get() % m);
    uint32_t* d_idx = arena_alloc_t<uint32_t>(ctx, nq);
    if (!d_idx) return SIPP_E_NOMEM;
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_idx, qidx.data(), nq * 4, hipMemcpyHostToDevice, ctx->stream));
    const uint32_t nsib0 = log_m - p.cap_height;
    // staging layout on the device, then one D2H
    size_t st_words = 0;
    std::vector<size_t> off_rows(n_oracles), off_sib(n_oracles), off_leaf(R), off_rsib(R);
    std::vector<uint32_t> rsib(R), rshift(R);
    for (int o = 0; o < n_oracles; o++) {
        off_rows[o] = st_words;
        st_words += (size_t)nq * ors[o].ncols;
        off_sib[o] = st_words;
        st_words += (size_t)nq * nsib0 * 4;
    }
    {
        uint32_t sh = 0;
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t ab = p.arity_bits[r];
            sh += ab;
            rshift[r] = sh;
            const uint32_t lt = log_m - sh;
            rsib[r] = lt > p.cap_height ? lt - p.cap_height : 0;
            off_leaf[r] = st_words;
            st_words += (size_t)nq * ((size_t)2 << ab);
            off_rsib[r] = st_words;
            st_words += (size_t)nq * rsib[r] * 4;
        }
    }
    if (st_words > ctx->h_pinned_words) return sipp_fail(ctx, SIPP_E_NOMEM, "query staging exceeds the pinned buffer");
    uint64_t* d_st = arena_alloc_t<uint64_t>(ctx, st_words);
    if (!d_st) return SIPP_E_NOMEM;
    for (int o = 0; o < n_oracles; o++) {
        SIPP_TRY(sipp_k_gather_rows(ctx, ors[o].lde, ors[o].stride, ors[o].ncols, d_idx, nq, d_st + off_rows[o]));
        SIPP_TRY(sipp_k_gather_siblings(ctx, ors[o].tree, log_m, nsib0, 0, d_idx, nq, d_st + off_sib[o]));
    }
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t ab = p.arity_bits[r];
        SIPP_TRY(sipp_k_gather_fri_leaf(ctx, r_vals[r], (size_t)1 << r_loglen[r], rshift[r], ab, d_idx, nq, d_st + off_leaf[r]));
        SIPP_TRY(sipp_k_gather_siblings(ctx, r_tree[r], r_loglen[r] - ab, rsib[r], rshift[r], d_idx, nq, d_st + off_rsib[r]));
    }
    uint64_t* hst = ctx->h_pinned;
    SIPP_CHECK_HIP(ctx, hipMemcpyAsync(hst, d_st, st_words * 8, hipMemcpyDeviceToHost, ctx->stream));
    SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t qi = 0; qi < nq; qi++) {
        for (int o = 0; o < n_oracles; o++) {
            SIPP_TRY(push(hst + off_rows[o] + (size_t)qi * ors[o].ncols, (size_t)ors[o].ncols));
            SIPP_TRY(push(hst + off_sib[o] + (size_t)qi * nsib0 * 4, (size_t)nsib0 * 4));
        }
        for (uint32_t r = 0; r < R; r++) {
            const size_t lw = (size_t)2 << p.arity_bits[r];
            SIPP_TRY(push(hst + off_leaf[r] + (size_t)qi * lw, lw));
            SIPP_TRY(push(hst + off_rsib[r] + (size_t)qi * rsib[r] * 4, (size_t)rsib[r] * 4));
        }
    }
    return SIPP_OK;
}
