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
    // one launch for all of them (sipp_k_gather_tasks): the task table goes up with one small copy (`tasks` stays alive until
    // the stream has been synchronised for the read-back below)
    std::vector<QueryGatherTask> tasks;
    {
        for (int o = 0; o < n_oracles; o++) {
            tasks.push_back(QueryGatherTask{ors[o].lde, d_st + off_rows[o], (uint64_t)ors[o].stride, 0, ors[o].ncols, 0, 0});
            if (nsib0) tasks.push_back(QueryGatherTask{ors[o].tree, d_st + off_sib[o], 0, 1, log_m, nsib0, 0});
        }
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t ab = p.arity_bits[r];
            tasks.push_back(QueryGatherTask{r_vals[r], d_st + off_leaf[r], (uint64_t)1 << r_loglen[r], 2, rshift[r], ab, 0});
            if (rsib[r]) tasks.push_back(QueryGatherTask{r_tree[r], d_st + off_rsib[r], 0, 1, r_loglen[r] - ab, rsib[r], rshift[r]});
        }
        QueryGatherTask* d_tasks = arena_alloc_t<QueryGatherTask>(ctx, tasks.size());
        if (!d_tasks) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_tasks, tasks.data(), tasks.size() * sizeof(QueryGatherTask), hipMemcpyHostToDevice, ctx->stream));
        SIPP_TRY(sipp_k_gather_tasks(ctx, d_tasks, (uint32_t)tasks.size(), d_idx, nq));
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


// =====================================================================================================================
// generic opening proofs: the C ABI of include/sipp_hip.h (PolynomialBatch::from_values / prove_openings for any FriParams)
// =====================================================================================================================
#define SIPP_FRI_MAGIC 0x5349505046524931ULL /* "SIPPFRI1" */

static size_t batch_len(const sipp_fri_batch& b) {
    size_t k = 0;
    for (uint32_t r = 0; r < b.n_ranges; r++) k += b.ranges[r].col_end - b.ranges[r].col_begin;
    return k;
}

static int check_params(sipp_ctx* ctx, const sipp_fri_params* p, uint32_t log_n) {
    if (!p || p->rate_bits < 1 || p->rate_bits > 3 || p->cap_height > 8 || p->pow_bits > 32 || p->pow_rule > SIPP_POW_HASH ||
        p->num_queries == 0 || p->num_queries > 1024 || p->n_rounds > SIPP_FRI_MAX_ROUNDS)
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri: parameters out of the supported range");
    uint32_t sum = 0;
    for (uint32_t r = 0; r < p->n_rounds; r++) {
        if (p->arity_bits[r] < 1 || p->arity_bits[r] > 4) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri: arity must be 2, 4, 8 or 16");
        sum += p->arity_bits[r];
        // every committed layer keeps at least 2^cap_height leaves (plonky2 asserts this) ...
        if (log_n + p->rate_bits < sum + p->cap_height)
            return sipp_fail(ctx, SIPP_E_BADARG, "fri: reduction arities leave a layer smaller than its cap");
        // ... and at least 16 values (a limit of this implementation's layer kernels, not of the protocol)
        if (log_n + p->rate_bits - sum + p->arity_bits[r] < 4)
            return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri: a committed layer of fewer than 16 values is not supported");
    }
    if (sum > log_n) return sipp_fail(ctx, SIPP_E_BADARG, "fri: the reduction arities exceed the degree bits");
    if (log_n < 10 || log_n > 24) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri: degree bits outside the supported range [10, 24]");
    return SIPP_OK;
}

extern "C" {

void sipp_fri_const_arity(sipp_fri_params* p, uint32_t arity_bits, uint32_t final_poly_bits, uint32_t degree_bits) {
    if (!p) return;
    p->n_rounds = 0;
    while (degree_bits > final_poly_bits && degree_bits + p->rate_bits - arity_bits >= p->cap_height && degree_bits >= arity_bits &&
           p->n_rounds < SIPP_FRI_MAX_ROUNDS) {
        p->arity_bits[p->n_rounds++] = arity_bits;
        degree_bits -= arity_bits;
    }
}

int sipp_commit_batch_ex(sipp_ctx* ctx, const uint64_t* d_in, int from_coeffs, uint64_t* d_coeffs, uint64_t* d_lde, uint64_t* d_tree,
                         size_t ncols, uint32_t log_n, uint32_t rate_bits, uint32_t cap_height, const uint64_t* d_salt,
                         uint32_t n_salt, uint64_t* cap_out) {
    if (!ctx || !d_in || !d_coeffs || !d_lde || !d_tree || !cap_out || ncols == 0) return SIPP_E_BADARG;
    if (rate_bits < 1 || rate_bits > 3 || cap_height > 8 || (n_salt != 0 && (n_salt != SIPP_SALT_SIZE || !d_salt)))
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "commit_batch_ex: blowup 2 / 4 / 8, cap height <= 8, salt 0 or 4 columns");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t log_m = log_n + rate_bits;
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m;
    ArenaScope scope(ctx);
    int rc;
    if (from_coeffs) {
        if (d_in != d_coeffs) SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_coeffs, d_in, ncols * n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        rc = sipp_lde_from_coeffs(ctx, d_coeffs, d_lde, ncols, log_n, rate_bits);
        if (rc == SIPP_E_UNSUPPORTED) rc = sipp_ntt_dif(ctx, d_coeffs, n, log_n, d_lde, m, log_m, ncols, false, NttDiag{gl::GEN, 0});
    } else {
        rc = d_in != d_coeffs ? sipp_lde_from_values(ctx, d_in, d_coeffs, d_lde, ncols, log_n, rate_bits) : SIPP_E_UNSUPPORTED;
        if (rc == SIPP_E_UNSUPPORTED) {
            const uint64_t* src = d_in;
            if (d_in == d_coeffs) {
                uint64_t* tmp = arena_alloc_t<uint64_t>(ctx, n * ncols);
                if (!tmp) return SIPP_E_NOMEM;
                SIPP_CHECK_HIP(ctx, hipMemcpyAsync(tmp, d_in, n * ncols * 8, hipMemcpyDeviceToDevice, ctx->stream));
                src = tmp;
            }
            rc = sipp_bitrev_cols(ctx, src, n, d_coeffs, n, log_n, ncols);
            if (rc == SIPP_OK) rc = sipp_ntt_dit(ctx, d_coeffs, n, log_n, ncols, true, NttDiag{});
            if (rc == SIPP_OK) rc = sipp_ntt_dif(ctx, d_coeffs, n, log_n, d_lde, m, log_m, ncols, false, NttDiag{gl::GEN, 0});
        }
    }
    // salt columns: natural LDE order in, leaf order (= bit-reversed rows) behind the polynomial columns
    if (rc == SIPP_OK && n_salt) rc = sipp_bitrev_cols(ctx, d_salt, m, d_lde + ncols * m, m, log_m, n_salt);
    if (rc == SIPP_OK) rc = sipp_k_poseidon_leaves(ctx, d_lde, m, ncols + n_salt, log_m, d_tree);
    if (rc == SIPP_OK) rc = sipp_k_merkle_levels(ctx, d_tree, log_m, cap_height);
    if (rc == SIPP_OK) rc = read_cap(ctx, d_tree, log_m, cap_height, cap_out);
    else (void)hipStreamSynchronize(ctx->stream);
    return rc;
}

size_t sipp_fri_proof_size(const sipp_oracle* oracles, size_t n_oracles, const sipp_fri_batch* batches, size_t n_batches,
                           uint32_t log_n, const sipp_fri_params* p) {
    if (!oracles || !batches || !p || n_oracles == 0 || n_oracles > 8 || n_batches == 0) return 0;
    FriParamsDev fp;
    fp.rate_bits = p->rate_bits; fp.cap_height = p->cap_height; fp.pow_bits = p->pow_bits; fp.num_queries = p->num_queries;
    fp.pow_rule = p->pow_rule;
    uint32_t sum = 0;
    for (uint32_t r = 0; r < p->n_rounds && r < SIPP_FRI_MAX_ROUNDS; r++) {
        fp.arity_bits.push_back(p->arity_bits[r]);
        sum += p->arity_bits[r];
    }
    if (sum > log_n) return 0;
    uint32_t lw[8];
    for (size_t o = 0; o < n_oracles; o++) lw[o] = oracles[o].n_polys + oracles[o].n_salt;
    size_t w = 8 + sipp_fri_core_words(fp, log_n, lw, (int)n_oracles);
    for (size_t b = 0; b < n_batches; b++) w += 2 * batch_len(batches[b]);
    return w;
}

int sipp_fri_prove_openings(sipp_ctx* ctx, const sipp_oracle* oracles, size_t n_oracles, const sipp_fri_batch* batches,
                            size_t n_batches, uint32_t log_n, const sipp_fri_params* p, sipp_challenger* chs, uint64_t* proof_out,
                            size_t proof_cap, size_t* proof_len) {
    if (!ctx || !oracles || !batches || !p || !chs || !proof_out || !proof_len || n_oracles == 0 || n_oracles > 8 || n_batches == 0)
        return SIPP_E_BADARG;
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    SIPP_TRY(check_params(ctx, p, log_n));
    if (chs->n_in > 7 || chs->n_out > 8) return sipp_fail(ctx, SIPP_E_BADARG, "fri: malformed challenger state");
    const size_t total = sipp_fri_proof_size(oracles, n_oracles, batches, n_batches, log_n, p);
    if (total == 0) return SIPP_E_BADARG;
    if (proof_cap < total) return sipp_fail(ctx, SIPP_E_BUFSZ, "fri: proof buffer too small (see sipp_fri_proof_size)");
    for (size_t b = 0; b < n_batches; b++)
        for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
            const sipp_poly_range& rg = batches[b].ranges[r];
            if (rg.oracle >= n_oracles || rg.col_begin > rg.col_end || rg.col_end > oracles[rg.oracle].n_polys)
                return sipp_fail(ctx, SIPP_E_BADARG, "fri: polynomial range outside its oracle");
        }
    const size_t n = (size_t)1 << log_n, m = n << p->rate_bits;
    FriParamsDev fp;
    fp.rate_bits = p->rate_bits; fp.cap_height = p->cap_height; fp.pow_bits = p->pow_bits; fp.num_queries = p->num_queries;
    fp.pow_rule = p->pow_rule;
    for (uint32_t r = 0; r < p->n_rounds; r++) fp.arity_bits.push_back(p->arity_bits[r]);
    host::Challenger ch;
    memcpy(ch.state, chs->state, sizeof ch.state);
    memcpy(ch.in_buf, chs->in_buf, sizeof ch.in_buf);
    memcpy(ch.out_buf, chs->out_buf, sizeof ch.out_buf);
    ch.n_in = (uint32_t)chs->n_in;
    ch.n_out = (uint32_t)chs->n_out;
    struct Release {
        sipp_ctx* c;
        ArenaMark mk;
        ~Release() {
            (void)hipStreamSynchronize(c->stream);
            arena_release(c, mk);
        }
    } release{ctx, arena_mark(ctx)};
    size_t pos = 0;
    uint64_t* pf = proof_out;
    {
        uint64_t hdr[8] = {SIPP_FRI_MAGIC, p->n_rounds, 0, p->num_queries, (uint64_t)n_oracles, (uint64_t)n_batches, total, log_n};
        memcpy(pf, hdr, sizeof hdr);
        pos = 8;
    }
    // ---- opened values: per batch a power table of its point, one dot product per polynomial ----
    std::vector<uint64_t*> d_zp(n_batches), d_zip(n_batches);
    std::vector<gl::E2> points(n_batches);
    for (size_t b = 0; b < n_batches; b++) {
        points[b] = gl::E2{batches[b].point[0], batches[b].point[1]};
        if (gl::eq(gl::pow(points[b], (uint64_t)n), gl::e2(1))) return sipp_fail(ctx, SIPP_E_SUBGROUP, "fri: opening point in the subgroup");
        d_zp[b] = arena_alloc_t<uint64_t>(ctx, 2 * n);
        d_zip[b] = arena_alloc_t<uint64_t>(ctx, 2 * n);
        if (!d_zp[b] || !d_zip[b]) return SIPP_E_NOMEM;
        SIPP_TRY(sipp_k_pow_table(ctx, points[b], n, d_zp[b]));
        SIPP_TRY(sipp_k_pow_table(ctx, gl::inv(points[b]), n, d_zip[b]));
    }
    for (size_t b = 0; b < n_batches; b++) {
        const size_t k = batch_len(batches[b]);
        uint64_t* d_open = arena_alloc_t<uint64_t>(ctx, 4 * k + 4);
        if (!d_open) return SIPP_E_NOMEM;
        size_t at = 0;
        for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
            const sipp_poly_range& rg = batches[b].ranges[r];
            const size_t cnt = rg.col_end - rg.col_begin;
            SIPP_TRY(sipp_k_openings(ctx, oracles[rg.oracle].d_coeffs + (size_t)rg.col_begin * n, cnt, n, d_zp[b], nullptr, d_open + 4 * at));
            at += cnt;
        }
        std::vector<uint64_t> hop(4 * k + 4);
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(hop.data(), d_open, (4 * k) * 8, hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t j = 0; j < k; j++) {
            memcpy(pf + pos, &hop[4 * j], 16);
            pos += 2;
            ch.observe_many(&hop[4 * j], 2);
        }
    }
    // ---- final polynomial: sum over batches of alpha-shifted quotients, times X ----
    const gl::E2 alpha = ch.get_ext();
    uint64_t* d_acc = arena_alloc_t<uint64_t>(ctx, 2 * n);
    uint64_t* d_final = arena_alloc_t<uint64_t>(ctx, 2 * n);
    if (!d_acc || !d_final) return SIPP_E_NOMEM;
    for (size_t b = 0; b < n_batches; b++) {
        const size_t k = batch_len(batches[b]);
        std::vector<const uint64_t*> cols;
        cols.reserve(k);
        for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
            const sipp_poly_range& rg = batches[b].ranges[r];
            for (uint32_t c = rg.col_begin; c < rg.col_end; c++) cols.push_back(oracles[rg.oracle].d_coeffs + (size_t)c * n);
        }
        std::vector<uint32_t> apow(6 * k + 6);
        gl::E2 ap = gl::e2(1);
        for (size_t c = 0; c < k; c++) {
            const uint64_t comp[2] = {ap.c0, ap.c1};
            for (int q = 0; q < 2; q++) {
                apow[6 * c + 3 * q] = (uint32_t)comp[q] & 0x3FFFFFu;
                apow[6 * c + 3 * q + 1] = (uint32_t)(comp[q] >> 22) & 0x3FFFFFu;
                apow[6 * c + 3 * q + 2] = (uint32_t)(comp[q] >> 44);
            }
            ap = gl::mul(ap, alpha);
        }
        const uint64_t** d_cols = reinterpret_cast<const uint64_t**>(arena_alloc(ctx, (k + 1) * sizeof(uint64_t*)));
        uint32_t* d_apow = arena_alloc_t<uint32_t>(ctx, apow.size());
        if (!d_cols || !d_apow) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_cols, cols.data(), k * sizeof(uint64_t*), hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_apow, apow.data(), apow.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host vectors go out of scope
        // alpha.shift_poly: final = final * alpha^(len of THIS batch) + quotient; `ap` is alpha^k here
        SIPP_TRY(sipp_k_fri_batch_quotient(ctx, d_cols, (int)k, n, d_apow, d_zp[b], d_zip[b], ap, b == 0, d_acc));
    }
    SIPP_TRY(sipp_k_fri_mulx(ctx, d_acc, n, d_final));
    // ---- commit phase, proof of work, queries ----
    FriOracleDev ors[8];
    for (size_t o = 0; o < n_oracles; o++)
        ors[o] = FriOracleDev{oracles[o].d_lde, m, oracles[o].n_polys + oracles[o].n_salt, oracles[o].d_tree};
    size_t flen = 0;
    SIPP_TRY(sipp_fri_prove_core(ctx, ors, (int)n_oracles, log_n, fp, d_final, ch, pf, pos, total, &flen, nullptr));
    pf[2] = flen;
    if (pos != total) return sipp_fail(ctx, SIPP_E_BUFSZ, "internal: opening proof length mismatch");
    *proof_len = pos;
    memcpy(chs->state, ch.state, sizeof ch.state);
    memcpy(chs->in_buf, ch.in_buf, sizeof ch.in_buf);
    memcpy(chs->out_buf, ch.out_buf, sizeof ch.out_buf);
    chs->n_in = ch.n_in;
    chs->n_out = ch.n_out;
    return SIPP_OK;
}

}  // extern "C"
