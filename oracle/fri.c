/*
 * oracle/fri.c -- see fri.h.  Follows plonky2 fri/oracle.rs (prove_openings), fri/prover.rs (fri_committed_trees,
 * fri_proof_of_work, fri_prover_query_rounds), fri/verifier.rs (fri_combine_initial, compute_evaluation,
 * fri_verifier_query_round) and fri/reduction_strategies.rs as recorded in SURVEY.md App. A.8.  TEST INFRASTRUCTURE ONLY;
 * PARITY UNPINNED.  Deterministic: the proof-of-work witness is the SMALLEST valid nonce (upstream: rayon find_any).
 */
#include "fri.h"
#include <stdlib.h>
#include <string.h>

#define FRI_MAGIC 0x5349505046524931ULL /* "SIPPFRI1" */

void orc_wb_push(orc_wbuf *b, const uint64_t *v, size_t n) {
    if (b->len + n > b->cap) {
        while (b->len + n > b->cap) b->cap = b->cap ? b->cap * 2 : 4096;
        b->w = (uint64_t *)realloc(b->w, b->cap * sizeof(uint64_t));
    }
    memcpy(b->w + b->len, v, n * sizeof(uint64_t));
    b->len += n;
}
static void wb_push1(orc_wbuf *b, uint64_t v) { orc_wb_push(b, &v, 1); }
static void wb_push_ext(orc_wbuf *b, gl2 v) { wb_push1(b, v.c0); wb_push1(b, v.c1); }

void orc_fri_const_arity(orc_fri_params *p, unsigned arity_bits, unsigned final_poly_bits, unsigned degree_bits) {
    p->n_rounds = 0;
    while (degree_bits > final_poly_bits && degree_bits + p->rate_bits - arity_bits >= p->cap_height &&
           degree_bits >= arity_bits && p->n_rounds < ORC_FRI_MAX_ROUNDS) {
        p->arity_bits[p->n_rounds++] = arity_bits;
        degree_bits -= arity_bits;
    }
}

orc_batch *orc_batch_salted(const uint64_t *data, int from_values, size_t ncols, unsigned log_n, unsigned rate_bits,
                            unsigned cap_height, const uint64_t *salt, size_t n_salt) {
    orc_batch *b = from_values ? orc_batch_from_values(data, ncols, log_n, rate_bits, cap_height)
                               : orc_batch_from_coeffs(data, ncols, log_n, rate_bits, cap_height);
    if (!n_salt) return b;
    /* rebuild the tree over leaves with the salt words appended (leaf j = natural LDE row bitrev(j)) */
    const unsigned log_m = log_n + rate_bits;
    const size_t m = (size_t)1 << log_m, ll = ncols + n_salt;
    uint64_t *leaves = (uint64_t *)malloc(m * ll * sizeof(uint64_t));
    for (size_t j = 0; j < m; j++) {
        memcpy(leaves + j * ll, b->tree->leaves + j * ncols, ncols * sizeof(uint64_t));
        for (size_t s = 0; s < n_salt; s++) leaves[j * ll + ncols + s] = salt[s * m + bitrev32((uint32_t)j, log_m)];
    }
    orc_merkle_free(b->tree);
    b->tree = orc_merkle_new(leaves, log_m, ll, cap_height);
    free(leaves);
    return b;
}

/* ---------------- composition polynomial ---------------- */
static size_t batch_len(const orc_fri_batch *bt) {
    size_t k = 0;
    for (uint32_t r = 0; r < bt->n_ranges; r++) k += bt->ranges[r].col_end - bt->ranges[r].col_begin;
    return k;
}

gl2 *orc_fri_final_poly(const orc_batch *const *oracles, const orc_fri_batch *batches, size_t n_batches, unsigned log_n, gl2 alpha) {
    const size_t n = (size_t)1 << log_n;
    gl2 *fin = (gl2 *)calloc(n, sizeof(gl2)), *comp = (gl2 *)malloc(n * sizeof(gl2)), *quo = (gl2 *)malloc(n * sizeof(gl2));
    for (size_t b = 0; b < n_batches; b++) {
        const orc_fri_batch *bt = &batches[b];
        /* F = sum_j alpha^j f_j over the batch's polynomials in order (ReducingFactor::reduce_polys_base) */
        for (size_t k = 0; k < n; k++) comp[k] = gl2_from(0);
        gl2 ap = gl2_from(1);
        for (uint32_t r = 0; r < bt->n_ranges; r++) {
            const orc_batch *o = oracles[bt->ranges[r].oracle];
            for (uint32_t c = bt->ranges[r].col_begin; c < bt->ranges[r].col_end; c++) {
                const uint64_t *co = o->coeffs + (size_t)c * n;
                for (size_t k = 0; k < n; k++) comp[k] = gl2_add(comp[k], gl2_scale(ap, co[k]));
                ap = gl2_mul(ap, alpha);
            }
        }
        /* (F(X) - F(z)) / (X - z): q_{k-1} = F_k + z q_k */
        gl2 acc = gl2_from(0);
        quo[n - 1] = gl2_from(0);
        for (size_t k = n; k-- > 1;) { acc = gl2_add(comp[k], gl2_mul(acc, bt->point)); quo[k - 1] = acc; }
        /* alpha.shift_poly(final): final = final * alpha^(len of THIS batch) + quotient */
        gl2 sh = gl2_pow(alpha, (uint64_t)batch_len(bt));
        for (size_t k = 0; k < n; k++) fin[k] = gl2_add(gl2_mul(fin[k], sh), quo[k]);
    }
    /* times X (plonky2 PR 436): the top coefficient is zero by construction */
    for (size_t k = n - 1; k > 0; k--) fin[k] = fin[k - 1];
    fin[0] = gl2_from(0);
    free(comp); free(quo);
    return fin;
}

/* ---------------- proof of work ---------------- */
static uint64_t pow_response(const orc_challenger *ch, const uint64_t cur_hash[4], unsigned rule, uint64_t w) {
    if (rule == ORC_POW_HASH) {
        uint64_t in[5] = {cur_hash[0], cur_hash[1], cur_hash[2], cur_hash[3], w}, out[4];
        orc_hash_no_pad(in, 5, out);
        return out[0];
    }
    orc_challenger c2 = *ch;
    orc_chal_observe(&c2, w);
    return orc_chal_get(&c2);
}

/* ---------------- prover core ---------------- */
int orc_fri_prove_core(const orc_batch *const *oracles, size_t n_oracles, unsigned log_n, const orc_fri_params *p,
                       const gl2 *final_coeffs, orc_challenger *ch, orc_wbuf *out, size_t *final_len) {
    const unsigned log_m = log_n + p->rate_bits;
    const size_t m = (size_t)1 << log_m, n = (size_t)1 << log_n;
    orc_merkle **trees = (orc_merkle **)calloc(p->n_rounds ? p->n_rounds : 1, sizeof(orc_merkle *));
    gl2 *coeffs = (gl2 *)calloc(m, sizeof(gl2)), *values = (gl2 *)malloc(m * sizeof(gl2));
    memcpy(coeffs, final_coeffs, n * sizeof(gl2));
    size_t len = m;
    unsigned log_len = log_m;
    uint64_t shift = GL_GEN;
    for (unsigned r = 0; r <= p->n_rounds; r++) {
        uint64_t s = 1;
        for (size_t i = 0; i < len; i++) { values[i] = gl2_scale(coeffs[i], s); s = gl_mul(s, shift); }
        orc_fft_ext(values, log_len);
        if (r == p->n_rounds) break;
        const unsigned ab = p->arity_bits[r];
        const size_t arity = (size_t)1 << ab, n_leaves = len >> ab;
        uint64_t *leaves = (uint64_t *)malloc(len * 2 * sizeof(uint64_t));
        for (size_t j = 0; j < len; j++) {
            gl2 v = values[bitrev32((uint32_t)j, log_len)];
            leaves[2 * j] = v.c0; leaves[2 * j + 1] = v.c1;
        }
        trees[r] = orc_merkle_new(leaves, log_len - ab, 2 * arity, p->cap_height);
        free(leaves);
        size_t cap_n = (size_t)1 << trees[r]->cap_height;
        orc_chal_observe_cap(ch, trees[r]->cap, cap_n);
        orc_wb_push(out, trees[r]->cap, cap_n * 4);
        gl2 beta = orc_chal_get_ext(ch);
        for (size_t k = 0; k < n_leaves; k++) {
            gl2 acc = gl2_from(0);
            for (size_t i = arity; i-- > 0;) acc = gl2_add(gl2_mul(acc, beta), coeffs[arity * k + i]);
            coeffs[k] = acc;
        }
        len = n_leaves; log_len -= ab;
        shift = gl_pow(shift, arity);
    }
    free(values);
    const size_t flen = len >> p->rate_bits;
    if (final_len) *final_len = flen;
    for (size_t i = 0; i < flen; i++) { orc_chal_observe_ext(ch, coeffs[i]); wb_push_ext(out, coeffs[i]); }
    free(coeffs);
    /* proof of work: smallest witness */
    uint64_t w = 0, cur_hash[4] = {0, 0, 0, 0};
    if (p->pow_rule == ORC_POW_HASH) for (int i = 0; i < 4; i++) cur_hash[i] = orc_chal_get(ch);
    for (;; w++) {
        uint64_t resp = pow_response(ch, cur_hash, p->pow_rule, w);
        if (p->pow_bits == 0 || (resp >> (64 - p->pow_bits)) == 0) break;
    }
    if (p->pow_rule != ORC_POW_HASH) {
        orc_chal_observe(ch, w);
        (void)orc_chal_get(ch);
    }
    wb_push1(out, w);
    /* queries */
    int err = 0;
    uint64_t *sib = (uint64_t *)malloc((size_t)(log_m + 1) * 4 * sizeof(uint64_t));
    for (unsigned qi = 0; qi < p->num_queries; qi++) {
        size_t x = (size_t)(orc_chal_get(ch) % m);
        for (size_t o = 0; o < n_oracles; o++) {
            const orc_merkle *t = oracles[o]->tree;
            orc_wb_push(out, t->leaves + x * t->leaf_len, t->leaf_len);
            size_t ns = orc_merkle_prove(t, x, sib);
            if (ns != log_m - t->cap_height) err = -21;
            orc_wb_push(out, sib, ns * 4);
        }
        size_t xi = x;
        for (unsigned r = 0; r < p->n_rounds; r++) {
            xi >>= p->arity_bits[r];
            orc_wb_push(out, trees[r]->leaves + xi * trees[r]->leaf_len, trees[r]->leaf_len);
            size_t ns = orc_merkle_prove(trees[r], xi, sib);
            orc_wb_push(out, sib, ns * 4);
        }
    }
    free(sib);
    for (unsigned r = 0; r < p->n_rounds; r++) orc_merkle_free(trees[r]);
    free(trees);
    return err;
}

/* ---------------- verifier core ---------------- */
static const uint64_t *take(const uint64_t *proof, size_t len, size_t *pos, size_t n, int *bad) {
    if (*pos + n > len) { *bad = 1; return proof; }
    const uint64_t *p = proof + *pos;
    *pos += n;
    return p;
}

static gl2 eval_poly_ext(const gl2 *c, size_t n, gl2 x) {
    gl2 acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, x), c[i]);
    return acc;
}

int orc_fri_verify_core(const uint64_t *proof, size_t len, size_t *pos, const uint64_t *const *caps, const int *ncols,
                        const int *n_salt, size_t n_oracles, const orc_fri_batch *batches, const gl2 *const *opened,
                        size_t n_batches, unsigned log_n, const orc_fri_params *p, gl2 alpha, orc_challenger *ch) {
    const unsigned log_m = log_n + p->rate_bits;
    const size_t m = (size_t)1 << log_m, n = (size_t)1 << log_n;
    const size_t cap_n = (size_t)1 << p->cap_height;
    int bad = 0, rc = 0;
    const uint64_t *rcaps[ORC_FRI_MAX_ROUNDS];
    gl2 betas[ORC_FRI_MAX_ROUNDS];
    unsigned sum_ab = 0;
    for (unsigned r = 0; r < p->n_rounds; r++) {
        rcaps[r] = take(proof, len, pos, cap_n * 4, &bad);
        if (bad) return -120;
        orc_chal_observe_cap(ch, rcaps[r], cap_n);
        betas[r] = orc_chal_get_ext(ch);
        sum_ab += p->arity_bits[r];
    }
    if (sum_ab > log_n) return -120;
    const size_t flen = n >> sum_ab;
    gl2 *fpoly = (gl2 *)malloc((flen ? flen : 1) * sizeof(gl2));
    for (size_t i = 0; i < flen; i++) {
        const uint64_t *e = take(proof, len, pos, 2, &bad);
        fpoly[i] = gl2_make(e[0], e[1]);
        orc_chal_observe_ext(ch, fpoly[i]);
    }
    uint64_t pw = *take(proof, len, pos, 1, &bad), resp;
    if (bad) { free(fpoly); return -120; }
    if (p->pow_rule == ORC_POW_HASH) {
        uint64_t cur_hash[4];
        for (int i = 0; i < 4; i++) cur_hash[i] = orc_chal_get(ch);
        resp = pow_response(ch, cur_hash, ORC_POW_HASH, pw);
    } else {
        orc_chal_observe(ch, pw);
        resp = orc_chal_get(ch);
    }
    if (p->pow_bits && (resp >> (64 - p->pow_bits)) != 0) { free(fpoly); return -121; }
    /* PrecomputedReducedOpenings: per batch sum_j alpha^j opened_j */
    gl2 *red = (gl2 *)malloc(n_batches * sizeof(gl2)), *shf = (gl2 *)malloc(n_batches * sizeof(gl2));
    for (size_t b = 0; b < n_batches; b++) {
        size_t k = batch_len(&batches[b]);
        gl2 acc = gl2_from(0);
        for (size_t j = k; j-- > 0;) acc = gl2_add(gl2_mul(acc, alpha), opened[b][j]);
        red[b] = acc;
        shf[b] = gl2_pow(alpha, (uint64_t)k);
    }
    const uint64_t wm = gl_root_of_unity(log_m);
    const uint64_t **rows = (const uint64_t **)malloc(n_oracles * sizeof(*rows));
    for (unsigned qi = 0; qi < p->num_queries && !rc; qi++) {
        size_t x = (size_t)(orc_chal_get(ch) % m);
        for (size_t o = 0; o < n_oracles; o++) {
            const size_t ll = (size_t)ncols[o] + (size_t)n_salt[o];
            const unsigned ns = log_m - p->cap_height;
            rows[o] = take(proof, len, pos, ll, &bad);
            const uint64_t *sib = take(proof, len, pos, (size_t)ns * 4, &bad);
            if (bad) { rc = -122; break; }
            if (!orc_merkle_verify(rows[o], ll, x, sib, ns, caps[o], p->cap_height)) { rc = -123 - (int)(o < 3 ? o : 3); break; }
        }
        if (rc) break;
        uint64_t sub_x = gl_mul(GL_GEN, gl_pow(wm, bitrev32((uint32_t)x, log_m)));
        /* fri_combine_initial: salt words are never combined (unsalted_eval) */
        gl2 sum = gl2_from(0);
        for (size_t b = 0; b < n_batches; b++) {
            gl2 acc = gl2_from(0), ap = gl2_from(1);
            for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
                const orc_poly_range *rg = &batches[b].ranges[r];
                for (uint32_t c = rg->col_begin; c < rg->col_end; c++) {
                    acc = gl2_add(acc, gl2_scale(ap, rows[rg->oracle][c]));
                    ap = gl2_mul(ap, alpha);
                }
            }
            gl2 num = gl2_sub(acc, red[b]), den = gl2_sub(gl2_from(sub_x), batches[b].point);
            sum = gl2_add(gl2_mul(sum, shf[b]), gl2_mul(num, gl2_inv(den)));
        }
        gl2 old = gl2_scale(sum, sub_x); /* the final polynomial was multiplied by X */
        size_t xi = x;
        unsigned log_tree = log_m;
        for (unsigned r = 0; r < p->n_rounds; r++) {
            const unsigned ab = p->arity_bits[r];
            const size_t arity = (size_t)1 << ab;
            log_tree -= ab;
            const unsigned ns = log_tree > p->cap_height ? log_tree - p->cap_height : 0;
            const uint64_t *ev = take(proof, len, pos, 2 * arity, &bad);
            const uint64_t *sib = take(proof, len, pos, (size_t)ns * 4, &bad);
            if (bad) { rc = -130; break; }
            size_t within = xi & (arity - 1);
            gl2 evs[16];
            for (size_t k = 0; k < arity; k++) evs[k] = gl2_make(ev[2 * k], ev[2 * k + 1]);
            if (!gl2_eq(evs[within], old)) { rc = -131; break; }
            /* compute_evaluation: interpolate {(coset_start g^i, evs[bitrev(i)])} and evaluate at beta */
            uint64_t g = gl_root_of_unity(ab);
            uint64_t rev_within = bitrev32((uint32_t)within, ab);
            uint64_t coset_start = gl_mul(sub_x, gl_pow(g, arity - rev_within));
            uint64_t pts[16];
            for (size_t i = 0; i < arity; i++) pts[i] = gl_mul(coset_start, gl_pow(g, (uint64_t)i));
            gl2 acc = gl2_from(0);
            for (size_t i = 0; i < arity; i++) {
                gl2 numr = evs[bitrev32((uint32_t)i, ab)];
                uint64_t den = 1;
                for (size_t k = 0; k < arity; k++) {
                    if (k == i) continue;
                    numr = gl2_mul(numr, gl2_sub(betas[r], gl2_from(pts[k])));
                    den = gl_mul(den, gl_sub(pts[i], pts[k]));
                }
                acc = gl2_add(acc, gl2_scale(numr, gl_inv(den)));
            }
            old = acc;
            xi >>= ab;
            unsigned cap_h = p->cap_height < log_tree ? p->cap_height : log_tree;
            if (!orc_merkle_verify(ev, 2 * arity, xi, sib, ns, rcaps[r], cap_h)) { rc = -132; break; }
            sub_x = gl_pow(sub_x, arity);
        }
        if (rc) break;
        if (!gl2_eq(eval_poly_ext(fpoly, flen, gl2_from(sub_x)), old)) { rc = -133; break; }
    }
    free(rows); free(red); free(shf); free(fpoly);
    return rc;
}

/* ---------------- generic opening proofs ---------------- */
static gl2 eval_poly_base(const uint64_t *c, size_t n, gl2 x) {
    gl2 acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, x), gl2_from(c[i]));
    return acc;
}

int orc_fri_prove_openings(const orc_batch *const *oracles, size_t n_oracles, const orc_fri_batch *batches, size_t n_batches,
                           unsigned log_n, const orc_fri_params *p, orc_challenger *ch, uint64_t **proof, size_t *len) {
    const size_t n = (size_t)1 << log_n;
    orc_wbuf pf = {0, 0, 0};
    uint64_t hdr[8] = {FRI_MAGIC, p->n_rounds, 0, p->num_queries, n_oracles, n_batches, 0, log_n};
    orc_wb_push(&pf, hdr, 8);
    /* opened values, observed batch by batch (Challenger::observe_openings) */
    for (size_t b = 0; b < n_batches; b++)
        for (uint32_t r = 0; r < batches[b].n_ranges; r++) {
            const orc_poly_range *rg = &batches[b].ranges[r];
            for (uint32_t c = rg->col_begin; c < rg->col_end; c++) {
                gl2 v = eval_poly_base(oracles[rg->oracle]->coeffs + (size_t)c * n, n, batches[b].point);
                wb_push_ext(&pf, v);
                orc_chal_observe_ext(ch, v);
            }
        }
    gl2 alpha = orc_chal_get_ext(ch);
    gl2 *fin = orc_fri_final_poly(oracles, batches, n_batches, log_n, alpha);
    size_t flen = 0;
    int rc = orc_fri_prove_core(oracles, n_oracles, log_n, p, fin, ch, &pf, &flen);
    free(fin);
    if (rc) { free(pf.w); return rc; }
    pf.w[2] = flen;
    pf.w[6] = pf.len;
    *proof = pf.w;
    *len = pf.len;
    return 0;
}

int orc_fri_verify_openings(const uint64_t *proof, size_t len, const uint64_t *const *caps, const int *ncols, const int *n_salt,
                            size_t n_oracles, const orc_fri_batch *batches, size_t n_batches, unsigned log_n,
                            const orc_fri_params *p, orc_challenger *ch) {
    if (len < 8 || proof[0] != FRI_MAGIC || proof[1] != p->n_rounds || proof[3] != p->num_queries || proof[4] != n_oracles ||
        proof[5] != n_batches || proof[6] != len || proof[7] != log_n)
        return -100;
    {   /* header word 2 = length of the final polynomial: fixed by the parameters (an unchecked word would be a second encoding of the proof) */
        unsigned sum_ab = 0;
        for (unsigned r = 0; r < p->n_rounds; r++) sum_ab += p->arity_bits[r];
        if (sum_ab > log_n || proof[2] != (((uint64_t)1 << log_n) >> sum_ab)) return -100;
    }
    for (size_t i = 8; i < len; i++)
        if (proof[i] >= GL_P) return -141; /* canonical field elements only: x + p would be a second encoding of x */
    size_t pos = 8;
    int bad = 0;
    gl2 **opened = (gl2 **)calloc(n_batches, sizeof(gl2 *));
    for (size_t b = 0; b < n_batches; b++) {
        size_t k = batch_len(&batches[b]);
        opened[b] = (gl2 *)malloc((k ? k : 1) * sizeof(gl2));
        for (size_t j = 0; j < k; j++) {
            const uint64_t *e = take(proof, len, &pos, 2, &bad);
            opened[b][j] = gl2_make(e[0], e[1]);
            orc_chal_observe_ext(ch, opened[b][j]);
        }
    }
    int rc = bad ? -106 : 0;
    if (!rc) {
        gl2 alpha = orc_chal_get_ext(ch);
        rc = orc_fri_verify_core(proof, len, &pos, caps, ncols, n_salt, n_oracles, batches, (const gl2 *const *)opened, n_batches,
                                 log_n, p, alpha, ch);
    }
    if (!rc && pos != len) rc = -140;
    for (size_t b = 0; b < n_batches; b++) free(opened[b]);
    free(opened);
    return rc;
}
