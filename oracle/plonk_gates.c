/*
 * oracle/plonk_gates.c -- the outer prover WITH its gates, the gate set given as data (plonk.h, "GATES AS DATA").
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED: restated from the published structure of plonky2 @ 541e127 -- plonk/vanishing_poly.rs
 * (evaluate_gate_constraints, eval_vanishing_poly_base_batch), gates/gate.rs (eval_filtered), gates/selectors.rs (selector groups,
 * UNUSED_SELECTOR), plonk/prover.rs (prove), plonk/verifier.rs -- which the reference reaches through `data.prove(pw)` /
 * `data.verify(proof)` at src/verifier_circuit.rs:253-254 on the circuit it builds at :213-226.  The reference's own gate set lives in
 * un-vendored crates; a circuit here is whatever the caller's programs say (tests: tools/plonk_synth.py -- arithmetic, base-sum,
 * x^7 and public-input style gates in two selector groups).
 */
#include "plonk.h"
#include <stdlib.h>
#include <string.h>

#define PLONK_MAGIC3 0x334b4c5050504953ULL /* "SIPPPLK3" */

uint32_t orc_plonk_num_gate_constraints(const orc_plonk_circuit *c) {
    uint32_t m = 0;
    for (uint32_t g = 0; g < c->num_gates; g++)
        if (c->gates[g].num_constraints > m) m = c->gates[g].num_constraints;
    return m;
}

int orc_plonk_circuit_check(const orc_plonk_circuit *c, const orc_plonk_params *p) {
    if (!c || !p || c->num_wires < p->num_routed_wires || c->num_selectors == 0 || c->num_selectors > c->num_constants || c->num_gates == 0 ||
        !c->gates || (!c->programs && c->program_words))
        return -1;
    for (uint32_t g = 0; g < c->num_gates; g++) {
        const orc_plonk_gate *ga = &c->gates[g];
        if (ga->selector_index >= c->num_selectors || ga->group_lo > ga->row || ga->row >= ga->group_hi || ga->group_hi > c->num_gates) return -2;
        size_t w = ga->prog_offset;
        for (uint32_t j = 0; j < ga->num_constraints; j++) {
            if (w >= c->program_words) return -3;
            const int64_t nm = c->programs[w++];
            if (nm < 0 || nm > 4096) return -3;
            for (int64_t m = 0; m < nm; m++) {
                if (w + 2 > c->program_words) return -3;
                const int64_t nf = c->programs[w + 1];
                w += 2;
                if (nf < 0 || nf > 64 || w + 2 * (size_t)nf > c->program_words) return -3;
                for (int64_t f = 0; f < nf; f++, w += 2) {
                    const int64_t kind = c->programs[w], idx = c->programs[w + 1];
                    if (kind < 0 || kind > 2 || idx < 0 || (kind == 0 && idx >= c->num_wires) || (kind == 1 && idx >= c->num_constants) ||
                        (kind == 2 && idx >= 4))
                        return -4;
                }
            }
        }
    }
    return 0;
}

/* compute_filter (gates/selectors.rs): prod_{i in group, i != row} (i - s), times (UNUSED - s) when the circuit has several selector columns */
static uint64_t filter_base(const orc_plonk_gate *g, uint64_t s, int many) {
    uint64_t f = 1;
    for (uint32_t i = g->group_lo; i < g->group_hi; i++)
        if (i != g->row) f = gl_mul(f, gl_sub(i, s));
    if (many) f = gl_mul(f, gl_sub(ORC_UNUSED_SELECTOR, s));
    return f;
}
static gl2 filter_ext(const orc_plonk_gate *g, gl2 s, int many) {
    gl2 f = gl2_from(1);
    for (uint32_t i = g->group_lo; i < g->group_hi; i++)
        if (i != g->row) f = gl2_mul(f, gl2_sub(gl2_from(i), s));
    if (many) f = gl2_mul(f, gl2_sub(gl2_from(ORC_UNUSED_SELECTOR), s));
    return f;
}

void orc_plonk_gate_constraints_base(const orc_plonk_circuit *c, const uint64_t *wires, const uint64_t *consts, const uint64_t pih[4], uint64_t *out) {
    const uint32_t ngc = orc_plonk_num_gate_constraints(c);
    for (uint32_t j = 0; j < ngc; j++) out[j] = 0;
    for (uint32_t g = 0; g < c->num_gates; g++) {
        const orc_plonk_gate *ga = &c->gates[g];
        const uint64_t f = filter_base(ga, consts[ga->selector_index], c->num_selectors > 1);
        const int64_t *w = c->programs + ga->prog_offset;
        for (uint32_t j = 0; j < ga->num_constraints; j++) {
            const int64_t nm = *w++;
            uint64_t sum = 0;
            for (int64_t m = 0; m < nm; m++) {
                uint64_t t = gl_from_i64(w[0]);
                const int64_t nf = w[1];
                w += 2;
                for (int64_t k = 0; k < nf; k++, w += 2) {
                    const uint64_t v = w[0] == 0 ? wires[w[1]] : w[0] == 1 ? consts[w[1]] : pih[w[1]];
                    t = gl_mul(t, v);
                }
                sum = gl_add(sum, t);
            }
            out[j] = gl_add(out[j], gl_mul(f, sum));
        }
    }
}

void orc_plonk_gate_constraints_ext(const orc_plonk_circuit *c, const gl2 *wires, const gl2 *consts, const uint64_t pih[4], gl2 *out) {
    const uint32_t ngc = orc_plonk_num_gate_constraints(c);
    for (uint32_t j = 0; j < ngc; j++) out[j] = gl2_from(0);
    for (uint32_t g = 0; g < c->num_gates; g++) {
        const orc_plonk_gate *ga = &c->gates[g];
        const gl2 f = filter_ext(ga, consts[ga->selector_index], c->num_selectors > 1);
        const int64_t *w = c->programs + ga->prog_offset;
        for (uint32_t j = 0; j < ga->num_constraints; j++) {
            const int64_t nm = *w++;
            gl2 sum = gl2_from(0);
            for (int64_t m = 0; m < nm; m++) {
                gl2 t = gl2_from(gl_from_i64(w[0]));
                const int64_t nf = w[1];
                w += 2;
                for (int64_t k = 0; k < nf; k++, w += 2) {
                    const gl2 v = w[0] == 0 ? wires[w[1]] : w[0] == 1 ? consts[w[1]] : gl2_from(pih[w[1]]);
                    t = gl2_mul(t, v);
                }
                sum = gl2_add(sum, t);
            }
            out[j] = gl2_add(out[j], gl2_mul(f, sum));
        }
    }
}

static void draw(orc_challenger *ch, uint32_t C, uint64_t *v) {
    for (uint32_t c = 0; c < C; c++) v[c] = orc_chal_get(ch);
}

/* the circuit's constraint terms on the quotient coset 7 <w_{N D}>, natural order: out [num_gate_constraints][N D] */
static void gate_terms_coset(const orc_plonk_circuit *c, const uint64_t *wires_c, const uint64_t *consts_c, unsigned log_n, unsigned log_d,
                             const uint64_t pih[4], uint64_t *out) {
    const size_t n = (size_t)1 << log_n, nd = n << log_d;
    const uint32_t W = c->num_wires, K = c->num_constants, ngc = orc_plonk_num_gate_constraints(c);
    uint64_t *wl = (uint64_t *)malloc((size_t)W * nd * 8), *cl = (uint64_t *)malloc((size_t)K * nd * 8);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t j = 0; j < W + K; j++) {
        if (j < W) orc_coset_lde(wires_c + (size_t)j * n, log_n, log_d, 7, wl + (size_t)j * nd);
        else orc_coset_lde(consts_c + (size_t)(j - W) * n, log_n, log_d, 7, cl + (size_t)(j - W) * nd);
    }
#pragma omp parallel
    {
        uint64_t *wv = (uint64_t *)malloc(W * 8), *cv = (uint64_t *)malloc(K * 8), *t = (uint64_t *)malloc((ngc ? ngc : 1) * 8);
#pragma omp for schedule(static)
        for (size_t i = 0; i < nd; i++) {
            for (uint32_t j = 0; j < W; j++) wv[j] = wl[(size_t)j * nd + i];
            for (uint32_t j = 0; j < K; j++) cv[j] = cl[(size_t)j * nd + i];
            orc_plonk_gate_constraints_base(c, wv, cv, pih, t);
            for (uint32_t j = 0; j < ngc; j++) out[(size_t)j * nd + i] = t[j];
        }
        free(wv); free(cv); free(t);
    }
    free(wl); free(cl);
}

int orc_plonk_prove_gates(const uint64_t *wires, const uint64_t *constants_sigmas, unsigned log_n, const orc_plonk_params *p,
                          const orc_fri_params *fp, const orc_plonk_circuit *c, const uint64_t circuit_digest[4], const uint64_t *public_inputs,
                          uint32_t n_public_inputs, uint64_t **proof, size_t *len) {
    if (orc_plonk_circuit_check(c, p)) return -1;
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, nz = orc_plonk_zs_cols(p);
    const uint32_t W = c->num_wires, K = c->num_constants, ngc = orc_plonk_num_gate_constraints(c);
    if (C == 0 || C > 8 || R == 0 || D < 2) return -1;
    unsigned log_d = 0;
    while ((1u << log_d) < D) log_d++;
    const size_t n = (size_t)1 << log_n, cap_n = (size_t)1 << fp->cap_height;
    uint64_t pih[4];
    orc_hash_no_pad(public_inputs, n_public_inputs, pih);
    orc_batch *bs = orc_batch_from_values(constants_sigmas, (size_t)K + R, log_n, fp->rate_bits, fp->cap_height);
    orc_batch *bw = orc_batch_from_values(wires, W, log_n, fp->rate_bits, fp->cap_height);
    orc_challenger ch;
    orc_chal_init(&ch);
    orc_chal_observe_many(&ch, circuit_digest, 4);
    orc_chal_observe_many(&ch, pih, 4);
    orc_chal_observe_cap(&ch, orc_batch_cap(bw), cap_n);
    uint64_t betas[8], gammas[8], alphas[8];
    draw(&ch, C, betas);
    draw(&ch, C, gammas);
    uint64_t *zs = (uint64_t *)malloc((size_t)nz * n * 8);
    orc_plonk_zs_partial_products(wires, constants_sigmas + (size_t)K * n, log_n, p, betas, gammas, zs);   /* the first R wires; the sigmas */
    orc_batch *bz = orc_batch_from_values(zs, nz, log_n, fp->rate_bits, fp->cap_height);
    free(zs);
    orc_chal_observe_cap(&ch, orc_batch_cap(bz), cap_n);
    draw(&ch, C, alphas);
    uint64_t *qc = (uint64_t *)malloc((size_t)C * D * n * 8);
    uint64_t *gt = (uint64_t *)malloc((size_t)(ngc ? ngc : 1) * (n << log_d) * 8);
    gate_terms_coset(c, orc_batch_coeffs(bw), orc_batch_coeffs(bs), log_n, log_d, pih, gt);
    int rc = orc_plonk_quotient_chunks_ex(orc_batch_coeffs(bw), orc_batch_coeffs(bs) + (size_t)K * n, orc_batch_coeffs(bz), log_n, p, betas, gammas,
                                          alphas, gt, ngc, qc);
    free(gt);
    orc_batch *bq = rc == 0 ? orc_batch_from_coeffs(qc, (size_t)C * D, log_n, fp->rate_bits, fp->cap_height) : NULL;
    free(qc);
    if (rc) { orc_batch_free(bs); orc_batch_free(bw); orc_batch_free(bz); return rc; }
    orc_chal_observe_cap(&ch, orc_batch_cap(bq), cap_n);
    const gl2 zeta = orc_chal_get_ext(&ch);
    const orc_batch *oracles[4] = {bs, bw, bz, bq};
    const orc_poly_range r0[4] = {{0, 0, K + R}, {1, 0, W}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    orc_fri_batch batches[2] = {{zeta, 4, r0}, {gl2_scale(zeta, gl_root_of_unity(log_n)), 1, r1}};
    uint64_t *op = NULL;
    size_t op_len = 0;
    rc = orc_fri_prove_openings(oracles, 4, batches, 2, log_n, fp, &ch, &op, &op_len);
    if (rc == 0) {
        const size_t total = 16 + 3 * cap_n * 4 + op_len + n_public_inputs;
        uint64_t *out = (uint64_t *)malloc(total * 8);
        const uint64_t h[16] = {PLONK_MAGIC3, log_n, R, D, C, total, W, K, c->num_selectors, c->num_gates, ngc, n_public_inputs, 0, 0, 0, 0};
        memcpy(out, h, sizeof h);
        memcpy(out + 16, orc_batch_cap(bw), cap_n * 32);
        memcpy(out + 16 + cap_n * 4, orc_batch_cap(bz), cap_n * 32);
        memcpy(out + 16 + 2 * cap_n * 4, orc_batch_cap(bq), cap_n * 32);
        memcpy(out + 16 + 3 * cap_n * 4, op, op_len * 8);
        if (n_public_inputs) memcpy(out + 16 + 3 * cap_n * 4 + op_len, public_inputs, (size_t)n_public_inputs * 8);
        *proof = out;
        *len = total;
    }
    free(op);
    orc_batch_free(bs); orc_batch_free(bw); orc_batch_free(bz); orc_batch_free(bq);
    return rc;
}

/* plonk/verifier.rs: the gate constraints at zeta are evaluated from the OPENED constants and wires */
int orc_plonk_verify_gates(const uint64_t *proof, size_t len, const uint64_t *cs_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                           const orc_plonk_circuit *c, const uint64_t circuit_digest[4]) {
    if (orc_plonk_circuit_check(c, p)) return -201;
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, np = orc_plonk_num_prods(p), nz = orc_plonk_zs_cols(p);
    const uint32_t W = c->num_wires, K = c->num_constants, ngc = orc_plonk_num_gate_constraints(c);
    const size_t cap_n = (size_t)1 << fp->cap_height;
    if (len < 16 + 3 * cap_n * 4 || proof[0] != PLONK_MAGIC3 || proof[2] != R || proof[3] != D || proof[4] != C || proof[5] != len || proof[6] != W ||
        proof[7] != K || proof[8] != c->num_selectors || proof[9] != c->num_gates || proof[10] != ngc || proof[12] | proof[13] | proof[14] | proof[15])
        return -201;
    const size_t n_pi = (size_t)proof[11];
    if (n_pi > len - (16 + 3 * cap_n * 4)) return -201;
    const unsigned log_n = (unsigned)proof[1];
    if (log_n < 1 || log_n > 26 || C == 0 || C > 8) return -202;
    for (size_t i = 16; i < len; i++)
        if (proof[i] >= GL_P) return -141;   /* canonical field elements only: x + p would be a second encoding of x */
    uint64_t pih[4];
    orc_hash_no_pad(proof + len - n_pi, n_pi, pih);
    const uint64_t *wcap = proof + 16, *zcap = wcap + cap_n * 4, *qcap = zcap + cap_n * 4, *op = qcap + cap_n * 4;
    const size_t op_len = len - (size_t)(op - proof) - n_pi;
    orc_challenger ch;
    orc_chal_init(&ch);
    orc_chal_observe_many(&ch, circuit_digest, 4);
    orc_chal_observe_many(&ch, pih, 4);
    orc_chal_observe_cap(&ch, wcap, cap_n);
    uint64_t betas[8], gammas[8], alphas[8];
    draw(&ch, C, betas);
    draw(&ch, C, gammas);
    orc_chal_observe_cap(&ch, zcap, cap_n);
    draw(&ch, C, alphas);
    orc_chal_observe_cap(&ch, qcap, cap_n);
    const gl2 zeta = orc_chal_get_ext(&ch);
    const uint64_t *caps[4] = {cs_cap, wcap, zcap, qcap};
    const int ncols[4] = {(int)(K + R), (int)W, (int)nz, (int)(C * D)}, n_salt[4] = {0, 0, 0, 0};
    const orc_poly_range r0[4] = {{0, 0, K + R}, {1, 0, W}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    orc_fri_batch batches[2] = {{zeta, 4, r0}, {gl2_scale(zeta, gl_root_of_unity(log_n)), 1, r1}};
    const size_t n0 = (size_t)K + R + W + nz + (size_t)C * D, n_open = n0 + C;
    if (op_len < 8 + 2 * n_open) return -203;
    gl2 *v = (gl2 *)malloc(n_open * sizeof(gl2));
    for (size_t k = 0; k < n_open; k++) v[k] = gl2_make(op[8 + 2 * k], op[8 + 2 * k + 1]);
    const gl2 *cv = v, *sg = v + K, *wv = v + K + R, *zs = wv + W, *pps = zs + C, *qs = zs + nz, *zs_next = v + n0;
    (void)np;
    gl2 van[8];
    gl2 *gt = (gl2 *)malloc((ngc + 1) * sizeof(gl2));
    orc_plonk_gate_constraints_ext(c, wv, cv, pih, gt);
    orc_plonk_eval_vanishing_ex(zeta, wv, sg, zs, zs_next, pps, log_n, p, betas, gammas, alphas, gt, ngc, van);
    free(gt);
    const gl2 zeta_n = gl2_pow(zeta, (uint64_t)1 << log_n), zh = gl2_sub(zeta_n, gl2_from(1));
    int rc = 0;
    for (uint32_t cc = 0; cc < C && rc == 0; cc++) {
        gl2 acc = gl2_from(0);
        for (uint32_t d = D; d-- > 0;) acc = gl2_add(gl2_mul(acc, zeta_n), qs[cc * D + d]);
        if (!gl2_eq(van[cc], gl2_mul(zh, acc))) rc = -210;
    }
    free(v);
    if (rc) return rc;
    return orc_fri_verify_openings(op, op_len, caps, ncols, n_salt, 4, batches, 2, log_n, fp, &ch);
}
