/*
 * oracle/plonk.c -- see plonk.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (restated from the published structure of plonky2's
 * plonk/prover.rs, plonk/vanishing_poly.rs, plonk/plonk_common.rs, plonk/verifier.rs @ 541e127, which the reference reaches through
 * `data.prove(pw)` / `data.verify(proof)` at src/verifier_circuit.rs:253-254).
 */
#include "plonk.h"
#include <stdlib.h>
#include <string.h>

#define PLONK_MAGIC 0x314b4c5050504953ULL  /* "SIPPPLK1" */
#define PLONK_MAGIC2 0x324b4c5050504953ULL /* "SIPPPLK2": with gate-constraint terms and public inputs */

static int plonk_prove_impl(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p, const orc_fri_params *fp,
                            const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], const uint64_t *public_inputs,
                            uint32_t n_public_inputs, const orc_plonk_gates *g, uint64_t **proof, size_t *len);

uint64_t orc_plonk_k_i(uint32_t j) { return gl_pow(7, j); }   /* get_unique_coset_shifts: g^j, g = MULTIPLICATIVE_GROUP_GENERATOR */

void orc_plonk_sigmas_from_perm(const uint32_t *perm, uint32_t num_routed, unsigned log_n, uint64_t *sigmas) {
    const size_t n = (size_t)1 << log_n;
    uint64_t *pw = (uint64_t *)malloc(n * sizeof(uint64_t));
    const uint64_t w = gl_root_of_unity(log_n);
    pw[0] = 1;
    for (size_t i = 1; i < n; i++) pw[i] = gl_mul(pw[i - 1], w);
    for (uint32_t j = 0; j < num_routed; j++)
        for (size_t i = 0; i < n; i++) {
            const uint32_t t = perm[(size_t)j * n + i];
            sigmas[(size_t)j * n + i] = gl_mul(orc_plonk_k_i(t >> log_n), pw[t & (n - 1)]);
        }
    free(pw);
}

/* plonk/prover.rs wires_permutation_partial_products_and_zs (per challenge), all_wires_permutation_partial_products + the
 * re-ordering in prove(): Z columns first */
void orc_plonk_zs_partial_products(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p,
                                   const uint64_t *betas, const uint64_t *gammas, uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, np = orc_plonk_num_prods(p), m = np + 1;
    uint64_t *k_is = (uint64_t *)malloc(R * sizeof(uint64_t));
    for (uint32_t j = 0; j < R; j++) k_is[j] = orc_plonk_k_i(j);
    const uint64_t w = gl_root_of_unity(log_n);
    uint64_t *chunk = (uint64_t *)malloc(n * m * sizeof(uint64_t));   /* quotient_chunk_products of every row */
    for (uint32_t c = 0; c < C; c++) {
        const uint64_t beta = betas[c], gamma = gammas[c];
        uint64_t x = 1;
        for (size_t i = 0; i < n; i++) {
            for (uint32_t q = 0; q < m; q++) {
                uint64_t num = 1, den = 1;
                for (uint32_t j = q * D; j < (q + 1) * D && j < R; j++) {
                    const uint64_t wv = wires[(size_t)j * n + i];
                    num = gl_mul(num, gl_add(gl_add(wv, gl_mul(beta, gl_mul(k_is[j], x))), gamma));
                    den = gl_mul(den, gl_add(gl_add(wv, gl_mul(beta, sigmas[(size_t)j * n + i])), gamma));
                }
                chunk[i * m + q] = gl_mul(num, gl_inv(den));
            }
            x = gl_mul(x, w);
        }
        uint64_t z_x = 1;
        for (size_t i = 0; i < n; i++) {
            /* partial_products_and_z_gx, then the last entry (Z(g x)) is swapped for Z(x) */
            uint64_t acc = z_x;
            out[(size_t)c * n + i] = z_x;
            for (uint32_t q = 0; q < m; q++) {
                acc = gl_mul(acc, chunk[i * m + q]);
                if (q < np) out[((size_t)C + (size_t)c * np + q) * n + i] = acc;
            }
            z_x = acc;
        }
    }
    free(chunk);
    free(k_is);
}

/* the permutation terms of eval_vanishing_poly at one point (base field or extension, by macro-free duplication below) */
static void vanishing_terms_base(uint64_t x, uint64_t l0, const uint64_t *wv, const uint64_t *sg, const uint64_t *zs, const uint64_t *zs_next,
                                 const uint64_t *pps, const orc_plonk_params *p, const uint64_t *k_is, const uint64_t *betas,
                                 const uint64_t *gammas, uint64_t *terms) {
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, np = orc_plonk_num_prods(p), m = np + 1;
    size_t t = 0;
    for (uint32_t c = 0; c < C; c++) terms[t++] = gl_mul(l0, gl_sub(zs[c], 1));          /* vanishing_z_1_terms */
    for (uint32_t c = 0; c < C; c++)                                                       /* check_partial_products */
        for (uint32_t q = 0; q < m; q++) {
            uint64_t num = 1, den = 1;
            for (uint32_t j = q * D; j < (q + 1) * D && j < R; j++) {
                num = gl_mul(num, gl_add(gl_add(wv[j], gl_mul(betas[c], gl_mul(k_is[j], x))), gammas[c]));
                den = gl_mul(den, gl_add(gl_add(wv[j], gl_mul(betas[c], sg[j])), gammas[c]));
            }
            const uint64_t prev = q == 0 ? zs[c] : pps[c * np + q - 1], next = q == np ? zs_next[c] : pps[c * np + q];
            terms[t++] = gl_sub(gl_mul(prev, num), gl_mul(next, den));
        }
}

int orc_plonk_quotient_chunks(const uint64_t *wires_c, const uint64_t *sigmas_c, const uint64_t *zs_c, unsigned log_n,
                              const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                              uint64_t *out) {
    return orc_plonk_quotient_chunks_ex(wires_c, sigmas_c, zs_c, log_n, p, betas, gammas, alphas, NULL, 0, out);
}

/* the product gates of the synthetic test circuit on the quotient coset 7 <w_{N D}>, natural order: term k = w_{3k} w_{3k+1} - w_{3k+2} */
void orc_plonk_gate_terms_coset(const uint64_t *wires_c, unsigned log_n, unsigned log_d, const orc_plonk_gates *g, uint64_t *out) {
    const size_t n = (size_t)1 << log_n, nd = n << log_d;
    uint64_t *wl = (uint64_t *)malloc((size_t)3 * g->num_mul * nd * 8);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t j = 0; j < 3 * g->num_mul; j++) orc_coset_lde(wires_c + (size_t)j * n, log_n, log_d, 7, wl + (size_t)j * nd);
    for (uint32_t k = 0; k < g->num_mul; k++)
        for (size_t i = 0; i < nd; i++)
            out[(size_t)k * nd + i] = gl_sub(gl_mul(wl[(size_t)(3 * k) * nd + i], wl[(size_t)(3 * k + 1) * nd + i]), wl[(size_t)(3 * k + 2) * nd + i]);
    free(wl);
}

/* eval_vanishing_poly_base_batch with the circuit's gate-constraint terms supplied by the caller: vanishing_terms =
 * vanishing_z_1_terms ++ vanishing_partial_products_terms ++ constraint_terms (plonk/vanishing_poly.rs), ONE reduce_with_powers over all
 * of them per challenge.  gate_terms: [n_gate_terms][N D], natural order of the coset (NULL / 0: the permutation argument alone). */
int orc_plonk_quotient_chunks_ex(const uint64_t *wires_c, const uint64_t *sigmas_c, const uint64_t *zs_c, unsigned log_n,
                                 const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                                 const uint64_t *gate_terms, uint32_t n_gate_terms, uint64_t *out) {
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, np = orc_plonk_num_prods(p), m = np + 1;
    unsigned log_d = 0;
    while ((1u << log_d) < D) log_d++;
    if ((1u << log_d) != D || D < 2) return -2;
    const size_t n = (size_t)1 << log_n, nd = n << log_d;
    const uint32_t nz = C * (1 + np), n_perm = C + C * m, n_terms = n_perm + n_gate_terms;
    /* coset LDEs on 7 <w_{N D}>, natural order */
    uint64_t *wl = (uint64_t *)malloc((size_t)R * nd * 8), *sl = (uint64_t *)malloc((size_t)R * nd * 8), *zl = (uint64_t *)malloc((size_t)nz * nd * 8);
#pragma omp parallel for schedule(dynamic)
    for (uint32_t j = 0; j < R; j++) {
        orc_coset_lde(wires_c + (size_t)j * n, log_n, log_d, 7, wl + (size_t)j * nd);
        orc_coset_lde(sigmas_c + (size_t)j * n, log_n, log_d, 7, sl + (size_t)j * nd);
    }
#pragma omp parallel for schedule(dynamic)
    for (uint32_t j = 0; j < nz; j++) orc_coset_lde(zs_c + (size_t)j * n, log_n, log_d, 7, zl + (size_t)j * nd);
    uint64_t *k_is = (uint64_t *)malloc(R * 8);
    for (uint32_t j = 0; j < R; j++) k_is[j] = orc_plonk_k_i(j);
    uint64_t *qv = (uint64_t *)malloc((size_t)C * nd * 8);
    const uint64_t w = gl_root_of_unity(log_n + log_d);
    uint64_t *xs = (uint64_t *)malloc(nd * 8);                 /* the coset points 7 w^i */
    xs[0] = 7;
    for (size_t i = 1; i < nd; i++) xs[i] = gl_mul(xs[i - 1], w);
#pragma omp parallel
    {
        uint64_t *wv = (uint64_t *)malloc(R * 8), *sg = (uint64_t *)malloc(R * 8), *zs = (uint64_t *)malloc(nz * 8), *zn = (uint64_t *)malloc(C * 8);
        uint64_t *terms = (uint64_t *)malloc(n_terms * 8);
#pragma omp for schedule(static)
        for (size_t i = 0; i < nd; i++) {
            const uint64_t x = xs[i];
            const uint64_t zh = gl_sub(gl_pow(x, n), 1);
            const uint64_t l0 = gl_mul(zh, gl_inv(gl_mul((uint64_t)n, gl_sub(x, 1))));   /* L_0(x) = (x^n - 1) / (n (x - 1)) */
            for (uint32_t j = 0; j < R; j++) { wv[j] = wl[(size_t)j * nd + i]; sg[j] = sl[(size_t)j * nd + i]; }
            for (uint32_t j = 0; j < nz; j++) zs[j] = zl[(size_t)j * nd + i];
            for (uint32_t c = 0; c < C; c++) zn[c] = zl[(size_t)c * nd + ((i + D) & (nd - 1))];   /* Z(g x): D steps on the coset */
            vanishing_terms_base(x, l0, wv, sg, zs, zn, zs + C, p, k_is, betas, gammas, terms);
            for (uint32_t k = 0; k < n_gate_terms; k++) terms[n_perm + k] = gate_terms[(size_t)k * nd + i];
            const uint64_t zhi = gl_inv(zh);
            for (uint32_t c = 0; c < C; c++) {
                uint64_t acc = 0;                               /* reduce_with_powers: sum_k alpha^k term_k */
                for (size_t k = n_terms; k-- > 0;) acc = gl_add(gl_mul(acc, alphas[c]), terms[k]);
                qv[(size_t)c * nd + i] = gl_mul(acc, zhi);
            }
        }
        free(wv); free(sg); free(zs); free(zn); free(terms);
    }
    /* coset_ifft(7): values on 7 <w> -> coefficients; chunks of N coefficients */
    const uint64_t g_inv = gl_inv(7);
    for (uint32_t c = 0; c < C; c++) {
        uint64_t *a = qv + (size_t)c * nd;
        orc_ifft(a, log_n + log_d);
        uint64_t f = 1;
        for (size_t j = 0; j < nd; j++) {
            a[j] = gl_mul(a[j], f);
            f = gl_mul(f, g_inv);
        }
        memcpy(out + (size_t)c * nd, a, nd * 8);                /* [C][D][N] == [C D][N] */
    }
    free(wl); free(sl); free(zl); free(k_is); free(qv); free(xs);
    return 0;
}

void orc_plonk_eval_vanishing(gl2 x, const gl2 *wv, const gl2 *sg, const gl2 *zs, const gl2 *zs_next, const gl2 *pps, unsigned log_n,
                              const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas, gl2 *out) {
    orc_plonk_eval_vanishing_ex(x, wv, sg, zs, zs_next, pps, log_n, p, betas, gammas, alphas, NULL, 0, out);
}

void orc_plonk_eval_vanishing_ex(gl2 x, const gl2 *wv, const gl2 *sg, const gl2 *zs, const gl2 *zs_next, const gl2 *pps, unsigned log_n,
                                 const orc_plonk_params *p, const uint64_t *betas, const uint64_t *gammas, const uint64_t *alphas,
                                 const gl2 *gate_terms, uint32_t n_gate_terms, gl2 *out) {
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, np = orc_plonk_num_prods(p), m = np + 1;
    const uint64_t n = (uint64_t)1 << log_n;
    const uint32_t n_terms = C + C * m + n_gate_terms;
    gl2 *terms = (gl2 *)malloc(n_terms * sizeof(gl2));
    const gl2 zh = gl2_sub(gl2_pow(x, n), gl2_from(1));
    const gl2 l0 = gl2_mul(zh, gl2_inv(gl2_scale(gl2_sub(x, gl2_from(1)), n)));
    size_t t = 0;
    for (uint32_t c = 0; c < C; c++) terms[t++] = gl2_mul(l0, gl2_sub(zs[c], gl2_from(1)));
    for (uint32_t c = 0; c < C; c++)
        for (uint32_t q = 0; q < m; q++) {
            gl2 num = gl2_from(1), den = gl2_from(1);
            for (uint32_t j = q * D; j < (q + 1) * D && j < R; j++) {
                const gl2 sid = gl2_scale(x, gl_mul(betas[c], orc_plonk_k_i(j)));
                num = gl2_mul(num, gl2_add(gl2_add(wv[j], sid), gl2_from(gammas[c])));
                den = gl2_mul(den, gl2_add(gl2_add(wv[j], gl2_scale(sg[j], betas[c])), gl2_from(gammas[c])));
            }
            const gl2 prev = q == 0 ? zs[c] : pps[c * np + q - 1], next = q == np ? zs_next[c] : pps[c * np + q];
            terms[t++] = gl2_sub(gl2_mul(prev, num), gl2_mul(next, den));
        }
    for (uint32_t k = 0; k < n_gate_terms; k++) terms[t++] = gate_terms[k];
    for (uint32_t c = 0; c < C; c++) {
        gl2 acc = gl2_from(0);
        for (size_t k = n_terms; k-- > 0;) acc = gl2_add(gl2_scale(acc, alphas[c]), terms[k]);
        out[c] = acc;
    }
    free(terms);
}

/* ---- the flow ---- */
static void draw(orc_challenger *ch, uint32_t C, uint64_t *v) {
    for (uint32_t c = 0; c < C; c++) v[c] = orc_chal_get(ch);
}

int orc_plonk_perm_prove(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p, const orc_fri_params *fp,
                         const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], uint64_t **proof, size_t *len) {
    return plonk_prove_impl(wires, sigmas, log_n, p, fp, circuit_digest, public_inputs_hash, NULL, 0, NULL, proof, len);
}

/* prove() with the gates' constraint terms in the quotient and the public inputs in the transcript (plonk/prover.rs: public_inputs_hash =
 * hash_n_to_hash_no_pad(public_inputs); the challenger observes circuit_digest, that hash, the wires cap).  Flat proof "SIPPPLK2":
 *   header[8]: magic, log_n, num_routed_wires, max_degree, num_challenges, total_len, n_gate_terms, n_public_inputs
 *   wires cap | zs_partial_products cap | quotient cap | opening proof | public_inputs */
int orc_plonk_prove_ex(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p, const orc_fri_params *fp,
                       const uint64_t circuit_digest[4], const uint64_t *public_inputs, uint32_t n_public_inputs, const orc_plonk_gates *g,
                       uint64_t **proof, size_t *len) {
    uint64_t pih[4];
    if (!g || 3 * g->num_mul > p->num_routed_wires || p->max_degree < 2) return -1;
    orc_hash_no_pad(public_inputs, n_public_inputs, pih);
    return plonk_prove_impl(wires, sigmas, log_n, p, fp, circuit_digest, pih, public_inputs, n_public_inputs, g, proof, len);
}

static int plonk_prove_impl(const uint64_t *wires, const uint64_t *sigmas, unsigned log_n, const orc_plonk_params *p, const orc_fri_params *fp,
                            const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], const uint64_t *public_inputs,
                            uint32_t n_public_inputs, const orc_plonk_gates *g, uint64_t **proof, size_t *len) {
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, nz = orc_plonk_zs_cols(p);
    if (C == 0 || C > 8 || R == 0 || D < 2) return -1;
    const size_t n = (size_t)1 << log_n, cap_n = (size_t)1 << fp->cap_height;
    orc_batch *bs = orc_batch_from_values(sigmas, R, log_n, fp->rate_bits, fp->cap_height);
    orc_batch *bw = orc_batch_from_values(wires, R, log_n, fp->rate_bits, fp->cap_height);
    orc_challenger ch;
    orc_chal_init(&ch);
    orc_chal_observe_many(&ch, circuit_digest, 4);
    orc_chal_observe_many(&ch, public_inputs_hash, 4);
    orc_chal_observe_cap(&ch, orc_batch_cap(bw), cap_n);
    uint64_t betas[8], gammas[8], alphas[8];
    draw(&ch, C, betas);
    draw(&ch, C, gammas);
    uint64_t *zs = (uint64_t *)malloc((size_t)nz * n * 8);
    orc_plonk_zs_partial_products(wires, sigmas, log_n, p, betas, gammas, zs);
    orc_batch *bz = orc_batch_from_values(zs, nz, log_n, fp->rate_bits, fp->cap_height);
    free(zs);
    orc_chal_observe_cap(&ch, orc_batch_cap(bz), cap_n);
    draw(&ch, C, alphas);
    uint64_t *qc = (uint64_t *)malloc((size_t)C * D * n * 8);
    uint64_t *gt = NULL;
    const uint32_t n_gt = g ? g->num_mul : 0;
    if (n_gt) {
        unsigned log_d = 0;
        while ((1u << log_d) < D) log_d++;
        gt = (uint64_t *)malloc((size_t)n_gt * (n << log_d) * 8);
        orc_plonk_gate_terms_coset(orc_batch_coeffs(bw), log_n, log_d, g, gt);
    }
    int rc = orc_plonk_quotient_chunks_ex(orc_batch_coeffs(bw), orc_batch_coeffs(bs), orc_batch_coeffs(bz), log_n, p, betas, gammas, alphas, gt,
                                          n_gt, qc);
    free(gt);
    orc_batch *bq = rc == 0 ? orc_batch_from_coeffs(qc, (size_t)C * D, log_n, fp->rate_bits, fp->cap_height) : NULL;
    free(qc);
    if (rc) { orc_batch_free(bs); orc_batch_free(bw); orc_batch_free(bz); return rc; }
    orc_chal_observe_cap(&ch, orc_batch_cap(bq), cap_n);
    const gl2 zeta = orc_chal_get_ext(&ch);
    const orc_batch *oracles[4] = {bs, bw, bz, bq};
    const orc_poly_range r0[4] = {{0, 0, R}, {1, 0, R}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    orc_fri_batch batches[2] = {{zeta, 4, r0}, {gl2_scale(zeta, gl_root_of_unity(log_n)), 1, r1}};
    uint64_t *op = NULL;
    size_t op_len = 0;
    rc = orc_fri_prove_openings(oracles, 4, batches, 2, log_n, fp, &ch, &op, &op_len);
    if (rc == 0) {
        const size_t total = 8 + 3 * cap_n * 4 + op_len + (g ? n_public_inputs : 0);
        uint64_t *out = (uint64_t *)malloc(total * 8);
        const uint64_t h[8] = {g ? PLONK_MAGIC2 : PLONK_MAGIC, log_n, R, D, C, total, n_gt, g ? n_public_inputs : 0};
        memcpy(out, h, sizeof h);
        memcpy(out + 8, orc_batch_cap(bw), cap_n * 32);
        memcpy(out + 8 + cap_n * 4, orc_batch_cap(bz), cap_n * 32);
        memcpy(out + 8 + 2 * cap_n * 4, orc_batch_cap(bq), cap_n * 32);
        memcpy(out + 8 + 3 * cap_n * 4, op, op_len * 8);
        if (g && n_public_inputs) memcpy(out + 8 + 3 * cap_n * 4 + op_len, public_inputs, (size_t)n_public_inputs * 8);
        *proof = out;
        *len = total;
    }
    free(op);
    orc_batch_free(bs); orc_batch_free(bw); orc_batch_free(bz); orc_batch_free(bq);
    return rc;
}

static int plonk_verify_impl(const uint64_t *proof, size_t len, const uint64_t *sigmas_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                             const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], const orc_plonk_gates *g);

int orc_plonk_perm_verify(const uint64_t *proof, size_t len, const uint64_t *sigmas_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                          const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4]) {
    return plonk_verify_impl(proof, len, sigmas_cap, p, fp, circuit_digest, public_inputs_hash, NULL);
}

/* plonk/verifier.rs with the gates: the public inputs come with the proof, their hash enters the transcript, and the gate-constraint terms
 * at zeta are evaluated from the opened wires (the synthetic circuit's product gates) */
int orc_plonk_verify_ex(const uint64_t *proof, size_t len, const uint64_t *sigmas_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                        const uint64_t circuit_digest[4], const orc_plonk_gates *g) {
    if (!g || len < 8 || proof[0] != PLONK_MAGIC2 || proof[6] != g->num_mul || proof[7] > len) return -201;
    uint64_t pih[4];
    const size_t n_pi = (size_t)proof[7];
    orc_hash_no_pad(proof + len - n_pi, n_pi, pih);
    return plonk_verify_impl(proof, len, sigmas_cap, p, fp, circuit_digest, pih, g);
}

static int plonk_verify_impl(const uint64_t *proof, size_t len, const uint64_t *sigmas_cap, const orc_plonk_params *p, const orc_fri_params *fp,
                             const uint64_t circuit_digest[4], const uint64_t public_inputs_hash[4], const orc_plonk_gates *g) {
    const uint32_t R = p->num_routed_wires, D = p->max_degree, C = p->num_challenges, np = orc_plonk_num_prods(p), nz = orc_plonk_zs_cols(p);
    const size_t cap_n = (size_t)1 << fp->cap_height;
    if (len < 8 + 3 * cap_n * 4 || proof[0] != (g ? PLONK_MAGIC2 : PLONK_MAGIC) || proof[2] != R || proof[3] != D || proof[4] != C || proof[5] != len)
        return -201;
    if (g && 3 * g->num_mul > R) return -201;
    const size_t n_pi = g ? (size_t)proof[7] : 0;
    if (n_pi > len - (8 + 3 * cap_n * 4)) return -201;
    const unsigned log_n = (unsigned)proof[1];
    if (log_n < 1 || log_n > 26 || C == 0 || C > 8) return -202;
    const uint64_t *wcap = proof + 8, *zcap = wcap + cap_n * 4, *qcap = zcap + cap_n * 4, *op = qcap + cap_n * 4;
    const size_t op_len = len - (size_t)(op - proof) - n_pi;
    orc_challenger ch;
    orc_chal_init(&ch);
    orc_chal_observe_many(&ch, circuit_digest, 4);
    orc_chal_observe_many(&ch, public_inputs_hash, 4);
    orc_chal_observe_cap(&ch, wcap, cap_n);
    uint64_t betas[8], gammas[8], alphas[8];
    draw(&ch, C, betas);
    draw(&ch, C, gammas);
    orc_chal_observe_cap(&ch, zcap, cap_n);
    draw(&ch, C, alphas);
    orc_chal_observe_cap(&ch, qcap, cap_n);
    const gl2 zeta = orc_chal_get_ext(&ch);
    const uint64_t *caps[4] = {sigmas_cap, wcap, zcap, qcap};
    const int ncols[4] = {(int)R, (int)R, (int)nz, (int)(C * D)}, n_salt[4] = {0, 0, 0, 0};
    const orc_poly_range r0[4] = {{0, 0, R}, {1, 0, R}, {2, 0, nz}, {3, 0, C * D}}, r1[1] = {{2, 0, C}};
    orc_fri_batch batches[2] = {{zeta, 4, r0}, {gl2_scale(zeta, gl_root_of_unity(log_n)), 1, r1}};
    /* the opened values sit behind the 8-word header of the opening proof, batch by batch (two words per extension element) */
    const size_t n0 = (size_t)2 * R + nz + (size_t)C * D, n_open = n0 + C;
    if (op_len < 8 + 2 * n_open) return -203;
    gl2 *v = (gl2 *)malloc(n_open * sizeof(gl2));
    for (size_t k = 0; k < n_open; k++) v[k] = gl2_make(op[8 + 2 * k], op[8 + 2 * k + 1]);
    const gl2 *sg = v, *wv = v + R, *zs = v + 2 * R, *pps = zs + C, *qs = v + 2 * R + nz, *zs_next = v + n0;
    (void)np;
    /* plonk/verifier.rs: vanishing(zeta) == Z_H(zeta) * sum_d zeta^(N d) q_{c,d}(zeta) for every challenge */
    gl2 van[8];
    const uint32_t n_gt = g ? g->num_mul : 0;
    gl2 *gt = (gl2 *)malloc((n_gt + 1) * sizeof(gl2));
    for (uint32_t k = 0; k < n_gt; k++) gt[k] = gl2_sub(gl2_mul(wv[3 * k], wv[3 * k + 1]), wv[3 * k + 2]);
    orc_plonk_eval_vanishing_ex(zeta, wv, sg, zs, zs_next, pps, log_n, p, betas, gammas, alphas, gt, n_gt, van);
    free(gt);
    const gl2 zeta_n = gl2_pow(zeta, (uint64_t)1 << log_n), zh = gl2_sub(zeta_n, gl2_from(1));
    int rc = 0;
    for (uint32_t c = 0; c < C && rc == 0; c++) {
        gl2 acc = gl2_from(0);
        for (uint32_t d = D; d-- > 0;) acc = gl2_add(gl2_mul(acc, zeta_n), qs[c * D + d]);
        if (!gl2_eq(van[c], gl2_mul(zh, acc))) rc = -210;
    }
    free(v);
    if (rc) return rc;
    return orc_fri_verify_openings(op, op_len, caps, ncols, n_salt, 4, batches, 2, log_n, fp, &ch);
}
