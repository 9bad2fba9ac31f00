/*
 * oracle/stark.c -- CPU restatement of starky::prover::prove / verifier::verify_stark_proof and plonky2's
 * FRI (prove_openings, fri_proof, verify_fri_proof) for the three SIPP AIRs.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED: upstream (starky + plonky2 @ InternetMaximalism/plonky2
 * 541e127, reference Cargo.toml:21,24) is not vendored; the sequence below follows SURVEY.md App. A.6-A.8
 * and A.11.  Reached from the reference only through src/verifier_circuit.rs:133-135.
 * Deterministic: the proof-of-work witness is the SMALLEST valid nonce.
 *
 * Fiat-Shamir binds the STATEMENT first (this repository's proof format, cfg->fs_rule = 0; upstream's starky of mid-2023
 * started from the trace cap: cfg->fs_rule = 1 reproduces that order): before the trace cap the challenger observes
 *   kind, log_n, num_io, W, P, Q, rate_bits, cap_height, pow_bits, arity_bits, final_poly_bits, num_queries,
 *   num_challenges, pow_rule, pi_per_io, lookup_rule                             (16 elements)
 *   pi_root[4] = Merkle root (two_to_one) over d_io = hash_no_pad(the pi_per_io u32 words of IO record io)
 * Public inputs must be canonical: every Fq element of a record < p (checked by prover and verifier), so the
 * result cells, which the AIR only range-checks to < 2^256, cannot smuggle a non-canonical output into the statement.
 *
 * Flat proof layout (u64 words; shared with the HIP prover):
 *   header[16]: magic, kind, log_n, num_io, W, P, Q, cap_height, n_fri_rounds, final_poly_len, num_queries,
 *               pi_per_io, total_len, rate_bits, arity_bits, fs_rule | lookup_rule << 1
 *   trace_cap | z_cap | quotient_cap                       (each 2^cap_height x 4)
 *   openings: local[W] next[W] z[P] z_next[P] quotient[Q]  (ext: 2 words each)
 *   fri commit caps [n_rounds][2^cap_height x 4] | final_poly[len] ext | pow_witness
 *   queries[num_queries]: for oracle in (trace, z, quotient): leaf[ncols], siblings[(log_m - cap) x 4]
 *                         for round r: evals[16] ext, siblings[max(0, log_m - 4(r+1) - cap) x 4]
 *   public_inputs[num_io x pi_per_io]
 */
#include "stark.h"
#include "fri.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define MAGIC 0x5349505053544b31ULL /* "SIPPSTK1" */

void orc_default_config(orc_config *c) {
    c->rate_bits = 1; c->cap_height = 4; c->pow_bits = 16; c->arity_bits = 4; c->final_poly_bits = 5;
    c->num_queries = 84; c->num_challenges = 2; c->pow_rule = ORC_POW_DUPLEX;
    c->fs_rule = 0; c->lookup_rule = 0;
}

/* ---------------- statement binding ---------------- */
static const uint64_t BN_P64[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static int fq_words_canonical(const uint32_t *w) {
    for (int q = 3; q >= 0; q--) {
        uint64_t v = (uint64_t)w[2 * q] | ((uint64_t)w[2 * q + 1] << 32);
        if (v < BN_P64[q]) return 1;
        if (v > BN_P64[q]) return 0;
    }
    return 0; /* == p */
}
/* every Fq element of every record < p; the exponent (8 words) may be any 256-bit value.
 * Record layouts (reference src/verifier_circuit.rs:92-124): (x, offset, exp_val, output). */
int orc_pis_canonical(int kind, const uint32_t *pis, size_t num_io) {
    kind = orc_record_kind(kind);   /* hardened G1 / G2: the same records */
    if (kind == 3 || kind == 6) { /* MapToG2 records (u, x, y): six Fq elements; pairing records (P, Q, Z): eighteen; no exponent */
        for (size_t k = 0; k < (size_t)(kind == 3 ? 6 : 18) * num_io; k++)
            if (!fq_words_canonical(pis + 8 * k)) return 0;
        return 1;
    }
    const int fe = kind == 0 ? 2 : kind == 1 ? 4 : 12;       /* Fq elements per group element */
    const int ppi = 8 * (3 * fe + 1);
    for (size_t io = 0; io < num_io; io++) {
        const uint32_t *rec = pis + io * ppi;
        for (int k = 0; k < 3 * fe + 1; k++) {
            if (k >= 2 * fe && k < 2 * fe + 1) continue;     /* exp_val */
            if (!fq_words_canonical(rec + 8 * k)) return 0;
        }
    }
    return 1;
}

void orc_pi_root(const uint32_t *pis, size_t num_io, int ppi, uint64_t root[4]) {
    uint64_t *d = (uint64_t *)malloc(num_io * 4 * sizeof(uint64_t));
#pragma omp parallel
    {
        uint64_t *tmp = (uint64_t *)malloc((size_t)ppi * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (size_t io = 0; io < num_io; io++) {
            for (int k = 0; k < ppi; k++) tmp[k] = pis[io * ppi + k];
            orc_hash_no_pad(tmp, (size_t)ppi, d + 4 * io);
        }
        free(tmp);
    }
    for (size_t cnt = num_io; cnt > 1; cnt >>= 1)
        for (size_t i = 0; i < cnt / 2; i++) {
            uint64_t out[4];
            orc_two_to_one(d + 8 * i, d + 8 * i + 4, out);
            memcpy(d + 4 * i, out, 32);
        }
    memcpy(root, d, 32);
    free(d);
}

static void observe_statement(orc_challenger *ch, int kind, unsigned log_n, size_t num_io, int W, int P, int Q,
                              const orc_config *cfg, int ppi, const uint32_t *pis) {
    if (cfg->fs_rule) return;   /* starky's recalled order: nothing before the trace cap (the public inputs stay outside the transcript) */
    uint64_t st[16] = {(uint64_t)kind, log_n, num_io, (uint64_t)W, (uint64_t)P, (uint64_t)Q, cfg->rate_bits, cfg->cap_height,
                       cfg->pow_bits, cfg->arity_bits, cfg->final_poly_bits, cfg->num_queries, cfg->num_challenges,
                       cfg->pow_rule, (uint64_t)ppi, cfg->lookup_rule};
    uint64_t root[4];
    orc_chal_observe_many(ch, st, 16);
    orc_pi_root(pis, num_io, ppi, root);
    orc_chal_observe_many(ch, root, 4);
}

/* ---------------- small helpers ---------------- */
typedef orc_wbuf wbuf;
#define wb_push orc_wb_push
static void wb_push1(wbuf *b, uint64_t v) { wb_push(b, &v, 1); }

static void fri_params_of(const orc_config *c, unsigned log_n, orc_fri_params *p) {
    memset(p, 0, sizeof *p);
    p->rate_bits = c->rate_bits; p->cap_height = c->cap_height; p->pow_bits = c->pow_bits; p->num_queries = c->num_queries;
    p->pow_rule = c->pow_rule; p->hiding = 0;
    orc_fri_const_arity(p, c->arity_bits, c->final_poly_bits, log_n);
}
static void wb_push_ext(wbuf *b, gl2 v) { wb_push1(b, v.c0); wb_push1(b, v.c1); }
static unsigned fri_rounds(const orc_config *c, unsigned degree_bits) {
    orc_fri_params p;
    fri_params_of(c, degree_bits, &p);
    return p.n_rounds;
}

static gl2 eval_poly_base(const uint64_t *c, size_t n, gl2 x) {
    gl2 acc = gl2_from(0);
    for (size_t i = n; i-- > 0;) acc = gl2_add(gl2_mul(acc, x), gl2_from(c[i]));
    return acc;
}

/* selectors at a base / ext point */
static void selectors_base(unsigned log_n, uint64_t x, uint64_t *lf, uint64_t *ll, uint64_t *zl) {
    uint64_t N = (uint64_t)1 << log_n, g = gl_root_of_unity(log_n), gi = gl_inv(g);
    uint64_t zh = gl_sub(gl_pow(x, N), 1), ninv = gl_inv(N % GL_P);
    *lf = gl_mul(gl_mul(zh, ninv), gl_inv(gl_sub(x, 1)));
    *ll = gl_mul(gl_mul(gl_mul(zh, ninv), gi), gl_inv(gl_sub(x, gi)));
    *zl = gl_sub(x, gi);
}
static void selectors_ext(unsigned log_n, gl2 x, gl2 *lf, gl2 *ll, gl2 *zl) {
    uint64_t N = (uint64_t)1 << log_n, g = gl_root_of_unity(log_n), gi = gl_inv(g);
    gl2 zh = gl2_sub(gl2_pow(x, N), gl2_from(1));
    uint64_t ninv = gl_inv(N % GL_P);
    *lf = gl2_mul(gl2_scale(zh, ninv), gl2_inv(gl2_sub(x, gl2_from(1))));
    *ll = gl2_mul(gl2_scale(zh, gl_mul(ninv, gi)), gl2_inv(gl2_sub(x, gl2_from(gi))));
    *zl = gl2_sub(x, gl2_from(gi));
}

/* ---------------- test hook: one-shot tampering with a committed oracle the caller cannot reach ----------------
 * stage 1: Z column values (before their commitment), stage 2: quotient chunk coefficients.  tests/test_oracle_soundness.py */
static struct { int stage, col; size_t row; uint64_t delta; } g_tamper;
void orc_test_tamper(int stage, int col, size_t row, uint64_t delta) {
    g_tamper.stage = stage; g_tamper.col = col; g_tamper.row = row; g_tamper.delta = delta;
}
static void apply_tamper(int stage, uint64_t *cols, size_t n) {
    if (g_tamper.stage != stage) return;
    uint64_t *c = cols + (size_t)g_tamper.col * n + g_tamper.row;
    *c = gl_add(*c, g_tamper.delta % GL_P);
    g_tamper.stage = 0;
}

/* ---------------- prover ---------------- */
int orc_stark_prove(int kind, const uint32_t *ios, size_t num_io, const orc_config *cfg, uint64_t **proof_out,
                    size_t *proof_len) {
    int err = 0;
    orc_trace *t = orc_trace_build(kind, ios, num_io, &err);
    if (!t) return err ? err : -1;
    if (!orc_pis_canonical(kind, t->pis, t->num_io)) { orc_trace_free(t); return -9; }
    err = orc_stark_prove_trace(t, cfg, proof_out, proof_len);
    orc_trace_free(t);
    return err;
}

/* the prover proper, from a filled trace (tests tamper with the trace between orc_trace_build and this call:
 * whatever the cells hold is committed and proved as is; the verifier must then refuse) */
/* ORC_TIMING=1 in the environment: the prover's stages on stderr (seconds since the start of the proof) */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#define STAGE(name) do { if (timing) fprintf(stderr, "[oracle prove] %-18s %.2f s\n", name, now_s() - t0); } while (0)

int orc_stark_prove_trace(const orc_trace *t, const orc_config *cfg, uint64_t **proof_out, size_t *proof_len) {
    int err = 0;
    const int timing = getenv("ORC_TIMING") != NULL;
    const double t0 = now_s();
    const int kind = orc_air_api_kind(t->air);
    const air_spec_t *a = t->air;
    const unsigned log_n = t->log_n, log_m = log_n + cfg->rate_bits;
    const size_t n = (size_t)1 << log_n;
    const int W = t->width, nc = a->n_checked, P = 2 * nc, Q = 4, nm = a->n_main;
    const uint64_t g = gl_root_of_unity(log_n);
    wbuf pf = {0, 0, 0};
    uint64_t hdr[16] = {MAGIC, (uint64_t)kind, log_n, t->num_io, (uint64_t)W, (uint64_t)P, (uint64_t)Q, cfg->cap_height,
                        fri_rounds(cfg, log_n), 0, cfg->num_queries, (uint64_t)a->pi_per_io, 0, cfg->rate_bits, cfg->arity_bits,
                        (uint64_t)(cfg->fs_rule | (cfg->lookup_rule << 1))};
    wb_push(&pf, hdr, 16);
    const size_t cap_n = (size_t)1 << (cfg->cap_height < log_m ? cfg->cap_height : log_m);

    orc_challenger ch;
    orc_chal_init(&ch);
    observe_statement(&ch, kind, log_n, t->num_io, W, P, Q, cfg, a->pi_per_io, t->pis);
    /* 1. trace commitment */
    orc_batch *bt = orc_batch_from_values(t->trace, (size_t)W, log_n, cfg->rate_bits, cfg->cap_height);
    orc_chal_observe_cap(&ch, orc_batch_cap(bt), cap_n);
    wb_push(&pf, orc_batch_cap(bt), cap_n * 4);
    STAGE("trace commitment");
    /* 2. permutation challenges and Z polys */
    uint64_t beta[2], gamma[2];
    for (int i = 0; i < 2; i++) {
        beta[i] = orc_chal_get(&ch); gamma[i] = orc_chal_get(&ch);
        if (cfg->lookup_rule) beta[i] = gamma[i];   /* shared challenge: Z' (pin + g)(ptab + g) = Z (col + g)(tab + g) */
    }
    uint64_t *zv = (uint64_t *)malloc((size_t)P * n * sizeof(uint64_t));
#pragma omp parallel
    {
        uint64_t *num = (uint64_t *)malloc(n * sizeof(uint64_t)), *den = (uint64_t *)malloc(n * sizeof(uint64_t));
        uint64_t *pre = (uint64_t *)malloc(n * sizeof(uint64_t));
#pragma omp for schedule(dynamic)
        for (int zi = 0; zi < P; zi++) {
            int i = zi / nc, j = zi % nc;
            const uint64_t *col = t->trace + (size_t)(a->checked_base + j) * n, *tab = t->trace;
            const uint64_t *pin = t->trace + (size_t)(nm + j) * n, *ptab = t->trace + (size_t)(nm + nc + j) * n;
            for (size_t r = 0; r < n; r++) {
                num[r] = gl_mul(gl_add(col[r], gamma[i]), gl_add(tab[r], beta[i]));
                den[r] = gl_mul(gl_add(pin[r], gamma[i]), gl_add(ptab[r], beta[i]));
            }
            /* batch inversion of den */
            uint64_t acc = 1;
            for (size_t r = 0; r < n; r++) { pre[r] = acc; acc = gl_mul(acc, den[r]); }
            uint64_t inv = gl_inv(acc);
            for (size_t r = n; r-- > 0;) { uint64_t d = den[r]; den[r] = gl_mul(inv, pre[r]); inv = gl_mul(inv, d); }
            uint64_t z = 1;
            uint64_t *out = zv + (size_t)zi * n;
            for (size_t r = 0; r < n; r++) { out[r] = z; z = gl_mul(z, gl_mul(num[r], den[r])); }
        }
        free(num); free(den); free(pre);
    }
    apply_tamper(1, zv, n);
    orc_batch *bz = orc_batch_from_values(zv, (size_t)P, log_n, cfg->rate_bits, cfg->cap_height);
    free(zv);
    orc_chal_observe_cap(&ch, orc_batch_cap(bz), cap_n);
    wb_push(&pf, orc_batch_cap(bz), cap_n * 4);
    STAGE("lookup products");
    /* 3. alphas */
    uint64_t alpha[2];
    alpha[0] = orc_chal_get(&ch); alpha[1] = orc_chal_get(&ch);
    /* 4. quotient: evaluate all constraints on the 2N coset */
    const uint64_t *tl = orc_batch_leaves(bt), *zl = orc_batch_leaves(bz);
    /* The quotient is evaluated on the coset 7 <w_2N> whatever the blowup: its points are the natural LDE indices that are
     * multiples of 2^(rate_bits - 1), i.e. exactly the FIRST 2N leaves in leaf order (bit-reversal moves the zero low bits
     * to the top), at leaf position bitrev_(log_n + 1)(iq) for quotient-domain index iq. */
    const unsigned log_mq = log_n + 1;
    const size_t mq = (size_t)1 << log_mq;
    /* aux LDE, natural order: [n_aux][mq] */
    uint64_t *aux_lde = (uint64_t *)malloc((size_t)(a->n_aux ? a->n_aux : 1) * mq * sizeof(uint64_t));
    {
        size_t nio = t->num_io;
#pragma omp parallel for schedule(dynamic)
        for (int ai = 0; ai < a->n_aux; ai++) {
            uint64_t *co = (uint64_t *)calloc(n, sizeof(uint64_t));
            orc_aux_coeffs(a, t->pis, nio, log_n, ai, co);
            orc_coset_lde(co, log_n, 1, GL_GEN, aux_lde + (size_t)ai * mq);
            free(co);
        }
    }
    /* value-periodic columns on the quotient coset: P_k at (7 w_2N^i)^(N/R) = 7^(N/R) w_2R^i -- 2R distinct values, a coset LDE of P_k */
    const int n_vper = orc_air_n_vper(a), n_per = orc_air_n_per(a);
    const size_t R2 = (size_t)2 << a->log_rows;
    uint64_t *vper_tab = (uint64_t *)malloc((size_t)(n_vper ? n_vper : 1) * R2 * sizeof(uint64_t));
    {
        const uint64_t shift = gl_pow(GL_GEN, (uint64_t)1 << (log_n - (unsigned)a->log_rows));
#pragma omp parallel for schedule(dynamic)
        for (int k = 0; k < n_vper; k++) {
            uint64_t *co = (uint64_t *)malloc((R2 / 2) * sizeof(uint64_t));
            orc_vper_coeffs(a, k, co);
            orc_coset_lde(co, (unsigned)a->log_rows, 1, shift, vper_tab + (size_t)k * R2);
            free(co);
        }
    }
    STAGE("aux / periodic");
    uint64_t *qv = (uint64_t *)malloc(2 * mq * sizeof(uint64_t)); /* [2][mq] natural order */
    {
        uint64_t wm = gl_root_of_unity(log_mq);
        /* x^N on the coset takes 2 values */
        uint64_t sN = gl_pow(GL_GEN, n);
        uint64_t zh_inv[2] = {gl_inv(gl_sub(sN, 1)), gl_inv(gl_sub(gl_neg(sN), 1))};
#pragma omp parallel
        {
            uint64_t *aux = (uint64_t *)malloc(sizeof(uint64_t) * (a->n_aux ? a->n_aux : 1));
            uint64_t *per = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n_per);
#pragma omp for schedule(static)
            for (size_t i = 0; i < mq; i++) {
                uint64_t x = gl_mul(GL_GEN, gl_pow(wm, i));
                size_t j = bitrev32((uint32_t)i, log_mq), jn = bitrev32((uint32_t)((i + 2) & (mq - 1)), log_mq);
                uint64_t lf, ll, zlast, out[2];
                for (int k = 0; k < AIR_N_PERIODIC; k++) per[k] = orc_periodic_base(log_n, k, x);
                for (int k = 0; k < n_vper; k++) per[AIR_N_PERIODIC + k] = vper_tab[(size_t)k * R2 + (i & (R2 - 1))];
                selectors_base(log_n, x, &lf, &ll, &zlast);
                for (int ai = 0; ai < a->n_aux; ai++) aux[ai] = aux_lde[(size_t)ai * mq + i];
                orc_eval_base(a, tl + j * W, tl + jn * W, aux, per, zl + j * P, zl + jn * P, lf, ll, zlast, alpha, beta, gamma, out);
                qv[i] = gl_mul(out[0], zh_inv[i & 1]);
                qv[mq + i] = gl_mul(out[1], zh_inv[i & 1]);
            }
            free(aux); free(per);
        }
    }
    free(aux_lde); free(vper_tab);
    STAGE("quotient values");
    /* coset iFFT -> 2 polys of 2N coefficients -> 4 chunks of N */
    uint64_t *qc = (uint64_t *)malloc((size_t)Q * n * sizeof(uint64_t));
    {
        uint64_t si = gl_inv(GL_GEN);
        for (int i = 0; i < 2; i++) {
            uint64_t *p = qv + (size_t)i * mq;
            orc_ifft(p, log_mq);
            uint64_t s = 1;
            for (size_t k = 0; k < mq; k++) { p[k] = gl_mul(p[k], s); s = gl_mul(s, si); }
            memcpy(qc + (size_t)(2 * i) * n, p, n * sizeof(uint64_t));
            memcpy(qc + (size_t)(2 * i + 1) * n, p + n, n * sizeof(uint64_t));
        }
    }
    free(qv);
    apply_tamper(2, qc, n);
    orc_batch *bq = orc_batch_from_coeffs(qc, (size_t)Q, log_n, cfg->rate_bits, cfg->cap_height);
    free(qc);
    orc_chal_observe_cap(&ch, orc_batch_cap(bq), cap_n);
    wb_push(&pf, orc_batch_cap(bq), cap_n * 4);
    /* 5. zeta, openings */
    gl2 zeta = orc_chal_get_ext(&ch);
    if (gl2_eq(gl2_pow(zeta, n), gl2_from(1))) { err = -20; goto done; }
    gl2 gzeta = gl2_scale(zeta, g);
    const uint64_t *ct = orc_batch_coeffs(bt), *cz = orc_batch_coeffs(bz), *cq = orc_batch_coeffs(bq);
    size_t n_open = (size_t)(2 * W + 2 * P + Q);
    gl2 *op = (gl2 *)malloc(n_open * sizeof(gl2));
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < W; c++) { op[c] = eval_poly_base(ct + (size_t)c * n, n, zeta); op[W + c] = eval_poly_base(ct + (size_t)c * n, n, gzeta); }
#pragma omp parallel for schedule(dynamic)
    for (int c = 0; c < P; c++) { op[2 * W + c] = eval_poly_base(cz + (size_t)c * n, n, zeta); op[2 * W + P + c] = eval_poly_base(cz + (size_t)c * n, n, gzeta); }
    for (int c = 0; c < Q; c++) op[2 * W + 2 * P + c] = eval_poly_base(cq + (size_t)c * n, n, zeta);
    for (size_t k = 0; k < n_open; k++) wb_push_ext(&pf, op[k]);
    /* observe: batch0 = local | z | quotient ; batch1 = next | z_next */
    for (int c = 0; c < W; c++) orc_chal_observe_ext(&ch, op[c]);
    for (int c = 0; c < P; c++) orc_chal_observe_ext(&ch, op[2 * W + c]);
    for (int c = 0; c < Q; c++) orc_chal_observe_ext(&ch, op[2 * W + 2 * P + c]);
    for (int c = 0; c < W; c++) orc_chal_observe_ext(&ch, op[W + c]);
    for (int c = 0; c < P; c++) orc_chal_observe_ext(&ch, op[2 * W + P + c]);
    /* 6. FRI (oracle/fri.c, the generic PolynomialBatch::prove_openings): batch 0 = everything at zeta, batch 1 = trace | Z at g zeta */
    gl2 fa = orc_chal_get_ext(&ch);
    {
        const orc_batch *oracles[3] = {bt, bz, bq};
        const orc_poly_range r0[3] = {{0, 0, (uint32_t)W}, {1, 0, (uint32_t)P}, {2, 0, (uint32_t)Q}};
        const orc_fri_batch batches[2] = {{zeta, 3, r0}, {gzeta, 2, r0}};
        orc_fri_params fp;
        fri_params_of(cfg, log_n, &fp);
        gl2 *fin = orc_fri_final_poly(oracles, batches, 2, log_n, fa);
        size_t flen = 0;
        int frc = orc_fri_prove_core(oracles, 3, log_n, &fp, fin, &ch, &pf, &flen);
        free(fin);
        if (frc) err = frc;
        pf.w[9] = flen;
    }
    for (size_t k = 0; k < t->num_io * (size_t)a->pi_per_io; k++) wb_push1(&pf, t->pis[k]);
    pf.w[12] = pf.len;
    free(op);
done:
    orc_batch_free(bt); orc_batch_free(bz); orc_batch_free(bq);
    if (err) { free(pf.w); return err; }
    *proof_out = pf.w;
    *proof_len = pf.len;
    return 0;
}

void orc_free(void *p) { free(p); }

/* ---------------- verifier ---------------- */
typedef struct { const uint64_t *w; size_t pos, len; int bad; } rbuf;
static const uint64_t *rb_take(rbuf *b, size_t n) {
    if (b->pos + n > b->len) { b->bad = 1; return b->w; }
    const uint64_t *p = b->w + b->pos;
    b->pos += n;
    return p;
}
static gl2 rb_ext(rbuf *b) { const uint64_t *p = rb_take(b, 2); return gl2_make(p[0], p[1]); }

int orc_stark_verify(const uint64_t *proof, size_t len, const orc_config *cfg) {
    rbuf rb = {proof, 0, len, 0};
    const uint64_t *h = rb_take(&rb, 16);
    if (rb.bad || h[0] != MAGIC) return -100;
    for (int i = 1; i < 16; i++)
        if (h[i] >> 32) return -100; /* every header word is a small integer: no high bits to hide a second encoding in */
    /* every body word is a field element in canonical form: x + p hashes and computes like x, so a non-canonical word would be
     * a second encoding of the same proof */
    for (size_t i = 16; i < len; i++)
        if (proof[i] >= GL_P) return -141;
    int kind = (int)h[1];
    unsigned log_n = (unsigned)h[2];
    size_t num_io = (size_t)h[3];
    int W = (int)h[4], P = (int)h[5], Q = (int)h[6];
    const unsigned log_rows = orc_kind_log_rows(kind);
    if (kind < 0 || kind > 6 || log_n < 10 || log_n > 26 || log_n <= log_rows || num_io != ((size_t)1 << (log_n - log_rows))) return -101;
    const air_spec_t *a = orc_air_get(kind, log_n);
    if (!a || W != orc_air_width(a) || P != 2 * a->n_checked || Q != 4 || h[7] != cfg->cap_height ||
        h[10] != cfg->num_queries || (int)h[11] != a->pi_per_io || h[12] != len || h[13] != cfg->rate_bits ||
        h[14] != cfg->arity_bits || h[15] != (uint64_t)(cfg->fs_rule | (cfg->lookup_rule << 1)) || cfg->fs_rule > 1 || cfg->lookup_rule > 1)
        return -102;
    const size_t n = (size_t)1 << log_n;
    const unsigned rounds = fri_rounds(cfg, log_n);
    if (h[8] != rounds) return -103;
    const size_t cap_n = (size_t)1 << cfg->cap_height;
    /* public inputs are at the end */
    size_t n_pi = num_io * (size_t)a->pi_per_io;
    if (len < 16 + n_pi) return -104;
    uint32_t *pis = (uint32_t *)malloc((n_pi ? n_pi : 1) * sizeof(uint32_t));
    for (size_t k = 0; k < n_pi; k++) {
        uint64_t v = proof[len - n_pi + k];
        if (v >> 32) { free(pis); return -105; }
        pis[k] = (uint32_t)v;
    }
    int rc = 0;
    gl2 *op = NULL, *auxz = NULL;
    if (!orc_pis_canonical(kind, pis, num_io)) { free(pis); return -108; }
    /* the statement is about group elements (src/verifier_circuit.rs:92-124): a prover that skipped its own curve check must not
     * get a proof about points of another curve accepted (the row constraints alone are satisfiable for any two points) */
    if (!orc_records_on_curve(kind, pis, num_io)) { free(pis); return -109; }
    orc_challenger ch;
    orc_chal_init(&ch);
    observe_statement(&ch, kind, log_n, num_io, W, P, Q, cfg, a->pi_per_io, pis);
    const uint64_t *trace_cap = rb_take(&rb, cap_n * 4);
    orc_chal_observe_cap(&ch, trace_cap, cap_n);
    uint64_t beta[2], gamma[2], alpha[2];
    for (int i = 0; i < 2; i++) {
        beta[i] = orc_chal_get(&ch); gamma[i] = orc_chal_get(&ch);
        if (cfg->lookup_rule) beta[i] = gamma[i];
    }
    const uint64_t *z_cap = rb_take(&rb, cap_n * 4);
    orc_chal_observe_cap(&ch, z_cap, cap_n);
    alpha[0] = orc_chal_get(&ch); alpha[1] = orc_chal_get(&ch);
    const uint64_t *q_cap = rb_take(&rb, cap_n * 4);
    orc_chal_observe_cap(&ch, q_cap, cap_n);
    gl2 zeta = orc_chal_get_ext(&ch);
    size_t n_open = (size_t)(2 * W + 2 * P + Q);
    op = (gl2 *)malloc(n_open * sizeof(gl2));
    for (size_t k = 0; k < n_open; k++) op[k] = rb_ext(&rb);
    if (rb.bad) { rc = -106; goto out; }
    for (int c = 0; c < W; c++) orc_chal_observe_ext(&ch, op[c]);
    for (int c = 0; c < P; c++) orc_chal_observe_ext(&ch, op[2 * W + c]);
    for (int c = 0; c < Q; c++) orc_chal_observe_ext(&ch, op[2 * W + 2 * P + c]);
    for (int c = 0; c < W; c++) orc_chal_observe_ext(&ch, op[W + c]);
    for (int c = 0; c < P; c++) orc_chal_observe_ext(&ch, op[2 * W + P + c]);
    /* constraint check at zeta */
    {
        gl2 lf, ll, zlast, out[2];
        gl2 *per = (gl2 *)malloc(sizeof(gl2) * (size_t)orc_air_n_per(a));
        for (int k = 0; k < AIR_N_PERIODIC; k++) per[k] = orc_periodic_ext(log_n, k, zeta);
        {   /* the AIR's value-periodic columns (selectors and constants of the pairing schedule): interpolated from the tables */
            uint64_t *vc = (uint64_t *)malloc(sizeof(uint64_t) << a->log_rows);
            for (int k = 0; k < orc_air_n_vper(a); k++) {
                orc_vper_coeffs(a, k, vc);
                per[AIR_N_PERIODIC + k] = orc_vper_ext(a, log_n, vc, zeta);
            }
            free(vc);
        }
        selectors_ext(log_n, zeta, &lf, &ll, &zlast);
        auxz = (gl2 *)malloc(sizeof(gl2) * (a->n_aux ? a->n_aux : 1));
        uint64_t *co = (uint64_t *)malloc(num_io * sizeof(uint64_t));
        for (int ai = 0; ai < a->n_aux; ai++) {
            orc_aux_coeffs(a, pis, num_io, log_n, ai, co);
            auxz[ai] = eval_poly_base(co, num_io, zeta);
        }
        free(co);
        orc_eval_ext(a, op, op + W, auxz, per, op + 2 * W, op + 2 * W + P, lf, ll, zlast, alpha, beta, gamma, out);
        free(per);
        gl2 zh = gl2_sub(gl2_pow(zeta, n), gl2_from(1));
        if (gl2_eq(zh, gl2_from(0))) { rc = -107; goto out; }
        gl2 zn = gl2_pow(zeta, n);
        for (int i = 0; i < 2; i++) {
            gl2 qz = gl2_add(op[2 * W + 2 * P + 2 * i], gl2_mul(zn, op[2 * W + 2 * P + 2 * i + 1]));
            if (!gl2_eq(out[i], gl2_mul(zh, qz))) { rc = -110 - i; goto out; }
        }
    }
    /* FRI (oracle/fri.c) */
    {
        gl2 fa = orc_chal_get_ext(&ch);
        gl2 gzeta = gl2_scale(zeta, gl_root_of_unity(log_n));
        const uint64_t *caps3[3] = {trace_cap, z_cap, q_cap};
        const int ncols3[3] = {W, P, Q}, salt3[3] = {0, 0, 0};
        const orc_poly_range r0[3] = {{0, 0, (uint32_t)W}, {1, 0, (uint32_t)P}, {2, 0, (uint32_t)Q}};
        const orc_fri_batch batches[2] = {{zeta, 3, r0}, {gzeta, 2, r0}};
        /* opened values in batch order: local | z | quotient, then next | z_next */
        gl2 *o0 = (gl2 *)malloc((size_t)(W + P + Q) * sizeof(gl2)), *o1 = (gl2 *)malloc((size_t)(W + P) * sizeof(gl2));
        for (int c = 0; c < W; c++) { o0[c] = op[c]; o1[c] = op[W + c]; }
        for (int c = 0; c < P; c++) { o0[W + c] = op[2 * W + c]; o1[W + c] = op[2 * W + P + c]; }
        for (int c = 0; c < Q; c++) o0[W + P + c] = op[2 * W + 2 * P + c];
        const gl2 *opened[2] = {o0, o1};
        orc_fri_params fp;
        fri_params_of(cfg, log_n, &fp);
        if ((size_t)h[9] != (n >> (fp.n_rounds * cfg->arity_bits))) rc = -120;
        if (!rc) rc = orc_fri_verify_core(proof, len, &rb.pos, caps3, ncols3, salt3, 3, batches, opened, 2, log_n, &fp, fa, &ch);
        free(o0); free(o1);
    }
    if (!rc && rb.pos + n_pi != len) rc = -140;
out:
    free(op); free(auxz); free(pis);
    return rc;
}
