/*
 * oracle/pairing.c -- CPU restatement of the final-pairing AIR's primary witness (API kind 6): one optimal ate pairing
 * e(P, Q) per 512-row block, executed row by row from the schedule tools/pairing_sched.py emits into data/air_tables.h.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.  The reference asks for this value at src/bin/bls_aggregation.rs:76-77
 * (`pairing_circuit(final_A, final_B)` connected to `final_Z`) and natively at src/prover_native.rs:20 / src/verifier_native.rs:80
 * (`plonky2_bn254_pairing::pairing::pairing`, recalled to restate arkworks' `Bn254::pairing`); neither ark-ec 0.4 nor plonky2-bn254-pairing @ fe5c3a8 is
 * vendored.  Restated: the optimal ate Miller loop in affine coordinates (the algorithm of oracle/py/bn254.py::miller_loop) and
 * ark-ec's final exponentiation as recalled -- easy part, then the chain y0 .. y16 of Bn::final_exponentiation (Fuentes-Castaneda
 * et al.), whose value is f^(lambda (p^12 - 1)/r), lambda = 2u(6u^2 + 3u + 1).  The second reading is
 * tools/pairing_sched.py::simulate (big-int Python) and, for the value alone, oracle/py/bn254.py::pairing (a plain power);
 * tests/test_oracle_pairing_air.py compares the three.
 *
 * Arithmetic: Fq12 as Fq2[w]/(w^6 - xi), xi = 9 + u (six Fq2 coefficients; the cell order of the AIR: component t = 2 i + c).
 */
#include "pairing.h"
#include <string.h>

typedef struct { fq2 c[6]; } t6;
typedef struct { fq2 c[3]; } s3;     /* Fq6 = Fq2[v]/(v^3 - xi), v = w^2 */

static fq2 f2z(void) { fq2 r = {fq_zero(), fq_zero()}; return r; }
static fq2 f2_neg(fq2 a) { fq2 r = {fq_neg(a.c0), fq_neg(a.c1)}; return r; }
static fq2 f2_conj(fq2 a) { fq2 r = {a.c0, fq_neg(a.c1)}; return r; }
static int f2_eq(fq2 a, fq2 b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
static fq2 f2_small(uint64_t k, fq2 a) { fq s = fq_from_u64(k); fq2 r = {fq_mul(s, a.c0), fq_mul(s, a.c1)}; return r; }
static fq2 f2_mul_xi(fq2 a) { /* (a0 + a1 u)(9 + u) */
    fq nine = fq_from_u64(9);
    fq2 r = {fq_sub(fq_mul(nine, a.c0), a.c1), fq_add(a.c0, fq_mul(nine, a.c1))};
    return r;
}

static t6 t6_zero(void) { t6 r; for (int i = 0; i < 6; i++) r.c[i] = f2z(); return r; }

static t6 t6_mul(const t6 *a, const t6 *b) {
    fq2 d[11];
    for (int k = 0; k < 11; k++) d[k] = f2z();
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) d[i + j] = fq2_add(d[i + j], fq2_mul(a->c[i], b->c[j]));
    t6 r;
    for (int k = 0; k < 6; k++) r.c[k] = k < 5 ? fq2_add(d[k], f2_mul_xi(d[k + 6])) : d[k];
    return r;
}

static s3 s3_mul(const s3 *a, const s3 *b) {
    fq2 d[5];
    for (int k = 0; k < 5; k++) d[k] = f2z();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) d[i + j] = fq2_add(d[i + j], fq2_mul(a->c[i], b->c[j]));
    s3 r = {{fq2_add(d[0], f2_mul_xi(d[3])), fq2_add(d[1], f2_mul_xi(d[4])), d[2]}};
    return r;
}
static int s3_inv(const s3 *a, s3 *out) {
    fq2 a0 = a->c[0], a1 = a->c[1], a2 = a->c[2];
    fq2 c0 = fq2_sub(fq2_mul(a0, a0), f2_mul_xi(fq2_mul(a1, a2)));
    fq2 c1 = fq2_sub(f2_mul_xi(fq2_mul(a2, a2)), fq2_mul(a0, a1));
    fq2 c2 = fq2_sub(fq2_mul(a1, a1), fq2_mul(a0, a2));
    fq2 t = fq2_add(fq2_mul(a0, c0), f2_mul_xi(fq2_add(fq2_mul(a2, c1), fq2_mul(a1, c2))));
    if (fq2_is_zero(t)) return -1;
    fq2 ti = fq2_inv(t);
    out->c[0] = fq2_mul(c0, ti); out->c[1] = fq2_mul(c1, ti); out->c[2] = fq2_mul(c2, ti);
    return 0;
}
/* a = E + O w with E, O in Fq6 (w^2 = v): 1 / a = (E - O w) / (E^2 - v O^2) */
static int t6_inv(const t6 *a, t6 *out) {
    s3 E = {{a->c[0], a->c[2], a->c[4]}}, O = {{a->c[1], a->c[3], a->c[5]}};
    s3 E2 = s3_mul(&E, &E), O2 = s3_mul(&O, &O);
    s3 N = {{fq2_sub(E2.c[0], f2_mul_xi(O2.c[2])), fq2_sub(E2.c[1], O2.c[0]), fq2_sub(E2.c[2], O2.c[1])}};
    s3 Ni;
    if (s3_inv(&N, &Ni)) return -1;
    s3 Ei = s3_mul(&E, &Ni), Oi = s3_mul(&O, &Ni);
    for (int i = 0; i < 3; i++) { out->c[2 * i] = Ei.c[i]; out->c[2 * i + 1] = f2_neg(Oi.c[i]); }
    return 0;
}

static fq limbs_to_fq(const int64_t *l) {
    uint16_t v[16];
    for (int i = 0; i < 16; i++) v[i] = (uint16_t)l[i];
    return fq_from_limbs16(v);
}
static t6 gconst(int idx) {
    t6 r;
    for (int i = 0; i < 6; i++) {
        r.c[i].c0 = limbs_to_fq(AIR_PAIRING_GCONST[idx] + 32 * i);
        r.c[i].c1 = limbs_to_fq(AIR_PAIRING_GCONST[idx] + 32 * i + 16);
    }
    return r;
}

/* ---- curve membership: the PUBLIC conditions of a record (oracle/stark.c's verifier and both provers) ---- */
static fq2 read2(const uint32_t *w) { fq2 r = {fq_from_u32(w), fq_from_u32(w + 8)}; return r; }
static int on_twist(fq2 x, fq2 y) { /* (9 + u)(y^2 - x^3) = 3 */
    fq2 d = f2_mul_xi(fq2_sub(fq2_mul(y, y), fq2_mul(fq2_mul(x, x), x)));
    return fq_eq(d.c0, fq_from_u64(3)) && fq_is_zero(d.c1);
}
static int on_g1(fq x, fq y) { return fq_eq(fq_sub(fq_mul(y, y), fq_mul(fq_mul(x, x), x)), fq_from_u64(3)); }
/* [r - 1] Q == -Q by affine double-and-add; a point of order r never meets a degenerate step on the way, so any such step
 * (or a different end point) means Q is not in the r-torsion */
static const uint64_t BN_RM1[4] = {0x43e1f593f0000000ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static int g2_order_r(fq2 qx, fq2 qy) {
    fq2 tx = qx, ty = qy;
    for (int i = 252; i >= 0; i--) {        /* r - 1 has 254 bits; the top bit is the starting point */
        fq2 den = fq2_add(ty, ty);
        if (fq2_is_zero(den)) return 0;
        fq2 lam = fq2_mul(f2_small(3, fq2_mul(tx, tx)), fq2_inv(den));
        fq2 x3 = fq2_sub(fq2_sub(fq2_mul(lam, lam), tx), tx);
        ty = fq2_sub(fq2_mul(lam, fq2_sub(tx, x3)), ty);
        tx = x3;
        if ((BN_RM1[i >> 6] >> (i & 63)) & 1) {
            den = fq2_sub(qx, tx);
            if (fq2_is_zero(den)) return 0;
            lam = fq2_mul(fq2_sub(qy, ty), fq2_inv(den));
            x3 = fq2_sub(fq2_sub(fq2_mul(lam, lam), tx), qx);
            ty = fq2_sub(fq2_mul(lam, fq2_sub(tx, x3)), ty);
            tx = x3;
        }
    }
    return f2_eq(tx, qx) && f2_eq(ty, f2_neg(qy));
}
int orc_pairing_record_ok(const uint32_t *rec) {
    fq_init();
    if (!on_g1(fq_from_u32(rec), fq_from_u32(rec + 8))) return 0;
    fq2 qx = read2(rec + 16), qy = read2(rec + 32);
    return on_twist(qx, qy) && g2_order_r(qx, qy);
}

/* ---- cell writers (the column-major trace of oracle/air.c) ---- */
static void put_fq(uint64_t *tr, size_t n, int col, size_t row, fq v, int cpl) {
    uint16_t l[16];
    fq_to_limbs16(v, l);
    for (int i = 0; i < 16; i++) {
        if (cpl <= 1) tr[(size_t)(col + i) * n + row] = l[i];
        else { tr[(size_t)(col + 2 * i) * n + row] = l[i] & 0xff; tr[(size_t)(col + 2 * i + 1) * n + row] = l[i] >> 8; }
    }
}
static void put_f2(uint64_t *tr, size_t n, int col, size_t row, fq2 v, int cpl) { /* cpl 0: unchecked */
    put_fq(tr, n, col, row, v.c0, cpl);
    put_fq(tr, n, col + 16 * (cpl ? cpl : 1), row, v.c1, cpl);
}
static void put_t6(uint64_t *tr, size_t n, int col, size_t row, const t6 *v, int cpl) {
    for (int i = 0; i < 6; i++) put_f2(tr, n, col + 32 * (cpl ? cpl : 1) * i, row, v->c[i], cpl);
}

enum { L_PX, L_PY, L_QX, L_QY, L_Q1X, L_Q1Y, L_Q2X, L_Q2Y, L_TX, L_TY, L_QSX, L_QSY, L_FXC, L_FYC, L_A, L_B, L_G, L_REG, L_C, L_S0, L_N };

/* runs the schedule on one record (P: 16 words, Q: 32 words).  tr != NULL: every primary cell of rows [row0, row0 + 512) of the
 * column-major trace (n rows per column) is written.  out_words receives the 96 words of the result.  Returns 0, -1 for a
 * degenerate step / a value without inverse, -20 for tables this code does not understand. */
int orc_pairing_run(const air_spec_t *a, uint64_t *tr, size_t n, size_t row0, const uint32_t *rec, uint32_t *out_words) {
    fq_init();
    if (AIR_PAIRING_ROWS != 512 || AIR_PAIRING_NREG > 8) return -20;
    const int cpl = a ? a->cells_per_limb : 1;
    const int32_t *lay = (a && cpl == 2) ? AIR_PAIRING_LAYOUT_U8 : AIR_PAIRING_LAYOUT_U16;
    if (tr && (!a || a->kind != 6 || a->log_rows != 9 || lay[L_C] != a->checked_base)) return -20;
    const fq px = fq_from_u32(rec), py = fq_from_u32(rec + 8);
    const fq2 qx = read2(rec + 16), qy = read2(rec + 32);
    /* the twist's Frobenius constants xi^((p-1)/3), xi^((p-1)/2) = the w^2 and w^3 entries of the p-power constant vector */
    const t6 g1c = gconst(2);
    const fq2 frob_x = g1c.c[2], frob_y = g1c.c[3];
    t6 regs[8];
    for (int k = 0; k < 8; k++) regs[k] = t6_zero();
    fq2 tx = f2z(), ty = f2z(), q1x = f2z(), q1y = f2z(), q2x = f2z(), q2y = f2z();
    const fq2 pxe = {px, fq_zero()}, pye = {py, fq_zero()};
    for (int t = 0; t < 512; t++) {
        const int8_t *r = AIR_PAIRING_SCHED[t];
        const int fop = r[0], ra = r[1], rb = r[2], rd = r[4], gop = r[5];
        fq2 S[5] = {f2z(), f2z(), f2z(), f2z(), f2z()}, qsx = f2z(), qsy = f2z();
        if (gop == 5) {
            S[0] = fq2_mul(f2_conj(qx), frob_x);
            S[1] = fq2_mul(f2_conj(qy), frob_y);
            S[2] = fq2_mul(f2_conj(S[0]), frob_x);
            S[3] = f2_neg(fq2_mul(f2_conj(S[1]), frob_y));
        } else if (gop != 0) {
            fq2 num, den, xb;
            if (gop == 1) {
                den = fq2_add(ty, ty);
                num = f2_small(3, fq2_mul(tx, tx));
                xb = tx;
            } else {
                qsx = gop == 2 ? qx : gop == 3 ? q1x : q2x;
                qsy = gop == 2 ? qy : gop == 3 ? q1y : q2y;
                den = fq2_sub(qsx, tx);
                num = fq2_sub(qsy, ty);
                xb = qsx;
            }
            if (fq2_is_zero(den)) return -1;
            fq2 lam = fq2_mul(num, fq2_inv(den));
            S[0] = lam;
            S[1] = fq2_sub(fq2_sub(fq2_mul(lam, lam), tx), xb);
            S[2] = fq2_sub(fq2_mul(lam, fq2_sub(tx, S[1])), ty);
            S[3] = f2_neg(fq2_mul(lam, pxe));
            S[4] = fq2_sub(fq2_mul(lam, tx), ty);
        }
        const int gi = AIR_PAIRING_GIDX[t];
        const t6 G = gconst(gi);
        t6 A = ra >= 0 ? regs[ra] : t6_zero(), B = t6_zero(), C = t6_zero();
        if (fop == 1) {
            B = regs[rb];
            C = t6_mul(&A, &B);
        } else if (fop == 2) {
            B.c[0] = pye; B.c[1] = S[3]; B.c[3] = S[4];
            C = t6_mul(&A, &B);
        } else if (fop == 3) {
            if (t6_inv(&A, &C)) return -1;
            B = C;
        } else if (fop == 4) {
            for (int i = 0; i < 6; i++) C.c[i] = fq2_mul(AIR_PAIRING_GCONJ[gi] ? f2_conj(A.c[i]) : A.c[i], G.c[i]);
        }
        if (tr) {
            const size_t row = row0 + (size_t)t;
            put_fq(tr, n, lay[L_PX], row, px, 0); put_fq(tr, n, lay[L_PY], row, py, 0);
            put_f2(tr, n, lay[L_QX], row, qx, 0); put_f2(tr, n, lay[L_QY], row, qy, 0);
            put_f2(tr, n, lay[L_Q1X], row, q1x, 0); put_f2(tr, n, lay[L_Q1Y], row, q1y, 0);
            put_f2(tr, n, lay[L_Q2X], row, q2x, 0); put_f2(tr, n, lay[L_Q2Y], row, q2y, 0);
            put_f2(tr, n, lay[L_TX], row, tx, 0); put_f2(tr, n, lay[L_TY], row, ty, 0);
            put_f2(tr, n, lay[L_QSX], row, qsx, 0); put_f2(tr, n, lay[L_QSY], row, qsy, 0);
            put_f2(tr, n, lay[L_FXC], row, frob_x, 0); put_f2(tr, n, lay[L_FYC], row, frob_y, 0);
            put_t6(tr, n, lay[L_A], row, &A, 0); put_t6(tr, n, lay[L_B], row, &B, 0); put_t6(tr, n, lay[L_G], row, &G, 0);
            for (int k = 0; k < AIR_PAIRING_NREG; k++) put_t6(tr, n, lay[L_REG] + 192 * k, row, &regs[k], 0);
            put_t6(tr, n, lay[L_C], row, &C, cpl);
            for (int sl = 0; sl < 5; sl++) put_f2(tr, n, lay[L_S0] + 32 * cpl * sl, row, S[sl], cpl);
        }
        if (rd >= 0) regs[rd] = fop == 0 ? G : C;
        if (gop == 5) { tx = qx; ty = qy; q1x = S[0]; q1y = S[1]; q2x = S[2]; q2y = S[3]; }
        else if (gop == 1 || gop == 2 || gop == 3) { tx = S[1]; ty = S[2]; }
    }
    const t6 *res = &regs[AIR_PAIRING_RESULT_REG];
    fq nine = fq_from_u64(9);
    for (int i = 0; i < 6; i++) {           /* MyFq12 coefficients: c_i = a_i - 9 b_i, c_{i+6} = b_i */
        fq_to_u32(fq_sub(res->c[i].c0, fq_mul(nine, res->c[i].c1)), out_words + 8 * i);
        fq_to_u32(res->c[i].c1, out_words + 8 * (i + 6));
    }
    return 0;
}

/* the value alone: e(P, Q) of a record's first 48 words (points checked: on the curves, Q of order r) */
int orc_pairing(const uint32_t *pq, uint32_t *out96) {
    if (!orc_pairing_record_ok(pq)) return -2;
    return orc_pairing_run(NULL, NULL, 0, 0, pq, out96);
}
