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
static void put_fq(uint64_t *tr, size_t n, int col, size_t row, fq v, int cpl) {   /* cpl 0 / 1: one cell per limb; 2: two (lo, hi byte) */
    uint16_t l[16];
    fq_to_limbs16(v, l);
    for (int i = 0; i < 16; i++) {
        if (cpl <= 1) tr[(size_t)(col + i) * n + row] = l[i];
        else { tr[(size_t)(col + 2 * i) * n + row] = l[i] & 0xff; tr[(size_t)(col + 2 * i + 1) * n + row] = l[i] >> 8; }
    }
}

enum { L_PX, L_PY, L_QX, L_QY, L_Q1X, L_Q1Y, L_Q2X, L_Q2Y, L_TX, L_TY, L_QSX, L_QSY, L_FXC, L_FYC, L_SR, L_GC, L_A, L_B, L_CACC, L_REG, L_RES, L_N };
/* row types and descriptor fields of tools/pairing_rows.py */
enum { T_IDLE, T_FMUL, T_FFROB, T_FINVW, T_FINVC, T_FCOPY, T_FCOMMIT, T_GW, T_GSL, T_GX3, T_GY3, T_GL1, T_GL3, T_GFQ };
enum { F_TYP, F_T, F_RA, F_RB, F_BSEL, F_GC, F_LD, F_SK, F_CHM, F_END };

static t6 t6_from12(const fq v[12]) { t6 r; for (int i = 0; i < 6; i++) { r.c[i].c0 = v[2 * i]; r.c[i].c1 = v[2 * i + 1]; } return r; }
static void t6_to12(const t6 *a, fq v[12]) { for (int i = 0; i < 6; i++) { v[2 * i] = a->c[i].c0; v[2 * i + 1] = a->c[i].c1; } }

/* runs the ROW PROGRAM (AIR_PAIRING_ROWPROG: one modular identity per row, 2^13 rows) on one record (P: 16 words, Q: 32 words).
 * tr != NULL: every primary cell of rows [row0, row0 + 2^13) of the column-major trace (n rows per column) is written.  out_words
 * receives the 96 words of the result.  Returns 0, -1 for a degenerate step / a value without inverse / a failing check row, -20 for
 * tables this code does not understand. */
int orc_pairing_run(const air_spec_t *a, uint64_t *tr, size_t n, size_t row0, const uint32_t *rec, uint32_t *out_words) {
    fq_init();
    if (AIR_PAIRING_LOG_ROWS != 13 || AIR_PAIRING_NREG > 8 || AIR_PAIRING_NFIELDS < 10) return -20;
    const int cpl = a ? a->cells_per_limb : 1;
    const int32_t *lay = (a && cpl == 2) ? AIR_PAIRING_LAYOUT_U8 : AIR_PAIRING_LAYOUT_U16;
    if (tr && (!a || a->kind != 6 || a->log_rows != 13 || lay[L_RES] != a->checked_base)) return -20;
    const fq px = fq_from_u32(rec), py = fq_from_u32(rec + 8), zero = fq_zero();
    const fq2 qx = read2(rec + 16), qy = read2(rec + 32);
    const t6 g1c = gconst(2);       /* the twist's Frobenius constants = the w^2 and w^3 entries of the p-power constant vector */
    const fq2 frob_x = g1c.c[2], frob_y = g1c.c[3];
    fq regs[8][12], cacc[12], S[5][2], inv12[12], prod12[12];
    for (int k = 0; k < 8; k++) for (int t = 0; t < 12; t++) regs[k][t] = zero;
    for (int t = 0; t < 12; t++) { cacc[t] = zero; inv12[t] = zero; prod12[t] = zero; }
    for (int s = 0; s < 5; s++) S[s][0] = S[s][1] = zero;
    fq2 tx = f2z(), ty = f2z(), q1x = f2z(), q1y = f2z(), q2x = f2z(), q2y = f2z(), lamw = f2z();
    const fq2 pxe = {px, zero};
    for (int r = 0; r < AIR_PAIRING_ROWS; r++) {
        const int8_t *d = AIR_PAIRING_ROWPROG[r];
        const int typ = d[F_TYP], t = d[F_T], ra = d[F_RA], rb = d[F_RB], bsel = d[F_BSEL], gc = d[F_GC], ld = d[F_LD], sk = d[F_SK], chm = d[F_CHM],
                  end = d[F_END];
        fq A[12], B[12], res = zero;
        for (int i = 0; i < 12; i++) { A[i] = ra >= 0 ? regs[ra][i] : zero; B[i] = zero; }
        if (rb >= 0) for (int i = 0; i < 12; i++) B[i] = regs[rb][i];
        else if (bsel == 1) { B[0] = py; B[2] = S[3][0]; B[3] = S[3][1]; B[6] = S[4][0]; B[7] = S[4][1]; }
        else if (bsel == 2) for (int i = 0; i < 12; i++) B[i] = cacc[i];
        fq2 qsx = f2z(), qsy = f2z();
        if (sk == 2) { qsx = chm == 0 ? qx : chm == 1 ? q1x : q2x; qsy = chm == 0 ? qy : chm == 1 ? q1y : q2y; }
        const fq u = limbs_to_fq(AIR_PAIRING_GC[gc]), v = limbs_to_fq(AIR_PAIRING_GC[gc] + 16);
        const fq2 lam = {S[0][0], S[0][1]}, x3 = {S[1][0], S[1][1]};
        if (typ == T_FMUL || typ == T_FINVC) {
            if (t == 0) { t6 a6 = t6_from12(A), b6 = t6_from12(B), c6 = t6_mul(&a6, &b6); t6_to12(&c6, prod12); }
            if (typ == T_FMUL) res = prod12[t];
            else if (!fq_eq(prod12[t], u)) return -1;                 /* A * CACC = 1, component t */
        } else if (typ == T_FFROB) {
            res = fq_add(fq_mul(A[2 * (t / 2)], u), fq_mul(A[2 * (t / 2) + 1], v));
        } else if (typ == T_FINVW) {
            if (t == 0) { t6 a6 = t6_from12(A), i6; if (t6_inv(&a6, &i6)) return -1; t6_to12(&i6, inv12); }
            res = inv12[t];
        } else if (typ == T_FCOPY) {
            res = B[t];
        } else if (typ == T_GW) {
            if (t == 0) {
                fq2 num, den;
                if (sk == 1) { den = fq2_add(ty, ty); num = f2_small(3, fq2_mul(tx, tx)); }
                else { den = fq2_sub(qsx, tx); num = fq2_sub(qsy, ty); }
                if (fq2_is_zero(den)) return -1;
                lamw = fq2_mul(num, fq2_inv(den));
            }
            res = t ? lamw.c1 : lamw.c0;
        } else if (typ == T_GSL) {
            fq2 lhs = sk == 1 ? fq2_sub(f2_small(2, fq2_mul(lam, ty)), f2_small(3, fq2_mul(tx, tx)))
                              : fq2_sub(fq2_mul(lam, fq2_sub(qsx, tx)), fq2_sub(qsy, ty));
            if (!fq_is_zero(t ? lhs.c1 : lhs.c0)) return -1;
        } else if (typ == T_GX3) {
            fq2 w = fq2_sub(fq2_sub(fq2_mul(lam, lam), tx), sk == 1 ? tx : qsx);
            res = t ? w.c1 : w.c0;
        } else if (typ == T_GY3) {
            fq2 w = fq2_sub(fq2_mul(lam, fq2_sub(tx, x3)), ty);
            res = t ? w.c1 : w.c0;
        } else if (typ == T_GL1) {
            fq2 w = f2_neg(fq2_mul(lam, pxe));
            res = t ? w.c1 : w.c0;
        } else if (typ == T_GL3) {
            fq2 w = fq2_sub(fq2_mul(lam, tx), ty);
            res = t ? w.c1 : w.c0;
        } else if (typ == T_GFQ) {
            const int slot = t >> 1;
            const fq2 s0 = {S[0][0], S[0][1]}, s1 = {S[1][0], S[1][1]};
            fq2 w = slot == 0 ? fq2_mul(f2_conj(qx), frob_x) : slot == 1 ? fq2_mul(f2_conj(qy), frob_y)
                    : slot == 2 ? fq2_mul(f2_conj(s0), frob_x) : f2_neg(fq2_mul(f2_conj(s1), frob_y));
            res = (t & 1) ? w.c1 : w.c0;
        }
        if (tr) {
            const size_t row = row0 + (size_t)r;
            put_fq(tr, n, lay[L_PX], row, px, 0); put_fq(tr, n, lay[L_PY], row, py, 0);
            const fq2 pts[12] = {qx, qy, q1x, q1y, q2x, q2y, tx, ty, qsx, qsy, frob_x, frob_y};
            for (int i = 0; i < 12; i++) { put_fq(tr, n, lay[L_QX + i], row, pts[i].c0, 0); put_fq(tr, n, lay[L_QX + i] + 16, row, pts[i].c1, 0); }
            for (int s = 0; s < 5; s++) { put_fq(tr, n, lay[L_SR] + 32 * s, row, S[s][0], 0); put_fq(tr, n, lay[L_SR] + 32 * s + 16, row, S[s][1], 0); }
            put_fq(tr, n, lay[L_GC], row, u, 0); put_fq(tr, n, lay[L_GC] + 16, row, v, 0);
            for (int i = 0; i < 12; i++) {
                put_fq(tr, n, lay[L_A] + 16 * i, row, A[i], 0); put_fq(tr, n, lay[L_B] + 16 * i, row, B[i], 0);
                put_fq(tr, n, lay[L_CACC] + 16 * i, row, cacc[i], 0);
                for (int k = 0; k < AIR_PAIRING_NREG; k++) put_fq(tr, n, lay[L_REG] + 192 * k + 16 * i, row, regs[k][i], 0);
            }
            put_fq(tr, n, lay[L_RES], row, res, cpl);
        }
        /* end of the row: loads */
        if (typ == T_FMUL || typ == T_FFROB || typ == T_FINVW || typ == T_FCOPY) cacc[t] = res;
        else if (typ == T_GW) S[0][t] = res;
        else if (typ == T_GX3) S[1][t] = res;
        else if (typ == T_GY3) S[2][t] = res;
        else if (typ == T_GL1) S[3][t] = res;
        else if (typ == T_GL3) S[4][t] = res;
        else if (typ == T_GFQ) S[t >> 1][t & 1] = res;
        if (typ == T_FCOMMIT && ld >= 0) for (int i = 0; i < 12; i++) regs[ld][i] = cacc[i];
        if (end == 1) { tx.c0 = S[1][0]; tx.c1 = S[1][1]; ty.c0 = S[2][0]; ty.c1 = S[2][1]; }
        else if (end == 2) {
            tx = qx; ty = qy;
            q1x.c0 = S[0][0]; q1x.c1 = S[0][1]; q1y.c0 = S[1][0]; q1y.c1 = S[1][1];
            q2x.c0 = S[2][0]; q2x.c1 = S[2][1]; q2y.c0 = S[3][0]; q2y.c1 = S[3][1];
        }
    }
    const fq *res12 = regs[AIR_PAIRING_RESULT_REG];
    fq nine = fq_from_u64(9);
    for (int i = 0; i < 6; i++) {           /* MyFq12 coefficients: c_i = a_i - 9 b_i, c_{i+6} = b_i */
        fq_to_u32(fq_sub(res12[2 * i], fq_mul(nine, res12[2 * i + 1])), out_words + 8 * i);
        fq_to_u32(res12[2 * i + 1], out_words + 8 * (i + 6));
    }
    return 0;
}

/* the value alone: e(P, Q) of a record's first 48 words (points checked: on the curves, Q of order r) */
int orc_pairing(const uint32_t *pq, uint32_t *out96) {
    if (!orc_pairing_record_ok(pq)) return -2;
    return orc_pairing_run(NULL, NULL, 0, 0, pq, out96);
}
