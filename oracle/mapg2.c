/*
 * oracle/mapg2.c -- CPU restatement of the map Fp2 -> E'(Fp2) in front of SIPP's BLS example.
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.  The reference calls
 *   plonky2_bn254::curves::map_to_g2::map_to_g2_without_cofactor_mul   (src/bin/bls_aggregation.rs:21, :102)
 *   starky_bn254::curves::g2::batch_map_to_g2::batch_map_to_g2_circuit (src/bin/bls_aggregation.rs:31, :65)
 * and neither crate is under /root/reference.  Restated here is the published algorithm they are recalled to follow:
 * the Shallue - van de Woestijne map of RFC 9380 (section 6.6.1, straight-line form of appendix F.1) for
 * y^2 = x^3 + 3/(9+u) with Z = 1, sgn0 of section 4.1 (m = 2), is_square(0) = true.  oracle/py/map_to_g2.py is the
 * second reading (tests/test_oracle_mapg2.py compares the two value for value).
 */
#include "mapg2.h"
#include <string.h>

static const uint64_t PM3D4[4] = {0x4f082305b61f3f51ULL, 0x65e05aa45a1c72a3ULL, 0x6e14116da0605617ULL, 0x0c19139cb84c680aULL}; /* (p - 3) / 4 */

static fq fq_pow4(fq a, const uint64_t e[4]) {
    fq r = fq_one();
    for (int i = 255; i >= 0; i--) {
        r = fq_mul(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = fq_mul(r, a);
    }
    return r;
}

/* p = 3 mod 4: a^((p+1)/4) = a * a^((p-3)/4) is a root of a iff a is a square */
static int fq_sqrt(fq a, fq *r) {
    fq t = fq_mul(fq_pow4(a, PM3D4), a);
    *r = t;
    return fq_eq(fq_mul(t, t), a);
}

int fq2_sgn0(fq2 a) {
    int s0 = (int)(a.c0.l[0] & 1), z0 = fq_is_zero(a.c0);
    return s0 | (z0 & (int)(a.c1.l[0] & 1));
}

static fq2 fq2_neg(fq2 a) { fq2 r = {fq_neg(a.c0), fq_neg(a.c1)}; return r; }
static int fq2_eq(fq2 a, fq2 b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }

/* THE root with sgn0 = 0, by the norm: for a = a0 + a1 u with root x0 + x1 u:  x0^2 = (a0 +- sqrt(a0^2 + a1^2)) / 2,  x1 = a1 / (2 x0) */
int fq2_sqrt_even(fq2 a, fq2 *out) {
    fq2 r;
    if (fq_is_zero(a.c1)) {
        fq s;
        if (fq_sqrt(a.c0, &s)) { r.c0 = s; r.c1 = fq_zero(); }
        else if (fq_sqrt(fq_neg(a.c0), &s)) { r.c0 = fq_zero(); r.c1 = s; }     /* (s u)^2 = -s^2 */
        else return 0;
    } else {
        fq n, half = fq_inv(fq_from_u64(2));
        if (!fq_sqrt(fq_add(fq_mul(a.c0, a.c0), fq_mul(a.c1, a.c1)), &n)) return 0;
        fq x0, t = fq_mul(fq_add(a.c0, n), half);
        if (!fq_sqrt(t, &x0)) {
            t = fq_mul(fq_sub(a.c0, n), half);
            if (!fq_sqrt(t, &x0)) return 0;
        }
        r.c0 = x0;
        r.c1 = fq_mul(a.c1, fq_inv(fq_add(x0, x0)));
    }
    if (!fq2_eq(fq2_mul(r, r), a)) return 0;
    *out = fq2_sgn0(r) ? fq2_neg(r) : r;
    return 1;
}

static fq2 f2(uint64_t a, uint64_t b) { fq2 r = {fq_from_u64(a), fq_from_u64(b)}; return r; }

static fq2 g_of(fq2 x, fq2 b) { return fq2_add(fq2_mul(fq2_mul(x, x), x), b); }

void orc_mapg2_constants(orc_mapg2_consts *k) {
    fq_init();
    k->one = f2(1, 0);
    k->xi = f2(9, 1);
    k->b = fq2_mul(f2(3, 0), fq2_inv(k->xi));
    fq2 z = k->one;
    k->c1 = g_of(z, k->b);
    k->c2.c0 = fq_neg(fq_inv(fq_from_u64(2))); k->c2.c1 = fq_zero();
    fq2 three_z2 = f2(3, 0);
    fq2_sqrt_even(fq2_neg(fq2_mul(k->c1, three_z2)), &k->c3);
    k->c4 = fq2_mul(fq2_neg(fq2_mul(f2(4, 0), k->c1)), fq2_inv(three_z2));
}

int orc_mapg2_witness(fq2 u, orc_mapg2_wit *w) {
    orc_mapg2_consts k;
    orc_mapg2_constants(&k);
    memset(w, 0, sizeof *w);
    w->v[MG_T1] = fq2_mul(u, u);
    w->v[MG_TV1] = fq2_mul(k.c1, w->v[MG_T1]);
    fq2 tv2 = fq2_add(k.one, w->v[MG_TV1]), tv1 = fq2_sub(k.one, w->v[MG_TV1]);
    w->v[MG_W] = fq2_mul(tv1, tv2);
    w->z = fq2_is_zero(w->v[MG_W]);                       /* u^2 g(Z) = +-1: inv0(0) = 0 */
    w->v[MG_TV3] = w->z ? f2(0, 0) : fq2_inv(w->v[MG_W]);
    w->v[MG_A4] = fq2_mul(u, tv1);
    w->v[MG_B4] = fq2_mul(w->v[MG_A4], w->v[MG_TV3]);
    fq2 tv4 = fq2_mul(w->v[MG_B4], k.c3);
    w->v[MG_X1] = fq2_sub(k.c2, tv4);
    w->v[MG_X2] = fq2_add(k.c2, tv4);
    w->v[MG_S1] = fq2_mul(w->v[MG_X1], w->v[MG_X1]);
    w->v[MG_GX1] = fq2_add(fq2_mul(w->v[MG_S1], w->v[MG_X1]), k.b);
    w->v[MG_S2] = fq2_mul(w->v[MG_X2], w->v[MG_X2]);
    w->v[MG_GX2] = fq2_add(fq2_mul(w->v[MG_S2], w->v[MG_X2]), k.b);
    w->v[MG_D] = fq2_mul(tv2, tv2);
    w->v[MG_E] = fq2_mul(w->v[MG_D], w->v[MG_TV3]);
    w->v[MG_F] = fq2_mul(w->v[MG_E], w->v[MG_E]);
    w->v[MG_X3] = fq2_add(fq2_mul(w->v[MG_F], k.c4), k.one);
    w->v[MG_S3] = fq2_mul(w->v[MG_X3], w->v[MG_X3]);
    w->v[MG_GX3] = fq2_add(fq2_mul(w->v[MG_S3], w->v[MG_X3]), k.b);
    fq2 r1, r2, zero = f2(0, 0);
    w->e1 = fq2_sqrt_even(w->v[MG_GX1], &r1);
    w->e2 = !w->e1 && fq2_sqrt_even(w->v[MG_GX2], &r2);
    w->m1 = w->e1 ? zero : w->v[MG_GX1];
    w->m2 = (w->e1 || w->e2) ? zero : w->v[MG_GX2];
    if (!fq2_sqrt_even(fq2_mul(k.xi, w->m1), &w->v[MG_N1])) return -2;
    if (!fq2_sqrt_even(fq2_mul(k.xi, w->m2), &w->v[MG_N2])) return -2;
    w->xs = w->e1 ? w->v[MG_X1] : w->e2 ? w->v[MG_X2] : w->v[MG_X3];
    w->gxs = w->e1 ? w->v[MG_GX1] : w->e2 ? w->v[MG_GX2] : w->v[MG_GX3];
    fq2 y;
    if (!fq2_sqrt_even(w->gxs, &y)) return -2;
    if (fq2_sgn0(u) != fq2_sgn0(y)) y = fq2_neg(y);
    w->v[MG_Y] = y;
    return 0;
}

static fq2 read2(const uint32_t *w) { fq2 r = {fq_from_u32(w), fq_from_u32(w + 8)}; return r; }
static void write2(fq2 a, uint32_t *w) { fq_to_u32(a.c0, w); fq_to_u32(a.c1, w + 8); }

/* u (16 u32) -> x, y (16 u32 each) */
int orc_map_to_g2(const uint32_t *u, uint32_t *xy) {
    fq_init();
    orc_mapg2_wit w;
    int rc = orc_mapg2_witness(read2(u), &w);
    if (rc) return rc;
    write2(w.xs, xy);
    write2(w.v[MG_Y], xy + 16);
    return 0;
}

/* the public side of a record (u, x, y): the sign rule of the map (canonicity is orc_pis_canonical's) */
int orc_mapg2_record_sign_ok(const uint32_t *rec) {
    fq_init();
    return fq2_sgn0(read2(rec)) == fq2_sgn0(read2(rec + 32));
}
