/*
 * oracle/air.c -- CPU restatement of the three SIPP AIRs: trace generation (native BN254 arithmetic +
 * generic gadget witnesses from the AIR program) and constraint evaluation (air_eval.inc).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.  The reference reaches this code's upstream counterpart
 * (starky-bn254 @ 2d46f9e `generate_trace` / `eval_packed_generic`, not vendored) only through
 * src/verifier_circuit.rs:133-135.  Semantics follow src/verifier_circuit.rs:92-124:
 *   G1/G2: out = offset + [exp_val] x ;  Fq12: out = offset * x^exp_val ; exp_val = 8 x u32 LE limbs.
 * The AIR layout itself is this repository's specification (tools/air_gen.py).
 */
#include "air.h"
#include "mapg2.h"
#include "pairing.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* API kinds: 0 g1, 1 g2, 2 fq12, 3 mapg2, 4 / 5 = the hardened g1 / g2 AIRs (same records, same rows; air->kind stays 0 / 1),
 * 6 = the final pairing (pairing.c) */
const air_spec_t *orc_air_get(int kind, unsigned log_n) {
    int mode_u16 = log_n >= 16, hard = kind == 4 || kind == 5;
    if (kind < 0 || kind > 6) return NULL;
    int base = hard ? kind - 4 : kind;
    for (size_t i = 0; i < sizeof(AIR_AIRS) / sizeof(AIR_AIRS[0]); i++)
        if (AIR_AIRS[i].kind == base && AIR_AIRS[i].hardened == hard && (AIR_AIRS[i].table_bits == 16) == mode_u16) return &AIR_AIRS[i];
    return NULL;
}
int orc_air_api_kind(const air_spec_t *a) { return a->kind + 4 * a->hardened; }

int orc_air_width(const air_spec_t *a) { return a->n_main + 2 * a->n_checked; }

/* ---- column name lookup is positional: the generator allocates in a fixed order ---- */
typedef struct {
    int ext;           /* curve: 1 or 2 */
    int Rx, Ry, Px, Py, bit, e, lam, X3, Y3;
    int acc, pw, C;    /* fq12 */
    int gad0;          /* first gadget's unchecked sign column / checked q column are found from the program */
} layout_t;

static layout_t layout_of(const air_spec_t *a) {
    layout_t L;
    memset(&L, 0, sizeof L);
    int cpl = a->cells_per_limb;
    if (a->kind == 2) {
        L.acc = 1; L.pw = 1 + 192; L.bit = 1 + 384; L.e = L.bit + 1;
        L.C = a->checked_base;
    } else {
        int ext = a->kind == 0 ? 1 : 2, nc = 16 * ext;
        L.ext = ext;
        L.Rx = 1; L.Ry = 1 + nc; L.Px = 1 + 2 * nc; L.Py = 1 + 3 * nc; L.bit = 1 + 4 * nc; L.e = L.bit + 1;
        L.lam = a->checked_base; L.X3 = L.lam + nc * cpl; L.Y3 = L.X3 + nc * cpl;
    }
    return L;
}

static inline void put(uint64_t *tr, size_t n, int col, size_t row, uint64_t v) { tr[(size_t)col * n + row] = v; }

static void put_fq_u16(uint64_t *tr, size_t n, int col, size_t row, fq v) {
    uint16_t l[16];
    fq_to_limbs16(v, l);
    for (int i = 0; i < 16; i++) put(tr, n, col + i, row, l[i]);
}

static void put_fq_checked(uint64_t *tr, size_t n, int col, size_t row, fq v, int cpl) {
    uint16_t l[16];
    fq_to_limbs16(v, l);
    for (int i = 0; i < 16; i++) {
        if (cpl == 1) put(tr, n, col + i, row, l[i]);
        else { put(tr, n, col + 2 * i, row, l[i] & 0xff); put(tr, n, col + 2 * i + 1, row, l[i] >> 8); }
    }
}

/* ---- primary witness: the native chain ---- */
static int fill_exponent(uint64_t *tr, size_t n, const layout_t *L, size_t row0, const uint32_t exp[8], int *bits_out) {
    uint32_t e[8];
    memcpy(e, exp, sizeof e);
    for (int r = 0; r < 512; r++) {
        size_t row = row0 + r;
        int is_add = (r & 1) == 0;
        int bit = is_add ? (int)(e[0] & 1) : 0;
        put(tr, n, L->bit, row, bit);
        for (int i = 0; i < 8; i++) put(tr, n, L->e + i, row, e[i]);
        if (is_add) {
            bits_out[r >> 1] = bit;
            e[0] = (e[0] - bit) >> 1;
        }
        if ((r & 63) == 63) { /* limb end: rotate */
            for (int i = 0; i < 7; i++) e[i] = e[i + 1];
            e[7] = 0;
        }
    }
    return 0;
}

typedef struct { fq2 x, y; } pt2; /* Fq points use c1 = 0 */

static fq2 mk2(fq a) { fq2 r = {a, fq_zero()}; return r; }

static fq2 f2_mul(fq2 a, fq2 b, int ext) { return ext == 2 ? fq2_mul(a, b) : mk2(fq_mul(a.c0, b.c0)); }
static int f2_inv(fq2 a, int ext, fq2 *out) {
    if (ext == 2) { if (fq2_is_zero(a)) return -1; *out = fq2_inv(a); return 0; }
    if (fq_is_zero(a.c0)) return -1;
    *out = mk2(fq_inv(a.c0));
    return 0;
}

static void put_f2_u16(uint64_t *tr, size_t n, int col, size_t row, fq2 v, int ext) {
    put_fq_u16(tr, n, col, row, v.c0);
    if (ext == 2) put_fq_u16(tr, n, col + 16, row, v.c1);
}
static void put_f2_chk(uint64_t *tr, size_t n, int col, size_t row, fq2 v, int ext, int cpl) {
    put_fq_checked(tr, n, col, row, v.c0, cpl);
    if (ext == 2) put_fq_checked(tr, n, col + 16 * cpl, row, v.c1, cpl);
}

static fq2 read_f2(const uint32_t *w, int ext) {
    fq2 r;
    r.c0 = fq_from_u32(w);
    r.c1 = ext == 2 ? fq_from_u32(w + 8) : fq_zero();
    return r;
}

/* y^2 = x^3 + 3 on E(Fp), (9 + u)(y^2 - x^3) = 3 on the twist E'(Fp2): y^2 = x^3 + 3 / (9 + u) */
static int on_curve(pt2 p, int ext) {
    fq2 d = fq2_sub(f2_mul(p.y, p.y, ext), f2_mul(f2_mul(p.x, p.x, ext), p.x, ext));
    fq three = fq_from_u64(3);
    if (ext == 1) return fq_is_zero(fq_sub(d.c0, three));
    fq nine = fq_from_u64(9);
    fq re = fq_sub(fq_mul(nine, d.c0), d.c1), im = fq_add(d.c0, fq_mul(nine, d.c1));
    return fq_is_zero(fq_sub(re, three)) && fq_is_zero(im);
}

/* TEST HOOK (tests/test_oracle_soundness.py): what a prover that skips its own checks could commit to.  bit 0: records whose
 * points are not on the curve are filled anyway (the chord / tangent formulas are satisfiable for any two points); bit 1: the
 * output words of a record are taken from the chain instead of being compared with it.  Never set by product or bench code. */
static int g_forge;
void orc_test_forge(int flags) { g_forge = flags; }

/* x and offset of every G1 / G2 record on E(Fp) / E'(Fp2) (the verifier's side of the refusal in fill_curve_io) */
int orc_records_on_curve(int kind, const uint32_t *pis, size_t num_io) {
    kind = orc_record_kind(kind);   /* hardened G1 / G2: the same records */
    if (kind == 6) { /* (P, Q, Z): P on E(Fp), Q on E'(Fp2) AND of order r -- the chord rows of the pairing AIR are sound for such Q only */
        int ok = 1;
#pragma omp parallel for schedule(dynamic) reduction(&& : ok)
        for (size_t io = 0; io < num_io; io++) ok = ok && orc_pairing_record_ok(pis + io * 144);
        return ok;
    }
    if (kind == 3) { /* (u, x, y): the point on E'(Fp2) and the map's sign rule sgn0(y) = sgn0(u) (public checks, not constraints) */
        fq_init();
        for (size_t io = 0; io < num_io; io++) {
            const uint32_t *rec = pis + io * 48;
            pt2 Q = {read_f2(rec + 16, 2), read_f2(rec + 32, 2)};
            if (!on_curve(Q, 2) || !orc_mapg2_record_sign_ok(rec)) return 0;
        }
        return 1;
    }
    if (kind != 0 && kind != 1) return 1;
    fq_init();
    const int ext = kind + 1, w = 8 * ext, ppi = 8 * (6 * ext + 1);
    for (size_t io = 0; io < num_io; io++) {
        const uint32_t *rec = pis + io * ppi;
        pt2 P = {read_f2(rec, ext), read_f2(rec + w, ext)};
        pt2 R = {read_f2(rec + 2 * w, ext), read_f2(rec + 3 * w, ext)};
        if (!on_curve(P, ext) || !on_curve(R, ext)) return 0;
    }
    return 1;
}

static int fill_curve_io(const air_spec_t *a, uint64_t *tr, size_t n, size_t io, const uint32_t *rec, uint32_t *out_words) {
    layout_t L = layout_of(a);
    int ext = L.ext, cpl = a->cells_per_limb, w = 8 * ext;
    pt2 P = {read_f2(rec, ext), read_f2(rec + w, ext)};
    pt2 R = {read_f2(rec + 2 * w, ext), read_f2(rec + 3 * w, ext)};
    /* both points on the curve: the chord / tangent rules are a group law only there (a generator that sums in another order
     * than this chain -- the GPU's scan -- agrees with it only there), and the statement is about group elements */
    if (!(g_forge & 1) && (!on_curve(P, ext) || !on_curve(R, ext))) return -1;
    const uint32_t *exp = rec + 4 * w;
    int bits[256];
    size_t row0 = io * 512;
    fill_exponent(tr, n, &L, row0, exp, bits);
    fq2 three = mk2(fq_from_u64(3)), two = mk2(fq_from_u64(2));
    int pending = 0, inf = 0;          /* hardened: inf = the accumulator is the identity (R keeps its last finite value) */
    const int32_t *hl = !a->hardened ? NULL : a->kind == 0 ? (cpl == 1 ? AIR_HARD_LAYOUT_G1H_U16 : AIR_HARD_LAYOUT_G1H_U8)
                                                          : (cpl == 1 ? AIR_HARD_LAYOUT_G2H_U16 : AIR_HARD_LAYOUT_G2H_U8);
    for (int r = 0; r < 512; r++) {
        size_t row = row0 + r;
        int is_add = (r & 1) == 0;
        if (hl) put(tr, n, hl[7], row, (uint64_t)inf);
        put_f2_u16(tr, n, L.Rx, row, R.x, ext); put_f2_u16(tr, n, L.Ry, row, R.y, ext);
        put_f2_u16(tr, n, L.Px, row, P.x, ext); put_f2_u16(tr, n, L.Py, row, P.y, ext);
        fq2 lam, num, den, deninv, xa, ya, xb;
        if (is_add) {
            num = fq2_sub(P.y, R.y); den = fq2_sub(P.x, R.x);
            xa = R.x; ya = R.y; xb = P.x;
        } else {
            num = f2_mul(three, f2_mul(P.x, P.x, ext), ext); den = f2_mul(two, P.y, ext);
            xa = P.x; ya = P.y; xb = P.x;
        }
        int same = 0, opposite = 0;
        if (f2_inv(den, ext, &deninv)) {
            /* R = +-P or 2-torsion: not provable by the plain AIR.  The hardened AIR proves both: R = P (the sum is the double the next
             * row computes) and R = -P (the sum is the identity: the bit inf); slope cells 0, result unused.  TEST HOOK bit 2: what a
             * cheating prover does with the plain AIR where R = P -- the chord rule 0 lam = 0 holds for every lam, so it picks one (5)
             * and walks on from a point of its choosing */
            if (is_add && fq2_is_zero(num) && (a->hardened || (g_forge & 4))) {
                same = a->hardened && !(g_forge & 4);
                lam = same ? mk2(fq_zero()) : mk2(fq_from_u64(5));
            } else if (is_add && a->hardened && !(g_forge & 4)) {
                opposite = 1;
                lam = mk2(fq_zero());
            } else
                return -1;
        } else
            lam = f2_mul(num, deninv, ext);
        fq2 x3 = fq2_sub(fq2_sub(f2_mul(lam, lam, ext), xa), xb);
        fq2 y3 = fq2_sub(f2_mul(lam, fq2_sub(xa, x3), ext), ya);
        put_f2_chk(tr, n, L.lam, row, lam, ext, cpl);
        put_f2_chk(tr, n, L.X3, row, x3, ext, cpl);
        put_f2_chk(tr, n, L.Y3, row, y3, ext, cpl);
        if (is_add) {
            if (inf) {                                    /* the identity plus P: a copy of P (R's cells were stale) */
                if (bits[r >> 1]) { R = P; inf = 0; }
            } else if (same) pending = bits[r >> 1];      /* hardened, R = P: the accumulator takes the NEXT row's double */
            else if (opposite) inf = bits[r >> 1];        /* hardened, R = -P: the sum is the identity */
            else if (bits[r >> 1]) { R.x = x3; R.y = y3; }
        } else {
            if (pending) {
                if (r == 511) return -1;                 /* no row left to hand the double over */
                R.x = x3; R.y = y3; pending = 0;
            }
            if (r != 511) { P.x = x3; P.y = y3; }
        }
    }
    if (inf) return -1;                                   /* the output would be the identity: no affine record for it */
    /* output words */
    fq_to_u32(R.x.c0, out_words);
    if (ext == 2) { fq_to_u32(R.x.c1, out_words + 8); fq_to_u32(R.y.c0, out_words + 16); fq_to_u32(R.y.c1, out_words + 24); }
    else fq_to_u32(R.y.c0, out_words + 8);
    return 0;
}

static fq fq12_tower(const fq12 *x, int t) {
    int i = t >> 1;
    if (t & 1) return x->c[i + 6];
    return fq_add(x->c[i], fq_mul(fq_from_u64(9), x->c[i + 6]));
}

static int fill_fq12_io(const air_spec_t *a, uint64_t *tr, size_t n, size_t io, const uint32_t *rec, uint32_t *out_words) {
    layout_t L = layout_of(a);
    int cpl = a->cells_per_limb;
    fq12 pw, acc;
    for (int k = 0; k < 12; k++) { pw.c[k] = fq_from_u32(rec + 8 * k); acc.c[k] = fq_from_u32(rec + 96 + 8 * k); }
    int bits[256];
    size_t row0 = io * 512;
    fill_exponent(tr, n, &L, row0, rec + 192, bits);
    for (int r = 0; r < 512; r++) {
        size_t row = row0 + r;
        int is_add = (r & 1) == 0;
        fq12 c = is_add ? fq12_mul(&acc, &pw) : fq12_mul(&pw, &pw);
        /* cells hold the tower basis: component t = 2i: c_i + 9 c_{i+6}, t = 2i+1: c_{i+6} */
        for (int t = 0; t < 12; t++) {
            put_fq_u16(tr, n, L.acc + 16 * t, row, fq12_tower(&acc, t));
            put_fq_u16(tr, n, L.pw + 16 * t, row, fq12_tower(&pw, t));
            put_fq_checked(tr, n, L.C + 16 * cpl * t, row, fq12_tower(&c, t), cpl);
        }
        if (is_add) { if (bits[r >> 1]) acc = c; }
        else if (r != 511) pw = c;
    }
    for (int k = 0; k < 12; k++) fq_to_u32(acc.c[k], out_words + 8 * k);
    return 0;
}

/* MapToG2 (kind 3): eight rows per record (u, x, y); columns and schedule from the tables tools/air_gen.py::build_map_g2 emits:
 * row t of the block holds the witnesses AIR_MAPG2_SLOT_WIT[t][0..2] in its three result slots and AIR_MAPG2_REG_WIT[t][0..5]
 * in its six registers (zero where the table says -1); u, e1, e2 and the constants on every row; M1, M2, XS, GXS as the
 * (ungated) selection constraints define them from the registers of THAT row. */
static int fill_map_io(const air_spec_t *a, uint64_t *tr, size_t n, size_t io, const uint32_t *rec, uint32_t *out_words) {
    const int cpl = a->cells_per_limb;
    const int32_t *lay = cpl == 1 ? AIR_MAPG2_LAYOUT_U16 : AIR_MAPG2_LAYOUT_U8;
    enum { L_U, L_ONE, L_C1, L_C2, L_C3, L_C4, L_BB, L_E1, L_E2, L_M1, L_M2, L_XS, L_GXS, L_REG, L_RES, L_RX1, L_RX2, L_RX3, L_RG1, L_RG2, L_RG3,
           L_Z, L_ZV, L_TINV };
    if (a->log_rows != 3 || AIR_MAPG2_ROWS != 8 || AIR_MAPG2_NWIT != MG_NWIT || lay[L_RES] != a->checked_base) return -20;
    orc_mapg2_consts k;
    orc_mapg2_constants(&k);
    orc_mapg2_wit w;
    fq2 u = read_f2(rec, 2);
    if (orc_mapg2_witness(u, &w)) return -1;
    const fq2 zero = {fq_zero(), fq_zero()};
    for (int t = 0; t < 8; t++) {
        const size_t row = io * 8 + (size_t)t;
        put_f2_u16(tr, n, lay[L_U], row, u, 2);
        put_f2_u16(tr, n, lay[L_ONE], row, k.one, 2); put_f2_u16(tr, n, lay[L_C1], row, k.c1, 2); put_f2_u16(tr, n, lay[L_C2], row, k.c2, 2);
        put_f2_u16(tr, n, lay[L_C3], row, k.c3, 2); put_f2_u16(tr, n, lay[L_C4], row, k.c4, 2); put_f2_u16(tr, n, lay[L_BB], row, k.b, 2);
        put(tr, n, lay[L_E1], row, (uint64_t)w.e1); put(tr, n, lay[L_E2], row, (uint64_t)w.e2);
        const uint64_t zf = (t == lay[L_TINV] && w.z) ? 1 : 0;        /* inv0's flag, on the row that inverts */
        put(tr, n, lay[L_Z], row, zf);
        for (int j = 0; j < 32; j++) put(tr, n, lay[L_ZV] + j, row, j == 0 ? zf : 0);
        fq2 reg[6];
        for (int r = 0; r < 6; r++) {
            int wi = AIR_MAPG2_REG_WIT[t][r];
            reg[r] = wi < 0 ? zero : w.v[wi];
            put_f2_u16(tr, n, lay[L_REG] + 32 * r, row, reg[r], 2);
        }
        for (int sl = 0; sl < 3; sl++) {
            int wi = AIR_MAPG2_SLOT_WIT[t][sl];
            put_f2_chk(tr, n, lay[L_RES] + 32 * cpl * sl, row, wi < 0 ? zero : w.v[wi], 2, cpl);
        }
        const fq2 gx1 = reg[lay[L_RG1]], gx2 = reg[lay[L_RG2]], gx3 = reg[lay[L_RG3]];
        put_f2_u16(tr, n, lay[L_M1], row, w.e1 ? zero : gx1, 2);
        put_f2_u16(tr, n, lay[L_M2], row, (w.e1 || w.e2) ? zero : gx2, 2);
        put_f2_u16(tr, n, lay[L_XS], row, w.e1 ? reg[lay[L_RX1]] : w.e2 ? reg[lay[L_RX2]] : reg[lay[L_RX3]], 2);
        put_f2_u16(tr, n, lay[L_GXS], row, w.e1 ? gx1 : w.e2 ? gx2 : gx3, 2);
    }
    fq_to_u32(w.xs.c0, out_words); fq_to_u32(w.xs.c1, out_words + 8);
    fq_to_u32(w.v[MG_Y].c0, out_words + 16); fq_to_u32(w.v[MG_Y].c1, out_words + 24);
    return 0;
}

/* hardened curve AIRs (kinds 4 / 5): T3 = p - 1 - x3 with its borrow bits, and on add rows with bit = 1 the witness that R.x and P.x
 * differ in a limb: nz_j = 1 / (Px_j - Rx_j) at the first such limb (a Goldilocks inverse of a 17-bit difference), 0 elsewhere */
static int fill_harden_row(const air_spec_t *a, uint64_t *tr, size_t n, size_t row) {
    const int ext = a->kind == 0 ? 1 : 2, nc = 16 * ext, cpl = a->cells_per_limb;
    const int32_t *lay = a->kind == 0 ? (cpl == 1 ? AIR_HARD_LAYOUT_G1H_U16 : AIR_HARD_LAYOUT_G1H_U8)
                                      : (cpl == 1 ? AIR_HARD_LAYOUT_G2H_U16 : AIR_HARD_LAYOUT_G2H_U8);
    layout_t L = layout_of(a);
    for (int c = 0; c < ext; c++) {
        int64_t borrow = 0;
        for (int i = 0; i < 16; i++) {
            int64_t x = cpl == 1 ? (int64_t)tr[(size_t)(L.X3 + 16 * c + i) * n + row]
                                 : (int64_t)(tr[(size_t)(L.X3 + 2 * (16 * c + i)) * n + row] + 256 * tr[(size_t)(L.X3 + 2 * (16 * c + i) + 1) * n + row]);
            int64_t pm1 = (int64_t)AIR_BN_P_LIMBS[i] - (i == 0 ? 1 : 0);      /* p is odd: p - 1 only changes limb 0 */
            int64_t d = pm1 - x - borrow;
            borrow = d < 0;
            uint64_t t = (uint64_t)(d + (borrow << 16));
            if (cpl == 1) put(tr, n, lay[2] + 16 * c + i, row, t);
            else { put(tr, n, lay[2] + 2 * (16 * c + i), row, t & 0xff); put(tr, n, lay[2] + 2 * (16 * c + i) + 1, row, t >> 8); }
            if (i < 15) put(tr, n, lay[1] + 15 * c + i, row, (uint64_t)borrow);
        }
        if (borrow) return -30;                                              /* x3 >= p: cannot happen for the chain's canonical values */
    }
    for (int j = 0; j < nc; j++) put(tr, n, lay[0] + j, row, 0);
    /* flags (see tools/air_gen.py::build_curve): eq / ng on add rows: R = P / R = -P as limb vectors; inf was written by the chain;
     * t1 = bit (1 - inf), u = t1 (1 - eq - ng), v = bit inf, w = t1 ng; eqc on a double row = t1 eq of the add row before it */
    const int is_add = (row & 1) == 0;
    const size_t cr = is_add ? row : row - 1;
    int eq = 1, xeq = 1;
    for (int j = 0; j < nc; j++) {
        xeq = xeq && tr[(size_t)(L.Px + j) * n + cr] == tr[(size_t)(L.Rx + j) * n + cr];
        eq = eq && tr[(size_t)(L.Py + j) * n + cr] == tr[(size_t)(L.Ry + j) * n + cr];
    }
    eq = eq && xeq;
    /* R = -P: the x limbs equal and Ry + Py = p limb by limb (carries cn) */
    int ng = is_add && xeq && !eq;
    uint64_t cnv[30];
    memset(cnv, 0, sizeof cnv);
    for (int c = 0; c < ext && ng; c++) {
        int64_t carry = 0;
        for (int i = 0; i < 16; i++) {
            int64_t sum = (int64_t)tr[(size_t)(L.Ry + 16 * c + i) * n + row] + (int64_t)tr[(size_t)(L.Py + 16 * c + i) * n + row] + carry;
            int64_t d = sum - (int64_t)AIR_BN_P_LIMBS[i];
            if (d != 0 && d != 65536) { ng = 0; break; }
            carry = d == 65536;
            if (i < 15) cnv[15 * c + i] = (uint64_t)carry;
        }
        if (carry) ng = 0;
    }
    if (g_forge & 4) { eq = 0; ng = 0; }                                  /* the forger claims the chord */
    const int bit = tr[(size_t)L.bit * n + row] != 0, infc = tr[(size_t)lay[7] * n + row] != 0;
    const int eq_here = is_add && eq, t1 = bit && !infc, u = t1 && !eq_here && !ng, v = bit && infc, w = t1 && ng;
    int eqc = 0;
    if (!is_add) eqc = eq && !(g_forge & 4) && tr[(size_t)L.bit * n + row - 1] != 0 && tr[(size_t)lay[7] * n + row - 1] == 0;
    put(tr, n, lay[3], row, (uint64_t)eq_here); put(tr, n, lay[4], row, (uint64_t)u); put(tr, n, lay[5], row, (uint64_t)eqc);
    put(tr, n, lay[6], row, (uint64_t)ng); put(tr, n, lay[8], row, (uint64_t)t1); put(tr, n, lay[9], row, (uint64_t)v);
    put(tr, n, lay[10], row, (uint64_t)w);
    for (int j = 0; j < 16; j++) put(tr, n, lay[11] + j, row, j == 0 ? (uint64_t)ng : 0);
    for (int j = 0; j < 15 * ext; j++) put(tr, n, lay[12] + j, row, ng ? cnv[j] : 0);
    if (is_add && u) {
        int j;
        for (j = 0; j < nc; j++)
            if (tr[(size_t)(L.Px + j) * n + row] != tr[(size_t)(L.Rx + j) * n + row]) break;
        if (j == nc) return (g_forge & 4) ? 0 : -31;                         /* the x's equal where the chord is used: no witness */
        put(tr, n, lay[0] + j, row, gl_inv(gl_sub(tr[(size_t)(L.Px + j) * n + row], tr[(size_t)(L.Rx + j) * n + row])));
    }
    return 0;
}

/* ---- generic gadget witnesses (integers) ---- */
static size_t ivec(const int64_t *w, const uint64_t *tr, size_t n, size_t row, const int *per,
                   int64_t *out, int *n_out) {
    int nl = (int)w[0], nt = (int)w[1];
    for (int i = 0; i < nl; i++) out[i] = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t *tm = w + 2 + 5 * t;
        int64_t f = tm[0];
        int base = (int)tm[1], stride = (int)tm[2], flag = (int)tm[3], neg = (int)tm[4];
        if (flag >= 0) f *= neg ? 1 - per[flag] : per[flag];
        if (f == 0) continue;
        for (int i = 0; i < nl; i++) out[i] += f * (int64_t)tr[(size_t)(base + i * stride) * n + row];
    }
    *n_out = nl;
    return 2 + 5 * (size_t)nt;
}

static int fill_gadgets_row(const air_spec_t *a, uint64_t *tr, size_t n, size_t row) {
    int per[AIR_N_PERIODIC + 160];    /* closed-form selectors, then the AIR's selector columns (value-periodic, small integers) */
    for (int k = 0; k < AIR_N_PERIODIC; k++) per[k] = (int)(row % (size_t)AIR_PERIODIC[k][0]) == AIR_PERIODIC[k][1];
    if (a->n_vflag > 160) return -21;
    for (int k = 0; k < a->n_vflag; k++) per[AIR_N_PERIODIC + k] = (int)air_vper_value(a, k, (int)(row & (((size_t)1 << a->log_rows) - 1)));
    const int64_t *w = a->prog, *end = a->prog + a->prog_len;
    int cpl = a->cells_per_limb;
    /* p^-1 mod 2^16 */
    uint32_t pinv = 1;
    for (int i = 0; i < 5; i++) pinv = (pinv * (2 - AIR_BN_P_LIMBS[0] * pinv)) & 0xffff;
    while (w < end) {
        if (w[0] != 1) break; /* gadgets come first */
        int sign_col = (int)w[1], cbase = (int)w[2], ncl = (int)w[3], lb = (int)w[4];
        int64_t coffset = w[5];
        int grp = (int)w[6];
        w += 7;
        /* q vector descriptor: remember where its cells live */
        int q_base = (int)w[3]; /* first term's base */
        w += 2 + 5 * (size_t)w[1];
        int64_t e[34], va[17], vb[17];
        memset(e, 0, sizeof e);
        int na, nb;
        int np = (int)*w++;
        for (int p = 0; p < np; p++) {
            int64_t coef = *w++;
            w += ivec(w, tr, n, row, per, va, &na);
            w += ivec(w, tr, n, row, per, vb, &nb);
            for (int i = 0; i < na; i++) {
                if (!va[i]) continue;
                int64_t ai = coef * va[i];
                for (int j = 0; j < nb; j++) e[i + j] += ai * vb[j];
            }
        }
        int nl = (int)*w++;
        for (int p = 0; p < nl; p++) {
            int64_t coef = *w++;
            w += ivec(w, tr, n, row, per, va, &na);
            for (int i = 0; i < na; i++) e[i] += coef * va[i];
        }
        /* E = sum e_k 2^(16k) as a signed big integer in 16-bit limbs */
        int64_t limbs[40];
        int sign = 0;
        for (int pass = 0; pass < 2; pass++) {
            int64_t carry = 0;
            for (int k = 0; k < 40; k++) {
                int64_t t = (k < 31 ? (sign ? -e[k] : e[k]) : 0) + carry;
                limbs[k] = t & 0xffff;
                carry = t >> 16;
            }
            if (carry == 0) break;
            if (carry == -1 && pass == 0) { sign = 1; continue; }
            return -2;
        }
        /* exact division by p: q_i = limbs[i] * p^-1 mod 2^16 */
        uint32_t q[17];
        for (int i = 0; i < 17; i++) {
            q[i] = ((uint32_t)limbs[i] * pinv) & 0xffff;
            int64_t carry = 0;
            for (int j = 0; i + j < 40; j++) {
                int64_t t = limbs[i + j] - (j < 16 ? (int64_t)q[i] * AIR_BN_P_LIMBS[j] : 0) + carry;
                limbs[i + j] = t & 0xffff;
                carry = t >> 16;
                if (j >= 16 && carry == 0) break;
            }
        }
        for (int k = 0; k < 40; k++) if (limbs[k]) return -3; /* E not divisible by p: primary witness wrong */
        put(tr, n, sign_col, row, (uint64_t)sign);
        for (int i = 0; i < 17; i++) {
            if (cpl == 1) put(tr, n, q_base + i, row, q[i]);
            else { put(tr, n, q_base + 2 * i, row, q[i] & 0xff); put(tr, n, q_base + 2 * i + 1, row, q[i] >> 8); }
        }
        /* carries: sum_t 2^(16t) d_{g m + t} - c_{m-1} + 2^(16 g) c_m = 0,  d_k = e_k - sgn (q*p)_k */
        int64_t sgn = sign ? -1 : 1;
        __int128 cprev = 0;
        int nm_ = 32 / grp;
        for (int m = 0; m < nm_; m++) {
            __int128 dm = 0;
            for (int t = grp - 1; t >= 0; t--) {
                int k = grp * m + t;
                int64_t qp = 0;
                for (int i = 0; i < 17; i++) { int j = k - i; if (j >= 0 && j < 16) qp += (int64_t)q[i] * AIR_BN_P_LIMBS[j]; }
                dm = dm * 65536 + ((k < 31 ? e[k] : 0) - sgn * qp);
            }
            dm -= cprev; /* = -2^(16 g) c_m */
            __int128 mask = (((__int128)1) << (16 * grp)) - 1;
            if (dm & mask) return -4;
            __int128 ck = -(dm >> (16 * grp));
            if (m == nm_ - 1) { if (ck != 0) return -5; break; }
            __int128 v = ck + coffset;
            if (v < 0 || (v >> (ncl * lb)) != 0) return -6;
            for (int l = 0; l < ncl; l++) put(tr, n, cbase + m * ncl + l, row, (uint64_t)((v >> (lb * l)) & (((int64_t)1 << lb) - 1)));
            cprev = ck;
        }
    }
    return 0;
}

/* ---- permuted lookup columns (this repository's deterministic fill rule, DESIGN.md "lookups") ----
 * perm_in  = the column sorted ascending
 * perm_tab[i] = perm_in[i] where perm_in[i] is the first occurrence of its value; the remaining positions
 *               receive, in ascending position order, the table entries not matched that way, ascending. */
static void permuted_cols(const uint64_t *col, size_t n, unsigned tbits, uint64_t *pin, uint64_t *ptab) {
    size_t T = (size_t)1 << tbits;
    uint32_t *hist = (uint32_t *)calloc(T, sizeof(uint32_t));
    for (size_t i = 0; i < n; i++) hist[col[i]]++;
    /* unused table entries ascending: values with hist == 0, then the extra copies of T-1 */
    size_t pos = 0;
    size_t uz = 0; /* cursor over unused values */
    for (size_t v = 0; v < T; v++) {
        for (uint32_t k = 0; k < hist[v]; k++) {
            pin[pos] = v;
            if (k == 0) ptab[pos] = v;
            else {
                while (uz < T && hist[uz] != 0) uz++;
                if (uz < T) ptab[pos] = uz++;
                else ptab[pos] = T - 1;
            }
            pos++;
        }
    }
    free(hist);
}

orc_trace *orc_trace_build(int api_kind, const uint32_t *ios, size_t num_io, int *err) {
    fq_init();
    *err = 0;
    if (num_io == 0 || api_kind < 0 || api_kind > 6) { *err = -1; return NULL; }
    const int kind = orc_record_kind(api_kind);   /* the hardened variants fill the same primary cells */
    const unsigned log_rows = orc_kind_log_rows(kind);   /* rows per record: 512 (exponentiations), 8 (MapToG2), 8192 (pairing); = air->log_rows */
    size_t nio = 2; /* at least two IO blocks, at least 1024 rows */
    while (nio < num_io || (nio << log_rows) < 1024) nio <<= 1;
    unsigned log_n = log_rows;
    while (((size_t)1 << (log_n - log_rows)) < nio) log_n++;
    const air_spec_t *a = orc_air_get(api_kind, log_n);
    if (!a || (unsigned)a->log_rows != log_rows) { *err = -1; return NULL; }
    size_t n = (size_t)1 << log_n;
    if (n < ((size_t)1 << a->table_bits)) { *err = -7; return NULL; }
    int W = orc_air_width(a);
    orc_trace *t = (orc_trace *)calloc(1, sizeof *t);
    t->air = a; t->log_n = log_n; t->num_io = nio; t->width = W;
    t->trace = (uint64_t *)calloc((size_t)W * n, sizeof(uint64_t));
    t->pis = (uint32_t *)calloc(nio * a->pi_per_io, sizeof(uint32_t));
    int ppi = a->pi_per_io, out_words = kind == 0 ? 16 : (kind == 2 || kind == 6) ? 96 : 32;
    int rc_all = 0;
#pragma omp parallel for schedule(dynamic)
    for (size_t io = 0; io < nio; io++) {
        const uint32_t *rec = ios + (io < num_io ? io : num_io - 1) * ppi;
        uint32_t *pi = t->pis + io * ppi;
        memcpy(pi, rec, ppi * sizeof(uint32_t));
        uint32_t outw[96];
        int rc = kind == 2 ? fill_fq12_io(a, t->trace, n, io, rec, outw) : kind == 3 ? fill_map_io(a, t->trace, n, io, rec, outw)
                 : kind == 6 ? ((g_forge & 1) || orc_pairing_record_ok(rec) ? orc_pairing_run(a, t->trace, n, io << 13, rec, outw) : -1)
                             : fill_curve_io(a, t->trace, n, io, rec, outw);
        if (rc == 0 && (g_forge & 2)) memcpy(pi + ppi - out_words, outw, out_words * sizeof(uint32_t));
        else if (rc == 0 && memcmp(outw, rec + ppi - out_words, out_words * sizeof(uint32_t)) != 0) rc = -8; /* claimed output wrong */
        if (rc) {
#pragma omp critical
            rc_all = rc;
        }
    }
    if (rc_all) { *err = rc_all; orc_trace_free(t); return NULL; }
    size_t T = (size_t)1 << a->table_bits;
    for (size_t r = 0; r < n; r++) t->trace[r] = r < T ? r : T - 1;
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < n; r++) {
        int rc = a->hardened ? fill_harden_row(a, t->trace, n, r) : 0;
        if (!rc) rc = fill_gadgets_row(a, t->trace, n, r);
        if (rc) {
#pragma omp critical
            rc_all = rc;
        }
    }
    if (rc_all) { *err = rc_all; orc_trace_free(t); return NULL; }
#pragma omp parallel for schedule(dynamic)
    for (int j = 0; j < a->n_checked; j++)
        permuted_cols(t->trace + (size_t)(a->checked_base + j) * n, n, a->table_bits,
                      t->trace + (size_t)(a->n_main + j) * n, t->trace + (size_t)(a->n_main + a->n_checked + j) * n);
    return t;
}

void orc_trace_free(orc_trace *t) {
    if (!t) return;
    free(t->trace); free(t->pis); free(t);
}

/* ---- periodic selectors and public-input polynomials ---- */
/* S_{m,r0}(x) = (K/N) ((x g^-r0)^N - 1) / ((x g^-r0)^K - 1),  K = N/m  (1 on rows r = r0 mod m) */
uint64_t orc_periodic_base(unsigned log_n, int which, uint64_t x) {
    uint64_t N = (uint64_t)1 << log_n, m = (uint64_t)AIR_PERIODIC[which][0], r0 = (uint64_t)AIR_PERIODIC[which][1];
    uint64_t K = N / m, g = gl_root_of_unity(log_n);
    uint64_t y = gl_mul(x, gl_inv(gl_pow(g, r0)));
    uint64_t num = gl_sub(gl_pow(y, N), 1), den = gl_sub(gl_pow(y, K), 1);
    return gl_mul(gl_mul(num, gl_inv(den)), gl_mul(K % GL_P, gl_inv(N % GL_P)));
}
gl2 orc_periodic_ext(unsigned log_n, int which, gl2 x) {
    uint64_t N = (uint64_t)1 << log_n, m = (uint64_t)AIR_PERIODIC[which][0], r0 = (uint64_t)AIR_PERIODIC[which][1];
    uint64_t K = N / m, g = gl_root_of_unity(log_n);
    gl2 y = gl2_scale(x, gl_inv(gl_pow(g, r0)));
    gl2 num = gl2_sub(gl2_pow(y, N), gl2_from(1)), den = gl2_sub(gl2_pow(y, K), gl2_from(1));
    return gl2_scale(gl2_mul(num, gl2_inv(den)), gl_mul(K % GL_P, gl_inv(N % GL_P)));
}

/* value-periodic columns: column k takes the value air_vper_value(a, k, r mod R) on trace row r (R = 2^log_rows rows per record), so
 * it is the polynomial P_k(x^(N/R)) with P_k (degree < R) interpolating the R values over the order-R subgroup */
int orc_air_n_vper(const air_spec_t *a) { return a->n_vflag + a->n_vconst; }
int orc_air_n_per(const air_spec_t *a) { return AIR_N_PERIODIC + orc_air_n_vper(a); }
void orc_vper_coeffs(const air_spec_t *a, int k, uint64_t *coeffs) {
    const size_t R = (size_t)1 << a->log_rows;
    for (size_t r = 0; r < R; r++) coeffs[r] = gl_from_i64(air_vper_value(a, k, (int)r));
    orc_ifft(coeffs, (unsigned)a->log_rows);
}
gl2 orc_vper_ext(const air_spec_t *a, unsigned log_n, const uint64_t *coeffs, gl2 x) {
    const size_t R = (size_t)1 << a->log_rows;
    gl2 y = gl2_pow(x, (uint64_t)1 << (log_n - (unsigned)a->log_rows)), acc = gl2_from(0);
    for (size_t i = R; i-- > 0;) acc = gl2_add(gl2_mul(acc, y), gl2_from(coeffs[i]));
    return acc;
}

/* value of aux column `ai` for IO `io`: the lo/hi half or the whole of a public u32 word */
uint64_t orc_aux_value(const air_spec_t *a, const uint32_t *pis, size_t io, int ai) {
    int word = a->aux[4 * ai], part = a->aux[4 * ai + 1], sub = a->aux[4 * ai + 3];
    const uint32_t *rec = pis + io * a->pi_per_io;
    if (part == 3) { /* tower-basis limb of the MyFq12 value at words [word, word + 96) */
        fq12 x;
        fq_init();
        for (int k = 0; k < 12; k++) x.c[k] = fq_from_u32(rec + word + 8 * k);
        uint16_t l[16];
        fq_to_limbs16(fq12_tower(&x, sub / 16), l);
        return l[sub % 16];
    }
    uint32_t w = rec[word];
    return part == 0 ? (w & 0xffff) : part == 1 ? (w >> 16) : w;
}

/* coefficients (length num_io) of the aux polynomial A with A(g^(rows_per_io io + shift)) = value(io):
 * interpolate over the order-num_io subgroup, then substitute x -> x g^-shift. */
void orc_aux_coeffs(const air_spec_t *a, const uint32_t *pis, size_t num_io, unsigned log_n, int ai, uint64_t *coeffs) {
    unsigned log_io = log_n - (unsigned)a->log_rows;
    for (size_t io = 0; io < num_io; io++) coeffs[io] = orc_aux_value(a, pis, io, ai);
    orc_ifft(coeffs, log_io);
    int shift = a->aux[4 * ai + 2];
    if (shift) {
        uint64_t s = gl_inv(gl_pow(gl_root_of_unity(log_n), (uint64_t)shift)), f = 1;
        for (size_t j = 0; j < num_io; j++) { coeffs[j] = gl_mul(coeffs[j], f); f = gl_mul(f, s); }
    }
}

/* ---- evaluators ---- */
#define FT uint64_t
#define F_ADD gl_add
#define F_SUB gl_sub
#define F_MUL gl_mul
#define F_SCALE(a, s) gl_mul((a), (s) % GL_P)
#define F_FROM_I64(v) gl_from_i64(v)
#define SUFFIX(n) n##_base
#include "air_eval.inc"
#undef FT
#undef F_ADD
#undef F_SUB
#undef F_MUL
#undef F_SCALE
#undef F_FROM_I64
#undef SUFFIX

#define FT gl2
#define F_ADD gl2_add
#define F_SUB gl2_sub
#define F_MUL gl2_mul
#define F_SCALE(a, s) gl2_scale((a), (s) % GL_P)
#define F_FROM_I64(v) gl2_from(gl_from_i64(v))
#define SUFFIX(n) n##_ext
#include "air_eval.inc"

void orc_eval_base(const air_spec_t *air, const uint64_t *local, const uint64_t *next, const uint64_t *aux,
                   const uint64_t *per, const uint64_t *z_local, const uint64_t *z_next,
                   uint64_t lag_first, uint64_t lag_last, uint64_t z_last, const uint64_t alpha[2],
                   const uint64_t beta[2], const uint64_t gamma[2], uint64_t out[2]) {
    evalctx_base c;
    c.local = local; c.next = next; c.aux = aux; c.z_local = z_local; c.z_next = z_next;
    c.per = per;
    c.lag_first = lag_first; c.lag_last = lag_last; c.z_last = z_last;
    c.alpha[0] = alpha[0]; c.alpha[1] = alpha[1]; c.beta[0] = beta[0]; c.beta[1] = beta[1];
    c.gamma[0] = gamma[0]; c.gamma[1] = gamma[1];
    eval_all_base(air, &c);
    out[0] = c.acc[0]; out[1] = c.acc[1];
}

void orc_eval_ext(const air_spec_t *air, const gl2 *local, const gl2 *next, const gl2 *aux,
                  const gl2 *per, const gl2 *z_local, const gl2 *z_next, gl2 lag_first, gl2 lag_last,
                  gl2 z_last, const uint64_t alpha[2], const uint64_t beta[2], const uint64_t gamma[2], gl2 out[2]) {
    evalctx_ext c;
    c.local = local; c.next = next; c.aux = aux; c.z_local = z_local; c.z_next = z_next;
    c.per = per;
    c.lag_first = lag_first; c.lag_last = lag_last; c.z_last = z_last;
    c.alpha[0] = alpha[0]; c.alpha[1] = alpha[1]; c.beta[0] = beta[0]; c.beta[1] = beta[1];
    c.gamma[0] = gamma[0]; c.gamma[1] = gamma[1];
    eval_all_ext(air, &c);
    out[0] = c.acc[0]; out[1] = c.acc[1];
}

size_t orc_air_num_constraints(const air_spec_t *a) { return (size_t)a->n_constraints + 3 + 6 * (size_t)a->n_checked; }

/* Debug aid used by the tests: evaluate every constraint on trace row `row` (selectors in {0,1}); returns
 * the index of the first non-zero constraint or -1.  Z columns are not checked here (pass zeros -> skipped). */
long orc_trace_check_row(const orc_trace *t, size_t row) {
    const air_spec_t *a = t->air;
    size_t n = (size_t)1 << t->log_n;
    int W = t->width;
    uint64_t *local = (uint64_t *)malloc(sizeof(uint64_t) * W), *next = (uint64_t *)malloc(sizeof(uint64_t) * W);
    uint64_t *aux = (uint64_t *)calloc(a->n_aux ? a->n_aux : 1, sizeof(uint64_t));
    for (int c = 0; c < W; c++) { local[c] = t->trace[(size_t)c * n + row]; next[c] = t->trace[(size_t)c * n + (row + 1) % n]; }
    evalctx_base c;
    memset(&c, 0, sizeof c);
    c.local = local; c.next = next; c.aux = aux;
    uint64_t *perv = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)orc_air_n_per(a));
    for (int k = 0; k < AIR_N_PERIODIC; k++) perv[k] = (row % (size_t)AIR_PERIODIC[k][0]) == (size_t)AIR_PERIODIC[k][1];
    for (int k = 0; k < orc_air_n_vper(a); k++)
        perv[AIR_N_PERIODIC + k] = gl_from_i64(air_vper_value(a, k, (int)(row & (((size_t)1 << a->log_rows) - 1))));
    c.per = perv;
    size_t io = row >> a->log_rows;
    for (int ai = 0; ai < a->n_aux; ai++) aux[ai] = orc_aux_value(a, t->pis, io, ai);
    /* alpha = 0 turns acc into "the last emitted constraint": walk constraint by constraint instead */
    c.alpha[0] = c.alpha[1] = 0;
    /* run the program with a probing consumer: use alpha = 0 and stop at first non-zero by re-running prefix */
    long bad = -1;
    {
        /* cheap trick: evaluate with two random-ish alphas; if the combination is zero all constraints are (whp) zero */
        c.alpha[0] = 0x123456789abcdefULL % GL_P; c.alpha[1] = 0xfedcba987654321ULL % GL_P;
        c.acc[0] = c.acc[1] = 0; c.n_emitted = 0;
        eval_program_base(a, &c);
        if (c.acc[0] != 0 || c.acc[1] != 0) bad = (long)c.n_emitted;
    }
    free(local); free(next); free(aux); free(perv);
    return bad;
}
