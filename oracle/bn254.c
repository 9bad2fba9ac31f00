/*
 * oracle/bn254.c -- see bn254.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED.
 * Montgomery multiplication (CIOS, R = 2^256) behind a standard-representation interface.
 */
#include "bn254.h"
#include <string.h>

typedef unsigned __int128 u128;

/* p = 21888242871839275222246405745257275088696311157297823662689037894645226208583 */
static const uint64_t FQ_P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                 0x30644e72e131a029ULL};
static uint64_t FQ_INV;   /* -p^{-1} mod 2^64 */
static fq FQ_R2;          /* 2^512 mod p */
static int fq_ready = 0;

static int ge_p(const uint64_t a[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > FQ_P[i]) return 1;
        if (a[i] < FQ_P[i]) return 0;
    }
    return 1;
}

static void sub_p(uint64_t a[4]) {
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - FQ_P[i] - borrow;
        a[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
}

fq fq_add(fq a, fq b) {
    fq r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || ge_p(r.l)) sub_p(r.l);
    return r;
}

fq fq_sub(fq a, fq b) {
    fq r;
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
    if (borrow) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)r.l[i] + FQ_P[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}

fq fq_zero(void) { fq r; memset(&r, 0, sizeof r); return r; }
fq fq_from_u64(uint64_t v) { fq r = fq_zero(); r.l[0] = v; return r; }
fq fq_one(void) { return fq_from_u64(1); }
int fq_is_zero(fq a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
int fq_eq(fq a, fq b) { return memcmp(&a, &b, sizeof a) == 0; }
fq fq_neg(fq a) { return fq_sub(fq_zero(), a); }

/* a * b * R^-1 mod p */
static fq mont_mul(fq a, fq b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)t[j] + (u128)a.l[j] * b.l[i];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FQ_INV;
        c = (u128)t[0] + (u128)m * FQ_P[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)t[j] + (u128)m * FQ_P[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fq r;
    memcpy(r.l, t, 32);
    if (t[4] || ge_p(r.l)) sub_p(r.l);
    return r;
}

void fq_init(void) {
    if (fq_ready) return;
    uint64_t inv = 1; /* Newton: inv = p^-1 mod 2^64 */
    for (int i = 0; i < 6; i++) inv *= 2 - FQ_P[0] * inv;
    FQ_INV = (uint64_t)0 - inv;
    /* R2 = 2^512 mod p by repeated doubling */
    fq x = fq_one();
    for (int i = 0; i < 512; i++) x = fq_add(x, x);
    FQ_R2 = x;
    fq_ready = 1;
}

fq fq_mul(fq a, fq b) { return mont_mul(mont_mul(a, b), FQ_R2); }

fq fq_inv(fq a) {
    /* a^(p-2) */
    uint64_t e[4];
    memcpy(e, FQ_P, 32);
    e[0] -= 2;
    fq r = fq_one();
    for (int i = 255; i >= 0; i--) {
        r = fq_mul(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = fq_mul(r, a);
    }
    return r;
}

fq fq_from_u32(const uint32_t w[8]) {
    fq r;
    for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    return r;
}
void fq_to_u32(fq a, uint32_t w[8]) {
    for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)a.l[i]; w[2 * i + 1] = (uint32_t)(a.l[i] >> 32); }
}
void fq_to_limbs16(fq a, uint16_t l[16]) {
    for (int i = 0; i < 16; i++) l[i] = (uint16_t)(a.l[i / 4] >> (16 * (i % 4)));
}
fq fq_from_limbs16(const uint16_t l[16]) {
    fq r = fq_zero();
    for (int i = 0; i < 16; i++) r.l[i / 4] |= (uint64_t)l[i] << (16 * (i % 4));
    return r;
}

/* ---- Fq2 = Fq[u]/(u^2 + 1) ---- */
fq2 fq2_add(fq2 a, fq2 b) { fq2 r = {fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; return r; }
fq2 fq2_sub(fq2 a, fq2 b) { fq2 r = {fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; return r; }
fq2 fq2_mul(fq2 a, fq2 b) {
    fq2 r;
    r.c0 = fq_sub(fq_mul(a.c0, b.c0), fq_mul(a.c1, b.c1));
    r.c1 = fq_add(fq_mul(a.c0, b.c1), fq_mul(a.c1, b.c0));
    return r;
}
fq2 fq2_inv(fq2 a) {
    fq n = fq_add(fq_mul(a.c0, a.c0), fq_mul(a.c1, a.c1));
    fq ni = fq_inv(n);
    fq2 r = {fq_mul(a.c0, ni), fq_neg(fq_mul(a.c1, ni))};
    return r;
}
int fq2_is_zero(fq2 a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }

/* ---- Fq12 = Fq[w]/(w^12 - 18 w^6 + 82) ---- */
fq12 fq12_mul(const fq12 *a, const fq12 *b) {
    fq d[23];
    for (int m = 0; m < 23; m++) d[m] = fq_zero();
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) d[i + j] = fq_add(d[i + j], fq_mul(a->c[i], b->c[j]));
    fq c18 = fq_from_u64(18), c82 = fq_from_u64(82);
    /* w^m = 18 w^(m-6) - 82 w^(m-12), applied from the top down */
    for (int m = 22; m >= 12; m--) {
        d[m - 6] = fq_add(d[m - 6], fq_mul(c18, d[m]));
        d[m - 12] = fq_sub(d[m - 12], fq_mul(c82, d[m]));
    }
    fq12 r;
    for (int k = 0; k < 12; k++) r.c[k] = d[k];
    return r;
}
