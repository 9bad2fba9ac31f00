/*
 * oracle/gl.h -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic
 * extension F_p[X]/(X^2 - 7), plain C.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product path; it is the CPU restatement the HIP kernels are checked against
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
 *
 * PARITY UNPINNED: the arithmetic restated here lives in a third-party
 * dependency that is absent from /root/reference:
 *   plonky2 @ InternetMaximalism/plonky2 rev 541e127
 *   (reference Cargo.toml:21; field/src/goldilocks_field.rs,
 *    field/src/goldilocks_extensions.rs upstream).
 * The reference's own use of this field: src/transcript_native.rs:1-12
 * (GoldilocksField as `F`), src/prover_native.rs:7,12.
 * Constants are verified arithmetically:
 *   generator 7, two-adic root 1753635133440165772 of order 2^32, W = 7 for
 *   the extension.
 */
#ifndef ORACLE_GL_H
#define ORACLE_GL_H

#include <stdint.h>
#include <stddef.h>

typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^32 - 1 == 2^64 mod p */
#define GL_GEN 7ULL
#define GL_TWO_ADIC_ROOT 1753635133440165772ULL /* order 2^32 */
#define GL_EXT_W 7ULL

/* (branch-free: the operands are as good as random, a mispredicted branch costs more than the arithmetic) */
static inline uint64_t gl_canon(uint64_t a) { return a - (GL_P & (0 - (uint64_t)(a >= GL_P))); }

static inline uint64_t gl_add(uint64_t a, uint64_t b) {
    /* a, b canonical */
    uint64_t s = a + b;
    return s - (GL_P & (0 - (uint64_t)((s < a) | (s >= GL_P))));
}

static inline uint64_t gl_sub(uint64_t a, uint64_t b) {
    return a - b + (GL_P & (0 - (uint64_t)(a < b)));
}

static inline uint64_t gl_neg(uint64_t a) { return a ? GL_P - a : 0; }

static inline uint64_t gl_reduce128(u128 x) {
    /* x = lo + 2^64 * (hi_lo + 2^32 * hi_hi);  2^64 = 2^32 - 1, 2^96 = -1 */
    uint64_t lo = (uint64_t)x;
    uint64_t hi = (uint64_t)(x >> 64);
    uint64_t hi_hi = hi >> 32;
    uint64_t hi_lo = hi & GL_EPS;
    uint64_t t0 = lo - hi_hi;
    t0 -= GL_EPS & (0 - (uint64_t)(lo < hi_hi)); /* borrow: + p  == - (2^32 - 1) mod 2^64 */
    uint64_t t1 = hi_lo * GL_EPS;
    uint64_t r = t0 + t1;
    r += GL_EPS & (0 - (uint64_t)(r < t0)); /* carry: 2^64 == 2^32 - 1 */
    return gl_canon(r);
}

static inline uint64_t gl_mul(uint64_t a, uint64_t b) {
    return gl_reduce128((u128)a * (u128)b);
}

static inline uint64_t gl_sqr(uint64_t a) { return gl_mul(a, a); }

static inline uint64_t gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, a);
        a = gl_sqr(a);
        e >>= 1;
    }
    return r;
}

static inline uint64_t gl_inv(uint64_t a) { return gl_pow(a, GL_P - 2); }

/* primitive 2^k-th root of unity, plonky2 `primitive_root_of_unity(k)` */
static inline uint64_t gl_root_of_unity(unsigned k) {
    uint64_t r = GL_TWO_ADIC_ROOT;
    for (unsigned i = k; i < 32; i++) r = gl_sqr(r);
    return r;
}

static inline uint64_t gl_from_i64(int64_t v) {
    return v >= 0 ? (uint64_t)v % GL_P : GL_P - ((uint64_t)(-v) % GL_P);
}

/* ---- quadratic extension: a = c0 + c1 * X, X^2 = 7 ---- */
typedef struct { uint64_t c0, c1; } gl2;

static inline gl2 gl2_make(uint64_t c0, uint64_t c1) { gl2 r = {c0, c1}; return r; }
static inline gl2 gl2_from(uint64_t c0) { gl2 r = {c0, 0}; return r; }
static inline gl2 gl2_add(gl2 a, gl2 b) { return gl2_make(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
static inline gl2 gl2_sub(gl2 a, gl2 b) { return gl2_make(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
static inline gl2 gl2_neg(gl2 a) { return gl2_make(gl_neg(a.c0), gl_neg(a.c1)); }
static inline gl2 gl2_mul(gl2 a, gl2 b) {
    uint64_t c0 = gl_add(gl_mul(a.c0, b.c0), gl_mul(GL_EXT_W, gl_mul(a.c1, b.c1)));
    uint64_t c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
    return gl2_make(c0, c1);
}
static inline gl2 gl2_scale(gl2 a, uint64_t s) { return gl2_make(gl_mul(a.c0, s), gl_mul(a.c1, s)); }
static inline gl2 gl2_sqr(gl2 a) { return gl2_mul(a, a); }
static inline int gl2_eq(gl2 a, gl2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline gl2 gl2_inv(gl2 a) {
    /* 1/(c0 + c1 X) = (c0 - c1 X) / (c0^2 - 7 c1^2) */
    uint64_t n = gl_sub(gl_sqr(a.c0), gl_mul(GL_EXT_W, gl_sqr(a.c1)));
    uint64_t ni = gl_inv(n);
    return gl2_make(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
static inline gl2 gl2_pow(gl2 a, uint64_t e) {
    gl2 r = gl2_from(1);
    while (e) {
        if (e & 1) r = gl2_mul(r, a);
        a = gl2_sqr(a);
        e >>= 1;
    }
    return r;
}

static inline uint32_t bitrev32(uint32_t x, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

#endif
