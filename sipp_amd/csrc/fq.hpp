// sipp_amd/csrc/fq.hpp -- BN254 base field Fq on gfx950: 8 x u32 limbs, Montgomery form (R = 2^261).
//
// Used only by the trace-fill kernels (the native double-and-add / square-and-multiply chains whose
// intermediate values become trace cells).  Replaces ark-bn254's Fq (reference Cargo.toml:9) on the device.
// Values live in registers as 8 x 32-bit words; a product re-cuts its operands into 9 x 29-bit limbs so that every
// column of the schoolbook product AND of the Montgomery reduction is a plain chain of v_mad_u64_u32 into one 64-bit
// accumulator (18 terms < 2^58 never overflow): one instruction per limb product, no carry bookkeeping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fq {

struct Fq {
    uint32_t l[8];
};

// p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
__device__ __constant__ const uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
// p in 9 limbs of 29 bits, and -p^-1 mod 2^29
constexpr uint32_t P29[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u,
                             0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
constexpr uint32_t NINV29 = 0x04866389u;
constexpr uint32_t M29 = (1u << 29) - 1;
// R^2 mod p (R = 2^261): to Montgomery form
__device__ __constant__ const uint32_t R2[8] = {0x659bac10u, 0xe1a2a074u, 0x5406005au, 0x63985586u,
                                                0x2d3e2632u, 0xff54c580u, 0x34ea65a6u, 0x2a11a68cu};
// R mod p: one in Montgomery form
__device__ __constant__ const uint32_t ONE_M[8] = {0x157ccc21u, 0x4e8384ebu, 0x0ce148c3u, 0xfb90a602u,
                                                   0x819caa36u, 0x5301fa84u, 0x563d4475u, 0x0dc83629u};

__device__ __forceinline__ Fq zero() {
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
}
__device__ __forceinline__ Fq one_m() {
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = ONE_M[i];
    return r;
}
__device__ __forceinline__ bool is_zero(const Fq& a) {
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) t |= a.l[i];
    return t == 0;
}
// All branch-free: carry chains (v_add_co / v_addc_co) and selects, so the compiler never emits exec-mask branches.
// r = a - p and the final borrow (1 when a < p)
__device__ __forceinline__ uint32_t sub_p_borrow(Fq& r, const Fq& a) {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = __builtin_subc(a.l[i], P[i], bw, &bw);
    return bw;
}
__device__ __forceinline__ bool geq_p(const Fq& a) {
    Fq d;
    return sub_p_borrow(d, a) == 0;
}
// a mod p for a < 2p
__device__ __forceinline__ Fq reduce_once(const Fq& a, uint32_t carry = 0) {
    Fq d, r;
    const uint32_t bw = sub_p_borrow(d, a);
    const bool take = carry || !bw;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = take ? d.l[i] : a.l[i];
    return r;
}
__device__ __forceinline__ void sub_p(Fq& a) {
    Fq d;
    sub_p_borrow(d, a);
    a = d;
}
__device__ __forceinline__ Fq add(const Fq& a, const Fq& b) {
    Fq s;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = __builtin_addc(a.l[i], b.l[i], c, &c);
    return reduce_once(s, c);
}
__device__ __forceinline__ Fq sub(const Fq& a, const Fq& b) {
    Fq d, e, r;
    uint32_t bw = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d.l[i] = __builtin_subc(a.l[i], b.l[i], bw, &bw);
#pragma unroll
    for (int i = 0; i < 8; i++) e.l[i] = __builtin_addc(d.l[i], P[i], c, &c);
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = bw ? e.l[i] : d.l[i];
    return r;
}
__device__ __forceinline__ Fq neg(const Fq& a) { return sub(zero(), a); }
__device__ __forceinline__ Fq dbl(const Fq& a) { return add(a, a); }

// 8 x 32 -> 9 x 29 bits
__device__ __forceinline__ void cut29(uint32_t (&o)[9], const Fq& a) {
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int w = (29 * k) >> 5, sh = (29 * k) & 31;
        uint32_t v = a.l[w] >> sh;
        if (sh > 3 && w < 7) v |= a.l[w + 1] << (32 - sh);
        o[k] = v & M29;
    }
}
// 9 x 29 (each limb < 2^29) -> 8 x 32; bits above 256 are dropped (callers guarantee the value is < 2p < 2^255)
__device__ __forceinline__ Fq join29(const uint32_t (&r)[9]) {
    Fq o;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = (32 * j) / 29, sh = 32 * j - 29 * k;  // word j starts at bit sh of limb k
        uint32_t v = r[k] >> sh;
        v |= r[k + 1] << (29 - sh);
        if (58 - sh < 32 && k + 2 < 9) v |= r[k + 2] << (58 - sh);
        o.l[j] = v;
    }
    return o;
}

// Montgomery product a * b * R^-1 mod p, column-wise (Comba) in radix 2^29
__device__ __forceinline__ Fq mul(const Fq& a, const Fq& b) {
    uint32_t x[9], y[9], m[9], r[9];
    cut29(x, a);
    cut29(y, b);
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (j >= 0 && j < 9) acc += (uint64_t)x[i] * y[j];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = k - i;
            if (i < k && j >= 0 && j < 9) acc += (uint64_t)m[i] * P29[j];
        }
        if (k < 9) {
            m[k] = ((uint32_t)acc * NINV29) & M29;
            acc += (uint64_t)m[k] * P29[0];
        } else {
            r[k - 9] = (uint32_t)acc & M29;
        }
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
    return reduce_once(join29(r));
}
__device__ __forceinline__ Fq sqr(const Fq& a) { return mul(a, a); }

__device__ __forceinline__ Fq to_mont(const Fq& a) {
    Fq r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = R2[i];
    return mul(a, r2);
}
__device__ __forceinline__ Fq from_mont(const Fq& a) {
    Fq o = zero();
    o.l[0] = 1;
    return mul(a, o);
}
// small constant in Montgomery form
__device__ __forceinline__ Fq small_m(uint32_t v) {
    Fq o = zero();
    o.l[0] = v;
    return to_mont(o);
}

// a^(p-2), Montgomery in / out.  Not unrolled: ~380 products.
__device__ __noinline__ Fq inv(const Fq& a) {
    Fq r = one_m();
    for (int i = 255; i >= 0; i--) {
        r = sqr(r);
        uint32_t w = P[i >> 5];
        if (i < 32) w -= 2;  // exponent p - 2 (low word of p ends in ...47, no borrow)
        if ((w >> (i & 31)) & 1) r = mul(r, a);
    }
    return r;
}

// R^3 mod p: takes the plain inverse of a Montgomery value back to Montgomery form
__device__ __constant__ const uint32_t R3[8] = {0x4e2312b2u, 0x26d80b94u, 0x85af210eu, 0xfdde6f98u,
                                                0xad1d1adau, 0xb923065cu, 0xa668ff5au, 0x26c2d286u};
// Inversion by the binary extended Euclidean algorithm, Montgomery in / out, for code that runs on ONE lane (the single
// inversion of a final exponentiation): ~750 shift / subtract steps of ~50 instructions instead of the 380 field products
// (~300 instructions each) of a^(p-2).  Data-dependent control flow: not for waves whose lanes invert different values.
// inv_gcd(0) = 0.
__device__ __forceinline__ bool is_even(const Fq& a) { return (a.l[0] & 1u) == 0; }
__device__ __forceinline__ void shr1(Fq& a, uint32_t top) {
#pragma unroll
    for (int i = 0; i < 7; i++) a.l[i] = (a.l[i] >> 1) | (a.l[i + 1] << 31);
    a.l[7] = (a.l[7] >> 1) | (top << 31);
}
// x <- x / 2 mod p for x < p
__device__ __forceinline__ void half_mod(Fq& x) {
    uint32_t c = 0;
    if (!is_even(x)) {
#pragma unroll
        for (int i = 0; i < 8; i++) x.l[i] = __builtin_addc(x.l[i], P[i], c, &c);
    }
    shr1(x, c);
}
__device__ __forceinline__ bool geq(const Fq& a, const Fq& b) {
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) (void)__builtin_subc(a.l[i], b.l[i], bw, &bw);
    return bw == 0;
}
__device__ __forceinline__ void sub_raw(Fq& a, const Fq& b) {   // a >= b
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) a.l[i] = __builtin_subc(a.l[i], b.l[i], bw, &bw);
}
__device__ __noinline__ Fq inv_gcd(const Fq& a) {
    if (is_zero(a)) return a;
    Fq u = a, v, x1 = zero(), x2 = zero();
#pragma unroll
    for (int i = 0; i < 8; i++) v.l[i] = P[i];
    x1.l[0] = 1;
    auto is_one = [](const Fq& t) {
        uint32_t r = t.l[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < 8; i++) r |= t.l[i];
        return r == 0;
    };
    while (!is_one(u) && !is_one(v)) {
        while (is_even(u)) {
            shr1(u, 0);
            half_mod(x1);
        }
        while (is_even(v)) {
            shr1(v, 0);
            half_mod(x2);
        }
        if (geq(u, v)) {
            sub_raw(u, v);
            x1 = sub(x1, x2);
        } else {
            sub_raw(v, u);
            x2 = sub(x2, x1);
        }
    }
    // plain inverse of (a R) = a^-1 R^-1; times R^3 (Montgomery product) = a^-1 R
    Fq r3;
#pragma unroll
    for (int i = 0; i < 8; i++) r3.l[i] = R3[i];
    return mul(is_one(u) ? x1 : x2, r3);
}

// ---- Fq2 = Fq[u]/(u^2 + 1), Montgomery components ----
struct Fq2 {
    Fq c0, c1;
};
__device__ __forceinline__ Fq2 add(const Fq2& a, const Fq2& b) { return Fq2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
__device__ __forceinline__ Fq2 sub(const Fq2& a, const Fq2& b) { return Fq2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
__device__ __forceinline__ Fq2 mul(const Fq2& a, const Fq2& b) {
    Fq t0 = mul(a.c0, b.c0), t1 = mul(a.c1, b.c1);
    Fq s = mul(add(a.c0, a.c1), add(b.c0, b.c1));
    return Fq2{sub(t0, t1), sub(sub(s, t0), t1)};
}
// (a0 + a1 u)^2 = (a0 + a1)(a0 - a1) + 2 a0 a1 u : two base-field products instead of three
__device__ __forceinline__ Fq2 sqr(const Fq2& a) {
    Fq t = mul(a.c0, a.c1);
    return Fq2{mul(add(a.c0, a.c1), sub(a.c0, a.c1)), add(t, t)};
}
__device__ __forceinline__ Fq2 inv(const Fq2& a) {
    Fq n = add(sqr(a.c0), sqr(a.c1));
    Fq ni = inv(n);
    return Fq2{mul(a.c0, ni), neg(mul(a.c1, ni))};
}
__device__ __forceinline__ Fq2 inv_gcd(const Fq2& a) {
    Fq n = add(sqr(a.c0), sqr(a.c1));
    Fq ni = inv_gcd(n);
    return Fq2{mul(a.c0, ni), neg(mul(a.c1, ni))};
}
__device__ __forceinline__ bool is_zero(const Fq2& a) { return is_zero(a.c0) && is_zero(a.c1); }

}  // namespace fq
