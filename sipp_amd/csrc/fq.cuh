// sipp_amd/csrc/fq.cuh -- BN254 base field Fq on gfx950: 8 x u32 limbs, Montgomery form (R = 2^256).
//
// Used only by the trace-fill kernels (the native double-and-add / square-and-multiply chains whose
// intermediate values become trace cells).  Replaces ark-bn254's Fq (reference Cargo.toml:9) on the device.
// 32-bit limbs because the VALU is 32-bit: one limb product is ONE v_mad_u64_u32.
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
// -p^-1 mod 2^32
constexpr uint32_t NINV = 0xe4866389u;
// R^2 mod p (R = 2^256): to Montgomery form
__device__ __constant__ const uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                                0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
// R mod p: one in Montgomery form
__device__ __constant__ const uint32_t ONE_M[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                                   0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};

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
__device__ __forceinline__ bool geq_p(const Fq& a) {
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (a.l[i] > P[i]) return true;
        if (a.l[i] < P[i]) return false;
    }
    return true;
}
__device__ __forceinline__ void sub_p(Fq& a) {
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - P[i] - borrow;
        a.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
}
__device__ __forceinline__ Fq add(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    if (c || geq_p(r)) sub_p(r);
    return r;
}
__device__ __forceinline__ Fq sub(const Fq& a, const Fq& b) {
    Fq r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint32_t)t;
        borrow = (t >> 32) & 1;
    }
    if (borrow) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r.l[i] + P[i];
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}
__device__ __forceinline__ Fq neg(const Fq& a) { return sub(zero(), a); }
__device__ __forceinline__ Fq dbl(const Fq& a) { return add(a, a); }

// Montgomery product a * b * R^-1 mod p (CIOS, 32-bit words)
__device__ __forceinline__ Fq mul(const Fq& a, const Fq& b) {
    uint32_t t[10];
#pragma unroll
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.l[j] * b.l[i] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        t[9] = (uint32_t)(c >> 32);
        uint32_t m = t[0] * NINV;
        c = (uint64_t)m * P[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * P[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t[9] + (uint32_t)(c >> 32);
    }
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    if (t[8] || geq_p(r)) sub_p(r);
    return r;
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
__device__ __forceinline__ bool is_zero(const Fq2& a) { return is_zero(a.c0) && is_zero(a.c1); }

}  // namespace fq
