// sipp_amd/csrc/gl.hpp -- Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension
// F_p[X]/(X^2 - 7) for gfx950 device code and the host-side driver.
//
// All values that cross a kernel boundary are canonical (< p).  64-bit integer modular
// arithmetic on a 32-bit VALU: a mulmod is four 32x32->64 multiply-adds plus a
// shift/add reduction (2^64 = 2^32 - 1, 2^96 = -1 mod p) -- no MFMA anywhere.
//
// Replaces (on the GPU side) plonky2's GoldilocksField @ InternetMaximalism/plonky2 541e127
// (reference Cargo.toml:21; used as `F` at reference src/prover_native.rs:7,12).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace gl {

constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p
constexpr uint64_t GEN = 7;              // multiplicative generator == coset shift
constexpr uint64_t TWO_ADIC_ROOT = 1753635133440165772ULL;  // order 2^32
constexpr uint64_t EXT_W = 7;            // X^2 = 7

GL_HD uint64_t canon(uint64_t a) { return a >= P ? a - P : a; }

GL_HD uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t t = s - P;  // == s + EPS mod 2^64
    return (s < a || s >= P) ? t : s;
}

GL_HD uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    return a < b ? d + P : d;
}

GL_HD uint64_t neg(uint64_t a) { return a ? P - a : 0; }

GL_HD uint64_t dbl(uint64_t a) { return add(a, a); }

// (hi, lo) -> [0, 2^64), congruent mod p but NOT necessarily canonical.  hi:lo is any 128-bit value.
// x = lo + 2^64 hl + 2^96 hh  ==  lo - (hh + hl) + (hl << 32)      (2^64 = 2^32 - 1, 2^96 = -1)
// The borrow of the subtraction and the carry of the addition are netted into ONE correction of +-(2^32 - 1).
// (The correction as corr * EPS costs a fifth v_mad_u64_u32; two selects on the carry / borrow masks + one 64-bit addition
// have one instruction fewer and are SLOWER: 1.63-1.71 against 1.73-1.75 G leaf permutations/s, 65.5 against 62.3 ms per
// n = 128 instance -- round 2, same box, alternating.)
GL_HD uint64_t reduce128_nc(uint64_t hi, uint64_t lo) {
    const uint32_t hl = (uint32_t)hi, hh = (uint32_t)(hi >> 32);
    const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32);
    uint32_t s0 = hh + hl;
    const uint32_t s1 = s0 < hh;
    const uint32_t a0 = l0 - s0;
    const uint32_t bw0 = l0 < s0;
    const uint32_t a1a = l1 - s1;
    const uint32_t a1 = a1a - bw0;
    const bool b1 = (l1 < s1) | (a1a < bw0);
    const uint32_t b1h = a1 + hl;
    const bool c1 = b1h < a1;
    const uint64_t b = ((uint64_t)b1h << 32) | a0;
    const int32_t corr = (int32_t)c1 - (int32_t)b1;  // -1, 0, +1
    return b + (uint64_t)((int64_t)corr * (int64_t)EPS);
}

GL_HD uint64_t reduce128(uint64_t hi, uint64_t lo) { return canon(reduce128_nc(hi, lo)); }

// value = lo + 2^64 * hi32 with hi32 < 2^32 (a "96-bit" accumulator) -> [0, 2^64), not canonical
GL_HD uint64_t reduce96_nc(uint32_t hi32, uint64_t lo) {
    uint64_t t1 = ((uint64_t)hi32 << 32) - hi32;
    uint64_t r = lo + t1;
    if (r < lo) r += EPS;
    return r;
}
GL_HD uint64_t reduce96(uint32_t hi32, uint64_t lo) { return canon(reduce96_nc(hi32, lo)); }

// 64 x 64 -> 128: four 32 x 32 + 64 multiply-adds (v_mad_u64_u32), no separate low product
GL_HD void mul_wide(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    hi = (uint64_t)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    lo = (t2 << 32) | (uint32_t)t0;
#else
    unsigned __int128 p = (unsigned __int128)a * b;
    lo = (uint64_t)p;
    hi = (uint64_t)(p >> 64);
#endif
}

// product in [0, 2^64), not canonical; inputs may be any u64
GL_HD uint64_t mul_nc(uint64_t a, uint64_t b) {
    uint64_t hi, lo;
    mul_wide(a, b, hi, lo);
    return reduce128_nc(hi, lo);
}

GL_HD uint64_t mul(uint64_t a, uint64_t b) { return canon(mul_nc(a, b)); }

GL_HD uint64_t sqr(uint64_t a) { return mul(a, a); }

// a * b + c
GL_HD uint64_t mad_nc(uint64_t a, uint64_t b, uint64_t c) {
    uint64_t hi, lo;
    mul_wide(a, b, hi, lo);
    lo += c;
    hi += (lo < c);
    return reduce128_nc(hi, lo);
}
GL_HD uint64_t mad(uint64_t a, uint64_t b, uint64_t c) { return canon(mad_nc(a, b, c)); }
// a (any u64) + c (canonical) -> [0, 2^64)
GL_HD uint64_t add_nc(uint64_t a, uint64_t c) {
    uint64_t s = a + c;
    return s < a ? s + EPS : s;
}

GL_HD uint64_t pow(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

// a^(p - 2) with the addition chain for p - 2 = 2^64 - 2^32 - 1 = (31 ones) 0 (32 ones): 63 squarings + 9 products
// instead of the 126 of square-and-multiply.  inv(0) = 0.
GL_HD uint64_t sqr_n(uint64_t a, int n) {
    for (int i = 0; i < n; i++) a = mul_nc(a, a);
    return a;
}
GL_HD uint64_t inv(uint64_t a) {
    const uint64_t t2 = mul_nc(mul_nc(a, a), a);            // 2 ones
    const uint64_t t3 = mul_nc(mul_nc(t2, t2), a);          // 3 ones
    const uint64_t t6 = mul_nc(sqr_n(t3, 3), t3);
    const uint64_t t12 = mul_nc(sqr_n(t6, 6), t6);
    const uint64_t t24 = mul_nc(sqr_n(t12, 12), t12);
    const uint64_t t30 = mul_nc(sqr_n(t24, 6), t6);
    const uint64_t t31 = mul_nc(mul_nc(t30, t30), a);       // 31 ones
    const uint64_t t63 = mul_nc(sqr_n(t31, 32), t31);       // (31 ones)(0)(31 ones)
    return mul(mul_nc(t63, t63), a);                        // (31 ones)(0)(32 ones)
}

GL_HD uint64_t root_of_unity(unsigned k) {
    uint64_t r = TWO_ADIC_ROOT;
    for (unsigned i = k; i < 32; i++) r = sqr(r);
    return r;
}

// signed small integer -> field
GL_HD uint64_t from_i64(int64_t v) { return v >= 0 ? (uint64_t)v : P - (uint64_t)(-v); }

#if defined(__HIPCC__)
#define GL_D __device__ __forceinline__
// ---- lazy dot products on the 32-bit VALU ----
// sum of (64-bit value x) x (64-bit value c) products with c cut into limbs c0 + c1 2^22 + c2 2^44 (22 + 22 + 20 bits):
// value = a[0] + a[1] 2^22 + a[2] 2^44 + 2^32 (a[3] + a[4] 2^22 + a[5] 2^44).  Each product is < 2^54, so a[] holds
// 1024 of them.
struct Acc6 {
    uint64_t a[6];
    GL_D void zero() {
#pragma unroll
        for (int j = 0; j < 6; j++) a[j] = 0;
    }
    GL_D void set(uint32_t lo, uint32_t hi) {
        zero();
        a[0] = lo;
        a[3] = hi;
    }
    GL_D void mac(uint32_t xl, uint32_t xh, const uint32_t* __restrict__ c) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            a[j] += (uint64_t)xl * c[j];
            a[3 + j] += (uint64_t)xh * c[j];
        }
    }
    // t[0..3] = a0 + a1 2^22 + a2 2^44 (a's < 2^60: the sum is < 2^105)
    static GL_D void fold3(uint32_t (&t)[4], uint64_t a0, uint64_t a1, uint64_t a2) {
        const uint32_t a1l = (uint32_t)a1, a1h = (uint32_t)(a1 >> 32), a2l = (uint32_t)a2, a2h = (uint32_t)(a2 >> 32);
        uint32_t c = 0;
        t[0] = __builtin_addc((uint32_t)a0, a1l << 22, c, &c);
        t[1] = __builtin_addc((uint32_t)(a0 >> 32), (uint32_t)(a1 >> 10), c, &c);
        t[2] = (a1h >> 10) + c;
        c = 0;
        t[1] = __builtin_addc(t[1], a2l << 12, c, &c);
        t[2] = __builtin_addc(t[2], (uint32_t)(a2 >> 20), c, &c);
        t[3] = (a2h >> 20) + c;
    }
    // -> [0, 2^64), congruent to the value, not canonical
    GL_D uint64_t reduce() const {
        uint32_t l[4], h[4], v[5];
        fold3(l, a[0], a[1], a[2]);
        fold3(h, a[3], a[4], a[5]);
        uint32_t c = 0;
        v[0] = l[0];
        v[1] = __builtin_addc(l[1], h[0], c, &c);
        v[2] = __builtin_addc(l[2], h[1], c, &c);
        v[3] = __builtin_addc(l[3], h[2], c, &c);
        v[4] = h[3] + c;
        const uint64_t r = reduce128_nc(((uint64_t)v[3] << 32) | v[2], ((uint64_t)v[1] << 32) | v[0]);
        const uint64_t t = (uint64_t)v[4] << 32;  // 2^128 = -2^32 (mod p)
        const uint64_t d = r - t;
        return r < t ? d - EPS : d;
    }
};

// sum of full 64 x 64 products (both factors per-lane values) in 160 bits, one reduction at the end
struct Acc160 {
    uint64_t lo = 0, hi = 0;
    uint32_t c = 0;
    GL_D void mac(uint64_t a, uint64_t b) {
        uint64_t ph, pl;
        mul_wide(a, b, ph, pl);
        lo += pl;
        ph += (lo < pl);  // ph <= 2^64 - 2, cannot wrap
        hi += ph;
        c += (hi < ph);
    }
    // 2^128 = -2^32 (mod p); result in [0, 2^64), not canonical
    GL_D uint64_t reduce() const {
        const uint64_t r = reduce128_nc(hi, lo), t = (uint64_t)c << 32;
        const uint64_t d = r - t;
        return r < t ? d - EPS : d;  // r < t < 2^37: d wrapped is >= 2^64 - 2^37 > EPS
    }
};

// c -> limbs for Acc6::mac
GL_D void limbs3(uint32_t (&o)[3], uint64_t c) {
    o[0] = (uint32_t)c & 0x3FFFFFu;
    o[1] = (uint32_t)(c >> 22) & 0x3FFFFFu;
    o[2] = (uint32_t)(c >> 44);
}
#endif

// ---- quadratic extension ----
struct E2 {
    uint64_t c0, c1;
};

GL_HD E2 e2(uint64_t c0, uint64_t c1 = 0) { return E2{c0, c1}; }
GL_HD E2 add(E2 a, E2 b) { return E2{add(a.c0, b.c0), add(a.c1, b.c1)}; }
GL_HD E2 sub(E2 a, E2 b) { return E2{sub(a.c0, b.c0), sub(a.c1, b.c1)}; }
GL_HD E2 neg(E2 a) { return E2{neg(a.c0), neg(a.c1)}; }
GL_HD E2 mul(E2 a, E2 b) {
    uint64_t t = mul(a.c1, b.c1);
    uint64_t c0 = mad(a.c0, b.c0, mul(t, EXT_W));
    uint64_t c1 = mad(a.c0, b.c1, mul(a.c1, b.c0));
    return E2{c0, c1};
}
GL_HD E2 scale(E2 a, uint64_t s) { return E2{mul(a.c0, s), mul(a.c1, s)}; }
GL_HD E2 sqr(E2 a) { return mul(a, a); }
GL_HD bool eq(E2 a, E2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
GL_HD E2 inv(E2 a) {
    uint64_t n = sub(sqr(a.c0), mul(EXT_W, sqr(a.c1)));
    uint64_t ni = inv(n);
    return E2{mul(a.c0, ni), mul(neg(a.c1), ni)};
}
GL_HD E2 pow(E2 a, uint64_t e) {
    E2 r{1, 0};
    while (e) {
        if (e & 1) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

GL_HD uint32_t bitrev(uint32_t x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
#endif
}

}  // namespace gl
