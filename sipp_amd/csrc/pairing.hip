// sipp_amd/csrc/pairing.hip -- BN254 optimal ate pairing products on gfx950: the inner_product of the native SIPP prover
// (reference src/prover_native.rs:15-23: prod_i pairing(A_i, B_i), called 2 log2 n + 1 times per proof at :43,:51,:52; the
// reference takes `pairing` from plonky2-bn254-pairing @ fe5c3a8, not vendored).  SURVEY.md section 8(f) rank 3: first step
// of the widening beyond the STARK sub-provers -- a correct, oracle-checked path; not yet tuned.
//
// Mapping: one lane per (A_i, B_i) runs the Miller loop (affine twist arithmetic with one Fq2 inversion per step, the same
// algorithm step for step as oracle/py/bn254.py::miller_loop, so intermediate values can be compared), values in
// Fq12 = Fq2[w]/(w^6 - xi), xi = 9 + u, as six Fq2 coefficients in Montgomery form (fq.cuh).  One workgroup then multiplies
// the n Miller values (strided partial products + a tree) and lane 0 applies the final exponentiation: easy part
// (p^6 - 1)(p^2 + 1), hard part (p^4 - p^2 + 1)/r by the Devegili-Scott-Dahab chain (three powers by u); the result is
// exactly f^((p^12 - 1)/r).  Output: the 12 coefficients of the reference's MyFq12 form Fq[w]/(w^12 - 18 w^6 + 82)
// (c_i = a_i - 9 b_i, c_{i+6} = b_i for the Fq2 coefficient a_i + b_i u of w^i), 8 x u32 limbs each.
// The field routines are deliberately NOT inlined (one copy each, operands through pointers): the whole file is ~4 k
// Montgomery products per lane-step and would not fit the instruction cache otherwise.
#include "ctx.hpp"
#include "fq.cuh"
#include "pairing_constants.h"

namespace {

using fq::Fq;
using fq::Fq2;

struct T6 {
    Fq2 c[6];
};

__device__ __noinline__ void f2_mul(Fq2& r, const Fq2& a, const Fq2& b) { r = fq::mul(a, b); }
__device__ __noinline__ void f2_sqr(Fq2& r, const Fq2& a) { r = fq::sqr(a); }
__device__ __noinline__ void f2_inv(Fq2& r, const Fq2& a) { r = fq::inv(a); }
__device__ __noinline__ void fq_mul(Fq& r, const Fq& a, const Fq& b) { r = fq::mul(a, b); }

__device__ __forceinline__ Fq2 f2_zero() { return Fq2{fq::zero(), fq::zero()}; }
__device__ __forceinline__ Fq2 f2_one() { return Fq2{fq::one_m(), fq::zero()}; }
__device__ __forceinline__ Fq2 f2_neg(const Fq2& a) { return Fq2{fq::neg(a.c0), fq::neg(a.c1)}; }
__device__ __forceinline__ Fq2 f2_conj(const Fq2& a) { return Fq2{a.c0, fq::neg(a.c1)}; }
__device__ __forceinline__ Fq2 f2_dbl(const Fq2& a) { return fq::add(a, a); }
// (a0 + a1 u)(9 + u) = (9 a0 - a1) + (a0 + 9 a1) u
__device__ __noinline__ void f2_mul_xi(Fq2& r, const Fq2& a) {
    const Fq2 a2 = f2_dbl(a), a4 = f2_dbl(a2), a8 = f2_dbl(a4), a9 = fq::add(a8, a);
    r = Fq2{fq::sub(a9.c0, a.c1), fq::add(a9.c1, a.c0)};
}
// Fq2 times an Fq scalar
__device__ __forceinline__ void f2_scale(Fq2& r, const Fq2& a, const Fq& s) {
    fq_mul(r.c0, a.c0, s);
    fq_mul(r.c1, a.c1, s);
}

__device__ __noinline__ void t6_one(T6& r) {
    r.c[0] = f2_one();
    for (int i = 1; i < 6; i++) r.c[i] = f2_zero();
}

// r = a * b in Fq2[w]/(w^6 - xi); r may alias a or b
__device__ __noinline__ void t6_mul(T6& r, const T6& a, const T6& b) {
    Fq2 d[11];
    for (int k = 0; k < 11; k++) d[k] = f2_zero();
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            Fq2 t;
            f2_mul(t, a.c[i], b.c[j]);
            d[i + j] = fq::add(d[i + j], t);
        }
    for (int k = 0; k < 5; k++) {
        Fq2 t;
        f2_mul_xi(t, d[k + 6]);
        r.c[k] = fq::add(d[k], t);
    }
    r.c[5] = d[5];
}

// f *= l0 + l1 w + l3 w^3 with l0 in Fq (the line through twist points evaluated at a G1 point)
__device__ __noinline__ void t6_mul_line(T6& f, const Fq& l0, const Fq2& l1, const Fq2& l3) {
    Fq2 d[9];
    for (int k = 0; k < 9; k++) d[k] = f2_zero();
    for (int i = 0; i < 6; i++) {
        Fq2 t;
        f2_scale(t, f.c[i], l0);
        d[i] = fq::add(d[i], t);
        f2_mul(t, f.c[i], l1);
        d[i + 1] = fq::add(d[i + 1], t);
        f2_mul(t, f.c[i], l3);
        d[i + 3] = fq::add(d[i + 3], t);
    }
    for (int k = 0; k < 3; k++) {
        Fq2 t;
        f2_mul_xi(t, d[k + 6]);
        f.c[k] = fq::add(d[k], t);
    }
    for (int k = 3; k < 6; k++) f.c[k] = d[k];
}

// x -> x^(p^6): w -> -w
__device__ __noinline__ void t6_conj(T6& r, const T6& a) {
    for (int i = 0; i < 6; i++) r.c[i] = (i & 1) ? f2_neg(a.c[i]) : a.c[i];
}
// x -> x^(p^k), k = 1..3
__device__ __noinline__ void t6_frob(T6& r, const T6& a, int k) {
    for (int i = 0; i < 6; i++) {
        const Fq2 c = (k & 1) ? f2_conj(a.c[i]) : a.c[i];
        f2_mul(r.c[i], c, pairing_k::GAMMA[k - 1][i]);
    }
}

// ---- Fq6 = Fq2[v]/(v^3 - xi), v = w^2: only for the inversion ----
struct S3 {
    Fq2 c[3];
};
__device__ __noinline__ void s3_mul(S3& r, const S3& a, const S3& b) {
    Fq2 d[5];
    for (int k = 0; k < 5; k++) d[k] = f2_zero();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            Fq2 t;
            f2_mul(t, a.c[i], b.c[j]);
            d[i + j] = fq::add(d[i + j], t);
        }
    Fq2 t;
    f2_mul_xi(t, d[3]);
    r.c[0] = fq::add(d[0], t);
    f2_mul_xi(t, d[4]);
    r.c[1] = fq::add(d[1], t);
    r.c[2] = d[2];
}
__device__ __noinline__ void s3_inv(S3& r, const S3& a) {
    Fq2 c0, c1, c2, t, u;
    f2_sqr(c0, a.c[0]);
    f2_mul(t, a.c[1], a.c[2]);
    f2_mul_xi(u, t);
    c0 = fq::sub(c0, u);  // a0^2 - xi a1 a2
    f2_sqr(t, a.c[2]);
    f2_mul_xi(c1, t);
    f2_mul(t, a.c[0], a.c[1]);
    c1 = fq::sub(c1, t);  // xi a2^2 - a0 a1
    f2_sqr(c2, a.c[1]);
    f2_mul(t, a.c[0], a.c[2]);
    c2 = fq::sub(c2, t);  // a1^2 - a0 a2
    Fq2 n, v;
    f2_mul(n, a.c[0], c0);
    f2_mul(t, a.c[2], c1);
    f2_mul(u, a.c[1], c2);
    t = fq::add(t, u);
    f2_mul_xi(v, t);
    n = fq::add(n, v);  // a0 c0 + xi (a2 c1 + a1 c2)
    Fq2 ni;
    f2_inv(ni, n);
    f2_mul(r.c[0], c0, ni);
    f2_mul(r.c[1], c1, ni);
    f2_mul(r.c[2], c2, ni);
}
// a = E + O w (E, O in Fq6): 1/a = (E - O w) / (E^2 - v O^2)
__device__ __noinline__ void t6_inv(T6& r, const T6& a) {
    S3 E{{a.c[0], a.c[2], a.c[4]}}, O{{a.c[1], a.c[3], a.c[5]}}, E2, O2, N, Ni, Ei, Oi;
    s3_mul(E2, E, E);
    s3_mul(O2, O, O);
    Fq2 t;
    f2_mul_xi(t, O2.c[2]);  // v * O^2 = (xi O2_2, O2_0, O2_1)
    N.c[0] = fq::sub(E2.c[0], t);
    N.c[1] = fq::sub(E2.c[1], O2.c[0]);
    N.c[2] = fq::sub(E2.c[2], O2.c[1]);
    s3_inv(Ni, N);
    s3_mul(Ei, E, Ni);
    s3_mul(Oi, O, Ni);
    for (int i = 0; i < 3; i++) {
        r.c[2 * i] = Ei.c[i];
        r.c[2 * i + 1] = f2_neg(Oi.c[i]);
    }
}

// r = a^u (u = 4965661367192848881, 63 bits), most significant bit first
__device__ __noinline__ void t6_pow_u(T6& r, const T6& a) {
    T6 acc = a;
    for (int i = pairing_k::U_BITS - 2; i >= 0; i--) {
        t6_mul(acc, acc, acc);
        if ((pairing_k::U >> i) & 1) t6_mul(acc, acc, a);
    }
    r = acc;
}

// f^((p^12 - 1)/r): oracle/py/bn254.py::final_exp computes the same value by plain square-and-multiply
__device__ __noinline__ void final_exp(T6& f) {
    T6 g, t, u;
    t6_conj(t, f);
    t6_inv(u, f);
    t6_mul(g, t, u);  // f^(p^6 - 1)
    t6_frob(t, g, 2);
    t6_mul(g, t, g);  // ^(p^2 + 1)
    // hard part, Devegili-Scott-Dahab
    T6 fx, fx2, fx3, y0, y1, y2, y3, y4, y5, y6, t0, t1;
    t6_pow_u(fx, g);
    t6_pow_u(fx2, fx);
    t6_pow_u(fx3, fx2);
    t6_frob(y0, g, 1);
    t6_frob(t, g, 2);
    t6_mul(y0, y0, t);
    t6_frob(t, g, 3);
    t6_mul(y0, y0, t);
    t6_conj(y1, g);
    t6_frob(y2, fx2, 2);
    t6_frob(t, fx, 1);
    t6_conj(y3, t);
    t6_frob(t, fx2, 1);
    t6_mul(t, fx, t);
    t6_conj(y4, t);
    t6_conj(y5, fx2);
    t6_frob(t, fx3, 1);
    t6_mul(t, fx3, t);
    t6_conj(y6, t);
    t6_mul(t0, y6, y6);
    t6_mul(t0, t0, y4);
    t6_mul(t0, t0, y5);
    t6_mul(t1, y3, y5);
    t6_mul(t1, t1, t0);
    t6_mul(t0, t0, y2);
    t6_mul(t1, t1, t1);
    t6_mul(t1, t1, t0);
    t6_mul(t1, t1, t1);
    t6_mul(t0, t1, y1);
    t6_mul(t1, t1, y0);
    t6_mul(t0, t0, t0);
    t6_mul(f, t0, t1);
}

// ---- Miller loop (oracle/py/bn254.py::_line, _step, miller_loop) ----
struct G2A {
    Fq2 x, y;
};
// line through T and Q (tangent when `tangent`) evaluated at (xp, yp), f *= line, T <- T + Q
__device__ __noinline__ void line_step(T6& f, G2A& T, const G2A& Q, bool tangent, const Fq& xp_neg, const Fq& yp, bool advance) {
    Fq2 lam, num, den, deni;
    if (tangent) {
        Fq2 x2;
        f2_sqr(x2, T.x);
        num = fq::add(f2_dbl(x2), x2);  // 3 x^2
        den = f2_dbl(T.y);
    } else {
        num = fq::sub(Q.y, T.y);
        den = fq::sub(Q.x, T.x);
    }
    f2_inv(deni, den);
    f2_mul(lam, num, deni);
    Fq2 l1, l3;
    f2_scale(l1, lam, xp_neg);  // -lam x_P
    f2_mul(l3, lam, T.x);
    l3 = fq::sub(l3, T.y);      // lam x_T - y_T
    t6_mul_line(f, yp, l1, l3);
    if (advance) {
        Fq2 l2, x3, y3;
        f2_sqr(l2, lam);
        x3 = fq::sub(fq::sub(l2, T.x), tangent ? T.x : Q.x);
        f2_mul(y3, lam, fq::sub(T.x, x3));
        y3 = fq::sub(y3, T.y);
        T.x = x3;
        T.y = y3;
    }
}

__device__ __forceinline__ Fq load_fq(const uint32_t* w) {
    Fq r;
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
    return fq::to_mont(r);
}

// one lane per pair; g1 [n][16] = (x, y), g2 [n][32] = (x.c0, x.c1, y.c0, y.c1) (reference src/transcript_native.rs:42-54 order);
// all-zero coordinates stand for the point at infinity (Miller value 1)
__global__ void __launch_bounds__(64) miller_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                    T6* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* pw = g1 + (size_t)i * 16;
    const uint32_t* qw = g2 + (size_t)i * 32;
    T6 f;
    t6_one(f);
    uint32_t nzp = 0, nzq = 0;
    for (int k = 0; k < 16; k++) nzp |= pw[k];
    for (int k = 0; k < 32; k++) nzq |= qw[k];
    if (nzp != 0 && nzq != 0) {
        const Fq xp = load_fq(pw), yp = load_fq(pw + 8);
        const Fq xpn = fq::neg(xp);
        G2A Q{Fq2{load_fq(qw), load_fq(qw + 8)}, Fq2{load_fq(qw + 16), load_fq(qw + 24)}};
        G2A T = Q;
        for (int b = pairing_k::ATE_BITS - 2; b >= 0; b--) {
            t6_mul(f, f, f);
            line_step(f, T, T, true, xpn, yp, true);
            const uint32_t bit = b >= 64 ? (pairing_k::ATE_HI >> (b - 64)) & 1u : (uint32_t)(pairing_k::ATE_LO >> b) & 1u;
            if (bit) line_step(f, T, Q, false, xpn, yp, true);
        }
        // Q1 = pi(Q), Q2 = -pi^2(Q): pi(x, y) = (conj(x) gamma_1,2, conj(y) gamma_1,3)
        G2A Q1, Q2;
        f2_mul(Q1.x, f2_conj(Q.x), pairing_k::GAMMA[0][2]);
        f2_mul(Q1.y, f2_conj(Q.y), pairing_k::GAMMA[0][3]);
        f2_mul(Q2.x, f2_conj(Q1.x), pairing_k::GAMMA[0][2]);
        f2_mul(Q2.y, f2_conj(Q1.y), pairing_k::GAMMA[0][3]);
        Q2.y = f2_neg(Q2.y);
        line_step(f, T, Q1, false, xpn, yp, true);
        line_step(f, T, Q2, false, xpn, yp, false);
    }
    out[i] = f;
}

// one workgroup: product of the n Miller values, final exponentiation, MyFq12 coefficients out.  `count` independent
// products (blockIdx.x): product k covers values [k * n, (k + 1) * n)
__global__ void __launch_bounds__(256) product_final_kernel(T6* __restrict__ vals, uint32_t n, uint32_t* __restrict__ out) {
    T6* v = vals + (size_t)blockIdx.x * n;
    const uint32_t t = threadIdx.x;
    // strided partial products into v[t]
    if (t < n) {
        T6 acc = v[t];
        for (uint32_t j = t + 256; j < n; j += 256) t6_mul(acc, acc, v[j]);
        v[t] = acc;
    }
    __syncthreads();
    const uint32_t m = n < 256 ? n : 256;
    for (uint32_t s = 1; s < m; s <<= 1) {
        if ((t % (2 * s)) == 0 && t + s < m) {
            T6 a = v[t];
            t6_mul(a, a, v[t + s]);
            v[t] = a;
        }
        __syncthreads();
    }
    if (t == 0) {
        T6 f = v[0];
        final_exp(f);
        uint32_t* o = out + (size_t)blockIdx.x * 96;
        const Fq nine = fq::small_m(9);
        for (int i = 0; i < 6; i++) {
            Fq nb;
            fq_mul(nb, nine, f.c[i].c1);
            const Fq lo = fq::from_mont(fq::sub(f.c[i].c0, nb)), hi = fq::from_mont(f.c[i].c1);
            for (int l = 0; l < 8; l++) {
                o[8 * i + l] = lo.l[l];
                o[8 * (i + 6) + l] = hi.l[l];
            }
        }
    }
}

}  // namespace

extern "C" int sipp_inner_products(sipp_ctx* ctx, const uint32_t* g1, const uint32_t* g2, size_t n, size_t count, uint32_t* out) {
    if (!ctx || !g1 || !g2 || !out || n == 0 || count == 0) return SIPP_E_BADARG;
    const size_t total = n * count;
    if (total > ((size_t)1 << 24)) return sipp_fail(ctx, SIPP_E_BADARG, "inner_products: too many pairs");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    ArenaMark mk = arena_mark(ctx);
    uint32_t* d_g1 = arena_alloc_t<uint32_t>(ctx, total * 16);
    uint32_t* d_g2 = arena_alloc_t<uint32_t>(ctx, total * 32);
    T6* d_f = arena_alloc_t<T6>(ctx, total);
    uint32_t* d_out = arena_alloc_t<uint32_t>(ctx, count * 96);
    int rc = SIPP_OK;
    if (!d_g1 || !d_g2 || !d_f || !d_out) rc = SIPP_E_NOMEM;
    if (rc == SIPP_OK) {
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_g1, g1, total * 64, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_g2, g2, total * 128, hipMemcpyHostToDevice, ctx->stream));
        {
            ProfScope ps(ctx, "pairing_miller");
            hipLaunchKernelGGL(miller_kernel, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, ctx->stream, d_g1, d_g2,
                               (uint32_t)total, d_f);
        }
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        {
            ProfScope ps(ctx, "pairing_product_final");
            hipLaunchKernelGGL(product_final_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, d_f, (uint32_t)n, d_out);
        }
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(out, d_out, count * 96 * 4, hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    arena_release(ctx, mk);
    return rc;
}

extern "C" int sipp_inner_product(sipp_ctx* ctx, const uint32_t* g1, const uint32_t* g2, size_t n, uint32_t* out) {
    return sipp_inner_products(ctx, g1, g2, n, 1, out);
}
