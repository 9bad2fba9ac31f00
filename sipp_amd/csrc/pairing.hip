// sipp_amd/csrc/pairing.hip -- BN254 optimal ate pairing products on gfx950: the inner_product of the native SIPP prover
// (reference src/prover_native.rs:15-23: prod_i pairing(A_i, B_i), called 2 log2 n + 1 times per proof at :43,:51,:52; the
// reference takes `pairing` from plonky2-bn254-pairing @ fe5c3a8, not vendored).  SURVEY.md section 8(f) rank 3: first step
// of the widening beyond the STARK sub-provers.
//
// Mapping: one WAVE per (A_i, B_i) runs the Miller loop (homogeneous projective twist arithmetic, no inversion; the 36 / 18 Fq2
// products of every Fq12 product and the products of one dependency level of a point step spread over the lanes), values in Fq12 = Fq2[w]/(w^6 - xi), xi = 9 + u, as six
// Fq2 coefficients in Montgomery form (fq.hpp), resident in LDS.  One workgroup then multiplies
// the n Miller values (strided partial products + a tree) and one wave applies the final exponentiation: easy part
// (p^6 - 1)(p^2 + 1), hard part by the chain of ark-ec 0.4's Bn::final_exponentiation (Fuentes-Castaneda et al.; three powers by
// u), as recalled: what `plonky2_bn254_pairing::pairing::pairing` (reference src/prover_native.rs:8,20; a restatement of arkworks'
// Bn254::pairing) returns is f^(lambda (p^12 - 1)/r) with
// lambda = 2u(6u^2 + 3u + 1), NOT the plain reduced pairing (round 6).  Output: the 12 coefficients of the reference's MyFq12 form Fq[w]/(w^12 - 18 w^6 + 82)
// (c_i = a_i - 9 b_i, c_{i+6} = b_i for the Fq2 coefficient a_i + b_i u of w^i), 8 x u32 limbs each.
// The field routines are deliberately NOT inlined (one copy each, operands through pointers) to keep the code small.
#include <vector>

#include "air_tables.h"
#include "ctx.hpp"
#include "fq.hpp"
#include "pairing_constants.h"
#include "pairing_rowsrc.h"

namespace {

using fq::Fq;
using fq::Fq2;

struct T6 {
    Fq2 c[6];
};

__device__ __noinline__ void f2_mul(Fq2& r, const Fq2& a, const Fq2& b) { r = fq::mul(a, b); }
__device__ __noinline__ void f2_sqr(Fq2& r, const Fq2& a) { r = fq::sqr(a); }
// one lane inverts one value (the easy part of a final exponentiation): binary extended Euclid instead of a^(p-2)
__device__ __noinline__ void f2_inv(Fq2& r, const Fq2& a) { r = fq::inv_gcd(a); }
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

// ---- final exponentiation f^(lambda (p^12 - 1)/r), lambda = 2u(6u^2 + 3u + 1), by ONE WAVE (oracle/py/bn254.py::final_exp
// computes the same value by plain square-and-multiply): easy part (p^6 - 1)(p^2 + 1), hard part by arkworks' chain (three powers
// by u).  Every Fq12 value lives in LDS, the 36 Fq2 products of an Fq12 product are
// taken by 36 lanes (then 11 lanes sum the anti-diagonals and 6 fold w^6 = xi), conjugations and Frobenius maps by 6 lanes;
// only the single inversion of the easy part runs on one lane.  ~290 Fq12 products: 33 ms on one lane, ~1 ms this way.
// The base-field products are the latency of everything here (one wave, dependent multiply-add chains): every Fq2 product is
// therefore split into its three Karatsuba parts a0 b0, a1 b1, (a0 + a1)(b0 + b1) on THREE lanes (two waves per Fq12 value:
// 36 products = 108 lanes), assembled by two lanes per product, summed along the anti-diagonals by one lane per component.
constexpr uint32_t NT = 128;   // threads of a pairing workgroup
struct CoopScratch {
    Fq part[126];
    Fq2 prod[36];
    Fq2 dsum[11];
};
// the Karatsuba part `part` of a * b
__device__ __forceinline__ void karatsuba_part(Fq& r, const Fq2& a, const Fq2& b, uint32_t part) {
    // inlined product: the operands are register values (a call through pointers would park them in scratch memory)
    const Fq x = part == 0 ? a.c0 : part == 1 ? a.c1 : fq::add(a.c0, a.c1);
    const Fq y = part == 0 ? b.c0 : part == 1 ? b.c1 : fq::add(b.c0, b.c1);
    r = fq::mul(x, y);
}
// component `comp` of the product whose three parts start at p[0]
__device__ __forceinline__ Fq karatsuba_join(const Fq* p, uint32_t comp) {
    return comp == 0 ? fq::sub(p[0], p[1]) : fq::sub(fq::sub(p[2], p[0]), p[1]);
}
// component `comp` of a * (9 + u) + b
__device__ __forceinline__ Fq xi_add_comp(const Fq2& a, const Fq2& b, uint32_t comp) {
    const Fq& m = comp ? a.c1 : a.c0;
    const Fq m2 = fq::dbl(m), m4 = fq::dbl(m2), m9 = fq::add(fq::dbl(m4), m);
    return comp ? fq::add(fq::add(m9, a.c0), b.c1) : fq::add(fq::sub(m9, a.c1), b.c0);
}
__device__ __forceinline__ void coop_sums_and_fold(T6& r, CoopScratch& sc) {
    const uint32_t l = threadIdx.x;
    if (l < 22) {   // anti-diagonal k, component comp
        const uint32_t k = l >> 1, comp = l & 1;
        Fq d = fq::zero();
        for (int i = 0; i < 6; i++) {
            const int j = (int)k - i;
            if (j >= 0 && j < 6) d = fq::add(d, comp ? sc.prod[i * 6 + j].c1 : sc.prod[i * 6 + j].c0);
        }
        (comp ? sc.dsum[k].c1 : sc.dsum[k].c0) = d;
    }
    __syncthreads();
    if (l < 12) {   // w^6 = xi
        const uint32_t k = l >> 1, comp = l & 1;
        const Fq v = k < 5 ? xi_add_comp(sc.dsum[k + 6], sc.dsum[k], comp) : (comp ? sc.dsum[5].c1 : sc.dsum[5].c0);
        (comp ? r.c[k].c1 : r.c[k].c0) = v;
    }
    __syncthreads();
}
__device__ __forceinline__ void coop_mul(T6& r, const T6& a, const T6& b, CoopScratch& sc) {  // r may alias a or b
    const uint32_t l = threadIdx.x;
    if (l < 108) karatsuba_part(sc.part[l], a.c[(l / 3) / 6], b.c[(l / 3) % 6], l % 3);
    __syncthreads();
    if (l < 72) {
        const uint32_t k = l >> 1, comp = l & 1;
        (comp ? sc.prod[k].c1 : sc.prod[k].c0) = karatsuba_join(&sc.part[3 * k], comp);
    }
    __syncthreads();
    coop_sums_and_fold(r, sc);
}
// the final exponentiation runs as ONE wave (its barriers cost nothing then; measured 1.93 ms against 2.19 ms with the
// 128-thread products and cyclotomic squarings, whose additions on six lanes outweigh the saved products)
struct CoopScratch64 {
    Fq2 prod[36];
    Fq2 dsum[11];
};
__device__ __forceinline__ void coop_mul64(T6& r, const T6& a, const T6& b, CoopScratch64& sc) {  // r may alias a or b
    const uint32_t l = threadIdx.x;
    if (l < 36) f2_mul(sc.prod[l], a.c[l / 6], b.c[l % 6]);
    __syncthreads();
    if (l < 11) {
        Fq2 d = f2_zero();
        for (int i = 0; i < 6; i++) {
            const int j = (int)l - i;
            if (j >= 0 && j < 6) d = fq::add(d, sc.prod[i * 6 + j]);
        }
        sc.dsum[l] = d;
    }
    __syncthreads();
    if (l < 6) {
        if (l < 5) {
            Fq2 t;
            f2_mul_xi(t, sc.dsum[l + 6]);
            r.c[l] = fq::add(sc.dsum[l], t);
        } else {
            r.c[5] = sc.dsum[5];
        }
    }
    __syncthreads();
}
// Granger-Scott squaring of an element of the cyclotomic subgroup (everything after the easy part of the final exponentiation):
// with a = sum a_i w^i the pairs (x, y) = (a0, a3), (a2, a5), (a1, a4) are squared as elements of Fq4 = Fq2[Y]/(Y^2 - xi):
//   (x + y Y)^2 = A + B Y,  A = x^2 + xi y^2,  B = 2 x y
//   a0' = 3 A(a0, a3) - 2 a0    a3' = 3 B(a0, a3) + 2 a3
//   a4' = 3 A(a2, a5) - 2 a4    a1' = 3 xi B(a2, a5) + 2 a1
//   a2' = 3 A(a1, a4) - 2 a2    a5' = 3 B(a1, a4) + 2 a5
// (arkworks' Fp12::cyclotomic_square in the w-power basis).  One wave: the TEN base-field products of a pair -- x0^2, x1^2, x0 x1,
// y0^2, y1^2, y0 y1, x0 y0, x1 y1, x0 y1, x1 y0 -- take raw components as operands (no additions in front of the products), 30
// lanes in one phase; then one lane per output component adds them up.  30 base-field products instead of 108.
struct CycloScratch {
    Fq part[30];
};
__device__ __forceinline__ Fq fq_x9(const Fq& a) {
    const Fq a2 = fq::dbl(a), a4 = fq::dbl(a2);
    return fq::add(fq::dbl(a4), a);
}
__device__ __forceinline__ void coop_cyclo_sqr64(T6& r, const T6& a, CycloScratch& cs) {  // r may alias a
    const uint32_t l = threadIdx.x;
    if (l < 30) {
        const uint32_t q = l / 10, m = l % 10;
        const Fq2& x = a.c[q == 0 ? 0 : q == 1 ? 2 : 1];
        const Fq2& y = a.c[q == 0 ? 3 : q == 1 ? 5 : 4];
        // operand table: (x0 x0) (x1 x1) (x0 x1) (y0 y0) (y1 y1) (y0 y1) (x0 y0) (x1 y1) (x0 y1) (x1 y0)
        const Fq& u = (m == 0 || m == 2 || m == 6 || m == 8) ? x.c0 : (m == 1 || m == 7 || m == 9) ? x.c1 : m == 3 || m == 5 ? y.c0 : y.c1;
        const Fq& v = m == 0 ? x.c0 : (m == 1 || m == 2) ? x.c1 : (m == 3 || m == 6) ? y.c0 : (m == 4 || m == 5 || m == 7 || m == 8) ? y.c1 : y.c0;
        cs.part[l] = fq::mul(u, v);
    }
    __syncthreads();
    if (l < 12) {
        const uint32_t q = l >> 2, c = l & 3;
        const Fq* p = &cs.part[10 * q];
        Fq out;
        if (c < 2) {
            const Fq dy = fq::sub(p[3], p[4]), ey = fq::dbl(p[5]);                       // y^2 = dy + ey u
            const Fq A = c == 0 ? fq::sub(fq::add(fq::sub(p[0], p[1]), fq_x9(dy)), ey)    // x0^2 - x1^2 + 9 dy - ey
                                : fq::add(fq::add(fq::dbl(p[2]), fq_x9(ey)), dy);         // 2 x0 x1 + 9 ey + dy
            const Fq2& z = a.c[q == 0 ? 0 : q == 1 ? 4 : 2];
            const Fq d = fq::sub(A, c ? z.c1 : z.c0);
            out = fq::add(fq::dbl(d), A);                                                // 3 A - 2 z
            Fq2& dst = r.c[q == 0 ? 0 : q == 1 ? 4 : 2];
            (c ? dst.c1 : dst.c0) = out;
        } else {
            const Fq B0 = fq::dbl(fq::sub(p[6], p[7])), B1 = fq::dbl(fq::add(p[8], p[9]));
            const uint32_t comp = c - 2;
            Fq t;                                                                         // component of B, or of xi B for the pair (a2, a5)
            if (q == 1) t = comp ? fq::add(fq_x9(B1), B0) : fq::sub(fq_x9(B0), B1);
            else t = comp ? B1 : B0;
            const Fq2& z = a.c[q == 0 ? 3 : q == 1 ? 1 : 5];
            const Fq s2 = fq::add(t, comp ? z.c1 : z.c0);
            out = fq::add(fq::dbl(s2), t);                                               // 3 t + 2 z
            Fq2& dst = r.c[q == 0 ? 3 : q == 1 ? 1 : 5];
            (comp ? dst.c1 : dst.c0) = out;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void coop_conj(T6& r, const T6& a) {
    const uint32_t l = threadIdx.x;
    if (l < 6) r.c[l] = (l & 1) ? f2_neg(a.c[l]) : a.c[l];
    __syncthreads();
}
__device__ __forceinline__ void coop_frob(T6& r, const T6& a, int k) {
    const uint32_t l = threadIdx.x;
    if (l < 6) {
        const Fq2 c = (k & 1) ? f2_conj(a.c[l]) : a.c[l];
        f2_mul(r.c[l], c, pairing_k::GAMMA[k - 1][l]);
    }
    __syncthreads();
}
__device__ __forceinline__ void coop_copy(T6& r, const T6& a) {
    const uint32_t l = threadIdx.x;
    if (l < 6) r.c[l] = a.c[l];
    __syncthreads();
}
__device__ __forceinline__ void coop_pow_u(T6& r, const T6& a, CoopScratch64& sc, CycloScratch& cs) {  // r != a, a cyclotomic
    coop_copy(r, a);
    for (int i = pairing_k::U_BITS - 2; i >= 0; i--) {
        coop_cyclo_sqr64(r, r, cs);
        if ((pairing_k::U >> i) & 1) coop_mul64(r, r, a, sc);
    }
}
struct FinalExpLds {
    T6 f, g, t, u, fx, fx2, fx3, y0, y1, y2, y3, y4, y5, y6, t0, t1;
    CoopScratch64 sc;
    CycloScratch cs;
};
// L.f in: the Miller product; L.f out: f^(lambda (p^12 - 1)/r).  Called by all 64 lanes of a one-wave workgroup.
__device__ __forceinline__ void final_exp_coop(FinalExpLds& L) {
    CoopScratch64& sc = L.sc;
    coop_conj(L.t, L.f);
    if (threadIdx.x == 0) {  // the one inversion: ~40 Fq2 products and an Fq inversion, on one lane
        T6 a = L.f, b;
        t6_inv(b, a);
        L.u = b;
    }
    __syncthreads();
    coop_mul64(L.g, L.t, L.u, sc);  // f^(p^6 - 1)
    coop_frob(L.t, L.g, 2);
    coop_mul64(L.g, L.t, L.g, sc);  // ^(p^2 + 1)
    // hard part: arkworks' chain (ark-ec 0.4 models/bn: Fuentes-Castaneda et al., "Faster hashing to G2"), y_k as named there
    coop_pow_u(L.fx, L.g, sc, L.cs);
    coop_conj(L.y0, L.fx);                 // y0 = r^-u
    coop_cyclo_sqr64(L.y1, L.y0, L.cs);    // y1 = y0^2
    coop_cyclo_sqr64(L.y2, L.y1, L.cs);    // y2 = y1^2
    coop_mul64(L.y3, L.y2, L.y1, sc);      // y3 = y2 y1
    coop_pow_u(L.fx2, L.y3, sc, L.cs);
    coop_conj(L.y4, L.fx2);                // y4 = y3^-u
    coop_cyclo_sqr64(L.y5, L.y4, L.cs);    // y5 = y4^2
    coop_pow_u(L.y6, L.y5, sc, L.cs);      // conj(y6) = y5^u
    coop_conj(L.t, L.y3);                  // conj(y3)
    coop_mul64(L.t0, L.y6, L.y4, sc);      // y7 = conj(y6) y4
    coop_mul64(L.t0, L.t0, L.t, sc);       // y8 = y7 conj(y3)
    coop_mul64(L.t1, L.t0, L.y1, sc);      // y9 = y8 y1
    coop_mul64(L.fx3, L.t0, L.y4, sc);     // y10 = y8 y4
    coop_mul64(L.fx3, L.fx3, L.g, sc);     // y11 = y10 r
    coop_frob(L.t, L.t1, 1);               // y12 = y9^p
    coop_mul64(L.fx3, L.t, L.fx3, sc);     // y13 = y12 y11
    coop_frob(L.t, L.t0, 2);               // y8^(p^2)
    coop_mul64(L.fx3, L.t, L.fx3, sc);     // y14
    coop_conj(L.t, L.g);
    coop_mul64(L.t1, L.t, L.t1, sc);       // y15 = conj(r) y9
    coop_frob(L.t, L.t1, 3);
    coop_mul64(L.f, L.t, L.fx3, sc);       // y16 = y15^(p^3) y14
}

// ---- Miller loop: one WAVE per pair.  f lives in LDS and its products are spread over the lanes like in the final
// exponentiation; the twist point T = (X, Y, Z) is homogeneous projective and advanced without any inversion, the Fq2 products of
// one dependency level of a step on as many lanes (four levels for a tangent, five for a chord).
// Lines are scaled by Fq2 factors (2 Y Z^2 for a tangent, x_Q Z - X for a chord) that the final exponentiation removes:
//   tangent: N = 3 X^2, D = 2 Y Z:  l = D Z y_P - N Z x_P w + (N X - D Y) w^3
//            W = N^2 Z - 2 X D^2:   T <- (D W, N (X D^2 - W) - Y D^3, D^3 Z)
//   chord:   N = y_Q Z - Y, D = x_Q Z - X:  l = D y_P - N x_P w + (N x_Q - D y_Q) w^3
//            E = D^2 Z, W = N^2 Z - X D^2 - x_Q E:  T <- (D W, N (x_Q E - W) - y_Q D^3 Z, D^3 Z)
// (tools/pairing_prototype.py::miller_proj is the same algorithm in Python, checked against the oracle's affine loop.)
// The point arithmetic of a step is a short program of Fq2 products whose operands and results sit in LDS (PointLds::v);
// the products of one dependency level run on as many lanes, the additions between levels on lane 0.
enum PV : uint8_t {
    vX, vY, vZ, vXQ, vYQ,           // T = (X, Y, Z); the affine point of a chord step
    vN, vD, vD2, vN2, vD3, vXD2, vNZ, vNX, vDZ, vDY, vN2Z, vW, vU, vT0, vT1, vT2, vT3, vE, vXQE, vZ3,
    vL0, vL1, vL3,                  // the line: coefficients of w^0, w^1, w^3
    PV_COUNT
};
struct PointLds {
    Fq2 v[PV_COUNT];
    Fq xpn, yp;                     // -x_P, y_P (Montgomery)
};
struct POp {
    uint8_t out, a, b;              // v[out] = v[a] * v[b];  b == 0xFE: v[a] * yp, b == 0xFF: v[a] * xpn (Fq scalars)
};
struct LineLds {
    Fq2 prod[18];
    Fq2 dsum[9];
};
struct MillerLds {
    T6 f;
    CoopScratch sc;
    LineLds line;
    PointLds pt;
    Fq2 q[6];  // Q, pi(Q), -pi^2(Q): (x, y) each
};
constexpr uint32_t PL = 36;  // first PRODUCT SLOT of the point arithmetic (slots 0..35 belong to the products of f); slot k is
                             // worked by the three threads 3 k .. 3 k + 2 (Karatsuba parts); thread PL also does the point's additions

// the Fq2 products of lanes PL.. for one level of a point step
template <int M>
__device__ __forceinline__ void level_ops(PointLds& P, const POp (&ops)[M], const Fq2*& pa, const Fq2*& pb, Fq2*& po, int& scale) {
    const uint32_t t = threadIdx.x / 3;
#pragma unroll
    for (int k = 0; k < M; k++)
        if (t == PL + (uint32_t)k) {
            po = &P.v[ops[k].out];
            pa = &P.v[ops[k].a];
            if (ops[k].b >= 0xFE) scale = ops[k].b == 0xFE ? 1 : 2;
            else pb = &P.v[ops[k].b];
        }
}
// ONE call site of the base-field product for every thread that has a part in this phase (operands through per-thread pointers),
// so that products of f and products of the point step issue together; then two threads per slot assemble the components
__device__ __forceinline__ void phase(const PointLds& P, Fq* part_buf, const Fq2* pa, const Fq2* pb, Fq2* po, int scale) {
    const uint32_t t = threadIdx.x, part = t % 3;
    if (po) {
        // Fq2 times an Fq scalar s = the parts (a0 s, a1 s, unused) of a product with b = (s, s): one call site for both kinds
        Fq2 bb;
        uint32_t pt = part;
        if (scale) {
            const Fq& sv = scale == 1 ? P.yp : P.xpn;
            bb = Fq2{sv, sv};
            if (part == 2) pt = 0;
        } else {
            bb = *pb;
        }
        karatsuba_part(part_buf[t], *pa, bb, pt);
    }
    __syncthreads();
    if (po && part < 2) {
        const Fq v = scale ? part_buf[t] : karatsuba_join(&part_buf[t - part], part);
        (part ? po->c1 : po->c0) = v;
    }
    __syncthreads();
}
// sums of the 18 line products into the 9 anti-diagonals (lanes 0..8)
__device__ __forceinline__ void line_sums(LineLds& L) {
    const uint32_t t = threadIdx.x;
    if (t < 9) {
        Fq2 d = f2_zero();
        for (int i = 0; i < 6; i++) {
            const int off = (int)t - i;  // exponent of the line term: 0, 1 or 3
            if (off == 0 || off == 1) d = fq::add(d, L.prod[3 * i + off]);
            else if (off == 3) d = fq::add(d, L.prod[3 * i + 2]);
        }
        L.dsum[t] = d;
    }
}
__device__ __forceinline__ void line_fold(T6& f, const LineLds& L) {
    const uint32_t t = threadIdx.x;
    if (t < 6) {
        if (t < 3) {
            Fq2 x;
            f2_mul_xi(x, L.dsum[t + 6]);
            f.c[t] = fq::add(L.dsum[t], x);
        } else {
            f.c[t] = L.dsum[t];
        }
    }
}

// tangent step, fused: f <- f^2 * l_{T,T}(P), T <- 2 T in FOUR product phases
//   1: the 36 products of f^2 | X^2, Y Z          2: N^2, D^2, N Z, N X, D Z, D Y (f's anti-diagonals are summed and folded beside)
//   3: D^3, X D^2, N^2 Z, l0, l1                  4: the 18 products f * line | D W, N U, Y D^3, D^3 Z
__device__ __forceinline__ void step_double(MillerLds& S) {
    const uint32_t t = threadIdx.x, slot = t / 3;
    PointLds& P = S.pt;
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        if (slot < 36) {
            pa = &S.f.c[slot / 6];
            pb = &S.f.c[slot % 6];
            po = &S.sc.prod[slot];
        }
        constexpr POp L1[2] = {{vT0, vX, vX}, {vT1, vY, vZ}};
        level_ops(P, L1, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    if (t < 11) {
        Fq2 d = f2_zero();
        for (int i = 0; i < 6; i++) {
            const int j = (int)t - i;
            if (j >= 0 && j < 6) d = fq::add(d, S.sc.prod[i * 6 + j]);
        }
        S.sc.dsum[t] = d;
    } else if (t == PL) {
        P.v[vN] = fq::add(f2_dbl(P.v[vT0]), P.v[vT0]);  // N = 3 X^2
        P.v[vD] = f2_dbl(P.v[vT1]);                     // D = 2 Y Z
    }
    __syncthreads();
    {
        if (t < 6) {  // fold w^6 = xi into f (additions only)
            if (t < 5) {
                Fq2 x;
                f2_mul_xi(x, S.sc.dsum[t + 6]);
                S.f.c[t] = fq::add(S.sc.dsum[t], x);
            } else {
                S.f.c[5] = S.sc.dsum[5];
            }
        }
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        constexpr POp L2[6] = {{vD2, vD, vD}, {vN2, vN, vN}, {vNZ, vN, vZ}, {vNX, vN, vX}, {vDZ, vD, vZ}, {vDY, vD, vY}};
        level_ops(P, L2, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        constexpr POp L3[5] = {{vD3, vD2, vD}, {vXD2, vX, vD2}, {vN2Z, vN2, vZ}, {vL0, vDZ, 0xFE}, {vL1, vNZ, 0xFF}};
        level_ops(P, L3, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    if (t == PL) {
        P.v[vL3] = fq::sub(P.v[vNX], P.v[vDY]);
        P.v[vW] = fq::sub(P.v[vN2Z], f2_dbl(P.v[vXD2]));
        P.v[vU] = fq::sub(P.v[vXD2], P.v[vW]);
    }
    __syncthreads();
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        if (slot < 18) {
            pa = &S.f.c[slot / 3];
            pb = &P.v[vL0 + slot % 3];
            po = &S.line.prod[slot];
        }
        constexpr POp L4[4] = {{vT0, vD, vW}, {vT1, vN, vU}, {vT2, vY, vD3}, {vT3, vD3, vZ}};
        level_ops(P, L4, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    line_sums(S.line);
    if (t == PL) {
        P.v[vX] = P.v[vT0];
        P.v[vY] = fq::sub(P.v[vT1], P.v[vT2]);
        P.v[vZ] = P.v[vT3];
    }
    __syncthreads();
    line_fold(S.f, S.line);
    __syncthreads();
}
// chord step through the affine point (vXQ, vYQ), fused: f <- f * l_{T,Q}(P), T <- T + Q (unless it is the last line)
//   1: y_Q Z, x_Q Z        2: D^2, N^2, N x_Q, D y_Q, l0, l1        3: the 18 products f * line | D^3, D^2 Z, N^2 Z, X D^2
//   4: x_Q E, D^3 Z        5: D W, N U, y_Q Z3
__device__ __forceinline__ void step_add(MillerLds& S, bool advance) {
    const uint32_t t = threadIdx.x, slot = t / 3;
    PointLds& P = S.pt;
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        constexpr POp L1[2] = {{vT0, vYQ, vZ}, {vT1, vXQ, vZ}};
        level_ops(P, L1, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    if (t == PL) {
        P.v[vN] = fq::sub(P.v[vT0], P.v[vY]);  // N = y_Q Z - Y
        P.v[vD] = fq::sub(P.v[vT1], P.v[vX]);  // D = x_Q Z - X
    }
    __syncthreads();
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        constexpr POp L2[6] = {{vD2, vD, vD}, {vN2, vN, vN}, {vNX, vN, vXQ}, {vDY, vD, vYQ}, {vL0, vD, 0xFE}, {vL1, vN, 0xFF}};
        level_ops(P, L2, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    if (t == PL) P.v[vL3] = fq::sub(P.v[vNX], P.v[vDY]);
    __syncthreads();
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        if (slot < 18) {
            pa = &S.f.c[slot / 3];
            pb = &P.v[vL0 + slot % 3];
            po = &S.line.prod[slot];
        }
        if (advance) {
            constexpr POp L3[4] = {{vD3, vD2, vD}, {vE, vD2, vZ}, {vN2Z, vN2, vZ}, {vXD2, vX, vD2}};
            level_ops(P, L3, pa, pb, po, sc);
        }
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    line_sums(S.line);
    __syncthreads();
    line_fold(S.f, S.line);
    if (!advance) {
        __syncthreads();
        return;
    }
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        constexpr POp L4[2] = {{vXQE, vXQ, vE}, {vZ3, vD3, vZ}};
        level_ops(P, L4, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    if (t == PL) {
        P.v[vW] = fq::sub(fq::sub(P.v[vN2Z], P.v[vXD2]), P.v[vXQE]);
        P.v[vU] = fq::sub(P.v[vXQE], P.v[vW]);
    }
    __syncthreads();
    {
        const Fq2 *pa = nullptr, *pb = nullptr;
        Fq2* po = nullptr;
        int sc = 0;
        constexpr POp L5[3] = {{vT0, vD, vW}, {vT1, vN, vU}, {vT2, vYQ, vZ3}};
        level_ops(P, L5, pa, pb, po, sc);
        phase(P, S.sc.part, pa, pb, po, sc);
    }
    if (t == PL) {
        P.v[vX] = P.v[vT0];
        P.v[vY] = fq::sub(P.v[vT1], P.v[vT2]);
        P.v[vZ] = P.v[vZ3];
    }
    __syncthreads();
}

__device__ __forceinline__ Fq load_fq(const uint32_t* w) {
    Fq r;
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
    return fq::to_mont(r);
}

// one wave per pair; g1 [n][16] = (x, y), g2 [n][32] = (x.c0, x.c1, y.c0, y.c1) (reference src/transcript_native.rs:42-54 order);
// all-zero coordinates stand for the point at infinity (Miller value 1)
__global__ void __launch_bounds__(NT) miller_kernel(const uint32_t* __restrict__ g1, const uint32_t* __restrict__ g2, uint32_t n,
                                                    T6* __restrict__ out) {
    __shared__ MillerLds S;
    const uint32_t i = blockIdx.x, t = threadIdx.x;
    const uint32_t* pw = g1 + (size_t)i * 16;
    const uint32_t* qw = g2 + (size_t)i * 32;
    if (t < 6) S.f.c[t] = t == 0 ? f2_one() : f2_zero();
    uint32_t nzp = 0, nzq = 0;  // wave-uniform
    for (int k = 0; k < 16; k++) nzp |= pw[k];
    for (int k = 0; k < 32; k++) nzq |= qw[k];
    if (nzp != 0 && nzq != 0) {
        if (t == 0) {
            S.pt.xpn = fq::neg(load_fq(pw));
            S.pt.yp = load_fq(pw + 8);
        }
        if (t < 2) {  // Q
            S.q[t] = Fq2{load_fq(qw + 16 * t), load_fq(qw + 16 * t + 8)};
        }
        __syncthreads();
        if (t < 2) {  // Q1 = pi(Q): pi(x, y) = (conj(x) gamma_1,2, conj(y) gamma_1,3)
            f2_mul(S.q[2 + t], f2_conj(S.q[t]), pairing_k::GAMMA[0][2 + t]);
        }
        __syncthreads();
        if (t < 2) {  // Q2 = -pi^2(Q)
            Fq2 r;
            f2_mul(r, f2_conj(S.q[2 + t]), pairing_k::GAMMA[0][2 + t]);
            S.q[4 + t] = t ? f2_neg(r) : r;
        }
        if (t == 0) {
            S.pt.v[vX] = S.q[0];
            S.pt.v[vY] = S.q[1];
            S.pt.v[vZ] = f2_one();
            S.pt.v[vXQ] = S.q[0];
            S.pt.v[vYQ] = S.q[1];
        }
        __syncthreads();
        for (int b = pairing_k::ATE_BITS - 2; b >= 0; b--) {
            step_double(S);
            const uint32_t bit = b >= 64 ? (pairing_k::ATE_HI >> (b - 64)) & 1u : (uint32_t)(pairing_k::ATE_LO >> b) & 1u;
            if (bit) step_add(S, true);
        }
        if (t < 2) S.pt.v[vXQ + t] = S.q[2 + t];
        __syncthreads();
        step_add(S, true);
        if (t < 2) S.pt.v[vXQ + t] = S.q[4 + t];
        __syncthreads();
        step_add(S, false);
    }
    __syncthreads();
    if (t < 6) out[i].c[t] = S.f.c[t];
}

// one workgroup per product (blockIdx.x = k): the product of the Miller values [off[k], off[k + 1]) is left in vals[off[k]]
__global__ void __launch_bounds__(256) product_kernel(T6* __restrict__ vals, const uint32_t* __restrict__ off) {
    T6* v = vals + off[blockIdx.x];
    const uint32_t n = off[blockIdx.x + 1] - off[blockIdx.x];
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
}

// one wave per product: final exponentiation of vals[off[k]] (the product left there by product_kernel), MyFq12 coefficients out
__global__ void __launch_bounds__(64) final_exp_kernel(const T6* __restrict__ vals, const uint32_t* __restrict__ off,
                                                       uint32_t* __restrict__ out) {
    __shared__ FinalExpLds L;
    const uint32_t l = threadIdx.x;
    if (l < 6) L.f.c[l] = vals[off[blockIdx.x]].c[l];
    __syncthreads();
    final_exp_coop(L);
    if (l < 6) {
        // MyFq12 coefficients c_i = a_i - 9 b_i, c_{i+6} = b_i of the Fq2 coefficient a_i + b_i u of w^i
        uint32_t* o = out + (size_t)blockIdx.x * 96;
        const Fq nine = fq::small_m(9);
        Fq nb;
        fq_mul(nb, nine, L.f.c[l].c1);
        const Fq lo = fq::from_mont(fq::sub(L.f.c[l].c0, nb)), hi = fq::from_mont(L.f.c[l].c1);
        for (int k = 0; k < 8; k++) {
            o[8 * l + k] = lo.l[k];
            o[8 * (l + 6) + k] = hi.l[k];
        }
    }
}


// =====================================================================================================================================
// Primary witness of the final-pairing AIR (API kind 6; tools/air_gen.py::build_pairing, NARROW layout: one modular identity per row,
// 2^13 rows per pairing, row program tools/pairing_rows.py) -- the statement behind `pairing_circuit(final_A, final_B)` == final_Z of
// the reference's BLS example (src/bin/bls_aggregation.rs:76-77).  Two kernels:
//   pairing_values_kernel   ONE WAVE per record walks the OPERATION schedule (tools/pairing_sched.py: 458 operations): the G2 unit (an
//                           AFFINE tangent / chord step with its line coefficients on lane 0; the 102 slopes come from a projective pass
//                           over the point chain and ONE batched inversion in front), the Fq12 unit (36 lanes per product as in the
//                           final exponentiation above; the one inversion on lane 0; Frobenius / conjugation on 6 lanes).  Every value a
//                           row of the trace produces is LOGGED into the record's pool at the index of that row (pairing_rowsrc.h).
//   pairing_expand_kernel   one thread per (row, field element): looks the element's pool index up in the host-built source table (the
//                           row program replayed symbolically: pairing_rowsrc.h) and writes its 16 limb cells (32 for the checked
//                           result) -- consecutive lanes are consecutive rows of one column.
// oracle/pairing.c is the CPU reading the tests compare with cell for cell.
// =====================================================================================================================================
enum { PV_QX, PV_QY, PV_FXC, PV_FYC, PV_N };
struct PairVals {
    T6 reg[6];
    T6 A, B, G, C;
    Fq px, py;
    Fq2 pts[PV_N];
};
constexpr int PAIR_STEPS = 102;      // 64 tangent + 36 + 2 chord steps of the schedule
// device copy of the schedule tables (data/air_tables.h): [512][6] int8 | gconj [8] int8 | gconst [6][192] u16 | oprow [512] i16 |
// steprow [104] i16 | row constants [NGC][32] u16
constexpr int PT_SCHED = 0, PT_GCONJ = 3072, PT_GCONST = 3080, PT_OPROW = PT_GCONST + 6 * 192 * 2, PT_STEPROW = PT_OPROW + 1024,
              PT_GC = PT_STEPROW + 208, PT_BYTES = PT_GC + AIR_PAIRING_NGC * 64;
static_assert(PT_GCONST % 2 == 0 && PT_OPROW % 2 == 0 && PT_GC % 2 == 0, "16-bit tables are aligned");

__device__ __forceinline__ Fq load_words_mont(const uint32_t* w) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
    return fq::to_mont(r);
}
__device__ __forceinline__ void pool_put(uint32_t* __restrict__ pool, int idx, const Fq& mont) {
    const Fq v = fq::from_mont(mont);
    uint4* o = reinterpret_cast<uint4*>(pool + (size_t)idx * 8);
    o[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    o[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// ---- the point chain AHEAD of the walk (pairing_values_kernel): scratch of one level of independent Fq2 products ----
constexpr int PAIR_NS = 104;         // PAIR_STEPS rounded up
struct StepLevel {
    Fq2 a[4], b[4], o[4];
    Fq part[12];
};
// o[k] = a[k] * b[k], k < cnt <= 4: lanes 3 k .. 3 k + 2 take the three Karatsuba parts, two of them join.  Operands written by lane 0
// before the call; every lane of the block calls (barriers inside)
__device__ __forceinline__ void level_mul(StepLevel& W, uint32_t l, uint32_t cnt) {
    __syncthreads();
    const uint32_t k = l / 3, part = l % 3;
    if (k < cnt) karatsuba_part(W.part[l], W.a[k], W.b[k], part);
    __syncthreads();
    if (k < cnt && part < 2) {
        const Fq v = karatsuba_join(&W.part[3 * k], part);
        (part ? W.o[k].c1 : W.o[k].c0) = v;
    }
    __syncthreads();
}
// inclusive products along an array of n <= 128 entries by 64 lanes (Hillis-Steele; seven levels of two products per lane):
// forward: out[i] = in[0] ... in[i]; backward: out[i] = in[i] ... in[n - 1].  Returns the buffer that holds the result (b0 or b1; b0 = input)
__device__ __forceinline__ Fq2* scan_products(Fq2* b0, Fq2* b1, uint32_t n, bool backward, uint32_t l) {
    Fq2 *src = b0, *dst = b1;
    for (uint32_t d = 1; d < n; d <<= 1) {
        for (uint32_t i = l; i < n; i += 64) {
            const bool has = backward ? i + d < n : i >= d;
            if (has) f2_mul(dst[i], src[i], src[backward ? i + d : i - d]);
            else dst[i] = src[i];
        }
        __syncthreads();
        Fq2* t = src; src = dst; dst = t;
    }
    return src;
}

// pools: [num_io][PP_N][8] u32, zeroed by the caller (rows without a result read zero); mode 1: values only, Z written into the record
__global__ void __launch_bounds__(64) pairing_values_kernel(uint32_t* __restrict__ ios, uint32_t num_io, const uint8_t* __restrict__ tab,
                                                           uint32_t* __restrict__ pools, int mode, int* __restrict__ err) {
    __shared__ PairVals V;
    __shared__ CoopScratch64 sc;
    __shared__ int s_bad;
    // the 102 point steps' values, produced AHEAD of the walk (see below): slope, x3, y3, -lam x_P, lam x_T - y_T; 8 x 6.5 KB
    __shared__ Fq2 a_lam[PAIR_NS], a_den[PAIR_NS], a_x[PAIR_NS], a_y[PAIR_NS], a_p0[PAIR_NS], a_p1[PAIR_NS], a_q0[PAIR_NS], a_q1[PAIR_NS];
    __shared__ Fq2 s_frob[4], s_inv;
    static_assert(sizeof(StepLevel) <= sizeof(CoopScratch64), "the level scratch lives in the product scratch");
    const uint32_t l = threadIdx.x;
    const uint32_t io = blockIdx.x;
    if (io >= num_io) return;
    uint32_t* rec = ios + (size_t)io * 144;
    uint32_t* pool = pools + (size_t)io * PP_N * 8;
    const Fq2 zero2 = f2_zero();
    // ---- the record: P, Q (checked: on their curves; the verifier also wants [r] Q = O), constants, empty registers ----
    if (l == 0) {
        s_bad = 0;
        V.px = load_words_mont(rec);
        V.py = load_words_mont(rec + 8);
        V.pts[PV_QX] = Fq2{load_words_mont(rec + 16), load_words_mont(rec + 24)};
        V.pts[PV_QY] = Fq2{load_words_mont(rec + 32), load_words_mont(rec + 40)};
        V.pts[PV_FXC] = pairing_k::GAMMA[0][2];
        V.pts[PV_FYC] = pairing_k::GAMMA[0][3];
        // y^2 = x^3 + 3 ;  (9 + u)(y^2 - x^3) = 3
        const Fq three = fq::small_m(3);
        Fq t;
        fq_mul(t, V.px, V.px);
        fq_mul(t, t, V.px);
        Fq y2;
        fq_mul(y2, V.py, V.py);
        if (!fq::is_zero(fq::sub(fq::sub(y2, t), three))) s_bad = 1;
        Fq2 x3, yy, d;
        f2_sqr(x3, V.pts[PV_QX]);
        f2_mul(x3, x3, V.pts[PV_QX]);
        f2_sqr(yy, V.pts[PV_QY]);
        f2_mul_xi(d, fq::sub(yy, x3));
        if (!fq::is_zero(fq::sub(d.c0, three)) || !fq::is_zero(d.c1)) s_bad = 1;
    }
    if (l < 36) V.reg[l / 6].c[l % 6] = zero2;
    __syncthreads();
    if (s_bad) {
        if (l == 0) atomicExch(err, SIPP_E_WITNESS);
        return;
    }
    const int8_t* sched = reinterpret_cast<const int8_t*>(tab + PT_SCHED);
    const int8_t* gconj = reinterpret_cast<const int8_t*>(tab + PT_GCONJ);
    const uint16_t* gconst = reinterpret_cast<const uint16_t*>(tab + PT_GCONST);
    const int16_t* oprow = reinterpret_cast<const int16_t*>(tab + PT_OPROW);
    const int16_t* steprow = reinterpret_cast<const int16_t*>(tab + PT_STEPROW);
    // ---- the pool's fixed part: P, Q (the record's words as they are), the twist's Frobenius constants, the row constants ----
    if (mode == 0) {
        if (l < 6) {
            const uint32_t* w = rec + 8 * l;
            for (int i = 0; i < 8; i++) pool[(size_t)(PP_PX + l) * 8 + i] = w[i];
        } else if (l < 10) {
            const Fq2& c = V.pts[l < 8 ? PV_FXC : PV_FYC];
            pool_put(pool, PP_FROBC + (l - 6), (l & 1) ? c.c1 : c.c0);
        }
        const uint16_t* gcs = reinterpret_cast<const uint16_t*>(tab + PT_GC);
        for (int e = l; e < 2 * AIR_PAIRING_NGC; e += 64)
            for (int i = 0; i < 8; i++) pool[(size_t)(PP_GC + e) * 8 + i] = (uint32_t)gcs[16 * e + 2 * i] | ((uint32_t)gcs[16 * e + 2 * i + 1] << 16);
    }
    // ---- the 102 point steps AHEAD of the walk.  (a) The point chain in homogeneous coordinates, no inversion: slope of step s =
    // N_s / D_s with N = 3 X^2, D = 2 Y Z for a tangent, N = y_Q Z - Y, D = x_Q Z - X for a chord; the independent Fq2 products of a
    // step run level by level on 3 lanes each (4 levels for a tangent, 5 for a chord).  (b) Z_0 = 1 and Z' = D^3 Z: every 1 / Z_s is
    // the cube of 1 / (D_0 ... D_s), so ONE inversion (of D_0 ... D_101) and two product scans give every 1 / D_s and 1 / Z_s.  (c) One
    // lane per step: slope, affine (x3, y3) = (X', Y') / Z', the line coefficients.  On one lane with an inversion per step this was
    // 10 of the kernel's 18.5 ms, with the batched inversion alone (round 6a) 5.6 of 8.6 ms; now 0.9.
    {
        StepLevel& W = *reinterpret_cast<StepLevel*>(&sc);
        Fq2 X, Y, Z, q1x, q1y, q2x, q2y;
        const Fq2 qx = V.pts[PV_QX], qy = V.pts[PV_QY];
        if (l == 0) {
            W.a[0] = f2_conj(qx); W.b[0] = V.pts[PV_FXC];
            W.a[1] = f2_conj(qy); W.b[1] = V.pts[PV_FYC];
        }
        level_mul(W, l, 2);
        if (l == 0) {
            q1x = W.o[0]; q1y = W.o[1];
            W.a[0] = f2_conj(q1x); W.a[1] = f2_conj(q1y);
        }
        level_mul(W, l, 2);
        if (l == 0) {
            q2x = W.o[0]; q2y = f2_neg(W.o[1]);
            s_frob[0] = q1x; s_frob[1] = q1y; s_frob[2] = q2x; s_frob[3] = q2y;
            X = qx; Y = qy; Z = f2_one();
        }
        int ns = 0;
        for (int t = 1; t < AIR_PAIRING_OPS && ns < PAIR_STEPS; t++) {
            const int gop = sched[6 * t + 5];
            if (gop < 1 || gop > 4) continue;
            Fq2 N, D, D2, D3, U, Wv;
            if (gop == 1) {
                if (l == 0) { W.a[0] = X; W.b[0] = X; W.a[1] = Y; W.b[1] = Z; }
                level_mul(W, l, 2);
                if (l == 0) {
                    N = fq::add(f2_dbl(W.o[0]), W.o[0]);
                    D = f2_dbl(W.o[1]);
                    if (fq::is_zero(D)) { s_bad = 1; D = f2_one(); }      // a degenerate step (Q outside the r-torsion): no slope
                    W.a[0] = D; W.b[0] = D; W.a[1] = N; W.b[1] = N;
                }
                level_mul(W, l, 2);
                if (l == 0) {
                    D2 = W.o[0];
                    W.a[0] = D2; W.b[0] = D; W.a[1] = X; W.b[1] = D2; W.a[2] = W.o[1]; W.b[2] = Z;
                }
                level_mul(W, l, 3);
                if (l == 0) {
                    D3 = W.o[0];
                    Wv = fq::sub(W.o[2], f2_dbl(W.o[1]));
                    U = fq::sub(W.o[1], Wv);
                    W.a[0] = N; W.b[0] = U; W.a[1] = Y; W.b[1] = D3; W.a[2] = D3; W.b[2] = Z; W.a[3] = D; W.b[3] = Wv;
                }
                level_mul(W, l, 4);
                if (l == 0) {
                    Y = fq::sub(W.o[0], W.o[1]);
                    Z = W.o[2];
                    X = W.o[3];
                }
            } else {
                Fq2 xq, yq;
                if (l == 0) {
                    xq = gop == 2 ? qx : gop == 3 ? q1x : q2x;
                    yq = gop == 2 ? qy : gop == 3 ? q1y : q2y;
                    W.a[0] = yq; W.b[0] = Z; W.a[1] = xq; W.b[1] = Z;
                }
                level_mul(W, l, 2);
                if (l == 0) {
                    N = fq::sub(W.o[0], Y);
                    D = fq::sub(W.o[1], X);
                    if (fq::is_zero(D)) { s_bad = 1; D = f2_one(); }
                    W.a[0] = D; W.b[0] = D; W.a[1] = N; W.b[1] = N;
                }
                level_mul(W, l, 2);
                if (l == 0) {
                    D2 = W.o[0];
                    W.a[0] = D2; W.b[0] = D; W.a[1] = D2; W.b[1] = Z; W.a[2] = W.o[1]; W.b[2] = Z; W.a[3] = X; W.b[3] = D2;
                }
                level_mul(W, l, 4);
                if (l == 0) {
                    D3 = W.o[0];
                    U = W.o[2];                      // N^2 Z
                    Wv = W.o[3];                     // X D^2
                    W.a[0] = xq; W.b[0] = W.o[1]; W.a[1] = D3; W.b[1] = Z;
                }
                level_mul(W, l, 2);
                if (l == 0) {
                    const Fq2 xqE = W.o[0];
                    Z = W.o[1];
                    Wv = fq::sub(fq::sub(U, Wv), xqE);
                    W.a[0] = N; W.b[0] = fq::sub(xqE, Wv); W.a[1] = yq; W.b[1] = Z; W.a[2] = D; W.b[2] = Wv;
                }
                level_mul(W, l, 3);
                if (l == 0) {
                    Y = fq::sub(W.o[0], W.o[1]);
                    X = W.o[2];
                }
            }
            if (l == 0) {
                a_lam[ns] = N; a_den[ns] = D; a_x[ns] = X; a_y[ns] = Y;
                a_p0[ns] = D; a_q0[ns] = D;
            }
            ns++;
        }
        __syncthreads();
        // (b) prefix and suffix products of the denominators, one inversion
        const Fq2* pre = scan_products(a_p0, a_p1, (uint32_t)ns, false, l);
        const Fq2* suf = scan_products(a_q0, a_q1, (uint32_t)ns, true, l);
        if (l == 0) f2_inv(s_inv, pre[ns - 1]);
        __syncthreads();
        // (c) one lane per step: 1 / (D_0 ... D_s) = inv suf[s + 1]; 1 / D_s = that times D_0 ... D_(s-1); 1 / Z'_s = its cube
        Fq2 lam[2], x3[2], y3[2];
        for (int r = 0; r < 2; r++) {
            const int sidx = (int)l + 64 * r;
            if (sidx < ns) {
                Fq2 ip = s_inv, dinv, t2;
                if (sidx + 1 < ns) f2_mul(ip, s_inv, suf[sidx + 1]);
                dinv = ip;
                if (sidx > 0) f2_mul(dinv, ip, pre[sidx - 1]);
                f2_mul(lam[r], a_lam[sidx], dinv);
                f2_sqr(t2, ip);
                f2_mul(t2, t2, ip);
                f2_mul(x3[r], a_x[sidx], t2);
                f2_mul(y3[r], a_y[sidx], t2);
            }
        }
        __syncthreads();
        for (int r = 0; r < 2; r++) {
            const int sidx = (int)l + 64 * r;
            if (sidx < ns) { a_lam[sidx] = lam[r]; a_x[sidx] = x3[r]; a_y[sidx] = y3[r]; }
        }
        __syncthreads();
        for (int r = 0; r < 2; r++) {
            const int sidx = (int)l + 64 * r;
            if (sidx < ns) {
                const Fq2 tx = sidx ? a_x[sidx - 1] : qx, ty = sidx ? a_y[sidx - 1] : qy;
                Fq2 s3, t2;
                f2_scale(s3, lam[r], V.px);
                f2_mul(t2, lam[r], tx);
                a_p0[sidx] = f2_neg(s3);                 // -lam x_P
                a_p1[sidx] = fq::sub(t2, ty);            // lam x_T - y_T
            }
        }
        __syncthreads();
    }
    int step = 0;
    for (int t = 0; t < AIR_PAIRING_OPS; t++) {
        const int fop = sched[6 * t], ra = sched[6 * t + 1], rb = sched[6 * t + 2], gc = sched[6 * t + 3], rd = sched[6 * t + 4], gop = sched[6 * t + 5];
        const int gi = gc < 0 ? 0 : gc;
        if (fop == 0 && gop == 0 && rd < 0) continue;      // an idle operation (uniform over the wave)
        const int st = step < PAIR_STEPS ? step : PAIR_STEPS - 1;
        // ---- (1) the point step's five values (from the tables above) into the pool, at the rows that produce them: slope, x3, y3,
        // -lam x_P, lam x_T - y_T; the Frobenius images of Q on the first operation; the operation's constants (lanes 32 .. 43) ----
        if (mode == 0 && gop != 0 && l < 10) {
            if (gop == 5) {
                if (l < 8) pool_put(pool, PP_ROW + (int)l, (l & 1) ? s_frob[l >> 1].c1 : s_frob[l >> 1].c0);
            } else {
                const Fq2& v = l < 2 ? a_lam[st] : l < 4 ? a_x[st] : l < 6 ? a_y[st] : l < 8 ? a_p0[st] : a_p1[st];
                pool_put(pool, PP_ROW + steprow[st] + (int)(l < 2 ? l : l + 2), (l & 1) ? v.c1 : v.c0);
            }
        }
        if (l >= 32 && l < 44 && (fop == 4 || fop == 0)) {
            const uint32_t e = l - 32;          // Fq element e of the operation's constant vector (16-bit limbs in the table)
            const uint16_t* src = gconst + (size_t)gi * 192 + 16 * e;
            Fq v;
#pragma unroll
            for (int i = 0; i < 8; i++) v.l[i] = (uint32_t)src[2 * i] | ((uint32_t)src[2 * i + 1] << 16);
            Fq* g = reinterpret_cast<Fq*>(&V.G);
            g[e] = fq::to_mont(v);
        }
        // ---- (2) operands ----
        if (l < 6) {
            V.A.c[l] = ra >= 0 ? V.reg[ra].c[l] : zero2;
            Fq2 b = zero2;
            if (fop == 1) b = V.reg[rb].c[l];
            else if (fop == 2) b = l == 0 ? Fq2{V.py, fq::zero()} : l == 1 ? a_p0[st] : l == 3 ? a_p1[st] : zero2;
            V.B.c[l] = b;
        }
        if (gop >= 1 && gop <= 4) step++;
        __syncthreads();
        // ---- (3) the Fq12 unit ----
        if (fop == 1 || fop == 2) {
            coop_mul64(V.C, V.A, V.B, sc);
        } else if (fop == 3) {
            if (l == 0) {
                T6 a = V.A, b;
                bool zero = true;
                for (int i = 0; i < 6; i++) zero = zero && fq::is_zero(a.c[i]);
                if (zero) {
                    s_bad = 1;
                    for (int i = 0; i < 6; i++) b.c[i] = zero2;
                } else {
                    t6_inv(b, a);
                }
                V.C = b;
            }
            __syncthreads();
        } else {
            if (l < 6) {
                Fq2 c = zero2;
                if (fop == 4) f2_mul(c, gconj[gi] ? f2_conj(V.A.c[l]) : V.A.c[l], V.G.c[l]);
                V.C.c[l] = c;
            }
            __syncthreads();
        }
        // ---- (4) the operation's twelve results into the pool; registers ----
        if (mode == 0 && l < 12 && oprow[t] >= 0) pool_put(pool, PP_ROW + oprow[t] + (int)l, reinterpret_cast<const Fq*>(&V.C)[l]);
        if (l < 6 && rd >= 0) V.reg[rd].c[l] = fop == 0 ? V.G.c[l] : V.C.c[l];
        __syncthreads();
    }
    // ---- the result: MyFq12 coefficients c_i = a_i - 9 b_i, c_{i+6} = b_i; compared with the record (mode 0) or written into it ----
    if (l < 6) {
        const T6& res = V.reg[AIR_PAIRING_RESULT_REG];
        const Fq nine = fq::small_m(9);
        Fq nb;
        fq_mul(nb, nine, res.c[l].c1);
        const Fq lo = fq::from_mont(fq::sub(res.c[l].c0, nb)), hi = fq::from_mont(res.c[l].c1);
        uint32_t* z = rec + 48;
        bool bad = s_bad != 0;
        for (int i = 0; i < 8; i++) {
            if (mode == 1) {
                z[8 * l + i] = lo.l[i];
                z[8 * (l + 6) + i] = hi.l[i];
            } else {
                bad |= z[8 * l + i] != lo.l[i] || z[8 * (l + 6) + i] != hi.l[i];
            }
        }
        if (bad) atomicExch(err, SIPP_E_WITNESS);
    }
}

struct PairElemCols {
    int32_t col[PE_N];
    int32_t cpl;
};

// grid (n / 256, PE_N): thread = one row of one field element; src [PE_N][8192] u16; pools [num_io][PP_N][8]
__global__ void __launch_bounds__(256) pairing_expand_kernel(const uint16_t* __restrict__ src, const uint32_t* __restrict__ pools, PairElemCols k,
                                                            uint64_t* __restrict__ tr, size_t n) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t e = blockIdx.y;
    if (row >= n) return;
    const uint32_t r = (uint32_t)row & (AIR_PAIRING_ROWS - 1), io = (uint32_t)(row >> AIR_PAIRING_LOG_ROWS);
    const uint32_t id = src[(size_t)e * AIR_PAIRING_ROWS + r];
    const uint4* p = reinterpret_cast<const uint4*>(pools + ((size_t)io * PP_N + id) * 8);
    const uint4 lo = p[0], hi = p[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const int col = k.col[e];
    if (e == PE_RES && k.cpl == 2) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const uint32_t limb = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
            tr[(size_t)(col + 2 * i) * n + row] = limb & 0xffu;
            tr[(size_t)(col + 2 * i + 1) * n + row] = limb >> 8;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) tr[(size_t)(col + i) * n + row] = (w[i >> 1] >> (16 * (i & 1))) & 0xffffu;
    }
}

}  // namespace

// primary witness of the final-pairing AIR: d_ios [num_io][144] (padded: 8192 num_io == n rows).  outputs_only ctx: the records'
// Z words are written instead (sipp_exp_outputs with kind = SIPP_PAIRING)
int sipp_pairing_fill(sipp_ctx* ctx, const air_spec_t* a, const uint32_t* d_ios, uint32_t num_io, uint32_t log_n, uint64_t* d_trace,
                      int* d_err) {
    const size_t n = (size_t)1 << log_n;
    const bool outputs = ctx->outputs_only;
    const int32_t* lay = (a && a->cells_per_limb == 2) ? AIR_PAIRING_LAYOUT_U8 : AIR_PAIRING_LAYOUT_U16;
    static_assert(AIR_PAIRING_ROWS == 8192 && AIR_PAIRING_OPS == 512 && AIR_PAIRING_NREG == 6 && AIR_PAIRING_NGCONST == 6 && PP_N < 65536,
                  "pairing kernels: table shapes");
    if (!outputs && (!a || a->kind != 6 || a->log_rows != AIR_PAIRING_LOG_ROWS || a->checked_base != lay[20] || a->pi_per_io != 144 ||
                     (size_t)num_io * AIR_PAIRING_ROWS != n))
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "pairing AIR: table / shape mismatch");
    uint64_t* t = sipp_table_get(ctx, 102, 0, 0);
    if (!t) {
        std::vector<uint64_t> packed((PT_BYTES + 7) / 8, 0);
        uint8_t* b = reinterpret_cast<uint8_t*>(packed.data());
        memcpy(b + PT_SCHED, AIR_PAIRING_SCHED, 3072);
        memcpy(b + PT_GCONJ, AIR_PAIRING_GCONJ, AIR_PAIRING_NGCONST);
        uint16_t* gc = reinterpret_cast<uint16_t*>(b + PT_GCONST);
        for (int g = 0; g < AIR_PAIRING_NGCONST; g++)
            for (int i = 0; i < 192; i++) gc[g * 192 + i] = (uint16_t)AIR_PAIRING_GCONST[g][i];
        int16_t steprow[104] = {0};
        if (pairing_log_rows(reinterpret_cast<int16_t*>(b + PT_OPROW), steprow, PAIR_STEPS) != PAIR_STEPS)
            return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "pairing AIR: the row program does not pair up with the operation schedule");
        memcpy(b + PT_STEPROW, steprow, sizeof steprow);
        uint16_t* rc = reinterpret_cast<uint16_t*>(b + PT_GC);
        for (int g = 0; g < AIR_PAIRING_NGC; g++)
            for (int i = 0; i < 32; i++) rc[g * 32 + i] = (uint16_t)AIR_PAIRING_GC[g][i];
        SIPP_TRY(sipp_table_put(ctx, 102, 0, 0, packed, &t));
    }
    uint32_t* d_pools = nullptr;
    uint64_t* src = nullptr;
    if (!outputs) {
        src = sipp_table_get(ctx, 106, 0, 0);
        if (!src) {
            std::vector<uint64_t> packed(((size_t)PE_N * AIR_PAIRING_ROWS * 2 + 7) / 8, 0);
            if (pairing_row_sources(reinterpret_cast<uint16_t*>(packed.data())) != 0)
                return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "pairing AIR: the row program reads what the source replay does not know");
            SIPP_TRY(sipp_table_put(ctx, 106, 0, 0, packed, &src));
        }
        d_pools = arena_alloc_t<uint32_t>(ctx, (size_t)num_io * PP_N * 8);
        if (!d_pools) return sipp_fail(ctx, SIPP_E_NOMEM, "pairing AIR: value pools");
        SIPP_CHECK_HIP(ctx, hipMemsetAsync(d_pools, 0, (size_t)num_io * PP_N * 32, ctx->stream));
    }
    {
        ProfScope ps(ctx, "trace_pairing");
        hipLaunchKernelGGL(pairing_values_kernel, dim3(num_io), dim3(64), 0, ctx->stream, const_cast<uint32_t*>(d_ios), num_io,
                           reinterpret_cast<const uint8_t*>(t), d_pools, outputs ? 1 : 0, d_err);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    if (!outputs) {
        ProfScope ps(ctx, "trace_pairing_cells");
        PairElemCols k;
        for (int e = 0; e < PE_N; e++) k.col[e] = pairing_elem_col(lay, e);
        k.cpl = a->cells_per_limb;
        hipLaunchKernelGGL(pairing_expand_kernel, dim3((unsigned)((n + 255) / 256), PE_N), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const uint16_t*>(src), d_pools, k, d_trace, n);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    return SIPP_OK;
}

// `count` products over consecutive groups of pairs: group k = pairs [off[k], off[k + 1])
int sipp_inner_products_groups(sipp_ctx* ctx, const uint32_t* g1, const uint32_t* g2, const uint32_t* off, size_t count, uint32_t* out) {
    if (!ctx || !g1 || !g2 || !off || !out || count == 0) return SIPP_E_BADARG;
    const size_t total = off[count];
    for (size_t k = 0; k < count; k++)
        if (off[k + 1] <= off[k]) return sipp_fail(ctx, SIPP_E_BADARG, "inner_products: empty group");
    if (off[0] != 0 || total > ((size_t)1 << 24)) return sipp_fail(ctx, SIPP_E_BADARG, "inner_products: bad group offsets");
    SIPP_CHECK_HIP(ctx, hipSetDevice(ctx->device));
    ArenaScope scope(ctx);
    uint32_t* d_g1 = arena_alloc_t<uint32_t>(ctx, total * 16);
    uint32_t* d_g2 = arena_alloc_t<uint32_t>(ctx, total * 32);
    T6* d_f = arena_alloc_t<T6>(ctx, total);
    uint32_t* d_out = arena_alloc_t<uint32_t>(ctx, count * 96);
    uint32_t* d_off = arena_alloc_t<uint32_t>(ctx, count + 1);
    int rc = SIPP_OK;
    if (!d_g1 || !d_g2 || !d_f || !d_out || !d_off) rc = SIPP_E_NOMEM;
    if (rc == SIPP_OK) {
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_g1, g1, total * 64, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_g2, g2, total * 128, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_off, off, (count + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        {
            ProfScope ps(ctx, "pairing_miller");
            hipLaunchKernelGGL(miller_kernel, dim3((unsigned)total), dim3(NT), 0, ctx->stream, d_g1, d_g2, (uint32_t)total, d_f);
        }
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        {
            ProfScope ps(ctx, "pairing_product_final");
            hipLaunchKernelGGL(product_kernel, dim3((unsigned)count), dim3(256), 0, ctx->stream, d_f, d_off);
            hipLaunchKernelGGL(final_exp_kernel, dim3((unsigned)count), dim3(64), 0, ctx->stream, d_f, d_off, d_out);
        }
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(out, d_out, count * 96 * 4, hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return rc;
}

extern "C" int sipp_inner_products(sipp_ctx* ctx, const uint32_t* g1, const uint32_t* g2, size_t n, size_t count, uint32_t* out) {
    if (!ctx || n == 0 || count == 0 || n * count > ((size_t)1 << 24)) return SIPP_E_BADARG;
    std::vector<uint32_t> off(count + 1);
    for (size_t k = 0; k <= count; k++) off[k] = (uint32_t)(k * n);
    return sipp_inner_products_groups(ctx, g1, g2, off.data(), count, out);
}

extern "C" int sipp_inner_product(sipp_ctx* ctx, const uint32_t* g1, const uint32_t* g2, size_t n, uint32_t* out) {
    return sipp_inner_products(ctx, g1, g2, n, 1, out);
}
