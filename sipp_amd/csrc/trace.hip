// sipp_amd/csrc/trace.hip -- trace fill on the device for the three SIPP AIRs (tools/air_gen.py spec).
//
// Replaces starky-bn254's `generate_trace` (@ 2d46f9e, reached through reference
// src/verifier_circuit.rs:133-135).  The caller never materialises the N x W table: IO records go in,
// the column-major trace [W][N] (natural row order) is produced in HBM.
//
// Kernels
//   curve_dbl/scan   one lane per IO for the 255 serial doublings, then one 256-lane workgroup per IO for a parallel
//                    prefix scan of the selected powers (Jacobian, no inversions); stores (R_k, P_k) per row.
//   curve_rows       one lane per row: Jacobian -> affine, slope, x3, y3 (2 Fermat inversions), limbs
//   fq12_chain       one 288-lane workgroup per IO: the 2 x 144 limb products of acc*pw and pw^2 per bit in parallel
//   exp_rows         one lane per row: bit / remaining-exponent-limb cells (closed form)
//   gadget_rows      one lane per row: quotient / sign / carry witnesses of every modular gadget,
//                    interpreted from the AIR program (integer arithmetic)
//   table / lookups  range table column; per checked column histogram -> scan -> permuted columns
#include <algorithm>

#include "air_tables.h"
#include "ctx.hpp"
#include "fq.hpp"

namespace {

using fq::Fq;
using fq::Fq2;

// ---- generic field wrappers so the curve kernels are written once for Fq (G1) and Fq2 (G2) ----
template <int EXT>
struct Fld;
template <>
struct Fld<1> {
    using T = Fq;
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fq::add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fq::sub(a, b); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return fq::mul(a, b); }
    static __device__ __forceinline__ T sqr(const T& a) { return fq::sqr(a); }
    static __device__ __forceinline__ T inv(const T& a) { return fq::inv(a); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return fq::is_zero(a); }
    static __device__ __forceinline__ T load(const uint32_t* w) {  // 8 u32 standard form -> Montgomery
        Fq r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = w[i];
        return fq::to_mont(r);
    }
};
template <>
struct Fld<2> {
    using T = Fq2;
    static __device__ __forceinline__ T add(const T& a, const T& b) { return fq::add(a, b); }
    static __device__ __forceinline__ T sub(const T& a, const T& b) { return fq::sub(a, b); }
    static __device__ __forceinline__ T mul(const T& a, const T& b) { return fq::mul(a, b); }
    static __device__ __forceinline__ T sqr(const T& a) { return fq::sqr(a); }
    static __device__ __forceinline__ T inv(const T& a) { return fq::inv(a); }
    static __device__ __forceinline__ bool is_zero(const T& a) { return fq::is_zero(a); }
    static __device__ __forceinline__ T load(const uint32_t* w) {
        return Fq2{Fld<1>::load(w), Fld<1>::load(w + 8)};
    }
};

__device__ __forceinline__ Fq one_of(Fq*) { return fq::one_m(); }
__device__ __forceinline__ Fq2 one_of(Fq2*) { return Fq2{fq::one_m(), fq::zero()}; }

// the affine point at `w` (x, y as 8-limb words) lies on E(Fp): y^2 = x^3 + 3, or on the twist E'(Fp2): (9 + u)(y^2 - x^3) = 3.
// Records with a point off its curve are not provable (SIPP_E_WITNESS, as in the CPU restatement): the chord / tangent rules are
// a group law only on ONE curve, and the accumulator scan of the generator sums in another order than the AIR's chain.
template <int EXT>
__device__ __forceinline__ bool point_on_curve(const uint32_t* w) {
    using F = Fld<EXT>;
    const auto x = F::load(w), y = F::load(w + 8 * EXT);
    const auto d = F::sub(F::sqr(y), F::mul(F::sqr(x), x));
    const Fq three = fq::small_m(3);
    if constexpr (EXT == 1) {
        return fq::is_zero(fq::sub(d, three));
    } else {
        const Fq nine = fq::small_m(9);
        const Fq re = fq::sub(fq::mul(nine, d.c0), d.c1), im = fq::add(d.c0, fq::mul(nine, d.c1));
        return fq::is_zero(fq::sub(re, three)) && fq::is_zero(im);
    }
}


template <int EXT>
struct Jac {
    typename Fld<EXT>::T x, y, z;
};

template <int EXT>
__device__ __forceinline__ Jac<EXT> jac_dbl(const Jac<EXT>& p) {
    using F = Fld<EXT>;
    auto A = F::sqr(p.x), B = F::sqr(p.y), C = F::sqr(B);
    auto t = F::add(p.x, B);
    auto D = F::sub(F::sub(F::sqr(t), A), C);
    D = F::add(D, D);
    auto E = F::add(F::add(A, A), A);
    auto Fv = F::sqr(E);
    Jac<EXT> r;
    r.x = F::sub(Fv, F::add(D, D));
    auto C8 = F::add(C, C);
    C8 = F::add(C8, C8);
    C8 = F::add(C8, C8);
    r.y = F::sub(F::mul(E, F::sub(D, r.x)), C8);
    auto yz = F::mul(p.y, p.z);
    r.z = F::add(yz, yz);
    return r;
}

template <int EXT>
__device__ __forceinline__ Jac<EXT> jac_add(const Jac<EXT>& p, const Jac<EXT>& q) {
    using F = Fld<EXT>;
    auto z1z1 = F::sqr(p.z), z2z2 = F::sqr(q.z);
    auto u1 = F::mul(p.x, z2z2), u2 = F::mul(q.x, z1z1);
    auto s1 = F::mul(F::mul(p.y, q.z), z2z2), s2 = F::mul(F::mul(q.y, p.z), z1z1);
    auto h = F::sub(u2, u1);
    auto rr = F::sub(s2, s1);
    if (F::is_zero(h)) {
        // the same x: p == q (the chord formula would divide zero by zero: double instead) or p == -q (the sum is the point
        // at infinity).  The operands of the scan are sums of disjoint sets of 2^j x and the offset: equal whenever the
        // offset is a small multiple of x (offset = 3 x, bits 0 and 1 set) -- records the AIR's sequential chain proves.
        if (F::is_zero(rr)) return jac_dbl<EXT>(p);
        Jac<EXT> inf;
        inf.x = one_of((typename F::T*)nullptr);
        inf.y = inf.x;
        inf.z = F::sub(inf.x, inf.x);
        return inf;
    }
    auto i = F::add(h, h);
    i = F::sqr(i);
    auto j = F::mul(h, i);
    rr = F::add(rr, rr);
    auto v = F::mul(u1, i);
    Jac<EXT> r;
    r.x = F::sub(F::sub(F::sqr(rr), j), F::add(v, v));
    auto s1j = F::mul(s1, j);
    r.y = F::sub(F::mul(rr, F::sub(v, r.x)), F::add(s1j, s1j));
    auto zz = F::add(p.z, q.z);
    r.z = F::mul(F::sub(F::sub(F::sqr(zz), z1z1), z2z2), h);
    return r;
}


// scratch per row: Jacobian R_k and P_k (the values the row's constraints see)
template <int EXT>
struct RowPts {
    Jac<EXT> R, P;
};

template <int EXT>
__device__ __forceinline__ bool jac_is_inf(const Jac<EXT>& p) { return Fld<EXT>::is_zero(p.z); }

// group addition with the point at infinity (z == 0) as identity; jac_add is complete (p == q doubles, p == -q gives
// infinity), so the scan below is right for EVERY record and the records refused are exactly those whose rows have no
// witness (curve_rows: the AIR's own chain meets R = +-P), as in the CPU restatement.
template <int EXT>
__device__ __forceinline__ Jac<EXT> jac_add_id(const Jac<EXT>& p, const Jac<EXT>& q) {
    if (jac_is_inf<EXT>(p)) return q;
    if (jac_is_inf<EXT>(q)) return p;
    return jac_add<EXT>(p, q);
}

// rows[512 io + 2k] = (R_k, P_k) for the add row of bit k, rows[.. + 2k + 1] = (R_{k+1}, P_k) for its double row,
// where P_k = 2^k x and R_k = offset + sum_{j<k} bit_j P_j.
//   curve_dbl_par_kernel  4 (Fq) / 8 (Fq2) lanes per IO: the 255 sequential doublings, the base-field products of one
//                      doubling spread over the lanes (three dependency levels, exchanged through LDS)
//   curve_scan_kernel  one 256-lane workgroup per IO (lane k <-> bit k): inclusive Hillis-Steele scan of
//                      T_k = bit_k ? P_k : inf (8 point additions deep), then R_{k+1} = offset + scan_k
// 2,200 field-multiplication times on the critical path instead of 6,100 for a one-lane-per-IO double-and-add.
// ---- the doubling chain with several lanes per IO ----
// The 255 doublings are serial, but the 16 (Fq2) / 7 (Fq) base-field products inside one doubling are not: they fall
// into three dependency levels of 7 + 6 + 3 (Fq2) or 3 + 3 + 1 (Fq) independent products.  G lanes share an IO: every
// lane keeps the whole point, all lanes form the level's operand pairs (cheap additions, done redundantly), lane t
// multiplies pair t, the products are exchanged through LDS, and every lane finishes the level's additions.  One
// doubling then costs 3 product latencies instead of 16 / 7.
__device__ __forceinline__ void lds_put(Fq* dst, const Fq& v) {
    uint4* d = reinterpret_cast<uint4*>(dst);
    d[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    d[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
__device__ __forceinline__ Fq lds_get(const Fq* src) {
    const uint4* d = reinterpret_cast<const uint4*>(src);
    const uint4 a = d[0], b = d[1];
    Fq v;
    v.l[0] = a.x; v.l[1] = a.y; v.l[2] = a.z; v.l[3] = a.w;
    v.l[4] = b.x; v.l[5] = b.y; v.l[6] = b.z; v.l[7] = b.w;
    return v;
}

// one level: NT operand pairs (identical in every lane of the group) -> NT products, visible to every lane
template <int NT>
__device__ __forceinline__ void par_level(Fq* s_op, Fq* s_pr, int sub, const Fq (&a)[NT], const Fq (&b)[NT], Fq (&p)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; t++) {
        lds_put(s_op + 2 * t, a[t]);
        lds_put(s_op + 2 * t + 1, b[t]);
    }
    __syncthreads();
    const int task = sub < NT ? sub : NT - 1;
    const Fq x = lds_get(s_op + 2 * task), y = lds_get(s_op + 2 * task + 1);
    lds_put(s_pr + sub, fq::mul(x, y));
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NT; t++) p[t] = lds_get(s_pr + t);
}

__device__ __forceinline__ Jac<1> jac_dbl_par(const Jac<1>& P, Fq* s_op, Fq* s_pr, int sub) {
    Fq p1[3], p2[3], p3[1];
    {
        const Fq a[3] = {P.x, P.y, P.y}, b[3] = {P.x, P.y, P.z};
        par_level<3>(s_op, s_pr, sub, a, b, p1);  // A = X^2, B = Y^2, YZ
    }
    const Fq A = p1[0], B = p1[1];
    const Fq U = fq::add(P.x, B), E = fq::add(fq::add(A, A), A);
    {
        const Fq a[3] = {B, U, E};
        par_level<3>(s_op, s_pr, sub, a, a, p2);  // C = B^2, (X + B)^2, F = E^2
    }
    const Fq C = p2[0];
    Fq D = fq::sub(fq::sub(p2[1], A), C);
    D = fq::add(D, D);
    Jac<1> r;
    r.x = fq::sub(p2[2], fq::add(D, D));
    {
        const Fq a[1] = {E}, b[1] = {fq::sub(D, r.x)};
        par_level<1>(s_op, s_pr, sub, a, b, p3);
    }
    Fq C8 = fq::add(C, C);
    C8 = fq::add(C8, C8);
    C8 = fq::add(C8, C8);
    r.y = fq::sub(p3[0], C8);
    r.z = fq::add(p1[2], p1[2]);
    return r;
}

__device__ __forceinline__ Jac<2> jac_dbl_par(const Jac<2>& P, Fq* s_op, Fq* s_pr, int sub) {
    Fq p1[7], p2[6], p3[3];
    {
        // X^2 = (x0+x1)(x0-x1) + 2 x0 x1 u ; Y^2 likewise ; Y Z by Karatsuba
        const Fq a[7] = {fq::add(P.x.c0, P.x.c1), P.x.c0, fq::add(P.y.c0, P.y.c1), P.y.c0, P.y.c0, P.y.c1,
                         fq::add(P.y.c0, P.y.c1)};
        const Fq b[7] = {fq::sub(P.x.c0, P.x.c1), P.x.c1, fq::sub(P.y.c0, P.y.c1), P.y.c1, P.z.c0, P.z.c1,
                         fq::add(P.z.c0, P.z.c1)};
        par_level<7>(s_op, s_pr, sub, a, b, p1);
    }
    const Fq2 A{p1[0], fq::add(p1[1], p1[1])}, B{p1[2], fq::add(p1[3], p1[3])};
    const Fq2 YZ{fq::sub(p1[4], p1[5]), fq::sub(fq::sub(p1[6], p1[4]), p1[5])};
    const Fq2 U = fq::add(P.x, B), E = fq::add(fq::add(A, A), A);
    {
        const Fq a[6] = {fq::add(B.c0, B.c1), B.c0, fq::add(U.c0, U.c1), U.c0, fq::add(E.c0, E.c1), E.c0};
        const Fq b[6] = {fq::sub(B.c0, B.c1), B.c1, fq::sub(U.c0, U.c1), U.c1, fq::sub(E.c0, E.c1), E.c1};
        par_level<6>(s_op, s_pr, sub, a, b, p2);
    }
    const Fq2 C{p2[0], fq::add(p2[1], p2[1])}, S{p2[2], fq::add(p2[3], p2[3])}, Fv{p2[4], fq::add(p2[5], p2[5])};
    Fq2 D = fq::sub(fq::sub(S, A), C);
    D = fq::add(D, D);
    Jac<2> r;
    r.x = fq::sub(Fv, fq::add(D, D));
    const Fq2 Wd = fq::sub(D, r.x);
    {
        const Fq a[3] = {E.c0, E.c1, fq::add(E.c0, E.c1)}, b[3] = {Wd.c0, Wd.c1, fq::add(Wd.c0, Wd.c1)};
        par_level<3>(s_op, s_pr, sub, a, b, p3);
    }
    const Fq2 EW{fq::sub(p3[0], p3[1]), fq::sub(fq::sub(p3[2], p3[0]), p3[1])};
    Fq2 C8 = fq::add(C, C);
    C8 = fq::add(C8, C8);
    C8 = fq::add(C8, C8);
    r.y = fq::sub(EW, C8);
    r.z = fq::add(YZ, YZ);
    return r;
}

// G lanes per IO, one 64-lane block = 64 / G IOs
template <int EXT, int G>
__global__ void __launch_bounds__(64) curve_dbl_par_kernel(const uint32_t* __restrict__ ios, uint32_t num_io, uint32_t ppi,
                                                          RowPts<EXT>* __restrict__ rows) {
    using F = Fld<EXT>;
    using T = typename F::T;
    constexpr int GPB = 64 / G;
    __shared__ Fq s_op[GPB][14], s_pr[GPB][G];
    const int grp = threadIdx.x / G, sub = threadIdx.x % G;
    uint32_t io = blockIdx.x * GPB + grp;
    const bool live = io < num_io && sub == 0;
    if (io >= num_io) io = num_io - 1;  // spare groups redo the last IO (uniform control flow for the barriers)
    const uint32_t* rec = ios + (size_t)io * ppi;
    const int w = 8 * EXT;
    RowPts<EXT>* out = rows + (size_t)io * 512;
    Jac<EXT> P{F::load(rec), F::load(rec + w), one_of((T*)nullptr)};
    for (int b = 0; b < 256; b++) {
        if (live) out[2 * b].P = P;
        if (b != 255) P = jac_dbl_par(P, s_op[grp], s_pr[grp], sub);
    }
}

template <int EXT>
__global__ void __launch_bounds__(256) curve_scan_kernel(const uint32_t* __restrict__ ios, uint32_t num_io, uint32_t ppi,
                                                        RowPts<EXT>* __restrict__ rows) {
    using F = Fld<EXT>;
    using T = typename F::T;
    __shared__ Jac<EXT> pts[256];
    const uint32_t io = blockIdx.x, k = threadIdx.x;
    const uint32_t* rec = ios + (size_t)io * ppi;
    const int w = 8 * EXT;
    RowPts<EXT>* out = rows + (size_t)io * 512;
    const Jac<EXT> Pk = out[2 * k].P;
    const uint32_t* ex = rec + 4 * w;
    const int bit = (ex[k >> 5] >> (k & 31)) & 1;
    Jac<EXT> inf;
    inf.x = one_of((T*)nullptr);
    inf.y = one_of((T*)nullptr);
    inf.z = F::sub(inf.x, inf.x);
    Jac<EXT> v = bit ? Pk : inf;
    for (int off = 1; off < 256; off <<= 1) {
        pts[k] = v;
        __syncthreads();
        Jac<EXT> o = inf;
        if ((int)k >= off) o = pts[k - off];
        __syncthreads();
        if ((int)k >= off) v = jac_add_id<EXT>(o, v);
    }
    const Jac<EXT> offs{F::load(rec + 2 * w), F::load(rec + 3 * w), one_of((T*)nullptr)};
    const Jac<EXT> A = jac_add_id<EXT>(offs, v);  // R_{k+1}
    pts[k] = A;
    __syncthreads();
    const Jac<EXT> Rk = k ? pts[k - 1] : offs;
    out[2 * k].R = Rk;
    out[2 * k + 1].R = A;
    out[2 * k + 1].P = Pk;
}

__device__ __forceinline__ void store_limbs16(uint64_t* tr, size_t n, int col, size_t row, const Fq& std_form) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        tr[(size_t)(col + 2 * i) * n + row] = std_form.l[i] & 0xffffu;
        tr[(size_t)(col + 2 * i + 1) * n + row] = std_form.l[i] >> 16;
    }
}
// checked limb vector: cpl = 1 -> one cell per 16-bit limb, cpl = 2 -> (lo8, hi8) per limb
__device__ __forceinline__ void store_checked(uint64_t* tr, size_t n, int col, size_t row, const Fq& std_form, int cpl) {
    if (cpl == 1) {
        store_limbs16(tr, n, col, row, std_form);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint32_t v = std_form.l[i];
            tr[(size_t)(col + 4 * i) * n + row] = v & 0xffu;
            tr[(size_t)(col + 4 * i + 1) * n + row] = (v >> 8) & 0xffu;
            tr[(size_t)(col + 4 * i + 2) * n + row] = (v >> 16) & 0xffu;
            tr[(size_t)(col + 4 * i + 3) * n + row] = v >> 24;
        }
    }
}
template <int EXT>
__device__ __forceinline__ void store_f_u16(uint64_t* tr, size_t n, int col, size_t row, const typename Fld<EXT>::T& v);
template <>
__device__ __forceinline__ void store_f_u16<1>(uint64_t* tr, size_t n, int col, size_t row, const Fq& v) {
    store_limbs16(tr, n, col, row, fq::from_mont(v));
}
template <>
__device__ __forceinline__ void store_f_u16<2>(uint64_t* tr, size_t n, int col, size_t row, const Fq2& v) {
    store_limbs16(tr, n, col, row, fq::from_mont(v.c0));
    store_limbs16(tr, n, col + 16, row, fq::from_mont(v.c1));
}
template <int EXT>
__device__ __forceinline__ void store_f_chk(uint64_t* tr, size_t n, int col, size_t row, const typename Fld<EXT>::T& v,
                                            int cpl);
template <>
__device__ __forceinline__ void store_f_chk<1>(uint64_t* tr, size_t n, int col, size_t row, const Fq& v, int cpl) {
    store_checked(tr, n, col, row, fq::from_mont(v), cpl);
}
template <>
__device__ __forceinline__ void store_f_chk<2>(uint64_t* tr, size_t n, int col, size_t row, const Fq2& v, int cpl) {
    store_checked(tr, n, col, row, fq::from_mont(v.c0), cpl);
    store_checked(tr, n, col + 16 * cpl, row, fq::from_mont(v.c1), cpl);
}

// sipp_exp_outputs for the curves: the accumulator after the last bit (R_256 = offset + [exp] x, Jacobian) of every IO goes
// straight into the record's output words -- one inversion per IO, no trace rows.  A result at infinity is SIPP_E_WITNESS.
template <int EXT>
__global__ void __launch_bounds__(64) curve_outputs_kernel(const RowPts<EXT>* __restrict__ rows, uint32_t* __restrict__ ios,
                                                         uint32_t num_io, uint32_t ppi, int* __restrict__ err) {
    using F = Fld<EXT>;
    const uint32_t io = blockIdx.x * blockDim.x + threadIdx.x;
    if (io >= num_io) return;
    const Jac<EXT> J = rows[(size_t)io * 512 + 511].R;
    const uint32_t* rec = ios + (size_t)io * ppi;
    if (jac_is_inf<EXT>(J) || !point_on_curve<EXT>(rec) || !point_on_curve<EXT>(rec + 16 * EXT)) {
        atomicExch(err, SIPP_E_WITNESS);
        return;
    }
    // z^(p-2): the binary extended Euclid (fq::inv_gcd) was tried here -- with 64 different values per wave its lanes diverge
    // and the kernel takes 0.25-0.26 ms instead of 0.23 ms (round 2)
    const auto zi = F::inv(J.z), zi2 = F::sqr(zi);
    const auto x = F::mul(J.x, zi2), y = F::mul(J.y, F::mul(zi2, zi));
    uint32_t* outw = ios + (size_t)io * ppi + ppi - 16 * EXT;
    if constexpr (EXT == 1) {
        const Fq xs = fq::from_mont(x), ys = fq::from_mont(y);
        for (int l = 0; l < 8; l++) {
            outw[l] = xs.l[l];
            outw[8 + l] = ys.l[l];
        }
    } else {
        const Fq c[4] = {fq::from_mont(x.c0), fq::from_mont(x.c1), fq::from_mont(y.c0), fq::from_mont(y.c1)};
        for (int k = 0; k < 4; k++)
            for (int l = 0; l < 8; l++) outw[8 * k + l] = c[k].l[l];
    }
}

struct CurveCols {
    int Rx, Ry, Px, Py, lam, X3, Y3, cpl;
};

// one lane per row: affine coordinates, slope and result of the row's group operation
// the same point in Jacobian coordinates (neither at infinity): X1 Z2^2 = X2 Z1^2 and Y1 Z2^3 = Y2 Z1^3
template <int EXT>
__device__ __forceinline__ bool jac_same(const Jac<EXT>& p, const Jac<EXT>& q) {
    using F = Fld<EXT>;
    const auto z1 = F::sqr(p.z), z2 = F::sqr(q.z);
    if (!F::is_zero(F::sub(F::mul(p.x, z2), F::mul(q.x, z1)))) return false;
    return F::is_zero(F::sub(F::mul(p.y, F::mul(z2, q.z)), F::mul(q.y, F::mul(z1, p.z))));
}

// Inverses of one NONZERO base-field value per lane for a whole block (Montgomery's trick through LDS, round 5): prefix and suffix
// products by two Hillis-Steele scans, ONE inversion per block (lane 0, Euclid: ~37 k instructions) and two products per lane --
// 2 log2(B) + 3 products per lane instead of a 380-product exponentiation in EVERY lane.  curve_rows_kernel spent 9 of 10
// instructions in its two Fermat inversions per row (27 ms of an n = 4096 instance).  s, s2: B values of LDS each.
template <int B>
__device__ __forceinline__ Fq block_inverse(const Fq& v, Fq* s, Fq* s2) {
    const int t = threadIdx.x;
    const Fq one = fq::one_m();
    Fq pre = v, suf = v;
    s[t] = v;
    s2[t] = v;
    __syncthreads();
#pragma unroll 1
    for (int off = 1; off < B; off <<= 1) {
        const Fq lo = t >= off ? s[t - off] : one, hi = t + off < B ? s2[t + off] : one;
        __syncthreads();
        pre = fq::mul(pre, lo);
        suf = fq::mul(suf, hi);
        s[t] = pre;
        s2[t] = suf;
        __syncthreads();
    }
    // s[t] = v_0 .. v_t, s2[t] = v_t .. v_{B-1}
    if (t == 0) s2[0] = fq::inv_gcd(s[B - 1]);
    __syncthreads();
    Fq r = s2[0];
    r = fq::mul(r, t ? s[t - 1] : one);
    r = fq::mul(r, t + 1 < B ? s2[t + 1] : one);
    return r;
}
template <int B>
__device__ __forceinline__ Fq block_inverse_of(const Fq& w, Fq* s, Fq* s2) { return block_inverse<B>(w, s, s2); }
template <int B>
__device__ __forceinline__ Fq2 block_inverse_of(const Fq2& w, Fq* s, Fq* s2) {
    const Fq ni = block_inverse<B>(fq::add(fq::sqr(w.c0), fq::sqr(w.c1)), s, s2);   // 1 / (w0 + w1 u) = (w0 - w1 u) / (w0^2 + w1^2)
    return Fq2{fq::mul(w.c0, ni), fq::neg(fq::mul(w.c1, ni))};
}

// hard != 0: the hardened AIR (API kinds 4 / 5; tools/air_gen.py::build_curve).  Where the accumulator IS the running power on an add
// row (R = P) or its negative (R = -P) the row gets slope 0 instead of SIPP_E_WITNESS (its result is not used: flags eq / ng).  After
// R = P the double row still shows the OLD accumulator -- the sum, which is that row's own double, is handed over at the row's end (eqc).
// After R = -P the accumulator is the identity: the state bit inf (column col_inf, written here from the scan's true accumulator) is
// set and the R cells keep the last finite value until a used addition copies P.
// The row needs 1 / (ZR ZP) for the affine coordinates and 1 / den for the slope; both come out of ONE inverse per row, of
// w = (ZR ZP) b with b = ZR ZP D (add rows: den = D / (ZR ZP)^2, D = XP ZR^2 - XR ZP^2, slope = (YP ZR^3 - YR ZP^3) / b) or
// b = 2 YP ZP (double rows: slope = 3 XP^2 / b) -- and that inverse is shared by the block (block_inverse).
constexpr int ROWS_BLOCK = 256;
template <int EXT>
__global__ void __launch_bounds__(ROWS_BLOCK) curve_rows_kernel(const RowPts<EXT>* __restrict__ rows, uint64_t* __restrict__ tr,
                                                               size_t n, CurveCols c, int hard, int col_inf, int* __restrict__ err) {
    using F = Fld<EXT>;
    using T = typename F::T;
    __shared__ Fq s_pre[ROWS_BLOCK], s_suf[ROWS_BLOCK];
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // n is a multiple of the block: every lane has a row
    RowPts<EXT> rp = rows[row];
    const bool is_add = (row & 1) == 0;
    const bool inf_here = jac_is_inf<EXT>(rp.R);
    if (hard) {
        tr[(size_t)col_inf * n + row] = inf_here ? 1 : 0;
        if (!is_add) {
            const RowPts<EXT> prev = rows[row - 1];
            if (!jac_is_inf<EXT>(prev.R) && !jac_is_inf<EXT>(prev.P) && jac_same<EXT>(prev.R, prev.P)) rp.R = prev.R;
        }
        // the identity has no cells: the last finite accumulator of the block stands in (the block's first row is finite)
        size_t r2 = row;
        const size_t row0 = row & ~(size_t)511;
        while (jac_is_inf<EXT>(rp.R) && r2 > row0) rp.R = rows[--r2].R;
    }
    const T one = one_of((T*)nullptr);
    const T zz = F::mul(rp.R.z, rp.P.z);
    bool bad = F::is_zero(zz);
    T num, b;
    if (is_add) {
        const T rz2 = F::sqr(rp.R.z), pz2 = F::sqr(rp.P.z);
        num = F::sub(F::mul(rp.P.y, F::mul(rz2, rp.R.z)), F::mul(rp.R.y, F::mul(pz2, rp.P.z)));
        b = F::mul(zz, F::sub(F::mul(rp.P.x, rz2), F::mul(rp.R.x, pz2)));
    } else {
        const T pxx = F::sqr(rp.P.x);
        num = F::add(F::add(pxx, pxx), pxx);
        const T yz = F::mul(rp.P.y, rp.P.z);
        b = F::add(yz, yz);
    }
    // R = +-P on an add row: cases of the hardened AIR (flags eq / ng; slope cells 0, result unused).  Under inf the R cells are the
    // stale last finite accumulator and the row's result is unused as well (t1 = 0): the same slope-0 row, as oracle/air.c fills it
    const bool same = hard && is_add && !bad && F::is_zero(b);
    bad = bad || (F::is_zero(b) && !same);
    const T asel = bad ? one : zz, bsel = (bad || same) ? one : b;
    const T iw = block_inverse_of<ROWS_BLOCK>(F::mul(asel, bsel), s_pre, s_suf);
    if (bad) {   // (after the block's barriers)
        atomicExch(err, SIPP_E_WITNESS);
        return;
    }
    // affine: x = X / Z^2, y = Y / Z^3
    const T izz = F::mul(bsel, iw), ib = F::mul(asel, iw);
    const T izr = F::mul(izz, rp.P.z), izp = F::mul(izz, rp.R.z);
    const T izr2 = F::mul(izr, izr), izp2 = F::mul(izp, izp);
    const T rx = F::mul(rp.R.x, izr2), ry = F::mul(rp.R.y, F::mul(izr2, izr));
    const T px = F::mul(rp.P.x, izp2), py = F::mul(rp.P.y, F::mul(izp2, izp));
    const T xa = is_add ? rx : px, ya = is_add ? ry : py, xb = px;
    const T lam = same ? F::sub(num, num) : F::mul(num, ib);
    const T x3 = F::sub(F::sub(F::mul(lam, lam), xa), xb);
    const T y3 = F::sub(F::mul(lam, F::sub(xa, x3)), ya);
    store_f_u16<EXT>(tr, n, c.Rx, row, rx);
    store_f_u16<EXT>(tr, n, c.Ry, row, ry);
    store_f_u16<EXT>(tr, n, c.Px, row, px);
    store_f_u16<EXT>(tr, n, c.Py, row, py);
    store_f_chk<EXT>(tr, n, c.lam, row, lam, c.cpl);
    store_f_chk<EXT>(tr, n, c.X3, row, x3, c.cpl);
    store_f_chk<EXT>(tr, n, c.Y3, row, y3, c.cpl);
}

// ---- Fq12 chain: one 384-lane workgroup per IO ----
// Per exponent bit, on the critical path: (A) 144 products acc_i * pw_j + the 78 distinct products pw_i * pw_j (i <= j), ONE
// Montgomery product per lane on waves 0..3 (one wave per SIMD); (B) 46 lanes sum them into the 23 + 23 coefficients d_m of
// the two plain polynomial products; (C) 72 lanes form the three terms of every folded coefficient of
// Fq[w]/(w^12 - 18 w^6 + 82) (one small-constant product each); (D) 24 lanes combine them; (E) state update + snapshot.
// Beside the path: waves 4..5 convert and store the trace cells of the PREVIOUS bit's two rows from the snapshot while
// waves 0..3 are in (A).  Round 1's kernel (288 lanes, 24-lane fold with 36 additions and two products in sequence,
// conversion on the path) took 2.97 ms for the 14 records of n = 128, this one the time in HISTORY.md section 6b.
struct Fq12Cols {
    int acc, pw, C, cpl;
};

struct Fq12Snap {
    Fq acc[12], pw[12], res[2][12];
};

__device__ __forceinline__ int sq_index(int i, int j) { return i * 12 - i * (i - 1) / 2 + (j - i); }  // i <= j, 0..77

__global__ void __launch_bounds__(384) fq12_chain_kernel(const uint32_t* __restrict__ ios, uint32_t num_io,
                                                        uint64_t* __restrict__ tr, size_t n, Fq12Cols c) {
    __shared__ Fq s_acc[12], s_pw[12], s_res[2][12], s_mul[144], s_sq[78], s_d[2][23], s_t[2][12][3];
    __shared__ Fq12Snap s_snap[2];
    const uint32_t io = blockIdx.x;
    const int t = threadIdx.x;
    const uint32_t* rec = ios + (size_t)io * SIPP_FQ12_IO_WORDS;
    if (t < 12) {
        s_pw[t] = Fld<1>::load(rec + 8 * t);
        s_acc[t] = Fld<1>::load(rec + 96 + 8 * t);
    }
    __syncthreads();
    const uint32_t* ex = rec + 192;
    // product lanes: t < 144 -> acc_i pw_j; 144 <= t < 222 -> pw_i pw_j with i <= j
    int pi = 0, pj = 0;
    if (t < 144) {
        pi = t / 12;
        pj = t % 12;
    } else if (t < 222) {
        int r = t - 144;
        while (r >= 12 - pi) {
            r -= 12 - pi;
            pi++;
        }
        pj = pi + r;
    }
    const int ct = t - 256;  // conversion lanes 0..71 on waves 4..5
    // fold lanes (C): lane t < 72 = (product sq, coefficient k, term); terms 1, 2 carry one small constant and one source index
    const int f_sq = t / 36, f_k = (t % 36) / 3, f_term = t % 3;
    const int f_m = f_term == 0 ? f_k : f_k < 6 ? (f_term == 1 ? f_k + 12 : f_k + 18) : (f_term == 1 ? f_k + 6 : f_k + 12);
    const Fq f_c = fq::small_m(f_k < 6 ? (f_term == 1 ? 82u : 1476u) : (f_term == 1 ? 18u : 242u));
    for (int b = 0; b <= 256; b++) {
        // (A) products of bit b | cells of bit b - 1
        if (t < 144) {
            if (b < 256) s_mul[t] = fq::mul(s_acc[pi], s_pw[pj]);
        } else if (t < 222) {
            if (b < 256) s_sq[t - 144] = fq::mul(s_pw[pi], s_pw[pj]);
        } else if (ct >= 0 && ct < 72 && b > 0) {
            // task = (row 0/1, array 0 acc / 1 pw / 2 C, tower component tc); the cells hold the tower basis
            // Fq2[w]/(w^6 - (9 + u)): tc = 2i -> c_i + 9 c_{i+6}, tc = 2i + 1 -> c_{i+6}   (tools/air_gen.py build_fq12)
            const int pb = b - 1;
            const int bit = (int)((ex[pb >> 5] >> (pb & 31)) & 1);
            const Fq12Snap& sn = s_snap[pb & 1];
            const int rowsel = ct / 36, arr = (ct % 36) / 12, tc = ct % 12;
            const size_t row = (size_t)io * 512 + 2 * pb + rowsel;
            const Fq* src = arr == 0 ? ((rowsel && bit) ? sn.res[0] : sn.acc) : arr == 1 ? sn.pw : sn.res[rowsel];
            const int i = tc >> 1;
            Fq v = src[i + 6];
            if (!(tc & 1)) {
                // c_i + 9 c_{i+6}: 9 v = 8 v + v (three doublings and an addition are shorter than a Montgomery product)
                Fq v2 = fq::add(v, v);
                v2 = fq::add(v2, v2);
                v2 = fq::add(v2, v2);
                v = fq::add(src[i], fq::add(v2, v));
            }
            v = fq::from_mont(v);
            if (arr == 0) store_limbs16(tr, n, c.acc + 16 * tc, row, v);
            else if (arr == 1) store_limbs16(tr, n, c.pw + 16 * tc, row, v);
            else store_checked(tr, n, c.C + 16 * c.cpl * tc, row, v, c.cpl);
        }
        if (b == 256) break;
        __syncthreads();
        // (B) d_m = sum_{i + j = m} of the products, m = 0..22, for acc * pw (lanes 0..22) and pw * pw (lanes 23..45)
        if (t < 46) {
            const int sq = t >= 23, m = t % 23;
            const int lo = m > 11 ? m - 11 : 0, hi = m < 11 ? m : 11;  // i range with 0 <= m - i <= 11
            Fq d = fq::zero();
            if (!sq) {
                for (int i = lo; i <= hi; i++) d = fq::add(d, s_mul[i * 12 + (m - i)]);
            } else {
                for (int i = lo; 2 * i < m; i++) d = fq::add(d, s_sq[sq_index(i, m - i)]);
                d = fq::add(d, d);
                if (!(m & 1)) d = fq::add(d, s_sq[sq_index(m >> 1, m >> 1)]);
            }
            s_d[sq][m] = d;
        }
        __syncthreads();
        // (C) k < 6: d_k, 82 d_{k+12}, 1476 d_{k+18} (0 for k = 5: d_23 does not exist);  k >= 6: d_k, 18 d_{k+6}, 242 d_{k+12}
        if (t < 72) {
            Fq v = fq::zero();
            if (f_term == 0)
                v = s_d[f_sq][f_k];
            else if (f_m < 23)
                v = fq::mul(f_c, s_d[f_sq][f_m]);
            s_t[f_sq][f_k][f_term] = v;
        }
        __syncthreads();
        // (D) r_k = t0 -/+ t1 -/+ t2
        if (t < 24) {
            const int sq = t >= 12, k = t % 12;
            const Fq* tt = s_t[sq][k];
            s_res[sq][k] = k < 6 ? fq::sub(fq::sub(tt[0], tt[1]), tt[2]) : fq::add(fq::add(tt[0], tt[1]), tt[2]);
        }
        __syncthreads();
        // (E) snapshot for the conversion lanes, then the state of the next bit
        if (t < 12) {
            const int bit = (int)((ex[b >> 5] >> (b & 31)) & 1);
            Fq12Snap& sn = s_snap[b & 1];
            sn.acc[t] = s_acc[t];
            sn.pw[t] = s_pw[t];
            sn.res[0][t] = s_res[0][t];
            sn.res[1][t] = s_res[1][t];
            if (bit) s_acc[t] = s_res[0][t];
            if (b != 255) s_pw[t] = s_res[1][t];
        }
        __syncthreads();
    }
}

// ---- claimed outputs: the last row of every IO block must hold the output the record claims (the AIR binds those
// cells to the public inputs anyway; checking here makes a wrong record fail fast, like the CPU restatement) ----
__global__ void check_outputs_kernel(const uint32_t* __restrict__ ios, uint32_t num_io, uint32_t ppi, int kind,
                                     const uint64_t* __restrict__ tr, size_t n, int col_state, int* __restrict__ err) {
    const uint32_t io = blockIdx.x * blockDim.x + threadIdx.x;
    if (io >= num_io) return;
    const size_t row = (size_t)io * 512 + 511;
    const uint32_t* rec = ios + (size_t)io * ppi;
    bool bad = false;
    if (kind == 2) {
        const uint32_t* outw = rec + ppi - 96;   // MyFq12 coefficients; the cells hold the tower basis
        const Fq m9 = fq::small_m(9);
        for (int tc = 0; tc < 12; tc++) {
            const int i = tc >> 1;
            Fq v = Fld<1>::load(outw + 8 * (i + 6));
            if (!(tc & 1)) v = fq::add(Fld<1>::load(outw + 8 * i), fq::mul(m9, v));
            v = fq::from_mont(v);
            for (int l = 0; l < 16; l++) {
                const uint64_t want = (v.l[l >> 1] >> (16 * (l & 1))) & 0xffffu;
                bad |= tr[(size_t)(col_state + 16 * tc + l) * n + row] != want;
            }
        }
    } else {
        const int words = kind == 0 ? 16 : 32;   // (x, y) or (x.c0, x.c1, y.c0, y.c1): the Rx | Ry cells in order
        const uint32_t* outw = rec + ppi - words;
        bad |= kind == 0 ? !(point_on_curve<1>(rec) && point_on_curve<1>(rec + 16)) : !(point_on_curve<2>(rec) && point_on_curve<2>(rec + 32));
        for (int wd = 0; wd < words; wd++) {
            bad |= tr[(size_t)(col_state + 2 * wd) * n + row] != (outw[wd] & 0xffffu);
            bad |= tr[(size_t)(col_state + 2 * wd + 1) * n + row] != (outw[wd] >> 16);
        }
    }
    if (bad) atomicExch(err, SIPP_E_WITNESS);
}

// the same cells read back INTO the records (sipp_exp_outputs): the inverse of check_outputs_kernel
__global__ void write_outputs_kernel(uint32_t* __restrict__ ios, uint32_t num_io, uint32_t ppi, int kind,
                                     const uint64_t* __restrict__ tr, size_t n, int col_state) {
    const uint32_t io = blockIdx.x * blockDim.x + threadIdx.x;
    if (io >= num_io) return;
    const size_t row = (size_t)io * 512 + 511;
    uint32_t* rec = ios + (size_t)io * ppi;
    if (kind == 2) {
        // cells: tower basis (a_i = c_i + 9 c_{i+6}, b_i = c_{i+6}) as 16-bit limbs -> MyFq12 coefficients c_i = a_i - 9 b_i
        uint32_t* outw = rec + ppi - 96;
        const Fq m9 = fq::small_m(9);
        for (int i = 0; i < 6; i++) {
            Fq ab[2];
            for (int h = 0; h < 2; h++) {
                uint32_t w[8];
                for (int l = 0; l < 8; l++)
                    w[l] = (uint32_t)tr[(size_t)(col_state + 16 * (2 * i + h) + 2 * l) * n + row] |
                           ((uint32_t)tr[(size_t)(col_state + 16 * (2 * i + h) + 2 * l + 1) * n + row] << 16);
                ab[h] = Fld<1>::load(w);
            }
            const Fq lo = fq::from_mont(fq::sub(ab[0], fq::mul(m9, ab[1]))), hi = fq::from_mont(ab[1]);
            for (int l = 0; l < 8; l++) {
                outw[8 * i + l] = lo.l[l];
                outw[8 * (i + 6) + l] = hi.l[l];
            }
        }
    } else {
        const int words = kind == 0 ? 16 : 32;
        uint32_t* outw = rec + ppi - words;
        for (int wd = 0; wd < words; wd++)
            outw[wd] = (uint32_t)tr[(size_t)(col_state + 2 * wd) * n + row] |
                       ((uint32_t)tr[(size_t)(col_state + 2 * wd + 1) * n + row] << 16);
    }
}

struct SippBnP {
    uint32_t l[16];   // 16-bit limbs of the BN254 base-field prime
};

// ---- hardened curve AIRs (API kinds 4 / 5): per row T3 = p - 1 - x3 with its borrow bits, and on add rows whose bit is set the
// witness that R.x and P.x differ in a limb: nz_j = 1 / (Px_j - Rx_j) in the Goldilocks field at the first such limb.  Reads cells
// written by curve_rows_kernel and exp_rows_kernel; one lane per row, integer work only.
__global__ void __launch_bounds__(256) harden_rows_kernel(uint64_t* __restrict__ tr, size_t n, int ext, int cpl, int col_rx, int col_px,
                                                         int col_bit, int col_x3, int col_nz, int col_cb, int col_t3, int col_eq, int col_u,
                                                         int col_eqc, int col_ng, int col_inf, int col_t1, int col_v, int col_w, int col_ngv,
                                                         int col_cn, SippBnP pl, int* __restrict__ err) {
    const size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const int nc = 16 * ext;
    bool bad = false;
    for (int c = 0; c < ext; c++) {
        int64_t borrow = 0;
        for (int i = 0; i < 16; i++) {
            const int64_t x = cpl == 1 ? (int64_t)tr[(size_t)(col_x3 + 16 * c + i) * n + row]
                                       : (int64_t)(tr[(size_t)(col_x3 + 2 * (16 * c + i)) * n + row] + 256 * tr[(size_t)(col_x3 + 2 * (16 * c + i) + 1) * n + row]);
            const int64_t d = (int64_t)pl.l[i] - (i == 0 ? 1 : 0) - x - borrow;
            borrow = d < 0 ? 1 : 0;
            const uint64_t t = (uint64_t)(d + (borrow << 16));
            if (cpl == 1) {
                tr[(size_t)(col_t3 + 16 * c + i) * n + row] = t;
            } else {
                tr[(size_t)(col_t3 + 2 * (16 * c + i)) * n + row] = t & 0xffu;
                tr[(size_t)(col_t3 + 2 * (16 * c + i) + 1) * n + row] = t >> 8;
            }
            if (i < 15) tr[(size_t)(col_cb + 15 * c + i) * n + row] = (uint64_t)borrow;
        }
        bad |= borrow != 0;
    }
    // flags (tools/air_gen.py::build_curve): eq / ng on add rows: R = P / R = -P as limb vectors (the Ry cells follow the Rx cells, the Py
    // cells the Px cells; ng: the x limbs equal and Ry + Py = p limb by limb with the carries cn); inf was written by curve_rows_kernel;
    // t1 = bit (1 - inf), u = t1 (1 - eq - ng), v = bit inf, w = t1 ng; eqc on a double row = t1 eq of the add row before it
    const bool is_add = (row & 1) == 0;
    const size_t cr = is_add ? row : row - 1;
    bool xeq = true, yeq = true;
    for (int j = 0; j < nc; j++) {
        xeq &= tr[(size_t)(col_px + j) * n + cr] == tr[(size_t)(col_rx + j) * n + cr];
        yeq &= tr[(size_t)(col_px + nc + j) * n + cr] == tr[(size_t)(col_rx + nc + j) * n + cr];
    }
    const bool eqv = xeq && yeq;
    bool ng = is_add && xeq && !yeq;
    uint32_t carries = 0;    // bit 15 c + i
    for (int c = 0; c < ext; c++) {
        int64_t carry = 0;
        for (int i = 0; i < 16; i++) {
            const int64_t sum = (int64_t)tr[(size_t)(col_rx + nc + 16 * c + i) * n + row] + (int64_t)tr[(size_t)(col_px + nc + 16 * c + i) * n + row] + carry;
            const int64_t d = sum - (int64_t)pl.l[i];
            if (d != 0 && d != 65536) ng = false;
            carry = d == 65536 ? 1 : 0;
            if (i < 15 && carry) carries |= 1u << (15 * c + i);
        }
        if (carry) ng = false;
    }
    const bool bit = tr[(size_t)col_bit * n + row] != 0, infc = tr[(size_t)col_inf * n + row] != 0;
    const bool eq = is_add && eqv, t1 = bit && !infc, used = t1 && !eq && !ng;
    const bool eqc = !is_add && eqv && tr[(size_t)col_bit * n + row - 1] != 0 && tr[(size_t)col_inf * n + row - 1] == 0;
    tr[(size_t)col_eq * n + row] = eq ? 1 : 0;
    tr[(size_t)col_u * n + row] = used ? 1 : 0;
    tr[(size_t)col_eqc * n + row] = eqc ? 1 : 0;
    tr[(size_t)col_ng * n + row] = ng ? 1 : 0;
    tr[(size_t)col_t1 * n + row] = t1 ? 1 : 0;
    tr[(size_t)col_v * n + row] = (bit && infc) ? 1 : 0;
    tr[(size_t)col_w * n + row] = (t1 && ng) ? 1 : 0;
    for (int j = 0; j < 16; j++) tr[(size_t)(col_ngv + j) * n + row] = (j == 0 && ng) ? 1 : 0;
    for (int j = 0; j < 15 * ext; j++) tr[(size_t)(col_cn + j) * n + row] = (ng && (carries >> j & 1)) ? 1 : 0;
    int first = -1;
    uint64_t dpx = 0, drx = 0;
    for (int j = 0; j < nc; j++) {
        const uint64_t px = tr[(size_t)(col_px + j) * n + row], rx = tr[(size_t)(col_rx + j) * n + row];
        if (first < 0 && px != rx) {
            first = j;
            dpx = px;
            drx = rx;
        }
    }
    const bool chord = is_add && used;
    const uint64_t w = (chord && first >= 0) ? gl::inv(gl::sub(dpx, drx)) : 0;
    for (int j = 0; j < nc; j++) tr[(size_t)(col_nz + j) * n + row] = (chord && j == first) ? w : 0;
    bad |= chord && first < 0;    // the x's equal where the chord is used: no witness (curve_rows refused the row already)
    // R = P on a block's LAST add row (bit 255 set): no row is left to hand the double over -- the constraint PER_LAST eqc = 0 cannot
    // hold, the record has no proof (oracle/air.c fill_curve_io refuses it at the same place)
    bad |= eqc && (row & 511) == 511;
    if (bad) atomicExch(err, SIPP_E_WITNESS);
}

// ---- exponent cells: closed form per row ----
__global__ void exp_rows_kernel(const uint32_t* __restrict__ ios, uint32_t ppi, uint32_t exp_off, uint64_t* __restrict__ tr,
                                size_t n, int col_bit, int col_e) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const uint32_t* ex = ios + (row >> 9) * ppi + exp_off;
    const uint32_t kk = row & 511, w = kk >> 6, s = kk & 63;
    const uint32_t consumed = (s + 1) >> 1;  // add rows before this row inside the 64-row window
    uint32_t e0 = consumed >= 32 ? 0u : (ex[w] >> consumed);
    tr[(size_t)col_bit * n + row] = (kk & 1) ? 0u : (e0 & 1u);
    tr[(size_t)col_e * n + row] = e0;
    for (uint32_t i = 1; i < 8; i++) tr[(size_t)(col_e + i) * n + row] = (w + i < 8) ? ex[w + i] : 0u;
}

__global__ void table_kernel(uint64_t* __restrict__ tr, size_t n, uint32_t tbits) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    uint64_t t = ((uint64_t)1 << tbits) - 1;
    tr[row] = row < t ? row : t;
}

// ---- generic gadget witnesses: one lane per row, integer interpreter over the AIR program ----
struct GadgetArgs {
    const int64_t* prog;
    const uint32_t* gadget_off;  // word offset of every gadget in prog
    int prog_len;
    int cpl;
    uint32_t p_limbs[16];
    uint32_t pinv16;
    // the AIR's selector columns (flags AIR_N_PERIODIC + k of a VEC): [n_vflag][2^log_rows] int8, values -1 / 0 / 1 (the pairing AIR)
    const int8_t* vflag;
    uint32_t rows_mask, log_rows;
    // a gadget of MANY products (the pairing AIR's ~300, a dozen of them live on a row): gadget_products_kernel sums them with one WAVE
    // per row -- prod_off[p] = word offset of product p, prod_off[np] = of what follows the products -- into esum [n][32]
    const uint32_t* prod_off;
    int64_t* esum;
    int np;
};

// limb vector with a STATIC number of limbs (16 for operands, 17 for the quotient: fixed by tools/air_gen.py): the limbs
// stay in registers.  Every limb of the specification is a small combination of 16-bit cells, far below 2^31; `ovf`
// is set if that ever fails, and the row is then reported as SIPP_E_WITNESS instead of being computed wrongly.
template <int NLV>
__device__ __forceinline__ int ivec_dev(const int64_t* w, const uint64_t* tr, size_t n, size_t row, const int* per, const GadgetArgs& g,
                                        int32_t (&out)[NLV], bool& ovf) {
    const int nt = (int)w[1];
    int64_t acc[NLV];
#pragma unroll
    for (int i = 0; i < NLV; i++) acc[i] = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t* tm = w + 2 + 5 * t;
        int32_t f = (int32_t)tm[0];
        const int base = (int)tm[1], stride = (int)tm[2], flag = (int)tm[3], neg = (int)tm[4];
        if (flag >= 0) {
            const int pv = flag < AIR_N_PERIODIC ? per[flag]
                                                 : (int)g.vflag[((size_t)(flag - AIR_N_PERIODIC) << g.log_rows) + (row & g.rows_mask)];
            f *= neg ? 1 - pv : pv;
        }
        if (f == 0) continue;
#pragma unroll
        for (int i = 0; i < NLV; i++) acc[i] += (int64_t)f * (int32_t)(uint32_t)tr[(size_t)(base + i * stride) * n + row];
    }
#pragma unroll
    for (int i = 0; i < NLV; i++) {
        out[i] = (int32_t)acc[i];
        ovf |= acc[i] != (int64_t)out[i];
    }
    return 2 + 5 * nt;
}

// one lane per (row, gadget): blockIdx.y selects the gadget.  All limb loops are static and unrolled (registers, no
// scratch); 16 x 16-bit operands keep every product inside one v_mad_i64_i32 / v_mad_u64_u32.
__global__ void __launch_bounds__(64) gadget_rows_kernel(GadgetArgs g, uint64_t* __restrict__ tr, size_t n,
                                                        int* __restrict__ err) {
    size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    int per[AIR_N_PERIODIC];
    for (int k = 0; k < AIR_N_PERIODIC; k++) per[k] = (int)(row % (size_t)AIR_PERIODIC[k][0]) == AIR_PERIODIC[k][1];
    const int64_t* w = g.prog + g.gadget_off[blockIdx.y];
    bool bad = false;
    const int sign_col = (int)w[1], cbase = (int)w[2], ncl = (int)w[3], lb = (int)w[4];
    const int64_t coffset = w[5];
    const int grp = (int)w[6];
    w += 7;
    const int q_base = (int)w[3];
    w += 2 + 5 * (int)w[1];
    int64_t e[32];
#pragma unroll
    for (int i = 0; i < 32; i++) e[i] = 0;
    int np = (int)*w++;
    if (g.esum) {                    // the products are summed already (gadget_products_kernel)
#pragma unroll
        for (int i = 0; i < 31; i++) e[i] = g.esum[row * 32 + (size_t)i];
        w = g.prog + g.prod_off[np];
        np = 0;
    }
    for (int p = 0; p < np; p++) {
        const int64_t coef = *w++;
        int32_t va[16], vb[16];
        w += ivec_dev<16>(w, tr, n, row, per, g, va, bad);
        bool live = false;
#pragma unroll
        for (int i = 0; i < 16; i++) live |= va[i] != 0;
        if (!live) {                 // a product another row type switches on (the pairing AIR: one of ~300 per row): nothing to add
            w += 2 + 5 * (int)w[1];
            continue;
        }
        w += ivec_dev<16>(w, tr, n, row, per, g, vb, bad);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int64_t a64 = coef * va[i];
            const int32_t ai = (int32_t)a64;
            bad |= a64 != (int64_t)ai;
#pragma unroll
            for (int j = 0; j < 16; j++) e[i + j] += (int64_t)ai * vb[j];
        }
    }
    const int nl = (int)*w++;
    for (int p = 0; p < nl; p++) {
        const int64_t coef = *w++;
        int32_t va[16];
        w += ivec_dev<16>(w, tr, n, row, per, g, va, bad);
#pragma unroll
        for (int i = 0; i < 16; i++) e[i] += coef * va[i];
    }
    // sign of E = sum e_k 2^(16 k) and the 16-bit limbs of |E| (E < 2^(16*40))
    int32_t limbs[40];
    int sign = 0;
    for (int pass = 0; pass < 2; pass++) {
        int64_t carry = 0;
#pragma unroll
        for (int k = 0; k < 40; k++) {
            const int64_t t = (k < 31 ? (sign ? -e[k] : e[k]) : 0) + carry;
            limbs[k] = (int32_t)(t & 0xffff);
            carry = t >> 16;
        }
        if (carry == 0) break;
        if (carry == -1 && pass == 0) {
            sign = 1;
            continue;
        }
        bad = true;
        break;
    }
    // q = |E| / p by exact division from the low end (p odd: q_i = limb_i * p^-1 mod 2^16)
    uint32_t q[17];
#pragma unroll
    for (int i = 0; i < 17; i++) {
        q[i] = ((uint32_t)limbs[i] * g.pinv16) & 0xffffu;
        int64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 40; j++) {
            if (i + j < 40) {
                const int64_t t = (int64_t)limbs[i + j] - (j < 16 ? (int64_t)((uint64_t)q[i] * g.p_limbs[j]) : 0) + carry;
                limbs[i + j] = (int32_t)(t & 0xffff);
                carry = t >> 16;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 40; k++) bad |= limbs[k] != 0;
    tr[(size_t)sign_col * n + row] = (uint64_t)sign;
#pragma unroll
    for (int i = 0; i < 17; i++) {
        if (g.cpl == 1) {
            tr[(size_t)(q_base + i) * n + row] = q[i];
        } else {
            tr[(size_t)(q_base + 2 * i) * n + row] = q[i] & 0xffu;
            tr[(size_t)(q_base + 2 * i + 1) * n + row] = q[i] >> 8;
        }
    }
    // d_k = e_k -+ (q * p)_k
    int64_t d[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
        uint64_t qp = 0;
#pragma unroll
        for (int i = 0; i < 17; i++) {
            const int j = k - i;
            if (j >= 0 && j < 16) qp += (uint64_t)q[i] * g.p_limbs[j];
        }
        d[k] = (k < 31 ? e[k] : 0) - (sign ? -(int64_t)qp : (int64_t)qp);
    }
    // carries in base 2^(16 grp): sum_t 2^(16t) d_{grp m + t} - c_{m-1} + 2^(16 grp) c_m = 0
    __int128 cprev = 0;
    const int nmm = 32 / grp;
#pragma unroll
    for (int m = 0; m < 32; m++) {
        if (m < nmm) {
            __int128 dm = grp == 2 ? (__int128)d[(2 * m + 1) & 31] * 65536 + d[(2 * m) & 31] : (__int128)d[m];
            dm -= cprev;
            const __int128 mask = (((__int128)1) << (16 * grp)) - 1;
            bad |= (dm & mask) != 0;
            const __int128 ck = -(dm >> (16 * grp));
            if (m == nmm - 1) {
                bad |= ck != 0;
            } else {
                const __int128 v = ck + coffset;
                bad |= v < 0 || (v >> (ncl * lb)) != 0;
                const uint64_t vv = (uint64_t)v;
                for (int l = 0; l < ncl; l++)
                    tr[(size_t)(cbase + m * ncl + l) * n + row] = (vv >> (lb * l)) & (((uint64_t)1 << lb) - 1);
                cprev = ck;
            }
        }
    }
    if (bad) atomicExch(err, SIPP_E_WITNESS);
}

// one WAVE per row: lane l takes the products l, l + 64, ... of the row's (single) gadget -- almost all of them are switched off by the
// row's selectors and cost a flag lookup --, the 31 coefficient sums are added up across the wave
__global__ void __launch_bounds__(64) gadget_products_kernel(GadgetArgs g, const uint64_t* __restrict__ tr, size_t n, int* __restrict__ err) {
    const size_t row = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    int per[AIR_N_PERIODIC];
    for (int k = 0; k < AIR_N_PERIODIC; k++) per[k] = (int)(row % (size_t)AIR_PERIODIC[k][0]) == AIR_PERIODIC[k][1];
    bool bad = false;
    int64_t e[31];
#pragma unroll
    for (int i = 0; i < 31; i++) e[i] = 0;
    for (int p = (int)lane; p < g.np; p += 64) {
        const int64_t* w = g.prog + g.prod_off[p];
        const int64_t coef = *w++;
        int32_t va[16], vb[16];
        w += ivec_dev<16>(w, tr, n, row, per, g, va, bad);
        bool live = false;
#pragma unroll
        for (int i = 0; i < 16; i++) live |= va[i] != 0;
        if (!live) continue;
        (void)ivec_dev<16>(w, tr, n, row, per, g, vb, bad);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int64_t a64 = coef * va[i];
            const int32_t ai = (int32_t)a64;
            bad |= a64 != (int64_t)ai;
#pragma unroll
            for (int j = 0; j < 16; j++) e[i + j] += (int64_t)ai * vb[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 31; i++) {
        long long v = e[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
        e[i] = v;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 31; i++) g.esum[row * 32 + (size_t)i] = e[i];
    }
    if (bad) atomicExch(err, SIPP_E_WITNESS);
}

// ---- permuted lookup columns ----
__global__ void hist_kernel(const uint64_t* __restrict__ cols, size_t n, uint32_t ncols, uint32_t tbits,
                            uint32_t* __restrict__ hist) {
    size_t total = n * ncols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t c = i / n;
        uint64_t v = cols[i];
        atomicAdd(&hist[(c << tbits) + (v & (((uint64_t)1 << tbits) - 1))], 1u);
    }
}

// u16 table: the 65,536 bins do not fit LDS as u32, so one block counts ONE QUARTER of the value range (16,384 bins, 64 KiB) of one
// (column, chunk of rows); wave-uniform values (constant cells such as high carry limbs) take a single LDS atomic per wave instead of
// 64 serialised ones.  (Halves -- 128 KiB per block -- until round 4: such a block finds no CU while another proof's leaf-hash launch is
// resident, 22.5 KB of LDS per block and two or three blocks per CU: G1's histogram waited 12.7 ms for 0.3 ms of work behind G2's leaf
// hashing, 382 ms at n = 4096.  A quarter fits beside them; the column is read four times instead of twice: 0.5 MB each.)
constexpr uint32_t HIST16_BINS = 16384, HIST16_PARTS = 65536 / HIST16_BINS, HIST16_SHIFT = 14;
__global__ void __launch_bounds__(1024) hist_u16_kernel(const uint64_t* __restrict__ cols, size_t n, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t hs[];
    for (uint32_t i = threadIdx.x; i < HIST16_BINS; i += 1024) hs[i] = 0;
    __syncthreads();
    const size_t c = blockIdx.z;
    const uint32_t part = blockIdx.y;
    const uint64_t* col = cols + c * n;
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (size_t i0 = lo; i0 < hi; i0 += 1024) {
        const size_t i = i0 + threadIdx.x;
        const bool live = i < hi;
        const uint32_t v = live ? (uint32_t)col[i] & 0xffffu : 0xffffffffu;
        const uint32_t v0 = __builtin_amdgcn_readfirstlane(v);
        if (__all(v == v0)) {
            if (v0 != 0xffffffffu && (v0 >> HIST16_SHIFT) == part && (threadIdx.x & 63) == 0) atomicAdd(&hs[v0 & (HIST16_BINS - 1)], 64u);
        } else if (live && (v >> HIST16_SHIFT) == part) {
            atomicAdd(&hs[v & (HIST16_BINS - 1)], 1u);
        }
    }
    __syncthreads();
    uint32_t* out = hist + (c << 16) + ((size_t)part << HIST16_SHIFT);
    for (uint32_t i = threadIdx.x; i < HIST16_BINS; i += 1024) {
        const uint32_t x = hs[i];
        if (x) {
            if (gridDim.x == 1) out[i] = x;     // this block owns the bins (hist is zero-initialised)
            else atomicAdd(&out[i], x);
        }
    }
}

// u8 table: LDS-privatised histogram, one block per (column, chunk of rows)
__global__ void __launch_bounds__(256) hist_u8_kernel(const uint64_t* __restrict__ cols, size_t n, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t c = blockIdx.y;
    const uint64_t* col = cols + c * n;
    const size_t chunk = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) atomicAdd(&h[col[i] & 0xff], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[(c << 8) + threadIdx.x], h[threadIdx.x]);
}

// one block per column: start[v] = #values < v ; dist[v] = #distinct values <= v ; zlist = values with hist == 0 ascending
__global__ void __launch_bounds__(256) lookup_scan_kernel(const uint32_t* __restrict__ hist, uint32_t tbits,
                                                         uint32_t* __restrict__ start, uint32_t* __restrict__ dist,
                                                         uint32_t* __restrict__ zlist, uint32_t* __restrict__ nzero) {
    __shared__ uint32_t s_cnt[256], s_dst[256], s_zer[256];
    const uint32_t T = 1u << tbits, per = T / 256;
    const size_t c = blockIdx.x;
    const uint32_t* h = hist + (c << tbits);
    uint32_t* st = start + (c << tbits);
    uint32_t* ds = dist + (c << tbits);
    uint32_t* zl = zlist + (c << tbits);
    const uint32_t v0 = threadIdx.x * per;
    uint32_t cnt = 0, dst = 0, zer = 0;
    // a lane owns `per` consecutive bins (256 for the u16 table): 16-byte vector accesses, four bins at a time
    if (per % 4 == 0) {
        const uint4* h4 = reinterpret_cast<const uint4*>(h + v0);
#pragma unroll 8
        for (uint32_t k = 0; k < per / 4; k++) {
            const uint4 x = h4[k];
            cnt += x.x + x.y + x.z + x.w;
            const uint32_t nz = (x.x != 0) + (x.y != 0) + (x.z != 0) + (x.w != 0);
            dst += nz;
            zer += 4 - nz;
        }
    } else {
        for (uint32_t v = v0; v < v0 + per; v++) {
            uint32_t x = h[v];
            cnt += x;
            dst += x != 0;
            zer += x == 0;
        }
    }
    s_cnt[threadIdx.x] = cnt;
    s_dst[threadIdx.x] = dst;
    s_zer[threadIdx.x] = zer;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = 0, b = 0, z = 0;
        for (int i = 0; i < 256; i++) {
            uint32_t ta = s_cnt[i], tb = s_dst[i], tz = s_zer[i];
            s_cnt[i] = a;
            s_dst[i] = b;
            s_zer[i] = z;
            a += ta;
            b += tb;
            z += tz;
        }
        nzero[c] = z;
    }
    __syncthreads();
    cnt = s_cnt[threadIdx.x];
    dst = s_dst[threadIdx.x];
    zer = s_zer[threadIdx.x];
    if (per % 4 == 0) {
        const uint4* h4 = reinterpret_cast<const uint4*>(h + v0);
        uint4* st4 = reinterpret_cast<uint4*>(st + v0);
        uint4* ds4 = reinterpret_cast<uint4*>(ds + v0);
#pragma unroll 4
        for (uint32_t k = 0; k < per / 4; k++) {
            const uint4 x = h4[k];
            const uint32_t v = v0 + 4 * k;
            uint4 so, dd;
            so.x = cnt; cnt += x.x; dst += x.x != 0; dd.x = dst; if (x.x == 0) zl[zer++] = v;
            so.y = cnt; cnt += x.y; dst += x.y != 0; dd.y = dst; if (x.y == 0) zl[zer++] = v + 1;
            so.z = cnt; cnt += x.z; dst += x.z != 0; dd.z = dst; if (x.z == 0) zl[zer++] = v + 2;
            so.w = cnt; cnt += x.w; dst += x.w != 0; dd.w = dst; if (x.w == 0) zl[zer++] = v + 3;
            st4[k] = so;
            ds4[k] = dd;
        }
    } else {
        for (uint32_t v = v0; v < v0 + per; v++) {
            uint32_t x = h[v];
            st[v] = cnt;
            cnt += x;
            dst += x != 0;
            ds[v] = dst;
            if (x == 0) zl[zer++] = v;
        }
    }
}

// The permuted input / table columns by EXPANSION (round 3; a binary search per position took 16 dependent loads, 0.57 ms of G2's
// exposed head): one lane per (column, value v) writes v to the positions start[v] .. start[v] + hist[v] - 1 -- the first of them carries v in the table
// column too, the others a filler (the next unused table value from zlist, T - 1 when those run out).  No dependent loads (the search takes 16 per position), writes in increasing position order.
// Bins of more than 8 entries (constant cells, carries; every bin of a u8 table) are written by the whole wave, one bin after the other.
__global__ void __launch_bounds__(256) lookup_expand_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ start,
                                                           const uint32_t* __restrict__ dist, const uint32_t* __restrict__ zlist,
                                                           const uint32_t* __restrict__ nzero, size_t n, uint32_t tbits,
                                                           uint64_t* __restrict__ pin, uint64_t* __restrict__ ptab) {
    const uint32_t T = 1u << tbits;
    const size_t c = blockIdx.y;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = v < T;
    const size_t tb = (c << tbits);
    const uint32_t cnt = live ? hist[tb + v] : 0, s = live ? start[tb + v] : 0, d = live ? dist[tb + v] : 0;
    const uint32_t nz = nzero[c];
    const uint32_t* zl = zlist + tb;
    uint64_t* pi = pin + c * n;
    uint64_t* pt = ptab + c * n;
    const uint32_t LIGHT = 8;
    if (cnt && cnt <= LIGHT) {
        pi[s] = v;
        pt[s] = v;
        for (uint32_t j = 1; j < cnt; j++) {
            const uint32_t i = s + j, k = i - d;
            pi[i] = v;
            pt[i] = k < nz ? zl[k] : T - 1;
        }
    }
    uint64_t heavy = __ballot(cnt > LIGHT);
    const uint32_t lane = threadIdx.x & 63;
    while (heavy) {
        const int src = __ffsll((unsigned long long)heavy) - 1;
        heavy &= heavy - 1;
        const uint32_t hv = __shfl(v, src), hs = __shfl(s, src), hc = __shfl(cnt, src), hd = __shfl(d, src);
        // eight positions per lane and trip: the fillers' loads are in flight together (a constant column is ONE bin of n entries)
        uint32_t j = lane;
        for (; j + 7 * 64 < hc; j += 8 * 64) {
            uint32_t f[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t k = hs + j + 64 * q - hd;
                f[q] = k < nz ? zl[k] : T - 1;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t i = hs + j + 64 * q;
                pi[i] = hv;
                pt[i] = (j + 64 * q) == 0 ? hv : f[q];
            }
        }
        for (; j < hc; j += 64) {
            const uint32_t i = hs + j, k = i - hd;
            pi[i] = hv;
            pt[i] = j == 0 ? hv : (k < nz ? zl[k] : T - 1);
        }
    }
}

}  // namespace

// ---- host drivers -------------------------------------------------------------------------------------------
// API kinds: 0 G1, 1 G2, 2 Fq12, 3 MapToG2, 4 / 5 = the hardened G1 / G2 AIRs (the same records and rows; air->kind stays 0 / 1),
// 6 = the final pairing
const air_spec_t* sipp_air_get(int kind, uint32_t log_n) {
    const bool u16 = log_n >= 16;
    if (kind < 0 || kind > 6) return nullptr;
    const int hard = (kind == 4 || kind == 5) ? 1 : 0, base = hard ? kind - 4 : kind;
    for (size_t i = 0; i < sizeof(AIR_AIRS) / sizeof(AIR_AIRS[0]); i++)
        if (AIR_AIRS[i].kind == base && AIR_AIRS[i].hardened == hard && (AIR_AIRS[i].table_bits == 16) == u16) return &AIR_AIRS[i];
    return nullptr;
}

static int64_t* prog_on_device(sipp_ctx* ctx, const air_spec_t* a) {
    uint64_t* t = sipp_table_get(ctx, 100, (uint64_t)(a->kind + 8 * a->hardened), (uint64_t)a->table_bits);
    if (t) return (int64_t*)t;
    std::vector<uint64_t> v(a->prog_len);
    memcpy(v.data(), a->prog, (size_t)a->prog_len * 8);
    if (sipp_table_put(ctx, 100, (uint64_t)(a->kind + 8 * a->hardened), (uint64_t)a->table_bits, v, &t) != SIPP_OK) return nullptr;
    return (int64_t*)t;
}
const int64_t* sipp_air_prog_device(sipp_ctx* ctx, const air_spec_t* a) { return prog_on_device(ctx, a); }
const int8_t* sipp_air_vflag_device(sipp_ctx* ctx, const air_spec_t* a) {
    if (a->n_vflag == 0) return nullptr;
    uint64_t* t = sipp_table_get(ctx, 103, (uint64_t)a->kind, 0);
    if (!t) {
        const size_t bytes = (size_t)a->n_vflag << a->log_rows;
        std::vector<uint64_t> v((bytes + 7) / 8, 0);
        int8_t* f = reinterpret_cast<int8_t*>(v.data());       // flag k on row r: the weights of its terms the row descriptor meets
        for (int k = 0; k < a->n_vflag; k++)
            for (int r = 0; r < (1 << a->log_rows); r++) f[((size_t)k << a->log_rows) + (size_t)r] = (int8_t)air_vper_value(a, k, r);
        if (sipp_table_put(ctx, 103, (uint64_t)a->kind, 0, v, &t) != SIPP_OK) return nullptr;
    }
    return reinterpret_cast<const int8_t*>(t);
}

// bytes of the Jacobian row scratch sipp_trace_fill takes from the arena FIRST for a curve AIR (kind 0 / 1) of 2^log_n rows
size_t sipp_curve_rows_bytes(int kind, uint32_t log_n) {
    return ((size_t)1 << log_n) * (kind == 0 ? sizeof(RowPts<1>) : sizeof(RowPts<2>));
}

// ---- the folds of a native SIPP round in two phases (native.hip) ----------------------------------------------------------
// A' = A1 + [x] A2 and B' = B1 + [1/x] B2 need the challenge x only for the SELECTION of the powers 2^k A2 / 2^k B2, not for
// the 255 doublings that produce them: phase A (the doubling chains, 1.8 ms of pure latency) therefore runs while the round's
// pairing products -- which determine x -- are still being computed; phase B (scan over the exponent bits, outputs) follows.
size_t sipp_fold_rows_bytes(int kind, uint32_t num_io) {
    return (size_t)num_io * 512 * (kind == 0 ? sizeof(RowPts<1>) : sizeof(RowPts<2>));
}
int sipp_fold_chain_begin(sipp_ctx* ctx, int kind, const uint32_t* d_ios, uint32_t num_io, uint32_t ppi, void* rows, hipStream_t st) {
    // (no ProfScope: these launches go to a side stream, the event brackets belong to the main one)
    if (kind == 0)
        hipLaunchKernelGGL((curve_dbl_par_kernel<1, 4>), dim3((num_io + 15) / 16), dim3(64), 0, st, d_ios, num_io, ppi,
                           reinterpret_cast<RowPts<1>*>(rows));
    else
        hipLaunchKernelGGL((curve_dbl_par_kernel<2, 8>), dim3((num_io + 7) / 8), dim3(64), 0, st, d_ios, num_io, ppi,
                           reinterpret_cast<RowPts<2>*>(rows));
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}
int sipp_fold_chain_finish(sipp_ctx* ctx, int kind, uint32_t* d_ios, uint32_t num_io, uint32_t ppi, void* rows, int* d_err,
                           hipStream_t st) {
    if (kind == 0) {
        hipLaunchKernelGGL(curve_scan_kernel<1>, dim3(num_io), dim3(256), 0, st, d_ios, num_io, ppi, reinterpret_cast<RowPts<1>*>(rows));
        hipLaunchKernelGGL(curve_outputs_kernel<1>, dim3((num_io + 63) / 64), dim3(64), 0, st, reinterpret_cast<RowPts<1>*>(rows), d_ios,
                           num_io, ppi, d_err);
    } else {
        hipLaunchKernelGGL(curve_scan_kernel<2>, dim3(num_io), dim3(256), 0, st, d_ios, num_io, ppi, reinterpret_cast<RowPts<2>*>(rows));
        hipLaunchKernelGGL(curve_outputs_kernel<2>, dim3((num_io + 63) / 64), dim3(64), 0, st, reinterpret_cast<RowPts<2>*>(rows), d_ios,
                           num_io, ppi, d_err);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

// d_ios: [num_io][pi_per_io] u32 on the device (padded); d_trace: [W][n] zero-initialised by the caller is NOT
// required: every main column is written here.  d_err: device int, 0 on entry.
// a gated proof of an instance (stark.hip: G1 behind G2's trace fill) has its thin doubling / scan
// chain -- 16 waves, 1.5 ms of pure latency -- in flight while it waits; everything wider starts after the gate
static void gate_after_chain(sipp_ctx* ctx) {
    if (ctx->gate_wait) {
        ctx->gate_wait->wait();
        ctx->gate_wait = nullptr;
    }
}

int sipp_trace_fill(sipp_ctx* ctx, const air_spec_t* a, const uint32_t* d_ios, uint32_t num_io, uint32_t log_n,
                    uint64_t* d_trace, int* d_err) {
    const size_t n = (size_t)1 << log_n;
    const int cpl = a->cells_per_limb, nm = a->n_main, nc = a->n_checked;
    ArenaScope scope(ctx);   // the row scratch and the lookup tables go back on EVERY exit path
    int col_bit = 0, col_e = 0, exp_off = 0;
    const bool exp_air = a->kind != 3 && a->kind != 6;   // the exponentiation AIRs: exponent cells, an accumulator compared with the record
    if (a->kind == 3) {
        // MapToG2: eight rows per message, no exponent / accumulator cells (mapg2.hip); the claimed point is compared there
        if (ctx->outputs_only) {
            SIPP_TRY(sipp_mapg2_outputs(ctx, const_cast<uint32_t*>(d_ios), num_io, d_err));
            return SIPP_OK;
        }
        SIPP_TRY(sipp_mapg2_fill(ctx, a, d_ios, num_io, log_n, d_trace, d_err));
    } else if (a->kind == 6) {
        // the final pairing: one wave per record walks the schedule (pairing.hip); the claimed value is compared there
        SIPP_TRY(sipp_pairing_fill(ctx, a, d_ios, num_io, log_n, d_trace, d_err));
        if (ctx->outputs_only) return SIPP_OK;
    } else if (a->kind == 2) {
        Fq12Cols c{1, 1 + 192, a->checked_base, cpl};
        col_bit = 1 + 384;
        col_e = col_bit + 1;
        exp_off = 192;
        ProfScope ps(ctx, "trace_fq12_chain");
        hipLaunchKernelGGL(fq12_chain_kernel, dim3(num_io), dim3(384), 0, ctx->stream, d_ios, num_io, d_trace, n, c);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    } else {
        const int ext = a->kind == 0 ? 1 : 2, ncl = 16 * ext;
        const int32_t* hard_lay = !a->hardened ? nullptr
                                  : a->kind == 0 ? (cpl == 1 ? AIR_HARD_LAYOUT_G1H_U16 : AIR_HARD_LAYOUT_G1H_U8)
                                                 : (cpl == 1 ? AIR_HARD_LAYOUT_G2H_U16 : AIR_HARD_LAYOUT_G2H_U8);
        const int hard_inf_col = hard_lay ? hard_lay[7] : 0;
        CurveCols c{1, 1 + ncl, 1 + 2 * ncl, 1 + 3 * ncl, a->checked_base, a->checked_base + ncl * cpl,
                    a->checked_base + 2 * ncl * cpl, cpl};
        col_bit = 1 + 4 * ncl;
        col_e = col_bit + 1;
        exp_off = 32 * ext;
        if (ext == 1) {
            RowPts<1>* rows = arena_alloc_t<RowPts<1>>(ctx, n);
            if (!rows) return SIPP_E_NOMEM;
            {
                ProfScope ps(ctx, "trace_curve_chain");
                hipLaunchKernelGGL((curve_dbl_par_kernel<1, 4>), dim3((num_io + 15) / 16), dim3(64), 0, ctx->stream, d_ios, num_io,
                                   (uint32_t)a->pi_per_io, rows);
                hipLaunchKernelGGL(curve_scan_kernel<1>, dim3(num_io), dim3(256), 0, ctx->stream, d_ios, num_io,
                                   (uint32_t)a->pi_per_io, rows);
            }
            SIPP_CHECK_HIP(ctx, hipGetLastError());
            gate_after_chain(ctx);
            if (ctx->outputs_only) {
                hipLaunchKernelGGL(curve_outputs_kernel<1>, dim3((num_io + 63) / 64), dim3(64), 0, ctx->stream, rows,
                                   const_cast<uint32_t*>(d_ios), num_io, (uint32_t)a->pi_per_io, d_err);
                SIPP_CHECK_HIP(ctx, hipGetLastError());
                return SIPP_OK;
            }
            ProfScope ps(ctx, "trace_curve_rows");
            hipLaunchKernelGGL(curve_rows_kernel<1>, dim3((unsigned)(n / ROWS_BLOCK)), dim3(ROWS_BLOCK), 0, ctx->stream, rows,
                               d_trace, n, c, a->hardened, hard_inf_col, d_err);
            SIPP_CHECK_HIP(ctx, hipGetLastError());
        } else {
            RowPts<2>* rows = arena_alloc_t<RowPts<2>>(ctx, n);
            if (!rows) return SIPP_E_NOMEM;
            {
                ProfScope ps(ctx, "trace_curve_chain");
                hipLaunchKernelGGL((curve_dbl_par_kernel<2, 8>), dim3((num_io + 7) / 8), dim3(64), 0, ctx->stream, d_ios, num_io,
                                   (uint32_t)a->pi_per_io, rows);
                hipLaunchKernelGGL(curve_scan_kernel<2>, dim3(num_io), dim3(256), 0, ctx->stream, d_ios, num_io,
                                   (uint32_t)a->pi_per_io, rows);
            }
            SIPP_CHECK_HIP(ctx, hipGetLastError());
            gate_after_chain(ctx);
            if (ctx->outputs_only) {
                hipLaunchKernelGGL(curve_outputs_kernel<2>, dim3((num_io + 63) / 64), dim3(64), 0, ctx->stream, rows,
                                   const_cast<uint32_t*>(d_ios), num_io, (uint32_t)a->pi_per_io, d_err);
                SIPP_CHECK_HIP(ctx, hipGetLastError());
                return SIPP_OK;
            }
            ProfScope ps(ctx, "trace_curve_rows");
            hipLaunchKernelGGL(curve_rows_kernel<2>, dim3((unsigned)(n / ROWS_BLOCK)), dim3(ROWS_BLOCK), 0, ctx->stream, rows,
                               d_trace, n, c, a->hardened, hard_inf_col, d_err);
            SIPP_CHECK_HIP(ctx, hipGetLastError());
        }
    }
    if (ctx->outputs_only) {
        // sipp_exp_outputs: the accumulator cells of every block's last row are the outputs; nothing else is needed
        hipLaunchKernelGGL(write_outputs_kernel, dim3((num_io + 63) / 64), dim3(64), 0, ctx->stream, const_cast<uint32_t*>(d_ios),
                           num_io, (uint32_t)a->pi_per_io, a->kind, d_trace, n, 1);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        return SIPP_OK;
    }
    if (exp_air) {
        // accumulator state columns: R (curves, column 1) or acc (Fq12, column 1)
        ProfScope ps(ctx, "trace_check_outputs");
        hipLaunchKernelGGL(check_outputs_kernel, dim3((num_io + 63) / 64), dim3(64), 0, ctx->stream, d_ios, num_io,
                           (uint32_t)a->pi_per_io, a->kind, d_trace, n, 1, d_err);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    {
        ProfScope ps(ctx, "trace_exp_table");
        if (exp_air)
            hipLaunchKernelGGL(exp_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_ios,
                               (uint32_t)a->pi_per_io, (uint32_t)exp_off, d_trace, n, col_bit, col_e);
        hipLaunchKernelGGL(table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_trace, n,
                           (uint32_t)a->table_bits);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    if (a->hardened) {
        if (a->kind > 1) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "hardened AIR: curves only");
        const int ext = a->kind + 1, ncl = 16 * ext;
        const int32_t* lay = a->kind == 0 ? (cpl == 1 ? AIR_HARD_LAYOUT_G1H_U16 : AIR_HARD_LAYOUT_G1H_U8)
                                          : (cpl == 1 ? AIR_HARD_LAYOUT_G2H_U16 : AIR_HARD_LAYOUT_G2H_U8);
        SippBnP pl;
        for (int i = 0; i < 16; i++) pl.l[i] = AIR_BN_P_LIMBS[i];
        ProfScope ps(ctx, "trace_harden");
        hipLaunchKernelGGL(harden_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_trace, n, ext, cpl, 1,
                           1 + 2 * ncl, col_bit, a->checked_base + ncl * cpl, lay[0], lay[1], lay[2], lay[3], lay[4], lay[5], lay[6], lay[7], lay[8], lay[9], lay[10], lay[11], lay[12],
                           pl, d_err);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    {
        GadgetArgs g;
        g.prog = prog_on_device(ctx, a);
        if (!g.prog) return SIPP_E_HIP;
        {
            uint64_t* t = sipp_table_get(ctx, 101, (uint64_t)(a->kind + 8 * a->hardened), (uint64_t)a->table_bits);
            if (!t) {
                // gadgets come first in the program: record where each one starts
                std::vector<uint32_t> off;
                const int64_t* w = a->prog;
                const int64_t* end = a->prog + a->prog_len;
                while (w < end && w[0] == 1) {
                    off.push_back((uint32_t)(w - a->prog));
                    w += 7;
                    w += 2 + 5 * w[1];
                    int64_t np = *w++;
                    for (int64_t p = 0; p < np; p++) {
                        w += 1;
                        w += 2 + 5 * w[1];
                        w += 2 + 5 * w[1];
                    }
                    int64_t nl = *w++;
                    for (int64_t p = 0; p < nl; p++) {
                        w += 1;
                        w += 2 + 5 * w[1];
                    }
                }
                if ((int)off.size() != a->n_gadgets) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "AIR program: gadget count mismatch");
                std::vector<uint64_t> packed((off.size() + 1) / 2 + 1, 0);
                memcpy(packed.data(), off.data(), off.size() * 4);
                SIPP_TRY(sipp_table_put(ctx, 101, (uint64_t)(a->kind + 8 * a->hardened), (uint64_t)a->table_bits, packed, &t));
            }
            g.gadget_off = reinterpret_cast<const uint32_t*>(t);
        }
        g.prog_len = a->prog_len;
        g.cpl = cpl;
        for (int i = 0; i < 16; i++) g.p_limbs[i] = AIR_BN_P_LIMBS[i];
        uint32_t pinv = 1;
        for (int i = 0; i < 5; i++) pinv = (pinv * (2 - AIR_BN_P_LIMBS[0] * pinv)) & 0xffff;
        g.pinv16 = pinv;
        g.vflag = sipp_air_vflag_device(ctx, a);
        if (a->n_vflag && !g.vflag) return SIPP_E_HIP;
        g.log_rows = (uint32_t)a->log_rows;
        g.rows_mask = (1u << a->log_rows) - 1;
        ProfScope ps(ctx, "trace_gadgets");
        g.prod_off = nullptr;
        g.esum = nullptr;
        g.np = 0;
        if (a->n_gadgets == 1) {
            // the one gadget's product offsets (table 107); more than 64 products: a wave per row sums them first
            uint64_t* pt = sipp_table_get(ctx, 107, (uint64_t)(a->kind + 8 * a->hardened), (uint64_t)a->table_bits);
            if (!pt) {
                std::vector<uint32_t> po;
                const int64_t* w = a->prog + 7;
                w += 2 + 5 * w[1];
                const int64_t np = *w++;
                for (int64_t p = 0; p < np; p++) {
                    po.push_back((uint32_t)(w - a->prog));
                    w += 1;
                    w += 2 + 5 * w[1];
                    w += 2 + 5 * w[1];
                }
                po.push_back((uint32_t)(w - a->prog));
                std::vector<uint64_t> packed(po.size() / 2 + 2, 0);
                packed[0] = (uint64_t)np;
                memcpy(packed.data() + 1, po.data(), po.size() * 4);
                SIPP_TRY(sipp_table_put(ctx, 107, (uint64_t)(a->kind + 8 * a->hardened), (uint64_t)a->table_bits, packed, &pt));
            }
            // (the count sits in the device table; re-derive it on the host from the program: cheap)
            const int64_t* w = a->prog + 7;
            w += 2 + 5 * w[1];
            const int np = (int)*w;
            if (np > 64) {
                g.esum = arena_alloc_t<int64_t>(ctx, n * 32);
                if (!g.esum) return SIPP_E_NOMEM;
                g.prod_off = reinterpret_cast<const uint32_t*>(pt + 1);
                g.np = np;
                hipLaunchKernelGGL(gadget_products_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, g, d_trace, n, d_err);
            }
        }
        hipLaunchKernelGGL(gadget_rows_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)a->n_gadgets), dim3(64), 0, ctx->stream, g, d_trace, n, d_err);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    {
        // permuted lookup columns for all checked columns at once
        const uint32_t tb = (uint32_t)a->table_bits;
        const size_t T = (size_t)1 << tb;
        uint32_t* hist = arena_alloc_t<uint32_t>(ctx, (size_t)nc * T);
        uint32_t* start = arena_alloc_t<uint32_t>(ctx, (size_t)nc * T);
        uint32_t* dist = arena_alloc_t<uint32_t>(ctx, (size_t)nc * T);
        uint32_t* zlist = arena_alloc_t<uint32_t>(ctx, (size_t)nc * T);
        uint32_t* nzero = arena_alloc_t<uint32_t>(ctx, (size_t)nc);
        if (!hist || !start || !dist || !zlist || !nzero) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)nc * T * 4, ctx->stream));
        const uint64_t* cols = d_trace + (size_t)a->checked_base * n;
        {
            ProfScope ps(ctx, "lookup_hist");
            if (tb == 8) {
                const unsigned chunks = (unsigned)std::max<size_t>(1, std::min<size_t>(16, n / 2048));
                hipLaunchKernelGGL(hist_u8_kernel, dim3(chunks, (unsigned)nc), dim3(256), 0, ctx->stream, cols, n, hist);
            } else if (tb == 16) {
                const unsigned chunks = (unsigned)std::max<size_t>(1, std::min<size_t>(8, n >> 17));
                hipLaunchKernelGGL(hist_u16_kernel, dim3(chunks, HIST16_PARTS, (unsigned)nc), dim3(1024), HIST16_BINS * sizeof(uint32_t), ctx->stream,
                                   cols, n, hist);
            } else {
                hipLaunchKernelGGL(hist_kernel, dim3(4096), dim3(256), 0, ctx->stream, cols, n, (uint32_t)nc, tb, hist);
            }
        }
        {
            ProfScope ps(ctx, "lookup_scan");
            hipLaunchKernelGGL(lookup_scan_kernel, dim3((unsigned)nc), dim3(256), 0, ctx->stream, hist, tb, start, dist, zlist, nzero);
        }
        {
            ProfScope ps(ctx, "lookup_fill");
            hipLaunchKernelGGL(lookup_expand_kernel, dim3((unsigned)((T + 255) / 256), (unsigned)nc), dim3(256), 0, ctx->stream, hist,
                               start, dist, zlist, nzero, n, tb, d_trace + (size_t)nm * n, d_trace + (size_t)(nm + nc) * n);
        }
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    return SIPP_OK;
}
