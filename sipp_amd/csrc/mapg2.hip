// sipp_amd/csrc/mapg2.hip -- the map Fp2 -> E'(Fp2) in front of SIPP's BLS example, on the device: native values and the
// primary witness of the MapToG2 AIR (tools/air_gen.py::build_map_g2, eight trace rows per message).
//
// Replaces plonky2_bn254::curves::map_to_g2::map_to_g2_without_cofactor_mul (reference src/bin/bls_aggregation.rs:21, :102) and
// the trace generator behind starky_bn254's batch_map_to_g2_circuit (:31, :65); neither crate is under /root/reference, so
// what is built is the published algorithm they are recalled to follow: the Shallue - van de Woestijne map of RFC 9380
// (appendix F.1) with Z = 1, sgn0 of section 4.1, is_square(0) = true (PARITY UNPINNED; oracle/mapg2.c and
// oracle/py/map_to_g2.py are the two CPU readings the tests compare with).
//
// Per message ~20 Fp2 products, one Fp2 inversion and five candidate Fp2 square roots (two or three Fp powers each, by the norm), on
// eight lanes (mapg2_rows_kernel).  Every witness root is THE root with sgn0 = 0, so the cells do not depend on how a root was
// found.  Latency-bound scalar-style code on few waves; the field product is kept out of line so that everything shares one copy.
#include "air_tables.h"
#include "ctx.hpp"
#include "fq.hpp"
#include "mapg2_constants.h"

namespace {

using fq::Fq;
using fq::Fq2;

__device__ __noinline__ Fq mmul(const Fq& a, const Fq& b) { return fq::mul(a, b); }
__device__ __forceinline__ Fq msqr(const Fq& a) { return mmul(a, a); }
__device__ __forceinline__ bool feq(const Fq& a, const Fq& b) { return fq::is_zero(fq::sub(a, b)); }

__device__ __forceinline__ Fq2 m2(const Fq2& a, const Fq2& b) {
    const Fq t0 = mmul(a.c0, b.c0), t1 = mmul(a.c1, b.c1);
    const Fq s = mmul(fq::add(a.c0, a.c1), fq::add(b.c0, b.c1));
    return Fq2{fq::sub(t0, t1), fq::sub(fq::sub(s, t0), t1)};
}
__device__ __forceinline__ Fq2 neg2(const Fq2& a) { return Fq2{fq::neg(a.c0), fq::neg(a.c1)}; }

// a^((p-3)/4); times a: a root of a whenever a is a square (p = 3 mod 4)
__device__ __noinline__ Fq pow_pm3d4(const Fq& a) {
    constexpr uint32_t E[8] = {0xb61f3f51u, 0x4f082305u, 0x5a1c72a3u, 0x65e05aa4u, 0xa0605617u, 0x6e14116du, 0xb84c680au, 0x0c19139cu};
    Fq r = fq::one_m();
    for (int i = 255; i >= 0; i--) {
        r = msqr(r);
        if ((E[i >> 5] >> (i & 31)) & 1) r = mmul(r, a);
    }
    return r;
}
__device__ __forceinline__ bool fq_sqrt(const Fq& a, Fq& r) {
    r = mmul(pow_pm3d4(a), a);
    return feq(msqr(r), a);
}

// RFC 9380 section 4.1, m = 2 (on the canonical representatives)
__device__ __forceinline__ int sgn0(const Fq2& a) {
    const Fq s0 = fq::from_mont(a.c0), s1 = fq::from_mont(a.c1);
    const int z0 = fq::is_zero(s0) ? 1 : 0;
    return (int)(s0.l[0] & 1u) | (z0 & (int)(s1.l[0] & 1u));
}

// THE root with sgn0 = 0; false for a non-square.  By the norm: x0^2 = (a0 +- sqrt(a0^2 + a1^2)) / 2, x1 = a1 / (2 x0); the inverse of
// x0 is the power the root came from: c^((p-3)/4) * c^((p+1)/4) = c^((p-1)/2) = 1 for a non-zero square c.  `half` = 1/2.
__device__ __noinline__ bool sqrt_even(const Fq2& a, const Fq& half, Fq2& out) {
    Fq2 r;
    if (fq::is_zero(a.c1)) {
        Fq s;
        if (fq_sqrt(a.c0, s)) r = Fq2{s, fq::zero()};
        else if (fq_sqrt(fq::neg(a.c0), s)) r = Fq2{fq::zero(), s};
        else return false;
    } else {
        Fq n;
        if (!fq_sqrt(fq::add(msqr(a.c0), msqr(a.c1)), n)) return false;
        Fq t = mmul(fq::add(a.c0, n), half);
        Fq pw = pow_pm3d4(t);
        Fq x0 = mmul(pw, t);
        if (!feq(msqr(x0), t)) {
            t = mmul(fq::sub(a.c0, n), half);
            pw = pow_pm3d4(t);
            x0 = mmul(pw, t);
            if (!feq(msqr(x0), t)) return false;
        }
        r = Fq2{x0, mmul(mmul(a.c1, half), pw)};      // x0 != 0 here (a1 != 0), so pw = 1 / x0
    }
    const Fq2 rr = m2(r, r);
    if (!feq(rr.c0, a.c0) || !feq(rr.c1, a.c1)) return false;
    out = sgn0(r) ? neg2(r) : r;
    return true;
}

__device__ __forceinline__ Fq load_fq(const uint32_t* w) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = w[i];
    return fq::to_mont(r);
}
__device__ __forceinline__ Fq2 load2(const uint32_t* w) { return Fq2{load_fq(w), load_fq(w + 8)}; }

// 16-bit limbs of an Fp2 value into unchecked cells (one cell per limb) / checked cells (cpl cells per limb)
__device__ __forceinline__ void store_u16(uint64_t* tr, size_t n, int col, size_t row, const Fq2& v) {
    const Fq s[2] = {fq::from_mont(v.c0), fq::from_mont(v.c1)};
    for (int c = 0; c < 2; c++)
        for (int l = 0; l < 16; l++) tr[(size_t)(col + 16 * c + l) * n + row] = (s[c].l[l >> 1] >> (16 * (l & 1))) & 0xffffu;
}
__device__ __forceinline__ void store_chk(uint64_t* tr, size_t n, int col, size_t row, const Fq2& v, int cpl) {
    const Fq s[2] = {fq::from_mont(v.c0), fq::from_mont(v.c1)};
    for (int c = 0; c < 2; c++)
        for (int l = 0; l < 16; l++) {
            const uint32_t limb = (s[c].l[l >> 1] >> (16 * (l & 1))) & 0xffffu;
            if (cpl == 1) {
                tr[(size_t)(col + 16 * c + l) * n + row] = limb;
            } else {
                tr[(size_t)(col + 32 * c + 2 * l) * n + row] = limb & 0xffu;
                tr[(size_t)(col + 32 * c + 2 * l + 1) * n + row] = limb >> 8;
            }
        }
}

enum { MG_T1, MG_TV1, MG_W, MG_TV3, MG_A4, MG_B4, MG_X2, MG_X1, MG_S1, MG_GX1, MG_S2, MG_GX2, MG_D, MG_E, MG_F, MG_X3, MG_S3,
       MG_GX3, MG_N1, MG_N2, MG_Y, MG_NWIT };
static_assert(MG_NWIT == AIR_MAPG2_NWIT && AIR_MAPG2_ROWS == 8, "witness order / rows per message of tools/air_gen.py");
// index into AIR_MAPG2_LAYOUT_*: columns, then the registers that hold x1 x2 x3 g(x1) g(x2) g(x3) for the selections
enum { L_U, L_ONE, L_C1, L_C2, L_C3, L_C4, L_BB, L_E1, L_E2, L_M1, L_M2, L_XS, L_GXS, L_REG, L_RES, L_RX1, L_RX2, L_RX3, L_RG1, L_RG2,
       L_RG3, L_Z, L_ZV, L_TINV, L_N };

struct MapConsts {
    uint32_t c1[16], c2[16], c3[16], c4[16], b[16];
    // the schedule of tools/air_gen.py (MAP_SLOTS / MAP_REG_LOADS) and the column layout of the AIR variant in use
    int32_t slot_wit[8][3], reg_wit[8][6], lay[L_N];
};

// EIGHT LANES PER MESSAGE (a wave = eight messages).  (A) every lane of a group runs the chain u -> g(x1), g(x2), g(x3) (~450 field
// products, the same values in all eight: no exchange); (B) lanes 0 .. 4 take ONE of the five candidate roots each -- of g(x1),
// g(x2), g(x3), (9+u) g(x1), (9+u) g(x2): which of them exist decides the branch -- ~1150 products; (C) the roots meet in LDS, every lane
// derives the branch bits and the witness vector, and lane t writes trace row t of the block (mode 0, after the record's claimed
// point was compared) or lane 0 writes (x, y) into the record (mode 1).  One lane per message took 8.9 ms for 128 messages
// (two waves of ~10 k dependent products); this form 1.6 ms.
__global__ void __launch_bounds__(64) mapg2_rows_kernel(uint32_t* __restrict__ ios, uint32_t num_io, uint32_t ppi, MapConsts k,
                                                       uint64_t* __restrict__ tr, size_t n, int cpl, int checked_base, int mode,
                                                       int* __restrict__ err) {
    __shared__ Fq2 s_root[8][5];
    __shared__ int s_ok[8][5];
    __shared__ Fq2 s_v[8][MG_NWIT];
    const uint32_t grp = threadIdx.x >> 3, ln = threadIdx.x & 7;
    const uint32_t io0 = blockIdx.x * 8 + grp;
    const bool live = io0 < num_io;
    const uint32_t io = live ? io0 : num_io - 1;       // idle groups redo the last message and store nothing (barriers below)
    uint32_t* rec = ios + (size_t)io * ppi;
    const Fq2 one{fq::one_m(), fq::zero()}, zero{fq::zero(), fq::zero()};
    const Fq2 c1 = load2(k.c1), c2 = load2(k.c2), c3 = load2(k.c3), c4 = load2(k.c4), bb = load2(k.b);
    const Fq2 xi{fq::small_m(9), fq::one_m()};
    const Fq half = fq::neg(c2.c0);                    // c2 = -Z/2 = -1/2
    const Fq2 u = load2(rec);
    Fq2 v[MG_NWIT];
    v[MG_T1] = m2(u, u);
    v[MG_TV1] = m2(c1, v[MG_T1]);
    const Fq2 tv2 = fq::add(one, v[MG_TV1]), tv1 = fq::sub(one, v[MG_TV1]);
    v[MG_W] = m2(tv1, tv2);
    const bool w_zero = fq::is_zero(v[MG_W]);          // u^2 g(Z) = +-1: inv0(0) = 0 (0^(p-2) = 0 below), flagged by the z cell
    {
        const Fq nrm = fq::inv(fq::add(msqr(v[MG_W].c0), msqr(v[MG_W].c1)));
        v[MG_TV3] = Fq2{mmul(v[MG_W].c0, nrm), fq::neg(mmul(v[MG_W].c1, nrm))};
    }
    v[MG_A4] = m2(u, tv1);
    v[MG_B4] = m2(v[MG_A4], v[MG_TV3]);
    const Fq2 tv4 = m2(v[MG_B4], c3);
    v[MG_X1] = fq::sub(c2, tv4);
    v[MG_X2] = fq::add(c2, tv4);
    v[MG_S1] = m2(v[MG_X1], v[MG_X1]);
    v[MG_GX1] = fq::add(m2(v[MG_S1], v[MG_X1]), bb);
    v[MG_S2] = m2(v[MG_X2], v[MG_X2]);
    v[MG_GX2] = fq::add(m2(v[MG_S2], v[MG_X2]), bb);
    v[MG_D] = m2(tv2, tv2);
    v[MG_E] = m2(v[MG_D], v[MG_TV3]);
    v[MG_F] = m2(v[MG_E], v[MG_E]);
    v[MG_X3] = fq::add(m2(v[MG_F], c4), one);
    v[MG_S3] = m2(v[MG_X3], v[MG_X3]);
    v[MG_GX3] = fq::add(m2(v[MG_S3], v[MG_X3]), bb);
    // (B) candidate roots: 0: g(x1)  1: g(x2)  2: g(x3)  3: (9+u) g(x1)  4: (9+u) g(x2)
    if (ln < 5) {
        Fq2 arg = ln == 0 ? v[MG_GX1] : ln == 1 ? v[MG_GX2] : ln == 2 ? v[MG_GX3] : ln == 3 ? v[MG_GX1] : v[MG_GX2];
        if (ln >= 3) arg = m2(xi, arg);
        Fq2 r = zero;
        const bool ok = sqrt_even(arg, half, r);
        s_ok[grp][ln] = ok ? 1 : 0;
        s_root[grp][ln] = r;
    }
    __syncthreads();
    // (C)
    const bool e1 = s_ok[grp][0] != 0;
    const bool e2 = !e1 && s_ok[grp][1] != 0;
    bool ok = true;
    v[MG_N1] = zero;
    v[MG_N2] = zero;
    if (!e1) {                       // g(x1) is not a square: (9+u) g(x1) is
        ok = ok && s_ok[grp][3] != 0;
        v[MG_N1] = s_root[grp][3];
    }
    if (!e1 && !e2) {
        ok = ok && s_ok[grp][4] != 0;
        v[MG_N2] = s_root[grp][4];
    }
    if (!e1 && !e2) ok = ok && s_ok[grp][2] != 0;
    Fq2 y = e1 ? s_root[grp][0] : e2 ? s_root[grp][1] : s_root[grp][2];
    const Fq2 xs = e1 ? v[MG_X1] : e2 ? v[MG_X2] : v[MG_X3];
    if (sgn0(u) != sgn0(y)) y = neg2(y);
    v[MG_Y] = y;
    const Fq o[4] = {fq::from_mont(xs.c0), fq::from_mont(xs.c1), fq::from_mont(y.c0), fq::from_mont(y.c1)};
    if (mode == 1) {
        if (live && ln == 0) {
            if (!ok) atomicExch(err, SIPP_E_WITNESS);
            else
                for (int q = 0; q < 4; q++)
                    for (int l = 0; l < 8; l++) rec[16 + 8 * q + l] = o[q].l[l];
        }
        return;                      // no barrier below in this mode
    }
    bool bad = !ok;
    for (int q = 0; q < 4; q++)
        for (int l = 0; l < 8; l++) bad |= rec[16 + 8 * q + l] != o[q].l[l];   // the record claims another point
    if (bad) {
        if (live && ln == 0) atomicExch(err, SIPP_E_WITNESS);
    }
    // the witness vector through LDS: the schedule indexes it dynamically (registers would spill)
    if (ln == 0)
        for (int i = 0; i < MG_NWIT; i++) s_v[grp][i] = v[i];
    __syncthreads();
    if (!live || bad) return;
    {
        const int t = (int)ln;
        const size_t row = (size_t)io * 8 + t;
        store_u16(tr, n, k.lay[L_U], row, u);
        store_u16(tr, n, k.lay[L_ONE], row, one);
        store_u16(tr, n, k.lay[L_C1], row, c1);
        store_u16(tr, n, k.lay[L_C2], row, c2);
        store_u16(tr, n, k.lay[L_C3], row, c3);
        store_u16(tr, n, k.lay[L_C4], row, c4);
        store_u16(tr, n, k.lay[L_BB], row, bb);
        tr[(size_t)k.lay[L_E1] * n + row] = e1 ? 1 : 0;
        tr[(size_t)k.lay[L_E2] * n + row] = e2 ? 1 : 0;
        const uint64_t zf = (t == k.lay[L_TINV] && w_zero) ? 1 : 0;
        tr[(size_t)k.lay[L_Z] * n + row] = zf;
        for (int j = 0; j < 32; j++) tr[(size_t)(k.lay[L_ZV] + j) * n + row] = j == 0 ? zf : 0;
        Fq2 reg[6];
        for (int r = 0; r < 6; r++) {
            const int wi = k.reg_wit[t][r];
            reg[r] = wi < 0 ? zero : s_v[grp][wi];
            store_u16(tr, n, k.lay[L_REG] + 32 * r, row, reg[r]);
        }
        for (int sl = 0; sl < 3; sl++) {
            const int wi = k.slot_wit[t][sl];
            store_chk(tr, n, checked_base + 32 * cpl * sl, row, wi < 0 ? zero : s_v[grp][wi], cpl);
        }
        // the selection cells as the (ungated) constraints define them from the registers of THIS row
        const Fq2 rg1 = reg[k.lay[L_RG1]], rg2 = reg[k.lay[L_RG2]], rg3 = reg[k.lay[L_RG3]];
        store_u16(tr, n, k.lay[L_M1], row, e1 ? zero : rg1);
        store_u16(tr, n, k.lay[L_M2], row, (e1 || e2) ? zero : rg2);
        store_u16(tr, n, k.lay[L_XS], row, e1 ? reg[k.lay[L_RX1]] : e2 ? reg[k.lay[L_RX2]] : reg[k.lay[L_RX3]]);
        store_u16(tr, n, k.lay[L_GXS], row, e1 ? rg1 : e2 ? rg2 : rg3);
    }
}

MapConsts map_consts(const air_spec_t* a) {
    MapConsts k;
    memcpy(k.c1, SIPP_MAPG2_C1, 64);
    memcpy(k.c2, SIPP_MAPG2_C2, 64);
    memcpy(k.c3, SIPP_MAPG2_C3, 64);
    memcpy(k.c4, SIPP_MAPG2_C4, 64);
    memcpy(k.b, SIPP_MAPG2_B, 64);
    memcpy(k.slot_wit, AIR_MAPG2_SLOT_WIT, sizeof k.slot_wit);
    memcpy(k.reg_wit, AIR_MAPG2_REG_WIT, sizeof k.reg_wit);
    memcpy(k.lay, (a && a->cells_per_limb == 1) ? AIR_MAPG2_LAYOUT_U16 : AIR_MAPG2_LAYOUT_U8, sizeof k.lay);
    return k;
}

}  // namespace

// primary witness of the MapToG2 AIR: d_ios [num_io][48] (padded: 8 num_io == n rows), every primary cell of rows 8 io .. 8 io + 7
int sipp_mapg2_fill(sipp_ctx* ctx, const air_spec_t* a, const uint32_t* d_ios, uint32_t num_io, uint32_t log_n, uint64_t* d_trace,
                    int* d_err) {
    const size_t n = (size_t)1 << log_n;
    const MapConsts k = map_consts(a);
    if (a->kind != 3 || a->log_rows != 3 || a->checked_base != k.lay[L_RES] || a->pi_per_io != 48 || (size_t)num_io * 8 != n)
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "MapToG2: AIR table / shape mismatch");
    ProfScope ps(ctx, "trace_mapg2");
    hipLaunchKernelGGL(mapg2_rows_kernel, dim3((num_io + 7) / 8), dim3(64), 0, ctx->stream, const_cast<uint32_t*>(d_ios), num_io, 48u,
                       k, d_trace, n, a->cells_per_limb, a->checked_base, 0, d_err);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

// (x, y) of every record computed from its u and written into the record (device buffer [num_io][48])
int sipp_mapg2_outputs(sipp_ctx* ctx, uint32_t* d_ios, uint32_t num_io, int* d_err) {
    ProfScope ps(ctx, "mapg2_outputs");
    hipLaunchKernelGGL(mapg2_rows_kernel, dim3((num_io + 7) / 8), dim3(64), 0, ctx->stream, d_ios, num_io, 48u, map_consts(nullptr),
                       (uint64_t*)nullptr, (size_t)0, 1, 0, 1, d_err);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

extern "C" {

// What the reference computes natively per message (src/bin/bls_aggregation.rs:100-104): the map, then the cofactor -- as
// records the provers take (include/sipp_hip.h).  Three passes of the outputs-only trace kernels: the map, G + [h] Q, - G.
int sipp_map_to_g2(sipp_ctx* ctx, const uint32_t* msgs, size_t n, uint32_t* map_ios, uint32_t* g2_ios, uint32_t* cleared) {
    if (!ctx || !msgs || !map_ios || n == 0 || n > ((size_t)1 << 17)) return SIPP_E_BADARG;   // 2^17 records: the provers' limit
    for (size_t i = 0; i < n; i++) {
        uint32_t* r = map_ios + i * SIPP_MAP_G2_IO_WORDS;
        memcpy(r, msgs + i * 16, 64);
        memset(r + 16, 0, 128);
    }
    SIPP_TRY(sipp_exp_outputs(ctx, SIPP_MAP_G2, map_ios, n));
    if (!g2_ios && !cleared) return SIPP_OK;
    std::vector<uint32_t> own;
    uint32_t* g = g2_ios;
    if (!g) {
        try {
            own.resize(2 * n * SIPP_G2_IO_WORDS);      // n <= 2^17 (the call above refused anything larger)
        } catch (const std::exception&) {
            return sipp_fail(ctx, SIPP_E_NOMEM, "map_to_g2: host scratch for the cofactor obligations");
        }
        g = own.data();
    }
    // (x, offset, exp_val, output): 32 + 32 + 8 + 32 words
    for (size_t i = 0; i < n; i++) {
        uint32_t* r = g + i * SIPP_G2_IO_WORDS;
        memcpy(r, map_ios + i * SIPP_MAP_G2_IO_WORDS + 16, 128);
        memcpy(r + 32, SIPP_G2_GEN, 128);
        memcpy(r + 64, SIPP_G2_COFACTOR, 32);
        memset(r + 72, 0, 128);
    }
    SIPP_TRY(sipp_exp_outputs(ctx, SIPP_G2_EXP, g, n));
    for (size_t i = 0; i < n; i++) {
        uint32_t* r = g + (n + i) * SIPP_G2_IO_WORDS;
        memcpy(r, SIPP_G2_GEN_NEG, 128);
        memcpy(r + 32, g + i * SIPP_G2_IO_WORDS + 72, 128);
        memset(r + 64, 0, 32);
        r[64] = 1;
        memset(r + 72, 0, 128);
    }
    SIPP_TRY(sipp_exp_outputs(ctx, SIPP_G2_EXP, g + n * SIPP_G2_IO_WORDS, n));
    if (cleared)
        for (size_t i = 0; i < n; i++) memcpy(cleared + i * 32, g + (n + i) * SIPP_G2_IO_WORDS + 72, 128);
    return SIPP_OK;
}

}  // extern "C"
