// sipp_amd/csrc/ntt.hip -- batched radix-2 Goldilocks NTT for gfx950, LDS-staged.
//
// Replaces plonky2's per-column fft/ifft/coset_fft (field/src/fft.rs @ 541e127, reached from the
// reference only through starky::prover::prove behind src/verifier_circuit.rs:133-135).
//
// Design (MI355X-first, see DESIGN.md "NTT"):
//  * data stays column-major [col][row]; one workgroup owns one LDS tile of a column.
//  * a size-n transform is split into P passes over bit groups k_1 (highest index bits) .. k_P
//    (lowest, contiguous).  Each pass does ALL of its k_i butterfly stages inside LDS, so a
//    pass moves every element through HBM exactly once (read 8 B, write 8 B).
//      DIF (natural -> bit-reversed): passes run 1..P,  twiddle diagonal AFTER the small DIF
//      DIT (bit-reversed -> natural): passes run P..1,  twiddle diagonal BEFORE the small DIT
//    (four-step factorisation; the diagonal w_m^(i0 * o1) is produced from a two-level table by
//    one lookup per 16 elements plus a running product, never from a size-n table).
//  * strided passes use tiles of R rows x T >= 16 consecutive elements (>= 128 B contiguous per
//    row => coalesced); the contiguous pass takes G whole size-R blocks per tile.
//  * LDS index = e + (e >> 4): one u64 of padding per 16 keeps both the butterfly phases
//    (lanes along t) and the diagonal phases (one lane per 16-element segment) conflict-free
//    for ds_read_b64 / ds_write_b64 (64 banks x 4 B).
//  * optional power diagonal base^(natural index) * c on the natural-order side gives the coset
//    shift of the LDE (DIF input) and the shift^-j / n un-scaling of a coset iNTT (DIT output).
//  * zero padding of the LDE is never read from HBM: rows >= n_in are materialised as zeros in LDS.
#include "ctx.hpp"

namespace {

constexpr int SEG = 16;      // diagonal segment length (elements)
constexpr int LOG_SEG = 4;

struct PassArgs {
    const uint64_t* in;
    uint64_t* out;
    size_t in_stride, out_stride;
    uint32_t log_n;   // positions per column
    uint32_t log_m;   // current block size R * B
    uint32_t k;       // log R
    uint32_t lt;      // log T   (B > 1 layout: tile = [R][T])
    uint32_t lg;      // log G   (B == 1 layout: tile = [G][R])
    uint32_t dit;     // 0 = DIF stages, 1 = DIT stages
    uint64_t n_in;    // rows >= n_in read as zero
    const uint64_t* wr;
    // twiddle diagonal: 0 none, 1 before butterflies, 2 after
    uint32_t tw_mode, tw_L;
    const uint64_t* tw_lo;
    const uint64_t* tw_hi;
    // power diagonal: 0 none, 1 before butterflies, 2 after
    uint32_t pw_mode, pw_L;
    const uint64_t* pw_lo;
    const uint64_t* pw_hi;
    uint64_t pw_step;
    uint64_t cscale;  // 0 = none; multiplied in after the butterflies
};

__device__ __forceinline__ uint32_t lds_idx(uint32_t e) { return e + (e >> LOG_SEG); }

// for e = threadIdx.x; e < E; e += blockDim.x:  use(e, load(e)) -- with U loads in flight per lane.  The trip counts of the tile
// loops are run-time values, so the compiler keeps them rolled and waits for every load before it issues the next
// (global_load; s_waitcnt vmcnt(0); ds_write in the ISA): one 8-byte load in flight per lane caps a sweep at 2.5-3 TB/s
// (256 CUs x 16 waves x 512 B over ~0.8 us).  Batches of U keep U loads in flight.
struct Two {
    uint64_t a, b;
};
template <int U, class T, class Load, class Use>
__device__ __forceinline__ void tile_loop(uint32_t E, Load load, Use use) {
    const uint32_t step = blockDim.x;
    uint32_t e = threadIdx.x;
    for (; e + (U - 1) * step < E; e += U * step) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = load(e + u * step);
#pragma unroll
        for (int u = 0; u < U; u++) use(e + u * step, v[u]);
    }
    for (; e < E; e += step) use(e, load(e));
}
#ifndef SIPP_NTT_MLP
#define SIPP_NTT_MLP 8
#endif

// LDS placement of tile element e: one u64 of padding per 16 (lds_idx)
struct IdxPlain {
    __device__ __forceinline__ uint32_t operator()(uint32_t e) const { return lds_idx(e); }
};

// all butterfly stages of one bit group on an LDS tile of E elements: k stages over rows (element stride T = 2^lt)
template <class Idx>
__device__ __forceinline__ void tile_stages(uint64_t* tile, const uint64_t* wr_s, uint32_t E, uint32_t k, uint32_t lt, bool dit,
                                            Idx lidx) {
    const uint32_t R = 1u << k, T = 1u << lt;
    uint32_t s = 0;
    // radix-2^4 in registers: a lane loads the 16 elements of a hexadecuple, applies FOUR stages and writes them back -- a quarter
    // of the LDS round trips, index arithmetic and barriers of radix 2^2, 15 twiddle loads instead of 24 (same 32 products)
    const uint32_t sixteenth = E >> 4;
    for (; s + 3 < k; s += 4) {
        const uint32_t log_q = dit ? s : (k - 4 - s);
        const uint32_t q = 1u << log_q;
        for (uint32_t idx = threadIdx.x; idx < sixteenth; idx += blockDim.x) {
            const uint32_t t = idx & (T - 1);
            const uint32_t rest = idx >> lt;
            const uint32_t b = rest & ((R >> 4) - 1);
            const uint32_t g = rest >> (k - 4);
            const uint32_t j = b & (q - 1);
            const uint32_t r0 = ((b >> log_q) << (log_q + 4)) | j;
            const uint32_t e0 = (((g << k) | r0) << lt) | t;
            const uint32_t st = q << lt;
            uint64_t x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = tile[lidx(e0 + i * st)];
            if (dit) {
                // distances h, 2h, 4h, 8h (h = q): stage s + d pairs offsets (i, i + 2^d) with twiddle w^((j + (i mod 2^d) h) << (k - 1 - s - d))
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int half = 1 << d;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        if (i & half) continue;
                        const uint64_t w = wr_s[(j + (uint32_t)(i & (half - 1)) * q) << (k - 1 - s - d)];
                        const uint64_t v = gl::mul(x[i + half], w), u = x[i];
                        x[i] = gl::add(u, v);
                        x[i + half] = gl::sub(u, v);
                    }
                }
            } else {
                // distances 8q, 4q, 2q, q: stage s + d pairs offsets (i, i + 2^(3 - d)) with twiddle w^((j + (i mod 2^(3 - d)) q) << (s + d))
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const int half = 8 >> d;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        if (i & half) continue;
                        const uint64_t w = wr_s[(j + (uint32_t)(i & (half - 1)) * q) << (s + d)];
                        const uint64_t u = x[i], v = x[i + half];
                        x[i] = gl::add(u, v);
                        x[i + half] = gl::mul(gl::sub(u, v), w);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 16; i++) tile[lidx(e0 + i * st)] = x[i];
        }
        __syncthreads();
    }
    const uint32_t quarter = E >> 2;
    for (; s + 1 < k; s += 2) {
        const uint32_t log_q = dit ? s : (k - 2 - s);
        for (uint32_t idx = threadIdx.x; idx < quarter; idx += blockDim.x) {
            const uint32_t t = idx & (T - 1);
            const uint32_t rest = idx >> lt;
            const uint32_t b = rest & ((R >> 2) - 1);
            const uint32_t g = rest >> (k - 2);
            const uint32_t j = b & ((1u << log_q) - 1);
            const uint32_t r0 = ((b >> log_q) << (log_q + 2)) | j;
            const uint32_t e0 = (((g << k) | r0) << lt) | t;
            const uint32_t st = (1u << log_q) << lt;
            const uint32_t l0 = lidx(e0), l1 = lidx(e0 + st), l2 = lidx(e0 + 2 * st), l3 = lidx(e0 + 3 * st);
            uint64_t x0 = tile[l0], x1 = tile[l1], x2 = tile[l2], x3 = tile[l3];
            if (dit) {
                const uint64_t w1 = wr_s[j << (k - 1 - s)];
                const uint64_t w20 = wr_s[j << (k - 2 - s)], w21 = wr_s[(j + (1u << s)) << (k - 2 - s)];
                const uint64_t v1 = gl::mul(x1, w1), v3 = gl::mul(x3, w1);
                const uint64_t a0 = gl::add(x0, v1), a1 = gl::sub(x0, v1), a2 = gl::add(x2, v3), a3 = gl::sub(x2, v3);
                const uint64_t u2 = gl::mul(a2, w20), u3 = gl::mul(a3, w21);
                tile[l0] = gl::add(a0, u2);
                tile[l2] = gl::sub(a0, u2);
                tile[l1] = gl::add(a1, u3);
                tile[l3] = gl::sub(a1, u3);
            } else {
                const uint64_t w10 = wr_s[j << s], w11 = wr_s[(j + (1u << log_q)) << s];
                const uint64_t w2 = wr_s[j << (s + 1)];
                const uint64_t a0 = gl::add(x0, x2), a2 = gl::mul(gl::sub(x0, x2), w10);
                const uint64_t a1 = gl::add(x1, x3), a3 = gl::mul(gl::sub(x1, x3), w11);
                tile[l0] = gl::add(a0, a1);
                tile[l1] = gl::mul(gl::sub(a0, a1), w2);
                tile[l2] = gl::add(a2, a3);
                tile[l3] = gl::mul(gl::sub(a2, a3), w2);
            }
        }
        __syncthreads();
    }
    const uint32_t half = E >> 1;
    for (; s < k; s++) {
        const uint32_t log_hd = dit ? s : (k - 1 - s);
        const uint32_t tw_shift = k - 1 - log_hd;
        for (uint32_t idx = threadIdx.x; idx < half; idx += blockDim.x) {
            const uint32_t t = idx & (T - 1);
            const uint32_t rest = idx >> lt;
            const uint32_t b = rest & ((R >> 1) - 1);
            const uint32_t g = rest >> (k - 1);
            const uint32_t j = b & ((1u << log_hd) - 1);
            const uint32_t r0 = ((b >> log_hd) << (log_hd + 1)) | j;
            const uint32_t e0 = (((g << k) | r0) << lt) | t;
            const uint32_t e1 = e0 + ((1u << log_hd) << lt);
            const uint64_t w = wr_s[j << tw_shift];
            const uint32_t l0 = lidx(e0), l1 = lidx(e1);
            uint64_t u = tile[l0], v = tile[l1];
            if (dit) {
                v = gl::mul(v, w);
                tile[l0] = gl::add(u, v);
                tile[l1] = gl::sub(u, v);
            } else {
                tile[l0] = gl::add(u, v);
                tile[l1] = gl::mul(gl::sub(u, v), w);
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) ntt_pass_kernel(PassArgs a) {
    extern __shared__ uint64_t smem[];
    const uint32_t k = a.k, lt = a.lt, lg = a.lg;
    const uint32_t R = 1u << k, T = 1u << lt;
    const uint32_t E = 1u << (k + lt + lg);           // tile elements
    const uint32_t log_b = a.log_m - k;                // log B
    const bool strided = log_b > 0;                    // B > 1 layout
    uint64_t* tile = smem;
    uint64_t* wr_s = smem + (E + (E >> LOG_SEG));

    const uint32_t tiles_per_col = 1u << (a.log_n - (k + lt + lg));
    const uint32_t col = blockIdx.x / tiles_per_col;
    const uint32_t tix = blockIdx.x - col * tiles_per_col;
    const uint64_t* in = a.in + (size_t)col * a.in_stride;
    uint64_t* out = a.out + (size_t)col * a.out_stride;

    // tile -> global position mapping
    //   strided: tix = blk * (B/T) + i0_hi ; pos(r, t) = blk*m + r*B + i0_hi*T + t
    //   contiguous: pos(e) = tix * E + e
    uint64_t base_pos;
    uint32_t i0_base = 0;  // i0_hi * T (strided only)
    if (strided) {
        const uint32_t tiles_per_blk = 1u << (log_b - lt);
        const uint32_t blk = tix >> (log_b - lt);
        i0_base = (tix & (tiles_per_blk - 1)) << lt;
        base_pos = ((uint64_t)blk << a.log_m) + i0_base;
    } else {
        base_pos = (uint64_t)tix << (k + lg);
    }
    auto pos_of = [&](uint32_t e) -> uint64_t {
        if (strided) return base_pos + ((uint64_t)(e >> lt) << log_b) + (e & (T - 1));
        return base_pos + e;
    };

    // ---- load ----
    tile_loop<SIPP_NTT_MLP, uint64_t>(
        E,
        [&](uint32_t e) -> uint64_t {
            const uint64_t p = pos_of(e);
            return p < a.n_in ? in[p] : 0;
        },
        [&](uint32_t e, uint64_t v) { tile[lds_idx(e)] = v; });
    for (uint32_t i = threadIdx.x; i < (R >> 1); i += blockDim.x) wr_s[i] = a.wr[i];
    __syncthreads();

    const uint32_t nseg = E >> LOG_SEG;
    // ---- diagonal phases: one lane per 16-element segment, running product ----
    auto diag = [&](bool twiddle, bool power, uint64_t cs) {
        for (uint32_t s = threadIdx.x; s < nseg; s += blockDim.x) {
            uint32_t e0 = s << LOG_SEG;
            uint64_t f = 1, step = 1;
            bool active = true;
            if (power) {
                uint64_t p0 = pos_of(e0);
                active = p0 < a.n_in || a.pw_mode == 2;
                f = gl::mul(a.pw_hi[p0 >> a.pw_L], a.pw_lo[p0 & ((1ull << a.pw_L) - 1)]);
                step = a.pw_step;
            }
            if (twiddle) {
                uint32_t r = e0 >> lt;
                uint32_t o1 = gl::bitrev(r, k);
                uint64_t ex = (uint64_t)(i0_base + (e0 & (T - 1))) * o1;
                uint64_t t0 = gl::mul(a.tw_hi[ex >> a.tw_L], a.tw_lo[ex & ((1ull << a.tw_L) - 1)]);
                uint64_t ts = a.tw_lo[o1];  // o1 < R <= 2^tw_L
                f = power ? gl::mul(f, t0) : t0;
                step = power ? gl::mul(step, ts) : ts;
            }
            if (cs) f = gl::mul(f, cs);
            if (!active) continue;
            uint32_t li = lds_idx(e0);
#pragma unroll
            for (int t = 0; t < SEG; t++) {
                tile[li + t] = gl::mul(tile[li + t], f);
                f = gl::mul(f, step);
            }
        }
        __syncthreads();
    };

    bool cs_done = a.cscale == 0;
    if (a.tw_mode == 1 || a.pw_mode == 1) diag(a.tw_mode == 1, a.pw_mode == 1, 0);

    // ---- butterflies: k stages over the r dimension (element stride T), four (two, one) stages per LDS round trip ----
    tile_stages(tile, wr_s, E, k, lt, a.dit != 0, IdxPlain{});

    if (a.tw_mode == 2 || a.pw_mode == 2) {
        diag(a.tw_mode == 2, a.pw_mode == 2, a.cscale);
        cs_done = true;
    }

    // ---- store ----
    tile_loop<SIPP_NTT_MLP, uint64_t>(
        E, [&](uint32_t e) -> uint64_t { return tile[lds_idx(e)]; },
        [&](uint32_t e, uint64_t v) {
            if (!cs_done) v = gl::mul(v, a.cscale);
            out[pos_of(e)] = v;
        });
}


// =====================================================================================================================
// Fused commitment transforms (PolynomialBatch::from_values without intermediate sweeps over HBM)
//
//   lde_column_kernel   2^10 <= N <= 2^14: the WHOLE column lives in LDS.  values (natural) are scattered into LDS in
//                       bit-reversed position, the inverse DIT yields natural coefficients (x 1/N folded into the
//                       twiddle table), which are stored once; then for each of the 2^rate_bits cosets the coefficients
//                       are scaled by the coset's power table, a forward DIF runs in LDS and the half is stored in leaf
//                       order.  HBM per column: 8 N read + 8 N (coefficients) + 8 N 2^rate_bits (LDE) written
//                       (+ 8 N re-read of the just-written coefficients per extra coset: cache hits) instead of
//                       104 N for bitrev copy + 2 iNTT passes + 2 LDE passes.
//   per-element tables  twiddle / coset-power factors come from tables laid out like the data (one coalesced 8-B load
//                       and ONE product per element) instead of a running product per 16-element segment (two products).
// Tile layout = the pass kernel's: position p at LDS index p + p / 16; k1 high bits (rows r = p >> k2, strided stages),
// k2 low bits (contiguous blocks); conventions identical to ntt_pass_kernel (tested bit-exact against it and the oracle).
// =====================================================================================================================

struct ColArgs {
    const uint64_t* in;       // [ncols][in_stride]: values (natural rows) or, with from_coeffs, coefficients
    uint64_t* coeffs;         // [ncols][n] natural order (written unless from_coeffs)
    uint64_t* lde;            // [ncols][lde_stride] leaf order
    size_t in_stride, lde_stride;
    uint32_t log_n, rate_bits, k1, k2, from_coeffs;
    const uint64_t* wr1_inv;  // w_R1^-x, x < R1 / 2
    const uint64_t* wr2_inv;
    const uint64_t* wr1_fwd;
    const uint64_t* wr2_fwd;
    const uint64_t* tw_inv;   // [n]: w_n^-(t bitrev_k1(r)) / n   at position p = r 2^k2 + t
    const uint64_t* tw_fwd;   // [n]: w_n^(t bitrev_k1(r))
    const uint64_t* pw;       // [2^rate_bits][n]: (7 w_m^h)^p, half h = natural LDE index mod 2^rate_bits
};

__global__ void __launch_bounds__(1024) lde_column_kernel(ColArgs a) {
    extern __shared__ uint64_t smem[];
    const uint32_t n = 1u << a.log_n, k1 = a.k1, k2 = a.k2;
    const uint32_t R1 = 1u << k1, R2 = 1u << k2;
    uint64_t* tile = smem;
    uint64_t* w1i = smem + (n + (n >> LOG_SEG));
    uint64_t* w2i = w1i + (R1 >> 1);
    uint64_t* w1f = w2i + (R2 >> 1);
    uint64_t* w2f = w1f + (R1 >> 1);
    const uint32_t col = blockIdx.x;
    const uint64_t* in = a.in + (size_t)col * a.in_stride;
    for (uint32_t i = threadIdx.x; i < (R1 >> 1); i += blockDim.x) {
        w1i[i] = a.wr1_inv[i];
        w1f[i] = a.wr1_fwd[i];
    }
    for (uint32_t i = threadIdx.x; i < (R2 >> 1); i += blockDim.x) {
        w2i[i] = a.wr2_inv[i];
        w2f[i] = a.wr2_fwd[i];
    }
    const uint64_t* cf = in;   // where the coefficients of this column can be (re)read from
    if (!a.from_coeffs) {
        // values -> coefficients: scatter into bit-reversed position, DIT low bits, twiddle (x 1/n), DIT high bits
        tile_loop<SIPP_NTT_MLP, uint64_t>(
            n, [&](uint32_t i) -> uint64_t { return in[i]; },
            [&](uint32_t i, uint64_t v) { tile[lds_idx(gl::bitrev(i, a.log_n))] = v; });
        __syncthreads();
        tile_stages(tile, w2i, n, k2, 0, true, IdxPlain{});
        tile_loop<SIPP_NTT_MLP, uint64_t>(
            n, [&](uint32_t p) -> uint64_t { return a.tw_inv[p]; },
            [&](uint32_t p, uint64_t w) { tile[lds_idx(p)] = gl::mul(tile[lds_idx(p)], w); });
        __syncthreads();
        tile_stages(tile, w1i, n, k1, k2, true, IdxPlain{});
        uint64_t* co = a.coeffs + (size_t)col * n;
        tile_loop<SIPP_NTT_MLP, uint64_t>(
            n, [&](uint32_t p) -> uint64_t { return tile[lds_idx(p)]; }, [&](uint32_t p, uint64_t v) { co[p] = v; });
        cf = co;
    }
    const uint32_t halves = 1u << a.rate_bits;
    for (uint32_t h = 0; h < halves; h++) {
        // coefficients x coset powers -> LDS (the first coset of from_values takes them from LDS, the others re-read them)
        const uint64_t* pw = a.pw + (size_t)h * n;
        if (h == 0 && !a.from_coeffs) {
            tile_loop<SIPP_NTT_MLP, uint64_t>(
                n, [&](uint32_t p) -> uint64_t { return pw[p]; },
                [&](uint32_t p, uint64_t w) { tile[lds_idx(p)] = gl::mul(tile[lds_idx(p)], w); });
        } else {
            __syncthreads();   // the previous half's stores read the tile
            tile_loop<SIPP_NTT_MLP, Two>(
                n, [&](uint32_t p) -> Two { return Two{cf[p], pw[p]}; },
                [&](uint32_t p, Two v) { tile[lds_idx(p)] = gl::mul(v.a, v.b); });
        }
        __syncthreads();
        tile_stages(tile, w1f, n, k1, k2, false, IdxPlain{});
        tile_loop<SIPP_NTT_MLP, uint64_t>(
            n, [&](uint32_t p) -> uint64_t { return a.tw_fwd[p]; },
            [&](uint32_t p, uint64_t w) { tile[lds_idx(p)] = gl::mul(tile[lds_idx(p)], w); });
        __syncthreads();
        tile_stages(tile, w2f, n, k2, 0, false, IdxPlain{});
        // natural LDE index i = i' 2^rate_bits + h sits at leaf position bitrev(h) n + bitrev(i'): the DIF's own order
        uint64_t* out = a.lde + (size_t)col * a.lde_stride + (size_t)gl::bitrev(h, a.rate_bits) * n;
        tile_loop<SIPP_NTT_MLP, uint64_t>(
            n, [&](uint32_t p) -> uint64_t { return tile[lds_idx(p)]; }, [&](uint32_t p, uint64_t v) { out[p] = v; });
    }
}

__global__ void bitrev_cols_kernel(const uint64_t* in, size_t in_stride, uint64_t* out, size_t out_stride,
                                   uint32_t log_n, size_t ncols) {
    size_t n = (size_t)1 << log_n;
    size_t total = n * ncols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t c = i >> log_n;
        uint32_t j = (uint32_t)(i & (n - 1));
        out[c * out_stride + j] = in[c * in_stride + gl::bitrev(j, log_n)];
    }
}

// tiled bit-reversal: index = (a | b | c) with a, c of TB bits each; bitrev = (rev c | rev b | rev a).
// One block moves the 2^TB x 2^TB tile of a fixed (column, b) through LDS so that both the global reads
// (runs over c) and the global writes (runs over rev a) are contiguous 2^TB * 8 B segments.
constexpr int TB = 5;
__global__ void __launch_bounds__(256) bitrev_tiled_kernel(const uint64_t* __restrict__ in, size_t in_stride,
                                                          uint64_t* __restrict__ out, size_t out_stride, uint32_t log_n) {
    __shared__ uint64_t tile[1 << TB][(1 << TB) + 1];
    const uint32_t lb = log_n - 2 * TB;
    const uint32_t b = blockIdx.x, col = blockIdx.y;
    const uint32_t rb = gl::bitrev(b, lb);
    const uint64_t* src = in + (size_t)col * in_stride;
    uint64_t* dst = out + (size_t)col * out_stride;
    // out[j] = in[bitrev(j)]:  j = (c' | b' | a')  reads  i = (rev a' | rev b' | rev c')
    for (uint32_t e = threadIdx.x; e < (1u << (2 * TB)); e += 256) {
        const uint32_t a = e >> TB, c = e & ((1u << TB) - 1);
        tile[a][c] = src[((size_t)a << (lb + TB)) | ((size_t)b << TB) | c];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < (1u << (2 * TB)); e += 256) {
        const uint32_t cp = e >> TB, ap = e & ((1u << TB) - 1);
        dst[((size_t)cp << (lb + TB)) | ((size_t)rb << TB) | ap] = tile[gl::bitrev(ap, TB)][gl::bitrev(cp, TB)];
    }
}

// ---- host side: tables + pass planning -----------------------------------------------------

// Radix-2^4 register rounds in every transform kernel: ~100 VGPRs instead of 38 and NOT faster alone (lde_column 1.33 vs 1.09 ms for
// 67 M elements; the pass kernel 5-10 % faster), but fewer instructions, and the instance is bound by instruction issue: 66.5-67.1 ms
// per n = 128 instance against 67.8-67.9 ms (round 2, 25-step runs, same box, alternating).
// Tile = 2^12 elements for every size, measured (round 2): a 2^13-element tile would turn the THREE passes of log_n = 21, 22 into
// two, but every pass on it is slower than the saved sweep is worth (256 threads: n = 4096 1697 vs 1464 ms per instance; 512 threads:
// 315 + 143 ms of transform time against 285 + 128 ms).
constexpr uint32_t NTT_LTILE = 12;

// ks[0] = highest index bits ... ks.back() = lowest (contiguous) bits
std::vector<uint32_t> plan(uint32_t log_n) {
    const uint32_t LT = NTT_LTILE;
    std::vector<uint32_t> ks;
    if (log_n <= LT) {
        ks.push_back(log_n);
        return ks;
    }
    uint32_t rem = log_n - LT;
    uint32_t kmax = LT - LOG_SEG;  // strided passes keep T >= 16
    uint32_t q = (rem + kmax - 1) / kmax;
    for (uint32_t i = 0; i < q; i++) {
        uint32_t ki = rem / (q - i) + ((rem % (q - i)) ? 1 : 0);
        ks.push_back(ki);
        rem -= ki;
    }
    ks.push_back(LT);
    return ks;
}

uint64_t* wr_table(sipp_ctx* ctx, uint32_t k, bool inverse) {
    uint64_t* t = sipp_table_get(ctx, TAB_WR, k, inverse);
    if (t) return t;
    size_t h = k ? ((size_t)1 << (k - 1)) : 1;
    std::vector<uint64_t> v(h);
    uint64_t w = gl::root_of_unity(k);
    if (inverse) w = gl::inv(w);
    uint64_t x = 1;
    for (size_t i = 0; i < h; i++) {
        v[i] = x;
        x = gl::mul(x, w);
    }
    if (sipp_table_put(ctx, TAB_WR, k, inverse, v, &t) != SIPP_OK) return nullptr;
    return t;
}

// two-level power table of `base`: lo[x] = base^x * c (x < 2^L), hi[y] = base^(y << L) (y < 2^(bits-L))
int pow_tables(sipp_ctx* ctx, int kind_lo, int kind_hi, uint64_t key_a, uint64_t key_b, uint64_t base, uint64_t c,
               uint32_t bits, uint32_t L, uint64_t** lo, uint64_t** hi) {
    *lo = sipp_table_get(ctx, kind_lo, key_a, key_b);
    *hi = sipp_table_get(ctx, kind_hi, key_a, key_b);
    if (*lo && *hi) return SIPP_OK;
    std::vector<uint64_t> vl((size_t)1 << L), vh((size_t)1 << (bits > L ? bits - L : 0));
    uint64_t x = c ? c : 1;
    for (auto& e : vl) {
        e = x;
        x = gl::mul(x, base);
    }
    uint64_t bl = gl::pow(base, (uint64_t)1 << L);
    x = 1;
    for (auto& e : vh) {
        e = x;
        x = gl::mul(x, bl);
    }
    SIPP_TRY(sipp_table_put(ctx, kind_lo, key_a, key_b, vl, lo));
    SIPP_TRY(sipp_table_put(ctx, kind_hi, key_a, key_b, vh, hi));
    return SIPP_OK;
}

int run_pass(sipp_ctx* ctx, const char* name, PassArgs& a, size_t ncols) {
    const uint32_t log_e = a.k + a.lt + a.lg;
    const size_t E = (size_t)1 << log_e;
    size_t shmem = (E + (E >> LOG_SEG) + ((size_t)1 << (a.k ? a.k - 1 : 0))) * sizeof(uint64_t);
    size_t tiles = ((size_t)1 << (a.log_n - log_e)) * ncols;
    if (tiles == 0) return SIPP_OK;
    if (tiles > 0x7fffffffull) return sipp_fail(ctx, SIPP_E_BADARG, "ntt: too many tiles for one launch");
    if (shmem > 64 * 1024) {
        SIPP_CHECK_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_kernel,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    }
    ProfScope ps(ctx, name);
    hipLaunchKernelGGL(ntt_pass_kernel, dim3((unsigned)tiles), dim3(256), shmem, ctx->stream, a);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

// shared driver for DIF and DIT
int ntt_run(sipp_ctx* ctx, bool dit, const uint64_t* d_in, size_t in_stride, uint32_t log_n_in, uint64_t* d_out,
            size_t out_stride, uint32_t log_n, size_t ncols, bool inverse, NttDiag diag) {
    if (log_n < LOG_SEG || log_n > 26) return sipp_fail(ctx, SIPP_E_BADARG, "ntt: log_n out of range [4, 26]");
    if (log_n_in > log_n) return sipp_fail(ctx, SIPP_E_BADARG, "ntt: log_n_in > log_n");
    std::vector<uint32_t> ks = plan(log_n);
    const size_t P = ks.size();
    // lo[i] = sum of ks after i
    std::vector<uint32_t> lo(P, 0);
    for (size_t i = P; i-- > 0;) lo[i] = (i + 1 < P) ? lo[i + 1] + ks[i + 1] : 0;
    const uint32_t LT = NTT_LTILE;

    uint64_t *pw_lo = nullptr, *pw_hi = nullptr;
    uint32_t pw_L = 0;
    if (diag.base) {
        pw_L = (log_n + 1) / 2;
        // key: base and c uniquely determine the table together with the size
        uint64_t key_a = diag.base ^ (diag.c * 0x9E3779B97F4A7C15ull);
        SIPP_TRY(pow_tables(ctx, TAB_POW_LO, TAB_POW_HI, key_a, ((uint64_t)log_n << 8) | pw_L, diag.base, diag.c,
                            log_n, pw_L, &pw_lo, &pw_hi));
    }
    const uint64_t ninv = inverse ? gl::inv((uint64_t)1 << log_n) : 0;

    for (size_t step = 0; step < P; step++) {
        const size_t i = dit ? (P - 1 - step) : step;  // pass index in high->low numbering
        PassArgs a{};
        const bool first = step == 0, last = step == P - 1;
        a.in = first ? d_in : d_out;
        a.in_stride = first ? in_stride : out_stride;
        a.out = d_out;
        a.out_stride = out_stride;
        a.log_n = log_n;
        a.k = ks[i];
        a.log_m = lo[i] + ks[i];
        a.dit = dit ? 1 : 0;
        a.n_in = first ? ((uint64_t)1 << log_n_in) : ((uint64_t)1 << log_n);
        if (lo[i] > 0) {  // strided layout
            uint32_t lt = LT > a.k ? LT - a.k : 0;
            if (lt > lo[i]) lt = lo[i];
            if (lt < LOG_SEG) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "ntt: plan produced T < 16");
            a.lt = lt;
            a.lg = 0;
        } else {
            a.lt = 0;
            uint32_t lg = LT > a.k ? LT - a.k : 0;
            if (lg > log_n - a.k) lg = log_n - a.k;
            a.lg = lg;
        }
        a.wr = wr_table(ctx, a.k, inverse);
        if (!a.wr) return SIPP_E_HIP;
        // twiddle diagonal between passes: belongs to the pass that owns the HIGHER bits of the pair
        // (pass i, i < P-1): DIF applies it after pass i's butterflies, DIT before them.
        if (i + 1 < P) {
            uint32_t L = (a.log_m + 1) / 2;
            if (L < a.k) L = a.k;
            uint64_t w = gl::root_of_unity(a.log_m);
            if (inverse) w = gl::inv(w);
            uint64_t *tl, *th;
            SIPP_TRY(pow_tables(ctx, TAB_TW_LO, TAB_TW_HI, ((uint64_t)a.log_m << 8) | L, inverse, w, 0, a.log_m, L,
                                &tl, &th));
            a.tw_mode = dit ? 1 : 2;
            a.tw_L = L;
            a.tw_lo = tl;
            a.tw_hi = th;
        }
        if (diag.base && ((!dit && first) || (dit && last))) {
            a.pw_mode = dit ? 2 : 1;
            a.pw_L = pw_L;
            a.pw_lo = pw_lo;
            a.pw_hi = pw_hi;
            a.pw_step = diag.base;
        }
        if (last && ninv) a.cscale = ninv;
        SIPP_TRY(run_pass(ctx, dit ? "ntt_dit_pass" : "ntt_dif_pass", a, ncols));
    }
    return SIPP_OK;
}


// ---- fused commitment transforms: host side ----------------------------------------------------------------------
enum { TAB_COL_TW = 20, TAB_COL_PW = 21 };

// per-element twiddle table of the whole-column kernel: tab[p] = w^(t bitrev_k1(r)) c, p = r 2^k2 + t
uint64_t* col_tw_table(sipp_ctx* ctx, uint32_t log_n, uint32_t k1, bool inverse) {
    uint64_t* t = sipp_table_get(ctx, TAB_COL_TW, ((uint64_t)log_n << 8) | k1, inverse);
    if (t) return t;
    const uint32_t k2 = log_n - k1;
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> v(n);
    uint64_t w = gl::root_of_unity(log_n);
    if (inverse) w = gl::inv(w);
    const uint64_t c = inverse ? gl::inv((uint64_t)n) : 1;
    for (uint32_t r = 0; r < (1u << k1); r++) {
        const uint64_t step = gl::pow(w, (uint64_t)gl::bitrev(r, k1));
        uint64_t x = c;
        for (uint32_t tt = 0; tt < (1u << k2); tt++) {
            v[((size_t)r << k2) | tt] = x;
            x = gl::mul(x, step);
        }
    }
    if (sipp_table_put(ctx, TAB_COL_TW, ((uint64_t)log_n << 8) | k1, inverse, v, &t) != SIPP_OK) return nullptr;
    return t;
}

// coset power tables: tab[h][p] = (7 w_m^h)^p, m = n << rate_bits
uint64_t* col_pw_table(sipp_ctx* ctx, uint32_t log_n, uint32_t rate_bits) {
    uint64_t* t = sipp_table_get(ctx, TAB_COL_PW, log_n, rate_bits);
    if (t) return t;
    const size_t n = (size_t)1 << log_n;
    std::vector<uint64_t> v(n << rate_bits);
    const uint64_t wm = gl::root_of_unity(log_n + rate_bits);
    for (uint32_t h = 0; h < (1u << rate_bits); h++) {
        const uint64_t base = gl::mul(gl::GEN, gl::pow(wm, (uint64_t)h));
        uint64_t x = 1;
        for (size_t p = 0; p < n; p++) {
            v[(size_t)h * n + p] = x;
            x = gl::mul(x, base);
        }
    }
    if (sipp_table_put(ctx, TAB_COL_PW, log_n, rate_bits, v, &t) != SIPP_OK) return nullptr;
    return t;
}

int lde_column(sipp_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint64_t* d_coeffs, uint64_t* d_lde, size_t lde_stride,
               size_t ncols, uint32_t log_n, uint32_t rate_bits, bool from_coeffs) {
    ColArgs a{};
    a.in = d_in; a.coeffs = d_coeffs; a.lde = d_lde; a.in_stride = in_stride; a.lde_stride = lde_stride;
    a.log_n = log_n; a.rate_bits = rate_bits; a.from_coeffs = from_coeffs ? 1 : 0;
    a.k2 = 7; a.k1 = log_n - 7;
    a.wr1_inv = wr_table(ctx, a.k1, true); a.wr2_inv = wr_table(ctx, a.k2, true);
    a.wr1_fwd = wr_table(ctx, a.k1, false); a.wr2_fwd = wr_table(ctx, a.k2, false);
    a.tw_inv = col_tw_table(ctx, log_n, a.k1, true);
    a.tw_fwd = col_tw_table(ctx, log_n, a.k1, false);
    a.pw = col_pw_table(ctx, log_n, rate_bits);
    if (!a.wr1_inv || !a.wr2_inv || !a.wr1_fwd || !a.wr2_fwd || !a.tw_inv || !a.tw_fwd || !a.pw) return SIPP_E_HIP;
    const size_t n = (size_t)1 << log_n;
    const size_t shmem = (n + (n >> LOG_SEG) + ((size_t)1 << a.k1) + ((size_t)1 << a.k2)) * sizeof(uint64_t);
    if (shmem > 160 * 1024) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "lde_column: column does not fit LDS");
    if (shmem > 64 * 1024)
        SIPP_CHECK_HIP(ctx, hipFuncSetAttribute((const void*)lde_column_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    // threads per block: big blocks hide LDS latency when the kernel runs alone, small ones are placed sooner beside the other
    // proofs' resident hash waves; measured at n = 128 (3 proofs concurrent): 256 threads 71.2 ms, 512 72.2, 1024 75.1, 128 72.1
    const unsigned threads = n >= 16384 ? 512 : 256;
    ProfScope ps(ctx, "lde_column");
    hipLaunchKernelGGL(lde_column_kernel, dim3((unsigned)ncols), dim3(threads), shmem, ctx->stream, a);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

}  // namespace

// values [ncols][n] natural -> coefficients [ncols][n] natural + LDE [ncols][n << rate_bits] in leaf order.  Whole column in LDS for
// 2^10 .. 2^14 rows, the tree-of-rings sweeps of ntt_tree.hip from 2^15; anything else (shorter columns, d_coeffs aliasing d_values
// on a long column) is SIPP_E_UNSUPPORTED and the caller runs the pass-by-pass path (bit-reversal copy, DIT, DIF).
// Columns of 2^13 / 2^14 rows (the Fq12 STARK at n = 128 .. 256: 2524 + 1512 columns of 2^14) take the tree sweeps of ntt_tree.hip, not
// lde_column_kernel (round 5): that kernel holds the whole column in LDS -- ONE 1024-lane block per CU, a block-wide barrier per stage,
// separate twiddle passes over the tile -- and ran at half the tree sweeps' butterfly rate: 2.31 ms of transforms per n = 128 instance
// against 1.52 ms in three tree sweeps (gather | middle | contiguous), although those move 9 N words over HBM instead of 4 N.
// 2^10 .. 2^12 rows stay here (the fused tree needs a strided top sweep: 2^13).  sipp_ctx_set_kernel_routes(ctx,
// SIPP_ROUTE_LDE_COLUMN_WIDE) restores the old routing (lde_column_kernel up to 2^14 rows): the fallback stays tested
// (tests/test_gpu_generic.py).
static uint32_t lde_column_max(const sipp_ctx* ctx) { return (ctx->kernel_routes & SIPP_ROUTE_LDE_COLUMN_WIDE) ? 14u : 12u; }
// the tree sweeps move 16 bytes per lane (LDS-DMA tiles, ulonglong2 stores): every base pointer must be 16-byte aligned (column
// strides are multiples of 2^13 words there).  An 8-byte-aligned view (a tensor slice at an odd u64 offset) is refused, not
// left to the hardware's unaligned-access mode.
static bool tree_aligned(const void* a, const void* b, const void* c) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15u) == 0;
}
static int tree_misaligned(sipp_ctx* ctx) {
    return sipp_fail(ctx, SIPP_E_BADARG, "transform: device buffers of columns of 2^13 rows and more must be 16-byte aligned");
}
int sipp_lde_from_values(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t log_n,
                         uint32_t rate_bits) {
    if (log_n > lde_column_max(ctx) && log_n >= 13 && log_n <= 14 && d_values != d_coeffs)
        return tree_aligned(d_values, d_coeffs, d_lde) ? sipp_tree_lde_from_values(ctx, d_values, d_coeffs, d_lde, ncols, log_n, rate_bits)
                                                      : tree_misaligned(ctx);
    if (log_n >= 10 && log_n <= 14 && ncols <= 0x7fffffffu)
        return lde_column(ctx, d_values, (size_t)1 << log_n, d_coeffs, d_lde, (size_t)1 << (log_n + rate_bits), ncols, log_n, rate_bits, false);
    if (sipp_tree_ntt_enabled(log_n) && d_values != d_coeffs)
        return tree_aligned(d_values, d_coeffs, d_lde) ? sipp_tree_lde_from_values(ctx, d_values, d_coeffs, d_lde, ncols, log_n, rate_bits)
                                                      : tree_misaligned(ctx);
    return SIPP_E_UNSUPPORTED;
}
int sipp_lde_from_coeffs(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t log_n, uint32_t rate_bits) {
    if (log_n > lde_column_max(ctx) && log_n >= 13 && log_n <= 14)
        return tree_aligned(d_coeffs, d_lde, nullptr) ? sipp_tree_lde_from_coeffs(ctx, d_coeffs, d_lde, ncols, log_n, rate_bits) : tree_misaligned(ctx);
    if (log_n >= 10 && log_n <= 14 && ncols <= 0x7fffffffu)
        return lde_column(ctx, d_coeffs, (size_t)1 << log_n, nullptr, d_lde, (size_t)1 << (log_n + rate_bits), ncols, log_n, rate_bits, true);
    if (sipp_tree_ntt_enabled(log_n))
        return tree_aligned(d_coeffs, d_lde, nullptr) ? sipp_tree_lde_from_coeffs(ctx, d_coeffs, d_lde, ncols, log_n, rate_bits) : tree_misaligned(ctx);
    return SIPP_E_UNSUPPORTED;
}

int sipp_ntt_dif(sipp_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint32_t log_n_in, uint64_t* d_out,
                 size_t out_stride, uint32_t log_n, size_t ncols, bool inverse, NttDiag diag) {
    return ntt_run(ctx, false, d_in, in_stride, log_n_in, d_out, out_stride, log_n, ncols, inverse, diag);
}

int sipp_ntt_dit(sipp_ctx* ctx, uint64_t* d_io, size_t stride, uint32_t log_n, size_t ncols, bool inverse,
                 NttDiag diag) {
    return ntt_run(ctx, true, d_io, stride, log_n, d_io, stride, log_n, ncols, inverse, diag);
}

int sipp_bitrev_cols(sipp_ctx* ctx, const uint64_t* d_in, size_t in_stride, uint64_t* d_out, size_t out_stride,
                     uint32_t log_n, size_t ncols) {
    size_t total = ((size_t)1 << log_n) * ncols;
    unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (!grid) return SIPP_OK;
    ProfScope ps(ctx, "bitrev_cols");
    if (log_n >= 2 * TB && ncols <= 65535) {
        hipLaunchKernelGGL(bitrev_tiled_kernel, dim3(1u << (log_n - 2 * TB), (unsigned)ncols), dim3(256), 0, ctx->stream, d_in,
                           in_stride, d_out, out_stride, log_n);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        return SIPP_OK;
    }
    hipLaunchKernelGGL(bitrev_cols_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_in, in_stride, d_out, out_stride,
                       log_n, ncols);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}
