// sipp_amd/csrc/ntt_tree.hip -- commitment transforms for LONG columns (N >= 2^18) as a tree of polynomial rings, gfx950.
//
// What it replaces: plonky2's PolynomialBatch::from_values (ifft, then coset LDE; field/src/fft.rs, fri/oracle.rs @ 541e127,
// reached from the reference through starky::prover::prove behind src/verifier_circuit.rs:133-135) for the sizes where a column
// fits neither LDS (ntt.hip lde_column) nor the three-sweep kernels -- the n = 1024 / n = 4096 configurations of BASELINE.json
// and every IO shard of an 8-GPU run at n = 4096 (N = 2^18).
//
// Why another formulation: the pass-by-pass path of ntt.hip is a four-step FFT -- every sweep over HBM ends
// (or starts) with a diagonal of twiddles w^(i0 o1) that costs TWO modular products per element (one to advance the running
// power, one to apply it), the coset shift 7^j is a third diagonal, and the inverse needs a bit-reversal copy first.  At
// N = 2^21 that is 9 sweeps, 7 diagonals and 152 N bytes per column, and the kernels are bound by VALU issue, not by HBM.
// Here the transform is the tree of rings  F[x]/(x^n - r)  ->  F[x]/(x^(n/2) - s) x F[x]/(x^(n/2) + s),  s^2 = r:
//   forward   (u, v) -> (u + s v, u - s v)         one product per butterfly, the twiddle belongs to the BLOCK (tree node),
//   inverse   (a, b) -> (a + b, (a - b) / s)       not to the position inside it: no diagonal between sweeps at all.
// The leaves come out in bit-reversed order, i.e. directly in the leaf order of the LDE buffer.  Evaluating on
// the coset 7 <w> is the same tree with root r = 7^n: the shift lives in the node constants and costs nothing.  With blowup 2^b
// the top b levels see a zero upper half and are plain copies: the LDE is 2^b INDEPENDENT size-N trees over the same
// coefficients, written to the 2^b halves of the LDE column.  The inverse reads natural-order values through a tile that is
// 16 blocks apart in the top four position bits (= the low four bits of the natural index: full 128-byte lines), so the
// bit-reversal copy disappears as well.
//
// Node constants: node (level l, block i) of a size-2^L tree with root constant g^(2^L) has  s = g^(2^(L-1-l)) T[i],
// T[i] = w_(2^(l+1))^bitrev(i, l)  -- T does not depend on l (prefix property), so the plain tree (g = 1, the inverse) needs one flat
// table of n/2 entries, and a coset tree a heap-ordered table (index (Q << l) + i under the subtree root Q).  Tables are shared by
// every column; the grid is ordered column-fastest so that the blocks in flight work on the same few KB of them.
#include "ctx.hpp"
#include "gl_lazy.hpp"

namespace {

// forward butterfly (u, v) -> (u + s v, u - s v): u may be ANY u64 congruent to its value, the product is made canonical, the
// sum and the difference take one conditional correction each and stay in [0, 2^64) (gl_lazy.hpp); what leaves the last sweep is
// canonicalised at the store
__device__ __forceinline__ void bfly_fwd(uint64_t& u, uint64_t& v, uint64_t s) {
    const uint64_t w = gll::canon(gl::mul_nc(v, s)), a = u;
    u = gll::add_nc(a, w);
    v = gll::sub_nc(a, w);
}
// inverse butterfly (a, b) -> (a + b, (a - b) / s) on canonical values
__device__ __forceinline__ void bfly_inv(uint64_t& a, uint64_t& b, uint64_t sinv) {
    const uint64_t u = a, v = b;
    a = gl::add(u, v);
    b = gll::canon(gl::mul_nc(gll::sub_nc(u, v), sinv));
}

constexpr int LOG_SEG = 4;
__device__ __forceinline__ uint32_t lds_idx(uint32_t e) { return e + (e >> LOG_SEG); }

struct IdxPlain {
    __device__ __forceinline__ uint32_t operator()(uint32_t e) const { return lds_idx(e); }
};

#ifndef SIPP_NTT_MLP
#define SIPP_NTT_MLP 8
#endif

template <int U, class Load, class Use>
__device__ __forceinline__ void tile_loop(uint32_t E, Load load, Use use) {
    const uint32_t step = blockDim.x;
    uint32_t e = threadIdx.x;
    for (; e + (U - 1) * step < E; e += U * step) {
        uint64_t v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = load(e + u * step);
#pragma unroll
        for (int u = 0; u < U; u++) use(e + u * step, v[u]);
    }
    for (; e < E; e += step) use(e, load(e));
}

// node constants of the tile's levels: level `lev` below the tile's root (lev = 0: one block per group), block `blk` of group g
struct TwLds {          // staged in LDS: [G][R] local heaps (entry 0 of a heap unused)
    const uint64_t* tws;
    uint32_t k;
    __device__ __forceinline__ uint64_t operator()(uint32_t g, uint32_t lev, uint32_t blk) const {
        return tws[(g << k) + (1u << lev) + blk];
    }
};
struct TwGlobal {       // straight from the table (L2): the deep levels use every constant once or twice
    const uint64_t* tw;
    uint32_t z0, zstride;
    __device__ __forceinline__ uint64_t operator()(uint32_t g, uint32_t lev, uint32_t blk) const {
        return tw[((z0 + g * zstride) << lev) + blk];
    }
};

// one round of S levels (row strides 2^D .. 2^(D+S-1)) in registers: a lane owns the 2^S rows r0 + i 2^D of one column
template <bool FWD, int S, class Idx, class Tw>
__device__ __forceinline__ void tree_round(uint64_t* tile, uint32_t E, uint32_t k, uint32_t lt, uint32_t D, Idx lidx, Tw tw) {
    constexpr int M = 1 << S;
    const uint32_t T = 1u << lt, R = 1u << k, q = 1u << D;
    const uint32_t items = E >> S;
    for (uint32_t idx = threadIdx.x; idx < items; idx += blockDim.x) {
        const uint32_t t = idx & (T - 1);
        const uint32_t rest = idx >> lt;
        const uint32_t b = rest & ((R >> S) - 1);
        const uint32_t g = rest >> (k - S);
        const uint32_t j = b & (q - 1);
        const uint32_t r0 = ((b >> D) << (D + S)) | j;
        const uint32_t e0 = (((g << k) | r0) << lt) | t;
        const uint32_t st = q << lt;
        const uint32_t top = r0 >> (D + S);      // block index of the round's highest level, before the lane's own bits
        uint64_t x[M];
#pragma unroll
        for (int i = 0; i < M; i++) x[i] = tile[lidx(e0 + i * st)];
        if (FWD) {
#pragma unroll
            for (int sg = S - 1; sg >= 0; sg--) {
                const uint32_t lev = k - 1 - (D + sg);
#pragma unroll
                for (int i = 0; i < M; i++) {
                    if (i & (1 << sg)) continue;
                    const uint64_t w = tw(g, lev, (top << (S - 1 - sg)) + (uint32_t)(i >> (sg + 1)));
                    bfly_fwd(x[i], x[i + (1 << sg)], w);
                }
            }
        } else {
#pragma unroll
            for (int sg = 0; sg < S; sg++) {
                const uint32_t lev = k - 1 - (D + sg);
#pragma unroll
                for (int i = 0; i < M; i++) {
                    if (i & (1 << sg)) continue;
                    const uint64_t w = tw(g, lev, (top << (S - 1 - sg)) + (uint32_t)(i >> (sg + 1)));
                    bfly_inv(x[i], x[i + (1 << sg)], w);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < M; i++) tile[lidx(e0 + i * st)] = x[i];
    }
    __syncthreads();
}

// all k levels of a pass on an LDS tile of E = G R T elements, element e = ((g << k | r) << lt) | t
// d0: the INVERSE runs only the levels with row strides 2^d0 .. 2^(k-1) (the fused middle sweep: the lower row bits belong to
// an earlier inverse sweep and act as extra columns)
// and the FORWARD only the levels with row strides 2^(k-1) .. 2^d0 (the tile's lower row bits are left to the next sweep)
template <bool FWD, class Idx, class Tw>
__device__ __forceinline__ void tree_stages(uint64_t* tile, uint32_t E, uint32_t k, uint32_t lt, Idx lidx, Tw tw, uint32_t d0 = 0) {
    if (FWD) {
        uint32_t top = k;
        for (; top >= d0 + 4; top -= 4) tree_round<true, 4>(tile, E, k, lt, top - 4, lidx, tw);
        if (top >= d0 + 2) {
            tree_round<true, 2>(tile, E, k, lt, top - 2, lidx, tw);
            top -= 2;
        }
        if (top > d0) tree_round<true, 1>(tile, E, k, lt, d0, lidx, tw);
    } else {
        uint32_t bot = d0, rem = k - d0;
        for (; rem >= 4; rem -= 4, bot += 4) tree_round<false, 4>(tile, E, k, lt, bot, lidx, tw);
        if (rem >= 2) {
            tree_round<false, 2>(tile, E, k, lt, bot, lidx, tw);
            bot += 2;
            rem -= 2;
        }
        if (rem) tree_round<false, 1>(tile, E, k, lt, bot, lidx, tw);
    }
}

struct TreeArgs {
    const uint64_t* in;
    uint64_t* out;
    size_t in_stride, out_stride;         // per column (elements)
    size_t in_half, out_half;             // per subtree of blockIdx.y (elements)
    uint32_t L;                           // log2 positions of one (sub)tree
    uint32_t lo, k;                       // the pass owns position bits [lo, lo + k)
    uint32_t lt, lg;                      // tile = 2^k rows x 2^lt columns (lo > 0)  or  2^lg groups of 2^k positions (lo == 0)
    uint32_t q0;                          // root of subtree blockIdx.y in the table: q0 + blockIdx.y (0 = flat table of the plain tree)
    const uint64_t* tw;
    uint64_t scale;                       // 0 = none; applied to every element at the store
    uint32_t ncols, colfast;
};

template <bool FWD, bool TWLDS>
__global__ void __launch_bounds__(256) tree_pass_kernel(TreeArgs a) {
    extern __shared__ uint64_t smem[];
    const uint32_t k = a.k, lt = a.lt, lg = a.lg, lo = a.lo;
    const uint32_t log_e = k + lt + lg;
    const uint32_t E = 1u << log_e, T = 1u << lt;
    uint64_t* tile = smem;
    uint64_t* tws = smem + (E + (E >> LOG_SEG));
    const uint32_t tiles_per_col = 1u << (a.L - log_e);
    uint32_t col, tix;
    if (a.colfast) {
        tix = blockIdx.x / a.ncols;
        col = blockIdx.x - tix * a.ncols;
    } else {
        col = blockIdx.x / tiles_per_col;
        tix = blockIdx.x - col * tiles_per_col;
    }
    const uint64_t* in = a.in + (size_t)col * a.in_stride + (size_t)blockIdx.y * a.in_half;
    uint64_t* out = a.out + (size_t)col * a.out_stride + (size_t)blockIdx.y * a.out_half;
    const uint32_t Q = a.q0 ? a.q0 + blockIdx.y : 0;
    // tile -> positions.  strided (lo > 0): tix = hi 2^(lo - lt) + c,  p(r, t) = hi 2^(lo + k) + r 2^lo + c T + t
    //                     contiguous:       p(e) = tix E + e
    uint32_t base, hi;
    if (lo) {
        hi = tix >> (lo - lt);
        base = (hi << (lo + k)) + ((tix & ((1u << (lo - lt)) - 1)) << lt);
    } else {
        base = tix << log_e;
        hi = tix << lg;
    }
    auto pos_of = [&](uint32_t e) -> uint32_t { return lo ? base + ((e >> lt) << lo) + (e & (T - 1)) : base + e; };
    const uint32_t z0 = (Q << (a.L - lo - k)) + hi;   // the tile's (first group's) root node
    tile_loop<SIPP_NTT_MLP>(
        E, [&](uint32_t e) -> uint64_t { return in[pos_of(e)]; }, [&](uint32_t e, uint64_t v) { tile[lds_idx(e)] = v; });
    if (TWLDS) {
        const uint32_t R = 1u << k;
        for (uint32_t i = threadIdx.x; i < (R << lg); i += blockDim.x) {
            const uint32_t g = i >> k, J = i & (R - 1);
            if (!J) continue;
            const uint32_t lev = 31 - __clz(J);
            tws[i] = a.tw[((z0 + g) << lev) + (J - (1u << lev))];
        }
    }
    __syncthreads();
    if (TWLDS)
        tree_stages<FWD>(tile, E, k, lt, IdxPlain{}, TwLds{tws, k});
    else
        tree_stages<FWD>(tile, E, k, lt, IdxPlain{}, TwGlobal{a.tw, z0, 1});
    const uint64_t sc = a.scale;
    tile_loop<SIPP_NTT_MLP>(
        E, [&](uint32_t e) -> uint64_t { return tile[lds_idx(e)]; },
        [&](uint32_t e, uint64_t v) { out[pos_of(e)] = sc ? gl::mul(v, sc) : FWD ? gl::canon(v) : v; });
}

// ---- round 5: the sweeps over full 2^12-element tiles with the tile loaded by LDS-DMA ------------------------------------------------
// global_load_lds_dwordx4 has no VGPR destination: the whole 32-KB tile of a block is in flight at once (the register-staged tile_loop
// keeps 8 x 8 bytes per lane in flight) and the load phase costs no VALU work.  The DMA writes LDS lane-linear (base + 16 lane), so the
// tile cannot be padded: it is XOR-swizzled on element bits 1 .. 4 (pairs of elements stay together, 16-byte aligned) through the
// per-lane SOURCE address -- exactly 32 KB, five blocks per CU -- and wherever a lane owns consecutive elements (the contiguous sweep's
// last round, every store phase) an LDS access moves 16 bytes.  Measured (profiles/r05_ab_tree_dma.txt): contiguous sweep 3.84 -> 3.15 ms
// at 2^18 x 1024, 0.95 -> 0.74 ms at 2^16 x 1024.  A persistent DOUBLE-buffered form (tile k + 1 streaming in during tile k's rounds;
// 64 KB per block, two blocks per CU) was slower than the round-4 kernel at every size: same file.
__device__ __forceinline__ uint32_t swz(uint32_t e) { return e ^ (((e >> 5) & 7u) << 1) ^ (((e >> 8) & 1u) << 4); }
struct IdxSwz {
    __device__ __forceinline__ uint32_t operator()(uint32_t e) const { return swz(e); }
};
struct DmaArgs {
    const uint64_t* in;
    uint64_t* out;
    size_t in_stride, out_stride, in_half, out_half;
    uint32_t L, q0, ncols, halves;
    const uint64_t* tw;
    uint32_t work;               // tiles x columns x halves
};

__device__ __forceinline__ void dma_decode(const DmaArgs& a, uint32_t w, uint32_t& tix, uint32_t& col, uint32_t& h) {
    const uint32_t per_half = a.ncols << (a.L - 12);
    h = w / per_half;
    const uint32_t r = w - h * per_half;
    tix = r / a.ncols;             // column-fastest: the blocks in flight share node constants
    col = r - tix * a.ncols;
}

__device__ __forceinline__ void dma_issue(const DmaArgs& a, uint32_t w, uint64_t* buf) {
    uint32_t tix, col, h;
    dma_decode(a, w, tix, col, h);
    const uint64_t* src = a.in + (size_t)col * a.in_stride + (size_t)h * a.in_half + ((size_t)tix << 12);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t P = ((c * 4 + wave) << 6) + lane;                           // pair slot in LDS (lane-linear inside the piece)
        const uint32_t pe = P ^ ((P >> 4) & 7u) ^ (((P >> 7) & 1u) << 3);          // the pair of elements that lives there
        // inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell the two buffers of one LDS array apart and would
        // wait vmcnt(0) before the first ds_read after a DMA it knows of -- the transfer would never overlap the butterflies.  M0 = the
        // piece's LDS byte address (wave-uniform), saved and restored around the instruction (cdna_hip_programming.md, LDS-DMA recipe)
        const uint32_t dst = __builtin_amdgcn_readfirstlane(
            (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(buf + (((c * 4 + wave) << 6) << 1)));
        const uint64_t* g = src + 2 * pe;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
}

// the tile's top four levels (row strides 2^8 .. 2^11): fifteen node constants, the same for every lane -- scalar loads, so that no
// vector-memory wait (which would drain the DMA in flight: vmcnt counts in order) sits in this round
__device__ __forceinline__ void tree_round_top(uint64_t* tile, const uint64_t* tw_, uint32_t z0) {
    // the table is read-only for the whole launch: constant address space, so that uniform addresses become s_load (lgkmcnt, not vmcnt)
    const __attribute__((address_space(4))) uint64_t* tw = (const __attribute__((address_space(4))) uint64_t*)tw_;
    const uint32_t j = threadIdx.x;
    uint64_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = tile[swz(j + 256 * i)];
#pragma unroll
    for (int sg = 3; sg >= 0; sg--) {
        const uint32_t lev = 3 - sg;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i & (1 << sg)) continue;
            const uint64_t w = tw[(z0 << lev) + (uint32_t)(i >> (sg + 1))];
            bfly_fwd(x[i], x[i + (1 << sg)], w);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) tile[swz(j + 256 * i)] = x[i];
    __syncthreads();
}

// the last four levels: a lane owns sixteen consecutive elements = eight 16-byte pairs
template <class Tw>
__device__ __forceinline__ void tree_round_last(uint64_t* tile, uint32_t k, Tw tw) {
    const uint32_t b = threadIdx.x;
    const uint32_t e0 = b << 4;
    uint64_t x[16];
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + swz(e0 + 2 * m));
        x[2 * m] = v.x;
        x[2 * m + 1] = v.y;
    }
#pragma unroll
    for (int sg = 3; sg >= 0; sg--) {
        const uint32_t lev = k - 1 - sg;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i & (1 << sg)) continue;
            const uint64_t w = tw(0, lev, (b << (3 - sg)) + (uint32_t)(i >> (sg + 1)));
            bfly_fwd(x[i], x[i + (1 << sg)], w);
        }
    }
#pragma unroll
    for (int m = 0; m < 8; m++) {
        ulonglong2 v;
        v.x = x[2 * m];
        v.y = x[2 * m + 1];
        *reinterpret_cast<ulonglong2*>(tile + swz(e0 + 2 * m)) = v;
    }
    __syncthreads();
}

// the contiguous forward sweep (the tile's twelve leaf-most levels), one tile per block
__global__ void __launch_bounds__(256) tree_fwd_dma_kernel(DmaArgs a) {
    extern __shared__ uint64_t smem[];
    constexpr uint32_t E = 4096, K = 12;
    const uint32_t w = blockIdx.x;
    uint64_t* tile = smem;
    dma_issue(a, w, tile);
    uint32_t tix, col, h;
    dma_decode(a, w, tix, col, h);
    const uint32_t z0 = ((a.q0 + h) << (a.L - K)) + tix;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tree_round_top(tile, a.tw, __builtin_amdgcn_readfirstlane(z0));
    tree_round<true, 4>(tile, E, K, 0, 4, IdxSwz{}, TwGlobal{a.tw, z0, 1});
    tree_round_last(tile, K, TwGlobal{a.tw, z0, 1});
    uint64_t* out = a.out + (size_t)col * a.out_stride + (size_t)h * a.out_half + ((size_t)tix << 12);
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t pe = c * 256 + threadIdx.x;
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + swz(2 * pe));
        v.x = gl::canon(v.x);
        v.y = gl::canon(v.y);
        *reinterpret_cast<ulonglong2*>(out + 2 * pe) = v;
    }
}

// strided sweeps (lo > 0: tile = 2^k rows x 2^lt columns, lt >= 1 so that a 16-byte pair is contiguous in HBM), either direction:
// DMA load through the per-lane source address, node constants staged in LDS behind the tile, 16-byte stores
template <bool FWD>
__global__ void __launch_bounds__(256) tree_pass_dma_kernel(TreeArgs a) {
    extern __shared__ uint64_t smem[];
    const uint32_t k = a.k, lt = a.lt, lo = a.lo;
    constexpr uint32_t E = 4096;
    const uint32_t T = 1u << lt;
    uint64_t* tile = smem;
    uint64_t* tws = smem + E;
    uint32_t col, tix;
    tix = blockIdx.x / a.ncols;
    col = blockIdx.x - tix * a.ncols;
    const uint64_t* in = a.in + (size_t)col * a.in_stride + (size_t)blockIdx.y * a.in_half;
    uint64_t* out = a.out + (size_t)col * a.out_stride + (size_t)blockIdx.y * a.out_half;
    const uint32_t Q = a.q0 ? a.q0 + blockIdx.y : 0;
    const uint32_t hi = tix >> (lo - lt);
    const uint32_t base = (hi << (lo + k)) + ((tix & ((1u << (lo - lt)) - 1)) << lt);
    auto pos_of = [&](uint32_t e) -> uint32_t { return base + ((e >> lt) << lo) + (e & (T - 1)); };
    const uint32_t z0 = (Q << (a.L - lo - k)) + hi;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t P = ((c * 4 + wave) << 6) + lane;
        const uint32_t pe = P ^ ((P >> 4) & 7u) ^ (((P >> 7) & 1u) << 3);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(
            (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(tile + (((c * 4 + wave) << 6) << 1)));
        const uint64_t* g = in + pos_of(2 * pe);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
    {
        const uint32_t R = 1u << k;
        for (uint32_t J = threadIdx.x; J < R; J += blockDim.x) {
            if (!J) continue;
            const uint32_t lev = 31 - __clz(J);
            tws[J] = a.tw[(z0 << lev) + (J - (1u << lev))];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tree_stages<FWD>(tile, E, k, lt, IdxSwz{}, TwLds{tws, k});
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t pe = c * 256 + threadIdx.x;
        ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + swz(2 * pe));
        if (FWD) {
            v.x = gl::canon(v.x);
            v.y = gl::canon(v.y);
        }
        *reinterpret_cast<ulonglong2*>(out + pos_of(2 * pe)) = v;
    }
}

// Fused MIDDLE sweep: the inverse's top ki levels and, on the same tile, the top kf levels of every half's forward tree.
// (Round 5: tile by LDS-DMA, swizzled, 16-byte accesses where a lane owns a pair: 3.28 -> 2.96 ms at 2^18 x 1024; forcing four waves
// per SIMD costs 13 spilled registers and returns less: 3.08 ms.)
// Tile = 2^max(ki, kf) rows (the top position bits) x 2^lt columns of a column of coefficients-to-be.  Reads N, writes N (coefficients,
// scaled by 1/N) + 2^rate_bits N (first forward sweep of every half) -- instead of an inverse sweep (16 N) plus a forward sweep that
// reads the coefficients once per half (16 N + 16 N at blowup 2).  The scaled coefficients wait in registers (16 per lane) while the
// LDS tile runs one half after the other.
struct MidArgs {
    uint64_t* coeffs;                  // [ncols][n]  in: after the inverse's earlier sweeps; out: natural coefficients
    uint64_t* lde;                     // [ncols][n << rate_bits]
    uint32_t L, kf, ki, lt, rate_bits;
    const uint64_t* tw_inv;            // flat inverse table
    const uint64_t* tw_fwd;            // coset heap table of size 2^(L + rate_bits)
    uint64_t scale;                    // 1 / n
    uint32_t ncols, colfast;
};

__global__ void __launch_bounds__(256) tree_mid_kernel(MidArgs a) {
    extern __shared__ uint64_t smem[];
    const uint32_t k = a.kf > a.ki ? a.kf : a.ki, lt = a.lt, lo = a.L - k;
    constexpr uint32_t E = 4096;                           // k + lt = 12; lt >= 4 (k <= 8): a 16-byte pair is contiguous in HBM
    const uint32_t T = 1u << lt, R = 1u << k;
    uint64_t* tile = smem;                                 // swizzled like the DMA tiles (swz), no padding
    uint64_t* twi = smem + E;
    uint64_t* twf = twi + R;
    const uint32_t tiles_per_col = 1u << (lo - lt);
    uint32_t col, tix;
    if (a.colfast) {
        tix = blockIdx.x / a.ncols;
        col = blockIdx.x - tix * a.ncols;
    } else {
        col = blockIdx.x / tiles_per_col;
        tix = blockIdx.x - col * tiles_per_col;
    }
    const size_t n = (size_t)1 << a.L;
    uint64_t* co = a.coeffs + (size_t)col * n;
    uint64_t* lde = a.lde + ((size_t)col << (a.L + a.rate_bits));
    const uint32_t base = tix << lt;                       // the top pass: no bits above the tile's rows
    auto pos_of = [&](uint32_t e) -> uint32_t { return base + ((e >> lt) << lo) + (e & (T - 1)); };
    {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {                 // the tile by LDS-DMA (see tree_pass_dma_kernel)
            const uint32_t P = ((c * 4 + wave) << 6) + lane;
            const uint32_t pe = P ^ ((P >> 4) & 7u) ^ (((P >> 7) & 1u) << 3);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(size_t)(__attribute__((address_space(3))) void*)(tile + (((c * 4 + wave) << 6) << 1)));
            const uint64_t* g = co + pos_of(2 * pe);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
        }
    }
    for (uint32_t J = threadIdx.x; J < R; J += blockDim.x) {
        if (!J) continue;
        const uint32_t lev = 31 - __clz(J);
        twi[J] = a.tw_inv[J - (1u << lev)];                // root node 0 of the flat table
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tree_stages<false>(tile, E, k, lt, IdxSwz{}, TwLds{twi, k}, k - a.ki);
    // coefficients: scaled, stored, kept (eight 16-byte pairs per lane)
    constexpr int PER = 8;
    ulonglong2 c[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t e = 2 * (threadIdx.x + u * 256);
        c[u] = *reinterpret_cast<const ulonglong2*>(tile + swz(e));
        c[u].x = gl::mul(c[u].x, a.scale);
        c[u].y = gl::mul(c[u].y, a.scale);
    }
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const uint32_t e = 2 * (threadIdx.x + u * 256);
        *reinterpret_cast<ulonglong2*>(co + pos_of(e)) = c[u];
    }
    const uint32_t halves = 1u << a.rate_bits;
    for (uint32_t h = 0; h < halves; h++) {
        __syncthreads();                                   // the previous half's readers (and twf readers) are done with LDS
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t e = 2 * (threadIdx.x + u * 256);
            *reinterpret_cast<ulonglong2*>(tile + swz(e)) = c[u];
        }
        const uint32_t Q = halves + h;
        for (uint32_t J = threadIdx.x; J < R; J += blockDim.x) {
            if (!J) continue;
            const uint32_t lev = 31 - __clz(J);
            twf[J] = a.tw_fwd[(Q << lev) + (J - (1u << lev))];
        }
        __syncthreads();
        tree_stages<true>(tile, E, k, lt, IdxSwz{}, TwLds{twf, k}, k - a.kf);
        uint64_t* out = lde + (size_t)h * n;
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const uint32_t e = 2 * (threadIdx.x + u * 256);
            ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + swz(e));
            v.x = gl::canon(v.x);
            v.y = gl::canon(v.y);
            *reinterpret_cast<ulonglong2*>(out + pos_of(e)) = v;
        }
    }
}

// inverse, first sweep: natural-order values -> the k2 leaf-most levels.  Tile = 16 groups of 2^k2 positions that differ in their
// top four position bits, i.e. in the low four bits of the natural index: every global read is a full 128-byte line.
// (Round 5: node constants of both rounds requested ahead of the tile as 16-byte loads, swizzled 32-KB tile, 16-byte LDS / store
// accesses: 1.62 -> 1.20 ms at 2^18 x 1024, 0.40 -> 0.30 ms at 2^16 x 1024.)
struct GatherArgs {
    const uint64_t* in;     // [ncols][n] values, natural order
    uint64_t* out;          // [ncols][n] positions (leaf order) after the k2 lowest levels
    uint32_t L, k2;
    const uint64_t* tw;     // flat inverse table
    uint32_t ncols, colfast;
};

// tile layout: element e = (g << 8) | r at gswz(e) -- the swizzle of the DMA tiles plus the group's upper three bits on element bits
// 1 .. 3, so that the load phase (a wave writes sixteen groups at once) spreads over the banks as well; exactly 32 KB
__device__ __forceinline__ uint32_t gswz(uint32_t e) { return swz(e) ^ (((e >> 9) & 7u) << 1); }

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) tree_gather_kernel(GatherArgs a) {
    extern __shared__ uint64_t tile[];
    constexpr uint32_t k2 = 8, E = 16u << k2;
    const uint32_t mid_bits = a.L - 4 - k2;
    uint32_t col, mid;
    if (a.colfast) {
        mid = blockIdx.x / a.ncols;
        col = blockIdx.x - mid * a.ncols;
    } else {
        col = blockIdx.x >> mid_bits;
        mid = blockIdx.x & ((1u << mid_bits) - 1);
    }
    const uint64_t* in = a.in + ((size_t)col << a.L);
    uint64_t* out = a.out + ((size_t)col << a.L);
    const uint32_t nat_mid = gl::bitrev(mid, mid_bits) << 4;
    // lane (g, b): group g = top four position bits, its blocks hang under node zg of level L - k2
    const uint32_t b = threadIdx.x & 15, g = threadIdx.x >> 4;
    const uint32_t zg = (g << mid_bits) + mid;
    // node constants of BOTH rounds are requested before the tile: round 1's (levels 7 .. 4 below zg: 8 + 4 + 2 + 1 consecutive
    // entries per lane) come back ahead of the tile, round 2's (levels 3 .. 0: fifteen per group) wait in registers -- no round starts
    // with a table access in its critical path (the round-4 kernel: fifteen 8-byte loads at the head of each round)
    uint64_t w7[8], w6[4], w5[2], w4, wt[15];
    {
        const ulonglong2* p7 = reinterpret_cast<const ulonglong2*>(a.tw + ((size_t)zg << 7) + 8 * b);
        const ulonglong2* p6 = reinterpret_cast<const ulonglong2*>(a.tw + ((size_t)zg << 6) + 4 * b);
        const ulonglong2* p5 = reinterpret_cast<const ulonglong2*>(a.tw + ((size_t)zg << 5) + 2 * b);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const ulonglong2 v = p7[m];
            w7[2 * m] = v.x;
            w7[2 * m + 1] = v.y;
        }
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const ulonglong2 v = p6[m];
            w6[2 * m] = v.x;
            w6[2 * m + 1] = v.y;
        }
        const ulonglong2 v5 = p5[0];
        w5[0] = v5.x;
        w5[1] = v5.y;
        w4 = a.tw[((size_t)zg << 4) + b];
    }
    tile_loop<SIPP_NTT_MLP>(
        E,
        [&](uint32_t e) -> uint64_t {
            const uint32_t gq = e & 15, r = e >> 4;
            return in[(gl::bitrev(r, k2) << (a.L - k2)) | nat_mid | gq];
        },
        [&](uint32_t e, uint64_t v) {
            const uint32_t gq = e & 15, r = e >> 4;
            tile[gswz((gl::bitrev(gq, 4) << k2) | r)] = v;
        });
#pragma unroll
    for (int lev = 0; lev < 4; lev++)
#pragma unroll
        for (int q = 0; q < (1 << lev); q++) wt[(1 << lev) - 1 + q] = a.tw[((size_t)zg << lev) + q];
    __syncthreads();
    uint64_t x[16];
    // round 1: position strides 1 .. 8 -- the lane's sixteen consecutive positions, 16 bytes per LDS access
    {
        const uint32_t e0 = (g << k2) | (b << 4);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + gswz(e0 + 2 * m));
            x[2 * m] = v.x;
            x[2 * m + 1] = v.y;
        }
#pragma unroll
        for (int i = 0; i < 16; i += 2) bfly_inv(x[i], x[i + 1], w7[i >> 1]);
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (!(i & 2)) bfly_inv(x[i], x[i + 2], w6[i >> 2]);
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (!(i & 4)) bfly_inv(x[i], x[i + 4], w5[i >> 3]);
#pragma unroll
        for (int i = 0; i < 8; i++) bfly_inv(x[i], x[i + 8], w4);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            ulonglong2 v;
            v.x = x[2 * m];
            v.y = x[2 * m + 1];
            *reinterpret_cast<ulonglong2*>(tile + gswz(e0 + 2 * m)) = v;
        }
    }
    __syncthreads();
    // round 2: position strides 16 .. 128; the constants depend on the group alone
    {
        const uint32_t e0 = (g << k2) | b;
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = tile[gswz(e0 + 16 * i)];
#pragma unroll
        for (int sg = 0; sg < 4; sg++) {
            const int lev = 3 - sg;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (i & (1 << sg)) continue;
                bfly_inv(x[i], x[i + (1 << sg)], wt[(1 << lev) - 1 + (i >> (sg + 1))]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) tile[gswz(e0 + 16 * i)] = x[i];
    }
    __syncthreads();
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
        const uint32_t e = 2 * (c * 256 + threadIdx.x);
        const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(tile + gswz(e));
        *reinterpret_cast<ulonglong2*>(out + (((size_t)(e >> k2)) << (a.L - 4) | ((size_t)mid << k2) | (e & 255u))) = v;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
enum { TAB_TREE_FLAT = 30, TAB_TREE_COSET = 31 };

// T[i] = w^(+-bitrev(i, L - 1)), w = w_(2^L), i < 2^(L-1)
uint64_t* tree_flat_table(sipp_ctx* ctx, uint32_t L, bool inverse) {
    uint64_t* t = sipp_table_get(ctx, TAB_TREE_FLAT, L, inverse);
    if (t) return t;
    const size_t h = (size_t)1 << (L - 1);
    std::vector<uint64_t> pw(h), v(h);
    uint64_t w = gl::root_of_unity(L);
    if (inverse) w = gl::inv(w);
    uint64_t x = 1;
    for (size_t i = 0; i < h; i++) {
        pw[i] = x;
        x = gl::mul(x, w);
    }
    for (size_t i = 0; i < h; i++) v[i] = pw[gl::bitrev((uint32_t)i, L - 1)];
    if (sipp_table_put(ctx, TAB_TREE_FLAT, L, inverse, v, &t) != SIPP_OK) return nullptr;
    return t;
}

// heap table of the coset tree of size m = 2^LM, shift g = 7: C[2^l + i] = 7^(m / 2^(l+1)) T[i].  One table serves every split of LM
// into (column length, blowup): the levels above the blowup are simply never read (zero upper halves: copies)
uint64_t* tree_coset_table(sipp_ctx* ctx, uint32_t LM) {
    uint64_t* t = sipp_table_get(ctx, TAB_TREE_COSET, LM, 0);
    if (t) return t;
    const size_t m = (size_t)1 << LM, h = m >> 1;
    std::vector<uint64_t> pw(h), v(m, 0);
    const uint64_t w = gl::root_of_unity(LM);
    uint64_t x = 1;
    for (size_t i = 0; i < h; i++) {
        pw[i] = x;
        x = gl::mul(x, w);
    }
    for (uint32_t l = 0; l < LM; l++) {
        const uint64_t gs = gl::pow(gl::GEN, (uint64_t)1 << (LM - 1 - l));
        for (size_t i = 0; i < ((size_t)1 << l); i++)
            v[((size_t)1 << l) + i] = gl::mul(gs, pw[gl::bitrev((uint32_t)i, LM - 1)]);
    }
    if (sipp_table_put(ctx, TAB_TREE_COSET, LM, 0, v, &t) != SIPP_OK) return nullptr;
    return t;
}

// grid order: column-fastest, so that the blocks in flight share the same few KB of node constants (tile-fastest measured equal:
// 6.77 against 6.77 ms)
constexpr int tree_colfast() { return 1; }

// bits of the strided sweeps above the first `low` position bits, highest first, at most 8 each
std::vector<uint32_t> split_bits(uint32_t rem) {
    std::vector<uint32_t> ks;
    if (!rem) return ks;
    uint32_t q = (rem + 7) / 8;
    for (uint32_t i = 0; i < q; i++) {
        const uint32_t ki = rem / (q - i) + ((rem % (q - i)) ? 1 : 0);
        ks.push_back(ki);
        rem -= ki;
    }
    return ks;
}

int launch_fwd_dma(sipp_ctx* ctx, const TreeArgs& t, unsigned halves) {
    DmaArgs a{};
    a.in = t.in; a.out = t.out; a.in_stride = t.in_stride; a.out_stride = t.out_stride; a.in_half = t.in_half; a.out_half = t.out_half;
    a.L = t.L; a.q0 = t.q0; a.ncols = t.ncols; a.halves = halves; a.tw = t.tw;
    const size_t work = ((size_t)t.ncols << (t.L - 12)) * halves;
    if (work > 0x7fffffffull) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "tree ntt: too many tiles for one launch");
    a.work = (uint32_t)work;
    ProfScope ps(ctx, "ntt_tree_fwd");
    hipLaunchKernelGGL(tree_fwd_dma_kernel, dim3((unsigned)work), dim3(256), 4096 * sizeof(uint64_t), ctx->stream, a);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

template <bool FWD, bool TWLDS>
int launch_pass(sipp_ctx* ctx, const char* name, TreeArgs& a, unsigned halves) {
    const uint32_t log_e = a.k + a.lt + a.lg;
    const size_t E = (size_t)1 << log_e;
    const size_t shmem = (E + (E >> LOG_SEG) + (TWLDS ? ((size_t)1 << (a.k + a.lg)) : 0)) * sizeof(uint64_t);
    const size_t tiles = ((size_t)1 << (a.L - log_e)) * a.ncols;
    if (tiles > 0x7fffffffull) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "tree ntt: too many tiles for one launch");
    auto kern = tree_pass_kernel<FWD, TWLDS>;
    if (shmem > 64 * 1024)
        SIPP_CHECK_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    a.colfast = (uint32_t)tree_colfast();
    // full tiles of the strided sweeps: the DMA kernel (a pair of elements must be contiguous in HBM: lt >= 1)
    if (TWLDS && a.lo > 0 && log_e == 12 && a.lg == 0 && a.lt >= 1 && a.scale == 0) {
        const size_t sh = (4096 + ((size_t)1 << a.k)) * sizeof(uint64_t);
        ProfScope psd(ctx, name);
        hipLaunchKernelGGL(tree_pass_dma_kernel<FWD>, dim3((unsigned)tiles, halves), dim3(256), sh, ctx->stream, a);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        return SIPP_OK;
    }
    ProfScope ps(ctx, name);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles, halves), dim3(256), shmem, ctx->stream, a);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

constexpr uint32_t LTILE = 12;
// the coset heap table has 2^(log_n + rate_bits) words and is built on the host at the first use of a size (like every transform
// table: the one hipMalloc outside sipp_ctx_create, sipp_table_put): capped at 2^26 words = 512 MiB; beyond it the callers fall
// back to the pass-by-pass path, whose tables are O(sqrt) of the size.  The largest BASELINE config needs 2^22.
constexpr uint32_t TREE_MAX_LOG_M = 26;

// coefficients [ncols][n] natural -> the 2^rate_bits halves of the leaf-order LDE
int tree_forward(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t L, uint32_t rate_bits) {
    const size_t n = (size_t)1 << L;
    const uint64_t* tw = tree_coset_table(ctx, L + rate_bits);
    if (!tw) return SIPP_E_HIP;
    const uint32_t kc = L < LTILE ? L : LTILE;
    const std::vector<uint32_t> ks = split_bits(L - kc);
    const unsigned halves = 1u << rate_bits;
    uint32_t top = L;
    bool first = true;
    for (uint32_t k : ks) {
        TreeArgs a{};
        a.in = first ? d_coeffs : d_lde;
        a.in_stride = first ? n : (n << rate_bits);
        a.in_half = first ? 0 : n;
        a.out = d_lde; a.out_stride = n << rate_bits; a.out_half = n;
        a.L = L; a.k = k; a.lo = top - k;
        a.lt = LTILE - k < a.lo ? LTILE - k : a.lo;
        a.lg = 0; a.q0 = halves; a.tw = tw; a.scale = 0; a.ncols = (uint32_t)ncols;
        SIPP_TRY((launch_pass<true, true>(ctx, "ntt_tree_fwd", a, halves)));
        top -= k;
        first = false;
    }
    TreeArgs a{};
    a.in = first ? d_coeffs : d_lde;
    a.in_stride = first ? n : (n << rate_bits);
    a.in_half = first ? 0 : n;
    a.out = d_lde; a.out_stride = n << rate_bits; a.out_half = n;
    a.L = L; a.k = kc; a.lo = 0; a.lt = 0; a.lg = 0; a.q0 = halves; a.tw = tw; a.scale = 0; a.ncols = (uint32_t)ncols;
    if (kc == LTILE) return launch_fwd_dma(ctx, a, halves);      // shorter columns (sipp_tree_coset_eval): the register-staged kernel
    return launch_pass<true, false>(ctx, "ntt_tree_fwd", a, halves);
}

}  // namespace

// the tree sweeps take every column of 2^15 .. 2^25 rows here; ntt.hip routes 2^13 / 2^14 to them as well (shorter ones fit LDS whole: lde_column)
bool sipp_tree_ntt_enabled(uint32_t log_n) { return log_n >= 15 && log_n <= 25; }

// values -> coefficients + LDE with the fused middle sweep: gather (8 bits) | strided inverse sweeps | middle (inverse top ki bits +
// forward top kf bits of every half) | the forward tree's remaining sweeps.  Needs L >= 13 (a strided top sweep exists).
static int tree_from_values_fused(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t L,
                                  uint32_t rate_bits) {
    const size_t n = (size_t)1 << L;
    const uint64_t* twi = tree_flat_table(ctx, L, true);
    const uint64_t* twf = tree_coset_table(ctx, L + rate_bits);
    if (!twi || !twf) return SIPP_E_HIP;
    const uint32_t k2 = 8;
    std::vector<uint32_t> fks = split_bits(L - LTILE);                // forward strided sweeps, top first
    // nine levels above the contiguous sweep (2^21 rows: n = 4096) ALL inside the middle sweep -- tile 2^9 rows x 8 columns, 64-byte row
    // segments -- instead of five there and a strided sweep of four (memory-bound: 2 N read + 2 N written for four levels): all tree
    // sweeps of 2^21 x 128 10.03 -> 9.13 ms (round 5).  Ten levels (32-byte segments) are NOT worth it: the middle sweep of 2^18 x 1024
    // with ten inverse levels took 7.4 instead of 3.0 + 0.84 ms.
    if (L - LTILE == 9) fks = {9};
    const uint32_t kf = fks[0];
    // inverse levels inside the middle sweep: all that is left above the gather when they fit a tile's rows (L <= 17: three sweeps in
    // all), else as many as the forward's top sweep has
    const uint32_t ki = L - k2 <= 9 ? L - k2 : kf;                    // (nine: 2^17 rows in three sweeps, 3.90 -> 3.79 ms for 1024 columns)
    const uint32_t km = kf > ki ? kf : ki;
    const unsigned halves = 1u << rate_bits;
    {
        GatherArgs g{};
        g.in = d_values; g.out = d_coeffs; g.L = L; g.k2 = k2; g.tw = twi; g.ncols = (uint32_t)ncols;
        g.colfast = (uint32_t)tree_colfast();
        const size_t E = (size_t)16 << k2;
        const size_t shmem = E * sizeof(uint64_t);               // swizzled, no padding: five blocks per CU
        const size_t tiles = ((size_t)1 << (L - 4 - k2)) * ncols;
        if (tiles > 0x7fffffffull) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "tree ntt: too many tiles for one launch");
        ProfScope ps(ctx, "ntt_tree_gather");
        hipLaunchKernelGGL(tree_gather_kernel, dim3((unsigned)tiles), dim3(256), shmem, ctx->stream, g);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    {
        const std::vector<uint32_t> iks = split_bits(L - k2 - ki);    // inverse strided sweeps between the gather and the middle
        uint32_t lo = k2;
        for (size_t i = iks.size(); i-- > 0;) {
            TreeArgs a{};
            a.in = d_coeffs; a.out = d_coeffs; a.in_stride = a.out_stride = n;
            a.L = L; a.k = iks[i]; a.lo = lo;
            a.lt = LTILE - a.k < lo ? LTILE - a.k : lo;
            a.lg = 0; a.q0 = 0; a.tw = twi; a.ncols = (uint32_t)ncols; a.scale = 0;
            SIPP_TRY((launch_pass<false, true>(ctx, "ntt_tree_inv", a, 1)));
            lo += a.k;
        }
    }
    {
        MidArgs m{};
        m.coeffs = d_coeffs; m.lde = d_lde; m.L = L; m.kf = kf; m.ki = ki; m.lt = LTILE - km; m.rate_bits = rate_bits;
        m.tw_inv = twi; m.tw_fwd = twf; m.scale = gl::inv((uint64_t)1 << L); m.ncols = (uint32_t)ncols;
        m.colfast = (uint32_t)tree_colfast();
        const size_t E = (size_t)1 << LTILE;
        const size_t shmem = (E + ((size_t)2 << km)) * sizeof(uint64_t);       // swizzled tile + the two constant heaps
        const size_t tiles = ((size_t)1 << (L - LTILE)) * ncols;
        if (tiles > 0x7fffffffull) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "tree ntt: too many tiles for one launch");
        ProfScope ps(ctx, "ntt_tree_mid");
        hipLaunchKernelGGL(tree_mid_kernel, dim3((unsigned)tiles), dim3(256), shmem, ctx->stream, m);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
    }
    uint32_t top = L - kf;
    for (size_t i = 1; i < fks.size(); i++) {
        TreeArgs a{};
        a.in = d_lde; a.in_stride = n << rate_bits; a.in_half = n;
        a.out = d_lde; a.out_stride = n << rate_bits; a.out_half = n;
        a.L = L; a.k = fks[i]; a.lo = top - a.k;
        a.lt = LTILE - a.k < a.lo ? LTILE - a.k : a.lo;
        a.lg = 0; a.q0 = halves; a.tw = twf; a.scale = 0; a.ncols = (uint32_t)ncols;
        SIPP_TRY((launch_pass<true, true>(ctx, "ntt_tree_fwd", a, halves)));
        top -= a.k;
    }
    TreeArgs a{};
    a.in = d_lde; a.in_stride = n << rate_bits; a.in_half = n;
    a.out = d_lde; a.out_stride = n << rate_bits; a.out_half = n;
    a.L = L; a.k = LTILE; a.lo = 0; a.lt = 0; a.lg = 0; a.q0 = halves; a.tw = twf; a.scale = 0; a.ncols = (uint32_t)ncols;
    return launch_fwd_dma(ctx, a, halves);
}

int sipp_tree_lde_from_values(sipp_ctx* ctx, const uint64_t* d_values, uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols,
                              uint32_t log_n, uint32_t rate_bits) {
    if (d_values == d_coeffs || ncols == 0 || ncols > 0xffffffu || log_n + rate_bits > TREE_MAX_LOG_M) return SIPP_E_UNSUPPORTED;
    if (log_n < 13) return SIPP_E_UNSUPPORTED;     // no strided top sweep to fuse into (never asked for: sipp_tree_ntt_enabled)
    return tree_from_values_fused(ctx, d_values, d_coeffs, d_lde, ncols, log_n, rate_bits);
}

int sipp_tree_lde_from_coeffs(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_lde, size_t ncols, uint32_t log_n,
                              uint32_t rate_bits) {
    if (ncols == 0 || ncols > 0xffffffu || log_n + rate_bits > TREE_MAX_LOG_M) return SIPP_E_UNSUPPORTED;
    return tree_forward(ctx, d_coeffs, d_lde, ncols, log_n, rate_bits);
}

// SHORT polynomials on a LONG coset (the public-input polynomials of the quotient: 2^log_n coefficients, 2^log_m points of 7 <w_m>,
// leaf order): the same trees with log_m - log_n copy levels -- 2^(log_m - log_n) independent size-2^log_n transforms, ONE sweep
// that reads the coefficients from L2 and writes every point once, instead of a zero-padded 2^log_m transform in three (shorter
// polynomials would make one tiny block per subtree: they stay with the pass-by-pass path)
int sipp_tree_coset_eval(sipp_ctx* ctx, const uint64_t* d_coeffs, uint64_t* d_out, size_t ncols, uint32_t log_n, uint32_t log_m) {
    if (ncols == 0 || ncols > 0xffffffu || log_n < 10 || log_m < log_n || log_m - log_n > 15 || log_m > 25)   // <= TREE_MAX_LOG_M
        return SIPP_E_UNSUPPORTED;
    return tree_forward(ctx, d_coeffs, d_out, ncols, log_n, log_m - log_n);
}
