// sipp_amd/csrc/prover.hip -- device kernels of the STARK prover beyond commitment:
// permutation-Z columns, constraint/quotient evaluation, openings, FRI combine / divide / fold, query gathers.
//
// Replaces, on the GPU, starky::prover::prove's compute_permutation_z_polys, compute_quotient_polys,
// StarkOpeningSet::new and plonky2's PolynomialBatch::prove_openings / fri_committed_trees
// (@ InternetMaximalism/plonky2 541e127; reached from reference src/verifier_circuit.rs:133-135 only).
// All arithmetic is Goldilocks u64 / quadratic extension; nothing here is GEMM shaped.
#include <map>

#include "ctx.hpp"
#include "prover.hpp"

namespace {

using gl::E2;

// =====================================================================================================
// permutation Z:  Z[r] = prod_{r' < r} (c + g)(t + b) / ((pin + g)(ptab + b))   (g = gamma_i, b = beta_i: independent
// challenges for the column factor and the table factor, so the two multisets are permuted separately)
// Round 5: NO inversion per lane.  With num_r = (c + g)(t + b), den_r = (pin + g)(ptab + b) and T = prod over ALL rows of den,
//     Z[r] = (prod_{r' < r} num_r') (prod_{r' >= r} den_r') / T,
// an exclusive PREFIX product of the numerators times an inclusive SUFFIX product of the denominators: two scans and ONE field
// inversion per column (phase B) instead of a batch inversion in every lane (72 products per 16 rows) and its three products per row --
// 8 products per row and challenge instead of 14 (the kernel is bound by its products: 27 ms of an n = 4096 instance).
// phase A: 256 lanes x ZR rows per block: lane-local and block-local prefix / suffix products, their product per row, two block totals
// phase B: one block per Z column: exclusive prefix scan of the numerator totals, exclusive suffix scan of the denominator totals, 1 / T
// phase C: Z = (prefix before the block)(suffix after the block) / T  x  what phase A left
// =====================================================================================================
// inclusive multiplicative scan across 256 lanes (Hillis-Steele in LDS); REV: from the last lane down (suffix products)
template <bool REV>
__device__ __forceinline__ uint64_t block_scan_mul_256(uint64_t v, uint64_t* s) {
    const int t = REV ? 255 - (int)threadIdx.x : (int)threadIdx.x;
    s[t] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint64_t o = t >= off ? s[t - off] : 1;
        __syncthreads();
        v = gl::mul(v, o);
        s[t] = v;
        __syncthreads();
    }
    return v;     // s[t'] holds the inclusive product of scan positions 0 .. t'
}

// four inclusive scans in one pass of barriers: v[0], v[1] prefix products (lane 0 upwards), v[2], v[3] suffix products (lane 255
// downwards); afterwards s[q][lane] holds lane's inclusive value of scan q
__device__ __forceinline__ void block_scan4_mul_256(uint64_t (&v)[4], uint64_t (*s)[256]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; q++) s[q][t] = v[q];
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint64_t o[4];
        o[0] = t >= off ? s[0][t - off] : 1;
        o[1] = t >= off ? s[1][t - off] : 1;
        o[2] = t + off < 256 ? s[2][t + off] : 1;
        o[3] = t + off < 256 ? s[3][t + off] : 1;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v[q] = gl::mul(v[q], o[q]);
            s[q][t] = v[q];
        }
        __syncthreads();
    }
}

// one block = one checked column x 256 ZR rows, BOTH challenges from one load of the four cells of a row (the kernel is bound by its
// memory traffic: 8 + 2 words per checked cell with one block per (challenge, column), 4 + 2 here)
template <int ZR>
__global__ void __launch_bounds__(256) z_phase_a(const uint64_t* __restrict__ trace, size_t n, int nm, int nc, int cbase,
                                                uint64_t gamma0, uint64_t gamma1, uint64_t beta0, uint64_t beta1,
                                                uint64_t* __restrict__ zv, uint64_t* __restrict__ totals) {
    __shared__ uint64_t s[4][256];
    const int j = blockIdx.y;
    const size_t r0 = (size_t)blockIdx.x * (256 * ZR) + (size_t)threadIdx.x * ZR;
    // a lane owns ZR consecutive rows = ZR * 8 contiguous bytes of each column: 16-byte vector accesses
    const ulonglong2* col = reinterpret_cast<const ulonglong2*>(trace + (size_t)(cbase + j) * n + r0);
    const ulonglong2* tab = reinterpret_cast<const ulonglong2*>(trace + r0);
    const ulonglong2* pin = reinterpret_cast<const ulonglong2*>(trace + (size_t)(nm + j) * n + r0);
    const ulonglong2* ptab = reinterpret_cast<const ulonglong2*>(trace + (size_t)(nm + nc + j) * n + r0);
    uint64_t num[2][ZR], den[2][ZR];
#pragma unroll
    for (int k = 0; k < ZR; k += 2) {
        const ulonglong2 c2 = col[k >> 1], t2 = tab[k >> 1], p2 = pin[k >> 1], q2 = ptab[k >> 1];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint64_t g = i ? gamma1 : gamma0, b = i ? beta1 : beta0;
            num[i][k] = gl::mul(gl::add(c2.x, g), gl::add(t2.x, b));
            den[i][k] = gl::mul(gl::add(p2.x, g), gl::add(q2.x, b));
            num[i][k + 1] = gl::mul(gl::add(c2.y, g), gl::add(t2.y, b));
            den[i][k + 1] = gl::mul(gl::add(p2.y, g), gl::add(q2.y, b));
        }
    }
    // num[i][k] <- product of the lane's numerators BEFORE row k;  den[i][k] <- product of its denominators FROM row k on
    uint64_t v[4];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        uint64_t ntot = 1;
#pragma unroll
        for (int k = 0; k < ZR; k++) {
            const uint64_t x = num[i][k];
            num[i][k] = ntot;
            ntot = gl::mul(ntot, x);
        }
#pragma unroll
        for (int k = ZR - 2; k >= 0; k--) den[i][k] = gl::mul(den[i][k], den[i][k + 1]);
        v[i] = ntot;
        v[2 + i] = den[i][0];
    }
    // block-local: numerators of the lanes before this one, denominators of the lanes after it
    block_scan4_mul_256(v, s);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int zi = i * nc + j;
        const uint64_t nbefore = threadIdx.x ? s[i][threadIdx.x - 1] : 1, dafter = threadIdx.x < 255 ? s[2 + i][threadIdx.x + 1] : 1;
        const uint64_t ls = gl::mul(nbefore, dafter);
        ulonglong2* out = reinterpret_cast<ulonglong2*>(zv + (size_t)zi * n + r0);
#pragma unroll
        for (int k = 0; k < ZR; k += 2) {
            ulonglong2 o;
            o.x = gl::mul(ls, gl::mul(num[i][k], den[i][k]));
            o.y = gl::mul(ls, gl::mul(num[i][k + 1], den[i][k + 1]));
            out[k >> 1] = o;
        }
        // totals[zi][0][blk] = the block's numerators, totals[zi][1][blk] = its denominators
        if (threadIdx.x == 255) totals[((size_t)zi * 2) * gridDim.x + blockIdx.x] = v[i];
        if (threadIdx.x == 0) totals[((size_t)zi * 2 + 1) * gridDim.x + blockIdx.x] = v[2 + i];
    }
}

__global__ void __launch_bounds__(256) z_phase_b(uint64_t* __restrict__ totals, int nblk) {
    // per Z column (nblk <= 2^15 blocks, chunked by 256): tn[blk] <- (numerators of the blocks before blk) (denominators of the blocks
    // after it) / T, T = all denominators
    __shared__ uint64_t s[256];
    uint64_t* tn = totals + (size_t)blockIdx.x * 2 * nblk;
    uint64_t* td = tn + nblk;
    // exclusive prefix products of tn, in place
    uint64_t carry = 1;
    for (int base = 0; base < nblk; base += 256) {
        const int idx = base + threadIdx.x;
        const uint64_t v = idx < nblk ? tn[idx] : 1;
        (void)block_scan_mul_256<false>(v, s);
        const uint64_t excl = gl::mul(carry, threadIdx.x ? s[threadIdx.x - 1] : 1);
        const uint64_t last = s[255];
        __syncthreads();
        if (idx < nblk) tn[idx] = excl;
        carry = gl::mul(carry, last);
        __syncthreads();
    }
    // exclusive suffix products of td, from the last chunk down; `carry` ends as T
    carry = 1;
    for (int base = ((nblk - 1) / 256) * 256; base >= 0; base -= 256) {
        const int idx = base + threadIdx.x;
        const uint64_t v = idx < nblk ? td[idx] : 1;
        (void)block_scan_mul_256<true>(v, s);                    // s[255 - lane] = product of lanes lane .. 255
        const uint64_t excl = gl::mul(carry, threadIdx.x < 255 ? s[255 - threadIdx.x - 1] : 1);
        const uint64_t all = s[255];
        __syncthreads();
        if (idx < nblk) td[idx] = excl;
        carry = gl::mul(carry, all);
        __syncthreads();
    }
    const uint64_t tinv = gl::inv(carry);
    for (int idx = threadIdx.x; idx < nblk; idx += 256) tn[idx] = gl::mul(gl::mul(tn[idx], td[idx]), tinv);
}

template <int ZR>
__global__ void __launch_bounds__(256) z_phase_c(uint64_t* __restrict__ zv, size_t n, const uint64_t* __restrict__ totals) {
    constexpr int ZBLOCK = 256 * ZR;
    const int zi = blockIdx.y;
    ulonglong2* z = reinterpret_cast<ulonglong2*>(zv + (size_t)zi * n + (size_t)blockIdx.x * ZBLOCK);
    const uint64_t tb = totals[((size_t)zi * 2) * gridDim.x + blockIdx.x];
    for (int k = threadIdx.x; k < ZBLOCK / 2; k += 256) {
        ulonglong2 v = z[k];
        v.x = gl::mul(tb, v.x);
        v.y = gl::mul(tb, v.y);
        z[k] = v;
    }
}

// =====================================================================================================
// quotient: one lane per LDE point (leaf order).
//   quotient_prog_kernel : the AIR program (gadgets + polynomial constraints) interpreted over the base field.
//                          Limb vectors, the 31-coefficient accumulator and the quotient limbs live in REGISTERS:
//                          the limb loops are static (16 / 17 limbs, fixed by tools/air_gen.py) and fully unrolled,
//                          only the walk over terms / products is data driven (wave-uniform, no divergence).
//   quotient_rest_kernel : range table, permuted lookups and permutation-Z constraints (no program, no arrays,
//                          high occupancy, loads batched four columns at a time), then
//                          total = acc_prog * alpha^(#rest) + acc_rest, divided by Z_H.
// Constraint order == alpha-power order, identical to oracle/air_eval.inc.
// =====================================================================================================
struct QuotArgs {
    const uint64_t* lde;   // [W][m]
    const uint64_t* zlde;  // [P][m]
    const uint64_t* aux;   // [n_aux][m]
    const int64_t* prog;
    int prog_len;
    int W, nm, nc, cbase, tbits;
    uint32_t log_n, log_m;   // log_m = log_n + 1: the quotient domain (coset 7 <w_2N>) = the first 2N leaves of any larger LDE
    size_t lde_stride;       // column stride of lde / zlde (n << rate_bits)
    uint64_t alpha[2], beta[2], gamma[2];
    // program segments: one gadget, or a run of up to 64 POLY ops; each is folded with alpha from zero by one lane
    const uint32_t* seg_off;                   // word offset of the segment in prog
    const uint32_t* seg_cnt;                   // 0 = gadget, else number of POLY ops in the run
    const uint64_t* seg_pow;                   // [n_seg][2]: alpha_c^(#program constraints after the segment)
    const uint32_t* apoly3;                    // [2][64][3]: limbs (gl::limbs3) of alpha_c^e, e < 64: the fold inside a run of POLY ops
    int n_seg;
    uint64_t* part;                            // [n_seg][2][m] partial Horner sums
    const uint64_t* per_tab[AIR_N_PERIODIC];  // table k has 2 * m_k entries, indexed by natural i mod 2 m_k
    uint32_t per_mask[AIR_N_PERIODIC];
    // the AIR's value-periodic columns (periodic index AIR_N_PERIODIC + k; the pairing AIR's selectors and constants) on the
    // quotient coset: [n_vper][2 R], R = rows per record, indexed by natural i mod 2 R
    const uint64_t* vper_tab;
    uint32_t vper_mask;
    uint64_t zh_inv[2];   // 1 / (x^N - 1) for even / odd natural index
    uint64_t zh[2];       // x^N - 1
    uint64_t ninv;        // 1 / N
    uint64_t g_inv;       // g^-1 (g = generator of the trace subgroup)
    uint64_t w_m;         // 2N-th root of unity
    uint32_t p_limbs[16];
    uint64_t* out;        // [2][m] leaf order
    // quotient_rest: K constraints after the program; apow3 = their alpha-power weights (alpha_pow3_kernel layout)
    int K;
    const uint32_t* apow3;
    uint64_t alphaK[2];   // alpha_c^K
    // products of a gadget (group 2): limbs of gamma_c^i, i < 8, gamma_c = 1 / alpha_c; gamma_c; alpha_c^16 (see quotient_prog_kernel)
    uint32_t ginv3[2][8][3];
    uint64_t gamma_inv[2], alpha16[2];
    // quotient_rest over a THIN domain (the Fq12 STARK: 2^14 points are 64 blocks, each walking ~1500 checked columns): the columns are cut
    // into `chunks` ranges (blockIdx.y), every range leaves its three sums per challenge in rest_part [chunk][6][m], quotient_finish adds up
    int chunks;
    uint64_t* rest_part;
};

struct QCtx {
    // copies of the few kernel arguments the helpers need: a POINTER to the by-value argument struct would force the whole struct
    // (about 1 KB) into every lane's scratch memory
    const uint64_t* lde;
    size_t lde_stride;
    uint64_t alpha0, alpha1;
    size_t j, jn, m;
    uint64_t p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11;     // scalars, not an array: the object must stay in registers
    uint64_t acc0, acc1;
    const uint64_t* vper;       // the lane's entry of value-periodic column 0; column k sits k * vper_stride words further
    uint32_t vper_stride;
    // selects instead of a dynamically indexed private array (which would live in scratch memory); k is wave-uniform.  Indices
    // 0 .. 3: the exponentiation AIRs' selectors, 4 .. 11: MapToG2's eight row types; from AIR_N_PERIODIC on: the AIR's value-periodic
    // columns, read from their table
    __device__ __forceinline__ uint64_t periodic(int k) const {
        static_assert(AIR_N_PERIODIC == 12, "periodic(): update the select chain");
        if (k >= AIR_N_PERIODIC) return vper[(size_t)(k - AIR_N_PERIODIC) * vper_stride];
        if (k < 4) return k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
        if (k < 8) return k == 4 ? p4 : k == 5 ? p5 : k == 6 ? p6 : p7;
        return k == 8 ? p8 : k == 9 ? p9 : k == 10 ? p10 : p11;
    }
    __device__ __forceinline__ uint64_t local(int c) const { return lde[(size_t)c * lde_stride + j]; }
    __device__ __forceinline__ uint64_t next(int c) const { return lde[(size_t)c * lde_stride + jn]; }
    __device__ __forceinline__ void emit(uint64_t v) {
        acc0 = gl::mad(acc0, alpha0, v);
        acc1 = gl::mad(acc1, alpha1, v);
    }
};

// evaluates a VEC with exactly N limbs into registers; returns words consumed
template <int N>
__device__ __forceinline__ int qvec(const int64_t* w, const QCtx& c, uint64_t (&out)[N]) {
    const int nt = (int)w[1];
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = 0;
    for (int t = 0; t < nt; t++) {
        const int64_t* tm = w + 2 + 5 * t;
        const int cbase = (int)tm[1], stride = (int)tm[2], flag = (int)tm[3], neg = (int)tm[4];
        uint64_t f = gl::from_i64(tm[0]);
        if (flag >= 0) {
            uint64_t pv = c.periodic(flag);
            if (neg) pv = gl::sub(1, pv);
            f = gl::mul(f, pv);
        }
        uint64_t v[N];
#pragma unroll
        for (int i = 0; i < N; i++) v[i] = c.local(cbase + i * stride);
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = gl::mad(f, v[i], out[i]);
    }
    return 2 + 5 * nt;
}

// one lane per (LDE point, gadget): the gadget's 32/grp + 1 constraints folded with alpha from zero, then scaled by
// alpha^(number of gadget constraints that follow), so that the sum over gadgets equals the sequential Horner value.
// Two instantiations, two launches: POLY = the runs of polynomial constraints (few registers, many waves), !POLY = the gadgets (their
// limb arrays need ~150 VGPRs).  As ONE kernel the allocator gave both the gadget path's registers and more (250 VGPRs, 1.1 KB of
// scratch per lane, two waves per SIMD): 9.9 instead of 3.9 ms per n = 128 instance when the lazy fold below was added to it.
template <bool POLY>
__global__ void __launch_bounds__(64) quotient_prog_kernel(QuotArgs a, int seg0) {
    const size_t m = (size_t)1 << a.log_m;
    const size_t j = (size_t)blockIdx.x * 64 + threadIdx.x;
    const int g = seg0 + (int)blockIdx.y;
    const uint32_t i = gl::bitrev((uint32_t)j, a.log_m);  // natural LDE index
    QCtx c;
    c.lde = a.lde;
    c.lde_stride = a.lde_stride;
    c.alpha0 = a.alpha[0];
    c.alpha1 = a.alpha[1];
    c.m = m;
    c.j = j;
    c.jn = gl::bitrev((uint32_t)((i + 2) & (m - 1)), a.log_m);
    c.acc0 = c.acc1 = 0;
    c.vper = a.vper_tab + (i & a.vper_mask);
    c.vper_stride = a.vper_mask + 1;
#define SIPP_PER(k) c.p##k = a.per_tab[k][i & a.per_mask[k]]
    SIPP_PER(0); SIPP_PER(1); SIPP_PER(2); SIPP_PER(3); SIPP_PER(4); SIPP_PER(5);
    SIPP_PER(6); SIPP_PER(7); SIPP_PER(8); SIPP_PER(9); SIPP_PER(10); SIPP_PER(11);
#undef SIPP_PER
    const int64_t* w = a.prog + a.seg_off[g];
    const uint32_t n_poly = a.seg_cnt[g];
    if (POLY) {
        // A run of up to 64 polynomial constraints: acc_c = sum_k alpha_c^(n_poly - 1 - k) v_k is a linear combination with WAVE-UNIFORM
        // weights, taken lazily (six multiply-adds per constraint and challenge, one reduction per run) instead of two Horner
        // products per constraint.  Monomials: the coefficient is +-1 in all but a few per cent of them -- the product starts at the
        // first factor and is added or subtracted (a scalar branch); only other coefficients cost a product.
        gl::Acc6 f0, f1;
        f0.zero();
        f1.zero();
        // plain locals (not the QCtx object: behind a by-reference capture it ended up in scratch memory)
        const uint64_t* __restrict__ lde = a.lde;
        const uint64_t* __restrict__ aux = a.aux;
        const size_t stride = a.lde_stride, jl = c.j, jn = c.jn;
        const uint64_t p0 = c.p0, p1 = c.p1, p2 = c.p2, p3 = c.p3, p4 = c.p4, p5 = c.p5, p6 = c.p6, p7 = c.p7, p8 = c.p8, p9 = c.p9,
                       p10 = c.p10, p11 = c.p11;
        const uint64_t* __restrict__ vper = c.vper;
        const uint32_t vper_stride = c.vper_stride;
        auto periodic = [&](int k) -> uint64_t {
            static_assert(AIR_N_PERIODIC == 12, "periodic(): update the select chain");
            if (k >= AIR_N_PERIODIC) return vper[(size_t)(k - AIR_N_PERIODIC) * vper_stride];
            if (k < 4) return k == 0 ? p0 : k == 1 ? p1 : k == 2 ? p2 : p3;
            if (k < 8) return k == 4 ? p4 : k == 5 ? p5 : k == 6 ? p6 : p7;
            return k == 8 ? p8 : k == 9 ? p9 : k == 10 ? p10 : p11;
        };
        auto operand = [&](int kind, int idx) -> uint64_t {
            return kind == 0 ? lde[(size_t)idx * stride + jl] : kind == 1 ? lde[(size_t)idx * stride + jn] : kind == 2 ? aux[(size_t)idx * m + jl]
                                                                                                                  : periodic(idx);
        };
        for (uint32_t op = 0; op < n_poly; op++) {
            const int nmono = (int)w[1];
            w += 2;
            uint64_t sum = 0;
            for (int mo = 0; mo < nmono; mo++) {
                const int64_t coef = *w++;
                const int nf = (int)*w++;
                if (nf == 0) {
                    sum = gl::add(sum, gl::from_i64(coef));
                    continue;
                }
                uint64_t t = operand((int)w[0], (int)w[1]);
                w += 2;
                for (int f = 1; f < nf; f++) {
                    t = gl::mul(t, operand((int)w[0], (int)w[1]));
                    w += 2;
                }
                if (coef == 1) sum = gl::add(sum, t);          // operands are canonical (committed LDE cells, selector values), products too
                else if (coef == -1) sum = gl::sub(sum, t);
                else sum = gl::mad(gl::from_i64(coef), t, sum);
            }
            const uint32_t* __restrict__ pw = a.apoly3 + 3 * (n_poly - 1 - op);
            f0.mac((uint32_t)sum, (uint32_t)(sum >> 32), pw);
            f1.mac((uint32_t)sum, (uint32_t)(sum >> 32), pw + 192);
        }
        c.acc0 = f0.reduce();
        c.acc1 = f1.reduce();
    } else {
        const int sign_col = (int)w[1], cbase = (int)w[2], ncl = (int)w[3], lb = (int)w[4];
        const int64_t coffset = w[5];
        const int grp = (int)w[6];
        w += 7;
        const int64_t* qdesc = w;
        w += 2 + 5 * (int)w[1];
        uint64_t e[32];
#pragma unroll
        for (int k = 0; k < 32; k++) e[k] = 0;
        // The 16 x 16 limb products are never expanded into their 31 coefficients: the 16 coefficient equations of a gadget are
        // folded with powers of alpha anyway (acc = sum_m alpha^(16 - m) v_m + ...), and the product part of that fold factorises.
        // With gamma = 1 / alpha, e_k = sum_{i + j = k} a_i b_j and the pairing of limbs in base 2^32 (group 2):
        //   sum_m alpha^(16 - m) (e_2m + 2^16 e_2m+1) = alpha^16 [A_e B_e + 2^16 (A_e B_o + A_o B_e) + gamma A_o B_o],
        //   A_e = sum_i a_2i gamma^i, A_o = sum_i a_2i+1 gamma^i (likewise B): four 8-term linear forms with WAVE-UNIFORM weights
        // (lazy limb MACs) and four products per challenge instead of 256 limb products and 31 reductions.  Exact field
        // arithmetic: the value is the one the coefficient-wise Horner fold of oracle/air_eval.inc produces.
        uint64_t psum[2] = {0, 0};
        const int np = (int)*w++;
        // a gadget with many products (the pairing AIR's: ~300) is cut into slices, one lane each: slice k takes the products p = k mod
        // n_slices, slice 0 everything else of the gadget; the parts add up (the fold is linear in the products)
        const uint32_t n_slices = (n_poly >> 31) ? (n_poly >> 8) & 0xffu : 1u, slice = (n_poly >> 31) ? n_poly & 0xffu : 0u;
        for (int p = 0; p < np; p++) {
            if (n_slices > 1 && (uint32_t)p % n_slices != slice) {
                w += 1;
                w += 2 + 5 * (int)w[1];
                w += 2 + 5 * (int)w[1];
                continue;
            }
            const uint64_t coef = gl::from_i64(*w++);
            uint64_t va[16], vb[16];
            w += qvec<16>(w, c, va);
            w += qvec<16>(w, c, vb);
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {
                gl::Acc6 Ae, Ao, Be, Bo;
                Ae.zero(); Ao.zero(); Be.zero(); Bo.zero();
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t* g3 = a.ginv3[ch][i];
                    Ae.mac((uint32_t)va[2 * i], (uint32_t)(va[2 * i] >> 32), g3);
                    Ao.mac((uint32_t)va[2 * i + 1], (uint32_t)(va[2 * i + 1] >> 32), g3);
                    Be.mac((uint32_t)vb[2 * i], (uint32_t)(vb[2 * i] >> 32), g3);
                    Bo.mac((uint32_t)vb[2 * i + 1], (uint32_t)(vb[2 * i + 1] >> 32), g3);
                }
                const uint64_t ae = Ae.reduce(), ao = Ao.reduce(), be = Be.reduce(), bo = Bo.reduce();
                const uint64_t cross = gl::add(gl::mul(ae, bo), gl::mul(ao, be));
                uint64_t t = gl::mad(gl::mul(ao, bo), a.gamma_inv[ch], gl::mul(ae, be));
                t = gl::mad(cross, 65536, t);
                psum[ch] = gl::mad(coef, t, psum[ch]);
            }
        }
        const int nl = slice == 0 ? (int)*w++ : 0;
        for (int p = 0; p < nl; p++) {
            const uint64_t coef = gl::from_i64(*w++);
            uint64_t va[16];
            w += qvec<16>(w, c, va);
#pragma unroll
            for (int ii = 0; ii < 16; ii++) e[ii] = gl::mad(coef, va[ii], e[ii]);
        }
        if (slice == 0) {
        uint64_t q[17];
        (void)qvec<17>(qdesc, c, q);
        const uint64_t s = c.local(sign_col);
        const uint64_t sgn = gl::sub(1, gl::add(s, s));
        uint64_t cprev = 0;
        const uint64_t coff = gl::from_i64(coffset);
        uint64_t d[32];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            // q * p limbs: the p limbs are 16-bit constants, so the two 32-bit halves of q accumulate in plain u64
            uint64_t al = 0, ah = 0;
#pragma unroll
            for (int ii = 0; ii < 17; ii++) {
                const int jj = k - ii;
                if (jj >= 0 && jj < 16) {
                    al += (uint64_t)(uint32_t)q[ii] * a.p_limbs[jj];
                    ah += (uint64_t)(uint32_t)(q[ii] >> 32) * a.p_limbs[jj];
                }
            }
            const uint64_t l = al + (ah << 32);
            const uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
            const uint64_t qp = gl::reduce96(h, l);
            d[k] = gl::sub(e[k], gl::mul(sgn, qp));
        }
        const uint64_t wgt = (uint64_t)1 << (16 * grp);
#pragma unroll
        for (int mm = 0; mm < 32; mm++) {
            if (mm < 32 / grp) {
                uint64_t v;
                if (grp == 2) v = gl::mad(d[(2 * mm + 1) & 31], 65536, d[(2 * mm) & 31]);
                else v = d[mm];
                uint64_t ck = 0;
                if (mm < 32 / grp - 1) {
                    for (int l = 0; l < ncl; l++)
                        ck = gl::mad(c.local(cbase + mm * ncl + l), (uint64_t)1 << (lb * l), ck);
                    ck = gl::sub(ck, coff);
                }
                v = gl::sub(v, cprev);
                v = gl::mad(ck, wgt, v);
                c.emit(v);
                cprev = ck;
            }
        }
        c.emit(gl::mul(s, gl::sub(s, 1)));
        }
        // the product part of the fold (grp == 2: the generator's only setting, checked on the host)
        c.acc0 = gl::mad(a.alpha16[0], psum[0], c.acc0);
        c.acc1 = gl::mad(a.alpha16[1], psum[1], c.acc1);
    }
    a.part[((size_t)g * 2 + 0) * m + j] = gl::mul(c.acc0, a.seg_pow[2 * g]);
    a.part[((size_t)g * 2 + 1) * m + j] = gl::mul(c.acc1, a.seg_pow[2 * g + 1]);
}

// limbs (gl::limbs3) of the alpha powers the constraints after the program are weighted with, laid out in the order
// quotient_rest_kernel consumes them so that one checked column is ONE contiguous 36-dword scalar fetch:
//   head  [c][t][3]        t = 0..2  : range-table constraints, exponent K-1-t
//   body  [k][c][slot][3]  slot 0..5 : lookup (first row, transition), permutation challenge 0 (first row, running
//                                      product), challenge 1 (same) of checked column k
__global__ void alpha_pow3_kernel(uint64_t a0, uint64_t a1, int K, int nc, uint32_t* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 2 * 3 + nc * 2 * 6;
    if (t >= total) return;
    int c, e;
    if (t < 6) {
        c = t / 3;
        e = K - 1 - t % 3;
    } else {
        const int k = (t - 6) / 12, r = (t - 6) % 12, slot = r % 6;
        c = r / 6;
        e = K - 1 - (3 + 2 * k) - (slot >> 1) * 2 * nc - (slot & 1);
    }
    uint32_t l[3];
    gl::limbs3(l, gl::pow(c ? a1 : a0, (uint64_t)e));
    for (int q = 0; q < 3; q++) out[3 * t + q] = l[q];
}

// The K constraints after the program are a random linear combination with WAVE-UNIFORM coefficients:
//   acc_c = acc_prog * alpha_c^K + sum_i v_i * alpha_c^(K-1-i).
// The sum is taken lazily (gl::Acc6, six v_mad_u64_u32 per term and challenge, limbs of alpha^e from scalar loads)
// in three groups that share a multiplier -- first-row constraints (x lagrange_first), transition constraints that
// vanish on the last row (x z_last), and the rest -- so the per-constraint multiplications by lagrange_first / z_last
// and the two Horner mulmods per constraint disappear.
struct RestAcc {
    gl::Acc6 a[2];
    int terms;
    uint64_t folded[2];
    __device__ __forceinline__ void init() {
        a[0].zero();
        a[1].zero();
        terms = 0;
        folded[0] = folded[1] = 0;
    }
    // p0 / p1: limbs of the constraint's weight for challenge 0 / 1
    __device__ __forceinline__ void mac(uint64_t v, const uint32_t* __restrict__ p0, const uint32_t* __restrict__ p1) {
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        a[0].mac(lo, hi, p0);
        a[1].mac(lo, hi, p1);
    }
    // an accumulator holds 1024 terms: fold long before that (wave-uniform condition)
    __device__ __forceinline__ void maybe_fold(int added) {
        terms += added;
        if (terms >= 960) {
            for (int c = 0; c < 2; c++) {
                folded[c] = gl::add(folded[c], gl::canon(a[c].reduce()));
                a[c].zero();
            }
            terms = 0;
        }
    }
    __device__ __forceinline__ uint64_t value(int c) const { return gl::add(folded[c], gl::canon(a[c].reduce())); }
};

// the point's multipliers: lagrange_first, z_last, lagrange_last, 1 / Z_H
struct RestPoint {
    uint64_t lf, zl, ll, zh_inv;
};
__device__ __forceinline__ RestPoint rest_point(const QuotArgs& a, uint32_t i) {
    const uint64_t x = gl::mul(gl::GEN, gl::pow(a.w_m, i));
    const uint64_t zh = a.zh[i & 1];
    RestPoint r;
    r.zl = gl::sub(x, a.g_inv);
    r.lf = gl::mul(gl::mul(zh, a.ninv), gl::inv(gl::sub(x, 1)));
    r.ll = gl::mul(gl::mul(gl::mul(zh, a.ninv), a.g_inv), gl::inv(r.zl));
    r.zh_inv = a.zh_inv[i & 1];
    return r;
}
// program partials + the three weighted sums -> the quotient's value at the point
__device__ __forceinline__ void rest_store(const QuotArgs& a, size_t m, size_t j, const RestPoint& pt, const uint64_t F[2], const uint64_t T[2],
                                           const uint64_t R[2]) {
    uint64_t acc[2] = {0, 0};
    for (int g = 0; g < a.n_seg; g++) {
        acc[0] = gl::add(acc[0], a.part[((size_t)g * 2 + 0) * m + j]);
        acc[1] = gl::add(acc[1], a.part[((size_t)g * 2 + 1) * m + j]);
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint64_t v = gl::mul(acc[q], a.alphaK[q]);
        v = gl::add(v, R[q]);
        v = gl::mad(pt.lf, F[q], v);
        v = gl::mad(pt.zl, T[q], v);
        a.out[(size_t)q * m + j] = gl::mul(v, pt.zh_inv);
    }
}

template <bool SPLIT>
__global__ void __launch_bounds__(256) quotient_rest_kernel(QuotArgs a) {
    const size_t m = (size_t)1 << a.log_m;
    // Which 256 points a block takes (round 5).  The next row of the point with natural index i is i + 2, at leaf position bitrev(i + 2):
    // far from bitrev(i), so a block of 256 CONSECUTIVE leaf positions read every `next` cell from lines no lane of it reads as `local` --
    // nine words per checked column and point, four of them a second fetch of data another block fetches as its own (the kernel runs at
    // the memory's pace: 186 GB for an n = 4096 instance).  A block takes eight runs of 32 leaf positions instead, the runs being the
    // eight values of bits 1 .. 3 of i: seven of the eight `next` runs are then `local` runs of the same block, fetched once.
    constexpr uint32_t RB = 3, RW = 256u >> RB;
    const uint32_t tid = threadIdx.x, run = tid / RW, blk = blockIdx.x, half = blk / (uint32_t)(m >> 9);
    const size_t j = (size_t)half * (m >> 1) + (size_t)run * (m >> (RB + 1)) + (size_t)(blk % (uint32_t)(m >> 9)) * RW + tid % RW;
    const uint32_t i = gl::bitrev((uint32_t)j, a.log_m);
    QCtx c;
    c.lde = a.lde;
    c.lde_stride = a.lde_stride;
    c.alpha0 = a.alpha[0];
    c.alpha1 = a.alpha[1];
    c.m = m;
    c.j = j;
    c.jn = gl::bitrev((uint32_t)((i + 2) & (m - 1)), a.log_m);
    c.acc0 = c.acc1 = 0;
    const uint32_t* __restrict__ ap = a.apow3;
    RestAcc F, T, R;  // x lagrange_first, x z_last, plain
    F.init();
    T.init();
    R.init();
    // range table (the first chunk's)
    const uint64_t tl = c.local(0);
    RestPoint pt{};
    if (!SPLIT || blockIdx.y == 0) {
        pt = rest_point(a, i);
        const uint64_t tn = c.next(0), ll = pt.ll;
        const uint64_t d = gl::sub(tn, tl);
        F.mac(tl, ap, ap + 9);
        T.mac(gl::mul(d, gl::sub(d, 1)), ap + 3, ap + 12);
        R.mac(gl::mul(ll, gl::sub(tl, ((uint64_t)1 << a.tbits) - 1)), ap + 6, ap + 15);
    }
    // One pass over the checked columns evaluates the lookup constraints AND the permutation constraints of both
    // challenges (nine loads per column instead of fourteen); the exponent of each constraint follows from its
    // position in the oracle's order: lookups 3 + 2k (+1), then per challenge 3 + 2 nc + 2 nc ch + 2k (+1).
    const int nm = a.nm, nc = a.nc;
    const uint64_t g0 = a.gamma[0], g1 = a.gamma[1];
    const uint64_t b0 = a.beta[0], b1 = a.beta[1];
    const uint64_t tg0 = gl::add(tl, b0), tg1 = gl::add(tl, b1);
    constexpr int U = 2;
    // this block's range of checked columns: all of them, or the blockIdx.y-th of a.chunks ranges (multiples of U)
    const int per = SPLIT ? ((nc + a.chunks - 1) / a.chunks + U - 1) / U * U : nc;
    const int kbeg = SPLIT ? (int)blockIdx.y * per : 0, kend = SPLIT ? min(nc, kbeg + per) : nc;
    for (int k0 = kbeg; k0 < kend; k0 += U) {
        uint64_t pin[U], ptab[U], npin[U], nptab[U], col[U], z0[U], zn0[U], z1[U], zn1[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = min(k0 + u, nc - 1);
            pin[u] = c.local(nm + k);
            ptab[u] = c.local(nm + nc + k);
            npin[u] = c.next(nm + k);
            nptab[u] = c.next(nm + nc + k);
            col[u] = c.local(a.cbase + k);
            z0[u] = a.zlde[(size_t)k * a.lde_stride + j];
            zn0[u] = a.zlde[(size_t)k * a.lde_stride + c.jn];
            z1[u] = a.zlde[(size_t)(nc + k) * a.lde_stride + j];
            zn1[u] = a.zlde[(size_t)(nc + k) * a.lde_stride + c.jn];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = k0 + u;
            if (k < kend) {
                const uint32_t* __restrict__ w0 = ap + 18 + 36 * k;  // [c][slot][3]
                const uint32_t* __restrict__ w1 = w0 + 18;
                F.mac(gl::sub(pin[u], ptab[u]), w0, w1);
                T.mac(gl::mul_nc(gl::sub(npin[u], pin[u]), gl::sub(npin[u], nptab[u])), w0 + 3, w1 + 3);
                const uint64_t cg0 = gl::add(col[u], g0), cg1 = gl::add(col[u], g1);
                const uint64_t rhs0 = gl::mul_nc(gl::add(pin[u], g0), gl::add(ptab[u], b0));
                const uint64_t rhs1 = gl::mul_nc(gl::add(pin[u], g1), gl::add(ptab[u], b1));
                F.mac(gl::sub(z0[u], 1), w0 + 6, w1 + 6);
                R.mac(gl::sub(gl::mul(zn0[u], rhs0), gl::mul(z0[u], gl::mul_nc(cg0, tg0))), w0 + 9, w1 + 9);
                F.mac(gl::sub(z1[u], 1), w0 + 12, w1 + 12);
                R.mac(gl::sub(gl::mul(zn1[u], rhs1), gl::mul(z1[u], gl::mul_nc(cg1, tg1))), w0 + 15, w1 + 15);
            }
        }
        F.maybe_fold(3 * U);
        T.maybe_fold(U);
        R.maybe_fold(2 * U);
    }
    const uint64_t f[2] = {F.value(0), F.value(1)}, t[2] = {T.value(0), T.value(1)}, r[2] = {R.value(0), R.value(1)};
    if (SPLIT) {
        uint64_t* __restrict__ o = a.rest_part + (size_t)blockIdx.y * 6 * m + j;
        o[0] = f[0]; o[m] = f[1]; o[2 * m] = t[0]; o[3 * m] = t[1]; o[4 * m] = r[0]; o[5 * m] = r[1];
        return;
    }
    rest_store(a, m, j, pt, f, t, r);
}

// the thin form's second step: the chunks' sums added up (canonical field elements: the order does not matter)
__global__ void __launch_bounds__(256) quotient_finish_kernel(QuotArgs a) {
    const size_t m = (size_t)1 << a.log_m;
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t i = gl::bitrev((uint32_t)j, a.log_m);
    uint64_t v[6] = {0, 0, 0, 0, 0, 0};
    for (int ch = 0; ch < a.chunks; ch++)
#pragma unroll
        for (int q = 0; q < 6; q++) v[q] = gl::add(v[q], a.rest_part[((size_t)ch * 6 + q) * m + j]);
    const uint64_t f[2] = {v[0], v[1]}, t[2] = {v[2], v[3]}, r[2] = {v[4], v[5]};
    rest_store(a, m, j, rest_point(a, i), f, t, r);
}

// =====================================================================================================
// power tables, openings, FRI combine / divide / fold
// =====================================================================================================
// tab[k] = base^k (ext), SoA: c0 at [0, n), c1 at [n, 2n)
__global__ void pow_table_kernel(E2 base, size_t n, uint64_t* __restrict__ tab) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    E2 r = gl::pow(base, (uint64_t)k);
    tab[k] = r.c0;
    tab[n + k] = r.c1;
}

// the four tables of an opening (zeta, g zeta and their inverses) in ONE launch: blockIdx.y selects the base
struct Pow4Args {
    E2 base[4];
    uint64_t* tab[4];
    size_t n;
};
__global__ void pow_table4_kernel(Pow4Args a) {
    const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const int b = blockIdx.y;
    // select instead of a dynamically indexed kernel argument (which would be copied to scratch memory)
    const E2 base = b == 0 ? a.base[0] : b == 1 ? a.base[1] : b == 2 ? a.base[2] : a.base[3];
    uint64_t* tab = b == 0 ? a.tab[0] : b == 1 ? a.tab[1] : b == 2 ? a.tab[2] : a.tab[3];
    const E2 r = gl::pow(base, (uint64_t)k);
    tab[k] = r.c0;
    tab[a.n + k] = r.c1;
}

// one block per OCOLS columns: sum_k c[k] * t0[k] and sum_k c[k] * t1[k]; out[col] = (e0.c0, e0.c1, e1.c0, e1.c1).
// The power tables are read once per block and k, i.e. once per OCOLS coefficients; the products are summed lazily in
// 160 bits (gl::Acc160) and reduced once per lane.
// (more columns per block would re-use the tables but leave too few waves in flight: measured)
// (tn = the distance between the two components of a power table: the tables hold [c0[n] | c1[n]]; a block that sums a SEGMENT of
// the rows -- openings_seg_kernel -- passes pointers to the segment's first row and the whole table's n)
__device__ __forceinline__ void openings_column(const uint64_t* __restrict__ col, size_t n, const uint64_t* __restrict__ t0,
                                                const uint64_t* __restrict__ t1, size_t tn, uint64_t* __restrict__ out4) {
    __shared__ uint64_t s[4][256];
    gl::Acc160 acc[4];
    // OB coefficient indices per trip: all their loads are issued before the first product (the rolled loop waited for five
    // loads per 32 products; n / 256 trips of load latency were the kernel's run time)
    constexpr int OB = 4;
    size_t k = threadIdx.x;
    for (; k + (OB - 1) * 256 < n; k += OB * 256) {
        uint64_t p0[OB], p1[OB], q0[OB], q1[OB], v[OB];
#pragma unroll
        for (int b = 0; b < OB; b++) {
            const size_t kb = k + (size_t)b * 256;
            p0[b] = t0[kb];
            p1[b] = t0[tn + kb];
            q0[b] = t1 ? t1[kb] : 0;
            q1[b] = t1 ? t1[tn + kb] : 0;
            v[b] = col[kb];
        }
#pragma unroll
        for (int b = 0; b < OB; b++) {
            acc[0].mac(v[b], p0[b]);
            acc[1].mac(v[b], p1[b]);
            if (t1) {
                acc[2].mac(v[b], q0[b]);
                acc[3].mac(v[b], q1[b]);
            }
        }
    }
    for (; k < n; k += 256) {
        const uint64_t v = col[k];
        acc[0].mac(v, t0[k]);
        acc[1].mac(v, t0[tn + k]);
        if (t1) {
            acc[2].mac(v, t1[k]);
            acc[3].mac(v, t1[tn + k]);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) s[q][threadIdx.x] = gl::canon(acc[q].reduce());
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int q = 0; q < 4; q++) s[q][threadIdx.x] = gl::add(s[q][threadIdx.x], s[q][threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x < 4) out4[threadIdx.x] = s[threadIdx.x][0];
}

__global__ void __launch_bounds__(256) openings_kernel(const uint64_t* __restrict__ coeffs, size_t n, uint32_t ncols,
                                                      const uint64_t* __restrict__ t0, const uint64_t* __restrict__ t1,
                                                      uint64_t* __restrict__ out) {
    openings_column(coeffs + (size_t)blockIdx.x * n, n, t0, t1, n, out + (size_t)blockIdx.x * 4);
}
// Launches of FEW columns (the outer prover's oracles: 84 + 136 + 20 + 16 polynomials of 2^18 .. 2^20 coefficients, one launch per
// range of a batch) leave most of the 256 CUs idle with one block per column: 2.9 ms of a 37.7-ms proof at 2^18 rows.  Block (column,
// segment) sums gridDim.y-th of the rows into partial[column][segment][4]; openings_fold_kernel adds the segments (field sums: the
// result does not depend on the split).
__global__ void __launch_bounds__(256) openings_seg_kernel(const uint64_t* __restrict__ coeffs, size_t n, size_t seg_len,
                                                          const uint64_t* __restrict__ t0, const uint64_t* __restrict__ t1,
                                                          uint64_t* __restrict__ partial) {
    const size_t lo = (size_t)blockIdx.y * seg_len, len = min(seg_len, n - lo);
    openings_column(coeffs + (size_t)blockIdx.x * n + lo, len, t0 + lo, t1 ? t1 + lo : nullptr, n,
                    partial + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 4);
}
__global__ void __launch_bounds__(256) openings_fold_kernel(const uint64_t* __restrict__ partial, uint32_t ncols, uint32_t segs,
                                                           uint64_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;     // (column, component)
    if (i >= 4 * ncols) return;
    const uint64_t* p = partial + (size_t)(i >> 2) * segs * 4 + (i & 3);
    uint64_t acc = 0;
    for (uint32_t sg = 0; sg < segs; sg++) acc = gl::add(acc, p[4 * sg]);
    out[i] = acc;
}

// the three oracles of a STARK (trace, Z, quotient chunks) in ONE launch: block b opens column b of their concatenation; the
// quotient chunks are opened at the first point only
struct Open3Args {
    const uint64_t* coeffs[3];
    uint32_t ncols[3];
    size_t n;
    const uint64_t *t0, *t1;
    uint64_t* out;
};
__global__ void __launch_bounds__(256) openings3_kernel(Open3Args a) {
    const uint32_t b = blockIdx.x;
    const uint64_t* col;
    const uint64_t* t1 = a.t1;
    if (b < a.ncols[0]) {
        col = a.coeffs[0] + (size_t)b * a.n;
    } else if (b < a.ncols[0] + a.ncols[1]) {
        col = a.coeffs[1] + (size_t)(b - a.ncols[0]) * a.n;
    } else {
        col = a.coeffs[2] + (size_t)(b - a.ncols[0] - a.ncols[1]) * a.n;
        t1 = nullptr;
    }
    openings_column(col, a.n, a.t0, t1, a.n, a.out + (size_t)b * 4);
}

// The same three oracles with the power tables shared by a GROUP of columns (round 5).  With one block per column every block streams
// its column AND the four power tables (two points x two components): five words from memory per coefficient, four of them tables --
// at 2^21 rows the tables (64 MB) fit no cache and openings3_kernel ran at 0.4 of the bandwidth its columns alone need (29.8 ms per
// n = 4096 instance against 11.7 ms for 58 GB at 5 TB/s).  Block (group of OG columns of ONE oracle, segment of the rows): a lane loads
// the four table words of a row once and multiplies them into OG x 4 accumulators; partial[column][segment][4] is folded by
// openings_fold_kernel.  The quotient chunks (oracle 2) are opened at the first point only.
// (groups of 2 / 3 / 6 / 8 columns and 1 / 4 rows in flight per lane measured: 4 x 2 is the fastest at n = 128, 1024 and 4096 --
//  0.75 / 4.4 / 18.3 ms of openings per instance against 1.08 / 7.5 / 29.8 with one block per column; 8 columns: 1.3 / 5.4 / 19.7)
constexpr int OG = 4;
struct OpenGArgs {
    const uint64_t* coeffs[3];
    uint32_t ncols[3];
    size_t n, seg_len;
    const uint64_t *t0, *t1;
    uint64_t* partial;     // [total columns][segments][4]
};
__global__ void __launch_bounds__(256) openings_group_kernel(OpenGArgs a) {
    __shared__ uint64_t s[4][256];
    // which oracle, which columns
    uint32_t g = blockIdx.x, orc = 0, col_base = 0;
    while (orc < 2 && g >= (a.ncols[orc] + OG - 1) / OG) {
        g -= (a.ncols[orc] + OG - 1) / OG;
        col_base += a.ncols[orc];
        orc++;
    }
    const uint32_t c0 = g * OG, cnt = min((uint32_t)OG, a.ncols[orc] - c0);
    const bool two = orc < 2;
    const size_t n = a.n, lo = (size_t)blockIdx.y * a.seg_len, hi = min(n, lo + a.seg_len);
    const uint64_t* col = a.coeffs[orc] + (size_t)c0 * n;
    gl::Acc160 acc[OG][4];
    constexpr int OB = 2;
    size_t k = lo + threadIdx.x;
    for (; k + (OB - 1) * 256 < hi; k += OB * 256) {
        uint64_t p0[OB], p1[OB], q0[OB], q1[OB], v[OB][OG];
#pragma unroll
        for (int b = 0; b < OB; b++) {
            const size_t kb = k + (size_t)b * 256;
            p0[b] = a.t0[kb];
            p1[b] = a.t0[n + kb];
            q0[b] = two ? a.t1[kb] : 0;
            q1[b] = two ? a.t1[n + kb] : 0;
#pragma unroll
            for (int c = 0; c < OG; c++) v[b][c] = (uint32_t)c < cnt ? col[(size_t)c * n + kb] : 0;
        }
#pragma unroll
        for (int b = 0; b < OB; b++)
#pragma unroll
            for (int c = 0; c < OG; c++) {
                if ((uint32_t)c >= cnt) break;
                acc[c][0].mac(v[b][c], p0[b]);
                acc[c][1].mac(v[b][c], p1[b]);
                if (two) {
                    acc[c][2].mac(v[b][c], q0[b]);
                    acc[c][3].mac(v[b][c], q1[b]);
                }
            }
    }
    for (; k < hi; k += 256) {
        const uint64_t p0 = a.t0[k], p1 = a.t0[n + k], q0 = two ? a.t1[k] : 0, q1 = two ? a.t1[n + k] : 0;
#pragma unroll
        for (int c = 0; c < OG; c++) {
            if ((uint32_t)c >= cnt) break;
            const uint64_t v = col[(size_t)c * n + k];
            acc[c][0].mac(v, p0);
            acc[c][1].mac(v, p1);
            if (two) {
                acc[c][2].mac(v, q0);
                acc[c][3].mac(v, q1);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < OG; c++) {
        if ((uint32_t)c >= cnt) break;
#pragma unroll
        for (int q = 0; q < 4; q++) s[q][threadIdx.x] = gl::canon(acc[c][q].reduce());
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off)
                for (int q = 0; q < 4; q++) s[q][threadIdx.x] = gl::add(s[q][threadIdx.x], s[q][threadIdx.x + off]);
            __syncthreads();
        }
        if (threadIdx.x < 4) a.partial[((size_t)(col_base + c0 + c) * gridDim.y + blockIdx.y) * 4 + threadIdx.x] = s[threadIdx.x][0];
        __syncthreads();
    }
}

// partial[slice][{0,1}][{c0,c1}][k]: acc over the columns of this slice of alpha^c * coef_c[k];
// batch 1 (trace | Z) is the same sum restricted to c < n1.
struct CombArgs {
    const uint64_t* src[3];
    int cnt[3];
    size_t n;
    const uint32_t* apow3; // [total][2][3]: limbs (gl::limbs3) of alpha^c as (c0, c1)
    int n1;                // columns in batch 1 (W + P)
    int slices;
    uint64_t* partial;     // [slices][4][n]
};
// the weights alpha^c are wave-uniform (lanes run over the coefficient index k): lazy multiply-accumulate with the limbs
// of alpha^c (gl::Acc6, six v_mad_u64_u32 per extension component), one reduction per slice.  A column belongs either
// to batch 1 (c < n1) or not, so g (batch 1) and h (the rest) are accumulated separately and f = g + h.
__global__ void __launch_bounds__(256) fri_combine_kernel(CombArgs a) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int total = a.cnt[0] + a.cnt[1] + a.cnt[2];
    const int per = (total + a.slices - 1) / a.slices;   // <= 1024 terms per accumulator (host checks)
    const int c_lo = blockIdx.y * per, c_hi = min(total, c_lo + per);
    gl::Acc6 G0, G1, H0, H1;
    G0.zero(); G1.zero(); H0.zero(); H1.zero();
    const uint64_t *src0 = a.src[0], *src1 = a.src[1], *src2 = a.src[2];
    const int cnt0 = a.cnt[0], cnt01 = a.cnt[0] + a.cnt[1], n1 = a.n1;
    const size_t nn = a.n;
    const uint32_t* __restrict__ apow3 = a.apow3;
    auto col_of = [=](int c) -> const uint64_t* {
        return c < cnt0 ? src0 + (size_t)c * nn : c < cnt01 ? src1 + (size_t)(c - cnt0) * nn : src2 + (size_t)(c - cnt01) * nn;
    };
#define SIPP_COMB_TERM(cc, vv)                                             \
    do {                                                                   \
        const uint64_t v_ = (vv);                                          \
        const uint32_t lo_ = (uint32_t)v_, hi_ = (uint32_t)(v_ >> 32);     \
        const uint32_t* __restrict__ w_ = apow3 + 6 * (cc);                \
        if ((cc) < n1) { /* wave-uniform */                                \
            G0.mac(lo_, hi_, w_);                                          \
            G1.mac(lo_, hi_, w_ + 3);                                      \
        } else {                                                           \
            H0.mac(lo_, hi_, w_);                                          \
            H1.mac(lo_, hi_, w_ + 3);                                      \
        }                                                                  \
    } while (0)
    // eight columns per trip: their coefficients are loaded before the first multiply-add (one load per trip of a rolled
    // loop made the kernel wait out a memory latency per column)
    constexpr int CB = 8;
    int c = c_lo;
    for (; c + CB <= c_hi; c += CB) {
        uint64_t v[CB];
#pragma unroll
        for (int b = 0; b < CB; b++) v[b] = col_of(c + b)[k];
#pragma unroll
        for (int b = 0; b < CB; b++) SIPP_COMB_TERM(c + b, v[b]);
    }
    for (; c < c_hi; c++) SIPP_COMB_TERM(c, col_of(c)[k]);
#undef SIPP_COMB_TERM
    const uint64_t g0 = gl::canon(G0.reduce()), g1 = gl::canon(G1.reduce());
    uint64_t* p = a.partial + (size_t)blockIdx.y * 4 * a.n;
    p[k] = gl::add(g0, gl::canon(H0.reduce()));
    p[a.n + k] = gl::add(g1, gl::canon(H1.reduce()));
    p[2 * a.n + k] = g0;
    p[3 * a.n + k] = g1;
}

// generic batch composition F[k] = sum_j alpha^j col_j[k] for an arbitrary list of coefficient columns (pointer table):
// the same lazy accumulation, written in the [slices][4][n] layout the division kernels read (batch slot 0)
struct GCombArgs {
    const uint64_t* const* cols;
    int total;
    size_t n;
    const uint32_t* apow3;  // [total][2][3]
    int slices;
    uint64_t* partial;      // [slices][4][n]
};
__global__ void __launch_bounds__(256) fri_gcombine_kernel(GCombArgs a) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int per = (a.total + a.slices - 1) / a.slices;
    const int c_lo = blockIdx.y * per, c_hi = min(a.total, c_lo + per);
    gl::Acc6 G0, G1;
    G0.zero(); G1.zero();
    for (int c = c_lo; c < c_hi; c++) {
        const uint64_t v = a.cols[c][k];
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        const uint32_t* __restrict__ w = a.apow3 + 6 * c;
        G0.mac(lo, hi, w);
        G1.mac(lo, hi, w + 3);
    }
    uint64_t* p = a.partial + (size_t)blockIdx.y * 4 * a.n;
    p[k] = gl::canon(G0.reduce());
    p[a.n + k] = gl::canon(G1.reduce());
}

// acc = acc * shift + q   (ext, SoA [2][n])
__global__ void fri_accum_kernel(uint64_t* __restrict__ acc, const uint64_t* __restrict__ q, size_t n, E2 shift, int first) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    E2 r{q[k], q[n + k]};
    if (!first) r = gl::add(gl::mul(E2{acc[k], acc[n + k]}, shift), r);
    acc[k] = r.c0;
    acc[n + k] = r.c1;
}
// fin[0] = 0, fin[k + 1] = acc[k]   (multiplication by X; acc[n - 1] is zero by construction)
__global__ void fri_mulx_kernel(const uint64_t* __restrict__ acc, size_t n, uint64_t* __restrict__ fin) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    fin[k] = k ? acc[k - 1] : 0;
    fin[n + k] = k ? acc[n + k - 1] : 0;
}

// one block per batch b (0: point zeta, 1: point g zeta):
//   F = sum over slices of partial;  S_k = sum_{j >= k} F_j z^j ;  q_{k-1} = S_k z^-k
// zp = z^k table, zip = z^-k table (ext SoA).  qout: [2][n] (c0 | c1) per batch.
struct DivArgs {
    const uint64_t* partial;
    int slices;
    size_t n;
    const uint64_t* zp[2];
    const uint64_t* zip[2];
    uint64_t* qout;  // [2 batches][2][n]
};
// three launches so that the division scales with n:
//   fri_divide_tiles   one block per (1024-coefficient tile, batch): v_k = F_k z^k, inclusive suffix scan inside the
//                      tile (LDS), tile total
//   fri_divide_carry   one block per batch: exclusive suffix scan of the tile totals
//   fri_divide_finish  q_{k-1} = (tile scan + carry) z^-k
__global__ void __launch_bounds__(1024) fri_divide_tiles(DivArgs a, uint64_t* __restrict__ scan, uint64_t* __restrict__ totals) {
    __shared__ uint64_t s0[1024], s1[1024];
    const int b = blockIdx.y, t = threadIdx.x;
    const size_t n = a.n, k = (size_t)blockIdx.x * 1024 + t;
    const uint64_t* zp = a.zp[b];
    uint64_t c0 = 0, c1 = 0;
    for (int sl = 0; sl < a.slices; sl++) {
        const uint64_t* p = a.partial + (size_t)sl * 4 * n + (size_t)b * 2 * n;
        c0 = gl::add(c0, p[k]);
        c1 = gl::add(c1, p[n + k]);
    }
    E2 v = gl::mul(E2{c0, c1}, E2{zp[k], zp[n + k]});
    s0[t] = v.c0;
    s1[t] = v.c1;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        E2 o{0, 0};
        if (t + off < 1024) o = E2{s0[t + off], s1[t + off]};
        __syncthreads();
        v = gl::add(v, o);
        s0[t] = v.c0;
        s1[t] = v.c1;
        __syncthreads();
    }
    uint64_t* sc = scan + (size_t)b * 2 * n;
    sc[k] = v.c0;
    sc[n + k] = v.c1;
    if (t == 0) {
        totals[((size_t)b * gridDim.x + blockIdx.x) * 2] = v.c0;
        totals[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1] = v.c1;
    }
}

__global__ void __launch_bounds__(256) fri_divide_carry(uint64_t* __restrict__ totals, int ntiles) {
    // exclusive suffix sums over the tile totals of one batch (ntiles <= 2^14): serial chunks of a block-wide scan
    __shared__ uint64_t s0[256], s1[256];
    uint64_t* tt = totals + (size_t)blockIdx.x * ntiles * 2;
    const int t = threadIdx.x;
    E2 carry{0, 0};
    for (int base = ((ntiles + 255) / 256 - 1) * 256; base >= 0; base -= 256) {
        const int idx = base + t;
        E2 v{0, 0};
        if (idx < ntiles) v = E2{tt[2 * idx], tt[2 * idx + 1]};
        s0[t] = v.c0;
        s1[t] = v.c1;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            E2 o{0, 0};
            if (t + off < 256) o = E2{s0[t + off], s1[t + off]};
            __syncthreads();
            v = gl::add(v, o);
            s0[t] = v.c0;
            s1[t] = v.c1;
            __syncthreads();
        }
        const E2 total{s0[0], s1[0]};
        const E2 after = t + 1 < 256 ? E2{s0[t + 1], s1[t + 1]} : E2{0, 0};
        __syncthreads();
        if (idx < ntiles) {
            const E2 ex = gl::add(after, carry);  // sum over all tiles after idx
            tt[2 * idx] = ex.c0;
            tt[2 * idx + 1] = ex.c1;
        }
        carry = gl::add(carry, total);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(1024) fri_divide_finish(DivArgs a, const uint64_t* __restrict__ scan,
                                                         const uint64_t* __restrict__ totals) {
    const int b = blockIdx.y;
    const size_t n = a.n, k = (size_t)blockIdx.x * 1024 + threadIdx.x;
    const uint64_t* zip = a.zip[b];
    const uint64_t* sc = scan + (size_t)b * 2 * n;
    uint64_t* q0 = a.qout + (size_t)b * 2 * n;
    uint64_t* q1 = q0 + n;
    const E2 carry{totals[((size_t)b * gridDim.x + blockIdx.x) * 2], totals[((size_t)b * gridDim.x + blockIdx.x) * 2 + 1]};
    const E2 S = gl::add(E2{sc[k], sc[n + k]}, carry);  // S_k = sum_{j >= k} F_j z^j
    if (k >= 1) {
        const E2 q = gl::mul(S, E2{zip[k], zip[n + k]});
        q0[k - 1] = q.c0;
        q1[k - 1] = q.c1;
    }
    if (k == n - 1) {
        q0[n - 1] = 0;
        q1[n - 1] = 0;
    }
}

// final[0] = 0 ; final[k+1] = q0[k] * shift + q1[k]   (multiplication by X)
__global__ void fri_final_kernel(const uint64_t* __restrict__ q, size_t n, E2 shift, uint64_t* __restrict__ fin) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    if (k == 0) { fin[0] = 0; fin[n] = 0; return; }
    E2 a{q[k - 1], q[n + k - 1]}, b{q[2 * n + k - 1], q[3 * n + k - 1]};
    E2 r = gl::add(gl::mul(a, shift), b);
    fin[k] = r.c0;
    fin[n + k] = r.c1;
}

// coeffs'[k] = sum_{i < arity} beta^i coeffs[arity k + i]; in: [2][len_in] SoA, out: [2][len_in / arity]
__global__ void fri_fold_kernel(const uint64_t* __restrict__ in, size_t len_in, uint32_t ab, E2 beta, uint64_t* __restrict__ out) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t len_out = len_in >> ab;
    if (k >= len_out) return;
    const int arity = 1 << ab;
    E2 acc{0, 0};
    for (int i = arity - 1; i >= 0; i--)
        acc = gl::add(gl::mul(acc, beta), E2{in[((size_t)k << ab) + i], in[len_in + ((size_t)k << ab) + i]});
    out[k] = acc.c0;
    out[len_out + k] = acc.c1;
}

// =====================================================================================================
// query gathers
// =====================================================================================================
__global__ void gather_rows_kernel(const uint64_t* __restrict__ lde, size_t m, uint32_t ncols, const uint32_t* __restrict__ idx,
                                   uint64_t* __restrict__ out) {
    const uint32_t q = blockIdx.y;
    const size_t x = idx[q];
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += gridDim.x * blockDim.x)
        out[(size_t)q * ncols + c] = lde[(size_t)c * m + x];
}

// siblings of leaf idx[q] >> shift from level 0 up to (log_leaves - cap) levels; out [q][nsib][4]
__global__ void gather_siblings_kernel(const uint64_t* __restrict__ tree, uint32_t log_leaves, uint32_t nsib, uint32_t shift,
                                       const uint32_t* __restrict__ idx, uint64_t* __restrict__ out) {
    const uint32_t q = blockIdx.x;
    const uint32_t l = threadIdx.x >> 2, w = threadIdx.x & 3;
    if (l >= nsib) return;
    size_t off = 0;
    for (uint32_t k = 0; k < l; k++) off += (size_t)1 << (log_leaves - k);
    size_t node = (((size_t)idx[q] >> shift) >> l) ^ 1;
    out[((size_t)q * nsib + l) * 4 + w] = tree[(off + node) * 4 + w];
}

// FRI round leaf: 2^ab ext values (interleaved c0, c1) at leaf index idx[q] >> shift; vals: [2][len] leaf order
__global__ void gather_fri_leaf_kernel(const uint64_t* __restrict__ vals, size_t len, uint32_t shift, uint32_t ab,
                                       const uint32_t* __restrict__ idx, uint64_t* __restrict__ out) {
    const uint32_t q = blockIdx.x, t = threadIdx.x;  // 2 * arity threads
    const size_t leaf = (size_t)idx[q] >> shift;
    out[((size_t)q << (ab + 1)) + t] = vals[(size_t)(t & 1) * len + (leaf << ab) + (t >> 1)];
}

// every gather of a query phase in ONE launch: block (q, task) copies what one of the three kernels above copies for query q
// (a dozen launches of a few microseconds each were 0.2 ms of launch gaps in every proof's FRI tail)
__global__ void __launch_bounds__(256) gather_tasks_kernel(const QueryGatherTask* __restrict__ tasks, const uint32_t* __restrict__ idx) {
    const QueryGatherTask t = tasks[blockIdx.y];
    const uint32_t q = blockIdx.x;
    const size_t x = idx[q];
    if (t.type == 0) {                       // a row of an oracle: ncols = b, column stride = a
        for (uint32_t c = threadIdx.x; c < t.b; c += blockDim.x) t.out[(size_t)q * t.b + c] = t.src[(size_t)c * t.a + x];
    } else if (t.type == 1) {                // Merkle siblings: log_leaves = b, nsib = c, shift = d
        for (uint32_t e = threadIdx.x; e < 4 * t.c; e += blockDim.x) {
            const uint32_t l = e >> 2, w = e & 3;
            size_t off = 0;
            for (uint32_t k = 0; k < l; k++) off += (size_t)1 << (t.b - k);
            const size_t node = ((x >> t.d) >> l) ^ 1;
            t.out[((size_t)q * t.c + l) * 4 + w] = t.src[(off + node) * 4 + w];
        }
    } else {                                 // a FRI leaf: len = a, shift = b, arity bits = c
        const size_t leaf = x >> t.b;
        for (uint32_t e = threadIdx.x; e < (2u << t.c); e += blockDim.x)
            t.out[((size_t)q << (t.c + 1)) + e] = t.src[(size_t)(e & 1) * t.a + (leaf << t.c) + (e >> 1)];
    }
}

}  // namespace

// ---- host wrappers -----------------------------------------------------------------------------------
int sipp_k_gather_tasks(sipp_ctx* ctx, const QueryGatherTask* d_tasks, uint32_t n_tasks, const uint32_t* d_idx, uint32_t nq) {
    if (!n_tasks || !nq) return SIPP_OK;
    ProfScope ps(ctx, "query_gather");
    hipLaunchKernelGGL(gather_tasks_kernel, dim3(nq, n_tasks), dim3(256), 0, ctx->stream, d_tasks, d_idx);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_z_columns(sipp_ctx* ctx, const air_spec_t* a, const uint64_t* d_trace, uint32_t log_n, const uint64_t beta[2],
                     const uint64_t gamma[2], uint64_t* d_zv) {
    const size_t n = (size_t)1 << log_n;
    const int P = 2 * a->n_checked;
    // rows per lane: 8 on long traces (two challenges x (numerators, denominators) x 8 rows live in registers), 4 otherwise
    const int zr = (n >= 16384) ? 8 : 4;
    const size_t zblock = (size_t)256 * zr;
    const unsigned nblk = (unsigned)(n / zblock);
    if (n % zblock) return sipp_fail(ctx, SIPP_E_BADARG, "z_columns: n must be a multiple of 1024");
    ArenaScope scope(ctx);   // the scratch goes back on EVERY exit path (the stream is ordered: later users of the block wait)
    uint64_t* totals = arena_alloc_t<uint64_t>(ctx, (size_t)2 * P * nblk);     // numerator and denominator totals of every block
    if (!totals) return SIPP_E_NOMEM;
    {
        ProfScope ps(ctx, "z_phase_a");
        if (zr == 8)
            hipLaunchKernelGGL(z_phase_a<8>, dim3(nblk, P / 2), dim3(256), 0, ctx->stream, d_trace, n, a->n_main, a->n_checked,
                               a->checked_base, gamma[0], gamma[1], beta[0], beta[1], d_zv, totals);
        else
            hipLaunchKernelGGL(z_phase_a<4>, dim3(nblk, P / 2), dim3(256), 0, ctx->stream, d_trace, n, a->n_main, a->n_checked,
                               a->checked_base, gamma[0], gamma[1], beta[0], beta[1], d_zv, totals);
    }
    {
        ProfScope ps(ctx, "z_phase_bc");
        hipLaunchKernelGGL(z_phase_b, dim3(P), dim3(256), 0, ctx->stream, totals, (int)nblk);
        if (zr == 8)
            hipLaunchKernelGGL(z_phase_c<8>, dim3(nblk, P), dim3(256), 0, ctx->stream, d_zv, n, totals);
        else
            hipLaunchKernelGGL(z_phase_c<4>, dim3(nblk, P), dim3(256), 0, ctx->stream, d_zv, n, totals);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

// natural-order NTT on the host (sizes up to 2 R = 1024: the value-periodic columns below): out[i] = sum_j a[j] w^(i j), w a primitive
// 2^log_n-th root (inverse: w^-1 and the factor 1 / n)
static void host_ntt(uint64_t* a, uint32_t log_n, bool inverse) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) {
        const size_t j = gl::bitrev((uint32_t)i, log_n);
        if (i < j) std::swap(a[i], a[j]);
    }
    uint64_t root = gl::root_of_unity(log_n);
    if (inverse) root = gl::inv(root);
    for (uint32_t s = 1; s <= log_n; s++) {
        const size_t mlen = (size_t)1 << s, h = mlen >> 1;
        uint64_t wm = root;
        for (uint32_t k = s; k < log_n; k++) wm = gl::sqr(wm);
        for (size_t k = 0; k < n; k += mlen) {
            uint64_t w = 1;
            for (size_t j = 0; j < h; j++) {
                const uint64_t t = gl::mul(w, a[k + j + h]), u = a[k + j];
                a[k + j] = gl::add(u, t);
                a[k + j + h] = gl::sub(u, t);
                w = gl::mul(w, wm);
            }
        }
    }
    if (inverse) {
        const uint64_t ninv = gl::inv((uint64_t)n);
        for (size_t i = 0; i < n; i++) a[i] = gl::mul(a[i], ninv);
    }
}

// PRECONDITION: every cell of d_lde / d_zlde / d_aux is CANONICAL (< p).  The lazy POLY path adds and subtracts +-1-coefficient
// monomials with gl::add / gl::sub directly on raw operands (no product in between to canonicalise them): gl::sub(a, b) with b >= p
// would be off by 2^32 - 1.  Every producer keeps this: the tree sweeps (ntt_tree.hip) and lde_column canonicalise at their last store
// (gl::canon in the forward sweeps' store phase), the aux LDEs come from the same transforms.  A lazy store there would corrupt the
// quotient and show only as a proof mismatch.
int sipp_k_quotient(sipp_ctx* ctx, const air_spec_t* a, uint32_t log_n, const uint64_t* d_lde, const uint64_t* d_zlde,
                    size_t lde_stride, const uint64_t* d_aux, const uint64_t alpha[2], const uint64_t beta[2],
                    const uint64_t gamma[2], uint64_t* d_out) {
    const uint32_t log_m = log_n + 1;
    const size_t n = (size_t)1 << log_n, m = (size_t)1 << log_m;
    QuotArgs q{};
    q.lde = d_lde; q.zlde = d_zlde; q.aux = d_aux; q.lde_stride = lde_stride;
    q.prog = sipp_air_prog_device(ctx, a);
    if (!q.prog) return SIPP_E_HIP;
    q.prog_len = a->prog_len;
    q.W = a->n_main + 2 * a->n_checked; q.nm = a->n_main; q.nc = a->n_checked; q.cbase = a->checked_base;
    q.tbits = a->table_bits; q.log_n = log_n; q.log_m = log_m;
    q.alpha[0] = alpha[0]; q.alpha[1] = alpha[1]; q.gamma[0] = gamma[0]; q.gamma[1] = gamma[1];
    q.beta[0] = beta[0]; q.beta[1] = beta[1];
    const uint64_t g = gl::root_of_unity(log_n), wm = gl::root_of_unity(log_m);
    // periodic tables over natural LDE index i mod 2 m_k
    for (int k = 0; k < AIR_N_PERIODIC; k++) {
        const uint64_t mk = (uint64_t)AIR_PERIODIC[k][0], r0 = (uint64_t)AIR_PERIODIC[k][1];
        uint64_t* t = sipp_table_get(ctx, 200 + k, log_n, 0);
        if (!t) {
            const uint64_t K = n / mk;
            std::vector<uint64_t> v(2 * mk);
            const uint64_t gs = gl::inv(gl::pow(g, r0)), scale = gl::mul(K % gl::P, gl::inv(n % gl::P));
            for (uint64_t i = 0; i < 2 * mk; i++) {
                uint64_t x = gl::mul(gl::GEN, gl::pow(wm, i));
                uint64_t y = gl::mul(x, gs);
                uint64_t num = gl::sub(gl::pow(y, n), 1), den = gl::sub(gl::pow(y, K), 1);
                v[i] = gl::mul(gl::mul(num, gl::inv(den)), scale);
            }
            SIPP_TRY(sipp_table_put(ctx, 200 + k, log_n, 0, v, &t));
        }
        q.per_tab[k] = t;
        q.per_mask[k] = (uint32_t)(2 * mk - 1);
    }
    // value-periodic columns: column k is P_k(x^(N/R)) with P_k interpolating its R values over the order-R subgroup; on the
    // quotient coset x_i = 7 w_2N^i it takes the 2 R values P_k(7^(N/R) w_2R^i): a coset LDE of P_k, built once per trace length
    q.vper_tab = nullptr;
    q.vper_mask = 0;
    if (a->n_vflag + a->n_vconst > 0) {
        const uint32_t lr = (uint32_t)a->log_rows;
        const size_t R = (size_t)1 << lr, R2 = 2 * R;
        const int nv = a->n_vflag + a->n_vconst;
        uint64_t* t = sipp_table_get(ctx, 104, log_n, (uint64_t)a->kind);
        if (!t) {
            std::vector<uint64_t> tab((size_t)nv * R2), col(R2);
            const uint64_t shift = gl::pow(gl::GEN, (uint64_t)1 << (log_n - lr));
            std::map<std::vector<int64_t>, int> seen;      // many columns repeat (the constants are mostly zero)
            std::vector<int64_t> vals(R);
            for (int k = 0; k < nv; k++) {
                for (size_t r = 0; r < R; r++) vals[r] = air_vper_value(a, k, (int)r);
                auto it = seen.find(vals);
                if (it != seen.end()) {
                    memcpy(&tab[(size_t)k * R2], &tab[(size_t)it->second * R2], R2 * 8);
                    continue;
                }
                seen.emplace(vals, k);
                for (size_t r = 0; r < R; r++) col[r] = gl::from_i64(vals[r]);
                host_ntt(col.data(), lr, true);
                uint64_t f = 1;
                for (size_t j = 0; j < R; j++) {
                    col[j] = gl::mul(col[j], f);
                    f = gl::mul(f, shift);
                }
                for (size_t j = R; j < R2; j++) col[j] = 0;
                host_ntt(col.data(), lr + 1, false);
                memcpy(&tab[(size_t)k * R2], col.data(), R2 * 8);
            }
            SIPP_TRY(sipp_table_put(ctx, 104, log_n, (uint64_t)a->kind, tab, &t));
        }
        q.vper_tab = t;
        q.vper_mask = (uint32_t)(R2 - 1);
    }
    const uint64_t sN = gl::pow(gl::GEN, n);
    q.zh[0] = gl::sub(sN, 1);
    q.zh[1] = gl::sub(gl::neg(sN), 1);
    q.zh_inv[0] = gl::inv(q.zh[0]);
    q.zh_inv[1] = gl::inv(q.zh[1]);
    q.ninv = gl::inv(n % gl::P);
    q.g_inv = gl::inv(g);
    q.w_m = wm;
    for (int i = 0; i < 16; i++) q.p_limbs[i] = AIR_BN_P_LIMBS[i];
    q.out = d_out;
    for (int ch = 0; ch < 2; ch++) {
        // (alpha = 0 has probability 2^-64 per challenge; the factorised fold divides by alpha, so refuse it loudly)
        if (alpha[ch] == 0) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "quotient: a zero constraint challenge");
        const uint64_t gi = gl::inv(alpha[ch]);
        q.gamma_inv[ch] = gi;
        q.alpha16[ch] = gl::pow(alpha[ch], 16);
        uint64_t x = 1;
        for (int i = 0; i < 8; i++) {
            q.ginv3[ch][i][0] = (uint32_t)x & 0x3FFFFFu;
            q.ginv3[ch][i][1] = (uint32_t)(x >> 22) & 0x3FFFFFu;
            q.ginv3[ch][i][2] = (uint32_t)(x >> 44);
            x = gl::mul(x, gi);
        }
    }
    // segment table: offsets, kinds, alpha powers (tiny; rebuilt per proof because alpha changes)
    int n_gadget_segs = 0;
    ArenaScope scope(ctx);   // the scratch goes back on EVERY exit path (the stream is ordered: later users of the block wait)
    {
        std::vector<uint32_t> off, cnt;
        std::vector<int> ncons;
        const int64_t* w = a->prog;
        const int64_t* end = a->prog + a->prog_len;
        while (w < end && w[0] == 1) {
            off.push_back((uint32_t)(w - a->prog));
            cnt.push_back(0);
            if (w[6] != 2) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "quotient: gadgets must pair their limbs (group 2)");
            ncons.push_back(32 / (int)w[6] + 1);
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
            // many products: slices of about 40 (at most SIPP_QUOTIENT_MAX_SLICES), entries of the same offset with no constraints of their own
            if (np > 64) {
                const uint32_t nsl = (uint32_t)std::min<int64_t>(SIPP_QUOTIENT_MAX_SLICES, (np + 39) / 40);
                cnt.back() = 0x80000000u | (nsl << 8);
                for (uint32_t k = 1; k < nsl; k++) {
                    off.push_back(off.back());
                    cnt.push_back(0x80000000u | (nsl << 8) | k);
                    ncons.push_back(0);
                }
            }
        }
        while (w < end) {  // runs of POLY ops
            off.push_back((uint32_t)(w - a->prog));
            uint32_t k = 0;
            while (w < end && k < 64) {
                int64_t nmono = w[1];
                w += 2;
                for (int64_t mo = 0; mo < nmono; mo++) w += 2 + 2 * w[1];
                k++;
            }
            cnt.push_back(k);
            ncons.push_back((int)k);
        }
        const int ns = (int)off.size();
        q.n_seg = ns;
        n_gadget_segs = 0;
        auto is_gadget = [](uint32_t c) { return c == 0 || (c >> 31) != 0; };
        while (n_gadget_segs < ns && is_gadget(cnt[n_gadget_segs])) n_gadget_segs++;
        for (int g = n_gadget_segs; g < ns; g++)
            if (is_gadget(cnt[g])) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "quotient: gadgets must precede the polynomial constraints");
        std::vector<uint64_t> host((size_t)ns * 2 + ns + 2 + 192, 0);      // + [2][64][3] u32: limbs of alpha_c^e, e < 64
        int after = 0;
        for (int g = ns - 1; g >= 0; g--) {
            host[2 * g] = gl::pow(alpha[0], (uint64_t)after);
            host[2 * g + 1] = gl::pow(alpha[1], (uint64_t)after);
            after += ncons[g];
        }
        uint32_t* h32 = reinterpret_cast<uint32_t*>(&host[2 * ns]);
        memcpy(h32, off.data(), (size_t)ns * 4);
        memcpy(h32 + ns, cnt.data(), (size_t)ns * 4);
        const size_t ap_at = (size_t)ns * 2 + (((size_t)2 * ns + 1) >> 1);     // behind the two u32 tables (ns + ns words)
        {
            uint32_t* ap = reinterpret_cast<uint32_t*>(&host[ap_at]);
            for (int ch = 0; ch < 2; ch++) {
                uint64_t x = 1;
                for (int e = 0; e < 64; e++) {
                    ap[(ch * 64 + e) * 3 + 0] = (uint32_t)x & 0x3FFFFFu;
                    ap[(ch * 64 + e) * 3 + 1] = (uint32_t)(x >> 22) & 0x3FFFFFu;
                    ap[(ch * 64 + e) * 3 + 2] = (uint32_t)(x >> 44);
                    x = gl::mul(x, alpha[ch]);
                }
            }
        }
        uint64_t* d_tab = arena_alloc_t<uint64_t>(ctx, host.size());
        q.part = arena_alloc_t<uint64_t>(ctx, (size_t)ns * 2 * m);
        if (!d_tab || !q.part) return SIPP_E_NOMEM;
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(d_tab, host.data(), host.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));  // `host` goes out of scope
        q.seg_pow = d_tab;
        q.seg_off = reinterpret_cast<const uint32_t*>(d_tab + 2 * ns);
        q.seg_cnt = q.seg_off + ns;
        q.apoly3 = reinterpret_cast<const uint32_t*>(d_tab + ap_at);
    }
    {
        // alpha powers of the K constraints that follow the program (range table 3, lookups 2 nc, permutation 4 nc)
        q.K = 3 + 6 * a->n_checked;
        const int nw = 6 + 12 * a->n_checked;  // weights: one limb triple each
        uint32_t* d_ap = arena_alloc_t<uint32_t>(ctx, (size_t)3 * nw);
        if (!d_ap) return SIPP_E_NOMEM;
        hipLaunchKernelGGL(alpha_pow3_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream, alpha[0], alpha[1],
                           q.K, a->n_checked, d_ap);
        q.apow3 = d_ap;
        q.alphaK[0] = gl::pow(alpha[0], (uint64_t)q.K);
        q.alphaK[1] = gl::pow(alpha[1], (uint64_t)q.K);
    }
    {
        ProfScope ps(ctx, "quotient_prog");
        // the program's segments: gadgets first, then the runs of polynomial constraints (the order the table above was built in)
        if (n_gadget_segs)
            hipLaunchKernelGGL(quotient_prog_kernel<false>, dim3((unsigned)(m / 64), (unsigned)n_gadget_segs), dim3(64), 0, ctx->stream, q, 0);
        if (q.n_seg > n_gadget_segs)
            hipLaunchKernelGGL(quotient_prog_kernel<true>, dim3((unsigned)(m / 64), (unsigned)(q.n_seg - n_gadget_segs)), dim3(64), 0, ctx->stream,
                               q, n_gadget_segs);
    }
    {
        ProfScope ps(ctx, "quotient_rest");
        // a thin domain (<= 2^15 points = 128 blocks) with many checked columns: the columns in up to 16 ranges of at least 32, two kernels
        q.chunks = (int)sipp_quotient_rest_chunks(log_n, a->n_checked);
        if (q.chunks > 1) {
            q.rest_part = arena_alloc_t<uint64_t>(ctx, (size_t)q.chunks * 6 * m);
            if (!q.rest_part) return SIPP_E_NOMEM;
            hipLaunchKernelGGL(quotient_rest_kernel<true>, dim3((unsigned)(m / 256), (unsigned)q.chunks), dim3(256), 0, ctx->stream, q);
            hipLaunchKernelGGL(quotient_finish_kernel, dim3((unsigned)(m / 256)), dim3(256), 0, ctx->stream, q);
        } else {
            q.chunks = 1;
            hipLaunchKernelGGL(quotient_rest_kernel<false>, dim3((unsigned)(m / 256)), dim3(256), 0, ctx->stream, q);
        }
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_pow_table(sipp_ctx* ctx, gl::E2 base, size_t n, uint64_t* d_tab) {
    ProfScope ps(ctx, "pow_table");
    hipLaunchKernelGGL(pow_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, base, n, d_tab);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_pow_table4(sipp_ctx* ctx, const gl::E2 base[4], size_t n, uint64_t* const d_tab[4]) {
    ProfScope ps(ctx, "pow_table");
    Pow4Args a;
    for (int i = 0; i < 4; i++) {
        a.base[i] = base[i];
        a.tab[i] = d_tab[i];
    }
    a.n = n;
    hipLaunchKernelGGL(pow_table4_kernel, dim3((unsigned)((n + 255) / 256), 4), dim3(256), 0, ctx->stream, a);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_openings(sipp_ctx* ctx, const uint64_t* d_coeffs, size_t ncols, size_t n, const uint64_t* d_t0,
                    const uint64_t* d_t1, uint64_t* d_out) {
    if (!ncols) return SIPP_OK;
    ProfScope ps(ctx, "openings");
    // about 2048 blocks whatever the column count; a segment is at least 1024 rows
    const size_t want = (2048 + ncols - 1) / ncols, segs = std::min<size_t>(std::min<size_t>(64, want), n / 1024);
    if (segs <= 1) {
        hipLaunchKernelGGL(openings_kernel, dim3((unsigned)ncols), dim3(256), 0, ctx->stream, d_coeffs, n, (uint32_t)ncols, d_t0, d_t1,
                           d_out);
    } else {
        ArenaScope scope(ctx);   // (the stream is ordered: later users of the block wait)
        uint64_t* partial = arena_alloc_t<uint64_t>(ctx, ncols * segs * 4);
        if (!partial) return SIPP_E_NOMEM;
        const size_t seg_len = (n + segs - 1) / segs;
        hipLaunchKernelGGL(openings_seg_kernel, dim3((unsigned)ncols, (unsigned)segs), dim3(256), 0, ctx->stream, d_coeffs, n, seg_len,
                           d_t0, d_t1, partial);
        hipLaunchKernelGGL(openings_fold_kernel, dim3((unsigned)((4 * ncols + 255) / 256)), dim3(256), 0, ctx->stream, partial,
                           (uint32_t)ncols, (uint32_t)segs, d_out);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_openings3(sipp_ctx* ctx, const uint64_t* const d_coeffs[3], const uint32_t ncols[3], size_t n, const uint64_t* d_t0,
                     const uint64_t* d_t1, uint64_t* d_out) {
    const unsigned total = ncols[0] + ncols[1] + ncols[2];
    if (!total) return SIPP_OK;
    ProfScope ps(ctx, "openings");
    Open3Args a;
    for (int i = 0; i < 3; i++) {
        a.coeffs[i] = d_coeffs[i];
        a.ncols[i] = ncols[i];
    }
    a.n = n; a.t0 = d_t0; a.t1 = d_t1; a.out = d_out;
    const bool grouped = !(ctx->kernel_routes & SIPP_ROUTE_OPENINGS_UNGROUPED);   // the fallback route stays tested (test_gpu_stark.py)
    if (!grouped || n < 1024) {
        hipLaunchKernelGGL(openings3_kernel, dim3(total), dim3(256), 0, ctx->stream, a);
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        return SIPP_OK;
    }
    OpenGArgs g;
    unsigned groups = 0;
    for (int i = 0; i < 3; i++) {
        g.coeffs[i] = d_coeffs[i];
        g.ncols[i] = ncols[i];
        groups += (ncols[i] + OG - 1) / OG;
    }
    // about 4096 blocks; a segment is at least 1024 rows and there are at most SIPP_OPENINGS_MAX_SEGS of them (the arena contract)
    const size_t segs = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(SIPP_OPENINGS_MAX_SEGS, (4096 + groups - 1) / groups), n / 1024));
    ArenaScope scope(ctx);   // (the stream is ordered: later users of the block wait)
    g.partial = arena_alloc_t<uint64_t>(ctx, (size_t)total * segs * 4);
    if (!g.partial) return SIPP_E_NOMEM;
    g.n = n; g.seg_len = (n + segs - 1) / segs; g.t0 = d_t0; g.t1 = d_t1;
    hipLaunchKernelGGL(openings_group_kernel, dim3(groups, (unsigned)segs), dim3(256), 0, ctx->stream, g);
    hipLaunchKernelGGL(openings_fold_kernel, dim3((4 * total + 255) / 256), dim3(256), 0, ctx->stream, g.partial, (uint32_t)total,
                       (uint32_t)segs, d_out);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_fri_final(sipp_ctx* ctx, const uint64_t* const src[3], const int cnt[3], size_t n, const uint32_t* d_apow3,
                     int n1, gl::E2 shift1, const uint64_t* d_zp[2], const uint64_t* d_zip[2], uint64_t* d_final) {
    ArenaScope scope(ctx);   // the scratch goes back on EVERY exit path (the stream is ordered: later users of the block wait)
    // enough (block, slice) pairs to fill the chip: short traces with many columns (Fq12) get more slices
    // (and no slice above the kernel's 1024 columns: MapToG2's 13 631 columns at 2^16 rows)
    const int total_cols = cnt[0] + cnt[1] + cnt[2];
    int slices = 8;
    while (slices < 64 && ((n / 256) * (size_t)slices < 2048 || (total_cols + slices - 1) / slices > 1024)) slices *= 2;
    uint64_t* partial = arena_alloc_t<uint64_t>(ctx, (size_t)slices * 4 * n);
    uint64_t* q = arena_alloc_t<uint64_t>(ctx, 4 * n);
    if (!partial || !q) return SIPP_E_NOMEM;
    CombArgs c{};
    for (int i = 0; i < 3; i++) { c.src[i] = src[i]; c.cnt[i] = cnt[i]; }
    c.n = n; c.apow3 = d_apow3; c.n1 = n1; c.slices = slices; c.partial = partial;
    if ((cnt[0] + cnt[1] + cnt[2] + slices - 1) / slices > 1024) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri_combine: more than 1024 columns per slice");
    {
        ProfScope ps(ctx, "fri_combine");
        hipLaunchKernelGGL(fri_combine_kernel, dim3((unsigned)(n / 256), slices), dim3(256), 0, ctx->stream, c);
    }
    DivArgs d{};
    d.partial = partial; d.slices = slices; d.n = n;
    d.zp[0] = d_zp[0]; d.zp[1] = d_zp[1]; d.zip[0] = d_zip[0]; d.zip[1] = d_zip[1];
    d.qout = q;
    const unsigned ntiles = (unsigned)(n / 1024);
    uint64_t* scan = arena_alloc_t<uint64_t>(ctx, 4 * n);
    uint64_t* totals = arena_alloc_t<uint64_t>(ctx, (size_t)4 * ntiles);
    if (!scan || !totals) return SIPP_E_NOMEM;
    if (n % 1024) return sipp_fail(ctx, SIPP_E_BADARG, "fri_divide: n must be a multiple of 1024");
    {
        ProfScope ps(ctx, "fri_divide");
        hipLaunchKernelGGL(fri_divide_tiles, dim3(ntiles, 2), dim3(1024), 0, ctx->stream, d, scan, totals);
        hipLaunchKernelGGL(fri_divide_carry, dim3(2), dim3(256), 0, ctx->stream, totals, (int)ntiles);
        hipLaunchKernelGGL(fri_divide_finish, dim3(ntiles, 2), dim3(1024), 0, ctx->stream, d, scan, totals);
        hipLaunchKernelGGL(fri_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, q, n, shift1, d_final);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

// (F(X) - F(z)) / (X - z) of the composition F = sum_j alpha^j col_j of one generic batch, then acc = acc * shift + quotient
int sipp_k_fri_batch_quotient(sipp_ctx* ctx, const uint64_t* const* d_cols, int total, size_t n, const uint32_t* d_apow3,
                              const uint64_t* d_zp, const uint64_t* d_zip, gl::E2 shift, bool first, uint64_t* d_acc) {
    ArenaScope scope(ctx);   // the scratch goes back on EVERY exit path (the stream is ordered: later users of the block wait)
    int slices = 1;
    while (slices < 64 && ((n / 256) * (size_t)slices < 1024 || (total + slices - 1) / slices > 1024)) slices *= 2;
    if ((total + slices - 1) / slices > 1024) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri: more than 65536 polynomials in one batch");
    if (n % 1024) return sipp_fail(ctx, SIPP_E_BADARG, "fri: degree must be a multiple of 1024");
    uint64_t* partial = arena_alloc_t<uint64_t>(ctx, (size_t)slices * 4 * n);
    uint64_t* q = arena_alloc_t<uint64_t>(ctx, 4 * n);
    uint64_t* scan = arena_alloc_t<uint64_t>(ctx, 4 * n);
    const unsigned ntiles = (unsigned)(n / 1024);
    uint64_t* totals = arena_alloc_t<uint64_t>(ctx, (size_t)4 * ntiles);
    if (!partial || !q || !scan || !totals) return SIPP_E_NOMEM;
    GCombArgs c{};
    c.cols = d_cols; c.total = total; c.n = n; c.apow3 = d_apow3; c.slices = slices; c.partial = partial;
    DivArgs d{};
    d.partial = partial; d.slices = slices; d.n = n;
    d.zp[0] = d_zp; d.zp[1] = d_zp; d.zip[0] = d_zip; d.zip[1] = d_zip;
    d.qout = q;
    {
        ProfScope ps(ctx, "fri_combine");
        hipLaunchKernelGGL(fri_gcombine_kernel, dim3((unsigned)(n / 256), slices), dim3(256), 0, ctx->stream, c);
    }
    {
        ProfScope ps(ctx, "fri_divide");
        hipLaunchKernelGGL(fri_divide_tiles, dim3(ntiles, 1), dim3(1024), 0, ctx->stream, d, scan, totals);
        hipLaunchKernelGGL(fri_divide_carry, dim3(1), dim3(256), 0, ctx->stream, totals, (int)ntiles);
        hipLaunchKernelGGL(fri_divide_finish, dim3(ntiles, 1), dim3(1024), 0, ctx->stream, d, scan, totals);
        hipLaunchKernelGGL(fri_accum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_acc, q, n, shift, first ? 1 : 0);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}
int sipp_k_fri_mulx(sipp_ctx* ctx, const uint64_t* d_acc, size_t n, uint64_t* d_final) {
    hipLaunchKernelGGL(fri_mulx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_acc, n, d_final);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_fri_fold(sipp_ctx* ctx, const uint64_t* d_in, size_t len_in, uint32_t arity_bits, gl::E2 beta, uint64_t* d_out) {
    size_t len_out = len_in >> arity_bits;
    ProfScope ps(ctx, "fri_fold");
    hipLaunchKernelGGL(fri_fold_kernel, dim3((unsigned)((len_out + 255) / 256)), dim3(256), 0, ctx->stream, d_in, len_in, arity_bits,
                       beta, d_out);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_gather_rows(sipp_ctx* ctx, const uint64_t* d_lde, size_t m, uint32_t ncols, const uint32_t* d_idx, uint32_t nq,
                       uint64_t* d_out) {
    ProfScope ps(ctx, "query_gather");
    unsigned gx = (ncols + 255) / 256;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, nq), dim3(256), 0, ctx->stream, d_lde, m, ncols, d_idx, d_out);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_gather_siblings(sipp_ctx* ctx, const uint64_t* d_tree, uint32_t log_leaves, uint32_t nsib, uint32_t shift,
                           const uint32_t* d_idx, uint32_t nq, uint64_t* d_out) {
    if (!nsib) return SIPP_OK;
    ProfScope ps(ctx, "query_gather");
    hipLaunchKernelGGL(gather_siblings_kernel, dim3(nq), dim3(4 * 32), 0, ctx->stream, d_tree, log_leaves, nsib, shift, d_idx, d_out);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_gather_fri_leaf(sipp_ctx* ctx, const uint64_t* d_vals, size_t len, uint32_t shift, uint32_t arity_bits,
                           const uint32_t* d_idx, uint32_t nq, uint64_t* d_out) {
    ProfScope ps(ctx, "query_gather");
    hipLaunchKernelGGL(gather_fri_leaf_kernel, dim3(nq), dim3(2u << arity_bits), 0, ctx->stream, d_vals, len, shift, arity_bits, d_idx,
                       d_out);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}
