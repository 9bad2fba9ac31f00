// sipp_amd/csrc/poseidon.hip -- batched Poseidon leaf hashing and Merkle levels (one leaf / node per lane).
//
// Replaces plonky2's MerkleTree::new over PoseidonHash (hash/merkle_tree.rs, hashing.rs @ 541e127),
// reached from the reference through starky::prover::prove behind src/verifier_circuit.rs:133-135.
//
// Layout: the LDE is column-major [col][leaf] in LEAF order, so the 64 lanes of a wave read 64
// consecutive u64 of one column per load (512 B, fully coalesced) and no transpose ever exists.
// Digests are [node][4] u64 (32 B per lane, contiguous across the wave).
#include <type_traits>
#include "ctx.hpp"
// ten VGPRs for the 64-bit temporaries of the hand-scheduled Goldilocks product (gl_lazy.hpp): v140 .. v149 keep the one-state-per-lane
// leaf kernel at 150 VGPRs (152 without; its budget is 168 = three waves per SIMD)
#ifndef GLL_T
#define GLL_T 140
#endif
#include "poseidon.hpp"
#include "poseidon_pair.hpp"
#include "prover.hpp"

namespace {

// hash_or_noop(leaf): ncols <= 4 -> the zero-padded elements are the digest; otherwise the
// overwrite-mode sponge with rate 8 (hash_n_to_hash_no_pad).
#ifndef SIPP_LEAVES_WPE
#define SIPP_LEAVES_WPE 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SIPP_LEAVES_WPE, SIPP_LEAVES_WPE)))
poseidon_leaves_kernel(const uint64_t* __restrict__ lde, size_t col_stride,
                                                             uint32_t ncols, uint64_t n_leaves,
                                                             uint64_t* __restrict__ digests) {
    // no early exit: the matrix-pipe form of the linear layers needs every lane of the wave (poseidon.hpp::permute<true>); a lane
    // past the end hashes the last leaf again and skips the store
    const uint64_t j0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = j0 < n_leaves ? j0 : n_leaves - 1;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* p = lde + j;
    __shared__ uint64_t stash[11 * 256];      // poseidon.hpp::partial_rounds_blocked: eleven state words per lane wait here
    // one call site for the permutation (it is ~50 KB of code): the ragged last chunk only masks its loads
#pragma unroll 1
    for (uint32_t c = 0; c < ncols; c += 8) {
        const uint32_t m = ncols - c;  // wave-uniform
#pragma unroll
        for (int i = 0; i < 8; i++)
            if ((uint32_t)i < m) s[i] = p[(size_t)(c + i) * col_stride];
        poseidon::permute<true>(s, stash + threadIdx.x, blockDim.x);
    }
    if (j0 >= n_leaves) return;
    uint64_t* d = digests + 4 * j;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
    d[3] = s[3];
}

// hash_or_noop: leaves of at most four elements are not hashed, the "digest" is the zero-padded leaf itself (the quotient chunks at
// blowup 2).  A kernel of its own: a copy, in no launch count or launch average of the hash kernel
__global__ void __launch_bounds__(256) poseidon_leaves_copy_kernel(const uint64_t* __restrict__ lde, size_t col_stride, uint32_t ncols,
                                                                  uint64_t n_leaves, uint64_t* __restrict__ digests) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    uint64_t* d = digests + 4 * j;
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) d[q] = q < ncols ? lde[j + (size_t)q * col_stride] : 0;
}

// ---- two lanes per state (poseidon_pair.hpp): the thin trees; lanes l and l + 32 of a wave share leaf 32 wave + (l & 31) ----
__global__ void __launch_bounds__(256) poseidon_leaves_pair_kernel(const uint64_t* __restrict__ lde, size_t col_stride,
                                                                  uint32_t ncols, uint64_t n_leaves,
                                                                  uint64_t* __restrict__ digests) {
    __shared__ poseidon_pair::Tables T;
    poseidon_pair::load_tables(T);
    const uint32_t lane = threadIdx.x & 63, h = lane >> 5;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t leaf = wave * 32 + (lane & 31);
    if (leaf >= n_leaves) return;  // n_leaves is a multiple of 32: whole waves leave together (the matrix pipe needs every lane)
    // the thin trees' waves ahead of the fat kernels' waves they share a SIMD with (issue arbitration: priority, then age): the Fq12 proof
    // is the last of an instance to end.  scripts/ab_prebuilt.sh, three alternating passes: 55.1 - 55.6 against 56.0 - 56.4 ms single
    // (s_setprio 3: 55.6 - 56.4), five queued 49.0 either way.  (Round 3 saw no effect: its two-lane kernel was 40 % longer.)
    __builtin_amdgcn_s_setprio(1);
    const poseidon::mfma_v4i afrag = poseidon_pair::mds_fragment(lane);
    uint64_t s[6] = {0, 0, 0, 0, 0, 0};
    const uint64_t* p = lde + leaf;
#pragma unroll 1
    for (uint32_t c = 0; c < ncols; c += 8) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t e = 6 * h + j;
            if (e < 8 && c + e < ncols) s[j] = p[(size_t)(c + e) * col_stride];
        }
        poseidon_pair::permute(s, lane, T, afrag);
    }
    if (h == 0) {
        uint64_t* d = digests + 4 * leaf;
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
        d[3] = s[3];
    }
}

// ---- one launch per SUBTREE instead of one per level ------------------------------------------------------------------
// Block b owns the 2^lw consecutive nodes [b 2^lw, (b + 1) 2^lw) of level `level0` and climbs `nlev` <= lw levels: every
// level it produces is written to the tree buffer (queries need the siblings of every level) and read back by the same
// block after a barrier (__syncthreads orders the block's global writes).  A 2^17-leaf tree takes 2 launches instead of 13,
// a 2^22-leaf tree 2 instead of 18; the wide levels run with all lanes busy, only the tip of a subtree (< 64 / < 16 parents)
// leaves lanes idle.  tree: levels back to back, level l at digest offset sum_{i<l} (n_leaves >> i).
struct NoTables {};
template <bool THIN>
__global__ void __launch_bounds__(256) merkle_subtree_kernel(uint64_t* tree, uint32_t log_leaves,
                                                            uint32_t level0, uint32_t lw, uint32_t nlev) {
    // THIN (levels of at most 2^16 parents): two lanes per node (poseidon_pair.hpp), 128 parents per trip of the block
    __shared__ typename std::conditional<THIN, poseidon_pair::Tables, NoTables>::type T;
    // FAT: the leaf kernel's permutation (linear layers on the matrix pipe; eleven state words per lane wait in LDS).  Its launches
    // (sipp_k_merkle_levels) give every lane a parent on every level -- 1024, 512, 256 per block -- as the matrix-pipe form requires
    __shared__ uint64_t stash[THIN ? 1 : 11 * 256];
    poseidon::mfma_v4i afrag = {0, 0, 0, 0};
    if constexpr (THIN) {
        poseidon_pair::load_tables(T);
        afrag = poseidon_pair::mds_fragment(threadIdx.x & 63);
        // the thin levels are a chain of dependent permutations (one per level) run by a few waves: ahead of the fat kernels' waves they
        // share a SIMD with (issue arbitration: priority, then age), like the thin leaf kernel's.  scripts/ab_prebuilt.sh, two boxes, five
        // alternating passes: lone instance -1.1 ms (55.6 -> 54.3; 56.4 -> 55.2), queue -0.1 ... -0.4; levels 2 / 3 the same.  The same
        // level on every non-hash kernel costs 1 - 3 ms, on the thin chain kernels alone (curve chains, FRI tail) adds nothing.
        __builtin_amdgcn_s_setprio(1);
    }
    uint64_t off = 0;
    for (uint32_t l = 0; l < level0; l++) off += (uint64_t)1 << (log_leaves - l);
    uint64_t n_level = (uint64_t)1 << (log_leaves - level0);       // nodes of the current level in the whole tree
    uint64_t first = (uint64_t)blockIdx.x << lw;                  // this block's first node on the current level
    uint32_t cnt = 1u << lw;                                       // this block's nodes on the current level
    for (uint32_t s = 0; s < nlev; s++) {
        const uint64_t* child = tree + 4 * (off + first);
        uint64_t* parent = tree + 4 * (off + n_level + (first >> 1));
        cnt >>= 1;
        if constexpr (THIN) {
            const uint32_t lane = threadIdx.x & 63, h = lane >> 5, w0 = (threadIdx.x >> 6) * 32;
            for (uint32_t base = 0; base < cnt; base += 128) {
                // whole waves run the permutation together (the matrix pipe reads every lane's registers); a wave past the end skips it
                if (base + w0 >= cnt) continue;
                const uint32_t i = base + w0 + (lane & 31);
                const bool live = i < cnt;
                uint64_t st[6] = {0, 0, 0, 0, 0, 0};
                if (live) {
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        const uint32_t e = 6 * h + j;
                        if (e < 8) st[j] = child[8 * (uint64_t)i + e];
                    }
                }
                poseidon_pair::permute(st, lane, T, afrag);
                if (live && h == 0) {
                    uint64_t* d = parent + 4 * (uint64_t)i;
                    d[0] = st[0];
                    d[1] = st[1];
                    d[2] = st[2];
                    d[3] = st[3];
                }
            }
        } else {
#pragma unroll 1
            for (uint32_t i = threadIdx.x; i < cnt; i += 256) {
                uint64_t st[12];
                const uint64_t* c = child + 8 * (uint64_t)i;
#pragma unroll
                for (int k = 0; k < 8; k++) st[k] = c[k];
#pragma unroll
                for (int k = 8; k < 12; k++) st[k] = 0;
                poseidon::permute<true>(st, stash + threadIdx.x, 256);
                uint64_t* d = parent + 4 * (uint64_t)i;
                d[0] = st[0];
                d[1] = st[1];
                d[2] = st[2];
                d[3] = st[3];
            }
        }
        __syncthreads();
        off += n_level;
        n_level >>= 1;
        first >>= 1;
    }
}

__global__ void __launch_bounds__(256) poseidon_permute_kernel(uint64_t* states, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s[12];
#pragma unroll
    for (int q = 0; q < 12; q++) s[q] = states[12 * i + q];
    poseidon::permute(s);
#pragma unroll
    for (int q = 0; q < 12; q++) states[12 * i + q] = s[q];
}

// FRI layer tree leaves: leaf k = 16 consecutive (leaf-order) extension values, flattened (c0, c1)
__global__ void __launch_bounds__(256) fri_leaves_kernel(const uint64_t* __restrict__ vals, uint64_t len,
                                                        uint64_t* __restrict__ digests) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (len >> 4)) return;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    const uint64_t* c0 = vals + 16 * k;
    const uint64_t* c1 = vals + len + 16 * k;
#pragma unroll 1
    for (int chunk = 0; chunk < 4; chunk++) {
#pragma unroll
        for (int e = 0; e < 8; e++) s[e] = (e & 1) ? c1[4 * chunk + (e >> 1)] : c0[4 * chunk + (e >> 1)];
        poseidon::permute(s);
    }
    uint64_t* d = digests + 4 * k;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

// the same with two lanes per leaf (poseidon_pair.hpp): FRI trees are small (2^13, 2^9, 2^5 leaves at n = 128) and sit on the
// latency-bound tail of a proof, where a lone wave per SIMD takes 50 us per one-lane permutation and 23 us per two-lane one
__global__ void __launch_bounds__(256) fri_leaves_pair_kernel(const uint64_t* __restrict__ vals, uint64_t len,
                                                             uint64_t* __restrict__ digests) {
    __shared__ poseidon_pair::Tables T;
    poseidon_pair::load_tables(T);
    const uint32_t lane = threadIdx.x & 63, h = lane >> 5;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nl = len >> 4;
    if (wave * 32 >= nl) return;        // whole waves; inside the last one every lane runs the permutation (the matrix pipe reads them all)
    const uint64_t k = wave * 32 + (lane & 31);
    const bool live = k < nl;
    const uint64_t kk = live ? k : 0;
    const poseidon::mfma_v4i afrag = poseidon_pair::mds_fragment(lane);
    const uint64_t* c0 = vals + 16 * kk;
    const uint64_t* c1 = vals + len + 16 * kk;
    uint64_t s[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int chunk = 0; chunk < 4; chunk++) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const uint32_t e = 6 * h + j;
            if (e < 8) s[j] = (e & 1) ? c1[4 * chunk + (e >> 1)] : c0[4 * chunk + (e >> 1)];
        }
        poseidon_pair::permute(s, lane, T, afrag);
    }
    if (!live || h) return;
    uint64_t* d = digests + 4 * k;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
    d[3] = s[3];
}

// any arity: leaf k = 2^ab extension values = 2^(ab + 1) words; <= 4 words are the digest themselves (hash_or_noop)
__global__ void __launch_bounds__(64) fri_leaves_any_kernel(const uint64_t* __restrict__ vals, uint64_t len, uint32_t ab,
                                                           uint64_t* __restrict__ digests) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (len >> ab)) return;
    const uint32_t words = 2u << ab;
    const uint64_t* c0 = vals + (k << ab);
    const uint64_t* c1 = vals + len + (k << ab);
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    if (words <= 4) {
        s[0] = c0[0];
        s[1] = c1[0];
        if (words == 4) {
            s[2] = c0[1];
            s[3] = c1[1];
        }
    } else {
#pragma unroll 1
        for (uint32_t chunk = 0; chunk < words / 8; chunk++) {
#pragma unroll
            for (int e = 0; e < 8; e++) s[e] = (e & 1) ? c1[4 * chunk + (e >> 1)] : c0[4 * chunk + (e >> 1)];
            poseidon::permute(s);
        }
    }
    uint64_t* d = digests + 4 * k;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3];
}

// proof-of-work grind: candidate nonce w = base + lane; response = state[7] after absorbing (in_buf, w)
struct PowArgs {
    uint64_t state[12];
    uint64_t in_buf[8];
    uint32_t n_in, pow_bits, resp_word;
    uint64_t base;
    unsigned long long* result;
};
__global__ void __launch_bounds__(256) pow_kernel(PowArgs a) {
    const uint64_t w = a.base + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = a.state[i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if ((uint32_t)i < a.n_in) s[i] = a.in_buf[i];
        if ((uint32_t)i == a.n_in) s[i] = w;
    }
    poseidon::permute(s);
    const uint64_t resp = a.resp_word == 7 ? s[7] : s[0];
    if ((resp >> (64 - a.pow_bits)) == 0) atomicMin(a.result, (unsigned long long)w);
}

// thin launches (the Fq12 trees, the upper Merkle levels, FRI layers): up to this many states two lanes work on one state.
// Measured both ways (round 3): one state per lane for the Fq12 trees costs 74-75 against 58-59 ms per instance single and 60-61
// against 51 ms queued -- the 1374 sequential permutations per leaf become the critical path.
constexpr uint64_t thin_threshold() { return 65536; }

}  // namespace

int sipp_k_fri_leaves(sipp_ctx* ctx, const uint64_t* d_vals, size_t len, uint32_t arity_bits, uint64_t* d_digests) {
    uint64_t nl = len >> arity_bits;
    ProfScope ps(ctx, "fri_leaves");
    if (arity_bits != 4) {
        if (arity_bits < 1 || arity_bits > 4) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "fri_leaves: arity must be 2, 4, 8 or 16");
        hipLaunchKernelGGL(fri_leaves_any_kernel, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, ctx->stream, d_vals, (uint64_t)len,
                           arity_bits, d_digests);
    } else if (nl <= thin_threshold())
        hipLaunchKernelGGL(fri_leaves_pair_kernel, dim3((unsigned)((2 * nl + 255) / 256)), dim3(256), 0, ctx->stream, d_vals,
                           (uint64_t)len, d_digests);
    else
        hipLaunchKernelGGL(fri_leaves_kernel, dim3((unsigned)((nl + 63) / 64)), dim3(64), 0, ctx->stream, d_vals, (uint64_t)len,
                           d_digests);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_pow_search(sipp_ctx* ctx, const uint64_t state[12], const uint64_t* in_buf, uint32_t n_in, uint32_t resp_word,
                      uint32_t pow_bits, uint64_t* witness) {
    if (pow_bits == 0) { *witness = 0; return SIPP_OK; }
    if (pow_bits > 32 || n_in > 7 || (resp_word != 0 && resp_word != 7))
        return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "pow_search: unsupported parameters");
    ArenaMark mk = arena_mark(ctx);
    unsigned long long* d_res = arena_alloc_t<unsigned long long>(ctx, 1);
    if (!d_res) return SIPP_E_NOMEM;
    PowArgs a;
    memcpy(a.state, state, sizeof a.state);
    memset(a.in_buf, 0, sizeof a.in_buf);
    memcpy(a.in_buf, in_buf, n_in * sizeof(uint64_t));
    a.n_in = n_in; a.pow_bits = pow_bits; a.resp_word = resp_word; a.result = d_res;
    // expected 2^pow_bits trials: a batch of 2^(pow_bits + 1) succeeds with probability 1 - e^-2; batches go in
    // ascending order and atomicMin keeps the smallest valid nonce, so the witness is the global minimum
    // (at most 2^28 nonces per launch: the grid stays far inside what a launch accepts for every allowed pow_bits)
    const uint32_t log_batch = pow_bits + 1 < 12 ? 12 : pow_bits + 1 > 28 ? 28 : pow_bits + 1;
    const uint64_t batch = (uint64_t)1 << log_batch;
    unsigned long long h_res = ~0ull;
    SIPP_CHECK_HIP(ctx, hipMemsetAsync(d_res, 0xff, sizeof(unsigned long long), ctx->stream));
    for (uint64_t base = 0; base < ((uint64_t)1 << 44); base += batch) {
        a.base = base;
        {
            ProfScope ps(ctx, "pow_grind");
            hipLaunchKernelGGL(pow_kernel, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx->stream, a);
        }
        SIPP_CHECK_HIP(ctx, hipGetLastError());
        SIPP_CHECK_HIP(ctx, hipMemcpyAsync(&h_res, d_res, sizeof h_res, hipMemcpyDeviceToHost, ctx->stream));
        SIPP_CHECK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h_res != ~0ull) break;
    }
    arena_release(ctx, mk);
    if (h_res == ~0ull) return sipp_fail(ctx, SIPP_E_UNSUPPORTED, "pow_search: no witness found");
    *witness = (uint64_t)h_res;
    return SIPP_OK;
}

int sipp_poseidon_init_constants(sipp_ctx* ctx) {
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_rc), SIPP_POSEIDON_RC, sizeof(SIPP_POSEIDON_RC)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_fast_first), SIPP_POSEIDON_FAST_FIRST,
                                          sizeof(SIPP_POSEIDON_FAST_FIRST)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_fast_scalar), SIPP_POSEIDON_FAST_SCALAR,
                                          sizeof(SIPP_POSEIDON_FAST_SCALAR)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_fast_mi), SIPP_POSEIDON_FAST_MI,
                                          sizeof(SIPP_POSEIDON_FAST_MI)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_fast_vs), SIPP_POSEIDON_FAST_VS,
                                          sizeof(SIPP_POSEIDON_FAST_VS)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_fast_what), SIPP_POSEIDON_FAST_WHAT,
                                          sizeof(SIPP_POSEIDON_FAST_WHAT)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_blk3), SIPP_POSEIDON_BLK3, sizeof(SIPP_POSEIDON_BLK3)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_comb3), SIPP_POSEIDON_COMB3, sizeof(SIPP_POSEIDON_COMB3)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_comb_c), SIPP_POSEIDON_COMB_C, sizeof(SIPP_POSEIDON_COMB_C)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::d_dense_a), SIPP_POSEIDON_DENSE_A, sizeof(SIPP_POSEIDON_DENSE_A)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon::c_dense_start), SIPP_POSEIDON_DENSE_START, sizeof(SIPP_POSEIDON_DENSE_START)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon_pair::d_pair_a), SIPP_POSEIDON_PAIR_A, sizeof(SIPP_POSEIDON_PAIR_A)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon_pair::d_pair_mds_a), SIPP_POSEIDON_PAIR_MDS_A, sizeof(SIPP_POSEIDON_PAIR_MDS_A)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon_pair::d_pair_cc), SIPP_POSEIDON_PAIR_CC3, sizeof(SIPP_POSEIDON_PAIR_CC3)));
    SIPP_CHECK_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(poseidon_pair::d_pair_start), SIPP_POSEIDON_PAIR_START, sizeof(SIPP_POSEIDON_PAIR_START)));
    return SIPP_OK;
}

int sipp_k_poseidon_leaves(sipp_ctx* ctx, const uint64_t* d_lde, size_t col_stride, size_t ncols, uint32_t log_leaves,
                           uint64_t* d_digests) {
    if (ncols == 0 || ncols > 0xffffffffull) return sipp_fail(ctx, SIPP_E_BADARG, "poseidon_leaves: bad ncols");
    uint64_t n = (uint64_t)1 << log_leaves;
    // profile names: "poseidon_leaves" = one state per lane (the dominant kernel: every tree of > 2^16 leaves); the thin trees'
    // two-lane kernel and the unhashed <= 4-column leaves (hash_or_noop: a copy) have names of their own.
    // Thin launches (32 .. 2^16 leaves: the Fq12 trees, MapToG2's): two lanes per state (poseidon_pair.hpp) -- 8.7 k instructions per
    // lane and permutation, 23 us per dependent permutation alone, 13.3 ms for 2^14 leaves x 4942 columns.  History: four lanes over DPP
    // quads (round 1: 8.1 k per lane, also 23 us, twice the lanes) and two lanes over DPP pairs on the VALU alone (rounds 2 - 3: 12.9 k,
    // 30 us) -- the instance is bound by total instruction issue, the lone Fq12 proof by the count per lane.
    const bool thin = ncols > 4 && n <= thin_threshold() && n >= 32;
    ProfScope ps(ctx, ncols <= 4 ? "poseidon_leaves_noop" : thin ? "poseidon_leaves_pair" : "poseidon_leaves");
    if (ncols <= 4) {
        hipLaunchKernelGGL(poseidon_leaves_copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_lde, col_stride,
                           (uint32_t)ncols, n, d_digests);
    } else if (thin) {
        unsigned grid = (unsigned)((2 * n + 255) / 256);
        hipLaunchKernelGGL(poseidon_leaves_pair_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_lde, col_stride,
                           (uint32_t)ncols, n, d_digests);
    } else {
        // 256-lane blocks (one wave per SIMD of a CU) measured best: 64-lane blocks spread unevenly (1.10 vs 1.61 G perm/s
        // at 2^17 leaves), 512 is slightly slower; trees of fewer than 32 leaves: one wave
        const unsigned bs = n <= 65536 ? 64 : 256;
        // (cutting this launch into k launches over leaf ranges, so that its waves retire in batches and the other proofs' thin kernels
        // get placed in between, was measured in round 3: 58.8 -> 63.8 / 82.0 / 124.5 ms per instance for k = 2 / 4 / 8 -- a chunk has
        // too few waves to hide its own latency; one launch stays)
        unsigned grid = (unsigned)((n + bs - 1) / bs);
        hipLaunchKernelGGL(poseidon_leaves_kernel, dim3(grid), dim3(bs), 0, ctx->stream, d_lde, col_stride,
                           (uint32_t)ncols, n, d_digests);
    }
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}

int sipp_k_merkle_levels(sipp_ctx* ctx, uint64_t* d_tree, uint32_t log_leaves, uint32_t cap_height) {
    if (cap_height > log_leaves) cap_height = log_leaves;
    {   // one launch per SUBTREE (146 -> 28 launches per n = 128 instance; the per-level form's kernel time was 4.5 against 5.2 ms)
        uint32_t level = 0;
        const uint32_t top = log_leaves - cap_height;     // levels to produce
        while (level < top) {
            const uint32_t log_nodes = log_leaves - level;
            // subtree size: as many blocks as the level allows while a block still has >= 2^8 nodes (full lanes on its
            // wide levels); the last launch is a single block over <= 2^10 nodes that climbs to the cap
            uint32_t lw = log_nodes <= 10 ? log_nodes : log_nodes >= 18 ? 10 : 8;
            uint32_t nlev = lw < top - level ? lw : top - level;
            const uint64_t widest_parents = (uint64_t)1 << (log_nodes - 1);
            if (widest_parents > thin_threshold()) {
                // FAT levels (one state per lane): a block climbs only while every lane has a parent -- 2^11 nodes -> 1024, 512, 256
                // parents, three levels per launch.  Climbing ten levels in one block (round 1 - 4) left its lanes idle on seven of
                // them: eleven permutation times for 1023 permutations, 36 % of the lanes' time (7.8 -> 3.0 ms for the three
                // 2^21-leaf trees of bench.py's outer_plonk leg); the thin levels below keep the subtree form (few nodes, launch-bound)
                lw = 11;
                nlev = 3 < top - level ? 3 : top - level;
            }
            const unsigned blocks = 1u << (log_nodes - lw);
            ProfScope ps(ctx, "merkle_subtree");
            if (widest_parents <= thin_threshold())
                hipLaunchKernelGGL(merkle_subtree_kernel<true>, dim3(blocks), dim3(256), 0, ctx->stream, d_tree, log_leaves, level, lw, nlev);
            else
                hipLaunchKernelGGL(merkle_subtree_kernel<false>, dim3(blocks), dim3(256), 0, ctx->stream, d_tree, log_leaves, level, lw, nlev);
            SIPP_CHECK_HIP(ctx, hipGetLastError());
            level += nlev;
        }
        return SIPP_OK;
    }
}

int sipp_k_poseidon_permute(sipp_ctx* ctx, uint64_t* d_states, size_t n) {
    if (!n) return SIPP_OK;
    unsigned grid = (unsigned)((n + 255) / 256);
    ProfScope ps(ctx, "poseidon_permute");
    hipLaunchKernelGGL(poseidon_permute_kernel, dim3(grid), dim3(256), 0, ctx->stream, d_states, (uint64_t)n);
    SIPP_CHECK_HIP(ctx, hipGetLastError());
    return SIPP_OK;
}
