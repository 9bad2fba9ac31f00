// The full-round MDS layer of Poseidon-Goldilocks (12 x 12 circulant + diagonal, entries <= 49) on the MATRIX pipe (VERDICT r2 #3a):
// does moving the small-constant linear layer of the one-state-per-lane hash to v_mfma_i32_32x32x32_i8 pay?
//
//   VALU form (what poseidon.hpp::mds_full does): per output two chains of 12 v_mad_u64_u32 over the 32-bit halves + one 96-bit
//   reduction: 12 x (24 + ~10) instructions per state.
//   MFMA form: the 12 state words are cut into 8 byte planes (48 v_perm_b32 = six 4x4 byte transposes, 24 v_xor to make the bytes
//   signed: i8 operands are signed, the +128 per byte is repaid through the accumulator's start value); per byte plane ONE
//   v_mfma_i32_32x32x32_i8 with a block-diagonal A (rows 8g + 4h + .. x columns 16h + ..: the two wave halves are two independent
//   12 x 12 products, every lane finds its own twelve sums in accumulator registers 0 .. 11 -- no lane exchange); recombination
//   sum_b D_b 2^(8b) as two chains of four v_mad_u64_u32 with 2^(8b) in SGPRs, then the same 96-bit reduction: 12 x (8 + ~10).
// The kernel chains MDS layers (the output feeds the next layer), 64 states per wave, every CU busy; both forms are checked
// against each other mod p; the k index of the i8 operand bytes is found by trying both candidate maps against the VALU result.
// build: hipcc -O3 --offload-arch=gfx950 -I sipp_amd/csrc scripts/ubench/mfma_mds.hip -o scripts/ubench/bin/mfma_mds
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "gl.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__constant__ uint32_t c_circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

__device__ __forceinline__ void mds_valu(uint64_t s[12]) {
    constexpr uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            al += (uint64_t)lo[(i + r) % 12] * CIRC[i];
            ah += (uint64_t)hi[(i + r) % 12] * CIRC[i];
        }
        if (r == 0) {
            al += (uint64_t)lo[0] * 8u;
            ah += (uint64_t)hi[0] * 8u;
        }
        uint64_t l = al + (ah << 32);
        uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        s[r] = gl::reduce96_nc(h, l);
    }
}

// 4 x 4 byte transpose: in[i] = bytes (b0 b1 b2 b3) of element i  ->  out[b] = byte b of elements 0 .. 3
__device__ __forceinline__ void transpose4(const uint32_t in[4], uint32_t out[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(in[1], in[0], 0x05010400u);   // a0 b0 a1 b1   (a = in[0], b = in[1])
    const uint32_t t1 = __builtin_amdgcn_perm(in[1], in[0], 0x07030602u);   // a2 b2 a3 b3
    const uint32_t t2 = __builtin_amdgcn_perm(in[3], in[2], 0x05010400u);   // c0 d0 c1 d1
    const uint32_t t3 = __builtin_amdgcn_perm(in[3], in[2], 0x07030602u);   // c2 d2 c3 d3
    out[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);                     // a0 b0 c0 d0
    out[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);                     // a1 b1 c1 d1
    out[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    out[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// A fragment (constant per lane) in `afrag`; cstart = accumulator start values 128 * rowsum (register r -> output r)
__device__ __forceinline__ void mds_mfma(uint64_t s[12], v4i afrag, const v16i cstart, uint32_t p8, uint32_t p16, uint32_t p24) {
    uint32_t lo[12], hi[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        lo[i] = (uint32_t)s[i];
        hi[i] = (uint32_t)(s[i] >> 32);
    }
    // byte planes: plane[b][g] = byte b of elements 4g .. 4g+3 (b = 0..3 from the low words, 4..7 from the high words)
    uint32_t plane[8][3];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        uint32_t o[4];
        transpose4(lo + 4 * g, o);
#pragma unroll
        for (int b = 0; b < 4; b++) plane[b][g] = o[b] ^ 0x80808080u;
        transpose4(hi + 4 * g, o);
#pragma unroll
        for (int b = 0; b < 4; b++) plane[4 + b][g] = o[b] ^ 0x80808080u;
    }
    uint64_t al[12], ah[12];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        v4i bfrag = {(int)plane[b][0], (int)plane[b][1], (int)plane[b][2], (int)0x80808080u};   // elements 12..15: signed zero's twin, A is 0 there
        const v16i d = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag, bfrag, cstart, 0, 0, 0);
        const uint32_t sh = (b & 3) == 0 ? 1u : (b & 3) == 1 ? p8 : (b & 3) == 2 ? p16 : p24;
#pragma unroll
        for (int r = 0; r < 12; r++) {
            uint64_t& acc = b < 4 ? al[r] : ah[r];
            acc = (uint64_t)(uint32_t)d[r] * sh + ((b & 3) == 0 ? 0 : acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 12; r++) {
        uint64_t l = al[r] + (ah[r] << 32);
        uint32_t h = (uint32_t)(ah[r] >> 32) + (l < al[r] ? 1u : 0u);
        s[r] = gl::reduce96_nc(h, l);
    }
}

template <int V>
__global__ void __launch_bounds__(256) k(const uint64_t* in, uint64_t* out, int iters, const uint32_t* afrag_tab, uint32_t p8, uint32_t p16,
                                         uint32_t p24) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[12];
    for (int q = 0; q < 12; q++) s[q] = in[(size_t)q * gridDim.x * blockDim.x + i];
    const int lane = threadIdx.x & 63;
    v4i afrag = {(int)afrag_tab[4 * lane], (int)afrag_tab[4 * lane + 1], (int)afrag_tab[4 * lane + 2], (int)afrag_tab[4 * lane + 3]};
    v16i cstart;
    for (int r = 0; r < 16; r++) cstart[r] = r < 12 ? 128 * (256 + (r == 0 ? 8 : 0)) : 0;
    for (int it = 0; it < iters; it++) {
        if (V == 0) mds_valu(s);
        else mds_mfma(s, afrag, cstart, p8, p16, p24);
    }
    for (int q = 0; q < 12; q++) out[(size_t)q * gridDim.x * blockDim.x + i] = gl::canon(s[q]);
}

int main() {
    const int blocks = 256 * 6, threads = 256, n = blocks * threads, iters = 256;
    uint64_t *in, *o0, *o1;
    uint32_t* atab;
    (void)hipMallocManaged(&in, (size_t)12 * n * 8);
    (void)hipMallocManaged(&o0, (size_t)12 * n * 8);
    (void)hipMallocManaged(&o1, (size_t)12 * n * 8);
    (void)hipMallocManaged(&atab, 64 * 4 * 4);
    uint64_t x = 88172645463325252ULL;
    for (size_t i = 0; i < (size_t)12 * n; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        in[i] = (i % 5 == 0) ? ~0ull - (x & 0xfffff) : x;
    }
    const uint32_t CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms0 = 0, ms1 = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        k<0><<<blocks, threads>>>(in, o0, iters, atab, 1u << 8, 1u << 16, 1u << 24);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms0, e0, e1);
    }
    int good_map = -1;
    for (int map = 0; map < 2 && good_map < 0; map++) {
        // A[row][k]: row = (reg & 3) + 8 (reg >> 2) + 4 h carries output `reg` of the lanes of half h; its columns are the k's that half h supplies
        // candidate k maps of byte j (0..15) of lane half h:  0: k = 16 h + j    1: k = 8 h + (j & 7) + 16 (j >> 3)
        int8_t A[32][32] = {};
        auto kof = [&](int h, int j) { return map == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3); };
        for (int h = 0; h < 2; h++)
            for (int reg = 0; reg < 12; reg++) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                for (int i = 0; i < 12; i++) {      // out[reg] = sum_i s[(i + reg) % 12] CIRC[i] + s[reg] DIAG[reg]
                    const int e = (i + reg) % 12;
                    A[row][kof(h, e)] += (int8_t)CIRC[i];
                }
                if (reg == 0) A[row][kof(h, 0)] += 8;
            }
        for (int l = 0; l < 64; l++)
            for (int w = 0; w < 4; w++) {
                uint32_t v = 0;
                for (int bb = 0; bb < 4; bb++) v |= (uint32_t)(uint8_t)A[l & 31][kof(l >> 5, 4 * w + bb)] << (8 * bb);
                atab[4 * l + w] = v;
            }
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            k<1><<<blocks, threads>>>(in, o1, iters, atab, 1u << 8, 1u << 16, 1u << 24);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms1, e0, e1);
        }
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)12 * n; i++) bad += o0[i] != o1[i];
        printf("k map %d: %zu mismatches\n", map, bad);
        if (!bad) good_map = map;
    }
    const double layers = (double)n * iters;
    printf("MDS layers: VALU %.3f ms (%.2f G state-layers/s) ; MFMA i8 %.3f ms (%.2f G) ; k map %d ; ratio %.2f\n", ms0, layers / ms0 / 1e6, ms1,
           layers / ms1 / 1e6, good_map, ms0 / ms1);
    return good_map < 0;
}
