// Goldilocks modular add / sub: 64-bit formulation (v_lshl_add_u64 + v_cmp_u64) vs 32-bit carry chains
// (__builtin_addc / __builtin_subc -> v_add_co / v_addc_co).  Prints throughput and checks equality.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define P 0xFFFFFFFF00000001ULL
__device__ __forceinline__ uint64_t add_old(uint64_t a, uint64_t b) { uint64_t s = a + b; uint64_t t = s - P; return (s < a || s >= P) ? t : s; }
__device__ __forceinline__ uint64_t sub_old(uint64_t a, uint64_t b) { uint64_t d = a - b; return a < b ? d + P : d; }
__device__ __forceinline__ uint64_t add_new(uint64_t a, uint64_t b) {
    uint32_t c = 0, bw = 0;
    const uint32_t s0 = __builtin_addc((uint32_t)a, (uint32_t)b, 0u, &c);
    const uint32_t s1 = __builtin_addc((uint32_t)(a >> 32), (uint32_t)(b >> 32), c, &c);
    const uint32_t t0 = __builtin_subc(s0, 1u, 0u, &bw);
    const uint32_t t1 = __builtin_subc(s1, 0xFFFFFFFFu, bw, &bw);
    const bool take = c | !bw;
    return ((uint64_t)(take ? t1 : s1) << 32) | (take ? t0 : s0);
}
__device__ __forceinline__ uint64_t sub_new(uint64_t a, uint64_t b) {
    uint32_t bw = 0, k = 0;
    const uint32_t d0 = __builtin_subc((uint32_t)a, (uint32_t)b, 0u, &bw);
    const uint32_t d1 = __builtin_subc((uint32_t)(a >> 32), (uint32_t)(b >> 32), bw, &bw);
    const uint32_t r0 = __builtin_addc(d0, bw, 0u, &k);
    const uint32_t r1 = d1 - bw + k;
    return ((uint64_t)r1 << 32) | r0;
}
template <int V>
__global__ void __launch_bounds__(256) k(const uint64_t* in, uint64_t* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[4], y[4];
    for (int q = 0; q < 4; q++) { x[q] = in[4 * i + q] % P; y[q] = (x[q] * 0x9E3779B97F4A7C15ull) % P; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            uint64_t s = V ? add_new(x[q], y[q]) : add_old(x[q], y[q]);
            uint64_t d = V ? sub_new(x[q], y[q]) : sub_old(x[q], y[q]);
            x[q] = s; y[q] = d;
        }
    }
    for (int q = 0; q < 4; q++) out[4 * i + q] = x[q] ^ y[q];
}
int main() {
    const int n = 1 << 20, iters = 512;
    uint64_t *in, *o0, *o1;
    hipMallocManaged(&in, 4 * n * 8); hipMallocManaged(&o0, 4 * n * 8); hipMallocManaged(&o1, 4 * n * 8);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 4 * n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; in[i] = (i % 5 == 0) ? ~0ull - (s & 0xfffff) : s; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int v = 0; v < 2; v++) for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (v) k<1><<<n / 256, 256>>>(in, o1, iters); else k<0><<<n / 256, 256>>>(in, o0, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
    }
    long bad = 0;
    for (int i = 0; i < 4 * n; i++) bad += o0[i] != o1[i];
    const double ops = 8.0 * n * iters;   // add + sub per element per iteration
    printf("mismatches %ld ; 64-bit form %.3f ms (%.2f T ops/s), carry chains %.3f ms (%.2f T ops/s)\n", bad, ms[0], ops / ms[0] / 1e9,
           ms[1], ops / ms[1] / 1e9);
    return bad != 0;
}
