// Latency of the BN254 base-field operations (sipp_amd/csrc/fq.hpp) in ONE wave with nothing else on its SIMD -- the situation of the
// trace fill's doubling chains (trace.hip curve_dbl_par_kernel: 16 waves on the whole chip, 255 dependent doublings): shader-clock
// cycles (s_memtime) per operation of a dependent chain, and of two / four independent chains interleaved in the same wave.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I sipp_amd/csrc scripts/ubench/fq_chain.hip -o scripts/ubench/bin/fq_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "fq.hpp"

using fq::Fq;

__device__ __forceinline__ Fq seed(uint32_t s) {
    Fq r;
    for (int i = 0; i < 8; i++) r.l[i] = (threadIdx.x + 1) * 2654435761u * (s + i) + i;
    r.l[7] &= 0x1fffffffu;
    return r;
}

// OP: 0 mul chain, 1 add chain, 2 sub chain, 3 two mul chains, 4 four mul chains, 5 raw mad chain (171 per "operation"),
//     6 raw mad, two accumulators, 7 raw 8-limb addc chain (no reduction)
template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t* out, int iters, uint64_t* stamps) {
    Fq x = seed(1), y = seed(2), x2 = seed(3), x3 = seed(4), x4 = seed(5);
    uint64_t acc = threadIdx.x, acc2 = threadIdx.x * 3;
    const uint32_t ma = x.l[0] & 0x1fffffffu, mb = y.l[0] & 0x1fffffffu;
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) x = fq::mul(x, y);
        if (OP == 1) x = fq::add(x, y);
        if (OP == 2) x = fq::sub(x, y);
        if (OP == 3) { x = fq::mul(x, y); x2 = fq::mul(x2, y); }
        if (OP == 4) { x = fq::mul(x, y); x2 = fq::mul(x2, y); x3 = fq::mul(x3, y); x4 = fq::mul(x4, y); }
        if (OP == 5) {
#pragma unroll
            for (int q = 0; q < 171; q++) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc) : "v"(ma), "v"(mb) : "s20", "s21");
        }
        if (OP == 6) {
#pragma unroll
            for (int q = 0; q < 86; q++) {
                asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(acc) : "v"(ma), "v"(mb) : "s20", "s21");
                asm volatile("v_mad_u64_u32 %0, s[22:23], %1, %2, %0" : "+v"(acc2) : "v"(ma), "v"(mb) : "s22", "s23");
            }
        }
        if (OP == 7) {
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) x.l[i] = __builtin_addc(x.l[i], y.l[i], c, &c);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t s = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ (uint32_t)acc2;
    for (int i = 0; i < 8; i++) s ^= x.l[i] ^ x2.l[i] ^ x3.l[i] ^ x4.l[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int OP>
static void run(const char* name, int per_iter) {
    const int iters = 2000, blocks = 16;
    uint32_t* out;
    uint64_t* st;
    hipMalloc(&out, blocks * 64 * 4);
    hipMalloc(&st, blocks * 16);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, st);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, out, iters, st);
    hipDeviceSynchronize();
    uint64_t h[2 * blocks];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime counts a constant 100 MHz on this part when it equals s_memrealtime: print both, and the time per operation
    const double ticks = (double)h[0] / iters / per_iter, real_ns = (double)h[1] * 10.0 / iters / per_iter;
    printf("%-44s %9.1f memtime ticks   %8.1f ns per operation\n", name, ticks, real_ns);
    hipFree(out);
    hipFree(st);
}

int main() {
    run<0>("fq::mul, dependent chain", 1);
    run<3>("fq::mul, two independent chains", 2);
    run<4>("fq::mul, four independent chains", 4);
    run<1>("fq::add, dependent chain", 1);
    run<2>("fq::sub, dependent chain", 1);
    run<7>("8-limb add with carry, no reduction", 1);
    run<5>("171 dependent v_mad_u64_u32", 1);
    run<6>("172 v_mad_u64_u32 in two accumulators", 1);
    return 0;
}
