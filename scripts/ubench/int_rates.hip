// Micro-benchmark: issue rate of the 32/64-bit integer multiply forms on gfx950 (SURVEY 7.3.2 asks for it).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 4096
template <int OP>
__global__ void k(uint64_t* out, uint32_t a0, uint32_t b0) {
    uint32_t a = a0 + threadIdx.x, b = b0 ^ threadIdx.x;
    uint64_t acc[8];
    for (int i = 0; i < 8; i++) acc[i] = threadIdx.x * 7919u + i;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) acc[i] = (uint64_t)(uint32_t)acc[i] * a + acc[i];             // v_mad_u64_u32
            if (OP == 1) acc[i] = (uint32_t)acc[i] * a + b;                            // v_mul_lo_u32 (+add)
            if (OP == 2) acc[i] = __umulhi((uint32_t)acc[i], a) + b;                    // v_mul_hi_u32
            if (OP == 3) acc[i] = acc[i] + (acc[i] >> 3) + b;                           // 64-bit add/shift
            if (OP == 4) acc[i] = (uint32_t)acc[i] + a;                                 // v_add_u32
            if (OP == 5) acc[i] = __umul64hi(acc[i], ((uint64_t)a << 32) | b) + acc[i] * b;  // full 64x64 hi+lo
            if (OP == 6) acc[i] = __mul24((int)acc[i], (int)a) + b;                     // v_mul_u32_u24
            if (OP == 7) {                                                              // v_dot2_u32_u16
                typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
                const uint32_t x = (uint32_t)acc[i];
                acc[i] = __builtin_amdgcn_udot2(*(const v2u16*)&x, *(const v2u16*)&a, b, false);
            }
            if (OP == 8) acc[i] = __builtin_amdgcn_udot4((uint32_t)acc[i], a, b, false); // v_dot4_u32_u8
            if (OP == 9) acc[i] = __builtin_amdgcn_perm((uint32_t)acc[i], a, 0x05040100u) + 0; // v_perm_b32
        }
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int ops_per) {
    uint64_t* d;
    int blocks = 256 * 8, threads = 256;
    hipMalloc(&d, blocks * threads * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, threads>>>(d, 12345, 999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, threads>>>(d, 12345, 999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * threads * ITER * 8 * ops_per;
    printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.2f cycles per wave-instr per SIMD @2.4GHz)\n", name, ms,
           lane_ops / ms / 1e9, 2.4e9 * 1024 * 64 / (lane_ops / (ms * 1e-3)) );
    hipFree(d);
}

int main() {
    run<4>("v_add_u32", 1);
    run<3>("u64 add+shift+add", 1);
    run<0>("v_mad_u64_u32", 1);
    run<1>("v_mul_lo_u32+add", 1);
    run<2>("v_mul_hi_u32+add", 1);
    run<6>("v_mul_u32_u24+add", 1);
    run<5>("u64 mulhi + mullo", 1);
    run<7>("v_dot2_u32_u16", 1);
    run<8>("v_dot4_u32_u8", 1);
    run<9>("v_perm_b32", 1);
    return 0;
}
