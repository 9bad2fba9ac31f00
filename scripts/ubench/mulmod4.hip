// Goldilocks product a b mod p (result ANY u64 congruent): the compiler's sequence (gl::mul_nc), the VOP2 carry chain
// (gll::mul_nc) and the carry-avoiding form (gll::mul_nc2).  G products/s, all three checked against each other mod p.
// build: hipcc -O3 --offload-arch=gfx950 -I sipp_amd/csrc -I scripts/ubench scripts/ubench/mulmod4.hip -o scripts/ubench/bin/mulmod4
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "gl.hpp"
#include "gl_mul_variants.hpp"

template <int V>
__global__ void __launch_bounds__(256) k(const uint64_t* in, uint64_t* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[8];
    for (int q = 0; q < 8; q++) x[q] = in[8 * i + q];
    uint64_t s = in[8 * i] | 1;
    for (int it = 0; it < iters; it++) {
        if (V == 4) {
#pragma unroll
            for (int q = 0; q < 8; q += 2) gll::mul2_nc3(x[q], x[q + 1], x[q], s, x[q + 1], s);
        } else
#pragma unroll
        for (int q = 0; q < 8; q++) x[q] = V == 0 ? gl::mul_nc(x[q], s) : V == 1 ? gll::mul_nc_vop2(x[q], s) : gll::mul_nc(x[q], s);
        s += 0x9E3779B97F4A7C15ull;
    }
    for (int q = 0; q < 8; q++) out[8 * i + q] = gl::canon(x[q]);
}

int main() {
    const int n = 1 << 19, iters = 512;
    uint64_t *in, *o[5];
    (void)hipMallocManaged(&in, 8 * n * 8);
    for (int v = 0; v < 5; v++) (void)hipMallocManaged(&o[v], 8 * n * 8);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 8 * n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        in[i] = (i % 5 == 0) ? ~0ull - (s & 0xfffff) : (i % 7 == 0) ? (s & 0xffff) : (i % 11 == 0) ? 0xFFFFFFFF00000000ull + (s & 3) : s;
    }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms[5];
    for (int v = 0; v < 5; v++)
        for (int rep = 0; rep < 2; rep++) {
            (void)hipEventRecord(e0);
            if (v == 0) k<0><<<n / 256, 256>>>(in, o[0], iters);
            if (v == 1) k<1><<<n / 256, 256>>>(in, o[1], iters);
            if (v == 2) k<3><<<n / 256, 256>>>(in, o[2], iters);
            if (v == 3) k<3><<<n / 256, 256>>>(in, o[3], iters);
            if (v == 4) k<4><<<n / 256, 256>>>(in, o[4], iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms[v], e0, e1);
        }
    long bad = 0;
    for (int i = 0; i < 8 * n; i++) bad += (o[0][i] != o[1][i]);
    long bad2 = 0, bad3 = 0;
    for (int i = 0; i < 8 * n; i++) { bad2 += o[0][i] != o[2][i]; bad3 += (o[0][i] != o[3][i]) + (o[0][i] != o[4][i]); }
    const double ops = 8.0 * n * iters;
    printf("mismatches %ld %ld %ld ; compiler %.3f ms (%.1f G mul/s) ; VOP2 chain %.3f ms (%.1f G) ; (unused) %.3f ms (%.1f G) ; one block, fixed temporaries %.3f ms (%.1f G) ; two products per block %.3f ms (%.1f G)\n", bad, bad2, bad3,
           ms[0], ops / ms[0] / 1e6, ms[1], ops / ms[1] / 1e6, ms[2], ops / ms[2] / 1e6, ms[3], ops / ms[3] / 1e6, ms[4], ops / ms[4] / 1e6);
    return (bad | bad3) != 0;
}
