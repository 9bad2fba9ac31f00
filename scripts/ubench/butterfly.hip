// Forward tree butterfly (u, v) -> (u + s v, u - s v) over Goldilocks: what does one cost on gfx950?
//   A  canonical everywhere (gl::mul, gl::add, gl::sub -- the compiler's 64-bit forms: v_lshl_add_u64, v_cmp_*_u64, v_cndmask e64)
//   B  lazy in C++: w = canon(mul_nc), u +- w on ANY u64 u with one conditional +-EPS (compiler picks the encodings)
//   C  lazy with the add / sub / canon spelled in VOP2 carry-chain forms (inline asm, carries through VCC only)
//   D  C + the 128 -> 64 reduction of the product as one VOP2 chain (gll::mul_nc)
// Prints G butterflies/s per variant and checks that all three agree mod p.
// build: hipcc -O3 --offload-arch=gfx950 -I sipp_amd/csrc -I scripts/ubench scripts/ubench/butterfly.hip -o scripts/ubench/bin/butterfly
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "gl.hpp"
#include "gl_mul_variants.hpp"

template <int V>
__device__ __forceinline__ void bfly(uint64_t& u, uint64_t& v, uint64_t s) {
    if (V == 0) {
        const uint64_t w = gl::mul(v, s), a = u;
        u = gl::add(a, w);
        v = gl::sub(a, w);
    } else if (V == 1) {
        const uint64_t w = gl::canon(gl::mul_nc(v, s)), a = u;
        uint64_t t = a + w;
        u = t < a ? t + gl::EPS : t;
        uint64_t d = a - w;
        v = a < w ? d - gl::EPS : d;
    } else if (V == 2) {
        const uint64_t w = gll::canon(gl::mul_nc(v, s)), a = u;
        u = gll::add_nc(a, w);
        v = gll::sub_nc(a, w);
    } else {
        const uint64_t w = gll::canon(gll::mul_nc_vop2(v, s)), a = u;
        u = gll::add_nc(a, w);
        v = gll::sub_nc(a, w);
    }
}

template <int V>
__global__ void __launch_bounds__(256) k(const uint64_t* in, uint64_t* out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x[8];
    for (int q = 0; q < 8; q++) x[q] = in[8 * i + q];
    uint64_t s = (in[8 * i] | 1) % gl::P;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int sg = 0; sg < 3; sg++)
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (q & (1 << sg)) continue;
                bfly<V>(x[q], x[q + (1 << sg)], s);
            }
        s = gl::canon(s + 0x9E3779B97F4A7C15ull);
    }
    for (int q = 0; q < 8; q++) out[8 * i + q] = gl::canon(x[q]);
}

int main() {
    const int n = 1 << 19, iters = 256;
    uint64_t *in, *o[4];
    hipMallocManaged(&in, 8 * n * 8);
    for (int v = 0; v < 4; v++) hipMallocManaged(&o[v], 8 * n * 8);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 8 * n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        in[i] = (i % 5 == 0) ? ~0ull - (s & 0xfffff) : (i % 7 == 0) ? (s & 0xffff) : s;   // edge patterns: near 2^64, tiny
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[4];
    for (int v = 0; v < 4; v++)
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v == 0) k<0><<<n / 256, 256>>>(in, o[0], iters);
            if (v == 1) k<1><<<n / 256, 256>>>(in, o[1], iters);
            if (v == 2) k<2><<<n / 256, 256>>>(in, o[2], iters);
            if (v == 3) k<3><<<n / 256, 256>>>(in, o[3], iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[v], e0, e1);
        }
    long bad = 0;
    for (int i = 0; i < 8 * n; i++) bad += (o[0][i] != o[1][i]) + (o[0][i] != o[2][i]) + (o[0][i] != o[3][i]);
    const double ops = 12.0 * n * iters;
    printf("mismatches %ld ; canonical %.3f ms (%.1f G bfly/s) ; lazy C++ %.3f ms (%.1f G) ; lazy VOP2 asm %.3f ms (%.1f G) ; + VOP2 reduction %.3f ms (%.1f G)\n", bad,
           ms[0], ops / ms[0] / 1e6, ms[1], ops / ms[1] / 1e6, ms[2], ops / ms[2] / 1e6, ms[3], ops / ms[3] / 1e6);
    return bad != 0;
}
