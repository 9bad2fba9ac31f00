// Goldilocks reduction with VCC-only two-operand carry instructions (12 VOP2 + wait states) against the compiler's sequence
// (6 VOP3 + 5 VOP2): 1.38 against 1.45 T mulmod/s on MI355X, identical results -- the wait states between dependent VCC
// producers and consumers cost what the cheaper encodings save.  Kept as a measurement (round 2).
#include "../../sipp_amd/csrc/gl.hpp"
using namespace gl;
// VCC-only (VOP2) reduction
__device__ __forceinline__ uint64_t red_asm(uint64_t hi, uint64_t lo) {
    const uint32_t hl = (uint32_t)hi, hh = (uint32_t)(hi >> 32);
    const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32);
    uint32_t r0, r1, t, s0, a1, nb, corr, ah;
    const uint32_t z = 0;
    asm("v_add_co_u32_e32 %3, vcc, %8, %9\n\t"          // s0 = hh + hl, vcc = cs
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %2, vcc, 0, %12, vcc\n\t"     // t = cs
        "v_sub_co_u32_e32 %0, vcc, %10, %3\n\t"          // a0 = l0 - s0, vcc = b0
        "s_nop 1\n\t"
        "v_subb_co_u32_e32 %4, vcc, %11, %2, vcc\n\t"    // a1 = l1 - t - b0, vcc = b1
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %5, vcc, 0, %12, vcc\n\t"     // nb = b1
        "v_add_co_u32_e32 %1, vcc, %4, %9\n\t"           // b1h = a1 + hl, vcc = c1
        "v_sub_u32_e32 %5, %12, %5\n\t"                  // nb = -b1
        "s_nop 0\n\t"
        "v_addc_co_u32_e32 %6, vcc, 0, %5, vcc\n\t"      // corr = c1 - b1
        "v_ashrrev_i32_e32 %7, 31, %6\n\t"               // ah = corr >> 31
        "v_sub_co_u32_e32 %0, vcc, %0, %6\n\t"           // r0 = a0 - corr, vcc = beta
        "s_nop 1\n\t"
        "v_subb_co_u32_e32 %1, vcc, %1, %7, vcc\n\t"     // r1 = b1h - ah - beta
        "v_add_u32_e32 %1, %1, %6\n\t"                   // r1 += corr
        : "=&v"(r0), "=&v"(r1), "=&v"(t), "=&v"(s0), "=&v"(a1), "=&v"(nb), "=&v"(corr), "=&v"(ah)
        : "v"(hh), "v"(hl), "v"(l0), "v"(l1), "v"(z)
        : "vcc");
    return ((uint64_t)r1 << 32) | r0;
}
template <int V> __device__ __forceinline__ uint64_t mulv(uint64_t a, uint64_t b) {
    uint64_t hi, lo; mul_wide(a, b, hi, lo);
    return V ? red_asm(hi, lo) : reduce128_nc(hi, lo);
}
template <int V>
__global__ void __launch_bounds__(256) k(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x0 = in[3 * i], x1 = in[3 * i + 1], x2 = in[3 * i + 2];
    uint64_t y0 = x0 ^ 0x1234567, y1 = x1 + 99, y2 = ~x2;
    for (int it = 0; it < iters; it++) {
        uint64_t r0 = mulv<V>(x0, y0), r1 = mulv<V>(x1, y1), r2 = mulv<V>(x2, y2);
        y0 = x0; y1 = x1; y2 = x2; x0 = r0; x1 = r1; x2 = r2;
    }
    out[3 * i] = canon(x0); out[3 * i + 1] = canon(x1); out[3 * i + 2] = canon(x2);
}
#include <stdio.h>
int main() {
    const int n = 1 << 20, iters = 512;
    uint64_t *in, *o0, *o1;
    hipMallocManaged(&in, 3 * n * 8); hipMallocManaged(&o0, 3 * n * 8); hipMallocManaged(&o1, 3 * n * 8);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 3 * n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; in[i] = (i % 7 == 0) ? ~0ull - (s & 0xffff) : (i % 11 == 0) ? (s & 0xffffffffu) : (i % 13 == 0) ? 0 : s; }
    for (int rep = 0; rep < 2; rep++) {
        hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
        hipEventRecord(e0); k<0><<<n / 256, 256>>>(in, o0, iters); hipEventRecord(e1); k<1><<<n / 256, 256>>>(in, o1, iters); hipEventRecord(e2);
        hipDeviceSynchronize();
        float m0, m1; hipEventElapsedTime(&m0, e0, e1); hipEventElapsedTime(&m1, e1, e2);
        size_t bad = 0; for (int i = 0; i < 3 * n; i++) bad += o0[i] != o1[i];
        printf("compiler %.3f ms (%.2f T mulmod/s)   asm %.3f ms (%.2f T mulmod/s)   mismatches %zu\n", m0, 3.0 * n * iters / m0 / 1e9, m1, 3.0 * n * iters / m1 / 1e9, bad);
    }
}
