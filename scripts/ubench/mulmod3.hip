// Goldilocks mulmod: compiler-generated (gl::mul_nc) vs a hand-written 3-way interleaved sequence that uses the
// carry-out of v_mad_u64_u32 / v_add_co directly (17 instead of 23 instructions per product).  Result on MI355X: both
// 1.4-1.5 T mulmod/s -- the SGPR-carry forms issue slower, so the compiler sequence stays.  Kept as a measurement.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../sipp_amd/csrc/gl.hpp"

struct U3 { uint64_t a, b, c; };

// three independent products x[i] * y[i] mod p -> any u64 representative
__device__ __forceinline__ void mulmod3(uint64_t& r0, uint64_t& r1, uint64_t& r2, uint64_t x0, uint64_t y0, uint64_t x1,
                                        uint64_t y1, uint64_t x2, uint64_t y2) {
    uint64_t L0, M0, H0, L1, M1, H1, L2, M2, H2;
    uint64_t c0, c1, c2;  // lane masks: carry out of the middle sum
    const uint32_t x0l = (uint32_t)x0, x0h = (uint32_t)(x0 >> 32), y0l = (uint32_t)y0, y0h = (uint32_t)(y0 >> 32);
    const uint32_t x1l = (uint32_t)x1, x1h = (uint32_t)(x1 >> 32), y1l = (uint32_t)y1, y1h = (uint32_t)(y1 >> 32);
    const uint32_t x2l = (uint32_t)x2, x2h = (uint32_t)(x2 >> 32), y2l = (uint32_t)y2, y2h = (uint32_t)(y2 >> 32);
    asm("v_mad_u64_u32 %0, vcc, %12, %13, 0\n\t"
        "v_mad_u64_u32 %3, vcc, %16, %17, 0\n\t"
        "v_mad_u64_u32 %6, vcc, %20, %21, 0\n\t"
        "v_mad_u64_u32 %1, vcc, %12, %15, 0\n\t"
        "v_mad_u64_u32 %4, vcc, %16, %19, 0\n\t"
        "v_mad_u64_u32 %7, vcc, %20, %23, 0\n\t"
        "v_mad_u64_u32 %1, %9, %14, %13, %1\n\t"
        "v_mad_u64_u32 %4, %10, %18, %17, %4\n\t"
        "v_mad_u64_u32 %7, %11, %22, %21, %7\n\t"
        "v_mad_u64_u32 %2, vcc, %14, %15, 0\n\t"
        "v_mad_u64_u32 %5, vcc, %18, %19, 0\n\t"
        "v_mad_u64_u32 %8, vcc, %22, %23, 0\n\t"
        : "=&v"(L0), "=&v"(M0), "=&v"(H0), "=&v"(L1), "=&v"(M1), "=&v"(H1), "=&v"(L2), "=&v"(M2), "=&v"(H2), "=&s"(c0),
          "=&s"(c1), "=&s"(c2)
        : "v"(x0l), "v"(y0l), "v"(x0h), "v"(y0h), "v"(x1l), "v"(y1l), "v"(x1h), "v"(y1h), "v"(x2l), "v"(y2l), "v"(x2h),
          "v"(y2h)
        : "vcc");
    // stage B: 128-bit assembly and t = lo - hh ; all 32-bit operands
    uint32_t lo1_0, hi0_0, hi1_0, t0_0, t1_0, nb_0;
    uint32_t lo1_1, hi0_1, hi1_1, t0_1, t1_1, nb_1;
    uint32_t lo1_2, hi0_2, hi1_2, t0_2, t1_2, nb_2;
    uint64_t k0, k1, k2;
#define HI(x) ((uint32_t)((x) >> 32))
#define LO(x) ((uint32_t)(x))
    asm("v_add_co_u32 %0, %18, %21, %22\n\t"        // lo1 = L.hi + M.lo
        "v_add_co_u32 %6, %19, %27, %28\n\t"
        "v_add_co_u32 %12, %20, %33, %34\n\t"
        "v_addc_co_u32 %1, %18, %23, %24, %18\n\t"  // hi0 = H.lo + M.hi + k
        "v_addc_co_u32 %7, %19, %29, %30, %19\n\t"
        "v_addc_co_u32 %13, %20, %35, %36, %20\n\t"
        "v_addc_co_u32 %2, %18, %25, 0, %18\n\t"    // hi1 = H.hi + k
        "v_addc_co_u32 %8, %19, %31, 0, %19\n\t"
        "v_addc_co_u32 %14, %20, %37, 0, %20\n\t"
        "v_addc_co_u32 %2, %18, %2, 0, %39\n\t"     // hi1 += carry of the middle sum
        "v_addc_co_u32 %8, %19, %8, 0, %40\n\t"
        "v_addc_co_u32 %14, %20, %14, 0, %41\n\t"
        "v_sub_co_u32 %3, %18, %26, %2\n\t"         // t0 = L.lo - hi1
        "v_sub_co_u32 %9, %19, %32, %8\n\t"
        "v_sub_co_u32 %15, %20, %38, %14\n\t"
        "v_subb_co_u32 %4, %18, %0, 0, %18\n\t"     // t1 = lo1 - borrow
        "v_subb_co_u32 %10, %19, %6, 0, %19\n\t"
        "v_subb_co_u32 %16, %20, %12, 0, %20\n\t"
        "v_subb_co_u32 %5, %18, 0, 0, %18\n\t"      // nb = -borrow
        "v_subb_co_u32 %11, %19, 0, 0, %19\n\t"
        "v_subb_co_u32 %17, %20, 0, 0, %20\n\t"
        : "=&v"(lo1_0), "=&v"(hi0_0), "=&v"(hi1_0), "=&v"(t0_0), "=&v"(t1_0), "=&v"(nb_0), "=&v"(lo1_1), "=&v"(hi0_1),
          "=&v"(hi1_1), "=&v"(t0_1), "=&v"(t1_1), "=&v"(nb_1), "=&v"(lo1_2), "=&v"(hi0_2), "=&v"(hi1_2), "=&v"(t0_2),
          "=&v"(t1_2), "=&v"(nb_2), "=&s"(k0), "=&s"(k1), "=&s"(k2)
        : "v"(HI(L0)), "v"(LO(M0)), "v"(LO(H0)), "v"(HI(M0)), "v"(HI(H0)), "v"(LO(L0)),   // 21..26
          "v"(HI(L1)), "v"(LO(M1)), "v"(LO(H1)), "v"(HI(M1)), "v"(HI(H1)), "v"(LO(L1)),   // 27..32
          "v"(HI(L2)), "v"(LO(M2)), "v"(LO(H2)), "v"(HI(M2)), "v"(HI(H2)), "v"(LO(L2)),   // 33..38
          "s"(c0), "s"(c1), "s"(c2));                                                      // 39..41
    // stage C: r = hi0 * eps + t (carry c) ; net = c - borrow ; r += net * eps
    uint64_t t_0 = ((uint64_t)t1_0 << 32) | t0_0, t_1 = ((uint64_t)t1_1 << 32) | t0_1, t_2 = ((uint64_t)t1_2 << 32) | t0_2;
    uint64_t q0, q1, q2;
    uint32_t n0, n1, n2;
    asm("v_mad_u64_u32 %0, %6, %9, -1, %12\n\t"
        "v_mad_u64_u32 %1, %7, %10, -1, %13\n\t"
        "v_mad_u64_u32 %2, %8, %11, -1, %14\n\t"
        "v_addc_co_u32 %3, %6, %15, 0, %6\n\t"      // net = nb + c
        "v_addc_co_u32 %4, %7, %16, 0, %7\n\t"
        "v_addc_co_u32 %5, %8, %17, 0, %8\n\t"
        "v_mad_i64_i32 %0, %6, %3, -1, %0\n\t"      // r -= net
        "v_mad_i64_i32 %1, %7, %4, -1, %1\n\t"
        "v_mad_i64_i32 %2, %8, %5, -1, %2\n\t"
        : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&s"(k0), "=&s"(k1), "=&s"(k2)
        : "v"(hi0_0), "v"(hi0_1), "v"(hi0_2), "v"(t_0), "v"(t_1), "v"(t_2), "v"(nb_0), "v"(nb_1), "v"(nb_2));
    r0 = q0 + ((uint64_t)n0 << 32);
    r1 = q1 + ((uint64_t)n1 << 32);
    r2 = q2 + ((uint64_t)n2 << 32);
}

template <int ASM>
__global__ void __launch_bounds__(256) k(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, int iters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x0 = in[3 * i], x1 = in[3 * i + 1], x2 = in[3 * i + 2];
    uint64_t y0 = x0 ^ 0x1234567, y1 = x1 + 99, y2 = ~x2;
    for (int it = 0; it < iters; it++) {
        uint64_t r0, r1, r2;
        if (ASM) {
            mulmod3(r0, r1, r2, x0, y0, x1, y1, x2, y2);
        } else {
            r0 = gl::mul_nc(x0, y0);
            r1 = gl::mul_nc(x1, y1);
            r2 = gl::mul_nc(x2, y2);
        }
        y0 = x0; y1 = x1; y2 = x2;
        x0 = r0; x1 = r1; x2 = r2;
    }
    out[3 * i] = gl::canon(x0);
    out[3 * i + 1] = gl::canon(x1);
    out[3 * i + 2] = gl::canon(x2);
}

int main() {
    const int n = 1 << 20, iters = 512;
    uint64_t *in, *o0, *o1;
    hipMallocManaged(&in, 3 * n * 8); hipMallocManaged(&o0, 3 * n * 8); hipMallocManaged(&o1, 3 * n * 8);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 3 * n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        in[i] = (i % 7 == 0) ? ~0ull - (s & 0xffff) : (i % 11 == 0) ? (s & 0xffffffffu) : s;
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int v = 0; v < 2; v++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (v) k<1><<<n / 256, 256>>>(in, o1, iters); else k<0><<<n / 256, 256>>>(in, o0, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
        }
    }
    long bad = 0;
    for (int i = 0; i < 3 * n; i++) bad += o0[i] != o1[i];
    printf("mismatches %ld of %d ; compiler %.3f ms, asm %.3f ms ; %.2f / %.2f G mulmod/s\n", bad, 3 * n, ms[0], ms[1],
           3.0 * n * iters / ms[0] / 1e6, 3.0 * n * iters / ms[1] / 1e6);
    return bad != 0;
}
