// sipp_amd/csrc/gl_lazy.cuh -- lazy Goldilocks add / sub / canonicalisation as carry chains through VCC (gfx950, device only).
//
// Butterflies do not need canonical values in between: with w canonical (< p) and u ANY u64 congruent to the true value,
// u + w  and  u - w  need ONE conditional correction by 2^64 mod p = 2^32 - 1 and stay in [0, 2^64) -- five instructions each,
// against the compare-and-select forms of gl::add / gl::sub (eight and six, with 64-bit compares and SGPR-pair selects); only
// what is stored for good has to be canonical (four instructions).  The mask of a carry is  x - x - VCC  (v_subb): a carry-in read
// of VCC needs no wait states, a v_cndmask on a freshly written VCC needs two.
// Measured (scripts/ubench/butterfly.hip, enc_rates.hip): 794 -> 850-890 G forward butterflies/s.  The ENCODING is not what
// pays -- v_add_co in VOP2 (VCC) and VOP3 (SGPR pair) form issue at the same rate, ~1.7x a plain 32-bit add, and so does
// v_mad_u64_u32 -- the instruction count is.
#pragma once
#include <stdint.h>
#include "gl.hpp"

namespace gll {

__device__ __forceinline__ uint64_t pack(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// x in [0, 2^64) -> x mod p:  t = x + (2^32 - 1) carries out of 64 bits exactly when x >= p, and then t = x - p
__device__ __forceinline__ uint64_t canon(uint64_t x) {
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), tl, th;
    asm("v_add_co_u32_e32 %2, vcc, -1, %0\n\t"
        "v_addc_co_u32_e32 %3, vcc, 0, %1, vcc\n\t"
        "v_cndmask_b32_e32 %0, %0, %2, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %3, vcc"
        : "+v"(xl), "+v"(xh), "=&v"(tl), "=&v"(th)
        :
        : "vcc");
    return pack(xl, xh);
}

// u any u64, w < p  ->  any u64 congruent to u + w
__device__ __forceinline__ uint64_t add_nc(uint64_t u, uint64_t w) {
    uint32_t lo, hi, m;
    asm("v_add_co_u32_e32 %0, vcc, %3, %5\n\t"
        "v_addc_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %3, %3, vcc\n\t"      // m = -carry
        "v_add_co_u32_e32 %0, vcc, %2, %0\n\t"            // + (2^32 - 1) when the sum wrapped
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(lo), "=&v"(hi), "=&v"(m)
        : "v"((uint32_t)u), "v"((uint32_t)(u >> 32)), "v"((uint32_t)w), "v"((uint32_t)(w >> 32))
        : "vcc");
    return pack(lo, hi);
}

// u any u64, w < p  ->  any u64 congruent to u - w
__device__ __forceinline__ uint64_t sub_nc(uint64_t u, uint64_t w) {
    uint32_t lo, hi, m;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %3, %3, vcc\n\t"      // m = -borrow
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"            // - (2^32 - 1) when the difference wrapped
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(lo), "=&v"(hi), "=&v"(m)
        : "v"((uint32_t)u), "v"((uint32_t)(u >> 32)), "v"((uint32_t)w), "v"((uint32_t)(w >> 32))
        : "vcc");
    return pack(lo, hi);
}

}  // namespace gll
