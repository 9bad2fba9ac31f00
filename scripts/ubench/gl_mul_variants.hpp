// scripts/ubench/gl_mul_variants.hpp -- Goldilocks products tried against the compiler's sequence (gl::mul_nc), kept as measured
// experiments (scripts/ubench/mulmod4.hip, butterfly.hip).  NOT product code: none of them paid inside a kernel --
//   VOP2 carry chain (mul_nc)                  1.26-1.28 T products/s against the compiler's 1.41-1.42 T
//   one hand-scheduled block, fixed temporaries (mul_nc3)   1.65-1.67 T alone (+17 %), nothing inside the transform kernels of
//                                              ntt_tree.hip (6.65 against 6.66 ms): they are not bound by the product's instruction count
//   two products interleaved per block (mul2_nc3)           1.60 T: the single block is not latency-bound
#pragma once
#ifndef GLL_T
#define GLL_T 100
#endif
#define GLL_REGS_INC "gl_lazy_regs.inc"   /* scripts/ubench: every register window */
#include "gl_lazy.hpp"

namespace gll {

// a b mod p for ANY u64 a, b -> any u64 congruent to the product.  The four 32 x 32 + 64 multiply-adds stay with the compiler
// (v_mad_u64_u32); the last accumulation and the whole 128 -> 64 reduction
//     lo + 2^64 hi  ==  lo - hh + hl (2^32 - 1)      (hi = hh 2^32 + hl;  2^64 = 2^32 - 1,  2^96 = -1  mod p)
// are ONE chain of fourteen VOP2 instructions with the carry in VCC: a borrow of lo - hh is repaid by - (2^32 - 1), a carry of the
// final addition by + (2^32 - 1); neither correction can wrap again (the bounds are plonky2's reduce128).
__device__ __forceinline__ uint64_t mul_nc_vop2(uint64_t a, uint64_t b) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t t0 = (uint64_t)a0 * b0;
    const uint64_t t1 = (uint64_t)a0 * b1 + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * b0 + (uint32_t)t1;
    const uint64_t t3 = (uint64_t)a1 * b1 + (t1 >> 32);       // + (t2 >> 32) inside the chain
    uint32_t r0, r1, m, t, hl = (uint32_t)t3, hh = (uint32_t)(t3 >> 32);
    asm("v_add_co_u32_e32 %4, vcc, %4, %8\n\t"                // hi += t2 >> 32
        "v_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %6, %5\n\t"                // lo - hh
        "v_subbrev_co_u32_e32 %1, vcc, 0, %7, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %6, %6, vcc\n\t"          // m = -borrow
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"                // - (2^32 - 1) if it wrapped
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32_e32 %2, vcc, 0, %4\n\t"                 // hl (2^32 - 1) = (hl - [hl != 0]) 2^32 + (2^32 - hl) mod 2^32
        "v_subbrev_co_u32_e32 %3, vcc, 0, %4, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %3, vcc\n\t"
        "v_subb_co_u32_e32 %2, vcc, %6, %6, vcc\n\t"          // m = -carry
        "v_add_co_u32_e32 %0, vcc, %2, %0\n\t"                // + (2^32 - 1) if it wrapped
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(m), "=&v"(t), "+v"(hl), "+v"(hh)
        : "v"((uint32_t)t0), "v"((uint32_t)t2), "v"((uint32_t)(t2 >> 32))
        : "vcc");
    return pack(r0, r1);
}

// What the instructions cost on gfx950 (scripts/ubench/enc_rates.hip): everything with a carry-out, a third operand or 64 bits --
// v_add_co, v_addc, v_cndmask, v_cmp_*_u64, v_lshl_add_u64 AND v_mad_u64_u32 -- takes ~1.7x the issue time of a plain 32-bit VOP2
// (v_add_u32, v_sub_u32, shifts, logic, v_mov); the encoding (VOP2 with VCC against VOP3 with an SGPR pair) makes no difference.
// two independent products in one block, interleaved instruction by instruction (twenty fixed temporaries): one product alone is a
// single dependent chain of 19 instructions, which a SIMD with three or four resident waves cannot hide
__device__ __forceinline__ void mul2_nc3(uint64_t& r, uint64_t& q, uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    asm(
        "v_mov_b32_e32 " GLL2_R9 ", 0\n\t"
        "v_mov_b32_e32 " GLL2_R19 ", 0\n\t"
        "v_mad_u64_u32 " GLL2_P0 ", vcc, %2, %4, 0\n\t"
        "v_mad_u64_u32 " GLL2_P10 ", vcc, %6, %8, 0\n\t"
        "v_mov_b32_e32 " GLL2_R8 ", " GLL2_R1 "\n\t"
        "v_mov_b32_e32 " GLL2_R18 ", " GLL2_R11 "\n\t"
        "v_mad_u64_u32 " GLL2_P2 ", vcc, %2, %5, " GLL2_P8 "\n\t"
        "v_mad_u64_u32 " GLL2_P12 ", vcc, %6, %9, " GLL2_P18 "\n\t"
        "v_mov_b32_e32 " GLL2_R8 ", " GLL2_R2 "\n\t"
        "v_mov_b32_e32 " GLL2_R18 ", " GLL2_R12 "\n\t"
        "v_mad_u64_u32 " GLL2_P4 ", vcc, %3, %4, " GLL2_P8 "\n\t"
        "v_mad_u64_u32 " GLL2_P14 ", vcc, %7, %8, " GLL2_P18 "\n\t"
        "v_mov_b32_e32 " GLL2_R8 ", " GLL2_R3 "\n\t"
        "v_mov_b32_e32 " GLL2_R18 ", " GLL2_R13 "\n\t"
        "v_mad_u64_u32 " GLL2_P6 ", vcc, %3, %5, " GLL2_P8 "\n\t"
        "v_mad_u64_u32 " GLL2_P16 ", vcc, %7, %9, " GLL2_P18 "\n\t"
        "v_mov_b32_e32 " GLL2_R8 ", " GLL2_R5 "\n\t"
        "v_mov_b32_e32 " GLL2_R18 ", " GLL2_R15 "\n\t"
        "v_lshl_add_u64 " GLL2_P6 ", " GLL2_P6 ", 0, " GLL2_P8 "\n\t"
        "v_lshl_add_u64 " GLL2_P16 ", " GLL2_P16 ", 0, " GLL2_P18 "\n\t"
        "v_mov_b32_e32 " GLL2_R1 ", " GLL2_R4 "\n\t"
        "v_mov_b32_e32 " GLL2_R11 ", " GLL2_R14 "\n\t"
        "v_mad_u64_u32 " GLL2_P2 ", vcc, " GLL2_R6 ", -1, " GLL2_P0 "\n\t"
        "v_subb_co_u32_e32 " GLL2_R8 ", vcc, " GLL2_R0 ", " GLL2_R0 ", vcc\n\t"
        "v_mad_u64_u32 " GLL2_P12 ", vcc, " GLL2_R16 ", -1, " GLL2_P10 "\n\t"
        "v_subb_co_u32_e32 " GLL2_R18 ", vcc, " GLL2_R10 ", " GLL2_R10 ", vcc\n\t"
        "v_lshl_add_u64 " GLL2_P2 ", " GLL2_P2 ", 0, " GLL2_P8 "\n\t"
        "v_lshl_add_u64 " GLL2_P12 ", " GLL2_P12 ", 0, " GLL2_P18 "\n\t"
        "v_sub_co_u32_e32 " GLL2_R2 ", vcc, " GLL2_R2 ", " GLL2_R7 "\n\t"
        "v_subbrev_co_u32_e32 " GLL2_R3 ", vcc, 0, " GLL2_R3 ", vcc\n\t"
        "v_subb_co_u32_e32 " GLL2_R5 ", vcc, " GLL2_R0 ", " GLL2_R0 ", vcc\n\t"
        "v_sub_co_u32_e32 " GLL2_R12 ", vcc, " GLL2_R12 ", " GLL2_R17 "\n\t"
        "v_subbrev_co_u32_e32 " GLL2_R13 ", vcc, 0, " GLL2_R13 ", vcc\n\t"
        "v_subb_co_u32_e32 " GLL2_R15 ", vcc, " GLL2_R10 ", " GLL2_R10 ", vcc\n\t"
        "v_lshrrev_b32_e32 " GLL2_R4 ", 31, " GLL2_R5 "\n\t"
        "v_lshrrev_b32_e32 " GLL2_R14 ", 31, " GLL2_R15 "\n\t"
        "v_lshl_add_u64 %0, " GLL2_P2 ", 0, " GLL2_P4 "\n\t"
        "v_lshl_add_u64 %1, " GLL2_P12 ", 0, " GLL2_P14
        : "=&v"(r), "=&v"(q)
        : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)),
          "v"((uint32_t)c), "v"((uint32_t)(c >> 32)), "v"((uint32_t)d), "v"((uint32_t)(d >> 32))
        : "vcc", GLL2_R0, GLL2_R1, GLL2_R2, GLL2_R3, GLL2_R4, GLL2_R5, GLL2_R6, GLL2_R7, GLL2_R8, GLL2_R9,
          GLL2_R10, GLL2_R11, GLL2_R12, GLL2_R13, GLL2_R14, GLL2_R15, GLL2_R16, GLL2_R17, GLL2_R18, GLL2_R19);
}

}  // namespace gll
