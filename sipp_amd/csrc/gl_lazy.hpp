// sipp_amd/csrc/gl_lazy.hpp -- lazy Goldilocks add / sub / canonicalisation as carry chains through VCC (gfx950, device only).
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

#ifdef GLL_T
// ---- the product as ONE hand-scheduled block ------------------------------------------------------------------------------
// The compiler's code for the sequences above carries 23 - 31 instructions per product (it rebuilds every zero-extended 64-bit addend
// of v_mad_u64_u32 with moves, splits 64-bit additions, and selects through SGPR pairs); the block below has 16 (19 until round 4's last
// day).  Inline asm cannot name the halves of a 64-bit operand, so the 64-bit temporaries are FIXED physical registers listed as
// clobbers (GLL_T = first of ten consecutive VGPRs, even, defined by the translation unit BEFORE this header: poseidon.hip reserves
// v140 .. v149): v[T:T+1] .. v[T+6:T+7] are the partial products, v[T+8:T+9] a zero-extended addend; one SGPR pair (s94:95) carries cA.
//   t0 = a0 b0;  m = a0 b1 + a1 b0 mod 2^64 with carry cA (the second cross product takes the first as its 64-bit addend: no
//   zero-extended temporaries);  lo = t0 + 2^32 lo(m) with the carry into X = hi(m) + carry and its carry X':  hi = a1 b1 + (X, X')
//   -- the product is lo + 2^64 hi + 2^96 cA, hi < 2^64 - 2^32 cA.  Then, with 2^64 = 2^32 - 1 and 2^96 = -1 (mod p):
//   T  = lo(hi) (2^32 - 1) + lo        carry c      (one multiply-add)
//   T += c (2^32 - 1)                  cannot wrap: T <= 2^64 - 2^33 after a carry
//   T -= hi(hi) + cA                   borrow b     (cA rides as the borrow-in of the subtraction; hi(hi) + cA < 2^32)
//   T += b p                           = T - (2^32 - 1) mod 2^64, cannot wrap the other way
// Measured: 1.41 -> 1.65 T products/s alone (scripts/ubench/mulmod4.hip); the one-state-per-lane leaf hash 1.65 -> 1.85 G permutations/s
// at 2^17 leaves and 2.01 -> 2.21 G at 2^21 (its S-boxes are 472 of these per permutation); nothing in the transform kernels.
}  // namespace gll
// GLL_R<i> = "v<GLL_T + i>", GLL_P<i> = "v[<GLL_T + i>:<GLL_T + i + 1>]" (i even).  The product library uses ONE window (v140 .. v149,
// poseidon.hip); the microbenchmarks' other windows come from their own table (scripts/ubench/gl_lazy_regs.inc via GLL_REGS_INC).
#if GLL_T == 140 && !defined(GLL_REGS_INC)
#define GLL_R0 "v140"
#define GLL_R1 "v141"
#define GLL_R2 "v142"
#define GLL_R3 "v143"
#define GLL_R4 "v144"
#define GLL_R5 "v145"
#define GLL_R6 "v146"
#define GLL_R7 "v147"
#define GLL_R8 "v148"
#define GLL_R9 "v149"
#define GLL_P0 "v[140:141]"
#define GLL_P2 "v[142:143]"
#define GLL_P4 "v[144:145]"
#define GLL_P6 "v[146:147]"
#define GLL_P8 "v[148:149]"
#define GLL_SA "s[94:95]"
#define GLL_SA0 "s94"
#define GLL_SA1 "s95"
#elif defined(GLL_REGS_INC)
#include GLL_REGS_INC
#else
#error "gl_lazy.hpp: GLL_T must be 140 (or name a register table with GLL_REGS_INC)"
#endif
namespace gll {

__device__ __forceinline__ uint64_t mul_nc(uint64_t a, uint64_t b) {
    uint64_t r;
    asm("v_mad_u64_u32 " GLL_P0 ", vcc, %1, %3, 0\n\t"                           /* t0 = a0 b0 */
        "v_mad_u64_u32 " GLL_P2 ", vcc, %1, %4, 0\n\t"                           /* a0 b1 */
        "v_mad_u64_u32 " GLL_P4 ", " GLL_SA ", %2, %3, " GLL_P2 "\n\t"            /* m = a0 b1 + a1 b0 mod 2^64, carry cA (worth 2^96 = -1) */
        "v_add_co_u32_e32 " GLL_R1 ", vcc, " GLL_R1 ", " GLL_R4 "\n\t"            /* lo = (lo(t0), hi(t0) + lo(m)) */
        "v_addc_co_u32_e32 " GLL_R6 ", vcc, 0, " GLL_R5 ", vcc\n\t"               /* hi(m) + carry */
        "v_addc_co_u32_e64 " GLL_R7 ", vcc, 0, 0, vcc\n\t"                        /* ... and its carry: the pair is the addend of the last product */
        "v_mad_u64_u32 " GLL_P6 ", vcc, %2, %4, " GLL_P6 "\n\t"                   /* hi = a1 b1 + that */
        "v_mad_u64_u32 " GLL_P2 ", vcc, " GLL_R6 ", -1, " GLL_P0 "\n\t"            /* T = lo(hi) (2^32 - 1) + lo, carry c */
        "v_subb_co_u32_e32 " GLL_R8 ", vcc, " GLL_R0 ", " GLL_R0 ", vcc\n\t"     /* -c (carry-in reads of VCC need no wait states) */
        "v_mov_b32_e32 " GLL_R9 ", 0\n\t"
        "v_lshl_add_u64 " GLL_P2 ", " GLL_P2 ", 0, " GLL_P8 "\n\t"                /* T += c (2^32 - 1) */
        "v_subb_co_u32_e64 " GLL_R2 ", vcc, " GLL_R2 ", " GLL_R7 ", " GLL_SA "\n\t" /* T -= hi(hi) + cA, borrow b */
        "v_subbrev_co_u32_e32 " GLL_R3 ", vcc, 0, " GLL_R3 ", vcc\n\t"
        "v_subb_co_u32_e32 " GLL_R5 ", vcc, " GLL_R0 ", " GLL_R0 ", vcc\n\t"     /* -b */
        "v_lshrrev_b32_e32 " GLL_R4 ", 31, " GLL_R5 "\n\t"
        "v_lshl_add_u64 %0, " GLL_P2 ", 0, " GLL_P4                                /* T += b p */
        : "=v"(r)
        : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32))
        : "vcc", GLL_SA0, GLL_SA1, GLL_R0, GLL_R1, GLL_R2, GLL_R3, GLL_R4, GLL_R5, GLL_R6, GLL_R7, GLL_R8, GLL_R9);
    return r;
}

// lo + 2^64 hi32  (a "96-bit" accumulator of the small-constant MDS sums) -> any u64 congruent to it:  hi32 (2^32 - 1) + lo  is ONE
// multiply-add whose carry is repaid by + (2^32 - 1) (cannot wrap again: the wrapped sum is below (2^32 - 1)^2).  Four instructions
// against the compiler's eight for gl::reduce96_nc.
__device__ __forceinline__ uint64_t reduce96_nc(uint32_t hi32, uint64_t lo) {
    uint64_t r;
    asm("v_mad_u64_u32 " GLL_P0 ", vcc, %1, -1, %2\n\t"
        "v_subb_co_u32_e32 " GLL_R8 ", vcc, " GLL_R0 ", " GLL_R0 ", vcc\n\t"
        "v_mov_b32_e32 " GLL_R9 ", 0\n\t"
        "v_lshl_add_u64 %0, " GLL_P0 ", 0, " GLL_P8
        : "=v"(r)
        : "v"(hi32), "v"(lo)
        : "vcc", GLL_R0, GLL_R1, GLL_R8, GLL_R9);
    return r;
}

// (hi, lo) -> any u64 congruent to lo + 2^64 hi: the tail of the product block on its own (nine instructions against ~15 with wait states
// in the compiler's code for gl::reduce128_nc).  For the thin hash kernel, where a lone wave per SIMD pays for every instruction it issues;
// in the one-state-per-lane kernel (three waves per SIMD) it measured no gain (round 3) and is not used there.
__device__ __forceinline__ uint64_t reduce128_nc(uint64_t hi, uint64_t lo) {
    uint64_t r;
    asm("v_mad_u64_u32 " GLL_P2 ", vcc, %1, -1, %3\n\t"
        "v_subb_co_u32_e32 " GLL_R8 ", vcc, " GLL_R0 ", " GLL_R0 ", vcc\n\t"
        "v_mov_b32_e32 " GLL_R9 ", 0\n\t"
        "v_lshl_add_u64 " GLL_P2 ", " GLL_P2 ", 0, " GLL_P8 "\n\t"
        "v_sub_co_u32_e32 " GLL_R2 ", vcc, " GLL_R2 ", %2\n\t"
        "v_subbrev_co_u32_e32 " GLL_R3 ", vcc, 0, " GLL_R3 ", vcc\n\t"
        "v_subb_co_u32_e32 " GLL_R5 ", vcc, " GLL_R0 ", " GLL_R0 ", vcc\n\t"
        "v_lshrrev_b32_e32 " GLL_R4 ", 31, " GLL_R5 "\n\t"
        "v_lshl_add_u64 %0, " GLL_P2 ", 0, " GLL_P4
        : "=v"(r)
        : "v"((uint32_t)hi), "v"((uint32_t)(hi >> 32)), "v"(lo)
        : "vcc", GLL_R0, GLL_R2, GLL_R3, GLL_R4, GLL_R5, GLL_R8, GLL_R9);
    return r;
}

// N independent products as one block of interleaved chains (tools/gen_gl_muln.py): for kernels where a lone wave per SIMD has nothing
// else to issue between two dependent instructions of a chain (the two-lanes-per-state hash kernel).  Windows v140 .. v169, s80 .. s91.
#if GLL_T == 140
#include "gl_lazy_muln.inc"
#endif

// (The same tail as a stand-alone 128 -> 64 reduction for the lazy accumulators of the partial rounds -- nine instructions against the
// compiler's ~15, 55 uses per permutation -- measured no gain: 9.13-9.23 against 8.89-9.01 ms for 2^17 leaves x 1024 columns; not kept.)
#endif  // GLL_T

}  // namespace gll
