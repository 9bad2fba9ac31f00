// Issue cost of the integer instruction FORMS the Goldilocks arithmetic is made of (gfx950): the same operation in its two-operand
// (VOP2, carry in VCC) and three-operand (VOP3, carry / mask in an SGPR pair) encodings, 64-bit adds, compares, selects, and the
// multiply-add.  Eight independent destination registers per block, blocks repeated back to back, 8 waves per SIMD, every CU busy.
// Reports the time per wave-instruction per SIMD RELATIVE to v_add_u32 (the clock under load is not 2.4 GHz, so ratios, not cycles).
// Round 5: every block also stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around its loop into a buffer of its own
// (MI355X_MICROARCH.md, DVFS give-back item 6): the median quotient is the clock the chip HOLDS under that instruction stream, so the
// table gives cycles per wave-instruction as well as nanoseconds -- the calibration DESIGN.md section 9 / bench.py's `valu` object use.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/enc_rates.hip -o scripts/ubench/bin/enc_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define S(x) #x
// each OPn(i) is one instruction on registers v[8+i] (dst / src) with sources v[16+i], v[24+i]; 64-bit forms use pairs 32+2i
#define ADD_U32(i) "v_add_u32_e32 v" S(8) "+" #i ", v16, v8\n"

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint64_t* stamps) {
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t x[8], y[8];
    uint64_t z[8];
    for (int i = 0; i < 8; i++) {
        x[i] = threadIdx.x * 77u + i;
        y[i] = threadIdx.x * 31u + 7 * i + 1;
        z[i] = ((uint64_t)x[i] << 32) | y[i];
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 1) asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0" : "+v"(x[i]) : "v"(y[i]) : "vcc");
                if (OP == 2) asm volatile("v_add_co_u32_e64 %0, s[20:21], %1, %0" : "+v"(x[i]) : "v"(y[i]) : "s20", "s21");
                if (OP == 3) asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %0, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");
                if (OP == 4) asm volatile("v_addc_co_u32_e64 %0, s[20:21], %1, %0, s[20:21]" : "+v"(x[i]) : "v"(y[i]) : "s20", "s21");
                if (OP == 5) asm volatile("v_cndmask_b32_e32 %0, %1, %0, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");
                if (OP == 6) asm volatile("v_cndmask_b32_e64 %0, %1, %0, s[20:21]" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 7) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(z[i]) : "v"(z[(i + 1) & 7]));
                if (OP == 8) asm volatile("v_cmp_lt_u64_e64 s[20:21], %0, %1" : : "v"(z[i]), "v"(z[(i + 1) & 7]) : "s20", "s21");
                if (OP == 9) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(x[i]), "v"(y[i]) : "vcc");
                if (OP == 10) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(z[i]) : "v"(x[i]), "v"(y[i]) : "s20", "s21");
                if (OP == 11) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
                if (OP == 12) asm volatile("v_add3_u32 %0, %1, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 13) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 14) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 15) asm volatile("v_mad_u32_u24 %0, %1, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 16) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(z[i]));
                if (OP == 17) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 18) asm volatile("v_xor_b32_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 19) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 20) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 21) asm volatile("v_mad_i32_i24 %0, %1, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 22) asm volatile("v_mul_u32_u24_e32 %0, %1, %0" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 23) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 24) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(z[i]) : "v"(z[(i + 1) & 7]));
                if (OP == 25) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(z[i]) : "v"(z[(i + 1) & 7]));
                if (OP == 26) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(z[i]) : "v"(z[(i + 1) & 7]));
                if (OP == 27) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 28) asm volatile("v_dot4_u32_u8 %0, %1, %0, %0" : "+v"(x[i]) : "v"(y[i]));
                if (OP == 29) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(z[i]) : "v"(x[i]), "v"(y[i]) : "vcc");
            }
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + (uint32_t)z[i] + (uint32_t)(z[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {           // a buffer no other code reads
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static double base_ms = 0;
template <int OP>
void run(const char* name) {
    uint32_t* d;
    const int blocks = 256 * 8, threads = 256, iters = 2048;
    (void)hipMalloc(&d, blocks * threads * 4);
    uint64_t* st;
    (void)hipMalloc(&st, blocks * 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        k<OP><<<blocks, threads>>>(d, iters, st);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    if (OP == 0) base_ms = ms;
    const double winst = (double)blocks * (threads / 64) * iters * 32;
    static uint64_t h[256 * 8 * 2];
    (void)hipMemcpy(h, st, blocks * 16, hipMemcpyDeviceToHost);
    double q[256 * 8];
    for (int b = 0; b < blocks; b++) q[b] = h[2 * b + 1] ? (double)h[2 * b] / (double)h[2 * b + 1] * 0.1 : 0.0;   // GHz
    for (int i = 1; i < blocks; i++) {       // insertion sort: the median block
        double v = q[i];
        int j = i - 1;
        for (; j >= 0 && q[j] > v; j--) q[j + 1] = q[j];
        q[j + 1] = v;
    }
    const double ghz = q[blocks / 2], ns = ms * 1e6 / (winst / 1024);
    printf("%-28s %8.3f ms  %6.2f x v_add_u32   %.2f ns per wave-instruction per SIMD   in-kernel clock %.3f GHz   %.2f cycles\n", name, ms,
           ms / base_ms, ns, ghz, ns * ghz);
    (void)hipFree(d);
    (void)hipFree(st);
}

int main() {
    run<0>("v_add_u32_e32");
    run<18>("v_xor_b32_e32");
    run<11>("v_mov_b32_e32");
    run<1>("v_add_co_u32_e32 (vcc)");
    run<2>("v_add_co_u32_e64 (sgpr)");
    run<3>("v_addc_co_u32_e32 (vcc)");
    run<4>("v_addc_co_u32_e64 (sgpr)");
    run<5>("v_cndmask_b32_e32 (vcc)");
    run<6>("v_cndmask_b32_e64 (sgpr)");
    run<9>("v_cmp_lt_u32_e32");
    run<8>("v_cmp_lt_u64_e64");
    run<7>("v_lshl_add_u64");
    run<16>("v_lshlrev_b64");
    run<12>("v_add3_u32");
    run<19>("v_and_or_b32");
    run<17>("v_alignbit_b32");
    run<10>("v_mad_u64_u32 (sgpr)");
    run<29>("v_mad_u64_u32 (vcc)");
    run<13>("v_mul_lo_u32");
    run<14>("v_mul_hi_u32");
    run<15>("v_mad_u32_u24");
    run<21>("v_mad_i32_i24");
    run<22>("v_mul_u32_u24_e32");
    run<20>("v_pk_add_u16");
    run<27>("v_pk_mul_lo_u16");
    run<28>("v_dot4_u32_u8");
    run<23>("v_fma_f32");
    run<25>("v_pk_fma_f32");
    run<24>("v_fma_f64");
    run<26>("v_mul_f64");
    return 0;
}
