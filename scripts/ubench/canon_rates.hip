// What a conditional correction costs on gfx950 (round 4): gll::canon -- x >= p ? x - p : x -- as carry chain + two v_cndmask on VCC (the
// shipped form until round 4's last day) against carry chain + mask (v_subb x, x) + v_bfi, inside a dependent chain of butterfly-like
// work (a lazy add and a lazy subtract before it), 8 independent values per lane, 8 waves per SIMD, every CU busy.
// build: hipcc -O3 --offload-arch=gfx950 scripts/ubench/canon_rates.hip -o scripts/ubench/bin/canon_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint64_t pack(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

template <int V>
__device__ __forceinline__ uint64_t canon(uint64_t x) {
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), tl, th;
    if (V == 0) {
        asm("v_add_co_u32_e32 %2, vcc, -1, %0\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %1, vcc\n\t"
            "v_cndmask_b32_e32 %0, %0, %2, vcc\n\t"
            "v_cndmask_b32_e32 %1, %1, %3, vcc"
            : "+v"(xl), "+v"(xh), "=&v"(tl), "=&v"(th) : : "vcc");
    } else if (V == 1) {
        asm("v_add_co_u32_e32 %2, vcc, -1, %0\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %1, vcc\n\t"
            "v_subb_co_u32_e32 %2, vcc, %0, %0, vcc\n\t"      /* -carry: all ones when x >= p */
            "v_add_co_u32_e32 %0, vcc, %2, %0\n\t"            /* x + (2^32 - 1) = x - p mod 2^64 then, else x */
            "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
            : "+v"(xl), "+v"(xh), "=&v"(tl), "=&v"(th) : : "vcc");
    } else if (V == 2) {
        uint32_t m;
        asm("v_add_co_u32_e32 %2, vcc, -1, %0\n\t"
            "v_addc_co_u32_e32 %3, vcc, 0, %1, vcc\n\t"
            "v_subb_co_u32_e32 %4, vcc, %0, %0, vcc\n\t"
            "v_bfi_b32 %0, %4, %2, %0\n\t"
            "v_bfi_b32 %1, %4, %3, %1"
            : "+v"(xl), "+v"(xh), "=&v"(tl), "=&v"(th), "=&v"(m) : : "vcc");
    } else {
        asm("v_add_co_u32_e64 %2, s[20:21], -1, %0\n\t"
            "v_addc_co_u32_e64 %3, s[20:21], 0, %1, s[20:21]\n\t"
            "v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\t"
            "v_cndmask_b32_e64 %1, %1, %3, s[20:21]"
            : "+v"(xl), "+v"(xh), "=&v"(tl), "=&v"(th) : : "s20", "s21");
    }
    return pack(xl, xh);
}

template <int V>
__global__ void __launch_bounds__(256) k(uint64_t* out, int iters) {
    uint64_t x[8];
    for (int i = 0; i < 8; i++) x[i] = 0xFFFFFFFF00000000ull + threadIdx.x * 977u + i * 31u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) x[i] = canon<V>(x[i] + 0x9E3779B97F4A7C15ull * (uint64_t)(it + 1));
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int V>
void run(const char* name) {
    uint64_t* d;
    const int blocks = 256 * 8, threads = 256, iters = 2048;
    (void)hipMalloc(&d, (size_t)blocks * threads * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        k<V><<<blocks, threads>>>(d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    uint64_t h = 0;
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms   (checksum %016llx)\n", name, ms, (unsigned long long)h);
    (void)hipFree(d);
}

int main() {
    run<0>("canon: carry chain + 2 v_cndmask_e32 (vcc)");
    run<1>("canon: carry chain + mask + carry chain");
    run<2>("canon: carry chain + mask + 2 v_bfi");
    run<3>("canon: e64 carry chain + 2 v_cndmask_e64");
    return 0;
}
