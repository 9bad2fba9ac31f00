// Do back-to-back carry chains need wait states on gfx950?  (the compiler inserts s_nop after v_*_co writes)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void k(const uint64_t* a, const uint64_t* b, uint64_t* c, uint64_t* d, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t a0 = (uint32_t)a[i], a1 = (uint32_t)(a[i] >> 32), b0 = (uint32_t)b[i], b1 = (uint32_t)(b[i] >> 32);
    uint32_t r0, r1, r2;
    // 96-bit: (a1:a0) + (b1:b0) then + carry into r2, three dependent carry ops, no nops
    asm volatile(
        "v_add_co_u32 %0, vcc, %3, %5\n\t"
        "v_addc_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "v_addc_co_u32 %2, vcc, 0, 0, vcc\n\t"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "vcc");
    c[i] = ((uint64_t)r1 << 32) | r0;
    d[i] = r2;
    // mad with sgpr carry-out consumed immediately
    uint64_t m; uint32_t cy;
    asm volatile(
        "v_mad_u64_u32 %0, vcc, %2, %3, %4\n\t"
        "v_addc_co_u32 %1, vcc, 0, 0, vcc\n\t"
        : "=&v"(m), "=&v"(cy)
        : "v"(a1), "v"(b1), "v"(a[i])
        : "vcc");
    c[n + i] = m;
    d[n + i] = cy;
}
int main() {
    const int n = 1 << 20;
    uint64_t *a, *b, *c, *d;
    hipMallocManaged(&a, n * 8); hipMallocManaged(&b, n * 8); hipMallocManaged(&c, 2 * n * 8); hipMallocManaged(&d, 2 * n * 8);
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (i % 3 == 0) ? ~0ull - (s & 0xff) : s;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17; b[i] = (i % 5 == 0) ? ~0ull : s;
    }
    k<<<n / 256, 256>>>(a, b, c, d, n);
    hipDeviceSynchronize();
    long bad1 = 0, bad2 = 0;
    for (int i = 0; i < n; i++) {
        unsigned __int128 t = (unsigned __int128)a[i] + b[i];
        if (c[i] != (uint64_t)t || d[i] != (uint64_t)(t >> 64)) bad1++;
        unsigned __int128 u = (unsigned __int128)(a[i] >> 32) * (b[i] >> 32) + a[i];
        if (c[n + i] != (uint64_t)u || d[n + i] != (uint64_t)(u >> 64)) bad2++;
    }
    printf("carry chain mismatches: %ld ; mad carry-out mismatches: %ld (of %d)\n", bad1, bad2, n);
    return 0;
}
