// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 in THIS repository's access pattern:
// 8 bytes per lane (global_load_dwordx2), column-strided like poseidon_leaves (64 consecutive u64 per wave per column).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ void read8(const uint64_t* __restrict__ p, size_t n_leaves, uint32_t ncols, uint64_t* out) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    uint64_t acc = 0;
    for (uint32_t c = 0; c < ncols; c++) acc ^= p[(size_t)c * n_leaves + j];
    if (acc == 0x123456789) out[0] = acc;
}
__global__ void write8(uint64_t* __restrict__ p, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) p[j] = j;
}
int main() {
    const size_t n_leaves = 1 << 17;
    const uint32_t ncols = 1024;  // 1 GiB read
    uint64_t *p, *o;
    hipMalloc(&p, n_leaves * ncols * 8);
    hipMalloc(&o, 8);
    hipMemset(p, 1, n_leaves * ncols * 8);
    hipDeviceSynchronize();
    read8<<<n_leaves / 256, 256>>>(p, n_leaves, ncols, o);
    write8<<<(n_leaves * ncols) / 256, 256>>>(p, n_leaves * ncols);
    hipDeviceSynchronize();
    printf("read8: %zu bytes read; write8: %zu bytes written\n", n_leaves * ncols * 8, n_leaves * ncols * 8);
    return 0;
}
