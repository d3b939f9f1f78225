// calib_fetch.hip — calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE against a known byte count for the access widths
// this project uses (MI355X_MICROARCH.md §HBM: FETCH_SIZE is only calibrated for 16 B/lane streams).
//   k_read4  : coalesced 4 B/lane reads  (the LDS tile staging of k_fast_nms / k_resize_level / k_describe)
//   k_read16 : coalesced 16 B/lane reads (the guide's reference pattern: expected to report 1/2)
//   k_write4 : coalesced 4 B/lane writes
// Each kernel moves exactly BYTES bytes of a buffer larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define BYTES (1024ull << 20)
__global__ void k_read4(const uint32_t *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_read16(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_write4(uint32_t *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}
int main() {
    uint32_t *buf, *sink;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, BYTES);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read4, dim3(4096), dim3(256), 0, 0, buf, BYTES / 4, sink);
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(k_write4, dim3(4096), dim3(256), 0, 0, buf, BYTES / 4);
    }
    hipDeviceSynchronize();
    printf("moved %llu bytes per kernel\n", (unsigned long long)BYTES);
    return 0;
}
