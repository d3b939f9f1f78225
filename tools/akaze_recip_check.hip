// Checks on the device it runs on that akz_recip_ge1(d) (csrc/akz_recip.h: v_rcp_f32 + one fused Newton step) is bit-identical to the
// IEEE quotient 1.0f / d for EVERY float d in [1, 2^96): 96 exponents x 2^23 mantissas.  Prints one line per 32 exponents and
// "recip_check ok" / "recip_check FAILED"; exit code 0 / 1.  Built by __graft_entry__.build(), run by tests/test_gpu_akaze.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../anyfeature-vslam_amd/csrc/akz_recip.h"
__global__ void k_check(int e_lo, unsigned long long *bad) {
    const unsigned int mant = blockIdx.x * blockDim.x + threadIdx.x;
    const float d = __uint_as_float(((unsigned)(127 + e_lo + (int)blockIdx.y) << 23) | mant);
    const float ref = 1.0f / d;
    if (__builtin_amdgcn_rcpf(d) != ref) atomicAdd(&bad[0], 1ull);
    if (akz_recip_ge1(d) != ref) atomicAdd(&bad[1], 1ull);
}
int main() {
    unsigned long long *bad = nullptr, h[2], total = 0;
    if (hipMalloc(&bad, 16) != hipSuccess) { printf("recip_check FAILED: no device\n"); return 1; }
    for (int e0 = 0; e0 < 96; e0 += 32) {
        (void)hipMemset(bad, 0, 16);
        k_check<<<dim3((1 << 23) / 256, 32), 256>>>(e0, bad);
        if (hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost) != hipSuccess) { printf("recip_check FAILED: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        printf("exponents %d..%d (%llu values): v_rcp_f32 alone differs from 1.0f / d in %llu, akz_recip_ge1 in %llu\n", e0, e0 + 31, 32ull << 23, h[0], h[1]);
        total += h[1];
    }
    printf(total ? "recip_check FAILED\n" : "recip_check ok\n");
    return total ? 1 : 0;
}
