// probe: do v_mfma_i32_32x32x32_i8 and integer VALU (v_med3_i32 / v_min_i32) overlap on one SIMD?
//   mode 0: MFMA only (2 independent accumulator chains)     mode 1: VALU only (2 dependent top-4 insertion chains)
//   mode 2: both, scheduled by the compiler in ONE wave       mode 3: waves alternate (even waves MFMA only, odd waves VALU only)
//   mode 4: both in one wave, interleaved 1 MFMA : 8 VALU with sched_group_barrier
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probes/probe_mfma_valu.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int med3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }
// mode 5: VALU only, insertion as a min / max chain (4 v_min + 3 v_max, all VOP2) instead of 1 v_min + 3 v_med3
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int *out, v4i seed) {
    v4i a = seed, b = seed;
    a[0] += threadIdx.x;
    v16i acc0 = {0}, acc1 = {0};
    int k0 = 1 << 30, k1 = 1 << 30, k2 = 1 << 30, k3 = 1 << 30, j0 = 1 << 30, j1 = 1 << 30, j2 = 1 << 30, j3 = 1 << 30;
    int x = threadIdx.x * 2654435761u;
    const bool do_mfma = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && ((threadIdx.x >> 6) + blockIdx.x) % 2 == 0);
    if (MODE == 5) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int ka = x + r * 977, kb = x - r * 1319, t;
                t = max(k0, ka); k0 = min(k0, ka); ka = t;
                t = max(k1, ka); k1 = min(k1, ka); ka = t;
                t = max(k2, ka); k2 = min(k2, ka); ka = t;
                k3 = min(k3, ka);
                t = max(j0, kb); j0 = min(j0, kb); kb = t;
                t = max(j1, kb); j1 = min(j1, kb); kb = t;
                t = max(j2, kb); j2 = min(j2, kb); kb = t;
                j3 = min(j3, kb);
            }
            x = x * 1664525 + 1013904223;
        }
    }
    const bool do_valu = MODE == 1 || MODE == 2 || MODE == 4 || (MODE == 3 && ((threadIdx.x >> 6) + blockIdx.x) % 2 == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
            }
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // 32 insertions = 128 VALU, two independent chains
                const int ka = x + r * 977, kb = x - r * 1319;
                const int n3 = med3(k2, k3, ka), n2 = med3(k1, k2, ka), n1 = med3(k0, k1, ka);
                k0 = min(k0, ka); k1 = n1; k2 = n2; k3 = n3;
                const int m3 = med3(j2, j3, kb), m2 = med3(j1, j2, kb), m1 = med3(j0, j1, kb);
                j0 = min(j0, kb); j1 = m1; j2 = m2; j3 = m3;
            }
            x = x * 1664525 + 1013904223;
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[5] + k0 + k1 + k2 + k3 + j0 + j1 + j2 + j3;
}
template <int MODE>
static void run(const char *name, int waves_per_simd) {
    int *d;
    hipMalloc(&d, 4 * 256 * 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd, iters = 2000;   // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    v4i s = {1, 2, 3, 4};
    k<MODE><<<blocks, 256>>>(10, d, s);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(iters, d, s);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves_per_simd waves, each iters x (16 MFMA and/or 128 VALU)
    printf("%-44s waves/SIMD %d: %.3f ms  -> %.1f cycles per wave-iteration at 2.4 GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / iters / waves_per_simd);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("MFMA only (16 per iteration)", w);
        run<1>("VALU only (128 min/med3 per iteration)", w);
        run<2>("both in every wave", w);
        run<3>("half the waves MFMA, half VALU", w);
        run<4>("both, 1 MFMA : 8 VALU interleaved", w);
        run<5>("VALU only, 7 min/max per insertion", w);
    }
    return 0;
}
