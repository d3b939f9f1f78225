// afv_poison.h — TEST-ONLY build aid, never part of libafv_hip.so.  `tools/poison_build.py` compiles every translation unit of the
// library a second time with `-DAFV_POISON -include afv_poison.h` into anyfeature-vslam_amd/build_exp/libafv_poison.so; the product
// sources are untouched.  In that build
//   * every hipMalloc is followed by a fill of the allocation with the byte AFV_POISON_BYTE (environment, default 0xA5), every
//     hipHostMalloc likewise: a kernel that reads device or arena memory nobody wrote in this call computes with the poison instead
//     of whatever an earlier test left in the pages;
//   * every kernel launch is preceded, on the same stream, by k_poison_lds: one workgroup per CU that owns the CU's whole LDS, fills
//     it with the byte and holds it long enough for the other workgroups of the launch to be placed on the other CUs: a kernel that
//     reads LDS it did not write reads the poison instead of the residue of the previous kernel - which in a serial run of one
//     pipeline is the SAME kernel's data of the previous call and looks valid.
// Results must not depend on the byte: the whole `-m gpu` suite is run against the oracle with 0xA5 and with 0x5A
// (tools/poison_suite.sh; DESIGN_LOG round 6).
#pragma once
#ifdef AFV_POISON
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace afv_poison {

static __global__ __launch_bounds__(1024) void k_poison_lds(unsigned word, int nwords, int hold_ticks) {
    extern __shared__ unsigned s_poison[];
    volatile unsigned *s = s_poison;
    for (int i = threadIdx.x; i < nwords; i += 1024) s[i] = word;
    __syncthreads();
    const long long t0 = wall_clock64();  // 100 MHz
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(16);
}

struct State {
    int byte = 0xA5;
    size_t lds = 0;  // bytes one workgroup fills (0: LDS poisoning unavailable)
    int grid = 0;
    bool lds_on = true;
};

static inline State &state() {
    static State st;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char *e = std::getenv("AFV_POISON_BYTE")) st.byte = (int)std::strtol(e, nullptr, 0) & 0xff;
        if (const char *e = std::getenv("AFV_POISON_LDS")) st.lds_on = std::atoi(e) != 0;
        if (!st.lds_on) return;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const size_t tries[] = {160 * 1024, 159 * 1024, 128 * 1024, 80 * 1024, 64 * 1024};
        for (size_t b : tries) {
            if (b > 64 * 1024 &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            k_poison_lds<<<1, 1024, b, 0>>>(0u, (int)(b / 4), 0);
            if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
                st.lds = b;
                st.grid = cus * (int)((160 * 1024) / b);
                break;
            }
            (void)hipGetLastError();
        }
        if (std::getenv("AFV_POISON_VERBOSE")) fprintf(stderr, "afv_poison: byte 0x%02x, LDS %zu bytes x %d workgroups\n", st.byte, st.lds, st.grid);
    });
    return st;
}

static inline void lds(hipStream_t s) {
    const State &st = state();
    if (!st.lds) return;
    const unsigned w = 0x01010101u * (unsigned)st.byte;
    k_poison_lds<<<st.grid, 1024, st.lds, s>>>(w, (int)(st.lds / 4), 800 /* 8 us */);
}

static inline hipError_t device_malloc(void **p, size_t n) {
    const hipError_t e = hipMalloc(p, n);
    if (e != hipSuccess || !n) return e;
    const hipError_t m = hipMemset(*p, state().byte, n);
    if (m != hipSuccess) return m;
    return hipDeviceSynchronize();
}

static inline hipError_t host_malloc(void **p, size_t n, unsigned flags) {
    const hipError_t e = hipHostMalloc(p, n, flags);
    if (e == hipSuccess && n && *p) std::memset(*p, state().byte, n);
    return e;
}

}  // namespace afv_poison

#define hipMalloc(p, n) afv_poison::device_malloc(reinterpret_cast<void **>(p), (n))
#define hipHostMalloc(p, n, f) afv_poison::host_malloc(reinterpret_cast<void **>(p), (n), (f))
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, ...)                  \
    do {                                                                                 \
        afv_poison::lds(stream);                                                         \
        kernel<<<(grid), (block), (lds_bytes), (stream)>>>(__VA_ARGS__);                 \
    } while (0)
#endif
