// measurement: what a 1-read : 3-write float stream reaches on this device (the shape of k_akz_dhess), aligned full lines vs rows of
// 52 floats at 208-byte steps (the strip kernel's store shape)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_aligned(const float *__restrict__ in, float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = in[i];
        a[i] = v * 2.0f; b[i] = v + 1.0f; c[i] = v * v;
    }
}
// wave per strip of OW columns x 128 rows of a w x h plane, lanes >= OW idle (like the strip kernel's out_lane)
template <int OW>
__global__ __launch_bounds__(256) void k_strips(const float *__restrict__ in, float *__restrict__ a, float *__restrict__ b, float *__restrict__ c, int w, int h, int nframes) {
    const int lane = threadIdx.x & 63;
    const int nstr = (w + OW - 1) / OW, nband = (h + 127) / 128;
    int id = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (id >= nstr * nband * nframes) return;
    const int f = id / (nstr * nband); id -= f * nstr * nband;
    const int band = id / nstr, x = (id - band * nstr) * OW + lane, y0 = band * 128;
    if (lane >= OW || x >= w) return;
    const size_t fo = (size_t)f * w * h;
    for (int y = y0; y < min(y0 + 128, h); ++y) {
        const size_t o = fo + (size_t)y * w + x;
        const float v = in[o];
        a[o] = v * 2.0f; b[o] = v + 1.0f; c[o] = v * v;
    }
}
int main() {
    const int w = 1280, h = 720, nf = 64; const size_t n = (size_t)w * h * nf;
    float *in, *a, *b, *c;
    hipMalloc(&in, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
    hipMemset(in, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f us  %.2f TB/s (16 B per pixel)\n", name, ms * 100, n * 16.0 / (ms / 10 * 1e-3) / 1e12);
    };
    time("aligned grid-stride", [&] { k_aligned<<<256 * 16, 256>>>(in, a, b, c, n); });
    auto strips = [&](auto tag) { constexpr int OW = decltype(tag)::value; const int ns = ((w + OW - 1) / OW) * ((h + 127) / 128) * nf; k_strips<OW><<<(ns + 3) / 4, 256>>>(in, a, b, c, w, h, nf); };
    time("strips of 64 columns", [&] { strips(std::integral_constant<int, 64>{}); });
    time("strips of 56 columns", [&] { strips(std::integral_constant<int, 56>{}); });
    time("strips of 52 columns", [&] { strips(std::integral_constant<int, 52>{}); });
    time("strips of 48 columns", [&] { strips(std::integral_constant<int, 48>{}); });
    return 0;
}
