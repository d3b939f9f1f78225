// measurement: do half-line (64-byte) row stores of 64 x 32 u8 tiles cost against full-line (128-byte) rows of 128 x 16 tiles?
// (the store shape of k_resize_level; one read of ~1.44 x the area per output, like a 1.2 x pyramid level)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int TW, int TH>
__global__ __launch_bounds__(128) void k(const uint8_t *__restrict__ src, int spitch, size_t sframe, uint8_t *__restrict__ dst, int dw, int dh, int dpitch,
                                         size_t dframe, int nframes) {
    const int tx = (dw + TW - 1) / TW, ty = (dh + TH - 1) / TH;
    int id = blockIdx.x;
    if (id >= tx * ty * nframes) return;
    const int f = id / (tx * ty); id -= f * tx * ty;
    const int x0 = (id % tx) * TW, y0 = (id / tx) * TH;
    constexpr int CPR = TW / 4, RPP = 128 / CPR;  // dword columns per row, rows per pass
    const int cx = (threadIdx.x % CPR) * 4, ry = threadIdx.x / CPR;
    if (x0 + cx >= dw) return;
    for (int y = ry; y < TH && y0 + y < dh; y += RPP) {
        const int sy = ((y0 + y) * 6) / 5, sx = ((x0 + cx) * 6) / 5 & ~3;
        const uint32_t a = *reinterpret_cast<const uint32_t *>(src + (size_t)f * sframe + (size_t)sy * spitch + sx);
        const uint32_t b = *reinterpret_cast<const uint32_t *>(src + (size_t)f * sframe + (size_t)(sy + 1) * spitch + sx + 4);
        *reinterpret_cast<uint32_t *>(dst + (size_t)f * dframe + (size_t)(y0 + y) * dpitch + x0 + cx) = a + b;
    }
}
int main() {
    const int nf = 512, sw = 640, sh = 480, dw = 533, dh = 400;
    const size_t sframe = 640 * 482 + 256;
    uint8_t *src, *dst;
    hipMalloc(&src, sframe * nf + 4096); hipMalloc(&dst, (size_t)640 * 400 * nf + 4096);
    hipMemset(src, 1, sframe * nf);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %7.1f us per %d frames\n", name, ms * 100, nf);
    };
    for (int dpitch : {576, 640}) {
        const size_t dframe = (size_t)dpitch * dh;
        char nm[96];
        snprintf(nm, sizeof nm, "64 x 32 tiles, dst pitch %d", dpitch);
        time(nm, [&] { k<64, 32><<<((dw + 63) / 64) * ((dh + 31) / 32) * nf, 128>>>(src, 640, sframe, dst, dw, dh, dpitch, dframe, nf); });
        snprintf(nm, sizeof nm, "128 x 16 tiles, dst pitch %d", dpitch);
        time(nm, [&] { k<128, 16><<<((dw + 127) / 128) * ((dh + 15) / 16) * nf, 128>>>(src, 640, sframe, dst, dw, dh, dpitch, dframe, nf); });
        snprintf(nm, sizeof nm, "256 x 8 tiles, dst pitch %d", dpitch);
        time(nm, [&] { k<256, 8><<<((dw + 255) / 256) * ((dh + 7) / 8) * nf, 128>>>(src, 640, sframe, dst, dw, dh, dpitch, dframe, nf); });
    }
    (void)sw; (void)sh;
    return 0;
}
