// k_pyramid.hip — E2: pyramid level l from level l-1, bilinear INTER_LINEAR_EXACT (8.8 fixed point).
//
// Replaces the resize(prevImg, currImg, sz, 0, 0, INTER_LINEAR_EXACT) chain inside cv::ORB (called from
// Feature_orb32.cpp:34 and :48).  The per-column / per-row (offset, weight) tables are computed once per
// geometry on the host exactly as OpenCV's interpolation_linear<uchar>::getCoeffs does (IEEE double), so the
// kernel is pure integer arithmetic: horizontal pass in 8.8, vertical pass (+2^15)>>16.
//
// HBM-bound stage: each thread produces 4 horizontally adjacent output pixels and stores them as one dword;
// a wavefront therefore writes 256 contiguous bytes per row and reads two ~307-byte source row spans.
#include "afv_device.h"

// tables for one destination level: xo[w], xc[w], yo[h], yc[h] (int16 each, packed as {ofs, c1})
struct ResizeTab {
    const short2 *xt;  // [dw]  {src offset, weight of the right tap}
    const short2 *yt;  // [dh]
};

__global__ __launch_bounds__(256) void k_resize_level(const uint8_t *__restrict__ src, int sw, int sh, int spitch,
                                                      size_t sframe, uint8_t *__restrict__ dst, int dw, int dh,
                                                      int dpitch, size_t dframe, ResizeTab tab) {
    const int f = blockIdx.z;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (y >= dh || x4 >= dw) return;
    const uint8_t *s = src + (size_t)f * sframe;
    const short2 yt = tab.yt[y];
    const int y0 = yt.x, y1 = min(yt.x + 1, sh - 1);
    const uint32_t cy = (uint32_t)yt.y;
    const uint8_t *r0 = s + (size_t)y0 * spitch;
    const uint8_t *r1 = s + (size_t)y1 * spitch;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x4 + k;
        if (x < dw) {
            const short2 xt = tab.xt[x];
            const int o0 = xt.x, o1 = min(xt.x + 1, sw - 1);
            const uint32_t cx = (uint32_t)xt.y;
            const uint32_t h0 = (256u - cx) * r0[o0] + cx * r0[o1];
            const uint32_t h1 = (256u - cx) * r1[o0] + cx * r1[o1];
            const uint32_t v = (h0 * (256u - cy) + h1 * cy + 32768u) >> 16;
            out |= v << (8 * k);
        }
    }
    uint8_t *d = dst + (size_t)f * dframe + (size_t)y * dpitch + x4;
    *reinterpret_cast<uint32_t *>(d) = out;  // pitch is a multiple of 64: the dword store never leaves the row
}

extern "C" void afv_launch_resize(const uint8_t *src, int sw, int sh, int spitch, size_t sframe, uint8_t *dst, int dw,
                                  int dh, int dpitch, size_t dframe, const short2 *xt, const short2 *yt, int nframes,
                                  hipStream_t stream) {
    dim3 grid((dw + 255) / 256, (dh + 3) / 4, nframes);
    ResizeTab tab{xt, yt};
    hipLaunchKernelGGL(k_resize_level, grid, dim3(256), 0, stream, src, sw, sh, spitch, sframe, dst, dw, dh, dpitch,
                       dframe, tab);
}
