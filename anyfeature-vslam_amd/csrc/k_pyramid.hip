// k_pyramid.hip — E2: pyramid level l from level l-1, bilinear INTER_LINEAR_EXACT (8.8 fixed point).
//
// Replaces the resize(prevImg, currImg, sz, 0, 0, INTER_LINEAR_EXACT) chain inside cv::ORB (called from
// Feature_orb32.cpp:34 and :48).  The per-column / per-row (offset, weight) tables are computed once per
// geometry on the host exactly as OpenCV's interpolation_linear<uchar>::getCoeffs does (IEEE double), so the
// kernel is pure integer arithmetic: horizontal pass in 8.8, vertical pass (+2^15)>>16.
//
// HBM/latency-bound stage.  One 256-thread workgroup produces a 64x32 output tile: the source window it needs
// (<= 80 x 42 pixels for the 1.2 pyramid) is staged in LDS with coalesced dword loads issued together with the
// tile's slice of the coefficient tables — ONE global round trip per workgroup — then every thread blends
// 4 x 2 output pixels from LDS and stores two dwords.
#include "afv_device.h"

#define RT_W 64
#define RT_H 32
#define RS_W 96   // LDS source window pitch (bytes); source span of 64 outputs at scale <= 1.4 plus alignment slack
#define RS_H 48

struct ResizeTab {
    const short2 *xt;  // [dw]  {src offset, weight of the right tap}
    const short2 *yt;  // [dh]
};

__global__ __launch_bounds__(256) void k_resize_level(const uint8_t *__restrict__ src, int sw, int sh, int spitch,
                                                      size_t sframe, uint8_t *__restrict__ dst, int dw, int dh,
                                                      int dpitch, size_t dframe, ResizeTab tab, int total_blocks, int frame_base) {
    __shared__ __attribute__((aligned(16))) uint8_t win[RS_H * RS_W];
    __shared__ short2 s_xt[RT_W];
    __shared__ short2 s_yt[RT_H];
    // XCD-aware placement: whole frames per XCD (neighbouring tiles share source cache lines)
    const int tiles_x = (dw + RT_W - 1) / RT_W, tiles_y = (dh + RT_H - 1) / RT_H;
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    const int fl = work / (tiles_x * tiles_y), tt = work - fl * (tiles_x * tiles_y);
    const int f = frame_base + fl;
    const int x0 = (tt % tiles_x) * RT_W, y0 = (tt / tiles_x) * RT_H;
    const int nx = min(RT_W, dw - x0), ny = min(RT_H, dh - y0);
    const uint8_t *s = src + (size_t)f * sframe;
    // source window: rows [sy0, sy1], dword-aligned columns [sx0, ...).  Offsets are monotone in the tables.
    const int sx_first = tab.xt[x0].x, sx_last = min(tab.xt[x0 + nx - 1].x + 1, sw - 1);
    const int sy0 = tab.yt[y0].x, sy1 = min(tab.yt[y0 + ny - 1].x + 1, sh - 1);
    const int sx0 = sx_first & ~3;
    const int ndw = (sx_last - sx0) / 4 + 1, nrows = sy1 - sy0 + 1;  // <= RS_W/4, <= RS_H (host checks the scale)
    for (int i = threadIdx.x; i < nrows * ndw; i += 256) {
        const int r = i / ndw, q = i - r * ndw;
        const uint8_t *p = s + (size_t)(sy0 + r) * spitch + sx0 + q * 4;
        uint32_t v;
        if (sx0 + q * 4 + 3 < spitch) {
            v = *reinterpret_cast<const uint32_t *>(p);
        } else {
            v = 0;
            for (int k = 0; k < 4; ++k)
                if (sx0 + q * 4 + k < sw) v |= (uint32_t)p[k] << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(&win[r * RS_W + q * 4]) = v;
    }
    if (threadIdx.x < nx) {
        short2 e = tab.xt[x0 + threadIdx.x];
        e.x = (short)(e.x - sx0);
        s_xt[threadIdx.x] = e;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + ny) {
        short2 e = tab.yt[y0 + threadIdx.x - 64];
        e.x = (short)(e.x - sy0);
        s_yt[threadIdx.x - 64] = e;
    }
    __syncthreads();
    // thread -> 4 consecutive columns, rows ry and ry + 16
    const int cx = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;
    if (cx >= nx) return;
    const int xr_max = sx_last - sx0, yr_max = sy1 - sy0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int y = ry + half * 16;
        if (y >= ny) break;
        const short2 yt = s_yt[y];
        const uint8_t *r0 = &win[yt.x * RS_W];
        const uint8_t *r1 = &win[min(yt.x + 1, yr_max) * RS_W];
        const uint32_t cy = (uint32_t)yt.y;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (cx + k < nx) {
                const short2 xt = s_xt[cx + k];
                const int o0 = xt.x, o1 = min(xt.x + 1, xr_max);
                const uint32_t wx = (uint32_t)xt.y;
                const uint32_t h0 = (256u - wx) * r0[o0] + wx * r0[o1];
                const uint32_t h1 = (256u - wx) * r1[o0] + wx * r1[o1];
                out |= ((h0 * (256u - cy) + h1 * cy + 32768u) >> 16) << (8 * k);
            }
        }
        // pitch is a multiple of 64: the dword store never leaves the row
        *reinterpret_cast<uint32_t *>(dst + (size_t)f * dframe + (size_t)(y0 + y) * dpitch + x0 + cx) = out;
    }
}

extern "C" int afv_resize_window_ok(int sw, int sh, int dw, int dh) {
    // the LDS window must hold the source span of a 64x32 output tile (plus alignment slack and the +1 tap)
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    return (RT_W * fx + 8 <= RS_W) && (RT_H * fy + 3 <= RS_H);
}

extern "C" void afv_launch_resize(const uint8_t *src, int sw, int sh, int spitch, size_t sframe, uint8_t *dst, int dw,
                                  int dh, int dpitch, size_t dframe, const short2 *xt, const short2 *yt, int frame_base,
                                  int nframes, hipStream_t stream) {
    const int total = ((dw + RT_W - 1) / RT_W) * ((dh + RT_H - 1) / RT_H) * nframes;
    dim3 grid((total + 7) / 8 * 8);
    ResizeTab tab{xt, yt};
    hipLaunchKernelGGL(k_resize_level, grid, dim3(256), 0, stream, src, sw, sh, spitch, sframe, dst, dw, dh, dpitch,
                       dframe, tab, total, frame_base);
}
