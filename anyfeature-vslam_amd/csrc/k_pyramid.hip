// k_pyramid.hip — E2: pyramid level l from level l-1, bilinear INTER_LINEAR_EXACT (8.8 fixed point).
//
// Replaces the resize(prevImg, currImg, sz, 0, 0, INTER_LINEAR_EXACT) chain inside cv::ORB (called from
// Feature_orb32.cpp:34 and :48).  The per-column / per-row (offset, weight) tables are computed once per
// geometry on the host exactly as OpenCV's interpolation_linear<uchar>::getCoeffs does (IEEE double), so the
// kernel is pure integer arithmetic: horizontal pass in 8.8, vertical pass (+2^15)>>16.
//
// One workgroup (RT_T threads) produces a 64x32 output tile: the source window it needs (<= 80 x 42 pixels for the 1.2 pyramid) is
// staged in LDS with coalesced dword loads issued together with the tile's slice of the coefficient tables — ONE global round
// trip per workgroup.  The blend is separable, exactly as OpenCV evaluates it:
//   horizontal  h[r][c] = (256 - wx_c) * S[r][o_c] + wx_c * S[r][o_c + 1]      (8.8 fixed point, <= 65280: exact in u16)
//   vertical    out = (h[ya][c] * (256 - wy) + h[ya + 1][c] * wy + 2^15) >> 16
// so every source row is filtered ONCE (a 1.2x level reuses a filtered row for ~1.6 output rows): the horizontal pass works on
// packed column pairs (v_pk_mul_lo_u16 + v_pk_mad_u16 per pair and row), the vertical pass is one v_dot2_u32_u16 per pixel on
// the (upper, lower) pair with the 2^15 rounding term as its accumulator.
#include "afv_device.h"

#define RT_W 64
#define RT_H 32
#define RS_W 96   // LDS source window pitch (bytes); source span of 64 outputs at scale <= 1.4 plus alignment slack
#define RS_H 48
#define RT_T 128  // threads per workgroup: the kernel waits on its global loads most of the time, so what counts is how many tiles a
                  // CU has in flight (LDS 10.9 KB, 2 waves per tile -> 14 tiles per CU instead of 8 with 256 threads)

typedef unsigned short ushort2r __attribute__((ext_vector_type(2)));

struct ResizeTab {
    const short2 *xt;  // [dw]  {src offset, weight of the right tap}
    const short2 *yt;  // [dh]
    // the first launch of a frame range also clears that range's per-level candidate counters and its Harris queue counter (the FAST
    // tiles that add to them run after the whole pyramid): two fill kernels per chunk less on the stream
    int *zero_counts;
    int n_zero;
    int *zero_one;
    DivMagic dv_tiles, dv_tiles_x;  // / (tiles per frame), / tiles_x
};

__global__ __launch_bounds__(RT_T) void k_resize_level(const uint8_t *__restrict__ src, int sw, int sh, int spitch,
                                                      size_t sframe, uint8_t *__restrict__ dst, int dw, int dh,
                                                      int dpitch, size_t dframe, ResizeTab tab, int total_blocks, int frame_base) {
    __shared__ __attribute__((aligned(16))) uint8_t win[RS_H * RS_W];
    __shared__ __attribute__((aligned(16))) uint32_t hrow[RS_H * (RT_W / 2)];  // horizontally filtered rows, column pairs (u16, u16)
    __shared__ short2 s_xt[RT_W];
    __shared__ short2 s_yt[RT_H];
    // XCD-aware placement: whole frames per XCD (neighbouring tiles share source cache lines)
    const int tiles_x = (dw + RT_W - 1) / RT_W, tiles_y = (dh + RT_H - 1) / RT_H;
    if (tab.zero_counts && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < tab.n_zero; i += RT_T) tab.zero_counts[i] = 0;
        if (threadIdx.x == 0 && tab.zero_one) *tab.zero_one = 0;
    }
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    const int fl = (int)afv_udiv((uint32_t)work, tab.dv_tiles), tt = work - fl * (tiles_x * tiles_y);
    const int f = frame_base + fl;
    const int tty = (int)afv_udiv((uint32_t)tt, tab.dv_tiles_x);
    const int x0 = (tt - tty * tiles_x) * RT_W, y0 = tty * RT_H;
    const int nx = min(RT_W, dw - x0), ny = min(RT_H, dh - y0);
    const uint8_t *s = src + (size_t)f * sframe;
    // source window: rows [sy0, sy1], dword-aligned columns [sx0, ...).  Offsets are monotone in the tables.
    const int sx_first = tab.xt[x0].x, sx_last = min(tab.xt[x0 + nx - 1].x + 1, sw - 1);
    const int sy0 = tab.yt[y0].x, sy1 = min(tab.yt[y0 + ny - 1].x + 1, sh - 1);
    const int sx0 = sx_first & ~3;
    const int ndw = (sx_last - sx0) / 4 + 1, nrows = sy1 - sy0 + 1;  // <= RS_W/4, <= RS_H (host checks the scale)
    // all loads of the window first (one global round trip), then the LDS writes.  Thread (q, rr) = (tid % 24, tid / 24) owns the
    // dword column q of the rows rr, rr + 5, ... (120 of the 128 threads): no division by a run-time width, addresses by increments
    constexpr int SQ = RS_W / 4, SG = RT_T / SQ, STG = (RS_H + SG - 1) / SG;  // 24 dword columns, 5 row groups, 10 rows per thread
    const int srr = (int)(__umul24(threadIdx.x, 2731u) >> 16);  // tid / 24 (exact for tid < 128)
    const int sq = (int)threadIdx.x - srr * SQ;
    const bool s_on = srr < SG && sq < ndw;
    const bool s_dword = sx0 + sq * 4 + 3 < spitch;  // else: the last bytes of a row whose pitch is not a multiple of 4
    uint32_t stg[STG];
    {
        const uint8_t *p = s + (size_t)(sy0 + srr) * spitch + sx0 + sq * 4;
#pragma unroll
        for (int k = 0; k < STG; ++k) {
            stg[k] = 0;
            if (s_on && srr + k * SG < nrows) {
                if (s_dword) {
                    stg[k] = *reinterpret_cast<const uint32_t *>(p);
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (sx0 + sq * 4 + b < sw) stg[k] |= (uint32_t)p[b] << (8 * b);
                }
            }
            p += (size_t)SG * spitch;
        }
    }
    if (s_on) {
#pragma unroll
        for (int k = 0; k < STG; ++k)
            if (srr + k * SG < nrows) *reinterpret_cast<uint32_t *>(&win[(srr + k * SG) * RS_W + sq * 4]) = stg[k];
    }
    if (threadIdx.x < RT_W) {  // columns past the image edge: offset 0, weight 0 (their outputs are never stored)
        short2 e;
        e.x = 0;
        e.y = 0;
        if (threadIdx.x < nx) {
            e = tab.xt[x0 + threadIdx.x];
            e.x = (short)(e.x - sx0);
        }
        s_xt[threadIdx.x] = e;
    } else if (threadIdx.x >= 64 && threadIdx.x < 64 + ny) {
        short2 e = tab.yt[y0 + threadIdx.x - 64];
        e.x = (short)(e.x - sy0);
        s_yt[threadIdx.x - 64] = e;
    }
    __syncthreads();
    const int xr_max = sx_last - sx0, yr_max = sy1 - sy0;
    // ---- horizontal pass: thread = column pair (cp) x row group (rg); rows rg, rg + 8, ... of the staged window ----
    {
        const int cp = threadIdx.x & 31, rg = threadIdx.x >> 5;
        const short2 xa = s_xt[2 * cp], xb = s_xt[2 * cp + 1];
        const int a0 = xa.x, a1 = min(xa.x + 1, xr_max), b0 = xb.x, b1 = min(xb.x + 1, xr_max);
        ushort2r WR, WL;
        WR.x = (unsigned short)xa.y;
        WR.y = (unsigned short)xb.y;
        WL.x = (unsigned short)(256 - xa.y);
        WL.y = (unsigned short)(256 - xb.y);
        for (int r = rg; r < nrows; r += RT_T / 32) {
            const uint8_t *row = &win[r * RS_W];
            ushort2r L, R;
            L.x = row[a0];
            L.y = row[b0];
            R.x = row[a1];
            R.y = row[b1];
            const ushort2r hv = WL * L + WR * R;  // <= 256 * 255: no overflow
            hrow[r * (RT_W / 2) + cp] = __builtin_bit_cast(uint32_t, hv);
        }
    }
    __syncthreads();
    // ---- vertical pass: thread -> 4 consecutive columns, rows ry, ry + RT_T / 16, ... ----
    const int cx = (threadIdx.x & 15) * 4, ry = threadIdx.x >> 4;
    if (cx >= nx) return;
#pragma unroll
    for (int part = 0; part < RT_H / (RT_T / 16); ++part) {
        const int y = ry + part * (RT_T / 16);
        if (y >= ny) break;
        const short2 yt = s_yt[y];
        const uint32_t *h0 = &hrow[yt.x * (RT_W / 2) + (cx >> 1)];
        const uint32_t *h1 = &hrow[min(yt.x + 1, yr_max) * (RT_W / 2) + (cx >> 1)];
        ushort2r WY;
        WY.x = (unsigned short)(256 - yt.y);
        WY.y = (unsigned short)yt.y;
        const uint2 u = *reinterpret_cast<const uint2 *>(h0), l = *reinterpret_cast<const uint2 *>(h1);  // 8-byte aligned: cx % 4 == 0
        // (upper, lower) pairs of the four columns
        const ushort2r p0 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l.x, u.x, 0x05040100u));
        const ushort2r p1 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l.x, u.x, 0x07060302u));
        const ushort2r p2 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l.y, u.y, 0x05040100u));
        const ushort2r p3 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l.y, u.y, 0x07060302u));
        const uint32_t o0 = __builtin_amdgcn_udot2(p0, WY, 32768u, false) >> 16, o1 = __builtin_amdgcn_udot2(p1, WY, 32768u, false) >> 16;
        const uint32_t o2 = __builtin_amdgcn_udot2(p2, WY, 32768u, false) >> 16, o3 = __builtin_amdgcn_udot2(p3, WY, 32768u, false) >> 16;
        // pitch is a multiple of 64: the dword store never leaves the row (columns past nx hold filtered padding, never read)
        *reinterpret_cast<uint32_t *>(dst + (size_t)f * dframe + (size_t)(y0 + y) * dpitch + x0 + cx) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
    }
}

extern "C" int afv_resize_window_ok(int sw, int sh, int dw, int dh) {
    // the LDS window must hold the source span of a 64x32 output tile (plus alignment slack and the +1 tap)
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    return (RT_W * fx + 8 <= RS_W) && (RT_H * fy + 3 <= RS_H);
}

extern "C" void afv_launch_resize(const uint8_t *src, int sw, int sh, int spitch, size_t sframe, uint8_t *dst, int dw,
                                  int dh, int dpitch, size_t dframe, const short2 *xt, const short2 *yt, int frame_base,
                                  int nframes, int *zero_counts, int n_zero, int *zero_one, hipStream_t stream) {
    const int tx = (dw + RT_W - 1) / RT_W, per_frame = tx * ((dh + RT_H - 1) / RT_H);
    const int total = per_frame * nframes;
    dim3 grid((total + 7) / 8 * 8);
    ResizeTab tab{xt, yt, zero_counts, n_zero, zero_one, afv_div_magic((uint32_t)per_frame), afv_div_magic((uint32_t)tx)};
    hipLaunchKernelGGL(k_resize_level, grid, dim3(RT_T), 0, stream, src, sw, sh, spitch, sframe, dst, dw, dh, dpitch,
                       dframe, tab, total, frame_base);
}
