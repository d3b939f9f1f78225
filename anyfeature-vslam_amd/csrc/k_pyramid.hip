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
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

#define RT_W 64
#define RT_H 32
// LDS source window (bytes per row x rows) = source span of a 64 x 32 output tile plus alignment slack; three instantiations:
// 88 x 44 for level ratios up to 1.25 (the ORB pyramid's 1.2, 2^(1/4): 9.9 KB of LDS = 8 wavefronts per SIMD), 96 x 48 up to 1.37, 160 x 80 for ratios up to 2.37 (the reference's settings files
// go up to FeatureExtractor.scaleFactor 2.0: settings/sift128_settings.yaml:7)
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

constexpr bool rs_div_exact(int sq) {  // tid / sq == (tid * ceil(65536 / sq)) >> 16 for every tid of the workgroup
    for (int t = 0; t < RT_T; ++t)
        if ((int)(((unsigned)t * (unsigned)((65536 + sq - 1) / sq)) >> 16) != t / sq) return false;
    return true;
}

template <int RS_W, int RS_H>
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
    constexpr int SQ = RS_W / 4, SG = RT_T / SQ, STG = (RS_H + SG - 1) / SG;  // 96 x 48: 24 dword columns, 5 row groups, 10 rows per thread
    static_assert(rs_div_exact(SQ), "tid / SQ by multiplication");
    const int srr = (int)(__umul24(threadIdx.x, (unsigned)((65536 + SQ - 1) / SQ)) >> 16);  // tid / SQ
    const int sq = (int)threadIdx.x - srr * SQ;
    const bool s_on = srr < SG && sq < ndw;
    const bool s_dword = sx0 + sq * 4 + 3 < spitch;  // else: the last bytes of a row whose pitch is not a multiple of 4
    uint32_t stg[STG];
    {
        // 32-bit byte offsets into the frame's level image (scalar frame base + vector offset; a level is far below 4 GB): one 24-bit
        // multiply for the thread's first row, scalar steps after it - no 64-bit vector address arithmetic (quarter / half rate)
        uint32_t po = __umul24((uint32_t)(sy0 + srr), (uint32_t)spitch) + (uint32_t)(sx0 + sq * 4);
        const uint32_t pstep = (uint32_t)(SG * spitch);
#pragma unroll
        for (int k = 0; k < STG; ++k) {
            stg[k] = 0;
            if (s_on && srr + k * SG < nrows) {
                if (s_dword) {
                    stg[k] = *reinterpret_cast<const uint32_t *>(s + po);
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (sx0 + sq * 4 + b < sw) stg[k] |= (uint32_t)s[po + b] << (8 * b);
                }
            }
            po += pstep;
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
    uint8_t *const dstf = dst + (size_t)f * dframe;  // scalar
    uint32_t doff = __umul24((uint32_t)(y0 + ry), (uint32_t)dpitch) + (uint32_t)(x0 + cx);
    const uint32_t dstep = (uint32_t)((RT_T / 16) * dpitch);
#pragma unroll
    for (int part = 0; part < RT_H / (RT_T / 16); ++part, doff += dstep) {
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
        // the output byte is byte 2 of each sum (<= 255 << 16 + 65535): the four of them are gathered with two byte permutes + one OR
        const uint32_t q0 = __builtin_amdgcn_udot2(p0, WY, 32768u, false), q1 = __builtin_amdgcn_udot2(p1, WY, 32768u, false);
        const uint32_t q2 = __builtin_amdgcn_udot2(p2, WY, 32768u, false), q3 = __builtin_amdgcn_udot2(p3, WY, 32768u, false);
        const uint32_t out4 = __builtin_amdgcn_perm(q1, q0, 0x0c0c0602u) | __builtin_amdgcn_perm(q3, q2, 0x06020c0cu);
        // pitch is a multiple of 64: the dword store never leaves the row (columns past nx hold filtered padding, never read)
        *reinterpret_cast<uint32_t *>(dstf + doff) = out4;
    }
}

// which instantiation holds the source span of a 64x32 output tile (plus alignment slack and the +1 tap): 1 = 96 x 48, 2 = 160 x 80, 0 = none
extern "C" int afv_resize_window_ok(int sw, int sh, int dw, int dh) {
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    if ((RT_W * fx + 8 <= 96) && (RT_H * fy + 3 <= 48)) return 1;
    if ((RT_W * fx + 8 <= 160) && (RT_H * fy + 3 <= 80)) return 2;
    return 0;
}

extern "C" void afv_launch_resize(const uint8_t *src, int sw, int sh, int spitch, size_t sframe, uint8_t *dst, int dw,
                                  int dh, int dpitch, size_t dframe, const short2 *xt, const short2 *yt, int frame_base,
                                  int nframes, int *zero_counts, int n_zero, int *zero_one, hipStream_t stream) {
    const int tx = (dw + RT_W - 1) / RT_W, per_frame = tx * ((dh + RT_H - 1) / RT_H);
    const int total = per_frame * nframes;
    dim3 grid((total + 7) / 8 * 8);
    ResizeTab tab{xt, yt, zero_counts, n_zero, zero_one, afv_div_magic((uint32_t)per_frame), afv_div_magic((uint32_t)tx)};
    const double fx = (double)sw / dw, fy = (double)sh / dh;
    if ((RT_W * fx + 8 <= 88) && (RT_H * fy + 3 <= 44))  // ratios up to 1.25 (the ORB pyramid's 1.2): 9.9 KB of LDS, 16 tiles = 8 wavefronts per SIMD in flight
        hipLaunchKernelGGL((k_resize_level<88, 44>), grid, dim3(RT_T), 0, stream, src, sw, sh, spitch, sframe, dst, dw, dh, dpitch, dframe, tab,
                           total, frame_base);
    else if (afv_resize_window_ok(sw, sh, dw, dh) == 1)
        hipLaunchKernelGGL((k_resize_level<96, 48>), grid, dim3(RT_T), 0, stream, src, sw, sh, spitch, sframe, dst, dw, dh, dpitch, dframe, tab,
                           total, frame_base);
    else
        hipLaunchKernelGGL((k_resize_level<160, 80>), grid, dim3(RT_T), 0, stream, src, sw, sh, spitch, sframe, dst, dw, dh, dpitch, dframe, tab,
                           total, frame_base);
}

// ---------------- the whole pyramid in ONE launch (the per-frame plugin call: Frame.cc:186 extracts one frame at a time) ----------------
// Seven dependent k_resize_level launches cost ~4.8 us each on one frame (34 us of a 110 us extraction) although the arithmetic is a
// few hundred nanoseconds of the chip: what is paid is seven kernel ramps and seven cache write-back / invalidate boundaries.  Here
// one workgroup owns one tile of the TOP level and computes, level by level, everything that tile depends on: its regions of levels
// 1 .. L live in LDS (ping-pong), only level 0 is read from memory.  Neighbouring workgroups recompute the overlap of their regions
// (a halo of ~2 px per level) instead of waiting for each other - no workgroup ever reads what another one wrote, so there is no
// hand-off at all.  Every pixel is produced by the same two expressions as in k_resize_level from the same source pixels, so the
// levels are bit-identical whichever workgroup writes them: level l is partitioned into OWNED rectangles (own_l(t) =
// [off_(l+1)[own_(l+1)(t).lo], off_(l+1)[own_(l+1)(t + 1).lo)): monotone tables make them a disjoint cover) and a workgroup stores
// exactly its rectangle, widened to whole dwords - two workgroups may store the same dword, with the same bytes.
// Latency shape (round 4, measured with -DAFV_PF_STATS; the first form took 16 us):
//   * the plan is a device blob laid out for a flat copy (afv_device.h): a workgroup's prologue is TWO scalar loads (its level-0
//     window) and ONE round of vector loads (its share of the blob + the window), nothing else is fetched through the scalar cache -
//     per-level scalar loads (kernel arguments, geometry) each cost an exposed miss and serialise behind LDS traffic (one counter);
//   * one barrier per level, and it waits for LDS only: __syncthreads() also drains the level's global stores (0.5 us per level);
//   * a thread owns four consecutive columns (their x coefficients stay in registers for the level) and walks down the rows; an
//     output dword is evaluated directly from its 2 x 8 source bytes, fetched as three aligned dwords per source row;
//   * 16 x 8 top tiles: 204 workgroups for a 640 x 480 frame - the level loop is bound by the vector ALU of the CUs it runs on.

#define PF_T 1024
#define PF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(PF_T) void k_pyramid_fused(FrameSrc src0, uint8_t *__restrict__ pyr, PyrFuseArgs A, int frame_base, int total_blocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pf_smem[];
    const int tid = threadIdx.x;
#ifdef AFV_PF_STATS
    long long st[12];
    int sti = 0;
    st[sti++] = wall_clock64();
#endif
    if (blockIdx.x == 0) {
        if (A.zero_counts)
            for (int i = tid; i < A.n_zero; i += PF_T) A.zero_counts[i] = 0;
        if (tid == 0 && A.zero_one) *A.zero_one = 0;
        if (tid == 1 && A.zero_two) *A.zero_two = 0;
    }
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);  // a frame's tiles share level-0 cache lines: keep them on one XCD
    if (work >= total_blocks) return;
    const int per_frame = A.ntx * A.nty;
    const int fl = work / per_frame, t = work - fl * per_frame, ty = t / A.ntx, tx = t - ty * A.ntx;
    const int f = frame_base + fl;
    const int NL = A.nlevels;
    const uint8_t *bx = A.blob + A.off_x + (size_t)tx * A.sx, *by = A.blob + A.off_y + (size_t)ty * A.sy;
    // the level-0 window of this tile (the first 8 bytes of its x / y part): uniform addresses -> two scalar loads
    const short4 rx0 = *reinterpret_cast<const short4 *>(bx), ry0 = *reinterpret_cast<const short4 *>(by);
    // flat copy of [common | x part | y part] into LDS, <= 2 dwords per thread, and up to PF_W0 dwords of the window: every global load of
    // the kernel is issued here, before the first LDS store
    constexpr int NC = (int)(sizeof(PfLevelC) * AFV_MAX_LEVELS / 4);
    const int nX = A.sx >> 2, nY = A.sy >> 2;
    uint32_t cpv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * PF_T;
        cpv[k] = 0;
        if (i < NC) cpv[k] = reinterpret_cast<const uint32_t *>(A.blob)[i];
        else if (i < NC + nX) cpv[k] = reinterpret_cast<const uint32_t *>(bx)[i - NC];
        else if (i < NC + nX + nY) cpv[k] = reinterpret_cast<const uint32_t *>(by)[i - NC - nX];
    }
    constexpr int PF_W0 = 8;
    uint32_t w0v[PF_W0];
    const uint8_t *img = src0.base + (size_t)f * src0.frame_stride;
    const PfLevelC *s_C = reinterpret_cast<const PfLevelC *>(pf_smem);
    const int pitch0 = (rx0.y - rx0.x + 4) & ~3;  // level 0 keeps the window's own dword-rounded width as its LDS pitch
    int lg0 = 0;
    while ((4 << lg0) < pitch0) ++lg0;           // uniform: a handful of scalar steps
    const int ndw0 = pitch0 >> 2, nrows0 = ry0.y - ry0.x + 1;
    const int q0 = tid & ((1 << lg0) - 1), gx0 = rx0.x + 4 * q0, rstep0 = PF_T >> lg0;
    const bool dw_ok0 = gx0 + 3 < src0.stride;
#pragma unroll
    for (int k = 0; k < PF_W0; ++k) {
        const int r = (tid >> lg0) + k * rstep0;
        w0v[k] = 0;
        if (q0 < ndw0 && r < nrows0) {
            const uint8_t *p = img + (size_t)(ry0.x + r) * src0.stride + gx0;
            if (dw_ok0) {
                w0v[k] = *reinterpret_cast<const uint32_t *>(p);
            } else {
                for (int b = 0; b < 4; ++b)
                    if (gx0 + b < A.w0) w0v[k] |= (uint32_t)p[b] << (8 * b);
            }
        }
    }
#ifdef AFV_PF_STATS
    st[sti++] = wall_clock64();
#endif
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = tid + k * PF_T;
        if (i < NC) reinterpret_cast<uint32_t *>(pf_smem)[i] = cpv[k];
        else if (i < NC + nX) reinterpret_cast<uint32_t *>(pf_smem + A.lds_x)[i - NC] = cpv[k];
        else if (i < NC + nX + nY) reinterpret_cast<uint32_t *>(pf_smem + A.lds_y)[i - NC - nX] = cpv[k];
    }
    {
        uint8_t *S = pf_smem + A.off_buf[0];
#pragma unroll
        for (int k = 0; k < PF_W0; ++k) {
            const int r = (tid >> lg0) + k * rstep0;
            if (q0 < ndw0 && r < nrows0) *reinterpret_cast<uint32_t *>(S + r * pitch0 + 4 * q0) = w0v[k];
        }
        for (int r = (tid >> lg0) + PF_W0 * rstep0; r < nrows0; r += rstep0) {  // windows taller than PF_W0 row groups
            if (q0 < ndw0) {
                const uint8_t *p = img + (size_t)(ry0.x + r) * src0.stride + gx0;
                uint32_t v = 0;
                if (dw_ok0) {
                    v = *reinterpret_cast<const uint32_t *>(p);
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (gx0 + b < A.w0) v |= (uint32_t)p[b] << (8 * b);
                }
                *reinterpret_cast<uint32_t *>(S + r * pitch0 + 4 * q0) = v;
            }
        }
    }
    const int buf0 = A.off_buf[0], buf1 = A.off_buf[1];
    const uint8_t *s_X = pf_smem + A.lds_x, *s_Y = pf_smem + A.lds_y;
    PF_LDS_BARRIER();
#ifdef AFV_PF_STATS
    st[sti++] = wall_clock64();
#endif
    short4 rxp = rx0, ryp = ry0;
    int sp = pitch0;
    for (int l = 1; l < NL; ++l) {
        const PfLevelC v = s_C[l];
        const short4 rx = *reinterpret_cast<const short4 *>(s_X + v.x_rx), ry = *reinterpret_cast<const short4 *>(s_Y + v.y_ry);
        const uint8_t *S = pf_smem + ((l & 1) ? buf0 : buf1);
        uint8_t *D = pf_smem + ((l & 1) ? buf1 : buf0);
        const short2 *xt = reinterpret_cast<const short2 *>(s_X + v.x_xt), *yt = reinterpret_cast<const short2 *>(s_Y + v.y_yt);
        const int dp = v.lds_pitch, lgq = v.lg_q;
        const int sw = rxp.y - rxp.x + 1, sh = ryp.y - ryp.x + 1;  // source region
        const int dwp = rx.y - rx.x + 1, dh = ry.y - ry.x + 1;     // this level's region (width a multiple of 4)
        const int cq = tid & ((1 << lgq) - 1);
        if (4 * cq < dwp && (tid >> lgq) < dh) {
            uint8_t *dst = pyr + v.pyr_off + (unsigned long long)f * v.fstride;
            const int gpitch = v.gpitch;
            const int gx = rx.x + 4 * cq;
            const bool own_x = gx >= (rx.z & ~3) && gx < rx.w;  // owned columns [own.lo & ~3, align4(own.hi)): whole dwords
            const uint2 xe = *reinterpret_cast<const uint2 *>(xt + 4 * cq), xf = *reinterpret_cast<const uint2 *>(xt + 4 * cq + 2);  // 4 x (offset, weight)
            const int o0 = (short)(xe.x & 0xffffu), o1 = (short)(xe.y & 0xffffu), o2 = (short)(xf.x & 0xffffu), o3 = (short)(xf.y & 0xffffu);
            ushort2r WR01, WL01, WR23, WL23;
            WR01.x = (unsigned short)(xe.x >> 16);
            WR01.y = (unsigned short)(xe.y >> 16);
            WR23.x = (unsigned short)(xf.x >> 16);
            WR23.y = (unsigned short)(xf.y >> 16);
            WL01.x = (unsigned short)(256 - WR01.x);
            WL01.y = (unsigned short)(256 - WR01.y);
            WL23.x = (unsigned short)(256 - WR23.x);
            WL23.y = (unsigned short)(256 - WR23.y);
            // The eight source bytes of a row (left and right tap of four neighbouring columns) lie within the 8 bytes that start at the
            // first column's left tap when the level ratio is below 2 (o3 + 1 - o0 <= 7): THREE aligned dwords per source row shifted
            // into an 8-byte window by two v_alignbyte, then one byte permute per packed operand with selectors that are constant for
            // the level.  A right tap past the region's last column has weight 0: whatever byte the window holds there is multiplied
            // by 0.  Wider spans (ratio >= 2 ...) take the byte-read form.
            const bool narrow = v.narrow != 0;  // host: every group of four columns of this level spans <= 8 bytes
            const int base = o0 & ~3, bsh = o0 & 3;
            const uint32_t k1 = (uint32_t)(o1 - o0), k2 = (uint32_t)(o2 - o0), k3 = (uint32_t)(o3 - o0);
            const uint32_t selL01 = 0x0c000c00u | (k1 << 16), selR01 = 0x0c000c01u | ((k1 + 1) << 16);            // (byte 0, byte k1), (byte 1, byte k1 + 1)
            const uint32_t selL23 = 0x0c000c00u | k2 | (k3 << 16), selR23 = 0x0c000c00u | (k2 + 1) | ((k3 + 1) << 16);
            const int r0 = min(o0 + 1, sw - 1), r1 = min(o1 + 1, sw - 1), r2 = min(o2 + 1, sw - 1), r3 = min(o3 + 1, sw - 1);
            for (int y = tid >> lgq; y < dh; y += PF_T >> lgq) {
                const short2 e = yt[y];
                const uint8_t *ru = S + e.x * sp, *rl = S + min(e.x + 1, sh - 1) * sp;
                ushort2r Lu01, Ru01, Lu23, Ru23, Ll01, Rl01, Ll23, Rl23;
                if (narrow) {
                    const uint32_t *du = reinterpret_cast<const uint32_t *>(ru + base), *dl = reinterpret_cast<const uint32_t *>(rl + base);
                    const uint32_t u0 = du[0], u1 = du[1], u2 = du[2], l0 = dl[0], l1 = dl[1], l2 = dl[2];
                    const uint32_t wu0 = __builtin_amdgcn_alignbyte(u1, u0, bsh), wu1 = __builtin_amdgcn_alignbyte(u2, u1, bsh);
                    const uint32_t wl0 = __builtin_amdgcn_alignbyte(l1, l0, bsh), wl1 = __builtin_amdgcn_alignbyte(l2, l1, bsh);
                    Lu01 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wu1, wu0, selL01));
                    Ru01 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wu1, wu0, selR01));
                    Lu23 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wu1, wu0, selL23));
                    Ru23 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wu1, wu0, selR23));
                    Ll01 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wl1, wl0, selL01));
                    Rl01 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wl1, wl0, selR01));
                    Ll23 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wl1, wl0, selL23));
                    Rl23 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(wl1, wl0, selR23));
                } else {
                    Lu01.x = ru[o0]; Lu01.y = ru[o1]; Ru01.x = ru[r0]; Ru01.y = ru[r1];
                    Lu23.x = ru[o2]; Lu23.y = ru[o3]; Ru23.x = ru[r2]; Ru23.y = ru[r3];
                    Ll01.x = rl[o0]; Ll01.y = rl[o1]; Rl01.x = rl[r0]; Rl01.y = rl[r1];
                    Ll23.x = rl[o2]; Ll23.y = rl[o3]; Rl23.x = rl[r2]; Rl23.y = rl[r3];
                }
                const uint32_t u01 = __builtin_bit_cast(uint32_t, (ushort2r)(WL01 * Lu01 + WR01 * Ru01)), u23 = __builtin_bit_cast(uint32_t, (ushort2r)(WL23 * Lu23 + WR23 * Ru23));
                const uint32_t l01 = __builtin_bit_cast(uint32_t, (ushort2r)(WL01 * Ll01 + WR01 * Rl01)), l23 = __builtin_bit_cast(uint32_t, (ushort2r)(WL23 * Ll23 + WR23 * Rl23));
                ushort2r WY;
                WY.x = (unsigned short)(256 - e.y);
                WY.y = (unsigned short)e.y;
                const ushort2r p0 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l01, u01, 0x05040100u));
                const ushort2r p1 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l01, u01, 0x07060302u));
                const ushort2r p2 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l23, u23, 0x05040100u));
                const ushort2r p3 = __builtin_bit_cast(ushort2r, __builtin_amdgcn_perm(l23, u23, 0x07060302u));
                const uint32_t q0_ = __builtin_amdgcn_udot2(p0, WY, 32768u, false) >> 16, q1_ = __builtin_amdgcn_udot2(p1, WY, 32768u, false) >> 16;
                const uint32_t q2_ = __builtin_amdgcn_udot2(p2, WY, 32768u, false) >> 16, q3_ = __builtin_amdgcn_udot2(p3, WY, 32768u, false) >> 16;
                const uint32_t o = q0_ | (q1_ << 8) | (q2_ << 16) | (q3_ << 24);
                *reinterpret_cast<uint32_t *>(D + y * dp + 4 * cq) = o;
                const int gy = ry.x + y;
                if (own_x && gy >= ry.z && gy < ry.w) *reinterpret_cast<uint32_t *>(dst + (size_t)gy * gpitch + gx) = o;
            }
        }
        rxp = rx;
        ryp = ry;
        sp = dp;
        PF_LDS_BARRIER();
#ifdef AFV_PF_STATS
        if (sti < 11) st[sti++] = wall_clock64();
#endif
    }
#ifdef AFV_PF_STATS
    __builtin_amdgcn_s_waitcnt(0);
    st[sti++] = wall_clock64();
    if (tid == 0 && (work == 0 || work == total_blocks - 1)) {
        printf("pyramid_fused block %d (work %d): x10 ns since start:", (int)blockIdx.x, work);
        for (int i = 1; i < sti; ++i) printf(" %lld", st[i] - st[0]);
        printf("\n");
    }
#endif
}

extern "C" void afv_launch_pyramid_fused(const FrameSrc *src0, uint8_t *pyr, const PyrFuseArgs *args, size_t lds_bytes, int frame_base, int nframes,
                                         hipStream_t stream) {
    const int total = args->ntx * args->nty * nframes;
    hipLaunchKernelGGL(k_pyramid_fused, dim3((total + 7) / 8 * 8), dim3(PF_T), lds_bytes, stream, *src0, pyr, *args, frame_base, total);
}

extern "C" int afv_pyramid_fused_prepare(size_t lds_bytes) {
    if (lds_bytes > 160 * 1024) return 0;
    if (lds_bytes > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_pyramid_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return 1;
}
