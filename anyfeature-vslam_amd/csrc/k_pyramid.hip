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
// LDS source window (bytes per row x rows) = source span of a 64 x 32 output tile plus alignment slack; two instantiations:
// 96 x 48 for level ratios up to 1.37 (the ORB pyramid's 1.2, 2^(1/4)), 160 x 80 for ratios up to 2.37 (the reference's settings files
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
    if (afv_resize_window_ok(sw, sh, dw, dh) == 1)
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
// (a halo of ~2 px per level: 1.7x the pixels at a 32x16 top tile) instead of waiting for each other - no workgroup ever
// reads what another one wrote, so there is no hand-off at all.  Every pixel is produced by the same two expressions as in
// k_resize_level from the same source pixels, so the levels are bit-identical whichever workgroup writes them: level l is
// partitioned into OWNED rectangles (own_l(t) = [off_(l+1)[own_(l+1)(t).lo], off_(l+1)[own_(l+1)(t + 1).lo)): monotone tables make
// them a disjoint cover) and a workgroup stores exactly its rectangle, widened to whole dwords - two workgroups may store the same
// dword, with the same bytes.  Region bookkeeping (per level and tile index: needed range, owned range) is host work, done once per
// geometry (afv_api.hip: plan_pyr_fuse) and handed over IN THE KERNEL ARGUMENTS: the region descriptors arrive by scalar loads, so the
// kernel's only global round trip is the one that fetches its level-0 window and its slices of the coefficient tables.
// Latency shape: one barrier per level.  A thread owns four consecutive columns (their x coefficients stay in registers for the
// level) and walks down the rows; an output dword is evaluated directly from its 2 x 8 source bytes (horizontal blend of the two source
// rows on packed column pairs, vertical blend by v_dot2 - the separable form of k_resize_level would save a third of the LDS reads
// and cost a second barrier and two more dependent LDS round trips per level).

#define PF_T 1024

// what the level loop needs about one level, parked in LDS by the prologue: inside the loop (run-time level index) nothing is fetched
// through the scalar cache any more - with the region descriptors and the level geometry read there, every level paid two or three
// exposed scalar-load misses (kernel arguments / geometry: a new cache line per level), 10 of the kernel's 16 us
struct PfLevel {
    short4 rx, ry;                // region of this workgroup at the level (need.lo, need.hi, own.lo, own.hi)
    int lds_pitch, lg_q, off_xt, off_yt;
    int gpitch, pad;              // row pitch of the level image in memory
    unsigned long long dst_off;   // byte offset of this frame's level image in the pyramid buffer
};

static_assert(sizeof(PfLevel) == 48, "afv_api.hip reserves 48 bytes of LDS per level");
static_assert(sizeof(PyrFuseArgs) + sizeof(PyrFuseRegions) + sizeof(FrameSrc) + 32 <= 4096, "kernel argument block");

__global__ __launch_bounds__(PF_T) void k_pyramid_fused(FrameSrc src0, uint8_t *__restrict__ pyr, PyrFuseArgs A, PyrFuseRegions R, int frame_base,
                                                        int total_blocks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pf_smem[];
    const int tid = threadIdx.x;
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
    const short4 *Rx = R.r + tx, *Ry = R.r + NL * A.ntx + ty;  // level l: Rx[l * ntx], Ry[l * nty] (uniform: scalar loads from the argument block)
    PfLevel *s_lv = reinterpret_cast<PfLevel *>(pf_smem + A.off_lv);
    // Prologue, unrolled over the levels (compile-time level index: the scalar loads of all levels are independent and issued together).
    // Per level: park its descriptor in LDS; fetch this thread's entry of the level's coefficient-table slices (a thread owns at most
    // one entry per level: region width + height <= 1024).  ALL global loads of the kernel - the table entries and the level-0
    // window - are issued before the first LDS store of loaded data: one round trip.
    short2 te[AFV_MAX_LEVELS];
    short4 rxs[AFV_MAX_LEVELS], rys[AFV_MAX_LEVELS];
    bool fallback = false;  // a region wider + taller than the workgroup (not at any supported geometry): plain loop
#pragma unroll
    for (int l = 0; l < AFV_MAX_LEVELS; ++l) {
        te[l].x = 0;
        te[l].y = 0;
        rxs[l] = short4{0, 0, 0, 0};
        rys[l] = short4{0, 0, 0, 0};
        if (l < NL) {
            const short4 rx = Rx[l * A.ntx], ry = Ry[l * A.nty];
            rxs[l] = rx;
            rys[l] = ry;
            if (tid == l) {
                PfLevel v;
                v.rx = rx;
                v.ry = ry;
                v.lds_pitch = A.pitch[l];
                v.lg_q = A.lg_q[l];
                v.off_xt = A.off_xt[l];
                v.off_yt = A.off_yt[l];
                v.gpitch = A.gpitch[l];
                v.pad = 0;
                v.dst_off = (unsigned long long)A.pyr_off[l] + (unsigned long long)f * A.fstride[l];
                s_lv[l] = v;
            }
            if (l > 0) {
                const int nxe = rx.y - rx.x + 1, nye = ry.y - ry.x + 1;
                fallback = fallback || nxe + nye > PF_T;
                if (tid < nxe) {
                    if (rx.x + tid < A.lw[l]) te[l] = A.tab[A.tabx[l] + rx.x + tid];  // columns past the level's width (dword padding): offset 0, weight 0
                } else if (tid < nxe + nye) {
                    te[l] = A.tab[A.taby[l] + ry.x + (tid - nxe)];
                }
            }
        }
    }
    // ... and up to PF_W0 dwords of the level-0 window
    constexpr int PF_W0 = 8;
    uint32_t w0v[PF_W0];
    const short4 rx0 = rxs[0], ry0 = rys[0];
    const uint8_t *img = src0.base + (size_t)f * src0.frame_stride;
    const int lg0 = A.lg_q[0], ndw0 = (rx0.y - rx0.x + 4) >> 2, nrows0 = ry0.y - ry0.x + 1;
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
                    if (gx0 + b < A.lw[0]) w0v[k] |= (uint32_t)p[b] << (8 * b);
            }
        }
    }
#pragma unroll
    for (int l = 1; l < AFV_MAX_LEVELS; ++l) {
        if (l < NL) {
            const short4 rx = rxs[l], rxp = rxs[l - 1], ry = rys[l], ryp = rys[l - 1];
            short2 *xt = reinterpret_cast<short2 *>(pf_smem + A.off_xt[l]), *yt = reinterpret_cast<short2 *>(pf_smem + A.off_yt[l]);
            const int nxe = rx.y - rx.x + 1, nye = ry.y - ry.x + 1;
            short2 e = te[l];
            if (tid < nxe) {
                if (rx.x + tid < A.lw[l]) e.x = (short)(e.x - rxp.x);
                xt[tid] = e;
            } else if (tid < nxe + nye) {
                e.x = (short)(e.x - ryp.x);
                yt[tid - nxe] = e;
            }
            if (fallback) {
                for (int i = tid + PF_T; i < nxe + nye; i += PF_T) {
                    short2 g;
                    g.x = 0;
                    g.y = 0;
                    if (i < nxe) {
                        if (rx.x + i < A.lw[l]) {
                            g = A.tab[A.tabx[l] + rx.x + i];
                            g.x = (short)(g.x - rxp.x);
                        }
                        xt[i] = g;
                    } else {
                        g = A.tab[A.taby[l] + ry.x + (i - nxe)];
                        g.x = (short)(g.x - ryp.x);
                        yt[i - nxe] = g;
                    }
                }
            }
        }
    }
    {
        uint8_t *S = pf_smem + A.off_buf[0];
        const int sp = A.pitch[0];
#pragma unroll
        for (int k = 0; k < PF_W0; ++k) {
            const int r = (tid >> lg0) + k * rstep0;
            if (q0 < ndw0 && r < nrows0) *reinterpret_cast<uint32_t *>(S + r * sp + 4 * q0) = w0v[k];
        }
        for (int r = (tid >> lg0) + PF_W0 * rstep0; r < nrows0; r += rstep0) {  // windows taller than PF_W0 row groups
            if (q0 < ndw0) {
                const uint8_t *p = img + (size_t)(ry0.x + r) * src0.stride + gx0;
                uint32_t v = 0;
                if (dw_ok0) {
                    v = *reinterpret_cast<const uint32_t *>(p);
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (gx0 + b < A.lw[0]) v |= (uint32_t)p[b] << (8 * b);
                }
                *reinterpret_cast<uint32_t *>(S + r * sp + 4 * q0) = v;
            }
        }
    }
    __syncthreads();
    const int buf0 = A.off_buf[0], buf1 = A.off_buf[1];
    PfLevel vp = s_lv[0];
    for (int l = 1; l < NL; ++l) {
        const PfLevel v = s_lv[l];
        const short4 rx = v.rx, ry = v.ry, rxp = vp.rx, ryp = vp.ry;
        const uint8_t *S = pf_smem + ((l & 1) ? buf0 : buf1);
        uint8_t *D = pf_smem + ((l & 1) ? buf1 : buf0);
        const short2 *xt = reinterpret_cast<const short2 *>(pf_smem + v.off_xt), *yt = reinterpret_cast<const short2 *>(pf_smem + v.off_yt);
        const int sp = vp.lds_pitch, dp = v.lds_pitch, lgq = v.lg_q;
        const int sw = rxp.y - rxp.x + 1, sh = ryp.y - ryp.x + 1;  // source region
        const int dwp = rx.y - rx.x + 1, dh = ry.y - ry.x + 1;     // this level's region (width a multiple of 4)
        const int cq = tid & ((1 << lgq) - 1);
        if (4 * cq < dwp) {
            uint8_t *dst = pyr + v.dst_off;
            const int gpitch = v.gpitch;
            const int gx = rx.x + 4 * cq;
            const bool own_x = gx >= (rx.z & ~3) && gx < rx.w;  // owned columns [own.lo & ~3, align4(own.hi)): whole dwords
            const uint2 xe = *reinterpret_cast<const uint2 *>(xt + 4 * cq), xf = *reinterpret_cast<const uint2 *>(xt + 4 * cq + 2);  // 4 x (offset, weight)
            const int o0 = (short)(xe.x & 0xffffu), o1 = (short)(xe.y & 0xffffu), o2 = (short)(xf.x & 0xffffu), o3 = (short)(xf.y & 0xffffu);
            const int r0 = min(o0 + 1, sw - 1), r1 = min(o1 + 1, sw - 1), r2 = min(o2 + 1, sw - 1), r3 = min(o3 + 1, sw - 1);
            ushort2r WR01, WL01, WR23, WL23;
            WR01.x = (unsigned short)(xe.x >> 16);
            WR01.y = (unsigned short)(xe.y >> 16);
            WR23.x = (unsigned short)(xf.x >> 16);
            WR23.y = (unsigned short)(xf.y >> 16);
            WL01.x = (unsigned short)(256 - WR01.x);
            WL01.y = (unsigned short)(256 - WR01.y);
            WL23.x = (unsigned short)(256 - WR23.x);
            WL23.y = (unsigned short)(256 - WR23.y);
            for (int y = tid >> lgq; y < dh; y += PF_T >> lgq) {
                const short2 e = yt[y];
                const uint8_t *ru = S + e.x * sp, *rl = S + min(e.x + 1, sh - 1) * sp;
                ushort2r Lu01, Ru01, Lu23, Ru23, Ll01, Rl01, Ll23, Rl23;
                Lu01.x = ru[o0]; Lu01.y = ru[o1]; Ru01.x = ru[r0]; Ru01.y = ru[r1];
                Lu23.x = ru[o2]; Lu23.y = ru[o3]; Ru23.x = ru[r2]; Ru23.y = ru[r3];
                Ll01.x = rl[o0]; Ll01.y = rl[o1]; Rl01.x = rl[r0]; Rl01.y = rl[r1];
                Ll23.x = rl[o2]; Ll23.y = rl[o3]; Rl23.x = rl[r2]; Rl23.y = rl[r3];
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
        vp = v;
        __syncthreads();
    }
}

extern "C" void afv_launch_pyramid_fused(const FrameSrc *src0, uint8_t *pyr, const PyrFuseArgs *args, const PyrFuseRegions *regions,
                                         size_t lds_bytes, int frame_base, int nframes, hipStream_t stream) {
    const int total = args->ntx * args->nty * nframes;
    hipLaunchKernelGGL(k_pyramid_fused, dim3((total + 7) / 8 * 8), dim3(PF_T), lds_bytes, stream, *src0, pyr, *args, *regions, frame_base, total);
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
