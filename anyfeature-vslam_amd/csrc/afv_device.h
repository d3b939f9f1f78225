// afv_device.h — structures shared by the host runtime and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/afv_hip.h"

#define AFV_WAVE 64

// FAST / Harris tile: 64 x 32 output pixels per 256-thread workgroup, 4 px halo
#define FT_W 64
#ifndef FT_H
#define FT_H 32   // (round-5 experiment: -DFT_H=64, see DESIGN "measured and dropped")
#endif
#define FT_HALO 4
#define FT_LW (FT_W + 2 * FT_HALO)  // 72
#define FT_LH (FT_H + 2 * FT_HALO)  // 40
#define FT_MAXC 1024                 // candidates per tile after NMS: <= (64/2)*(32/2) = 512 (strict 3x3 maxima)

// k_describe: keypoints (= wavefronts) per workgroup
#ifndef AFV_KP_PER_BLOCK
#define AFV_KP_PER_BLOCK 4
#endif

// quadtree limits
#define QT_MAX_NODES 1536  // alive nodes <= N+3 ; supports per-level quotas up to ~1500 (nfeatures <= ~6900)

// Division of a work index by a launch-invariant divisor without the float-reciprocal sequence the compiler emits (about 20 vector
// instructions per wavefront and division; on block-uniform values this form stays on the scalar unit): q = (n * m) >> k with
// k = 26 + ceil(log2 d), m = ceil(2^k / d) < 2^27 — exact for n < 2^26 (launchers keep their work lists below that).
struct DivMagic {
    uint32_t m, k;
};
static inline DivMagic afv_div_magic(uint32_t d) {
    uint32_t lg = 0;
    while ((1u << lg) < d) ++lg;
    DivMagic r;
    r.k = 26 + lg;
    r.m = (uint32_t)((((uint64_t)1 << r.k) + d - 1) / d);
    return r;
}
static inline __host__ __device__ uint32_t afv_udiv(uint32_t n, DivMagic dm) { return (uint32_t)(((uint64_t)n * dm.m) >> dm.k); }
#define AFV_MAX_WORK (1 << 26)

struct LevelGeo {
    int w, h, pitch;        // level size, row pitch in bytes
    DivMagic dv_tiles_x;    // / tiles_x
    int tiles_x, tiles_y;   // FAST tiling
    int tile_base;          // first flat tile index of this level
    int quota;              // mnFeaturesPerLevel (FeatureExtractor.cpp:97-108)
    int cv_quota;           // cv::ORB nfeaturesPerLevel for nfeatures*10
    int cand_cap;           // candidate slots per frame at this level
    int sel_cap;            // quota + 3
    int sel_base;           // first selected slot of this level inside a frame's `sel` row
    int desc_blk_base;      // first k_describe block of this level in a frame's block list
    float scale;            // (float)pow(1.2f, l)
    float inv_scale;        // 1.f / scale
    size_t pyr_off;         // byte offset of frame 0 of this level in the pyramid buffer (level 0: unused)
    size_t pyr_frame_stride;
    size_t cand_off;        // element offset of frame 0 of this level in the candidate arrays
    size_t cand_frame_stride;
};

struct Geo {
    int nlevels, width, height;
    int total_tiles;
    DivMagic dv_total_tiles;     // / total_tiles
    DivMagic dv_desc_per_frame;  // / k_describe blocks per frame
    int sel_per_frame;      // sum of sel_cap
    int n_ini;              // DistributeOctTree: round(w/h)
    float h_x;              // (float)w / n_ini
    int fast_threshold;
    float harris_scale4;    // (1/(4*7*255))^4
    LevelGeo lv[AFV_MAX_LEVELS];
};

// level-0 image comes straight from the caller's buffer
struct FrameSrc {
    const uint8_t *base;
    int stride;
    size_t frame_stride;
};

struct SelPoint {  // quadtree survivor, level coordinates
    uint16_t x, y;
    float response;
};

// ---- k_pyramid_fused (k_pyramid.hip: the whole pyramid of a frame in one launch); plan and device image: afv_api.hip ----
// The work plan lives in ONE device buffer ("blob"), written once per geometry, laid out so that a workgroup's share is a flat copy into
// LDS: [common: one PfLevelC per level][x part of tile column 0][x part of tile column 1] ... [y part of tile row 0] ...
// x part of tile column tx = per level the region descriptor (need.lo, need.hi inclusive, own.lo, own.hi exclusive; level 0: the source
// window) and the level's coefficient-table slice with offsets already relative to the source region; y part likewise.
struct PfLevelC {
    int lds_pitch, lg_q, narrow, gpitch;      // LDS row pitch of the region, log2 dword slots per row, dword-read form allowed, pitch in memory
    unsigned long long pyr_off, fstride;      // level image of frame f = pyramid buffer + pyr_off + f * fstride
    int x_rx, x_xt, y_ry, y_yt;               // byte offsets inside the x / y part: region descriptor, table slice
};
struct PyrFuseArgs {
    const uint8_t *blob;
    int nlevels, ntx, nty;
    int sx, sy;              // bytes of one x / y part (multiples of 16)
    int off_x, off_y;        // byte offsets of the first x / y part inside the blob
    int lds_x, lds_y;        // LDS byte offsets of the parked x / y part (the common part sits at 0)
    int off_buf[2];          // LDS byte offsets of the two region buffers (level parity)
    int w0;                  // level-0 width
    int *zero_counts;        // as in ResizeTab: the candidate / queue counters of the frame range are cleared here
    int n_zero;
    int *zero_one, *zero_two;
};
// host view of the plan (tests replay it on the CPU: afv_debug_pyramid_plan)
struct PyrFusePlan {
    int nlevels, ntx, nty;
    int pitch[AFV_MAX_LEVELS], lg_q[AFV_MAX_LEVELS], narrow[AFV_MAX_LEVELS], maxh[AFV_MAX_LEVELS];
    int tabx[AFV_MAX_LEVELS], taby[AFV_MAX_LEVELS];
};

// XCD-aware block -> work-item mapping (cdna_hip_programming.md T1): hardware places linear block b on XCD b % 8, each
// XCD has its own L2.  Work items that share cache lines (neighbouring tiles of one frame) must therefore be given
// to blocks that are congruent mod 8: XCD k takes the contiguous range [k*ceil(n/8), ...) of the work list.
// Pure performance device: any placement computes the same result.
static inline __device__ int afv_xcd_remap(int b, int n) {
    const int per = (n + 7) >> 3;
    const int v = (b & 7) * per + (b >> 3);
    return v;  // may be >= n for the last XCDs when n % 8 != 0: caller skips
}

// inclusive prefix sum over the 64 lanes of a wavefront on DPP (no LDS crossbar: a __shfl_up step is a ds_bpermute, ~100+ cycles of
// latency each; the latency-bound kernels — quadtree, retainBest — run dozens of these scans back to back): 4 Hillis-Steele steps
// inside each row of 16 lanes, then the row totals are carried over with row_bcast15 (rows 1, 3) and row_bcast31 (rows 2, 3)
#ifdef __HIPCC__
static inline __device__ int afv_wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142 /*row_bcast:15*/, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143 /*row_bcast:31*/, 0xc, 0xf, false);
    return v;
}
#endif

static inline __host__ __device__ int afv_reflect101(int p, int n) {
    // BORDER_REFLECT_101 for |overshoot| < n (apron 23 px / patch halo <= 21 px, levels >= 32 px)
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return p;
}
