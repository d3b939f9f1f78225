// k_blur.hip — E9 dense: the 7x7 Gaussian blur of every pyramid level, written to HBM once, together with the apron cv::ORB keeps
// around each level.
//
// Replaces the GaussianBlur(7x7, sigma 2, BORDER_REFLECT_101) of every level inside each cv::ORB::compute call
// (Feature_orb32.cpp:48: one compute per level, 36 level blurs per frame in the reference) and the copyMakeBorder apron of
// cv::ORB's pyramid buffer.  Two planes per (frame, level), both (w + 2 A) x (h + 2 A) with A = AFV_APRON = 20:
//   raw plane    the unblurred level with a BORDER_REFLECT_101 apron (the intensity-centroid disc of a keypoint near the edge);
//   blur plane   ROI = the blurred level; apron = the UNBLURRED reflected level — exactly the memory image OpenCV's
//                computeOrbDescriptors samples (only the ROI is blurred in place, a rotated test point that leaves the level
//                reads the unblurred apron).
// k_describe then has no border case at all.  This spends HBM bytes (about 4.5 MB moved per 640 x 480 frame; the pipeline uses
// < 10 % of the HBM roofline) to save vector instructions: the per-keypoint row filter and the 1024 scattered 7-tap column filters
// of the previous describe kernel are gone, and the blur touches every pixel once instead of every keypoint window (1.4 x the
// pixels of a frame at 1000 keypoints).
//
//   k_apron_copy   one wavefront per 64 dword columns x 16 rows of a plane: copies the level (plain dword loads; the ten apron dword
//                  columns gather their reflected bytes, row reflection is scalar) into BOTH planes.
//   k_blur_strips  one LANE per (row strip, 4-pixel column group) of the ROI, marching down its strip with the whole filter state in
//                  registers: per source row three aligned dwords of the raw plane (the apron supplies the reflected neighbours: no
//                  border logic), row pass exact in u16 (two v_dot4_u32_u8 per output on funnel-shifted dwords), rows paired
//                  vertically (rows 2p, 2p + 1 share a dword) so that the 7-tap column pass is four v_dot2_u32_u16, then
//                  round-half-even(S / 65536) saturated to 255 = one v_cvt_pk_u8_f32 after an exact u32 -> f32 conversion (S < 2^24
//                  whenever the result is not saturated; checked against the integer rule for all 16 842 496 possible S:
//                  tools/probes/probe_cvt_pk_u8.hip).  No LDS, no barriers, no idle lanes: lanes are a flat list over (strip, group).
// Arithmetic is bit-exact with the oracle / OpenCV's 8U separable path: integer taps [18,34,49,55,49,34,18] in both passes.
#include "afv_device.h"

typedef unsigned short ushort2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ushort2d as_us2(uint32_t v) { return __builtin_bit_cast(ushort2d, v); }

__device__ __forceinline__ const uint8_t *level_image(const LevelGeo &L, int l, const FrameSrc &src0, const uint8_t *pyr, int f, int &pitch) {
    if (l == 0) {
        pitch = src0.stride;
        return src0.base + (size_t)f * src0.frame_stride;
    }
    pitch = L.pitch;
    return pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
}

__global__ __launch_bounds__(64) void k_apron_copy(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                   uint8_t *__restrict__ blur, uint8_t *__restrict__ raw, int total_blocks, int frame_base) {
    const Geo &geo = *geo_p;
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    const int f0 = work / geo.ap_blocks, blk = work - f0 * geo.ap_blocks;
    const int f = frame_base + f0;
    int l = 0;
#pragma unroll
    for (int i = 1; i < AFV_MAX_LEVELS; ++i)
        if (i < geo.nlevels && blk >= geo.lv[i].ap_blk_base) l = i;
    const LevelGeo &L = geo.lv[l];
    const int t = blk - L.ap_blk_base;
    const int strip = t / L.ap_chunks, chunk = t - strip * L.ap_chunks;
    const int lw = L.w, lh = L.h, bh = lh + 2 * AFV_APRON;
    int pitch;
    const uint8_t *img = level_image(L, l, src0, pyr, f, pitch);
    const int c = chunk * 64 + (int)threadIdx.x;  // dword column of the plane
    if (c * 4 >= L.bpitch) return;
    const int x = c * 4 - AFV_APRON;  // level x of byte 0
    const bool interior = x >= 0 && x + 3 < lw;
    int xr[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) xr[b] = min(max(afv_reflect101(x + b, lw), 0), lw - 1);  // columns beyond the plane width: clamped, never read back
    const size_t plane = L.b_off + (size_t)f * L.b_frame_stride;
    const int oy0 = strip * AFV_AP_ROWS;
#pragma unroll 4
    for (int r = 0; r < AFV_AP_ROWS; ++r) {
        const int oy = oy0 + r;
        if (oy >= bh) break;  // uniform
        const int gy = afv_reflect101(oy - AFV_APRON, lh);  // |overshoot| <= 20 < 32 <= lh
        const uint8_t *row = img + (size_t)gy * pitch;
        uint32_t v;
        if (interior) {
            v = *reinterpret_cast<const uint32_t *>(row + x);
        } else {
            v = (uint32_t)row[xr[0]] | ((uint32_t)row[xr[1]] << 8) | ((uint32_t)row[xr[2]] << 16) | ((uint32_t)row[xr[3]] << 24);
        }
        const size_t o = plane + (size_t)oy * L.bpitch + (size_t)c * 4;
        *reinterpret_cast<uint32_t *>(raw + o) = v;
        *reinterpret_cast<uint32_t *>(blur + o) = v;
    }
}

struct RowPair {      // two consecutive row-filtered source rows of one 4-pixel group
    uint32_t v[4];    // v[k] = H[row a][x + k] | H[row b][x + k] << 16
    uint32_t ra, rb;  // the raw (centre) dwords of the two rows
};

// row pass of one source row: the dwords left of / at / right of the group -> four u16 sums
__device__ __forceinline__ void blur_row(const uint8_t *p, uint32_t (&h)[4], uint32_t &centre) {
    const uint32_t T_LO = 18u | (34u << 8) | (49u << 16) | (55u << 24), T_HI = 49u | (34u << 8) | (18u << 16);
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
    const uint32_t dp = q[-1], d = q[0], dn = q[1];
    centre = d;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // output x + k: taps over the columns x + k - 3 .. x + k + 3
        const uint32_t lo = k < 3 ? __builtin_amdgcn_alignbit(d, dp, 8 * (k + 1)) : d;
        const uint32_t hi = k < 3 ? __builtin_amdgcn_alignbit(dn, d, 8 * (k + 1)) : dn;
        h[k] = __builtin_amdgcn_udot4(hi, T_HI, __builtin_amdgcn_udot4(lo, T_LO, 0u, false), false);
    }
}

__device__ __forceinline__ void load_pair(const uint8_t *p, int bpitch, RowPair &P) {
    uint32_t ha[4], hb[4];
    blur_row(p, ha, P.ra);
    blur_row(p + bpitch, hb, P.rb);
#pragma unroll
    for (int k = 0; k < 4; ++k) P.v[k] = ha[k] | (hb[k] << 16);
}

// column pass over four row pairs = source rows 2 j .. 2 j + 7: output rows 2 j (taps on rows 2 j .. 2 j + 6) and 2 j + 1
__device__ __forceinline__ void blur_emit(const RowPair &A, const RowPair &B, const RowPair &C, const RowPair &D, uint32_t &be, uint32_t &bo) {
    const ushort2d TA0 = as_us2(18u | (34u << 16)), TA1 = as_us2(49u | (55u << 16)), TA2 = as_us2(49u | (34u << 16)), TA3 = as_us2(18u);
    const ushort2d TB0 = as_us2(18u << 16), TB1 = as_us2(34u | (49u << 16)), TB2 = as_us2(55u | (49u << 16)), TB3 = as_us2(34u | (18u << 16));
    be = 0;
    bo = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t se = __builtin_amdgcn_udot2(as_us2(A.v[k]), TA0, 0u, false);
        se = __builtin_amdgcn_udot2(as_us2(B.v[k]), TA1, se, false);
        se = __builtin_amdgcn_udot2(as_us2(C.v[k]), TA2, se, false);
        se = __builtin_amdgcn_udot2(as_us2(D.v[k]), TA3, se, false);
        uint32_t so = __builtin_amdgcn_udot2(as_us2(A.v[k]), TB0, 0u, false);
        so = __builtin_amdgcn_udot2(as_us2(B.v[k]), TB1, so, false);
        so = __builtin_amdgcn_udot2(as_us2(C.v[k]), TB2, so, false);
        so = __builtin_amdgcn_udot2(as_us2(D.v[k]), TB3, so, false);
        // round-half-even(S / 65536), saturated: exact u32 -> f32 (S < 2^24 unless saturated), exact scaling, RNE + clamp
        be = __builtin_amdgcn_cvt_pk_u8_f32((float)se * (1.0f / 65536.0f), k, be);
        bo = __builtin_amdgcn_cvt_pk_u8_f32((float)so * (1.0f / 65536.0f), k, bo);
    }
}

__global__ __launch_bounds__(64) void k_blur_strips(const Geo *__restrict__ geo_p, const uint8_t *__restrict__ raw, uint8_t *__restrict__ blur,
                                                    int total_waves, int frame_base) {
    const Geo &geo = *geo_p;
    const int work = afv_xcd_remap(blockIdx.x, total_waves);
    if (work >= total_waves) return;
    const int f0 = work / geo.bl_waves, wave = work - f0 * geo.bl_waves;
    const int f = frame_base + f0;
    int l = 0;
#pragma unroll
    for (int i = 1; i < AFV_MAX_LEVELS; ++i)
        if (i < geo.nlevels && wave >= geo.lv[i].bl_wave_base) l = i;
    const LevelGeo &L = geo.lv[l];
    const int nitems = L.bl_nstrips * L.bl_groups;
    const int item_raw = (wave - L.bl_wave_base) * 64 + (int)threadIdx.x;
    const bool live = item_raw < nitems;
    const int item = live ? item_raw : nitems - 1;  // tail lanes shadow the last item and store nothing
    const int strip = item / L.bl_groups, g = item - strip * L.bl_groups;
    const int lw = L.w, lh = L.h, bpitch = L.bpitch, SR = L.bl_sr;
    const int y0 = strip * SR, x = 4 * g;
    const size_t plane = L.b_off + (size_t)f * L.b_frame_stride;
    // source row k of the strip = plane row y0 + AFV_APRON - 3 + k; this lane's centre dword at plane column AFV_APRON + x
    const uint8_t *sp = raw + plane + (size_t)(y0 + AFV_APRON - 3) * bpitch + AFV_APRON + x;
    uint8_t *dp = blur + plane + (size_t)(y0 + AFV_APRON) * bpitch + AFV_APRON + x;
    // bytes of this dword inside the level (the last group of a row may straddle the right edge: keep the apron bytes)
    const int nb = min(lw - x, 4);
    const uint32_t m = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
    const int rows_left = live ? lh - y0 : 0;  // output rows y0 + r with r < rows_left are stored

    RowPair P0, P1, P2, P3;
    load_pair(sp, bpitch, P0);
    load_pair(sp + 2 * bpitch, bpitch, P1);
    load_pair(sp + 4 * bpitch, bpitch, P2);
    load_pair(sp + 6 * bpitch, bpitch, P3);
    sp += 8 * (size_t)bpitch;
#define BLUR_STEP(A, B, C, D, r)                                                                  \
    {                                                                                             \
        uint32_t be, bo;                                                                          \
        blur_emit(A, B, C, D, be, bo);                                                            \
        /* centres of the output rows r, r + 1: source rows r + 3 (second row of B), r + 4 (first row of C) */ \
        if ((r) < rows_left) *reinterpret_cast<uint32_t *>(dp) = (be & m) | (B.rb & ~m);          \
        if ((r) + 1 < rows_left) *reinterpret_cast<uint32_t *>(dp + bpitch) = (bo & m) | (C.ra & ~m); \
        dp += 2 * (size_t)bpitch;                                                                 \
        load_pair(sp, bpitch, A); /* source rows r + 8, r + 9 (the last step's pair is never used: rows inside the apron) */ \
        sp += 2 * (size_t)bpitch;                                                                 \
    }
    for (int r = 0; r < SR; r += 8) {
        BLUR_STEP(P0, P1, P2, P3, r)
        BLUR_STEP(P1, P2, P3, P0, r + 2)
        BLUR_STEP(P2, P3, P0, P1, r + 4)
        BLUR_STEP(P3, P0, P1, P2, r + 6)
    }
#undef BLUR_STEP
}

extern "C" void afv_launch_blur_planes(const Geo *geo, int ap_blocks, int bl_waves, const FrameSrc *src0, const uint8_t *pyr, uint8_t *blur,
                                       uint8_t *raw, int frame_base, int nframes, hipStream_t stream) {
    const int t1 = ap_blocks * nframes;
    hipLaunchKernelGGL(k_apron_copy, dim3((t1 + 7) / 8 * 8), dim3(64), 0, stream, geo, *src0, pyr, blur, raw, t1, frame_base);
    const int t2 = bl_waves * nframes;
    hipLaunchKernelGGL(k_blur_strips, dim3((t2 + 7) / 8 * 8), dim3(64), 0, stream, geo, raw, blur, t2, frame_base);
}
