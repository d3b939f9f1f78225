// k_fast.hip — E3 + E5 fused: FAST-9/16 score, 3x3 non-max suppression and Harris response.
//
// Replaces, for every pyramid level of every frame of the batch, the per-level work of cv::ORB::detect
// (Feature_orb32.cpp:34): FastFeatureDetector(20, true) (OpenCV fast.cpp FAST_t<16> + cornerScore<16>) and
// HarrisResponses(blockSize 7, k 0.04) (OpenCV orb.cpp).  One 256-thread workgroup owns a 64x32 tile:
//   1. the tile plus a 4 px halo is staged in LDS with coalesced dword loads (reflect-101 at the image edge —
//      FAST never reads it, Harris does, exactly like cv::ORB's 23 px apron);
//   2. every pixel of the 66x34 inner ring gets its corner score = largest threshold for which it is still a
//      9-arc corner (sliding-window min/max over the 16 ring differences, packed 2 x i16 per VGPR);
//   3. strict 3x3 maxima are compacted into an LDS list (<= 512 per tile);
//   4. the list is processed densely: integer Harris sums a,b,c over the 7x7 block from LDS, one float
//      expression for the response;
//   5. one global atomic per tile reserves output slots in the (frame, level) candidate array.
// Candidates leave the kernel unordered; everything downstream is order-independent (ties are broken by the
// raster index), see DESIGN.md "canonical order".
#include "afv_device.h"

typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2v pk(int lo, int hi) {
    short2v r;
    r.x = (short)lo;
    r.y = (short)hi;
    return r;
}
__device__ __forceinline__ short2v pkmin(short2v a, short2v b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ short2v pkmax(short2v a, short2v b) { return __builtin_elementwise_max(a, b); }


// score = max over the 16 arcs of 9 contiguous ring pixels of min(v - ring) [dark arc] and of min(ring - v)
// [bright arc]; corner iff score > threshold; cornerScore<16> returns score - 1.
__device__ __forceinline__ int fast_score(const uint8_t *c, int threshold) {
    const int v = c[0];
    short2v d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        // compile-time ring offsets (the loop is fully unrolled)
        constexpr int dx[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
        constexpr int dy[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};
        const int diff = v - (int)c[dy[k] * FT_LW + dx[k]];
        d[k] = pk(diff, -diff);
    }
    short2v m2[16], m4[16], m8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m2[k] = pkmin(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m4[k] = pkmin(m2[k], m2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) m8[k] = pkmin(m4[k], m4[(k + 4) & 15]);
    short2v best = pkmin(m8[0], d[8]);
#pragma unroll
    for (int k = 1; k < 16; ++k) best = pkmax(best, pkmin(m8[k], d[(k + 8) & 15]));
    const int s = max((int)best.x, (int)best.y);
    return s > threshold ? s - 1 : 0;
}

__global__ __launch_bounds__(256) void k_fast_harris(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                     uint32_t *__restrict__ cand_packed, float *__restrict__ cand_resp,
                                                     int *__restrict__ cand_count) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[FT_LH * FT_LW];
    __shared__ uint8_t sc[(FT_H + 2) * 68];
    __shared__ uint32_t list[512];
    __shared__ int list_n, out_base;

    const Geo &geo = *geo_p;
    const int f = blockIdx.y;
    int l = 0;
#pragma unroll
    for (int i = 1; i < AFV_MAX_LEVELS; ++i)
        if (i < geo.nlevels && (int)blockIdx.x >= geo.lv[i].tile_base) l = i;
    const LevelGeo &L = geo.lv[l];
    const int t = blockIdx.x - L.tile_base;
    const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
    const int gx0 = tx * FT_W - FT_HALO, gy0 = ty * FT_H - FT_HALO;
    const int lw = L.w, lh = L.h;

    const uint8_t *img;
    int pitch;
    if (l == 0) {
        img = src0.base + (size_t)f * src0.frame_stride;
        pitch = src0.stride;
    } else {
        img = pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
        pitch = L.pitch;
    }
    if (threadIdx.x == 0) list_n = 0;

    // 1. stage 72x40 bytes: 18 dwords per row
    for (int i = threadIdx.x; i < FT_LH * (FT_LW / 4); i += 256) {
        const int ry = i / (FT_LW / 4), rq = i - ry * (FT_LW / 4);
        int gy = gy0 + ry;
        const int gx = gx0 + rq * 4;
        gy = min(max(afv_reflect101(gy, lh), 0), lh - 1);
        uint32_t v;
        const uint8_t *row = img + (size_t)gy * pitch;
        if (gx >= 0 && gx + 3 < lw) {
            v = *reinterpret_cast<const uint32_t *>(row + gx);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x = min(max(afv_reflect101(gx + k, lw), 0), lw - 1);
                v |= (uint32_t)row[x] << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t *>(&tile[ry * FT_LW + rq * 4]) = v;
    }
    __syncthreads();

    // 2. scores on the 66x34 ring-extended tile
    const int thr = geo.fast_threshold;
    for (int i = threadIdx.x; i < (FT_W + 2) * (FT_H + 2); i += 256) {
        const int sy = i / (FT_W + 2), sx = i - sy * (FT_W + 2);  // position (sx-1, sy-1) relative to the tile
        const int gx = gx0 + FT_HALO - 1 + sx, gy = gy0 + FT_HALO - 1 + sy;
        int s = 0;
        if (gx >= 3 && gx < lw - 3 && gy >= 3 && gy < lh - 3) s = fast_score(&tile[(sy + 3) * FT_LW + (sx + 3)], thr);
        sc[sy * 68 + sx] = (uint8_t)s;
    }
    __syncthreads();

    // 3. strict 3x3 maxima -> LDS list
    for (int i = threadIdx.x; i < FT_W * FT_H; i += 256) {
        const int py = i >> 6, px = i & 63;
        const uint8_t *p = &sc[(py + 1) * 68 + (px + 1)];
        const int s = p[0];
        if (s != 0 && s > p[-1] && s > p[1] && s > p[-69] && s > p[-68] && s > p[-67] && s > p[67] && s > p[68] &&
            s > p[69]) {
            const int slot = atomicAdd(&list_n, 1);
            list[slot] = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)s << 16);
        }
    }
    __syncthreads();
    const int n = list_n;
    if (n == 0) return;
    if (threadIdx.x == 0) out_base = atomicAdd(&cand_count[f * AFV_MAX_LEVELS + l], n);
    __syncthreads();

    // 4. Harris response for the compacted candidates
    const size_t obase = L.cand_off + (size_t)f * L.cand_frame_stride + (size_t)out_base;
    for (int ci = threadIdx.x; ci < n; ci += 256) {
        const uint32_t e = list[ci];
        const int px = e & 255, py = (e >> 8) & 255, s = e >> 16;
        const uint8_t *c = &tile[(py + FT_HALO) * FT_LW + (px + FT_HALO)];
        int a = 0, b = 0, cc = 0;
        // rows r-1, r, r+1 of 9 pixels slide down over the 7 block rows
        int r0[9], r1[9], r2[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            r0[j] = c[-4 * FT_LW + (j - 4)];
            r1[j] = c[-3 * FT_LW + (j - 4)];
        }
#pragma unroll
        for (int i = -3; i <= 3; ++i) {
#pragma unroll
            for (int j = 0; j < 9; ++j) r2[j] = c[(i + 1) * FT_LW + (j - 4)];
#pragma unroll
            for (int j = 1; j <= 7; ++j) {
                const int Ix = (r1[j + 1] - r1[j - 1]) * 2 + (r0[j + 1] - r0[j - 1]) + (r2[j + 1] - r2[j - 1]);
                const int Iy = (r2[j] - r0[j]) * 2 + (r2[j - 1] - r0[j - 1]) + (r2[j + 1] - r0[j + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                cc += Ix * Iy;
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                r0[j] = r1[j];
                r1[j] = r2[j];
            }
        }
        const float fa = (float)a, fb = (float)b, fc = (float)cc;
        const float sum = fa + fb;
        const float resp = ((fa * fb - fc * fc) - (0.04f * sum) * sum) * geo.harris_scale4;
        const int gx = gx0 + FT_HALO + px, gy = gy0 + FT_HALO + py;
        cand_packed[obase + ci] = (uint32_t)gx | ((uint32_t)gy << 12) | ((uint32_t)s << 24);
        cand_resp[obase + ci] = resp;
    }
}

// `geo` is the DEVICE copy of the geometry
extern "C" void afv_launch_fast_harris(const Geo *geo, int total_tiles, const FrameSrc *src0, const uint8_t *pyr, uint32_t *cand_packed,
                                       float *cand_resp, int *cand_count, int nframes, hipStream_t stream) {
    dim3 grid(total_tiles, nframes);
    hipLaunchKernelGGL(k_fast_harris, grid, dim3(256), 0, stream, geo, *src0, pyr, cand_packed, cand_resp, cand_count);
}
