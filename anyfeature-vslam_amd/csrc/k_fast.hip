// k_fast.hip — E3 + E5 fused: FAST-9/16 score, 3x3 non-max suppression and Harris response.
//
// Replaces, for every pyramid level of every frame of the batch, the per-level work of cv::ORB::detect
// (Feature_orb32.cpp:34): FastFeatureDetector(20, true) (OpenCV fast.cpp FAST_t<16> + cornerScore<16>) and
// HarrisResponses(blockSize 7, k 0.04) (OpenCV orb.cpp).  One 256-thread workgroup owns a 64x32 tile:
//   1. the tile plus a 4 px halo is staged in LDS with coalesced dword loads (reflect-101 at the image edge —
//      FAST never reads it, Harris does, exactly like cv::ORB's 23 px apron);
//   2a. every pixel of the 66x34 ring-extended tile takes OpenCV's necessary pre-test (each of the 4 even antipodal
//      ring pairs must hold a darker / a brighter pixel), evaluated on packed (v-r, r-v) i16 pairs; survivors are
//      compacted into an LDS list so that
//   2b. the exact corner score = largest threshold for which the pixel is still a 9-arc corner runs on dense
//      wavefronts: 16 ring differences packed as (v-r, r-v), van-Herk prefix/suffix minima over the two ring halves
//      (59 packed min/max for both polarities), corner iff score > threshold;
//   3. strict 3x3 maxima are compacted into a second LDS list (<= 512 per tile);
//   4. that list is processed densely: integer Harris sums a,b,c over the 7x7 block from LDS, one float
//      expression for the response;
//   5. one global atomic per tile reserves output slots in the (frame, level) candidate array.
// Thread (tx, ty) = (tid & 63, tid >> 6) walks rows ty, ty+4, ...: no integer division, row validity is wave-uniform.
// Candidates leave the kernel unordered; everything downstream is order-independent (ties are broken by the
// raster index), see DESIGN.md "canonical order".
#include "afv_device.h"


typedef short short2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2v pkmin(short2v a, short2v b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ short2v pkmax(short2v a, short2v b) { return __builtin_elementwise_max(a, b); }

// packed ring difference (v - r, r - v) = (r, r) * (-1, +1) + (v, -v): one v_pk_mad_i16 per ring pixel
__device__ __forceinline__ short2v ring_diff(const uint8_t *c, int off, short2v vn) {
    const short r = (short)c[off];
    short2v R;
    R.x = r;
    R.y = r;
    short2v K;
    K.x = -1;
    K.y = 1;
    return R * K + vn;
}

#define RING_OFF(dx, dy) ((dy) * FT_LW + (dx))

// OpenCV pre-test on the even antipodal pairs (0,8) (2,10) (4,12) (6,14): lo halves carry "darker", hi "brighter"
__device__ __forceinline__ bool fast_pretest(const uint8_t *c, int threshold) {
    const short v = (short)c[0];
    short2v vn;
    vn.x = v;
    vn.y = (short)-v;
    const short2v m0 = pkmax(ring_diff(c, RING_OFF(0, 3), vn), ring_diff(c, RING_OFF(0, -3), vn));
    const short2v m2 = pkmax(ring_diff(c, RING_OFF(2, 2), vn), ring_diff(c, RING_OFF(-2, -2), vn));
    const short2v m4 = pkmax(ring_diff(c, RING_OFF(3, 0), vn), ring_diff(c, RING_OFF(-3, 0), vn));
    const short2v m6 = pkmax(ring_diff(c, RING_OFF(2, -2), vn), ring_diff(c, RING_OFF(-2, 2), vn));
    const short2v m = pkmin(pkmin(m0, m2), pkmin(m4, m6));
    return max((int)m.x, (int)m.y) > threshold;
}

// score = max over the 16 arcs of 9 contiguous ring pixels of min(v - ring) [dark arc] and of min(ring - v)
// [bright arc]; corner iff score > threshold; cornerScore<16> returns score - 1.
__device__ __forceinline__ int fast_score(const uint8_t *c, int threshold) {
    const short v = (short)c[0];
    short2v vn;
    vn.x = v;
    vn.y = (short)-v;
    short2v d[16];
    d[0] = ring_diff(c, RING_OFF(0, 3), vn);
    d[1] = ring_diff(c, RING_OFF(1, 3), vn);
    d[2] = ring_diff(c, RING_OFF(2, 2), vn);
    d[3] = ring_diff(c, RING_OFF(3, 1), vn);
    d[4] = ring_diff(c, RING_OFF(3, 0), vn);
    d[5] = ring_diff(c, RING_OFF(3, -1), vn);
    d[6] = ring_diff(c, RING_OFF(2, -2), vn);
    d[7] = ring_diff(c, RING_OFF(1, -3), vn);
    d[8] = ring_diff(c, RING_OFF(0, -3), vn);
    d[9] = ring_diff(c, RING_OFF(-1, -3), vn);
    d[10] = ring_diff(c, RING_OFF(-2, -2), vn);
    d[11] = ring_diff(c, RING_OFF(-3, -1), vn);
    d[12] = ring_diff(c, RING_OFF(-3, 0), vn);
    d[13] = ring_diff(c, RING_OFF(-3, 1), vn);
    d[14] = ring_diff(c, RING_OFF(-2, 2), vn);
    d[15] = ring_diff(c, RING_OFF(-1, 3), vn);
    // van Herk: window k (9 long, circular) = suffix of its ring half starting at k + prefix of the other half
    short2v suf0[8], pre0[8], suf1[8], pre1[8];
    suf0[7] = d[7];
    suf1[7] = d[15];
    pre0[0] = d[0];
    pre1[0] = d[8];
#pragma unroll
    for (int k = 6; k >= 0; --k) {
        suf0[k] = pkmin(d[k], suf0[k + 1]);
        suf1[k] = pkmin(d[8 + k], suf1[k + 1]);
    }
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        pre0[k] = pkmin(d[k], pre0[k - 1]);
        pre1[k] = pkmin(d[8 + k], pre1[k - 1]);
    }
    short2v best = pkmin(suf0[0], pre1[0]);
#pragma unroll
    for (int k = 1; k < 8; ++k) best = pkmax(best, pkmin(suf0[k], pre1[k]));
#pragma unroll
    for (int k = 0; k < 8; ++k) best = pkmax(best, pkmin(suf1[k], pre0[k]));
    const int s = max((int)best.x, (int)best.y);
    return s > threshold ? s - 1 : 0;
}

// wave-aggregated append of `val` to an LDS list (one LDS atomic per wavefront)
__device__ __forceinline__ void wave_push(bool pred, unsigned short *list, int *count, unsigned short val, int lane) {
    const unsigned long long m = __ballot(pred);
    if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(count, __popcll(m));
        base = __shfl(base, 0, 64);
        if (pred) list[base + __popcll(m & ((1ull << lane) - 1ull))] = val;
    }
}

__global__ __launch_bounds__(256) void k_fast_harris(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                     uint32_t *__restrict__ cand_packed, float *__restrict__ cand_resp,
                                                     int *__restrict__ cand_count, int total_blocks, int frame_base) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[FT_LH * FT_LW];
    __shared__ __attribute__((aligned(16))) uint8_t sc[FT_LH * FT_LW];  // same geometry as `tile`
    __shared__ unsigned short pre[(FT_W + 2) * (FT_H + 2)];
    __shared__ uint32_t list[512];
    __shared__ int list_n, out_base, pre_n;

    const Geo &geo = *geo_p;
    // XCD-aware placement: every XCD works on whole frames, so the halo / cache-line sharing between neighbouring
    // tiles stays inside one L2
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    const int f0 = work / geo.total_tiles, tile_id = work - f0 * geo.total_tiles;
    const int f = frame_base + f0;
    int l = 0;
#pragma unroll
    for (int i = 1; i < AFV_MAX_LEVELS; ++i)
        if (i < geo.nlevels && tile_id >= geo.lv[i].tile_base) l = i;
    const LevelGeo &L = geo.lv[l];
    const int t = tile_id - L.tile_base;
    const int tyi = t / L.tiles_x, txi = t - tyi * L.tiles_x;
    const int gx0 = txi * FT_W - FT_HALO, gy0 = tyi * FT_H - FT_HALO;
    const int lw = L.w, lh = L.h;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, lane = tx;

    const uint8_t *img;
    int pitch;
    if (l == 0) {
        img = src0.base + (size_t)f * src0.frame_stride;
        pitch = src0.stride;
    } else {
        img = pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
        pitch = L.pitch;
    }
    if (threadIdx.x == 0) {
        list_n = 0;
        pre_n = 0;
    }

    // 1. stage 72x40 bytes: 18 dwords per row; clear the score plane
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
        *reinterpret_cast<uint32_t *>(&sc[ry * FT_LW + rq * 4]) = 0u;
    }
    __syncthreads();

    // 2a. pre-test on LDS rows 3..36 x columns 3..68 (= tile pixels -1..64 x -1..32).  Each lane owns one column and
    //     walks rows ty, ty+4, ...; its pass bits are collected in a register and compacted once per wavefront.
    const int thr = geo.fast_threshold;
    {
        const int col = 3 + tx, gx = gx0 + col;
        const bool col_ok = gx >= 3 && gx < lw - 3;
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < (FT_H + 2 + 3) / 4; ++k) {
            const int r = ty + 4 * k;  // wave-uniform row
            const int gy = gy0 + 3 + r;
            if (r < FT_H + 2 && gy >= 3 && gy < lh - 3) {
                if (col_ok && fast_pretest(&tile[(r + 3) * FT_LW + col], thr)) bits |= 1u << k;
            }
        }
        // the two extra columns 67, 68: 34 rows x 2 = 68 positions, handled by waves 0 and 1 (bit 15)
        int p_extra = 0;
        if (threadIdx.x < 128) {
            const int r = threadIdx.x >> 1, c2 = 67 + (threadIdx.x & 1);
            const int gy = gy0 + 3 + r, gx2 = gx0 + c2;
            const bool ok = r < FT_H + 2 && gy >= 3 && gy < lh - 3 && gx2 >= 3 && gx2 < lw - 3;
            p_extra = (r + 3) * FT_LW + c2;
            if (ok && fast_pretest(&tile[p_extra], thr)) bits |= 1u << 15;
        }
        // wave-level exclusive prefix of popcounts -> one LDS atomic per wavefront
        const int cnt = __popc(bits);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t2 = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t2;
        }
        int base = 0;
        if (lane == 63) base = atomicAdd(&pre_n, incl);
        base = __shfl(base, 63, 64) + incl - cnt;
        uint32_t b = bits;
        while (b) {
            const int k = __builtin_ctz(b);
            b &= b - 1;
            pre[base++] = (unsigned short)(k == 15 ? p_extra : (ty + 4 * k + 3) * FT_LW + col);
        }
    }
    __syncthreads();
    // 2b. exact corner score for the survivors (dense)
    const int npre = pre_n;
    for (int i = threadIdx.x; i < npre; i += 256) {
        const int p = pre[i];
        sc[p] = (uint8_t)fast_score(&tile[p], thr);
    }
    __syncthreads();

    // 3. strict 3x3 maxima -> LDS list.  Only scored positions can be corners, so the NMS walks the survivor list of
    //    step 2 (interior positions only) instead of all 2048 pixels.
    for (int i0 = 0; i0 < npre; i0 += 256) {
        const int i = i0 + threadIdx.x;
        bool keep = false;
        int px = 0, py = 0, s0 = 0;
        if (i < npre) {
            const int p = pre[i];
            py = p / FT_LW - FT_HALO;
            px = p - (py + FT_HALO) * FT_LW - FT_HALO;
            const uint8_t *q = &sc[p];
            s0 = q[0];
            keep = s0 != 0 && px >= 0 && px < FT_W && py >= 0 && py < FT_H && s0 > q[-1] && s0 > q[1] && s0 > q[-FT_LW - 1] &&
                   s0 > q[-FT_LW] && s0 > q[-FT_LW + 1] && s0 > q[FT_LW - 1] && s0 > q[FT_LW] && s0 > q[FT_LW + 1];
        }
        const unsigned long long m = __ballot(keep);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&list_n, __popcll(m));
            base = __shfl(base, 0, 64);
            if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)px | ((uint32_t)py << 8) | ((uint32_t)s0 << 16);
        }
    }
    __syncthreads();
    const int n = list_n;
    if (n == 0) return;
    if (threadIdx.x == 0) out_base = atomicAdd(&cand_count[f * AFV_MAX_LEVELS + l], n);
    __syncthreads();

    // 4. Harris response for the compacted candidates: 8 lanes per candidate, lane `sub` owns block row sub-3 (3 source
    //    rows of 9 pixels -> 7 gradient pairs), partial integer sums are combined with 3 xor-shuffles.  All four
    //    wavefronts share the work, so no wave is left with a long serial tail.
    const size_t obase = L.cand_off + (size_t)f * L.cand_frame_stride + (size_t)out_base;
    const int sub = threadIdx.x & 7;
    for (int c0 = 0; c0 < n; c0 += 32) {
        const int ci = c0 + (threadIdx.x >> 3);
        const bool act = ci < n;
        const uint32_t e = list[act ? ci : 0];
        const int px = e & 255, py = (e >> 8) & 255, s = e >> 16;
        int a = 0, b = 0, cc = 0;
        if (act && sub < 7) {
            const uint8_t *c = &tile[(py + FT_HALO + (sub - 3)) * FT_LW + (px + FT_HALO)];
            // 3 source rows x 9 pixels [x-4, x+4]: three aligned dwords per row, funnel-shifted so that byte k of the
            // row sits at a lane-independent position (bytes are then picked with static SDWA selects)
            int r0[9], r1[9], r2[9];
            const int xa = (px + FT_HALO - 4) & ~3, sh = ((px + FT_HALO - 4) & 3) * 8;
            const uint8_t *rowp = &tile[(py + FT_HALO + (sub - 3) - 1) * FT_LW + xa];
#define LOAD_ROW(R, PTR)                                                                      \
    {                                                                                         \
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(PTR);                          \
        const uint32_t w1 = *reinterpret_cast<const uint32_t *>((PTR) + 4);                    \
        const uint32_t w2 = *reinterpret_cast<const uint32_t *>((PTR) + 8);                    \
        const uint32_t a0 = __builtin_amdgcn_alignbit(w1, w0, sh), a1 = __builtin_amdgcn_alignbit(w2, w1, sh); \
        const uint32_t a2 = w2 >> sh;                                                          \
        R[0] = a0 & 255; R[1] = (a0 >> 8) & 255; R[2] = (a0 >> 16) & 255; R[3] = a0 >> 24;     \
        R[4] = a1 & 255; R[5] = (a1 >> 8) & 255; R[6] = (a1 >> 16) & 255; R[7] = a1 >> 24;     \
        R[8] = a2 & 255;                                                                       \
    }
            LOAD_ROW(r0, rowp)
            LOAD_ROW(r1, rowp + FT_LW)
            LOAD_ROW(r2, rowp + 2 * FT_LW)
#undef LOAD_ROW
            (void)c;
#pragma unroll
            for (int j = 1; j <= 7; ++j) {
                const int Ix = (r1[j + 1] - r1[j - 1]) * 2 + (r0[j + 1] - r0[j - 1]) + (r2[j + 1] - r2[j - 1]);
                const int Iy = (r2[j] - r0[j]) * 2 + (r2[j - 1] - r0[j - 1]) + (r2[j + 1] - r0[j + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                cc += Ix * Iy;
            }
        }
#pragma unroll
        for (int m = 1; m <= 4; m <<= 1) {
            a += __shfl_xor(a, m, 64);
            b += __shfl_xor(b, m, 64);
            cc += __shfl_xor(cc, m, 64);
        }
        if (act && sub == 0) {
            const float fa = (float)a, fb = (float)b, fc = (float)cc;
            const float sum = fa + fb;
            const float resp = ((fa * fb - fc * fc) - (0.04f * sum) * sum) * geo.harris_scale4;
            const int gx = gx0 + FT_HALO + px, gy = gy0 + FT_HALO + py;
            cand_packed[obase + ci] = (uint32_t)gx | ((uint32_t)gy << 12) | ((uint32_t)s << 24);
            cand_resp[obase + ci] = resp;
        }
    }
}

// `geo` is the DEVICE copy of the geometry
extern "C" void afv_launch_fast_harris(const Geo *geo, int total_tiles, const FrameSrc *src0, const uint8_t *pyr, uint32_t *cand_packed,
                                       float *cand_resp, int *cand_count, int frame_base, int nframes, hipStream_t stream) {
    const int total = total_tiles * nframes;
    dim3 grid((total + 7) / 8 * 8);
    hipLaunchKernelGGL(k_fast_harris, grid, dim3(256), 0, stream, geo, *src0, pyr, cand_packed, cand_resp, cand_count, total, frame_base);
}
