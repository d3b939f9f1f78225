// k_harris.hip — E4a + E5: retainBest on the FAST score, then the Harris response of the survivors.
//
// Replaces, inside cv::ORB::detect (Feature_orb32.cpp:34; OpenCV orb.cpp computeKeyPoints), per pyramid level:
//   KeyPointsFilter::retainBest(keypoints, 2 * featuresNum)   — on the FAST score, ties at the threshold are all kept;
//   HarrisResponses(img, keypoints, 7, HARRIS_K)             — for what is left.
// On a dense scene more than 40 % of the FAST + NMS candidates fall to the first retainBest (cv::ORB is asked for 10x the final
// budget, Feature_orb32.cpp:28), so the 7x7 structure tensor is evaluated after it, not inside the FAST tiles:
//   k_retain_score   one workgroup per (frame, level): 256-bin histogram of the scores -> threshold T1 -> order-free compaction
//                    of the survivors into the level's `l1` list, and one work item per 64 survivors appended to a queue;
//   k_harris         a fixed grid walks the queue, one wavefront per item, one lane per candidate: the 9 x 9 window is read
//                    straight from the level image in global memory (the FAST tiles read it a moment ago), one aligned 12-byte
//                    load per row, and the integer sums a, b, c of the 7 x 7 block are built with packed i16 Sobel rows and
//                    v_dot2_i32_i16 (see harris_response).  The response is the float expression OpenCV evaluates:
//                    ((a*b - c*c) - k*(a+b)^2) * scale^4.
//                    Outside the image the window follows BORDER_REFLECT_101 (cv::ORB's apron): only the first pixel past an
//                    edge is ever needed, FAST corners being at least 3 px inside.
#include <algorithm>

#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

typedef short short2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ uint32_t as_u32(short2v v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ short2v as_s2(uint32_t v) { return __builtin_bit_cast(short2v, v); }

#define RS_CPT 36    // candidates per thread k_retain_score keeps in registers (lists up to 9216 entries)
#define HQ_CHUNK 64  // candidates per queue item {frame * AFV_MAX_LEVELS + level, first candidate | count << 24}: one pass of a 256-thread workgroup

__device__ __forceinline__ int wave_incl_scan_shfl(int v) { return afv_wave_incl_scan(v); }

__global__ __launch_bounds__(256) void k_retain_score(const Geo *__restrict__ geo_p, const uint32_t *__restrict__ cand_packed,
                                                      const int *__restrict__ cand_count, uint32_t *__restrict__ l1,
                                                      int *__restrict__ l1_count, uint2 *__restrict__ queue, int *__restrict__ queue_n,
                                                      int frame_base, int total_blocks) {
    __shared__ int hist[256];
    __shared__ int wsum[4];
    __shared__ int s_T1, s_n1, s_qb;
    const Geo &geo = *geo_p;
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);  // same placement as k_select_quadtree: a frame's levels on one XCD
    if (work >= total_blocks) return;
    const int l = work % geo.nlevels, f = frame_base + work / geo.nlevels;
    const LevelGeo &L = geo.lv[l];
    const size_t base = L.cand_off + (size_t)f * L.cand_frame_stride;
    const uint32_t *cp = cand_packed + base;
    uint32_t *out = l1 + base;
    const int n = min(cand_count[f * AFV_MAX_LEVELS + l], L.cand_cap);
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = 2 * L.cv_quota;
    int n1 = n;
    // the candidate list is read ONCE: every thread keeps its share (items tid, tid + 256, ...) in registers for the histogram and
    // the compaction pass; lists longer than 256 * RS_CPT (noise-like images) stream from memory a second time instead
    const bool reg = n <= 256 * RS_CPT;
    uint32_t cpk[RS_CPT];
    if (reg) {
#pragma unroll
        for (int k = 0; k < RS_CPT; ++k) {
            cpk[k] = 0;
            if (k * 256 < n && k * 256 + tid < n) cpk[k] = cp[k * 256 + tid];
        }
    }
    if (n > K) {  // uniform
        hist[tid] = 0;
        if (tid == 0) {
            s_T1 = 0;
            s_n1 = 0;
        }
        __syncthreads();
        if (reg) {
#pragma unroll
            for (int k = 0; k < RS_CPT; ++k)
                if (k * 256 < n && k * 256 + tid < n) atomicAdd(&hist[cpk[k] >> 24], 1);
        } else {
            for (int i = tid; i < n; i += 256) atomicAdd(&hist[cp[i] >> 24], 1);
        }
        __syncthreads();
        // thread t owns bin 255 - t: the threshold is the bin at which the count from the top reaches K
        const int h = hist[255 - tid];
        const int incl = wave_incl_scan_shfl(h);
        if (lane == 63) wsum[tid >> 6] = incl;
        __syncthreads();
        int before = incl - h;
        for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
        if (before < K && K <= before + h) s_T1 = 255 - tid;
        __syncthreads();
        const int T1 = s_T1;
        // order-free, wave-aggregated compaction (everything downstream is order-independent)
        auto emit = [&](bool in, uint32_t e) {
            const bool keep = in && (int)(e >> 24) >= T1;
            const unsigned long long m = __ballot(keep);
            int wbase = 0;
            if (lane == 0 && m) wbase = atomicAdd(&s_n1, __popcll(m));
            wbase = __shfl(wbase, 0, 64);
            if (keep) out[wbase + __popcll(m & ((1ull << lane) - 1ull))] = e;
        };
        if (reg) {
#pragma unroll
            for (int k = 0; k < RS_CPT; ++k) {
                if (k * 256 >= n) break;
                emit(k * 256 + tid < n, cpk[k]);
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += 256) {
                const int i = i0 + tid;
                emit(i < n, i < n ? cp[i] : 0u);
            }
        }
        __syncthreads();
        n1 = s_n1;
    } else if (reg) {
#pragma unroll
        for (int k = 0; k < RS_CPT; ++k)
            if (k * 256 < n && k * 256 + tid < n) out[k * 256 + tid] = cpk[k];
    } else {
        for (int i = tid; i < n; i += 256) out[i] = cp[i];
    }
    const int nchunks = (n1 + HQ_CHUNK - 1) / HQ_CHUNK;
    if (tid == 0) {
        l1_count[f * AFV_MAX_LEVELS + l] = n1;
        s_qb = nchunks ? atomicAdd(queue_n, nchunks) : 0;
    }
    __syncthreads();
    const int qb = s_qb;
    for (int i = tid; i < nchunks; i += 256) queue[qb + i] = make_uint2((uint32_t)(f * AFV_MAX_LEVELS + l), (uint32_t)(i * HQ_CHUNK) | ((uint32_t)min(HQ_CHUNK, n1 - i * HQ_CHUNK) << 24));
}

// Uniform (scalar) description of one queue item: up to 64 candidates of one level image, one lane each.
struct HarrisItem {
    const uint8_t *img;
    __amdgpu_buffer_rsrc_t rsrc;
    int pitch, lw, lh, cnt;
    size_t base;  // first l1 slot of the item
};

// raw buffer over one frame's level image: reads past its last byte return 0 instead of faulting.  Pointer, pitch and size go through
// v_readfirstlane: after the l == 0 / l > 0 join the compiler keeps them in vector registers, and with a resource it takes for divergent
// every window load sits in a waterfall loop (four v_readfirstlane, two 64-bit compares, a branch: 12 loads x 6 vector instructions per
// wavefront of candidates; round 5)
__device__ __forceinline__ void harris_item_rsrc(HarrisItem &it) {
    const uint64_t ip = reinterpret_cast<uint64_t>(it.img);
    const uint32_t ip_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ip), ip_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ip >> 32));
    it.img = reinterpret_cast<const uint8_t *>(((uint64_t)ip_hi << 32) | ip_lo);
    it.pitch = __builtin_amdgcn_readfirstlane(it.pitch);
    it.lw = __builtin_amdgcn_readfirstlane(it.lw);
    it.lh = __builtin_amdgcn_readfirstlane(it.lh);
    it.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(it.img), 0, (it.lh - 1) * it.pitch + it.lw, 0x00027000);
}

__device__ __forceinline__ HarrisItem harris_item(const Geo &geo, const FrameSrc &src0, const uint8_t *pyr, uint2 item) {
    HarrisItem it;
    const int fl = (int)item.x, f = fl / AFV_MAX_LEVELS, l = fl - f * AFV_MAX_LEVELS;
    const LevelGeo &L = geo.lv[l];
    if (l == 0) {
        it.img = src0.base + (size_t)f * src0.frame_stride;
        it.pitch = src0.stride;
    } else {
        it.img = pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
        it.pitch = L.pitch;
    }
    it.lw = L.w;
    it.lh = L.h;
    it.cnt = (int)(item.y >> 24);
    it.base = L.cand_off + (size_t)f * L.cand_frame_stride + (size_t)(item.y & 0x00ffffffu);
    harris_item_rsrc(it);
    return it;
}

// One lane = one candidate.  The nine window columns px - 4 .. px + 4 lie inside the 12 bytes that start at the aligned column xa:
// ONE dwordx3 load per window row, all nine in flight together.  Each row is expanded to packed u16 pairs P_k = (x[2k], x[2k+1])
// by v_perm_b32 with lane-dependent selectors (byte sh + 2k, byte sh + 2k + 1): k = 0, 1 from (w1:w0), k = 2, 3 from (w2:w1), k = 4
// from (w2:w1) too.  Block row r uses the window rows r, r + 1, r + 2:  S = r0 + 2 r1 + r2,  D = r2 - r0,
//   Ix = S_{k+1} - S_k,   Iy = D_k + D_{k+1} + 2 * (D_k.hi, D_{k+1}.lo)     for the block columns (2k + 1, 2k + 2)
// and a, b, c accumulate with v_dot2_i32_i16.
__device__ __forceinline__ float harris_response(const HarrisItem &it, uint32_t e, float scale4) {
    const int px = (int)(e & 4095u), py = (int)((e >> 12) & 4095u);  // 3 <= px <= lw - 4, 3 <= py <= lh - 4
    // BORDER_REFLECT_101, one pixel at most: column -1 (px == 3) is column 1, column lw (px == lw - 4) is column lw - 2 = window
    // column 6; row -1 is row 1, row lh is row lh - 2
    const bool left = px < 4, right = px + 4 >= it.lw;
    const int xa = left ? 0 : ((px - 4) & ~3), sh = left ? 0 : ((px - 4) & 3);
    const uint32_t selA = 0x0c010c00u + (uint32_t)sh * 0x00010001u, selB = selA + 0x00020002u;
    const uint32_t selC = (right ? 0x0c0c0c02u : 0x0c0c0c04u) + (uint32_t)sh;  // window column 8 (or 6) inside (w2:w1)
    uint32_t off[9];
    off[1] = (uint32_t)(py - 3) * (uint32_t)it.pitch + (uint32_t)xa;
#pragma unroll
    for (int r = 2; r < 8; ++r) off[r] = off[r - 1] + (uint32_t)it.pitch;
    off[0] = (py == 3) ? off[2] : off[1] - (uint32_t)it.pitch;
    off[8] = (py + 4 >= it.lh) ? off[6] : off[7] + (uint32_t)it.pitch;
    uint32_t w[9][3];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const uint3v v = __builtin_amdgcn_raw_buffer_load_b96(it.rsrc, off[r], 0, 0);
        w[r][0] = v.x;
        w[r][1] = v.y;
        w[r][2] = v.z;
    }
    // the third dword of a load may lie past the end of the image's memory (bottom rows of an image whose pitch is its width): it
    // came back as 0 or not at all; rebuild those rows from two dwords and single bytes
    if (py + 5 >= it.lh && xa + 12 > it.lw) {
#pragma unroll
        for (int r = 6; r < 9; ++r) {
            const uint2v v = __builtin_amdgcn_raw_buffer_load_b64(it.rsrc, off[r], 0, 0);
            uint32_t w2 = 0;
            for (int j = 0; j < 4; ++j)
                if (xa + 8 + j < it.lw) w2 |= (uint32_t)it.img[off[r] + 8 + j] << (8 * j);
            w[r][0] = v.x;
            w[r][1] = v.y;
            w[r][2] = w2;
        }
    }
    if (left) {  // shift the rows up by one byte: bytes (c1, c0, c1, c2 | c3 .. c6 | c7 ..)
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            w[r][2] = __builtin_amdgcn_alignbit(w[r][2], w[r][1], 24);
            w[r][1] = __builtin_amdgcn_alignbit(w[r][1], w[r][0], 24);
            w[r][0] = (w[r][0] << 8) | ((w[r][0] >> 8) & 0xffu);
        }
    }
    short2v two;
    two.x = two.y = 2;
    short2v P[3][5];  // sliding window of three expanded rows
    int a = 0, b = 0, c = 0;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        short2v *row = P[r % 3];
        row[0] = as_s2(__builtin_amdgcn_perm(w[r][1], w[r][0], selA));
        row[1] = as_s2(__builtin_amdgcn_perm(w[r][1], w[r][0], selB));
        row[2] = as_s2(__builtin_amdgcn_perm(w[r][2], w[r][1], selA));
        row[3] = as_s2(__builtin_amdgcn_perm(w[r][2], w[r][1], selB));
        row[4] = as_s2(__builtin_amdgcn_perm(w[r][2], w[r][1], selC));
        if (r >= 2) {
            const short2v *r0 = P[(r - 2) % 3], *r1 = P[(r - 1) % 3], *r2 = row;
            short2v Sp[5], Dp[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                Sp[k] = r1[k] * two + r0[k] + r2[k];
                Dp[k] = r2[k] - r0[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                short2v Ix = Sp[k + 1] - Sp[k];
                const short2v O = as_s2(__builtin_amdgcn_alignbit(as_u32(Dp[k + 1]), as_u32(Dp[k]), 16));
                short2v Iy = O * two + Dp[k] + Dp[k + 1];
                if (k == 3) {  // column 8 is outside the block
                    Ix = as_s2(as_u32(Ix) & 0xffffu);
                    Iy = as_s2(as_u32(Iy) & 0xffffu);
                }
                a = __builtin_amdgcn_sdot2(Ix, Ix, a, false);
                b = __builtin_amdgcn_sdot2(Iy, Iy, b, false);
                c = __builtin_amdgcn_sdot2(Ix, Iy, c, false);
            }
        }
    }
    const float fa = (float)a, fb = (float)b, fc = (float)c;
    const float sum = fa + fb;
    return ((fa * fb - fc * fc) - (0.04f * sum) * sum) * scale4;
}

__global__ __launch_bounds__(256) void k_harris(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                const uint32_t *__restrict__ l1, float *__restrict__ l1_resp,
                                                const uint2 *__restrict__ queue, const int *__restrict__ queue_n) {
    const Geo &geo = *geo_p;
    const int qn = *queue_n;
    const int lane = threadIdx.x & 63;
    // one queue item per wavefront and step; a contiguous range of the queue per wavefront (neighbouring items are neighbouring
    // candidates of one level image)
    // XCD-aware: the workgroups of one XCD take one contiguous eighth of the queue, so the chunks of a level image (contiguous in
    // the queue) are fetched through one L2 instead of eight (gridDim.x is a multiple of 8)
    const int nwaves = (int)gridDim.x * 4;
    const int wv = __builtin_amdgcn_readfirstlane(afv_xcd_remap((int)blockIdx.x, (int)gridDim.x) * 4 + (int)(threadIdx.x >> 6));
    const int per = (qn + nwaves - 1) / nwaves;
    const int w_end = min(qn, (wv + 1) * per);
    for (int w = wv * per; w < w_end; ++w) {
        const HarrisItem it = harris_item(geo, src0, pyr, queue[w]);
        if (lane < it.cnt) l1_resp[it.base + lane] = harris_response(it, l1[it.base + lane], geo.harris_scale4);
    }
}

// ---------------- retainBest + Harris in ONE launch (the small-batch path: one or a few frames) ----------------
// k_retain_score + k_harris cost one frame 11 + 5 us plus a launch boundary: one workgroup per level walks a list of up to ~11 000
// candidates in 36 dependent compaction steps, then a second kernel picks the survivors up from a queue.  Here a level is dealt to
// RH_SLICES workgroups of 1024 threads.  Every one of them reads the WHOLE level once (44 KB out of L2, a thread's <= 12 loads in
// flight together, the values stay in registers), histograms it, derives the same threshold T1, counts the survivors in front of its
// slice (its output offset: deterministic, no counter to clear), compacts the survivors of its slice in LDS and computes their Harris
// responses - one lane per survivor, the same harris_response as k_harris.  The last slice publishes the level's survivor count.
// Global round trips on the critical path: candidate count -> candidates -> Harris windows.
#define RH_T 1024
#define RH_SLICES 4
#define RH_CPT 12     // candidates a thread keeps in registers (lists up to 12 288 entries; longer ones are re-read from memory)
#define RH_SURV 3072  // survivors compacted per pass (= the slice of a 12 288-entry list)

template <bool REG>
__device__ __forceinline__ void retain_harris_small_body(const Geo &geo, const FrameSrc &src0, const uint8_t *__restrict__ pyr,
                                                         const uint32_t *__restrict__ cp, int n, int l, int f, int slice, size_t base,
                                                         uint32_t *__restrict__ l1, int *__restrict__ l1_count, float *__restrict__ l1_resp) {
    __shared__ int hist[4][256];  // four copies (lane & 3): FAST scores crowd a few bins just above the threshold
    __shared__ int wsum[4];
    __shared__ int s_T1, s_before, s_mine;
    __shared__ uint32_t s_surv[RH_SURV];
    const LevelGeo &L = geo.lv[l];
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = 2 * L.cv_quota;
    const int start = (int)((long)n * slice / RH_SLICES), end = (int)((long)n * (slice + 1) / RH_SLICES);
    const int nk = (n + RH_T - 1) / RH_T;  // items per thread
    uint32_t cpk[RH_CPT];
    if (REG) {
#pragma unroll
        for (int k = 0; k < RH_CPT; ++k) cpk[k] = (k < nk && k * RH_T + tid < n) ? cp[k * RH_T + tid] : 0u;
    }
#define RH_FOR_ITEMS(BODY)                                            \
    if (REG) {                                                        \
        _Pragma("unroll") for (int k = 0; k < RH_CPT; ++k) {          \
            const int i = k * RH_T + tid;                             \
            const uint32_t e = cpk[k];                                \
            if (k < nk) { BODY }                                      \
        }                                                             \
    } else {                                                          \
        for (int k = 0; k < nk; ++k) {                                \
            const int i = k * RH_T + tid;                             \
            const uint32_t e = i < n ? cp[i] : 0u;                    \
            { BODY }                                                  \
        }                                                             \
    }
    int T1 = 0;
    if (tid == 0) {
        s_before = 0;
        s_mine = 0;
    }
    if (n > K) {  // uniform
        if (tid < 256) hist[0][tid] = hist[1][tid] = hist[2][tid] = hist[3][tid] = 0;
        if (tid == 0) s_T1 = 0;
        __syncthreads();
        RH_FOR_ITEMS(if (i < n) atomicAdd(&hist[lane & 3][e >> 24], 1);)
        __syncthreads();
        // thread t < 256 owns bin 255 - t: the threshold is the bin at which the count from the top reaches K
        int h = 0, incl = 0;
        if (tid < 256) {
            h = hist[0][255 - tid] + hist[1][255 - tid] + hist[2][255 - tid] + hist[3][255 - tid];
            incl = afv_wave_incl_scan(h);
            if (lane == 63) wsum[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            int before = incl - h;
            for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
            if (before < K && K <= before + h) s_T1 = 255 - tid;
        }
        __syncthreads();
        T1 = s_T1;  // (the barriers above also published s_before / s_mine = 0)
    } else {
        __syncthreads();  // s_before / s_mine cleared
    }
    {   // survivors in front of the slice: this workgroup's first output slot
        int c = 0;
        RH_FOR_ITEMS(c += (i < start && (int)(e >> 24) >= T1) ? 1 : 0;)
        c = afv_wave_incl_scan(c);
        if (lane == 63 && c) atomicAdd(&s_before, c);
    }
    __syncthreads();
    int out_base = s_before;
    // raw buffer over this frame's level image, as in harris_item
    HarrisItem it;
    if (l == 0) {
        it.img = src0.base + (size_t)f * src0.frame_stride;
        it.pitch = src0.stride;
    } else {
        it.img = pyr + L.pyr_off + (size_t)f * L.pyr_frame_stride;
        it.pitch = L.pitch;
    }
    it.lw = L.w;
    it.lh = L.h;
    it.cnt = 0;
    it.base = 0;
    harris_item_rsrc(it);
    for (int s0 = start; s0 < end; s0 += RH_SURV) {  // uniform trip count (one pass for lists that fit the registers)
        const int s1 = min(s0 + RH_SURV, end);
        RH_FOR_ITEMS(
            if (k * RH_T < s1 && (k + 1) * RH_T > s0) {  // uniform
                const bool keep = i >= s0 && i < s1 && (int)(e >> 24) >= T1;
                const unsigned long long m = __ballot(keep);
                int wbase = 0;
                if (lane == 0 && m) wbase = atomicAdd(&s_mine, __popcll(m));
                wbase = __shfl(wbase, 0, 64);
                if (keep) s_surv[wbase + __popcll(m & ((1ull << lane) - 1ull))] = e;
            })
        __syncthreads();
        const int cnt = s_mine;
        for (int j = tid; j < cnt; j += RH_T) {
            const uint32_t es = s_surv[j];
            l1[base + out_base + j] = es;
            l1_resp[base + out_base + j] = harris_response(it, es, geo.harris_scale4);
        }
        out_base += cnt;
        if (s0 + RH_SURV < end) {  // uniform: another pass follows (lists beyond the registers only) - the counter is re-armed for it
            __syncthreads();
            if (tid == 0) s_mine = 0;
            __syncthreads();
        }
    }
#undef RH_FOR_ITEMS
    if (slice == RH_SLICES - 1 && tid == 0) l1_count[f * AFV_MAX_LEVELS + l] = out_base;
}

__global__ __launch_bounds__(RH_T) void k_retain_harris_small(const Geo *__restrict__ geo_p, FrameSrc src0, const uint8_t *__restrict__ pyr,
                                                              const uint32_t *__restrict__ cand_packed, const int *__restrict__ cand_count,
                                                              uint32_t *__restrict__ l1, int *__restrict__ l1_count, float *__restrict__ l1_resp,
                                                              int frame_base, int total_blocks) {
    const Geo &geo = *geo_p;
    const int work = (int)blockIdx.x;
    if (work >= total_blocks) return;
    const int slice = work % RH_SLICES, fl = work / RH_SLICES;
    const int l = fl % geo.nlevels, f = frame_base + fl / geo.nlevels;
    const LevelGeo &L = geo.lv[l];
    const size_t base = L.cand_off + (size_t)f * L.cand_frame_stride;
    const int n = min(cand_count[f * AFV_MAX_LEVELS + l], L.cand_cap);
    if (n <= RH_T * RH_CPT) retain_harris_small_body<true>(geo, src0, pyr, cand_packed + base, n, l, f, slice, base, l1, l1_count, l1_resp);
    else retain_harris_small_body<false>(geo, src0, pyr, cand_packed + base, n, l, f, slice, base, l1, l1_count, l1_resp);
}

// queue capacity per frame: every level can hand over all its candidate slots
extern "C" size_t afv_harris_queue_per_frame(const Geo *g) {
    size_t n = 0;
    for (int l = 0; l < g->nlevels; ++l) n += (size_t)(g->lv[l].cand_cap + HQ_CHUNK - 1) / HQ_CHUNK;
    return n;
}

// `queue` / `queue_n` belong to this launch (the runtime hands every chunk of a split batch its own); *queue_n must be 0 on entry
extern "C" void afv_launch_retain_harris(const Geo *geo_dev, int nlevels, const FrameSrc *src0, const uint8_t *pyr,
                                         const uint32_t *cand_packed, const int *cand_count, uint32_t *l1, int *l1_count,
                                         float *l1_resp, uint2 *queue, int *queue_n, int frame_base, int nframes, int small, hipStream_t stream) {
    if (small) {
        const int total = nlevels * nframes * RH_SLICES;
        hipLaunchKernelGGL(k_retain_harris_small, dim3(total), dim3(RH_T), 0, stream, geo_dev, *src0, pyr, cand_packed, cand_count, l1, l1_count,
                           l1_resp, frame_base, total);
        return;
    }
    const int total = nlevels * nframes;
    hipLaunchKernelGGL(k_retain_score, dim3((total + 7) / 8 * 8), dim3(256), 0, stream, geo_dev, cand_packed, cand_count, l1, l1_count,
                       queue, queue_n, frame_base, total);
    const int grid = (int)std::min<long>(std::max<long>((long)nframes * 80, 256), 65536) & ~7;  // 4 items per workgroup and step
    hipLaunchKernelGGL(k_harris, dim3(grid), dim3(256), 0, stream, geo_dev, *src0, pyr, l1, l1_resp, queue, queue_n);
}
