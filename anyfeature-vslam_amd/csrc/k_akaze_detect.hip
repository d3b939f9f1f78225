// k_akaze_detect.hip — AKAZE Feature_Detection: scale-space extrema, ordered duplicate suppression,
// upper-level filter and sub-pixel refinement.  Restates libAKAZE 1.5 AKAZE::Find_Scale_Space_Extrema and
// Do_Subpixel_Refinement (the calls behind FeatureExtractor_akaze61::detectKeypoints, Feature_akaze61.cpp:38-47); the
// operation order is the one written down in oracle/akaze.c.
//
//  1. candidates: a streaming pass turns every level's Ldet plane into a bitmap of (3x3 strict maximum + thresholds +
//     descriptor-border rule); one workgroup per (level, frame) then counts the rows from the bitmap, scans them and
//     expands the bits, which leaves the candidates in RASTER order without a sort (upstream's loop order is part of
//     the result).
//  2. k_akz_suppress: upstream inserts the candidates one by one into kpts_aux, comparing each against the FIRST earlier
//     entry of the same / previous level within its radius (replace it or drop the newcomer).  One workgroup per (frame,
//     level) replays that loop in speculative rounds of AKD_R (256) consecutive candidates: the (candidate, grid cell) pairs
//     of a round are scanned by all threads in two uniform grids (the level below, this level; entries inline in the cell
//     lists); then every candidate checks exactly whether an earlier candidate of the round changes what its search saw (a
//     new / moved entry inside its radius, or a replaced entry that lay inside it) and the round commits, in parallel, up to
//     the first such candidate.  The levels of a frame run as a pipeline (see the kernel); the upper-level filter of a level
//     pair runs in the workgroup that finishes it.
//  3. k_akz_refine_a / _b: the 2x2 sub-pixel solve and the ordered compaction over the slot space.
#include <cstdlib>
#include "afv_device.h"
#include "akz_jobs.h"
#include "../../include/afv_hip.h"



__device__ __forceinline__ int akd_fround(float x) { return (int)(x + 0.5f); }

__device__ __forceinline__ bool akd_is_candidate(const AkdParams &P, const AkdLevel &L, const float *ld, int jx, int iy) {
    const int w = L.w;
    const float *c = ld + (size_t)iy * w, *m = c - w, *q = c + w;
    const float v = c[jx];
    if (!(v > P.dthreshold && v >= P.min_dthreshold && v > c[jx - 1] && v > c[jx + 1] && v > m[jx - 1] && v > m[jx] && v > m[jx + 1] &&
          v > q[jx - 1] && v > q[jx] && v > q[jx + 1]))
        return false;
    // descriptor-border rule ("is_out"): such a point never changes kpts_aux, so it is dropped before the ordered pass
    const float smax = 10.0f * sqrtf(2.0f);
    const float px = (float)jx, py = (float)iy, r = smax * (float)L.sigma_size;
    const int left_x = akd_fround(px - r) - 1, right_x = akd_fround(px + r) + 1, up_y = akd_fround(py - r) - 1, down_y = akd_fround(py + r) + 1;
    return !(left_x < 0 || right_x >= L.w || up_y < 0 || down_y >= L.h);
}

// pass A: candidate bitmap of all levels in one launch.  One wavefront per strip of 64 columns x AKM_ROWS rows: lane = column,
// the rows slide through registers (row above / centre / below), the horizontal neighbours come from DPP wave shifts plus one
// edge column either side, so the Ldet plane is streamed once with full-line loads and no LDS.  v > all eight neighbours  <=>
// v > max(left, right) of its own row and v > the 3-wide row maxima above and below.  Bit x of word
// mask[frame][row][x / 64] = pixel (x, row) is a candidate.
#define AKD_MAXCHUNKS 32  // 64-column chunks per row: levels up to 2048 pixels wide
#ifndef AKM_ROWS
#define AKM_ROWS 32
#endif
#ifndef AKM_G
#define AKM_G 8
#endif
struct AkmWork {
    int strip_off[17];  // first strip of every level (strips of a level: chunks x row blocks x frames)
};
__device__ __forceinline__ float akm_from_left(float v) {   // lane i <- lane i - 1 (DPP wave_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float akm_from_right(float v) {  // lane i <- lane i + 1 (DPP wave_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__global__ __launch_bounds__(256) void k_akz_cand_mask(AkdParams P, AkmWork W, int nframes, unsigned long long *__restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int strip = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);  // wave-uniform: level / strip / row arithmetic on the scalar unit
    if (strip >= W.strip_off[P.nlevels]) return;
    int level = 0;
    while (strip >= W.strip_off[level + 1]) ++level;
    const AkdLevel L = P.lv[level];
    const int nchunks = (L.w + 63) >> 6, nrb = (L.h + AKM_ROWS - 1) / AKM_ROWS;
    int t = strip - W.strip_off[level];
    const int f = t / (nchunks * nrb);
    t -= f * nchunks * nrb;
    const int rb = t / nchunks, ch = t - rb * nchunks;
    const int tx0 = ch << 6, ty0 = rb * AKM_ROWS;
    const float *ld = L.ldet + (size_t)f * L.w * L.h;
    const int jx = tx0 + lane, xc = min(jx, L.w - 1), xl = max(tx0 - 1, 0), xr = min(tx0 + 64, L.w - 1);
    const float smax = 10.0f * sqrtf(2.0f), r = smax * (float)L.sigma_size;
    // descriptor-border rule ("is_out"), column part: such a point never changes kpts_aux, so it is dropped before the ordered pass
    const float px = (float)jx;
    const bool col_ok = jx >= 1 && jx < L.w - 1 && !(akd_fround(px - r) - 1 < 0 || akd_fround(px + r) + 1 >= L.w);
    unsigned long long *mk = mask + ((size_t)f * P.rows_stride + L.row_off) * AKD_MAXCHUNKS + ch;
    // row y of the window: centre value, max(left, right), 3-wide maximum
    auto reduce_row = [&](float c, float e, float &side, float &m3) {  // e: the edge column (lanes 0 and 63 only)
        const float dl = akm_from_left(c), dr = akm_from_right(c);
        const float l = lane == 0 ? e : dl, rr = lane == 63 ? e : dr;
        side = fmaxf(l, rr);
        m3 = fmaxf(side, c);
    };
    // 32-bit byte offsets into the frame's plane (scalar row part + the lane's column part; the plane is far below 4 GB): the frame base
    // stays a scalar pointer, a row costs one scalar multiply and the loads one vector add each
    const char *ldb = reinterpret_cast<const char *>(ld);
    const uint32_t w4 = (uint32_t)L.w * 4u;
    auto row_off = [&](int y) { return (uint32_t)min(max(y, 0), L.h - 1) * w4; };
    auto at = [&](uint32_t roff, uint32_t xoff) { return *reinterpret_cast<const float *>(ldb + (roff + xoff)); };
    const int xe = lane == 0 ? xl : xr;
    const uint32_t xc4 = (uint32_t)xc * 4u, xe4 = (uint32_t)xe * 4u;
    float c_m, side_m, m3_u, m3_m;
    {
        const uint32_t r0 = row_off(ty0 - 1), r1 = row_off(ty0);
        const float c0 = at(r0, xc4), e0 = at(r0, xe4), c1 = at(r1, xc4), e1 = at(r1, xe4);
        float side_u;
        reduce_row(c0, e0, side_u, m3_u);
        reduce_row(c1, e1, side_m, m3_m);
        c_m = c1;
    }
    const int rows = min(AKM_ROWS, L.h - ty0);
    constexpr int G = AKM_G;  // rows fetched together: a wavefront keeps G full lines in flight
    for (int g0 = 0; g0 < rows; g0 += G) {
        float cd[G], ed[G];
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const uint32_t rp = row_off(ty0 + g0 + k + 1);
            cd[k] = at(rp, xc4);
            ed[k] = at(rp, xe4);
        }
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int iy = ty0 + g0 + k;
            float side_d, m3_d;
            reduce_row(cd[k], ed[k], side_d, m3_d);
            const float v = c_m;
            // v > every neighbour and > dthreshold as ONE comparison against their maximum (the planes hold no NaN: the determinant of
            // finite derivatives of an image in [0, 1]); five compares, each with its own mask to AND on the scalar unit, were as many
            // scalar as vector instructions in a kernel whose scalar unit is the busier one (0.86 against 0.64, SQ counters)
            const float nb = fmaxf(fmaxf(side_m, m3_u), fmaxf(m3_d, P.dthreshold));
            bool ok = col_ok && iy >= 1 && iy < L.h - 1 && v > nb && v >= P.min_dthreshold;
            if (ok) {
                const float py = (float)iy;
                ok = !(akd_fround(py - r) - 1 < 0 || akd_fround(py + r) + 1 >= L.h);
            }
            const unsigned long long m = __ballot(ok);
            if (lane == 0 && g0 + k < rows) mk[(size_t)iy * AKD_MAXCHUNKS] = m;
            m3_u = m3_m; c_m = cd[k]; side_m = side_d; m3_m = m3_d;
        }
    }
}

// pass B: one workgroup per (level, frame): row counts from the bitmap, exclusive scan over the rows, then every wavefront
// expands its rows in raster order (lane = 64-column chunk, each lane walks the bits of its own word), and a last flat pass
// fetches |response| for the emitted indices (independent loads; fetched inside the expansion they serialised it)
#define AKE_T 1024
__global__ __launch_bounds__(AKE_T) void k_akz_cand_emit(AkdParams P, const unsigned long long *__restrict__ mask, int *__restrict__ row_start,
                                                       unsigned short *__restrict__ wpre, int *__restrict__ cand, float *__restrict__ cand_resp,
                                                       int *__restrict__ cand_count, int *__restrict__ status) {
    __shared__ int s_w[AKE_T / 64];
    const int level = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const AkdLevel L = P.lv[level];
    const int nchunks = (L.w + 63) >> 6;
    const unsigned long long *mk = mask + ((size_t)f * P.rows_stride + L.row_off) * AKD_MAXCHUNKS;
    int *rs = row_start + (size_t)f * P.rows_stride + L.row_off;
    const int per = (L.h + AKE_T - 1) / AKE_T, b = min(tid * per, L.h), e = min(L.h, b + per);
    int sum = 0;
    for (int r = b; r < e; ++r) {
        int c = 0;
        const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(mk + (size_t)r * AKD_MAXCHUNKS);  // 16-byte loads, all in flight
#pragma unroll
        for (int k = 0; k < AKD_MAXCHUNKS / 2; ++k) {
            if (2 * k < nchunks) {
                const ulonglong2 v = w2[k];
                c += __popcll(v.x) + (2 * k + 1 < nchunks ? __popcll(v.y) : 0);
            }
        }
        rs[r] = c;  // row count for now
        sum += c;
    }
    const int incl = afv_wave_incl_scan(sum);
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int acc = incl - sum, total = 0;
#pragma unroll
    for (int w = 0; w < AKE_T / 64; ++w) {
        const int t = s_w[w];
        if (w < wv) acc += t;
        total += t;
    }
    if (tid == 0) {
        cand_count[f * 16 + level] = min(total, L.cand_cap);
        if (total > L.cand_cap) atomicExch(status, 1);
    }
    for (int r = b; r < e; ++r) {
        const int c = rs[r];
        rs[r] = acc;
        acc += c;
    }
    __threadfence_block();
    __syncthreads();
    const float *ld = L.ldet + (size_t)f * L.w * L.h;
    int *co = cand + (size_t)f * P.cand_stride + L.cand_off;
    float *cr = cand_resp + (size_t)f * P.cand_stride + L.cand_off;
    constexpr int ER = 4;  // rows a wavefront fetches together (mask words + row offsets: one memory round trip for four rows)
    for (int r0 = wv * ER; r0 < L.h; r0 += ER * (AKE_T / 64)) {
        // lane k holds chunk k's word; exclusive scan of the popcounts gives every chunk its offset inside the row
        unsigned long long mw[ER];
        int ro[ER];
#pragma unroll
        for (int u = 0; u < ER; ++u) {
            const int r = min(r0 + u, L.h - 1);
            mw[u] = (lane < nchunks && r0 + u < L.h) ? mk[(size_t)r * AKD_MAXCHUNKS + lane] : 0ull;
            ro[u] = rs[r];
        }
#pragma unroll
        for (int u = 0; u < ER; ++u) {
            unsigned long long m = mw[u];
            const int cnt = __popcll(m);
            const int in = afv_wave_incl_scan(cnt);
            int o = ro[u] + in - cnt;
            // candidates of the row in front of this word: the fixed-point engine turns a bitmap position into a candidate index with it
            if (wpre && lane < nchunks && r0 + u < L.h)
                wpre[((size_t)f * P.rows_stride + L.row_off + r0 + u) * AKD_MAXCHUNKS + lane] = (unsigned short)(in - cnt);
            const int idx0 = (r0 + u) * L.w + (lane << 6);
            while (m) {
                const int bit = (int)__builtin_ctzll(m);
                m &= m - 1;
                if (o < L.cand_cap) co[o] = idx0 + bit;
                ++o;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    const int ntot = min(total, L.cand_cap);
    // |response| of the emitted indices: independent gathers, four in flight per thread
    for (int k0 = tid; k0 < ntot; k0 += 4 * AKE_T) {
        int ci[4];
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ci[u] = k0 + u * AKE_T < ntot ? co[k0 + u * AKE_T] : 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld[ci[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * AKE_T < ntot) cr[k0 + u * AKE_T] = fabsf(v[u]);
    }
}

// ---------------- ordered suppression + upper-level filter: one workgroup per (frame, level), levels pipelined ----------------
// Upstream runs the levels one after the other.  A level-c candidate q only looks at (and only changes) list entries of level
// c-1 / c inside its radius r_c, and a level-(c-1) candidate p only looks at / changes entries inside r_(c-1) of p.  So q sees
// the final state as soon as every level-(c-1) candidate on rows <= y_q + r_c + r_(c-1) is committed, and nothing q does can be
// seen by the level-(c-1) candidates that are still to come (they sit further down).  The eight levels of a frame therefore
// run as a software pipeline of eight workgroups.  What keeps the result identical to the serial loop:
//   * slots: level c appends into its own slot range [sum of the candidate counts of the levels below, + its own count), so
//     slot order == upstream's insertion order whatever the interleaving;
//   * one cell grid per level: grid c holds the entries whose level is c.  Only workgroup c appends to it (list lengths in its
//     LDS).  A list element is {x, y, response, tag}: tag = slot | launch epoch << 17 | dead << 31.  A reader that does not own
//     the grid trusts an element only if its epoch matches (the host clears the grids when the 14-bit epoch wraps, so a
//     matching epoch means "written by this launch"); the owner also publishes its list lengths, but only as a hint that keeps
//     the reader away from empty cells (whose lines would come cold from HBM): a length that is ahead of the elements the
//     reader can see, or left over from the previous launch, only makes it look at elements whose epoch does not match -
//     by the argument above never one it needs.
//   * hand-off (cdna_hip_programming.md §6 Guideline 16): the producer's waves drain their stores, barrier, then ONE lane
//     issues an agent-scope release fence and stores the number of committed candidates (relaxed, agent scope); the consumer's
//     communication lane polls that number relaxed, issues ONE agent-scope acquire fence, barrier, plain loads.  Correct for any
//     placement of the workgroups; the ticket order below only makes it fast (one frame's levels share an XCD's L2).
//   * no dispatch-order assumption: a workgroup draws a ticket from its XCD's counter (falling over to the other XCDs' counters
//     if its own list is used up); a list holds the levels of the frames x, x + 8, ... in groups of G <= 8 frames, level-major
//     inside a group (decoding below).  Whoever holds ticket t is running, and only ever waits for tickets t - G and t - 2G of
//     the same list (the two levels below of the same frame), whose holders started earlier.
// A replaced entry is not unlinked: its old list element is marked dead and a fresh element goes into the list of its new cell
// (in the replacing candidate's level grid), so every list only grows and all commits of a round run in parallel.

#define AKD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define AKD_T 1024
#define AKD_COMM (AKD_T - 64)    // the lane that talks to the neighbouring levels (first lane of the last wavefront)
#define AKD_PAIRS 8   // 2 grids x the (at most) 2 x 2 cells a candidate's disc overlaps
#ifndef AKD_R
#define AKD_R 256      // candidates per speculative round (AKD_R / 64 wavefronts decide / commit); AKD_R * AKD_PAIRS / AKD_T pairs per thread.
                       // 128 -> 256 (round 4): 5.56 -> 5.49 ms per 64-frame step; the static LDS arrays below grow to ~10 KB (akaze_api's gate)
#endif
#define AKD_CPER (AKD_T / (AKD_R / 2))  // threads per row pair of the conflict test
#define AKD_DEADBIT 0x80000000u
#define AKD_SLOTMASK 0x1ffffu   // 17 bits: entry_cap <= 131072
#define AKD_EPOCH_MAX 0x3fffu
#define AKD_NONE 0xffffffffffffffffull
#define AKD_SPIN_LIMIT (1 << 21)  // x ~0.5 us: about a second, not forever

__device__ __forceinline__ unsigned akd_tag_epoch(unsigned w) { return (w >> 17) & AKD_EPOCH_MAX; }

// one lane: wait (relaxed polls) until *p >= need; -1 after AKD_SPIN_LIMIT tries.  The caller issues the acquire fence.
__device__ __forceinline__ int akd_poll(const int *p, int need) {
    for (int it = 0; it < AKD_SPIN_LIMIT; ++it) {
        const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v >= need) return v;
        __builtin_amdgcn_s_sleep(8);
    }
    return -1;
}
// one lane, after a barrier behind which every wave has drained its stores: publish `v`
__device__ __forceinline__ void akd_publish(int *p, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // restated where the compiler cannot drop it (ROCm 7.2)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the (at most 2 x 2) cells of grid G a disc of radius rq around (x, y) overlaps: x0 | y0 << 16 | (x1 - x0) << 30 | (y1 - y0) << 31
__device__ __forceinline__ unsigned akd_box(const AkdLevel &G, float x, float y, float rq) {
    const int x0 = min((int)(fmaxf(x - rq, 0.f) * G.ginv), G.gw - 1), x1 = min((int)((x + rq) * G.ginv), G.gw - 1);
    const int y0 = min((int)(fmaxf(y - rq, 0.f) * G.ginv), G.gh - 1), y1 = min((int)((y + rq) * G.ginv), G.gh - 1);
    return (unsigned)x0 | ((unsigned)y0 << 16) | ((unsigned)(x1 > x0) << 30) | ((unsigned)(y1 > y0) << 31);
}
__device__ __forceinline__ int akd_box_y0(unsigned b) { return (int)((b >> 16) & 0x3fffu); }
__device__ __forceinline__ int akd_box_y1(unsigned b) { return (int)((b >> 16) & 0x3fffu) + (int)(b >> 31); }

// first match of (qx, qy) in one cell list: smallest slot among the live elements within the radius; key = slot << 32 | element index
__device__ __forceinline__ void akd_consider(const uint4 v, unsigned idx, float qx, float qy, float size2, unsigned long long &best) {
    if (v.w & AKD_DEADBIT) return;  // replaced earlier: the entry lives on in another list
    const float dx = qx - __uint_as_float(v.x), dy = qy - __uint_as_float(v.y);
    if (dx * dx + dy * dy <= size2) {
        const unsigned long long key = ((unsigned long long)(v.w & AKD_SLOTMASK) << 32) | idx;
        best = key < best ? key : best;
    }
}

__global__ __launch_bounds__(AKD_T, 8) void k_akz_suppress(AkdParams P, AkdState S, int nframes, const int *__restrict__ cand,
                                                        const float *__restrict__ cand_resp, const int *__restrict__ cand_count,
                                                        const int *__restrict__ row_start, int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char akd_smem[];
    __shared__ int s_item;
    const int NL = P.nlevels;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        const int share = (int)gridDim.x / 8;  // items per XCD list
#ifdef AFV_AKZ_MIX_XCD  // stress build (tools/experiments.py): every list is served by workgroups of ALL XCDs - the hand-off must not care
        const int xcc = ((int)blockIdx.x >> 3) & 7;
#else
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u);  // HW_REG_XCC_ID[3:0]
#endif
        int item = -1;
        for (int k = 0; k < 8 && item < 0; ++k) {
            const int x = (xcc + k) & 7;
            if (__hip_atomic_load(S.ticket + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= share) continue;
            const int t = atomicAdd(S.ticket + x, 1);
            if (t < share) item = t * 8 + x;
        }
        s_item = item;
    }
    __syncthreads();
    if (s_item < 0) return;
    // ticket t of list x: the list's frames (x, x + 8, ...) in groups of G, level-major inside a group: (frame 0, level 0), (frame 1,
    // level 0), ..., (frame G-1, level 0), (frame 0, level 1), ...  The level below is G tickets back.  (Workgroups that start one
    // after the other tend to land on different CUs and the k-th and (k + #CUs)-th on the same one: level-major order puts a busy low
    // level and a mostly waiting high level together - speed only.)
    const int nfx = (nframes + 7) / 8, G = min(8, nfx);
    const int t_ = s_item >> 3, r_ = t_ % (G * NL);
    const int f = ((t_ / (G * NL)) * G + r_ % G) * 8 + (s_item & 7), c = r_ / G;
    if (f >= nframes) return;
    const AkdLevel L = P.lv[c];
    const AkdLevel Lp = P.lv[c > 0 ? c - 1 : 0];  // the level below
    const int ncells = L.gw * L.gh, pcells = c > 0 ? Lp.gw * Lp.gh : 0;
    unsigned char *s_cnt = reinterpret_cast<unsigned char *>(akd_smem);  // [ncells] list lengths of this level's grid ...
    unsigned int *s_cnt32 = reinterpret_cast<unsigned int *>(akd_smem);  // ... four to a word for the LDS atomics
    unsigned char *s_pcnt = s_cnt + ((ncells + 15) & ~15);               // [pcells] length hints of the grid below, refreshed row-wise
    __shared__ unsigned long long s_best[AKD_R];  // per candidate of the round: slot << 32 | element index (min = first match)
    __shared__ float s_sx[AKD_R], s_sy[AKD_R];
    __shared__ unsigned s_boxo[AKD_R], s_boxp[AKD_R];  // cells the disc overlaps in this level's grid / the grid below (akd_box)
    __shared__ float4 s_moved[AKD_R];  // what a candidate changes if it commits: {new entry x, y, replaced entry's old x, y}, 1e30 where nothing
    __shared__ int s_apps[AKD_R / 64], s_napp[AKD_R / 64];  // per deciding wavefront: appends, committed appends
    __shared__ int s_stop[2];                                // first candidate of the round that must be redone (by round parity)
    __shared__ int s_reacq;                                  // the communication lane had to wait (and acquire) this round
    float4 *entry = S.entry + (size_t)f * P.entry_cap;
    uint4 *cells = S.cells + (size_t)f * P.gelems;
    int *gcnt_own = S.gcnt + (size_t)f * P.gcells + L.gcell_off;
    const int *gcnt_prev = S.gcnt + (size_t)f * P.gcells + Lp.gcell_off;
    int *prog_own = S.ticket + 8 + f * 16 + c;
    const int *prog_prev = prog_own - 1;
    unsigned char *keep = S.keep + (size_t)f * P.entry_cap;
    const unsigned epoch = S.epoch;
    const int n = cand_count[f * 16 + c];
    int slot_base = 0, n_prev = 0, n_prev2 = 0;
    for (int k = 0; k < c; ++k) {
        n_prev2 = n_prev;
        n_prev = cand_count[f * 16 + k];
        slot_base += n_prev;
    }
    for (int i = tid; i < (ncells + 3) / 4; i += AKD_T) s_cnt32[i] = 0;
    for (int i = tid; i < pcells; i += AKD_T) s_pcnt[i] = 0;
    for (int i = tid; i < ncells; i += AKD_T) gcnt_own[i] = 0;
    if (tid < AKD_R / 64) s_napp[tid] = 0;
    __syncthreads();
    // rows of the level below that must be final before a candidate on (scaled) row y may look: y + r_c + r_(c-1), one row spare
    const int *rs_prev = row_start + (size_t)f * P.rows_stride + Lp.row_off;
    const int *cd = cand + (size_t)f * P.cand_stride + L.cand_off;
    const float *cr = cand_resp + (size_t)f * P.cand_stride + L.cand_off;
    const float reach = L.psize + Lp.psize + 2.0f * Lp.ratio;
    const float size2 = L.psize * L.psize;
    const float rq = L.psize * 1.0001f + 1e-3f;  // what "dx * dx + dy * dy <= size2" can reach in float arithmetic, with room
    const bool has_reader = c + 1 < NL;  // the last level's progress is nobody's business
    int nE = 0;       // appends of this level so far
    int pos = 0, par = 0;
    int seen = 0;      // (communication lane) progress of the level below as last acquired; -1: gave up
    int published = 0; // (communication lane)
    int pub_cell = -1;  // cell whose list this thread extended in the previous round: its length is published one barrier later
    int pf_pos = -1, pf_idx = 0;
    float pf_resp = 0.f;
#ifdef AFV_AKZ_STATS
    const long long st_t0 = wall_clock64();
    long long st_p[6] = {0, 0, 0, 0, 0, 0};
    int st_rounds = 0;
#endif
    while (pos < n) {
        const int nround = min(AKD_R, n - pos);
#ifdef AFV_AKZ_STATS
        ++st_rounds;
        const long long ph0 = wall_clock64();
#endif
        // ---- 1. the round's candidates (first AKD_R threads); the next round's are prefetched while this one is scanned.
        //         Meanwhile the communication lane makes sure the level below is far enough ahead of the round's last candidate.
        //         There is no barrier between a round's commit and this point: the stores of the commit drain here. ----
        float sx = 0, sy = 0, resp = 0;
        int ci = 0;  // the candidate's own cell
        const bool act = tid < nround;
        if (tid < AKD_R) {
            if (act) {
                int idx;
                if (pf_pos == pos) {
                    idx = pf_idx;
                    resp = pf_resp;
                } else {
                    idx = cd[pos + tid];
                    resp = cr[pos + tid];
                }
                const int iy = idx / L.w, jx = idx - iy * L.w;
                sx = (float)jx * L.ratio;
                sy = (float)iy * L.ratio;
                ci = min((int)(sy * L.ginv), L.gh - 1) * L.gw + min((int)(sx * L.ginv), L.gw - 1);
                s_boxo[tid] = akd_box(L, sx, sy, rq);
                s_boxp[tid] = akd_box(Lp, sx, sy, rq);
            }
            s_sx[tid] = sx; s_sy[tid] = sy;
            s_best[tid] = AKD_NONE;
            if (tid == 0) s_stop[par] = nround;
            pf_pos = pos + AKD_R;  // valid if this round commits all of its candidates (the common case)
            if (pf_pos + tid < n) {
                pf_idx = cd[pf_pos + tid];
                pf_resp = cr[pf_pos + tid];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the previous round's stores have left
        } else if (tid == AKD_COMM && c > 0) {
            const int row = (int)(((float)(cd[pos + nround - 1] / L.w) * L.ratio + reach) / Lp.ratio) + 1;  // first row of the level below that may still be open
            const int need = row >= Lp.h ? n_prev : min(rs_prev[row], n_prev);
            s_reacq = (seen >= 0 && seen < need) ? 1 : 0;
            if (seen >= 0 && seen < need) {
                // waiting costs an acquire (this CU's L1 is dropped): ask for two rounds more than needed so that a level running
                // in lock-step with the one below does not pay it every round.  Here and only here: no other wavefront has a
                // grid load in flight between a round's commit and the barrier below.
                seen = akd_poll(prog_prev, min(n_prev, need + 2 * AKD_R));
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (seen < 0) atomicExch(status, 5);  // the level below never got there: report and run on without waiting
            }
        }
        // length hints of the cell rows (grid below) this round can touch: raster order, first .. last candidate.  Read before the
        // barrier, i.e. possibly before the communication lane's acquire - then they are read again behind it.  Agent-scope
        // loads (L2-served): these loads may be in flight while that acquire drops the L1, and nothing stale may settle there.
        int h0 = 0, h1 = 0;
        if (c > 0) {
            const float yf = (float)(cd[pos] / L.w) * L.ratio, yl = (float)(cd[pos + nround - 1] / L.w) * L.ratio;
            h0 = min((int)(fmaxf(yf - rq, 0.f) * Lp.ginv), Lp.gh - 1) * Lp.gw;
            h1 = (min((int)((yl + rq) * Lp.ginv), Lp.gh - 1) + 1) * Lp.gw;
            for (int i = h0 + tid; i < h1; i += AKD_T)
                s_pcnt[i] = (unsigned char)min(__hip_atomic_load(gcnt_prev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 255);
        }
        __syncthreads();
#ifdef AFV_AKZ_STATS
        const long long ph1 = wall_clock64();
#endif
#pragma unroll
        for (int w = 0; w < AKD_R / 64; ++w) nE += s_napp[w];  // the previous round's appends
        // every commit of the previous round is behind the barrier: the lengths of the lists it extended can go out (plain
        // stores; drained before the barrier after the scan, published with the round's count)
        if (tid < AKD_R && pub_cell >= 0) {
            gcnt_own[pub_cell] = s_cnt[pub_cell];
            pub_cell = -1;
        }
        if (c > 0 && s_reacq) {
            for (int i = h0 + tid; i < h1; i += AKD_T)
                s_pcnt[i] = (unsigned char)min(__hip_atomic_load(gcnt_prev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 255);
            __syncthreads();
        }
        // ---- 2. neighbourhood scan: one (candidate, cell) pair per thread ----
        for (int t = tid; t < nround * AKD_PAIRS; t += AKD_T) {
            const int q = t >> 3, k = t & 7;
            const bool below = k < 4;
            if (below && c == 0) continue;
            const unsigned box = below ? s_boxp[q] : s_boxo[q];
            if (((k & 1) && !((box >> 30) & 1u)) || ((k & 2) && !(box >> 31))) continue;
            const float qx = s_sx[q], qy = s_sy[q];
            unsigned long long best = AKD_NONE;
            // one 16-byte load per element: a wavefront's 64 lanes look at 64 different lists, and the lists are short
            if (below) {  // hinted length, every element checked against the epoch
                const int cell = (akd_box_y0(box) + ((k >> 1) & 1)) * Lp.gw + (int)(box & 0xffffu) + (k & 1);
                const int cn = min((int)s_pcnt[cell], Lp.gcap);
                const unsigned lb = (unsigned)Lp.gelem_off + (unsigned)cell * Lp.gcap;
                for (int e = 0; e < cn; ++e) {
                    const uint4 v = cells[lb + e];
                    if (akd_tag_epoch(v.w) != epoch) break;
                    akd_consider(v, lb + e, qx, qy, size2, best);
                }
            } else {
                const int cell = (akd_box_y0(box) + ((k >> 1) & 1)) * L.gw + (int)(box & 0xffffu) + (k & 1);
                const int cn = min((int)s_cnt[cell], L.gcap);
                const unsigned lb = (unsigned)L.gelem_off + (unsigned)cell * L.gcap;
                for (int e = 0; e < cn; ++e) akd_consider(cells[lb + e], lb + e, qx, qy, size2, best);
            }
            if (best != AKD_NONE) atomicMin(&s_best[q], best);
        }
        if (tid < AKD_R) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the published lengths have left
        __syncthreads();
        // communication lane: everything before this round is committed and drained.  The L2 write-back is started here and
        // runs behind the decision phases; the count is stored once it is through.
        const bool publish = tid == AKD_COMM && has_reader && pos > published;
        if (publish) asm volatile("buffer_wbl2 sc1" ::: "memory");
#ifdef AFV_AKZ_STATS
        const long long ph2 = wall_clock64();
#endif
        // ---- 3. decisions, exact conflict test against the earlier candidates of the round, commit ----
        // 3a. decisions (first AKD_R threads): the first match is re-read from its list (response, position, tag)
        int first = -1, type = 0;  // type: 0 drop, 1 append, 2 replace `first`
        unsigned mloc = 0, mtag = 0;
        if (tid < AKD_R) {
            const unsigned long long b = s_best[tid];
            float oex = 0, oey = 0;
            if (act) {
                if (b == AKD_NONE) {
                    type = 1;
                } else {
                    first = (int)(b >> 32);
                    mloc = (unsigned)(b & 0xffffffffu);
                    const uint4 v = cells[mloc];
                    mtag = v.w;
                    if (resp > __uint_as_float(v.z)) {
                        type = 2;
                        oex = __uint_as_float(v.x);
                        oey = __uint_as_float(v.y);
                    }
                }
            }
            s_moved[tid] = make_float4(type != 0 ? sx : 1e30f, sy, type == 2 ? oex : 1e30f, oey);
            const unsigned long long apm = __ballot(type == 1);
            if (lane == 0) s_apps[tid >> 6] = __popcll(apm);
        }
        // LDS-only barrier (round 4): what crosses it is s_moved / s_apps.  __syncthreads() also drains the vector-memory counter of
        // every wavefront - here that is the communication lane's L2 write-back, which is meant to run BEHIND the decision phases
        AKD_LDS_BARRIER();
#ifdef AFV_AKZ_STATS
        const long long ph3 = wall_clock64();
#endif
        // 3b. a candidate's decision stands unless an earlier candidate of the round changes what its search sees: a new /
        //     moved entry inside its radius, or a replaced entry that used to lie inside its radius.  The round commits up to
        //     the first candidate that was hit.  Rows i and AKD_R - 1 - i of the (j < i) triangle have AKD_R - 1 pairs together:
        //     AKD_CPER threads per such row pair, every thread the same number of pairs, one LDS read per pair.
        {
            const int rp = tid / AKD_CPER, part = tid % AKD_CPER;  // AKD_R / 2 row pairs
            const int i1 = rp, i2 = AKD_R - 1 - rp;
            const float x1 = s_sx[i1], y1 = s_sy[i1], x2 = s_sx[i2], y2 = s_sy[i2];
            bool hit1 = false, hit2 = false;
#pragma unroll 8
            for (int it = 0; it < (AKD_R + AKD_CPER - 2) / AKD_CPER; ++it) {  // several LDS reads in flight together
                const int m = part + AKD_CPER * it;
                if (m >= AKD_R - 1) break;
                const bool lo = m < i1;
                const float4 o = s_moved[lo ? m : m - i1];
                const float xi = lo ? x1 : x2, yi = lo ? y1 : y2;
                const float dx = xi - o.x, dy = yi - o.y, ux = xi - o.z, uy = yi - o.w;
                const bool hit = (dx * dx + dy * dy <= size2) || (ux * ux + uy * uy <= size2);
                hit1 = hit1 || (hit && lo);
                hit2 = hit2 || (hit && !lo);
            }
            if (hit1 && i1 < nround) atomicMin(&s_stop[par], i1);
            if (hit2 && i2 < nround) atomicMin(&s_stop[par], i2);
        }
#ifdef AFV_AKZ_STATS
        const long long ph3a = wall_clock64();
#endif
        AKD_LDS_BARRIER();  // s_stop (LDS atomics) is all that crosses; the write-back is waited for right below, by its own lane only
#ifdef AFV_AKZ_STATS
        const long long ph3b = wall_clock64();
#endif
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(prog_own, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            published = pos;
        }
        const int stop = s_stop[par];
        // 3c. commit (first AKD_R threads = AKD_R / 64 wavefronts).  No barrier follows: the next round's first phase touches
        //     nothing this one still reads (s_stop alternates), and s_napp is read behind the next barrier.
        if (tid < AKD_R) {
            const bool commit = act && tid < stop && type != 0;
            const unsigned long long am = __ballot(commit && type == 1);
            if (commit) {
                int slot;
                if (type == 1) {
                    // appends of the earlier deciding wavefronts all commit when this wavefront commits anything
                    int before = 0;
                    for (int w = 0; w < (tid >> 6); ++w) before += s_apps[w];
                    slot = slot_base + nE + before + __popcll(am & ((1ull << lane) - 1ull));
                } else {
                    slot = first;
                    cells[mloc].w = mtag | AKD_DEADBIT;
                }
                if (slot < P.entry_cap) {
                    entry[slot] = make_float4(sx, sy, resp, __int_as_float(c));
                    // list slot from an LDS atomic on the packed 8-bit counters (several commits may share a cell)
                    const unsigned int old = atomicAdd(&s_cnt32[ci >> 2], 1u << ((ci & 3) * 8));
                    const int cn = (int)((old >> ((ci & 3) * 8)) & 0xffu);
                    if (cn < L.gcap) {
                        cells[(unsigned)L.gelem_off + (unsigned)ci * L.gcap + cn] =
                            make_uint4(__float_as_uint(sx), __float_as_uint(sy), __float_as_uint(resp), (unsigned)slot | (epoch << 17));
                        pub_cell = ci;
                    } else {
                        atomicExch(status, 2);
                    }
                } else {
                    atomicExch(status, 3);
                }
            }
            if (lane == 0) s_napp[tid >> 6] = __popcll(am);
        }
        pos += stop;
        par ^= 1;
#ifdef AFV_AKZ_STATS
        { const long long ph4 = wall_clock64(); st_p[0] += ph1 - ph0; st_p[1] += ph2 - ph1; st_p[2] += ph3 - ph2; st_p[3] += ph4 - ph3; st_p[4] += ph3a - ph3; st_p[5] += ph3b - ph3a; }
#endif
    }
    // the last round's stores and list lengths, then "this level is final"; wait for the two levels below to be final as well
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int w = 0; w < AKD_R / 64; ++w) nE += s_napp[w];
    if (tid < AKD_R && pub_cell >= 0) gcnt_own[pub_cell] = s_cnt[pub_cell];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == AKD_COMM) {
        S.used[f * 16 + c] = nE;
        if (has_reader) akd_publish(prog_own, n);
        if (c > 0 && seen >= 0) {  // a level with few candidates finishes before the one below it
            int v = akd_poll(prog_prev, n_prev);
            if (c > 1 && v >= 0) v = akd_poll(prog_prev - 1, n_prev2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (v < 0) atomicExch(status, 5);
        }
    }
    __syncthreads();
#ifdef AFV_AKZ_STATS
    const long long st_t1 = wall_clock64();
#endif
    // ---- "Now filter points with the upper scale level": entry i of level a is repeated if a LATER entry of level a+1 lies
    //      within size_a and has a larger response.  Grid a holds exactly the level-a entries once level a+1 has run, grid a+1
    //      the level-(a+1) entries once level a+2 has run: pair (c-2, c-1) is closed here, after this level's own rounds; the
    //      last level also closes (c-1, c).
    for (int a = c - 2; a <= (c == NL - 1 ? c - 1 : c - 2); ++a) {
        if (a < 0) continue;
        const AkdLevel A = P.lv[a], B = P.lv[a + 1];
        const float sz2 = A.psize * A.psize, ra = A.psize * 1.0001f + 1e-3f;
        const int *gA = S.gcnt + (size_t)f * P.gcells + A.gcell_off, *gB = S.gcnt + (size_t)f * P.gcells + B.gcell_off;
        for (int cell = tid; cell < A.gw * A.gh; cell += AKD_T) {
            const int cna = min(gA[cell], A.gcap);
            for (int e0 = 0; e0 < cna; ++e0) {
                const uint4 u = cells[(unsigned)A.gelem_off + (unsigned)cell * A.gcap + e0];
                if (akd_tag_epoch(u.w) != epoch) break;
                if (u.w & AKD_DEADBIT) continue;
                const float x = __uint_as_float(u.x), y = __uint_as_float(u.y), r = __uint_as_float(u.z);
                const unsigned i = u.w & AKD_SLOTMASK;
                const unsigned box = akd_box(B, x, y, ra);
                bool rep = false;
                for (int k = 0; k < 4 && !rep; ++k) {
                    if (((k & 1) && !((box >> 30) & 1u)) || ((k & 2) && !(box >> 31))) continue;
                    const int c2 = (akd_box_y0(box) + (k >> 1)) * B.gw + (int)(box & 0xffffu) + (k & 1);
                    const int cnb = min(gB[c2], B.gcap);
                    const unsigned lb = (unsigned)B.gelem_off + (unsigned)c2 * B.gcap;
                    for (int e = 0; e < cnb; ++e) {
                        const uint4 v = cells[lb + e];
                        if (akd_tag_epoch(v.w) != epoch) break;
                        if ((v.w & AKD_DEADBIT) || (v.w & AKD_SLOTMASK) <= i) continue;
                        const float dx = x - __uint_as_float(v.x), dy = y - __uint_as_float(v.y);
                        if (dx * dx + dy * dy <= sz2 && r < __uint_as_float(v.z)) {
                            rep = true;
                            break;
                        }
                    }
                }
                if (rep) keep[i] = 0;
            }
        }
    }
#ifdef AFV_AKZ_STATS
    if (tid == 0 && f == 0)
        printf("akz_suppress level %d: candidates %d rounds %d appends %d | us: rounds %lld upper %lld | per-round parts: load+wait %lld scan %lld decide %lld conflicts+commit %lld (test %lld barrier %lld)\n", c, n,
               st_rounds, nE, (st_t1 - st_t0) / 100, (wall_clock64() - st_t1) / 100, st_p[0] / 100, st_p[1] / 100, st_p[2] / 100, st_p[3] / 100, st_p[4] / 100, st_p[5] / 100);
#endif
}

// ---------------- ordered suppression as a FIXED POINT (round 5): all candidates of all frames at once ----------------
// What upstream's loop does to candidate c (in loop order: level-major, raster inside a level) depends on the list as the candidates
// before it left it: c looks for the FIRST list entry (smallest slot) of its own level or the level below inside its radius; none ->
// c is appended (a new slot); one with a smaller response -> c takes over that slot (the entry now IS c: position, level, response);
// else c is dropped.  An entry is always "the candidate that holds the slot now", so the whole state is a function of what every
// candidate did:   act(c) in {APPEND, DROP, REPLACE(j)},  j = the holder c pushed out.
//   * j holds its slot at time c   <=>  act(j) != DROP and no candidate before c replaced j        (succ(j) = the smallest such, >= c)
//   * slot order = order of the candidates that opened the slots: slot(j) = root(j), root(c) = c if APPEND, root(j) if REPLACE(j)
//   * act(c) = f(the holders at time c among c's EARLIER IN-RANGE candidates): the one with the smallest root decides.
// act(c) only depends on candidates before c, so (induction on c) ANY assignment that satisfies all equations is the loop's outcome -
// and iterating "every candidate re-evaluates f against the current state" reaches it in (longest chain of decisions that depend on
// each other) + 1 passes, in any evaluation order (a pass that changes nothing has evaluated every equation against the final state).
// Strict 3 x 3 maxima lie at least two pixels apart and the radii are 2.4 - 4.8 pixels of the level scanned: a candidate has 0.6 - 0.8
// earlier in-range candidates on average (at most a few) and the chains are short: 8 passes on the bench frames (40 k candidates) and
// on band-limited noise (52 k), where the speculative rounds of k_akz_suppress need ~330 rounds of 5 barriers each on a pipeline of
// eight workgroups per frame.  The in-range candidates are found in the candidate BITMAPS (k_akz_cand_mask), once: no grids, no lists
// that grow, no hand-off between levels, nothing that depends on where or when a workgroup runs.
//   k_akz_fp_build    per candidate: its earlier in-range candidates (bitmap windows of its level and the level below; exact float
//                     test, upstream's expression) as gids; start state APPEND; the candidates that have any form the active list
//   k_akz_fp_pass     one launch per pass over the active candidates of ALL frames (a kernel boundary is the only synchronisation a
//                     pass needs); the succ arrays rotate read / write / clear; a frame whose previous pass changed nothing is done,
//                     its workgroups leave at once.  AKF_PASSES launches are enqueued; a frame that needs more reports status 7.
//   k_akz_fp_entries  the list: slot = root gid (same layout as k_akz_suppress's), keep = "this candidate opened a slot"
//   k_akz_fp_upper    the upper-level filter, again from the bitmap of the level above.
// Both engines leave entry / keep / used for k_akz_refine_a / _b; tests/test_gpu_akaze.py runs every case through both.
#define AKF_T 256
#define AKF_ROWS 12   // bitmap rows fetched together (a window is at most 2 * 4.8 + 2 rows high with the akaze61 radii; more rows: another group)
#define AKF_UP_K 32   // level-above candidates inside one disc of the upper-level filter: strict 3 x 3 maxima lie >= 2 pixels apart, radius <= 4.04
#define AKF_ROWS_UP 6  // ... of the rows up to the candidate's own (same level: radius <= 4.04 rows)

// the candidates of level G (bitmap rows mk) inside the disc (sx, sy, size2): rows up to y_last, on row y_last only the columns up to
// x_last (same level: the candidates in front of c); visit(x | y << 16).  A window is at most 13 columns wide: one or two 64-bit words
// per row, all rows' words in flight together.  Turning a position into the candidate's index costs three more loads (akf_index): the
// callers collect the hits first and resolve them together - inside the row loop every hit was a dependent memory round trip of its own
// (64 lanes, 18 rows: nearly every row has a hit in some lane) and k_akz_fp_build took 0.17 ms per 64 frames.
template <int ROWS, typename F>
__device__ __forceinline__ void akf_scan(const AkdLevel &G, const unsigned long long *__restrict__ mk, float sx, float sy, float rq, float size2,
                                         int y_last, int x_last, F visit) {
    const int xa = max((int)floorf((sx - rq) / G.ratio), 0), xb = min((int)((sx + rq) / G.ratio) + 1, G.w - 1);
    const int ya = max((int)floorf((sy - rq) / G.ratio), 0), yb = min(min((int)((sy + rq) / G.ratio) + 1, G.h - 1), y_last);
    if (xa > xb) return;
    const int w0 = xa >> 6, w1 = min(xb >> 6, w0 + 1);  // (a window wider than 64 columns would need more: not with these radii)
    // the column masks of the window's one or two words are the same for every row but the last (64-bit shifts are slow: once per scan)
    auto col_mask = [&](int wd, int xe) -> unsigned long long {
        if ((wd << 6) > xe) return 0ull;
        const int lo = max(xa - (wd << 6), 0), hi = min(xe - (wd << 6), 63);
        return (~0ull << lo) & (~0ull >> (63 - hi));
    };
    const unsigned long long cm0 = col_mask(w0, xb), cm1 = w1 != w0 ? col_mask(w1, xb) : 0ull;
    const int xl = min(xb, x_last);
    const unsigned long long lm0 = xa <= xl ? col_mask(w0, xl) : 0ull, lm1 = (w1 != w0 && xa <= xl) ? col_mask(w1, xl) : 0ull;
    for (int yg = ya; yg <= yb; yg += ROWS) {
        unsigned long long m0[ROWS], m1[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = min(yg + r, yb);
            m0[r] = yg + r <= yb ? mk[(size_t)y * AKD_MAXCHUNKS + w0] : 0ull;
            m1[r] = (w1 != w0 && yg + r <= yb) ? mk[(size_t)y * AKD_MAXCHUNKS + w1] : 0ull;  // one window in five straddles two words
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int y = yg + r;
            const bool last = y == y_last;
            unsigned long long a = m0[r] & (last ? lm0 : cm0), b = m1[r] & (last ? lm1 : cm1);
            if ((a | b) == 0ull) continue;
            const float ay = (float)y * G.ratio, dy = sy - ay;
            while (a) {
                const int bit = (int)__builtin_ctzll(a);
                a &= a - 1;
                const float ax = (float)((w0 << 6) + bit) * G.ratio, dx = sx - ax;
                if (dx * dx + dy * dy <= size2) visit((unsigned)((w0 << 6) + bit) | ((unsigned)y << 16));
            }
            while (b) {
                const int bit = (int)__builtin_ctzll(b);
                b &= b - 1;
                const float ax = (float)((w1 << 6) + bit) * G.ratio, dx = sx - ax;
                if (dx * dx + dy * dy <= size2) visit((unsigned)((w1 << 6) + bit) | ((unsigned)y << 16));
            }
        }
    }
}

// index of the candidate at bitmap position (x, y) inside its level: candidates of the rows above + of the words in front + of the bits below
struct AkfPos {
    int rs;
    unsigned short wp;
    unsigned long long word;
    int bit;
};
__device__ __forceinline__ AkfPos akf_index_load(const unsigned long long *__restrict__ mk, const int *__restrict__ rs,
                                                 const unsigned short *__restrict__ wp, unsigned code) {
    const int x = (int)(code & 0xffffu), y = (int)((code >> 16) & 0x7fffu);
    AkfPos p;
    p.rs = rs[y];
    p.wp = wp[(size_t)y * AKD_MAXCHUNKS + (x >> 6)];
    p.word = mk[(size_t)y * AKD_MAXCHUNKS + (x >> 6)];
    p.bit = x & 63;
    return p;
}
__device__ __forceinline__ int akf_index(const AkfPos &p) { return p.rs + (int)p.wp + __popcll(p.word & ((1ull << p.bit) - 1ull)); }

__device__ __forceinline__ void akf_bases(const int *__restrict__ cand_count, int f, int NL, int *s_base) {
    if (threadIdx.x == 0) {
        int b = 0;
        for (int k = 0; k < NL; ++k) {
            s_base[k] = b;
            b += cand_count[f * 16 + k];
        }
        for (int k = NL; k < 17; ++k) s_base[k] = b;
    }
    __syncthreads();
}

__global__ __launch_bounds__(AKF_T) void k_akz_fp_build(AkdParams P, AkdState S, const unsigned long long *__restrict__ mask,
                                                        const int *__restrict__ row_start, const int *__restrict__ cand,
                                                        const float *__restrict__ cand_resp, const int *__restrict__ cand_count,
                                                        int *__restrict__ status) {
    __shared__ unsigned s_hit[AKF_K][AKF_T];  // per thread: the positions its scans found (bit 31: in the level below)
    const int c = blockIdx.x, f = blockIdx.y, part = blockIdx.z, nparts = gridDim.z, tid = threadIdx.x, lane = tid & 63;
    const AkdLevel L = P.lv[c];
    const AkdLevel Lp = P.lv[c > 0 ? c - 1 : 0];
    const int n = cand_count[f * 16 + c];
    int base = 0, base_prev = 0;
    for (int k = 0; k < c; ++k) {
        base_prev = base;
        base += cand_count[f * 16 + k];
    }
    if (base + n > P.entry_cap) {
        if (tid == 0) atomicExch(status, 3);
        return;
    }
    const unsigned long long *mk = mask + ((size_t)f * P.rows_stride + L.row_off) * AKD_MAXCHUNKS;
    const unsigned long long *mkp = mask + ((size_t)f * P.rows_stride + Lp.row_off) * AKD_MAXCHUNKS;
    const int *rs = row_start + (size_t)f * P.rows_stride + L.row_off, *rsp = row_start + (size_t)f * P.rows_stride + Lp.row_off;
    const unsigned short *wp = S.wpre + ((size_t)f * P.rows_stride + L.row_off) * AKD_MAXCHUNKS;
    const unsigned short *wpp = S.wpre + ((size_t)f * P.rows_stride + Lp.row_off) * AKD_MAXCHUNKS;
    const int *cd = cand + (size_t)f * P.cand_stride + L.cand_off;
    const float *cr = cand_resp + (size_t)f * P.cand_stride + L.cand_off;
    const size_t fo = (size_t)f * P.entry_cap;
    const float size2 = L.psize * L.psize, rq = L.psize * 1.0001f + 1e-3f;
    for (int kb = part * AKF_T; kb < n; kb += nparts * AKF_T) {  // uniform trip count per workgroup
        const int k = kb + tid;
        bool active = false;
        int my_cnt = 0, n0 = 0, n1 = 0, n2 = 0;
        float my_resp = 0.f;
        if (k < n) {
            const int idx = cd[k], iy = idx / L.w, jx = idx - iy * L.w;
            const float sx = (float)jx * L.ratio, sy = (float)iy * L.ratio;
            const int gid = base + k;
            int *list = S.fp_nbr + (fo + gid) * AKF_K;  // only a candidate with more than three neighbours ever touches its 64-byte list
            int cnt = 0;
            // the FIRST entry in slot order decides, and slot order is the order of the roots: the order inside the list does not matter
            if (c > 0)
                akf_scan<AKF_ROWS>(Lp, mkp, sx, sy, rq, size2, Lp.h - 1, Lp.w - 1, [&](unsigned code) {
                    if (cnt < AKF_K) s_hit[cnt][tid] = code | 0x80000000u;
                    ++cnt;
                });
            akf_scan<AKF_ROWS_UP>(L, mk, sx, sy, rq, size2, iy, jx - 1, [&](unsigned code) {
                if (cnt < AKF_K) s_hit[cnt][tid] = code;
                ++cnt;
            });
            for (int h0 = 0; h0 < min(cnt, AKF_K); h0 += 4) {  // four hits' loads in flight together
                AkfPos ps[4];
                unsigned cd4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    cd4[u] = s_hit[min(h0 + u, AKF_K - 1)][tid];
                    const bool below = (cd4[u] >> 31) != 0;
                    if (h0 + u < cnt) ps[u] = akf_index_load(below ? mkp : mk, below ? rsp : rs, below ? wpp : wp, cd4[u]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int h = h0 + u;
                    if (h < cnt && h < AKF_K) {
                        const int v = ((cd4[u] >> 31) ? base_prev : base) + akf_index(ps[u]);
                        if (h == 0) n0 = v; else if (h == 1) n1 = v; else if (h == 2) n2 = v; else list[h] = v;
                    }
                }
            }
            if (cnt > (S.fp_nbr_cap > 0 ? min(S.fp_nbr_cap, AKF_K) : AKF_K)) atomicExch(status, 6);
            my_cnt = min(cnt, AKF_K);
            my_resp = cr[k];
            S.fp_state[(fo + gid) * 2] = make_int4(AKF_APPEND, gid, __float_as_int(my_resp), 0);
            S.fp_state[(fo + gid) * 2 + 1] = make_int4(-1, -1, 0, 0);  // 0xffffffff: above every stamped value, and no pass has stamp 255
            active = cnt > 0;
        }
        const unsigned long long am = __ballot(active);
        int o = 0;
        if (lane == 0 && am) o = atomicAdd(S.fp_ctl + (size_t)f * AKF_CTL + c, __popcll(am));  // one counter per (frame, level): 8 x fewer collisions
        o = __shfl(o, 0, 64);
        if (active) {  // the candidate's record: what a pass needs of it in one 32-byte load (the first three neighbours inline)
            // level c's records start at its gid base: a level has at most as many active candidates as candidates
            int4 *rec = S.fp_active + (fo + base + o + __popcll(am & ((1ull << lane) - 1ull))) * 2;
            rec[0] = make_int4(base + k, my_cnt, __float_as_int(my_resp), AKF_APPEND);
            rec[1] = make_int4(base + k, n0, n1, n2);
        }
    }
}

// fp_ctl[frame][AKF_CTL]: [0 .. 15] active candidates per level, [16] the pass that found the frame converged (0: not yet), [17 + p] pass p
// changed something
__global__ __launch_bounds__(AKF_T) void k_akz_fp_pass(AkdParams P, AkdState S, const int *__restrict__ cand_count, int pass) {
    __shared__ int s_base[17], s_nact[16];
    const int f = blockIdx.y, tid = threadIdx.x;
    int *ctl = S.fp_ctl + (size_t)f * AKF_CTL;
    if (pass > 0) {
        if (ctl[16] != 0) return;       // converged in an earlier pass
        if (ctl[17 + pass - 1] == 0) {  // the previous pass changed nothing: every equation holds
            if (blockIdx.x == 0 && tid == 0) ctl[16] = pass;
            return;
        }
    }
    if (tid < 16) s_nact[tid] = ctl[tid];
    akf_bases(cand_count, f, P.nlevels, s_base);
    const int total = min(s_base[P.nlevels], P.entry_cap);
    const size_t fo = (size_t)f * P.entry_cap;
    int4 *state = S.fp_state + fo * 2;  // per candidate 32 bytes: {act, root, response, -} {succ of the even passes, of the odd passes, -, -}
    const int *nbr = S.fp_nbr + fo * AKF_K;
    int4 *recs = S.fp_active + fo * 2;
    // succ(j) = the smallest candidate that replaces j, as the passes' atomic minima of  (254 - pass) << 24 | candidate : a later pass's
    // value is smaller than anything an earlier pass left, so nothing is ever cleared; pass p writes word p & 1 and reads the other one,
    // which holds pass p - 1's minima exactly where its stamp says so
    const unsigned want = (unsigned)(254 - (pass - 1)) & 0xffu, stamp = (unsigned)(254 - pass) << 24;
    const int rd = (pass + 1) & 1, wr = pass & 1;
    bool changed = false;
    for (int i = blockIdx.x * AKF_T + tid; i < total; i += gridDim.x * AKF_T) {
        int lv = 0;
        while (i >= s_base[lv + 1]) ++lv;
        if (i - s_base[lv] >= s_nact[lv]) continue;  // behind the level's records
        const int4 r0 = recs[2 * i], r1 = recs[2 * i + 1];  // {gid, count, response, act} {root, n0, n1, n2}
        const int c = r0.x, cn = r0.y;
        const int *list = nbr + (size_t)c * AKF_K;
        int best = -1, best_root = AKF_NONE, best_resp = 0;
        for (int k = 0; k < cn; ++k) {
            const int j = k == 0 ? r1.y : k == 1 ? r1.z : k == 2 ? r1.w : list[k];
            const int4 sj = state[2 * j], tj = state[2 * j + 1];
            const unsigned sv = (unsigned)(rd ? tj.y : tj.x);
            const int su = pass > 0 && (sv >> 24) == want ? (int)(sv & 0xffffffu) : AKF_NONE;
            if (sj.x != AKF_DROP && su >= c && sj.y < best_root) {
                best_root = sj.y;
                best = j;
                best_resp = sj.z;
            }
        }
        int2 nw;
        if (best < 0) nw = make_int2(AKF_APPEND, c);
        else if (__int_as_float(r0.z) > __int_as_float(best_resp)) nw = make_int2(best, best_root);
        else nw = make_int2(AKF_DROP, c);
        if (nw.x != r0.w || nw.y != r1.x) {
            recs[2 * i].w = nw.x;
            recs[2 * i + 1].x = nw.y;
            *reinterpret_cast<int2 *>(state + 2 * c) = nw;
            changed = true;
        }
        if (nw.x >= 0)
            __hip_atomic_fetch_min(reinterpret_cast<unsigned *>(state + 2 * nw.x + 1) + wr, stamp | (unsigned)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (changed) ctl[17 + pass] = 1;
}

// the last executed pass (its succ minima describe the final state); -1: the frame did not converge in the passes that were enqueued
__device__ __forceinline__ int akf_final(const int *ctl, int npass) {
    int q = ctl[16];
    if (q == 0 && ctl[17 + npass - 1] == 0) q = npass;  // the last launch was the pass that changed nothing
    return q - 1;
}
// has the candidate with this state been replaced at the end (last pass: fin)?
__device__ __forceinline__ bool akf_replaced(const int4 &t, int fin) {
    const unsigned sv = (unsigned)((fin & 1) ? t.y : t.x);
    return (sv >> 24) == ((unsigned)(254 - fin) & 0xffu);
}

__global__ __launch_bounds__(AKF_T) void k_akz_fp_entries(AkdParams P, AkdState S, const int *__restrict__ cand, const int *__restrict__ cand_count,
                                                          int npass, int *__restrict__ status) {
    __shared__ int s_base[17];
    const int f = blockIdx.y, tid = threadIdx.x, NL = P.nlevels;
    akf_bases(cand_count, f, NL, s_base);
    const int total = s_base[NL];
    if (total > P.entry_cap) return;  // k_akz_fp_build reported it
    const int *ctl = S.fp_ctl + (size_t)f * AKF_CTL;
    const int fin = akf_final(ctl, npass);
    if (fin < 0) {
        if (blockIdx.x == 0 && tid == 0) atomicExch(status, 7);
        return;
    }
    const size_t fo = (size_t)f * P.entry_cap;
    const int4 *state = S.fp_state + fo * 2;
    float4 *entry = S.entry + fo;
    unsigned char *keep = S.keep + fo;
    if (blockIdx.x == 0 && tid < NL) S.used[f * 16 + tid] = cand_count[f * 16 + tid];
    // slot = root gid: a slot exists for every candidate that appended, its entry is the candidate that holds it at the end
    for (int g = blockIdx.x * AKF_T + tid; g < total; g += gridDim.x * AKF_T) {
        int lv = 0;
        while (g >= s_base[lv + 1]) ++lv;
        const int4 st = state[2 * g];
        keep[g] = st.x == AKF_APPEND ? 1 : 0;
        if (st.x != AKF_DROP && !akf_replaced(state[2 * g + 1], fin)) {
            const AkdLevel &L = P.lv[lv];
            const int idx = cand[(size_t)f * P.cand_stride + L.cand_off + (g - s_base[lv])];
            const int iy = idx / L.w, jx = idx - iy * L.w;
            entry[st.y] = make_float4((float)jx * L.ratio, (float)iy * L.ratio, __int_as_float(st.z), __int_as_float(lv));
        }
    }
}

// "Now filter points with the upper scale level": entry i (holder a, level A) is repeated if a LATER entry (slot > i) of level A + 1
// lies within size_A of it and has a larger response
__global__ __launch_bounds__(AKF_T) void k_akz_fp_upper(AkdParams P, AkdState S, const unsigned long long *__restrict__ mask,
                                                        const int *__restrict__ row_start, const int *__restrict__ cand,
                                                        const int *__restrict__ cand_count, int npass, int *__restrict__ status) {
    __shared__ int s_base[17];
    __shared__ unsigned s_hit[AKF_UP_K][AKF_T];
    const int f = blockIdx.y, tid = threadIdx.x, NL = P.nlevels;
    akf_bases(cand_count, f, NL, s_base);
    const int total = s_base[NL];
    if (total > P.entry_cap) return;
    const int fin = akf_final(S.fp_ctl + (size_t)f * AKF_CTL, npass);
    if (fin < 0) return;
    const size_t fo = (size_t)f * P.entry_cap;
    const int4 *state = S.fp_state + fo * 2;
    unsigned char *keep = S.keep + fo;
    for (int g = blockIdx.x * AKF_T + tid; g < s_base[NL - 1]; g += gridDim.x * AKF_T) {  // the last level has no level above
        int lv = 0;
        while (g >= s_base[lv + 1]) ++lv;
        const int4 st = state[2 * g];
        if (st.x == AKF_DROP || akf_replaced(state[2 * g + 1], fin)) continue;
        const AkdLevel &A = P.lv[lv], &B = P.lv[lv + 1];
        const int idx = cand[(size_t)f * P.cand_stride + A.cand_off + (g - s_base[lv])];
        const int iy = idx / A.w, jx = idx - iy * A.w;
        const float x = (float)jx * A.ratio, y = (float)iy * A.ratio, r = __int_as_float(st.z);
        const unsigned long long *mkb = mask + ((size_t)f * P.rows_stride + B.row_off) * AKD_MAXCHUNKS;
        const int *rsb = row_start + (size_t)f * P.rows_stride + B.row_off;
        const unsigned short *wpb = S.wpre + ((size_t)f * P.rows_stride + B.row_off) * AKD_MAXCHUNKS;
        const int bb = s_base[lv + 1];
        int cnt = 0;
        akf_scan<AKF_ROWS>(B, mkb, x, y, A.psize * 1.0001f + 1e-3f, A.psize * A.psize, B.h - 1, B.w - 1, [&](unsigned code) {
            if (cnt < AKF_UP_K) s_hit[cnt][tid] = code;
            ++cnt;
        });
        bool rep = false;
        for (int h0 = 0; h0 < cnt; h0 += 4) {
            AkfPos ps[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (h0 + u < cnt) ps[u] = akf_index_load(mkb, rsb, wpb, s_hit[min(h0 + u, AKF_UP_K - 1)][tid]);
            int gb[4];
            int4 sb[4], tb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (h0 + u < cnt) {
                    gb[u] = bb + akf_index(ps[u]);
                    sb[u] = state[2 * gb[u]];
                    tb[u] = state[2 * gb[u] + 1];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (h0 + u < cnt && sb[u].x != AKF_DROP && !akf_replaced(tb[u], fin) && sb[u].y > st.y && r < __int_as_float(sb[u].z)) rep = true;
        }
        if (cnt > AKF_UP_K) atomicExch(status, 6);  // more level-above candidates in one disc than 2-pixel-apart maxima can be
        if (rep) keep[st.y] = 0;
    }
}

// ---------------- Do_Subpixel_Refinement + ordered compaction over the slot space, two passes ----------------
// pass A: one thread per slot: level range -> used? kept? -> the 2x2 solve; the refined position goes back into ex / ey, the
//         verdict into keep, the number of survivors of every 1024-slot chunk into chunk_cnt
// pass B: chunk offset = sum of the earlier chunks' counts; survivors written in slot order
#define AKD_CHUNK 1024
__global__ __launch_bounds__(AKD_CHUNK) void k_akz_refine_a(AkdParams P, AkdState S, const int *__restrict__ cand_count) {
    __shared__ int s_wsum[AKD_CHUNK / 64];
    const int f = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int slot = chunk * AKD_CHUNK + tid;
    {   // the slot space is 131 072 wide, a frame uses the first sum-of-candidates of it (~40 k): the chunks behind that have nothing to do
        int slots = 0;
        for (int k = 0; k < P.nlevels; ++k) slots += cand_count[f * 16 + k];
        if (chunk * AKD_CHUNK >= slots) {  // uniform
            if (tid == 0) S.chunk_cnt[(size_t)f * gridDim.x + chunk] = 0;  // k_akz_refine_b then never looks at `keep` of this chunk
            return;
        }
    }
    float4 *entry = S.entry + (size_t)f * P.entry_cap;
    unsigned char *keep = S.keep + (size_t)f * P.entry_cap;
    bool ok = false;
    {
        int base = 0;
        for (int k = 0; k < P.nlevels; ++k) {  // slot ranges: level k appended into [base, base + candidates of k)
            const int nk = cand_count[f * 16 + k];
            if (slot >= base && slot - base < S.used[f * 16 + k]) ok = true;
            base += nk;
        }
    }
    ok = ok && slot < P.entry_cap && keep[slot] != 0;
    if (ok) {
        const float4 en = entry[slot];
        const AkdLevel L = P.lv[__float_as_int(en.w)];
        const int x = akd_fround(en.x / L.ratio), y = akd_fround(en.y / L.ratio), w = L.w;
        const float *D = L.ldet + (size_t)f * L.w * L.h;
#define LD(yy, xx) D[(size_t)(yy) * w + (xx)]
        const float Dx = (float)(0.5 * (double)(LD(y, x + 1) - LD(y, x - 1)));
        const float Dy = (float)(0.5 * (double)(LD(y + 1, x) - LD(y - 1, x)));
        const float Dxx = (float)((double)(LD(y, x + 1) + LD(y, x - 1)) - 2.0 * (double)LD(y, x));
        const float Dyy = (float)((double)(LD(y + 1, x) + LD(y - 1, x)) - 2.0 * (double)LD(y, x));
        const float Dxy = (float)(0.25 * (double)(LD(y + 1, x + 1) + LD(y - 1, x - 1)) - 0.25 * (double)(LD(y - 1, x + 1) + LD(y + 1, x - 1)));
#undef LD
        const double det = (double)Dxx * (double)Dyy - (double)Dxy * (double)Dxy;
        if (det == 0.0) {
            ok = false;
        } else {
            const double b0 = -(double)Dx, b1 = -(double)Dy, inv = 1.0 / det;
            const float d0 = (float)((b0 * (double)Dyy - b1 * (double)Dxy) * inv);
            const float d1 = (float)((b1 * (double)Dxx - b0 * (double)Dxy) * inv);
            if (fabsf(d0) <= 1.0f && fabsf(d1) <= 1.0f) {
                const float power = (float)(1 << L.octave);
                entry[slot] = make_float4(((float)x + d0) * power, ((float)y + d1) * power, en.z, en.w);
            } else {
                ok = false;
            }
        }
    }
    if (slot < P.entry_cap) keep[slot] = ok ? 1 : 0;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) s_wsum[tid >> 6] = __popcll(m);
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int w = 0; w < AKD_CHUNK / 64; ++w) tot += s_wsum[w];
        S.chunk_cnt[(size_t)f * gridDim.x + chunk] = tot;
    }
}

__global__ __launch_bounds__(AKD_CHUNK) void k_akz_refine_b(AkdParams P, AkdState S, afv_keypoint *__restrict__ kps, int *__restrict__ kp_count,
                                                            int *__restrict__ status) {
    __shared__ int s_wsum[AKD_CHUNK / 64], s_base, s_total;
    const int f = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x, tid = threadIdx.x, lane = tid & 63;
    const int slot = chunk * AKD_CHUNK + tid;
    const int *cc = S.chunk_cnt + (size_t)f * nchunks;
    if (tid < 64) {  // nchunks <= 256: four per lane
        int before = 0, all = 0;
        for (int j = lane; j < nchunks; j += 64) {
            const int v = cc[j];
            all += v;
            if (j < chunk) before += v;
        }
        before = afv_wave_incl_scan(before);
        all = afv_wave_incl_scan(all);
        if (lane == 63) {
            s_base = before;
            s_total = all;
        }
    }
    // a chunk without survivors (k_akz_refine_a counted them; the chunks behind the used slot range never set `keep`) has nothing to write
    const bool ok = cc[chunk] > 0 && slot < P.entry_cap && S.keep[(size_t)f * P.entry_cap + slot] != 0;
    const unsigned long long m = __ballot(ok);
    if (lane == 0) s_wsum[tid >> 6] = __popcll(m);
    __syncthreads();
    if (chunk == 0 && tid == 0) {
        kp_count[f] = min(s_total, P.kp_cap);
        if (s_total > P.kp_cap) atomicExch(status, 4);
    }
    if (!ok) return;
    int o = s_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < (tid >> 6); ++w) o += s_wsum[w];
    if (o >= P.kp_cap) return;
    const size_t si = (size_t)f * P.entry_cap + slot;
    const float4 en = S.entry[si];
    const int klev = __float_as_int(en.w);
    const AkdLevel L = P.lv[klev];
    afv_keypoint k;
    k.x = en.x; k.y = en.y; k.size = L.psize * 2.0f; k.angle = 0.0f; k.response = en.z; k.octave = L.octave; k.class_id = klev;
    kps[(size_t)f * P.kp_cap + o] = k;
}

extern "C" void afv_akz_launch_candidates(const AkdParams *P, int nframes, unsigned long long *mask, int *row_start, unsigned short *wpre, int *cand,
                                          float *cand_resp, int *cand_count, int *status, hipStream_t st) {
    AkmWork W;
    int strips = 0;
    for (int l = 0; l < P->nlevels; ++l) {
        W.strip_off[l] = strips;
        strips += ((P->lv[l].w + 63) / 64) * ((P->lv[l].h + AKM_ROWS - 1) / AKM_ROWS) * nframes;
    }
    for (int l = P->nlevels; l < 17; ++l) W.strip_off[l] = strips;
    hipLaunchKernelGGL(k_akz_cand_mask, dim3((strips + 3) / 4), dim3(256), 0, st, *P, W, nframes, mask);
    hipLaunchKernelGGL(k_akz_cand_emit, dim3(P->nlevels, nframes), dim3(AKE_T), 0, st, *P, mask, row_start, wpre, cand, cand_resp, cand_count, status);
}

extern "C" void afv_akz_launch_suppress(const AkdParams *P, const AkdState *S, int nframes, int engine, const unsigned long long *mask, const int *cand,
                                        const float *cand_resp, const int *cand_count, const int *row_start, afv_keypoint *kps, int *kp_count,
                                        int *status, hipStream_t st) {
    if (engine == 1) {
        const int npass = S->fp_pass_cap > 0 ? (S->fp_pass_cap < AKF_PASSES ? S->fp_pass_cap : AKF_PASSES) : AKF_PASSES;
        // S->fp_ctl was cleared together with the status word (akaze_api.hip)
        int per = 4096 / (nframes > 0 ? nframes : 1);  // workgroups per frame
        per = per < 16 ? 16 : (per > 128 ? 128 : per);
        int parts = per / P->nlevels;
        parts = parts < 1 ? 1 : parts;
        hipLaunchKernelGGL(k_akz_fp_build, dim3(P->nlevels, nframes, parts), dim3(AKF_T), 0, st, *P, *S, mask, row_start, cand, cand_resp, cand_count,
                           status);
        for (int p = 0; p < npass; ++p) hipLaunchKernelGGL(k_akz_fp_pass, dim3(per, nframes), dim3(AKF_T), 0, st, *P, *S, cand_count, p);
        hipLaunchKernelGGL(k_akz_fp_entries, dim3(per, nframes), dim3(AKF_T), 0, st, *P, *S, cand, cand_count, npass, status);
        hipLaunchKernelGGL(k_akz_fp_upper, dim3(per, nframes), dim3(AKF_T), 0, st, *P, *S, mask, row_start, cand, cand_count, npass, status);
    } else {
        const size_t lds = (size_t)P->lds_bytes;  // list lengths (u8): a level's own grid + hints of the one below
        (void)hipMemsetAsync(S->ticket, 0, (size_t)(8 + nframes * 16) * sizeof(int), st);
        (void)hipMemsetAsync(S->keep, 1, (size_t)nframes * P->entry_cap, st);
        const int nfx = (nframes + 7) / 8, G = nfx < 8 ? nfx : 8;  // frames per XCD list, group size (see the ticket decoding in the kernel)
        const int blocks = (nfx + G - 1) / G * G * P->nlevels * 8;
        hipLaunchKernelGGL(k_akz_suppress, dim3(blocks), dim3(AKD_T), lds, st, *P, *S, nframes, cand, cand_resp, cand_count, row_start, status);
    }
    const int nchunks = (P->entry_cap + AKD_CHUNK - 1) / AKD_CHUNK;
    hipLaunchKernelGGL(k_akz_refine_a, dim3(nchunks, nframes), dim3(AKD_CHUNK), 0, st, *P, *S, cand_count);
    hipLaunchKernelGGL(k_akz_refine_b, dim3(nchunks, nframes), dim3(AKD_CHUNK), 0, st, *P, *S, kps, kp_count, status);
}
