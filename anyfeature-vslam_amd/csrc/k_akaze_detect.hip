// k_akaze_detect.hip — AKAZE Feature_Detection: scale-space extrema, ordered duplicate suppression,
// upper-level filter and sub-pixel refinement.  Restates libAKAZE 1.5 AKAZE::Find_Scale_Space_Extrema and
// Do_Subpixel_Refinement (the calls behind FeatureExtractor_akaze61::detectKeypoints, Feature_akaze61.cpp:38-47); the
// operation order is the one written down in oracle/akaze.c.
//
//  1. candidates: a tiled pass turns every level's Ldet plane into a bitmap of (3x3 strict maximum + thresholds +
//     descriptor-border rule); one workgroup per (level, frame) then counts the rows from the bitmap, scans them and
//     expands the bits, which leaves the candidates in RASTER order without a sort (upstream's loop order is part of
//     the result).
//  2. k_akz_suppress: upstream inserts the candidates one by one into kpts_aux, comparing each against the FIRST earlier
//     entry of the same / previous level within its radius (replace it or drop the newcomer).  One workgroup per frame
//     replays that loop in speculative rounds of AKD_R (128) consecutive candidates: the (candidate, grid cell) pairs of a round
//     are scanned by all threads in two uniform grids (previous level, current level; entries inline in the cell lists,
//     list lengths in LDS); then every candidate checks exactly whether an earlier candidate of the round changes what
//     its search saw (a new / moved entry inside its radius, or a replaced entry that lay inside it) and the round
//     commits, in parallel, up to the first such candidate.
//  3. the upper-level filter, the 2x2 sub-pixel solve and the ordered compaction run at the end of the same kernel.
#include "afv_device.h"
#include "../../include/afv_hip.h"

#define AKD_CELL 10.0f       // grid cell edge in level-0 pixels; must be >= the largest keypoint radius (esigma * derivative_factor)
#define AKD_CELLCAP 48       // list elements per cell and level: <= 25 live entries in a 10 px cell (3x3 strict maxima) + dead ones
#define AKD_MAX_CELLS 12288  // 2 x u16 list lengths in LDS (48 KB): 1280 x 960 at 10 px cells

struct AkdLevel {
    int w, h, octave, sigma_size;
    float psize, ratio;      // esigma * derivative_factor, 2^octave
    const float *ldet;       // [frame][h][w]
    int cand_off;            // offset of this level's candidate slice inside a frame's candidate array
    int cand_cap;
    int row_off;             // offset of this level's rows inside a frame's row-count array
};

struct AkdParams {
    int nlevels, W, H;
    float dthreshold, min_dthreshold;
    AkdLevel lv[16];
    int cand_stride;   // candidates per frame (all levels)
    int rows_stride;   // rows per frame (all levels)
    int gw, gh;        // grid geometry
    int entry_cap, kp_cap;
};

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

// pass A: candidate bitmap of one level.  64 x 32 tile + 1 px ring in LDS (the Ldet plane is streamed once); bit x of word
// mask[frame][row][x / 64] = pixel (x, row) is a candidate.
#define AKD_MAXCHUNKS 32  // 64-column chunks per row: levels up to 2048 pixels wide
__global__ __launch_bounds__(256) void k_akz_cand_mask(AkdParams P, int level, int nframes, unsigned long long *__restrict__ mask) {
    constexpr int LW = 66, LH = 34;
    __shared__ float s_d[LW * LH];
    const AkdLevel L = P.lv[level];
    const int tiles_x = (L.w + 63) / 64, tiles_y = (L.h + 31) / 32, total = tiles_x * tiles_y * nframes;
    const int work = afv_xcd_remap(blockIdx.x, total);
    if (work >= total) return;
    const int f = work / (tiles_x * tiles_y), t = work - f * (tiles_x * tiles_y);
    const int ty0 = (t / tiles_x) * 32, tx0 = (t - (t / tiles_x) * tiles_x) * 64;
    const float *ld = L.ldet + (size_t)f * L.w * L.h;
    for (int i = threadIdx.x; i < LW * LH; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gx = min(max(tx0 - 1 + lx, 0), L.w - 1), gy = min(max(ty0 - 1 + ly, 0), L.h - 1);
        s_d[i] = ld[(size_t)gy * L.w + gx];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float smax = 10.0f * sqrtf(2.0f), r = smax * (float)L.sigma_size;
    const int jx = tx0 + lane;
    // descriptor-border rule ("is_out"), column part: such a point never changes kpts_aux, so it is dropped before the ordered pass
    const float px = (float)jx;
    const bool col_ok = jx >= 1 && jx < L.w - 1 && !(akd_fround(px - r) - 1 < 0 || akd_fround(px + r) + 1 >= L.w);
    for (int ly = wv; ly < 32; ly += 4) {
        const int iy = ty0 + ly;
        if (iy >= L.h) break;
        bool ok = false;
        if (col_ok && iy >= 1 && iy < L.h - 1) {
            const float *c = &s_d[(ly + 1) * LW + lane + 1];
            const float v = c[0];
            ok = v > P.dthreshold && v >= P.min_dthreshold && v > c[-1] && v > c[1] && v > c[-LW - 1] && v > c[-LW] && v > c[-LW + 1] &&
                 v > c[LW - 1] && v > c[LW] && v > c[LW + 1];
            if (ok) {
                const float py = (float)iy;
                ok = !(akd_fround(py - r) - 1 < 0 || akd_fround(py + r) + 1 >= L.h);
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) mask[((size_t)f * P.rows_stride + L.row_off + iy) * AKD_MAXCHUNKS + (tx0 >> 6)] = m;
    }
}

// pass B: one workgroup per (level, frame): row counts from the bitmap, exclusive scan over the rows, then every wavefront
// expands its rows in raster order (index + |response|)
#define AKE_T 1024
__global__ __launch_bounds__(AKE_T) void k_akz_cand_emit(AkdParams P, const unsigned long long *__restrict__ mask, int *__restrict__ row_start,
                                                       int *__restrict__ cand, float *__restrict__ cand_resp, int *__restrict__ cand_count,
                                                       int *__restrict__ status) {
    __shared__ int s_part[AKE_T];
    const int level = blockIdx.x, f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const AkdLevel L = P.lv[level];
    const int nchunks = (L.w + 63) >> 6;
    const unsigned long long *mk = mask + ((size_t)f * P.rows_stride + L.row_off) * AKD_MAXCHUNKS;
    int *rs = row_start + (size_t)f * P.rows_stride + L.row_off;
    const int per = (L.h + AKE_T - 1) / AKE_T, b = min(tid * per, L.h), e = min(L.h, b + per);
    int sum = 0;
    for (int r = b; r < e; ++r) {
        int c = 0;
        for (int k = 0; k < nchunks; ++k) c += __popcll(mk[(size_t)r * AKD_MAXCHUNKS + k]);
        rs[r] = c;  // row count for now
        sum += c;
    }
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int i = 0; i < AKE_T; ++i) {
            const int t = s_part[i];
            s_part[i] = acc;
            acc += t;
        }
        cand_count[f * 16 + level] = min(acc, L.cand_cap);
        if (acc > L.cand_cap) atomicExch(status, 1);
    }
    __syncthreads();
    {
        int acc = s_part[tid];
        for (int r = b; r < e; ++r) {
            const int c = rs[r];
            rs[r] = acc;
            acc += c;
        }
    }
    __threadfence_block();
    __syncthreads();
    const float *ld = L.ldet + (size_t)f * L.w * L.h;
    int *co = cand + (size_t)f * P.cand_stride + L.cand_off;
    float *cr = cand_resp + (size_t)f * P.cand_stride + L.cand_off;
    for (int r = wv; r < L.h; r += AKE_T / 64) {
        // lane k holds chunk k's word; exclusive scan of the popcounts gives every chunk its offset inside the row
        const unsigned long long m = lane < nchunks ? mk[(size_t)r * AKD_MAXCHUNKS + lane] : 0ull;
        const int cnt = __popcll(m);
        const int incl = afv_wave_incl_scan(cnt);
        const int row_total = __shfl(incl, 63, 64);
        if (row_total == 0) continue;
        const int base = rs[r];
        unsigned long long live = __ballot(cnt != 0);
        while (live) {
            const int k = (int)__builtin_ctzll(live);
            live &= live - 1;
            const unsigned long long mkk = __shfl(m, k, 64);
            const int off = base + __shfl(incl - cnt, k, 64);
            if ((mkk >> lane) & 1ull) {
                const int kk = off + __popcll(mkk & ((1ull << lane) - 1ull));
                if (kk < L.cand_cap) {
                    const int idx = r * L.w + (k << 6) + lane;
                    co[kk] = idx;
                    cr[kk] = fabsf(ld[idx]);
                }
            }
        }
    }
}

// ---------------- ordered suppression + upper-level filter + sub-pixel refinement: one workgroup per frame ----------------
// Cell lists hold the entries inline (x, y, response bits, slot), so a neighbourhood scan is one global load per entry; the
// cell counts of both grids live in LDS.  A replaced entry is not unlinked: its old list element is marked dead and a fresh
// element goes into the list of its new cell, so every list only grows and all commits of a round run in parallel.
struct AkdState {
    float *ex, *ey, *eresp;   // [frame][entry_cap]
    int *elevel;              // [frame][entry_cap]
    uint4 *cells;             // [frame][2][ncells][AKD_CELLCAP] {x, y, response, slot}
    int *cell_cnt;            // [frame][ncells] scratch counts of the upper-level filter
    unsigned char *keep;      // [frame][entry_cap]
};

#define AKD_T 1024
#define AKD_PAIRS 18  // 2 grids x 3 x 3 cells per candidate
#define AKD_R 128      // candidates per speculative round (two wavefronts decide / commit)
#define AKD_ITERS ((AKD_R * AKD_PAIRS + AKD_T - 1) / AKD_T)
#define AKD_DEAD 0xffffffffu
#define AKD_NONE 0xffffffffffffffffull
#define AKD_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

__global__ __launch_bounds__(AKD_T) void k_akz_suppress(AkdParams P, AkdState S, const int *__restrict__ cand, const float *__restrict__ cand_resp,
                                                        const int *__restrict__ cand_count,
                                                        afv_keypoint *__restrict__ kps, int *__restrict__ kp_count, int *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char akd_smem[];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int ncells = P.gw * P.gh;
    unsigned short *s_cnt = reinterpret_cast<unsigned short *>(akd_smem);  // [2][ncells] list lengths, both grids
    unsigned int *s_cnt32 = reinterpret_cast<unsigned int *>(akd_smem);    // the same counters as packed pairs (LDS atomics)
    __shared__ unsigned long long s_best[AKD_R];  // per candidate of the round: slot << 32 | response bits (min = first match)
    __shared__ float s_sx[AKD_R], s_sy[AKD_R];
    __shared__ int s_cx[AKD_R], s_cy[AKD_R], s_loc[AKD_R], s_wsum[AKD_T / 64];
    __shared__ float s_mx[AKD_R], s_my[AKD_R];  // position of a candidate's first match
    __shared__ int s_type[AKD_R], s_conf[AKD_R];
    __shared__ int s_first[AKD_R / 64], s_apps[AKD_R / 64], s_napp[AKD_R / 64];  // per deciding wavefront: first conflict, appends, committed appends
    float *ex = S.ex + (size_t)f * P.entry_cap, *ey = S.ey + (size_t)f * P.entry_cap, *er = S.eresp + (size_t)f * P.entry_cap;
    int *el = S.elevel + (size_t)f * P.entry_cap;
    uint4 *cells = S.cells + (size_t)f * 2 * ncells * AKD_CELLCAP;
    int *ccnt = S.cell_cnt + (size_t)f * ncells;
    unsigned char *keep = S.keep + (size_t)f * P.entry_cap;
    for (int i = tid; i < 2 * ncells; i += AKD_T) s_cnt[i] = 0;
    __syncthreads();
    int nE = 0;
    int cur = 0;
    const float inv_cell = 1.0f / AKD_CELL;
#ifdef AFV_AKZ_STATS
    const long long st_t0 = wall_clock64();
    long long st_p[4] = {0, 0, 0, 0};
#endif
    for (int c = 0; c < P.nlevels; ++c) {
        const AkdLevel L = P.lv[c];
        if (c > 0) {  // the previous level's grid becomes "prev"; the other one is recycled
            cur ^= 1;
            for (int i = tid; i < ncells; i += AKD_T) s_cnt[cur * ncells + i] = 0;
            __syncthreads();
        }
        const int prv = cur ^ 1;
        const int *cd = cand + (size_t)f * P.cand_stride + L.cand_off;
        const float *cr = cand_resp + (size_t)f * P.cand_stride + L.cand_off;
        const int n = cand_count[f * 16 + c];
        const float size2 = L.psize * L.psize;
        int pos = 0;
        int pf_pos = -1, pf_idx = 0;
        float pf_resp = 0.f;
#ifdef AFV_AKZ_STATS
        int st_rounds = 0;
#endif
        while (pos < n) {
#ifdef AFV_AKZ_STATS
            ++st_rounds;
#endif
            const int nround = min(AKD_R, n - pos);
#ifdef AFV_AKZ_STATS
            const long long ph0 = wall_clock64();
#endif
            // ---- 1. the round's candidates (first AKD_R threads); the next round's are prefetched while this one is scanned ----
            float sx = 0, sy = 0, resp = 0;
            int cx = 0, cy = 0;
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
                    cx = min((int)(sx * inv_cell), P.gw - 1);
                    cy = min((int)(sy * inv_cell), P.gh - 1);
                }
                s_sx[tid] = sx; s_sy[tid] = sy; s_cx[tid] = cx; s_cy[tid] = cy;
                s_best[tid] = AKD_NONE;
                pf_pos = pos + AKD_R;  // valid if this round commits all of its candidates (the common case)
                if (pf_pos + tid < n) {
                    pf_idx = cd[pf_pos + tid];
                    pf_resp = cr[pf_pos + tid];
                }
            }
            __syncthreads();
#ifdef AFV_AKZ_STATS
            const long long ph1 = wall_clock64();
#endif
            // ---- 2. neighbourhood scan: the (candidate, cell) pairs of the round spread over the whole workgroup ----
            unsigned long long lbest[AKD_ITERS];
            int lloc[AKD_ITERS];
            float lmx[AKD_ITERS], lmy[AKD_ITERS];
#pragma unroll
            for (int it = 0; it < AKD_ITERS; ++it) {
                lbest[it] = AKD_NONE;
                lloc[it] = 0;
                lmx[it] = 0.f;
                lmy[it] = 0.f;
                const int t = tid + it * AKD_T;
                if (t >= nround * AKD_PAIRS) continue;
                const int q = t / AKD_PAIRS, k = t - q * AKD_PAIRS;
                const int g = k / 9, kk = k - g * 9;
                if (g == 0 && c == 0) continue;
                const int xx = s_cx[q] + (kk % 3) - 1, yy = s_cy[q] + (kk / 3) - 1;
                if (xx < 0 || yy < 0 || xx >= P.gw || yy >= P.gh) continue;
                const int gi = g == 0 ? prv : cur;
                const int cell = yy * P.gw + xx;
                const int cn = min((int)s_cnt[gi * ncells + cell], AKD_CELLCAP);
                if (cn == 0) continue;
                const size_t lb = ((size_t)gi * ncells + cell) * AKD_CELLCAP;
                const float qx = s_sx[q], qy = s_sy[q];
                unsigned long long best = AKD_NONE;
                int beste = 0;
                float bx = 0.f, by = 0.f;
                for (int e = 0; e < cn; ++e) {
                    const uint4 v = cells[lb + e];
                    if (v.w == AKD_DEAD) continue;  // replaced earlier: the entry lives on in another list
                    const float dx = qx - __uint_as_float(v.x), dy = qy - __uint_as_float(v.y);
                    if (dx * dx + dy * dy <= size2) {
                        const unsigned long long key = ((unsigned long long)v.w << 32) | v.z;
                        if (key < best) {
                            best = key;
                            beste = e;
                            bx = __uint_as_float(v.x);
                            by = __uint_as_float(v.y);
                        }
                    }
                }
                if (best != AKD_NONE) {
                    atomicMin(&s_best[q], best);
                    lbest[it] = best;
                    lloc[it] = (int)lb + beste;
                    lmx[it] = bx;
                    lmy[it] = by;
                }
            }
            __syncthreads();
#ifdef AFV_AKZ_STATS
            const long long ph2 = wall_clock64();
#endif
            // the thread that found a candidate's first match publishes where that list element sits (slots are unique)
#pragma unroll
            for (int it = 0; it < AKD_ITERS; ++it) {
                if (lbest[it] != AKD_NONE) {
                    const int q = (tid + it * AKD_T) / AKD_PAIRS;
                    if (s_best[q] == lbest[it]) {
                        s_loc[q] = lloc[it];
                        s_mx[q] = lmx[it];
                        s_my[q] = lmy[it];
                    }
                }
            }
            __syncthreads();
#ifdef AFV_AKZ_STATS
            const long long ph3 = wall_clock64();
#endif
            // ---- 3. decisions, exact conflict test against the earlier lanes of the round, commit (wave 0) ----
            // 3a. decisions (first AKD_R threads)
            int first = -1, type = 0;  // type: 0 drop, 1 append, 2 replace `first`
            float oex = 0, oey = 0;
            if (tid < AKD_R) {
                const unsigned long long b = s_best[tid];
                first = (act && b != AKD_NONE) ? (int)(b >> 32) : -1;
                if (act) {
                    if (first < 0) type = 1;
                    else if (resp > __uint_as_float((unsigned int)(b & 0xffffffffu))) {
                        type = 2;
                        oex = s_mx[tid];
                        oey = s_my[tid];
                    }
                }
                s_type[tid] = type;
                s_mx[tid] = oex;
                s_my[tid] = oey;
                s_conf[tid] = 0;
                const unsigned long long apm = __ballot(type == 1);
                if (lane == 0) s_apps[tid >> 6] = __popcll(apm);
            }
            __syncthreads();
            // 3b. a candidate's decision stands unless an earlier candidate of the round changes what its search sees: a new /
            //     moved entry inside its radius, or a replaced entry that used to lie inside its radius.  The ordered pairs (j < i)
            //     are spread over the workgroup (AKD_T / AKD_R threads per candidate i, strided over the earlier candidates j).
            {
                constexpr int PER = AKD_T / AKD_R;  // threads per candidate
                const int i = tid / PER;
                if (i < nround) {
                    const float xi = s_sx[i], yi = s_sy[i];
                    bool hit = false;
                    for (int j = tid % PER; j < i; j += PER) {
                        const int tj = s_type[j];
                        if (tj == 0) continue;
                        const float dx = xi - s_sx[j], dy = yi - s_sy[j];
                        hit = hit || (dx * dx + dy * dy <= size2);
                        if (tj == 2) {
                            const float ux = xi - s_mx[j], uy = yi - s_my[j];
                            hit = hit || (ux * ux + uy * uy <= size2);
                        }
                    }
                    if (hit) s_conf[i] = 1;
                }
            }
            __syncthreads();
            // 3c. commit up to the first conflicting candidate (first AKD_R threads = AKD_R / 64 wavefronts)
            if (tid < AKD_R) {
                const bool conflict = act && s_conf[tid] != 0;
                const unsigned long long cm = __ballot(conflict);
                if (lane == 0) s_first[tid >> 6] = cm ? (tid & ~63) + (int)__builtin_ctzll(cm) : AKD_R;
            }
            __syncthreads();
            int stop = nround;
#pragma unroll
            for (int w = 0; w < AKD_R / 64; ++w) stop = min(stop, s_first[w]);
            if (tid < AKD_R) {
                const bool commit = act && tid < stop && type != 0;
                const unsigned long long am = __ballot(commit && type == 1);
                if (commit) {
                    int slot;
                    if (type == 1) {
                        // appends of the earlier deciding wavefronts all commit when this wavefront commits anything
                        int before = 0;
                        for (int w = 0; w < (tid >> 6); ++w) before += s_apps[w];
                        slot = nE + before + __popcll(am & ((1ull << lane) - 1ull));
                    } else {
                        slot = first;
                        cells[s_loc[tid]].w = AKD_DEAD;
                    }
                    if (slot < P.entry_cap) {
                        ex[slot] = sx;
                        ey[slot] = sy;
                        er[slot] = resp;
                        el[slot] = c;
                        // list slot from an LDS atomic on the packed 16-bit counters (several commits may share a cell)
                        const int ci = cur * ncells + cy * P.gw + cx;
                        const unsigned int old = atomicAdd(&s_cnt32[ci >> 1], (ci & 1) ? 0x10000u : 1u);
                        const int cn = (int)((old >> ((ci & 1) * 16)) & 0xffffu);
                        if (cn < AKD_CELLCAP) {
                            cells[(size_t)ci * AKD_CELLCAP + cn] = make_uint4(__float_as_uint(sx), __float_as_uint(sy), __float_as_uint(resp), (unsigned)slot);
                        } else {
                            atomicExch(status, 2);
                        }
                    } else {
                        atomicExch(status, 3);
                    }
                }
                if (lane == 0) s_napp[tid >> 6] = __popcll(am);
            }
            __threadfence_block();
            __syncthreads();
            pos += stop;
#pragma unroll
            for (int w = 0; w < AKD_R / 64; ++w) nE += s_napp[w];
            nE = min(nE, P.entry_cap);
#ifdef AFV_AKZ_STATS
            { const long long ph4 = wall_clock64(); st_p[0] += ph1 - ph0; st_p[1] += ph2 - ph1; st_p[2] += ph3 - ph2; st_p[3] += ph4 - ph3; }
#endif
        }
#ifdef AFV_AKZ_STATS
        if (tid == 0 && f == 0) printf("akz_suppress: level %d candidates %d rounds %d entries %d\n", c, n, st_rounds, nE);
#endif
    }
#ifdef AFV_AKZ_STATS
    const long long st_t1 = wall_clock64();
#endif
    // ---- "Now filter points with the upper scale level": entry i of level c is repeated if a LATER entry of level c+1 lies
    //      within size_i and has a larger response.  Per level pair: grid of the level c+1 entries, then one thread per entry.
    for (int i = tid; i < nE; i += AKD_T) keep[i] = 1;
    for (int c = 0; c + 1 < P.nlevels; ++c) {
        for (int i = tid; i < ncells; i += AKD_T) ccnt[i] = 0;
        __threadfence_block();
        __syncthreads();
        for (int i = tid; i < nE; i += AKD_T)
            if (el[i] == c + 1) {
                const int cell = min((int)(ey[i] * inv_cell), P.gh - 1) * P.gw + min((int)(ex[i] * inv_cell), P.gw - 1);
                const int k = atomicAdd(&ccnt[cell], 1);
                if (k < AKD_CELLCAP) cells[(size_t)cell * AKD_CELLCAP + k] = make_uint4(__float_as_uint(ex[i]), __float_as_uint(ey[i]), __float_as_uint(er[i]), (unsigned)i);
                else atomicExch(status, 2);
            }
        __threadfence_block();
        __syncthreads();
        const float sz = P.lv[c].psize, sz2 = sz * sz;
        for (int i = tid; i < nE; i += AKD_T)
            if (el[i] == c) {
                const float x = ex[i], y = ey[i], r = er[i];
                const int cx = min((int)(x * inv_cell), P.gw - 1), cy = min((int)(y * inv_cell), P.gh - 1);
                bool rep = false;
                for (int yy = max(cy - 1, 0); yy <= min(cy + 1, P.gh - 1) && !rep; ++yy)
                    for (int xx = max(cx - 1, 0); xx <= min(cx + 1, P.gw - 1) && !rep; ++xx) {
                        const int cell = yy * P.gw + xx;
                        const int cn = min(ccnt[cell], AKD_CELLCAP);
                        for (int e = 0; e < cn; ++e) {
                            const uint4 v = cells[(size_t)cell * AKD_CELLCAP + e];
                            if ((int)v.w <= i) continue;
                            const float dx = x - __uint_as_float(v.x), dy = y - __uint_as_float(v.y);
                            if (dx * dx + dy * dy <= sz2 && r < __uint_as_float(v.z)) {
                                rep = true;
                                break;
                            }
                        }
                    }
                if (rep) keep[i] = 0;
            }
        __threadfence_block();
        __syncthreads();
    }
#ifdef AFV_AKZ_STATS
    const long long st_t2 = wall_clock64();
#endif
    // ---- Do_Subpixel_Refinement + ordered compaction ----
    int nout = 0;
    for (int i0 = 0; i0 < nE; i0 += AKD_T) {
        const int i = i0 + tid;
        bool ok = i < nE && keep[i] != 0;
        float kx = 0, ky = 0, ksize = 0, kresp = 0;
        int koct = 0, klev = 0;
        if (ok) {
            klev = el[i];
            const AkdLevel L = P.lv[klev];
            koct = L.octave;
            kresp = er[i];
            const int x = akd_fround(ex[i] / L.ratio), y = akd_fround(ey[i] / L.ratio), w = L.w;
            const float *D = L.ldet + (size_t)f * L.w * L.h;
#define LD(yy, xx) D[(size_t)(yy) * w + (xx)]
            const float Dx = (float)(0.5 * (double)(LD(y, x + 1) - LD(y, x - 1)));
            const float Dy = (float)(0.5 * (double)(LD(y + 1, x) - LD(y - 1, x)));
            const float Dxx = (float)((double)(LD(y, x + 1) + LD(y, x - 1)) - 2.0 * (double)LD(y, x));
            const float Dyy = (float)((double)(LD(y + 1, x) + LD(y - 1, x)) - 2.0 * (double)LD(y, x));
            const float Dxy =
                (float)(0.25 * (double)(LD(y + 1, x + 1) + LD(y - 1, x - 1)) - 0.25 * (double)(LD(y - 1, x + 1) + LD(y + 1, x - 1)));
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
                    kx = ((float)x + d0) * power;
                    ky = ((float)y + d1) * power;
                    ksize = L.psize * 2.0f;
                } else {
                    ok = false;
                }
            }
        }
        const unsigned long long m = __ballot(ok);
        if (lane == 0) s_wsum[tid >> 6] = __popcll(m);
        __syncthreads();
        int base = nout, tot = 0;
        for (int w = 0; w < AKD_T / 64; ++w) {
            if (w < (tid >> 6)) base += s_wsum[w];
            tot += s_wsum[w];
        }
        if (ok) {
            const int o = base + __popcll(m & ((1ull << lane) - 1ull));
            if (o < P.kp_cap) {
                afv_keypoint k;
                k.x = kx; k.y = ky; k.size = ksize; k.angle = 0.0f; k.response = kresp; k.octave = koct; k.class_id = klev;
                kps[(size_t)f * P.kp_cap + o] = k;
            } else {
                atomicExch(status, 4);
            }
        }
        nout += tot;
        __syncthreads();
    }
#ifdef AFV_AKZ_STATS
    if (tid == 0 && f == 0) printf("akz_suppress phases (us at 100 MHz): rounds %lld upper %lld subpixel %lld | per-round parts: load %lld scan %lld publish %lld decide+commit %lld\n", (st_t1 - st_t0) / 100, (st_t2 - st_t1) / 100, (wall_clock64() - st_t2) / 100, st_p[0] / 100, st_p[1] / 100, st_p[2] / 100, st_p[3] / 100);
#endif
    if (tid == 0) kp_count[f] = min(nout, P.kp_cap);
}

extern "C" void afv_akz_launch_candidates(const AkdParams *P, int nframes, unsigned long long *mask, int *row_start, int *cand, float *cand_resp,
                                          int *cand_count, int *status, hipStream_t st) {
    for (int l = 0; l < P->nlevels; ++l) {
        const int total = ((P->lv[l].w + 63) / 64) * ((P->lv[l].h + 31) / 32) * nframes;
        hipLaunchKernelGGL(k_akz_cand_mask, dim3((total + 7) / 8 * 8), dim3(256), 0, st, *P, l, nframes, mask);
    }
    hipLaunchKernelGGL(k_akz_cand_emit, dim3(P->nlevels, nframes), dim3(AKE_T), 0, st, *P, mask, row_start, cand, cand_resp, cand_count, status);
}

extern "C" void afv_akz_launch_suppress(const AkdParams *P, const AkdState *S, int nframes, const int *cand, const float *cand_resp,
                                        const int *cand_count,
                                        afv_keypoint *kps, int *kp_count, int *status, hipStream_t st) {
    const size_t lds = ((size_t)P->gw * P->gh * 2 * sizeof(unsigned short) + 15) & ~(size_t)15;  // two count grids (u16)
    hipLaunchKernelGGL(k_akz_suppress, dim3(nframes), dim3(AKD_T), lds, st, *P, *S, cand, cand_resp, cand_count, kps, kp_count, status);
}
