// k_akaze_detect.hip — AKAZE Feature_Detection: scale-space extrema, ordered duplicate suppression,
// upper-level filter and sub-pixel refinement.  Restates libAKAZE 1.5 AKAZE::Find_Scale_Space_Extrema and
// Do_Subpixel_Refinement (the calls behind FeatureExtractor_akaze61::detectKeypoints, Feature_akaze61.cpp:38-47); the
// operation order is the one written down in oracle/akaze.c.
//
//  1. candidates: one wavefront per image row tests the 3x3 strict maximum + thresholds + descriptor-border rule; a row
//     count pass, a per-(frame, level) scan over rows and a write pass leave every level's candidates in RASTER order
//     without a sort (upstream's loop order is part of the result).
//  2. k_akz_suppress: upstream inserts the candidates one by one into kpts_aux, comparing each against the FIRST earlier
//     entry of the same / previous level within its radius (replace it or drop the newcomer).  One wavefront per frame
//     replays that loop in speculative rounds of 64 consecutive candidates: every lane looks its first match up in two
//     uniform grids (previous level, current level), lanes that change the list mark the grid cells they touch, and the
//     round commits up to the first lane that sees a mark of an earlier lane in its 3x3 cell neighbourhood.  A candidate
//     only interacts inside its radius (<= one cell), so everything committed in a round is independent.
//  3. the upper-level filter, the 2x2 sub-pixel solve and the ordered compaction run at the end of the same kernel.
#include "afv_device.h"
#include "../../include/afv_hip.h"

#define AKD_CELL 10.0f       // grid cell edge in level-0 pixels; must be >= the largest keypoint radius (esigma * derivative_factor)
#define AKD_CELLCAP 32       // entries of one level per cell (strict 3x3 maxima are >= 2 px apart: <= 25 in a 10 px cell)
#define AKD_MAX_CELLS 12288  // LDS mark table (48 KB): 1280 x 960 at 10 px cells

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

// pass A (write == 0): candidates per row; pass C (write == 1): write them at row offset + rank.  One wavefront per row.
template <int WRITE>
__global__ __launch_bounds__(256) void k_akz_cand_rows(AkdParams P, int level, int *__restrict__ row_count, const int *__restrict__ row_start,
                                                       int *__restrict__ cand, int *__restrict__ status) {
    const AkdLevel L = P.lv[level];
    const int f = blockIdx.y, lane = threadIdx.x & 63;
    const int iy = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (iy >= L.h) return;
    const float *ld = L.ldet + (size_t)f * L.w * L.h;
    int total = 0;
    int base = 0;
    if (WRITE) base = row_start[(size_t)f * P.rows_stride + L.row_off + iy];
    if (iy >= 1 && iy < L.h - 1) {
        for (int x0 = 0; x0 < L.w; x0 += 64) {
            const int jx = x0 + lane;
            const bool ok = jx >= 1 && jx < L.w - 1 && akd_is_candidate(P, L, ld, jx, iy);
            const unsigned long long m = __ballot(ok);
            if (WRITE && ok) {
                const int k = base + total + __popcll(m & ((1ull << lane) - 1ull));
                if (k < L.cand_cap) cand[(size_t)f * P.cand_stride + L.cand_off + k] = iy * L.w + jx;
                else atomicExch(status, 1);
            }
            total += __popcll(m);
        }
    }
    if (!WRITE && lane == 0) row_count[(size_t)f * P.rows_stride + L.row_off + iy] = total;
}

// pass B: exclusive scan of the row counts of one (frame, level); also the level's candidate count
__global__ __launch_bounds__(256) void k_akz_cand_scan(AkdParams P, const int *__restrict__ row_count, int *__restrict__ row_start,
                                                       int *__restrict__ cand_count) {
    __shared__ int s_part[256];
    const int level = blockIdx.x, f = blockIdx.y;
    const AkdLevel L = P.lv[level];
    const int *rc = row_count + (size_t)f * P.rows_stride + L.row_off;
    int *rs = row_start + (size_t)f * P.rows_stride + L.row_off;
    const int per = (L.h + 255) / 256, b = threadIdx.x * per, e = min(L.h, b + per);
    int s = 0;
    for (int i = b; i < e; ++i) s += rc[i];
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < 256; ++i) {
            const int t = s_part[i];
            s_part[i] = acc;
            acc += t;
        }
        cand_count[f * 16 + level] = min(acc, L.cand_cap);
    }
    __syncthreads();
    int acc = s_part[threadIdx.x];
    for (int i = b; i < e; ++i) {
        rs[i] = acc;
        acc += rc[i];
    }
}

// ---------------- ordered suppression + upper-level filter + sub-pixel refinement: one wavefront per frame ----------------
struct AkdState {
    float *ex, *ey, *eresp;   // [frame][entry_cap]
    int *elevel;              // [frame][entry_cap]
    unsigned short *cells;    // [frame][2][ncells][AKD_CELLCAP] slots
    int *cell_cnt;            // [frame][2][ncells]
    unsigned char *keep;      // [frame][entry_cap]
};

#define AKD_WAVE_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)
// global-memory visibility inside the one wavefront that owns a frame's state
#define AKD_MEM_SYNC()                 \
    do {                               \
        __threadfence();               \
        __builtin_amdgcn_wave_barrier(); \
    } while (0)

__global__ __launch_bounds__(64) void k_akz_suppress(AkdParams P, AkdState S, const int *__restrict__ cand, const int *__restrict__ cand_count,
                                                     afv_keypoint *__restrict__ kps, int *__restrict__ kp_count, int *__restrict__ status) {
    __shared__ unsigned int s_mark[AKD_MAX_CELLS];
    const int f = blockIdx.x, lane = threadIdx.x;
    const int ncells = P.gw * P.gh;
    float *ex = S.ex + (size_t)f * P.entry_cap, *ey = S.ey + (size_t)f * P.entry_cap, *er = S.eresp + (size_t)f * P.entry_cap;
    int *el = S.elevel + (size_t)f * P.entry_cap;
    unsigned short *cells = S.cells + (size_t)f * 2 * ncells * AKD_CELLCAP;
    int *ccnt = S.cell_cnt + (size_t)f * 2 * ncells;
    unsigned char *keep = S.keep + (size_t)f * P.entry_cap;
    for (int i = lane; i < ncells; i += 64) s_mark[i] = 0xffu;
    for (int i = lane; i < 2 * ncells; i += 64) ccnt[i] = 0;
    AKD_MEM_SYNC();
    AKD_WAVE_SYNC();
    int nE = 0;
    int cur = 0;  // grid index of the current level
    const float inv_cell = 1.0f / AKD_CELL;
    for (int c = 0; c < P.nlevels; ++c) {
        const AkdLevel L = P.lv[c];
        if (c > 0) {  // the previous level's grid becomes "prev"; the other one is recycled
            cur ^= 1;
            for (int i = lane; i < ncells; i += 64) ccnt[cur * ncells + i] = 0;
            AKD_MEM_SYNC();
        }
        const int prv = cur ^ 1;
        const float *ld = L.ldet + (size_t)f * L.w * L.h;
        const int *cd = cand + (size_t)f * P.cand_stride + L.cand_off;
        const int n = cand_count[f * 16 + c];
        const float size2 = L.psize * L.psize;
        int pos = 0;
        while (pos < n) {
            const int q = pos + lane;
            const bool act = q < n;
            float sx = 0, sy = 0, resp = 0;
            int first = -1, cx = 0, cy = 0;
            if (act) {
                const int idx = cd[q];
                const int iy = idx / L.w, jx = idx - iy * L.w;
                resp = fabsf(ld[idx]);
                sx = (float)jx * L.ratio;
                sy = (float)iy * L.ratio;
                cx = min((int)(sx * inv_cell), P.gw - 1);
                cy = min((int)(sy * inv_cell), P.gh - 1);
                // first entry (smallest slot) of level c-1 / c within the radius
                unsigned int best = 0xffffffffu;
                for (int g = 0; g < 2; ++g) {
                    if (g == 0 && c == 0) continue;
                    const int gi = g == 0 ? prv : cur;
                    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, P.gh - 1); ++yy)
                        for (int xx = max(cx - 1, 0); xx <= min(cx + 1, P.gw - 1); ++xx) {
                            const int cell = yy * P.gw + xx;
                            const int cn = ccnt[gi * ncells + cell];
                            const unsigned short *sl = cells + ((size_t)gi * ncells + cell) * AKD_CELLCAP;
                            for (int e = 0; e < cn; ++e) {
                                const unsigned int slot = sl[e];
                                const float dx = sx - ex[slot], dy = sy - ey[slot];
                                const float dist = dx * dx + dy * dy;
                                if (dist <= size2) best = min(best, slot);
                            }
                        }
                }
                first = best == 0xffffffffu ? -1 : (int)best;
            }
            // 0 drop, 1 append, 2 replace `first`
            int type = 0;
            int ocell = -1;
            if (act) {
                if (first < 0) type = 1;
                else if (resp > er[first]) {
                    type = 2;
                    ocell = min((int)(ey[first] * inv_cell), P.gh - 1) * P.gw + min((int)(ex[first] * inv_cell), P.gw - 1);
                }
            }
            const int mycell = cy * P.gw + cx;
            if (type != 0) {  // lanes that change the list mark the cells they touch with their lane number
                atomicMin(&s_mark[mycell], (unsigned)lane);
                if (type == 2) atomicMin(&s_mark[ocell], (unsigned)lane);
            }
            AKD_WAVE_SYNC();
            bool conflict = false;
            if (act) {
                for (int yy = max(cy - 1, 0); yy <= min(cy + 1, P.gh - 1); ++yy)
                    for (int xx = max(cx - 1, 0); xx <= min(cx + 1, P.gw - 1); ++xx)
                        if (s_mark[yy * P.gw + xx] < (unsigned)lane) conflict = true;
            }
            const unsigned long long cm = __ballot(conflict);
            const int stop = cm ? (int)__builtin_ctzll(cm) : 64;
            const bool commit = act && lane < stop;
            const unsigned long long am = __ballot(commit && type == 1);
            AKD_WAVE_SYNC();
            // clear the marks (every marking lane resets its own cells)
            if (type != 0) {
                s_mark[mycell] = 0xffu;
                if (type == 2) s_mark[ocell] = 0xffu;
            }
            if (commit && type != 0) {
                int slot;
                if (type == 1) {
                    slot = nE + __popcll(am & ((1ull << lane) - 1ull));
                } else {
                    slot = first;
                    // take the slot out of its old cell list (grid of the old entry's level)
                    const int gi = el[first] == c ? cur : prv;
                    unsigned short *sl = cells + ((size_t)gi * ncells + ocell) * AKD_CELLCAP;
                    const int cn = ccnt[gi * ncells + ocell];
                    for (int e = 0; e < cn; ++e)
                        if (sl[e] == (unsigned short)first) {
                            sl[e] = sl[cn - 1];
                            break;
                        }
                    ccnt[gi * ncells + ocell] = cn - 1;
                }
                if (slot < P.entry_cap) {
                    ex[slot] = sx;
                    ey[slot] = sy;
                    er[slot] = resp;
                    el[slot] = c;
                    const int cn = ccnt[cur * ncells + mycell];
                    if (cn < AKD_CELLCAP) {
                        cells[((size_t)cur * ncells + mycell) * AKD_CELLCAP + cn] = (unsigned short)slot;
                        ccnt[cur * ncells + mycell] = cn + 1;
                    } else {
                        atomicExch(status, 2);
                    }
                } else {
                    atomicExch(status, 3);
                }
            }
            nE = min(nE + __popcll(am), P.entry_cap);
            AKD_MEM_SYNC();
            AKD_WAVE_SYNC();
            pos += stop;
        }
    }
    // ---- "Now filter points with the upper scale level": entry i of level c is repeated if a LATER entry of level c+1 lies
    //      within size_i and has a larger response.  Per level pair: grid of the level c+1 entries, then one lane per entry.
    for (int i = lane; i < nE; i += 64) keep[i] = 1;
    for (int c = 0; c + 1 < P.nlevels; ++c) {
        for (int i = lane; i < ncells; i += 64) ccnt[i] = 0;
        AKD_MEM_SYNC();
        for (int i = lane; i < nE; i += 64)
            if (el[i] == c + 1) {
                const int cell = min((int)(ey[i] * inv_cell), P.gh - 1) * P.gw + min((int)(ex[i] * inv_cell), P.gw - 1);
                const int k = atomicAdd(&ccnt[cell], 1);
                if (k < AKD_CELLCAP) cells[(size_t)cell * AKD_CELLCAP + k] = (unsigned short)i;
                else atomicExch(status, 2);
            }
        AKD_MEM_SYNC();
        const float sz = P.lv[c].psize, sz2 = sz * sz;
        for (int i = lane; i < nE; i += 64)
            if (el[i] == c) {
                const float x = ex[i], y = ey[i], r = er[i];
                const int cx = min((int)(x * inv_cell), P.gw - 1), cy = min((int)(y * inv_cell), P.gh - 1);
                bool rep = false;
                for (int yy = max(cy - 1, 0); yy <= min(cy + 1, P.gh - 1) && !rep; ++yy)
                    for (int xx = max(cx - 1, 0); xx <= min(cx + 1, P.gw - 1) && !rep; ++xx) {
                        const int cell = yy * P.gw + xx;
                        const int cn = min(ccnt[cell], AKD_CELLCAP);
                        for (int e = 0; e < cn; ++e) {
                            const int j = cells[(size_t)cell * AKD_CELLCAP + e];
                            if (j <= i) continue;
                            const float dx = x - ex[j], dy = y - ey[j];
                            if (dx * dx + dy * dy <= sz2 && r < er[j]) {
                                rep = true;
                                break;
                            }
                        }
                    }
                if (rep) keep[i] = 0;
            }
        AKD_MEM_SYNC();
    }
    // ---- Do_Subpixel_Refinement + ordered compaction ----
    int nout = 0;
    for (int i0 = 0; i0 < nE; i0 += 64) {
        const int i = i0 + lane;
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
        if (ok) {
            const int o = nout + __popcll(m & ((1ull << lane) - 1ull));
            if (o < P.kp_cap) {
                afv_keypoint k;
                k.x = kx; k.y = ky; k.size = ksize; k.angle = 0.0f; k.response = kresp; k.octave = koct; k.class_id = klev;
                kps[(size_t)f * P.kp_cap + o] = k;
            } else {
                atomicExch(status, 4);
            }
        }
        nout += __popcll(m);
    }
    if (lane == 0) kp_count[f] = min(nout, P.kp_cap);
}

extern "C" void afv_akz_launch_candidates(const AkdParams *P, int nframes, int *row_count, int *row_start, int *cand, int *cand_count,
                                          int *status, hipStream_t st) {
    for (int l = 0; l < P->nlevels; ++l)
        hipLaunchKernelGGL(k_akz_cand_rows<0>, dim3((P->lv[l].h + 3) / 4, nframes), dim3(256), 0, st, *P, l, row_count, row_start, cand, status);
    hipLaunchKernelGGL(k_akz_cand_scan, dim3(P->nlevels, nframes), dim3(256), 0, st, *P, row_count, row_start, cand_count);
    for (int l = 0; l < P->nlevels; ++l)
        hipLaunchKernelGGL(k_akz_cand_rows<1>, dim3((P->lv[l].h + 3) / 4, nframes), dim3(256), 0, st, *P, l, row_count, row_start, cand, status);
}

extern "C" void afv_akz_launch_suppress(const AkdParams *P, const AkdState *S, int nframes, const int *cand, const int *cand_count,
                                        afv_keypoint *kps, int *kp_count, int *status, hipStream_t st) {
    hipLaunchKernelGGL(k_akz_suppress, dim3(nframes), dim3(64), 0, st, *P, *S, cand, cand_count, kps, kp_count, status);
}
