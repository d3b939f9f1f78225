// k_project.hip — SURVEY §8f rank 1: projection-guided matching core (grid window + Hamming).
//
// Replaces the matching loops of FeatureMatcher::SearchByProjection(F, localMapPoints) (FeatureMatcher.cc:73-154, mode 0)
// and SearchByProjection(CurrentFrame, LastFrame) (:1291-1402, mode 1, mono) together with Frame::GetFeaturesInArea
// (Frame.cc:333-382) over the 64x48 grid of Frame::AssignFeaturesToGrid (Frame.cc:225-240).  The projection (pose x
// point, window radius, admissible size band) is evaluated by the caller as the reference does; each query arrives as
// (u, v, r, min_size, max_size, descriptor).
//
// Like SearchByBoW the loop is greedy: a feature taken by an earlier map point is skipped by later ones.  Same
// two-phase scheme as k_match.hip: phase 1 (one thread per query, all queries in parallel) walks the query's grid
// window in the reference's visiting order (cell column, cell row, ascending feature index) and keeps its 4 best
// (distance, visit rank, feature) keys; phase 2 replays the queries in order in speculative 64-query rounds
// (claim / replay on the feature-occupancy bitset), with an exact window rescan when a query runs out of keys.
#include "afv_device.h"

#define PT 256
#define PK 4
#define P_NO_KEY 0xffffffffffffffffull
#define P_MAX_FEATS 8192

#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

struct DevProjJob {
    const uint32_t *fdesc;
    int n, words;
    const float *x, *y, *size, *angle;
    const uint8_t *occupied;
    const float *inf;
    float min_x, min_y, inv_w, inv_h;
    int cols, rows;
    const int *cell_ptr, *cell_idx;  // grid CSR, cell = ix * rows + iy, ascending feature index inside a cell
    int nq;
    const uint32_t *qdesc;
    const uint8_t *qvalid;
    const float *qu, *qv, *qr, *qmin, *qmax, *qangle;
    const uint8_t *qocc;
    float th, ratio, tol, inv_tol;
    int check_ori, mode;
    unsigned long long *keys;  // [nq][PK]  dist << 32 | visit rank << 16 | feature
    int *ncand;                // [nq] candidates inside the window (geometry only)
    int *orilist;              // [nq][2] (feature, rotation bin) of accepted matches, mode 1
    int *assign;               // [n]
    int *nmatches;
};

__device__ __forceinline__ int proj_rotation_bin(float a1, float a2) {  // FeatureMatcher.cc:1587-1599
    const float rot_factor = 1.0f / 30.0f;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * rot_factor);
    if (bin == 30) bin = 0;
    return bin;
}

template <int W>
__device__ __forceinline__ int proj_hamming(const uint32_t *a, const uint32_t *b) {
    int d = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) d += __popc(a[i] ^ b[i]);
    return d;
}

struct Window {
    int cx0, cx1, cy0, cy1;
    bool ok;
};

// cell range of Frame::GetFeaturesInArea (Frame.cc:339-353)
__device__ __forceinline__ Window proj_window(const DevProjJob &J, float x, float y, float r) {
    Window w;
    w.ok = true;
    w.cx0 = max(0, (int)floorf((x - J.min_x - r) * J.inv_w));
    if (w.cx0 >= J.cols) w.ok = false;
    w.cx1 = min(J.cols - 1, (int)ceilf((x - J.min_x + r) * J.inv_w));
    if (w.cx1 < 0) w.ok = false;
    w.cy0 = max(0, (int)floorf((y - J.min_y - r) * J.inv_h));
    if (w.cy0 >= J.rows) w.ok = false;
    w.cy1 = min(J.rows - 1, (int)ceilf((y - J.min_y + r) * J.inv_h));
    if (w.cy1 < 0) w.ok = false;
    return w;
}

// walk the window of query q in visiting order; VISIT(idx, rank) is called for every feature that passes the
// geometric filters of GetFeaturesInArea (size band, |dx| < r, |dy| < r)
#define PROJ_FOR_WINDOW(J, q, VISIT)                                                                  \
    {                                                                                                 \
        const float x_ = J.qu[q], y_ = J.qv[q], r_ = J.qr[q], mn_ = J.qmin[q], mx_ = J.qmax[q];       \
        const Window w_ = proj_window(J, x_, y_, r_);                                                 \
        int rank_ = 0;                                                                                \
        if (w_.ok)                                                                                    \
            for (int ix_ = w_.cx0; ix_ <= w_.cx1; ++ix_)                                              \
                for (int iy_ = w_.cy0; iy_ <= w_.cy1; ++iy_) {                                        \
                    const int c_ = ix_ * J.rows + iy_;                                                \
                    for (int k_ = J.cell_ptr[c_]; k_ < J.cell_ptr[c_ + 1]; ++k_) {                    \
                        const int idx = J.cell_idx[k_];                                               \
                        const float s_ = J.size[idx];                                                 \
                        if (s_ < mn_ || s_ > mx_) continue;                                           \
                        if (!(fabsf(J.x[idx] - x_) < r_ && fabsf(J.y[idx] - y_) < r_)) continue;      \
                        const int rank = rank_++;                                                     \
                        VISIT                                                                         \
                    }                                                                                 \
                }                                                                                     \
    }

template <int W>
__device__ void proj_job(const DevProjJob &J) {
    __shared__ uint32_t s_occ[P_MAX_FEATS / 32];
    __shared__ int s_claim[P_MAX_FEATS];
    __shared__ int s_hist[32];
    __shared__ int s_nm, s_nori, s_drop[3];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;

    for (int i = tid; i < J.n; i += PT) {
        J.assign[i] = -1;
        s_claim[i] = 0x7fffffff;
    }
    for (int i = tid; i < (J.n + 31) / 32; i += PT) {
        uint32_t w = 0;
        if (J.occupied)
            for (int b = 0; b < 32 && i * 32 + b < J.n; ++b) w |= (uint32_t)(J.occupied[i * 32 + b] != 0) << b;
        s_occ[i] = w;
    }
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) {
        s_nm = 0;
        s_nori = 0;
    }
    // ---- phase 1: one thread per query, top-PK keys by (distance, visit rank) ----
    for (int q = tid; q < J.nq; q += PT) {
        unsigned long long k[PK] = {P_NO_KEY, P_NO_KEY, P_NO_KEY, P_NO_KEY};
        int visited = 0;
        if (!J.qvalid || J.qvalid[q]) {
            uint32_t qd[W];
#pragma unroll
            for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
            PROJ_FOR_WINDOW(J, q, {
                const int d = proj_hamming<W>(qd, J.fdesc + (size_t)idx * W);
                unsigned long long key = ((unsigned long long)d << 32) | ((unsigned long long)(rank & 0xffff) << 16) | (unsigned)idx;
                _Pragma("unroll") for (int s = 0; s < PK; ++s) {
                    if (key < k[s]) {
                        const unsigned long long t = k[s];
                        k[s] = key;
                        key = t;
                    }
                }
                visited = rank + 1;
            })
        }
#pragma unroll
        for (int s = 0; s < PK; ++s) J.keys[(size_t)q * PK + s] = k[s];
        J.ncand[q] = visited;
    }
    __syncthreads();
    __threadfence_block();
    if (wv != 0) goto finish;
    {
        // ---- phase 2: ordered walk in rounds of 64 queries ----
        int nm = 0;
        int pos = 0;
        while (pos < J.nq) {
            const int q = pos + lane;
            const bool act = q < J.nq && (!J.qvalid || J.qvalid[q]);
            unsigned long long k[PK] = {P_NO_KEY, P_NO_KEY, P_NO_KEY, P_NO_KEY};
            int visited = 0;
            if (act) {
#pragma unroll
                for (int s = 0; s < PK; ++s) k[s] = J.keys[(size_t)q * PK + s];
                visited = J.ncand[q];
            }
            int e0 = -1, e1 = -1, d0 = 0, d1 = 0;
            bool open = act, exhausted = act;
#pragma unroll
            for (int s = 0; s < PK; ++s) {
                if (open) {
                    if (k[s] == P_NO_KEY) {
                        open = false;
                        exhausted = false;
                    } else {
                        const int idx = (int)(k[s] & 0xffff);
                        if (!((s_occ[idx >> 5] >> (idx & 31)) & 1u)) {
                            if (e0 < 0) {
                                e0 = idx;
                                d0 = (int)(k[s] >> 32);
                                if (J.mode == 1) {  // best only
                                    open = false;
                                    exhausted = false;
                                }
                            } else {
                                e1 = idx;
                                d1 = (int)(k[s] >> 32);
                                open = false;
                                exhausted = false;
                            }
                        }
                    }
                }
            }
            if (visited <= PK) exhausted = false;  // the key list holds the whole window
            int type = 0;  // 0 no match, 1 accept e0, 2 exact rescan of the window
            if (act) {
                if (e0 >= 0 && !((float)d0 <= J.th)) {
                    e0 = -1;  // the best unoccupied candidate fails TH_HIGH: final
                    e1 = -1;
                } else if (exhausted) {
                    type = 2;
                } else if (e0 >= 0) {
                    type = 1;
                    if (J.mode == 0 && e1 >= 0) {  // FeatureMatcher.cc:142-148
                        const float bs = J.size[e0], bs2 = J.size[e1];
                        if ((bs / bs2 < J.tol) && (bs / bs2 > J.inv_tol) && (bs2 > 0.0f) && ((float)d0 > J.ratio * (float)d1)) type = 0;
                    }
                }
            }
            if (type == 1) atomicMin(&s_claim[e0], lane);
            WAVE_LDS_SYNC();
            bool stopper = type == 2;
            if (act && type != 2) {
                if (e0 >= 0 && s_claim[e0] < lane) stopper = true;
                if (e1 >= 0 && s_claim[e1] < lane) stopper = true;
            }
            const unsigned long long sm = __ballot(stopper);
            const int stop = sm ? (int)__builtin_ctzll(sm) : 64;
            const bool commit = type == 1 && lane < stop;
            if (commit) {
                J.assign[e0] = q;
                if (!J.qocc || J.qocc[q]) atomicOr(&s_occ[e0 >> 5], 1u << (e0 & 31));
                if (J.mode == 1 && J.check_ori) {
                    const int slot = atomicAdd(&s_nori, 1);
                    J.orilist[2 * slot] = e0;
                    J.orilist[2 * slot + 1] = proj_rotation_bin(J.qangle[q], J.angle[e0]);
                }
            }
            nm += __popcll(__ballot(commit));
            if (type == 1) s_claim[e0] = 0x7fffffff;
            WAVE_LDS_SYNC();
            if (stop == 0) {
                // exact rescan of the first query's window against the current occupancy (lane 0; rare)
                const int q0 = pos;
                int acc = 0;
                if (lane == 0) {
                    uint32_t qd[W];
#pragma unroll
                    for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q0 * W + i];
                    float best = 3.402823466e+38f, best2 = 3.402823466e+38f, bsz = -1.0f, bsz2 = -1.0f;
                    int bidx = -1;
                    PROJ_FOR_WINDOW(J, q0, {
                        (void)rank;
                        if ((s_occ[idx >> 5] >> (idx & 31)) & 1u) continue;
                        const float d = (float)proj_hamming<W>(qd, J.fdesc + (size_t)idx * W);
                        if (d < best) {
                            best2 = best; best = d; bidx = idx;
                            bsz2 = bsz; bsz = J.size[idx];
                        } else if (J.mode == 0 && d < best2) {
                            best2 = d; bsz2 = J.size[idx];
                        }
                    })
                    bool ok = best <= J.th;
                    if (ok && J.mode == 0 && (bsz / bsz2 < J.tol) && (bsz / bsz2 > J.inv_tol) && (bsz2 > 0.0f) && (best > J.ratio * best2)) ok = false;
                    if (ok) {
                        J.assign[bidx] = q0;
                        if (!J.qocc || J.qocc[q0]) s_occ[bidx >> 5] |= 1u << (bidx & 31);
                        if (J.mode == 1 && J.check_ori) {
                            const int slot = s_nori++;
                            J.orilist[2 * slot] = bidx;
                            J.orilist[2 * slot + 1] = proj_rotation_bin(J.qangle[q0], J.angle[bidx]);
                        }
                        acc = 1;
                    }
                }
                nm += __shfl(acc, 0, 64);
                WAVE_LDS_SYNC();
                pos += 1;
            } else {
                pos += stop;
            }
        }
        if (lane == 0) s_nm = nm;
    }
finish:
    __syncthreads();
    if (J.mode == 1 && J.check_ori) {
        // filterMatchesWithOrientation (Pt flavour, FeatureMatcher.cc:1601-1613) over the accepted-match list
        __threadfence_block();
        const int nori = s_nori;
        for (int i = tid; i < nori; i += PT) atomicAdd(&s_hist[J.orilist[2 * i + 1]], 1);
        __syncthreads();
        if (tid == 0) {
            int i1 = -1, i2 = -1, i3 = -1, max1 = 0, max2 = 0, max3 = 0;
            for (int i = 0; i < 30; ++i) {
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
                else if (sz > max3) { max3 = sz; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            s_drop[0] = i1; s_drop[1] = i2; s_drop[2] = i3;
        }
        __syncthreads();
        int dropped = 0;
        for (int i = tid; i < nori; i += PT) {
            const int b = J.orilist[2 * i + 1];
            if (b != s_drop[0] && b != s_drop[1] && b != s_drop[2]) {
                J.assign[J.orilist[2 * i]] = -1;
                ++dropped;
            }
        }
        if (dropped) atomicSub(&s_nm, dropped);
        __syncthreads();
    }
    if (tid == 0) *J.nmatches = s_nm;
}

__global__ __launch_bounds__(PT) void k_match_projection(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.words == 8) proj_job<8>(J);
    else proj_job<16>(J);
}

// Fuse: independent map points, one thread each; first minimal distance in visiting order wins (strict <, :905)
template <int W>
__device__ void fuse_job(const DevProjJob &J) {
    __shared__ int s_found;
    if (threadIdx.x == 0) s_found = 0;
    __syncthreads();
    int found = 0;
    for (int q = threadIdx.x; q < J.nq; q += PT) {
        int best_idx = -1, best = 0x7fffffff;
        if (!J.qvalid || J.qvalid[q]) {
            uint32_t qd[W];
#pragma unroll
            for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
            const float u = J.qu[q], v = J.qv[q];
            PROJ_FOR_WINDOW(J, q, {
                (void)rank;
                const float ex = u - J.x[idx];
                const float ey = v - J.y[idx];
                const float e2 = ex * ex + ey * ey;
                if (J.inf && (double)(e2 * J.inf[idx]) > 5.99) continue;  // FeatureMatcher.cc:897-898 (absent in Fuse(Sim3) / SearchBySim3)
                const int d = proj_hamming<W>(qd, J.fdesc + (size_t)idx * W);
                if (d < best) {
                    best = d;
                    best_idx = idx;
                }
            })
            if (!(best_idx >= 0 && (float)best <= J.th)) best_idx = -1;
        }
        J.assign[q] = best_idx;
        found += best_idx >= 0;
    }
    if (found) atomicAdd(&s_found, found);
    __syncthreads();
    if (threadIdx.x == 0) *J.nmatches = s_found;
}

__global__ __launch_bounds__(PT) void k_match_fuse(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.words == 8) fuse_job<8>(J);
    else fuse_job<16>(J);
}

// SearchForInitialization (FeatureMatcher.cc:399-557, active part :480-556): queries = level-0 features of F1 searched in a
// fixed window around vbPrevMatched in F2.  Inherently ordered: a candidate is skipped when an earlier query already
// matched it at a distance <= the current one (:513), and a later query steals the feature (:531-535).  One wave per
// job walks the queries in order; the 64 lanes split the window's cells (cell-major = the reference's visiting order),
// so a query costs ceil(cells / 64) rounds plus two 64-bit wave minima.
#define INIT_NO_KEY 0xffffffffffffffffull
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

template <int W>
__device__ void init_job(const DevProjJob &J) {
    __shared__ int s_m21[P_MAX_FEATS];
    __shared__ unsigned short s_mdist[P_MAX_FEATS];
    __shared__ int s_hist[32];
    const int lane = threadIdx.x;
    for (int i = lane; i < J.n; i += 64) {
        s_m21[i] = -1;
        s_mdist[i] = 0xffff;
    }
    for (int q = lane; q < J.nq; q += 64) J.assign[q] = -1;
    if (lane < 32) s_hist[lane] = 0;
    WAVE_LDS_SYNC();
    int nm = 0, nori = 0;
    for (int q = 0; q < J.nq; ++q) {
        if (J.qvalid && !J.qvalid[q]) continue;  // level1 > 0 (:489-491)
        const float x = J.qu[q], y = J.qv[q], r = J.qr[q], mn = J.qmin[q], mx = J.qmax[q];
        const Window w = proj_window(J, x, y, r);
        if (!w.ok) continue;
        uint32_t qd[W];
#pragma unroll
        for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
        const int ny = w.cy1 - w.cy0 + 1, ncells = (w.cx1 - w.cx0 + 1) * ny;
        unsigned long long k0 = INIT_NO_KEY, k1 = INIT_NO_KEY;
        for (int c = lane; c < ncells; c += 64) {
            const int cell = (w.cx0 + c / ny) * J.rows + (w.cy0 + c % ny);
            const int kb = J.cell_ptr[cell], ke = J.cell_ptr[cell + 1];
            for (int k = kb; k < ke; ++k) {
                const int idx = J.cell_idx[k];
                const float sz = J.size[idx];
                if (sz < mn || sz > mx) continue;
                if (!(fabsf(J.x[idx] - x) < r && fabsf(J.y[idx] - y) < r)) continue;
                const int d = proj_hamming<W>(qd, J.fdesc + (size_t)idx * W);
                if ((int)s_mdist[idx] <= d) continue;  // vMatchedDistance[i2] <= descDist (:513)
                // key: distance, then visiting order (cell rank, position in cell), then the feature itself
                const unsigned long long key = ((unsigned long long)d << 48) | ((unsigned long long)c << 32) |
                                               ((unsigned long long)((k - kb) & 0xffff) << 16) | (unsigned)idx;
                if (key < k0) {
                    k1 = k0;
                    k0 = key;
                } else if (key < k1) {
                    k1 = key;
                }
            }
        }
        const unsigned long long g0 = wave_min_u64(k0);
        if (g0 == INIT_NO_KEY) continue;
        const unsigned long long g1 = wave_min_u64(k0 == g0 ? k1 : k0);
        const float best = (float)(int)(g0 >> 48);
        const float best2 = g1 == INIT_NO_KEY ? 3.402823466e+38f : (float)(int)(g1 >> 48);
        const int bidx = (int)(g0 & 0xffff);
        if (best <= J.th && best < best2 * J.ratio) {  // :529-531
            if (lane == 0) {
                const int prev = s_m21[bidx];
                if (prev >= 0) J.assign[prev] = -1;
                J.assign[q] = bidx;
                s_m21[bidx] = q;
                s_mdist[bidx] = (unsigned short)(int)best;
                if (J.check_ori) {
                    const int bin = proj_rotation_bin(J.qangle[q], J.angle[bidx]);  // F1 keypoint first (:543)
                    J.orilist[2 * nori] = q;
                    J.orilist[2 * nori + 1] = bin;
                    s_hist[bin]++;
                }
            }
            nm += 1;
            nori += 1;
            // a stolen feature takes one match away again (:533-537); every lane tracks the count uniformly
            WAVE_LDS_SYNC();
        }
    }
    // the steal count: matches still standing = queries whose assign survived; recount exactly
    __threadfence_block();
    WAVE_LDS_SYNC();
    if (J.check_ori) {
        // filterMatchesWithOrientation (int flavour, :1615-1629): histogram over every accepted match, stolen ones included
        int i1 = -1, i2 = -1, i3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; ++i) {
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
            else if (sz > max3) { max3 = sz; i3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
        for (int i = lane; i < nori; i += 64) {
            const int b = J.orilist[2 * i + 1];
            if (b != i1 && b != i2 && b != i3) J.assign[J.orilist[2 * i]] = -1;
        }
        __threadfence_block();
        WAVE_LDS_SYNC();
    }
    int cnt = 0;
    for (int q = lane; q < J.nq; q += 64) cnt += J.assign[q] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) *J.nmatches = cnt;
    (void)nm;
}

__global__ __launch_bounds__(64) void k_match_init(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.words == 8) init_job<8>(J);
    else init_job<16>(J);
}

extern "C" void afv_launch_match_init(const DevProjJob *jobs, int njobs, hipStream_t stream) {
    hipLaunchKernelGGL(k_match_init, dim3(njobs), dim3(64), 0, stream, jobs);
}

extern "C" void afv_launch_match_fuse(const DevProjJob *jobs, int njobs, hipStream_t stream) {
    hipLaunchKernelGGL(k_match_fuse, dim3(njobs), dim3(PT), 0, stream, jobs);
}

extern "C" void afv_launch_match_projection(const DevProjJob *jobs, int njobs, hipStream_t stream) {
    hipLaunchKernelGGL(k_match_projection, dim3(njobs), dim3(PT), 0, stream, jobs);
}
