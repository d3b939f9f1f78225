// k_project.hip — SURVEY §8f rank 1: projection-guided matching cores (grid window + Hamming).
//
// Replaces the matching loops of
//   SearchByProjection(F, localMapPoints)            FeatureMatcher.cc:73-154     (k_proj_topk + k_proj_resolve, mode 0)
//   SearchByProjection(CurrentFrame, LastFrame)      :1291-1402 (mono)            (mode 1; also the relocalisation search
//                                                                                  :1404-1506 and the Sim3 search :287-397)
//   Fuse(pKF, vpMapPoints) / Fuse(pKF, Scw, ...)     :794-940 / :942-1064         (k_match_fuse)
//   SearchBySim3                                     :1066-1287                   (two k_match_fuse jobs + host agreement)
//   SearchForInitialization                          :399-557                     (k_proj_topk<8> + k_init_resolve)
// together with Frame/KeyFrame::GetFeaturesInArea (Frame.cc:333-382, KeyFrame.cc:613-652) over the 64x48 grid of
// Frame::AssignFeaturesToGrid (Frame.cc:225-240).  The projection (pose x point, window radius, admissible size band) is
// evaluated by the caller as the reference does; each query arrives as (u, v, r, min_size, max_size, descriptor).
//
// The greedy searches are order dependent (a feature taken by an earlier map point is skipped by later ones), so they
// run in two kernels like k_match.hip:
//   1. k_proj_topk — one WAVE per query, all queries of all jobs in parallel: the 64 lanes split the window's cells
//      (cell-major = the reference's visiting order), each lane keeps its K best keys
//      (distance << 48 | window cell rank << 32 | position in cell << 16 | feature), and K wave-minimum rounds extract
//      the query's K best.  Keys order candidates exactly like the reference's sequential best/second scan.
//   2. k_proj_resolve — one wave per job replays the queries in order in speculative 64-query rounds (claim / replay on
//      the feature-occupancy bitset); a query that runs out of keys is rescanned exactly, again with lanes over cells.
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch
#include "afv_jobs.h"

#define PT 256
#define PK 4
#define IK 8  // keys per query for SearchForInitialization
#define P_NO_KEY 0xffffffffffffffffull
#define P_MAX_FEATS 8192

#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)


__device__ __forceinline__ int key_dist(unsigned long long k) { return (int)(k >> 48); }
__device__ __forceinline__ int key_idx(unsigned long long k) { return (int)(k & 0xffff); }

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

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
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

// The 64 lanes of a wave walk the window of query q: lane l takes the window cells l, l + 64, ... (rank c in the
// reference's ix-outer / iy-inner order).  VISIT(idx, c, kpos) runs for every feature that passes the geometric filters of
// GetFeaturesInArea (size band, |dx| < r, |dy| < r); (c, kpos) orders the candidates like the reference's vIndices.
#define PROJ_WAVE_WINDOW(J, q, lane, VISIT)                                                           \
    {                                                                                                 \
        const float x_ = J.qu[q], y_ = J.qv[q], r_ = J.qr[q], mn_ = J.qmin[q], mx_ = J.qmax[q];       \
        const bool sg_ = J.stereo_gate != 0;                                                          \
        const float qur_ = sg_ ? J.q_ur[q] : 0.0f, qer_ = sg_ ? J.q_er[q] : 0.0f;                      \
        const Window w_ = proj_window(J, x_, y_, r_);                                                 \
        if (w_.ok) {                                                                                  \
            const int ny_ = w_.cy1 - w_.cy0 + 1, ncells_ = (w_.cx1 - w_.cx0 + 1) * ny_;               \
            for (int c = lane; c < ncells_; c += 64) {                                                \
                const int cell_ = (w_.cx0 + c / ny_) * J.rows + (w_.cy0 + c % ny_);                   \
                const int kb_ = J.cell_ptr[cell_], ke_ = J.cell_ptr[cell_ + 1];                       \
                for (int k_ = kb_; k_ < ke_; ++k_) {                                                  \
                    const int idx = J.cell_idx[k_];                                                   \
                    const float s_ = J.size[idx];                                                     \
                    if (s_ < mn_ || s_ > mx_) continue;                                               \
                    if (!(fabsf(J.x[idx] - x_) < r_ && fabsf(J.y[idx] - y_) < r_)) continue;          \
                    if (sg_) {                                                                        \
                        const float ur_ = J.u_right[idx];                                             \
                        if (ur_ > 0.0f && fabsf(qur_ - ur_) > qer_) continue;                          \
                    }                                                                                 \
                    const int kpos = k_ - kb_;                                                        \
                    VISIT                                                                             \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    }

__device__ __forceinline__ unsigned long long make_key(int d, int c, int kpos, int idx) {
    return ((unsigned long long)d << 48) | ((unsigned long long)(c & 0xffff) << 32) | ((unsigned long long)(kpos & 0xffff) << 16) |
           (unsigned)idx;
}

// ---------------- phase 1: K best keys per query, one wave per query ----------------
template <int W, int K>
__device__ void topk_query(const DevProjJob &J, int q, int lane) {
    unsigned long long k[K];
#pragma unroll
    for (int s = 0; s < K; ++s) k[s] = P_NO_KEY;
    int visited = 0;
    if (!J.qvalid || J.qvalid[q]) {
        uint32_t qd[W];
#pragma unroll
        for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
        PROJ_WAVE_WINDOW(J, q, lane, {
            unsigned long long key = make_key(proj_hamming<W>(qd, J.fdesc + (size_t)idx * W), c, kpos, idx);
            _Pragma("unroll") for (int s = 0; s < K; ++s) {
                if (key < k[s]) {
                    const unsigned long long t = k[s];
                    k[s] = key;
                    key = t;
                }
            }
            ++visited;
        })
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) visited += __shfl_xor(visited, o, 64);
    unsigned long long mine = P_NO_KEY;  // lane s ends up holding the query's s-th best key
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const unsigned long long m = wave_min_u64(k[0]);
        if (lane == s) mine = m;
        if (k[0] == m && m != P_NO_KEY) {  // keys are unique (the feature is part of the key): exactly one lane pops
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) k[t] = k[t + 1];
            k[K - 1] = P_NO_KEY;
        }
    }
    if (K == PK) {
        // projection searches: a 64-byte record per query so that the ordered phase never touches global memory on its fast
        // path: 4 keys | 4 x (candidate size [mode 0] or rotation bin [mode 1]) | #candidates | "occupies" flag
        uint32_t *rec = reinterpret_cast<uint32_t *>(J.keys) + (size_t)q * 16;
        if (lane < PK) {
            reinterpret_cast<unsigned long long *>(rec)[lane] = mine;
            uint32_t aux = 0;
            if (mine != P_NO_KEY) {
                const int idx = key_idx(mine);
                if (J.mode == 0) aux = __float_as_uint(J.size[idx]);
                else if (J.check_ori) aux = (uint32_t)proj_rotation_bin(J.qangle[q], J.angle[idx]);
            }
            rec[8 + lane] = aux;
        }
        if (lane == 0) {
            rec[12] = (uint32_t)visited;
            rec[13] = (!J.qocc || J.qocc[q]) ? 1u : 0u;
        }
    } else {
        if (lane < K) J.keys[(size_t)q * K + lane] = mine;
        if (lane == 0) J.ncand[q] = visited;
    }
}

template <int K>
__global__ __launch_bounds__(PT) void k_proj_topk(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, q = blockIdx.x * (PT / 64) + (threadIdx.x >> 6);
    if (q >= J.nq) return;
    if (J.words == 8) topk_query<8, K>(J, q, lane);
    else topk_query<16, K>(J, q, lane);
}

// ---------------- phase 2: ordered resolve, one wave per job ----------------
// dynamic LDS: claim table [P_MAX_FEATS] | occupancy bitset | histogram | (when the job fits) the queries' records, so that
// a replay round costs LDS latency only
#define PR_LDS_FIXED (P_MAX_FEATS * 4 + P_MAX_FEATS / 8 + 32 * 4)
#define PR_REC_BYTES 64
template <int W>
__device__ void proj_resolve(const DevProjJob &J, int stage_cap) {
    extern __shared__ __attribute__((aligned(16))) char pr_smem[];
    int *s_claim = reinterpret_cast<int *>(pr_smem);
    uint32_t *s_occ = reinterpret_cast<uint32_t *>(pr_smem + P_MAX_FEATS * 4);
    int *s_hist = reinterpret_cast<int *>(pr_smem + P_MAX_FEATS * 4 + P_MAX_FEATS / 8);
    uint4 *s_rec = reinterpret_cast<uint4 *>(pr_smem + PR_LDS_FIXED);
    const int tid = threadIdx.x, lane = tid & 63;
    const bool staged = J.nq <= stage_cap;
    // ---- all four waves: tables + staging ----
    for (int i = tid; i < J.n; i += PT) {
        J.assign[i] = -1;
        s_claim[i] = 0x7fffffff;
    }
    for (int i0 = (tid >> 6) * 64; i0 < J.n; i0 += PT) {  // occupancy bytes -> bitset, one coalesced load + ballot per 64 features
        const int i = i0 + lane;
        const unsigned long long m = __ballot(J.occupied && i < J.n && J.occupied[i] != 0);
        if (lane == 0) {
            s_occ[i0 >> 5] = (uint32_t)m;
            s_occ[(i0 >> 5) + 1] = (uint32_t)(m >> 32);
        }
    }
    if (tid < 32) s_hist[tid] = 0;
    const uint4 *grec = reinterpret_cast<const uint4 *>(J.keys);
    if (staged)
        for (int i = tid; i < J.nq * (PR_REC_BYTES / 16); i += PT) s_rec[i] = grec[i];
    __syncthreads();
    if (tid >= 64) return;
    // ---- wave 0: ordered walk ----
    const uint4 *rp = staged ? s_rec : grec;
    int nm = 0, nori = 0, pos = 0;
#ifdef AFV_PROJ_STATS
    int st_rounds = 0, st_rescans = 0;
#endif
    while (pos < J.nq) {
#ifdef AFV_PROJ_STATS
        ++st_rounds;
#endif
        const int q = pos + lane;
        const bool act = q < J.nq;  // an invalid query has no keys
        unsigned long long k[PK] = {P_NO_KEY, P_NO_KEY, P_NO_KEY, P_NO_KEY};
        uint32_t aux[PK] = {0, 0, 0, 0};
        int visited = 0;
        bool occupies = true;
        if (act) {
            const uint4 r0 = rp[(size_t)q * 4], r1 = rp[(size_t)q * 4 + 1], r2 = rp[(size_t)q * 4 + 2], r3 = rp[(size_t)q * 4 + 3];
            k[0] = ((unsigned long long)r0.y << 32) | r0.x;
            k[1] = ((unsigned long long)r0.w << 32) | r0.z;
            k[2] = ((unsigned long long)r1.y << 32) | r1.x;
            k[3] = ((unsigned long long)r1.w << 32) | r1.z;
            aux[0] = r2.x; aux[1] = r2.y; aux[2] = r2.z; aux[3] = r2.w;
            visited = (int)r3.x;
            occupies = (r3.y & 1u) != 0;
        }
        int e0 = -1, e1 = -1, d0 = 0, d1 = 0;
        uint32_t a0 = 0, a1 = 0;
        bool open = act, exhausted = act;
#pragma unroll
        for (int s = 0; s < PK; ++s) {
            if (open) {
                if (k[s] == P_NO_KEY) {
                    open = false;
                    exhausted = false;
                } else {
                    const int idx = key_idx(k[s]);
                    if (!((s_occ[idx >> 5] >> (idx & 31)) & 1u)) {
                        if (e0 < 0) {
                            e0 = idx;
                            d0 = key_dist(k[s]);
                            a0 = aux[s];
                            if (J.mode == 1) {  // best only
                                open = false;
                                exhausted = false;
                            }
                        } else {
                            e1 = idx;
                            d1 = key_dist(k[s]);
                            a1 = aux[s];
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
                    if ((float)d0 > J.ratio * (float)d1) {
                        const float bs = __uint_as_float(a0), bs2 = __uint_as_float(a1);
                        if ((bs / bs2 < J.tol) && (bs / bs2 > J.inv_tol) && (bs2 > 0.0f)) type = 0;
                    } else {
                        // the distance test passes against this second best; any later second is at least as far, so the
                        // outcome no longer depends on e1 staying free
                        e1 = -1;
                    }
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
        const unsigned long long cm = __ballot(commit);
        if (commit) {
            J.assign[e0] = q;
            if (occupies) atomicOr(&s_occ[e0 >> 5], 1u << (e0 & 31));
            if (J.mode == 1 && J.check_ori) {
                const int slot = nori + __popcll(cm & ((1ull << lane) - 1ull));
                J.orilist[2 * slot] = e0;
                J.orilist[2 * slot + 1] = (int)a0;
                atomicAdd(&s_hist[a0], 1);
            }
        }
        nm += __popcll(cm);
        nori += __popcll(cm);
        if (type == 1) s_claim[e0] = 0x7fffffff;
        WAVE_LDS_SYNC();
        if (stop == 0) {
#ifdef AFV_PROJ_STATS
            ++st_rescans;
#endif
            // exact rescan of the first query's window against the current occupancy, lanes over the window's cells
            const int q0 = pos;
            uint32_t qd[W];
#pragma unroll
            for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q0 * W + i];
            unsigned long long k0 = P_NO_KEY, k1 = P_NO_KEY;
            PROJ_WAVE_WINDOW(J, q0, lane, {
                if ((s_occ[idx >> 5] >> (idx & 31)) & 1u) continue;
                const unsigned long long key = make_key(proj_hamming<W>(qd, J.fdesc + (size_t)idx * W), c, kpos, idx);
                if (key < k0) {
                    k1 = k0;
                    k0 = key;
                } else if (key < k1) {
                    k1 = key;
                }
            })
            const unsigned long long g0 = wave_min_u64(k0);
            const unsigned long long g1 = wave_min_u64(k0 == g0 ? k1 : k0);
            if (g0 != P_NO_KEY) {
                const float best = (float)key_dist(g0);
                const int bidx = key_idx(g0);
                bool ok = best <= J.th;
                if (ok && J.mode == 0 && g1 != P_NO_KEY) {
                    const float best2 = (float)key_dist(g1), bsz = J.size[bidx], bsz2 = J.size[key_idx(g1)];
                    if ((bsz / bsz2 < J.tol) && (bsz / bsz2 > J.inv_tol) && (bsz2 > 0.0f) && (best > J.ratio * best2)) ok = false;
                }
                if (ok) {
                    if (lane == 0) {
                        J.assign[bidx] = q0;
                        if (!J.qocc || J.qocc[q0]) s_occ[bidx >> 5] |= 1u << (bidx & 31);
                        if (J.mode == 1 && J.check_ori) {
                            const int bin = proj_rotation_bin(J.qangle[q0], J.angle[bidx]);
                            J.orilist[2 * nori] = bidx;
                            J.orilist[2 * nori + 1] = bin;
                            s_hist[bin]++;
                        }
                    }
                    nm += 1;
                    nori += 1;
                }
            }
            WAVE_LDS_SYNC();
            pos += 1;
        } else {
            pos += stop;
        }
    }
    if (J.mode == 1 && J.check_ori) {
        // filterMatchesWithOrientation (Pt flavour, FeatureMatcher.cc:1601-1613) over the accepted-match list
        __threadfence_block();
        WAVE_LDS_SYNC();
        int i1 = -1, i2 = -1, i3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < 30; ++i) {
            const int sz = s_hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
            else if (sz > max3) { max3 = sz; i3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
        int dropped = 0;
        for (int i = lane; i < nori; i += 64) {
            const int b = J.orilist[2 * i + 1];
            if (b != i1 && b != i2 && b != i3) {
                J.assign[J.orilist[2 * i]] = -1;
                ++dropped;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dropped += __shfl_xor(dropped, o, 64);
        nm -= dropped;
    }
#ifdef AFV_PROJ_STATS
    if (lane == 0) printf("proj_resolve: mode %d nq %d rounds %d rescans %d matches %d\n", J.mode, J.nq, st_rounds, st_rescans, nm);
#endif
    if (lane == 0) *J.nmatches = nm;
}

__global__ __launch_bounds__(PT) void k_proj_resolve(const DevProjJob *__restrict__ jobs, int stage_cap) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.words == 8) proj_resolve<8>(J, stage_cap);
    else proj_resolve<16>(J, stage_cap);
}

// ---------------- Fuse / SearchBySim3: independent queries, one wave each; first minimum in visiting order (:905) ----------------
template <int W>
__device__ void fuse_query(const DevProjJob &J, int q, int lane) {
    unsigned long long k0 = P_NO_KEY;
    if (!J.qvalid || J.qvalid[q]) {
        uint32_t qd[W];
#pragma unroll
        for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
        const float u = J.qu[q], v = J.qv[q];
        const float qur = J.u_right ? J.q_ur[q] : 0.0f;
        PROJ_WAVE_WINDOW(J, q, lane, {
            if (J.inf) {  // reprojection gate of Fuse (:876-900); absent in Fuse(Sim3) / SearchBySim3
                const float ex = u - J.x[idx];
                const float ey = v - J.y[idx];
                const float kpr = J.u_right ? J.u_right[idx] : -1.0f;
                if (kpr >= 0.0f) {  // stereo keypoint: three degrees of freedom (:880-894)
                    const float er = qur - kpr;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if ((double)(e2 * J.inf[idx]) > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if ((double)(e2 * J.inf[idx]) > 5.99) continue;
                }
            }
            const unsigned long long key = make_key(proj_hamming<W>(qd, J.fdesc + (size_t)idx * W), c, kpos, idx);
            k0 = key < k0 ? key : k0;
        })
    }
    const unsigned long long g0 = wave_min_u64(k0);
    if (lane == 0) J.assign[q] = (g0 != P_NO_KEY && (float)key_dist(g0) <= J.th) ? key_idx(g0) : -1;
}

__global__ __launch_bounds__(PT) void k_match_fuse(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, q = blockIdx.x * (PT / 64) + (threadIdx.x >> 6);
    if (q >= J.nq) return;
    if (J.words == 8) fuse_query<8>(J, q, lane);
    else fuse_query<16>(J, q, lane);
}

// ---------------- SearchForInitialization (FeatureMatcher.cc:399-557, active part :480-556) ----------------
// queries = level-0 features of F1 searched in a fixed window around vbPrevMatched in F2.  Inherently ordered: a
// candidate is skipped when an earlier query already matched it at a distance <= the current one (:513), and a later
// query steals the feature (:531-535).  k_proj_topk<IK> ranks every query's window in parallel; one wave then walks the
// queries in order.  The gate only ever removes candidates, so best / second are the first two ungated keys; when the
// IK keys run out before two are found (and the window holds more) the window is rescanned exactly.
template <int W>
__device__ void init_resolve(const DevProjJob &J) {
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
    __threadfence_block();
    WAVE_LDS_SYNC();
    int nori = 0;
    const int sub = lane & (IK - 1);
    for (int base = 0; base < J.nq; base += 64 / IK) {
        // 8 queries x 8 keys per pass: lane = (query in pass) * 8 + key slot
        const int ql = base + lane / IK;
        unsigned long long kreg = P_NO_KEY;
        int ncand_l = 0;
        bool valid_l = false;
        if (ql < J.nq) {
            valid_l = !J.qvalid || J.qvalid[ql];
            if (valid_l) {
                kreg = J.keys[(size_t)ql * IK + sub];
                ncand_l = J.ncand[ql];
            }
        }
        unsigned long long vm = __ballot(valid_l && sub == 0 && ncand_l > 0);
        while (vm) {
            const int j = (int)__builtin_ctzll(vm) / IK;  // next live query of this pass
            vm &= vm - 1;
            const int q = base + j;
            const unsigned long long key = __shfl(kreg, j * IK + sub, 64);  // every lane group of 8 sees the 8 keys
            const int ncand = __shfl(ncand_l, j * IK, 64);
            const int ki = key == P_NO_KEY ? 0 : key_idx(key);
            const bool okk = key != P_NO_KEY && !((int)s_mdist[ki] <= key_dist(key));  // gate (:513)
            const unsigned m8 = (unsigned)(__ballot(okk) & 0xffull);  // lanes 0..7 hold slots 0..7
            unsigned long long g0 = P_NO_KEY, g1 = P_NO_KEY;
            if (__popc(m8) >= 2 || ncand <= IK) {
                if (m8) {
                    const int s0 = __builtin_ctz(m8);
                    g0 = __shfl(key, s0, 64);
                    const unsigned r = m8 & (m8 - 1);
                    if (r) g1 = __shfl(key, __builtin_ctz(r), 64);
                }
            } else {
                // exact rescan of the window with the gate applied, lanes over cells
                uint32_t qd[W];
#pragma unroll
                for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
                unsigned long long k0 = P_NO_KEY, k1 = P_NO_KEY;
                PROJ_WAVE_WINDOW(J, q, lane, {
                    const int d = proj_hamming<W>(qd, J.fdesc + (size_t)idx * W);
                    if ((int)s_mdist[idx] <= d) continue;
                    const unsigned long long kk = make_key(d, c, kpos, idx);
                    if (kk < k0) {
                        k1 = k0;
                        k0 = kk;
                    } else if (kk < k1) {
                        k1 = kk;
                    }
                })
                g0 = wave_min_u64(k0);
                g1 = wave_min_u64(k0 == g0 ? k1 : k0);
            }
            if (g0 == P_NO_KEY) continue;
            const float best = (float)key_dist(g0);
            const float best2 = g1 == P_NO_KEY ? 3.402823466e+38f : (float)key_dist(g1);
            const int bidx = key_idx(g0);
            if (best <= J.th && best < best2 * J.ratio) {  // :527-529
                if (lane == 0) {
                    const int prev = s_m21[bidx];
                    if (prev >= 0) J.assign[prev] = -1;  // stolen (:531-535)
                    J.assign[q] = bidx;
                    s_m21[bidx] = q;
                    s_mdist[bidx] = (unsigned short)key_dist(g0);
                    if (J.check_ori) {
                        const int bin = proj_rotation_bin(J.qangle[q], J.angle[bidx]);  // F1 keypoint first (:543)
                        J.orilist[2 * nori] = q;
                        J.orilist[2 * nori + 1] = bin;
                        s_hist[bin]++;
                    }
                }
                nori += 1;
                WAVE_LDS_SYNC();
            }
        }
    }
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
    // nMatches = entries still standing (accepts minus steals minus orientation drops)
    int cnt = 0;
    for (int q = lane; q < J.nq; q += 64) cnt += J.assign[q] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (lane == 0) *J.nmatches = cnt;
}

__global__ __launch_bounds__(64) void k_init_resolve(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.words == 8) init_resolve<8>(J);
    else init_resolve<16>(J);
}

extern "C" void afv_launch_match_init(const DevProjJob *jobs, int njobs, int max_nq, hipStream_t stream) {
    if (max_nq > 0) hipLaunchKernelGGL(k_proj_topk<IK>, dim3((max_nq + PT / 64 - 1) / (PT / 64), njobs), dim3(PT), 0, stream, jobs);
    hipLaunchKernelGGL(k_init_resolve, dim3(njobs), dim3(64), 0, stream, jobs);
}

extern "C" void afv_launch_match_fuse(const DevProjJob *jobs, int njobs, int max_nq, hipStream_t stream) {
    if (max_nq > 0) hipLaunchKernelGGL(k_match_fuse, dim3((max_nq + PT / 64 - 1) / (PT / 64), njobs), dim3(PT), 0, stream, jobs);
}

extern "C" void afv_launch_match_projection(const DevProjJob *jobs, int njobs, int max_nq, hipStream_t stream) {
    if (max_nq > 0) hipLaunchKernelGGL(k_proj_topk<PK>, dim3((max_nq + PT / 64 - 1) / (PT / 64), njobs), dim3(PT), 0, stream, jobs);
    // stage the query records in LDS when the largest job fits (64 B per query next to the 33.9 KB of tables; 160 KB per CU)
    int stage_cap = max_nq;
    if ((size_t)PR_LDS_FIXED + (size_t)stage_cap * PR_REC_BYTES > 128 * 1024) stage_cap = 0;
    hipLaunchKernelGGL(k_proj_resolve, dim3(njobs), dim3(PT), PR_LDS_FIXED + (size_t)stage_cap * PR_REC_BYTES, stream, jobs, stage_cap);
}
