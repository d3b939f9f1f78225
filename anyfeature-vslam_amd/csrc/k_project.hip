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
//   2. the ordered phase, two engines with identical results (afv_set_projection_resolve):
//      k_proj_resolve_wg (round 5, default) — ONE fixed point over all live queries of a job on a 1024-thread workgroup, the recipe of
//      k_match_resolve_wg (k_match.hip): a query's decision is a function of the features wanted by the OCCUPYING queries before it, so any
//      assignment that satisfies all these equations is the sequential outcome; every pass all live queries re-evaluate against the
//      claims of the previous pass (three rotating LDS arrays, one barrier per pass); queries whose keys are used up are rescanned
//      exactly after convergence, in order, up to the first that takes a feature.
//      k_proj_resolve (rounds 1-4) — one wave per job replays the queries in order in speculative 64-query rounds (claim / replay on
//      the feature-occupancy bitset); a query that runs out of keys is rescanned exactly, again with lanes over cells.
// The grid arrives as cell_ptr + cell_ent (k_frame.hip builds it on the device): an entry carries the feature's index, position and
// keyPtsSize, so a candidate costs ONE dependent load before its descriptor instead of two (index, then x / y / size).
#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch
#include "afv_jobs.h"

#include <type_traits>

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

// wave-wide minimum on DPP (row shifts inside the rows of 16 lanes, then row_bcast15 / row_bcast31 carry the row results: no LDS
// crossbar - the __shfl_xor butterfly of rounds 1-4 was six dependent ds_bpermute pairs per call); every lane receives the result
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#define AFV_MIN64_STEP(ctrl, rmask)                                                                                       \
    {                                                                                                                     \
        const unsigned lo_ = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)v, ctrl, rmask, 0xf, false);        \
        const unsigned hi_ = (unsigned)__builtin_amdgcn_update_dpp(-1, (int)(unsigned)(v >> 32), ctrl, rmask, 0xf, false); \
        const unsigned long long t_ = ((unsigned long long)hi_ << 32) | lo_;                                              \
        v = t_ < v ? t_ : v;                                                                                              \
    }
    AFV_MIN64_STEP(0x111, 0xf)  // row_shr:1
    AFV_MIN64_STEP(0x112, 0xf)  // row_shr:2
    AFV_MIN64_STEP(0x114, 0xf)  // row_shr:4
    AFV_MIN64_STEP(0x118, 0xf)  // row_shr:8
    AFV_MIN64_STEP(0x142, 0xa)  // row_bcast:15 into rows 1, 3
    AFV_MIN64_STEP(0x143, 0xc)  // row_bcast:31 into rows 2, 3
#undef AFV_MIN64_STEP
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
#define AFV_MIN32_STEP(ctrl, rmask) v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, ctrl, rmask, 0xf, false));
    AFV_MIN32_STEP(0x111, 0xf)
    AFV_MIN32_STEP(0x112, 0xf)
    AFV_MIN32_STEP(0x114, 0xf)
    AFV_MIN32_STEP(0x118, 0xf)
    AFV_MIN32_STEP(0x142, 0xa)
    AFV_MIN32_STEP(0x143, 0xc)
#undef AFV_MIN32_STEP
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(afv_wave_incl_scan(v), 63); }

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
// Cells are taken PW_CPL at a time per lane: their (begin, end) pairs are loaded first, all in flight together, then their entries are
// walked - a 100-pixel window (SearchForInitialization: ~400 cells, 6-7 per lane) was a chain of 6-7 x 2 dependent loads per lane.
#define PW_CPL 4
#define PROJ_WAVE_WINDOW(J, q, lane, VISIT)                                                           \
    {                                                                                                 \
        const float x_ = J.qu[q], y_ = J.qv[q], r_ = J.qr[q], mn_ = J.qmin[q], mx_ = J.qmax[q];       \
        const bool sg_ = J.stereo_gate != 0;                                                          \
        const float qur_ = sg_ ? J.q_ur[q] : 0.0f, qer_ = sg_ ? J.q_er[q] : 0.0f;                      \
        const Window w_ = proj_window(J, x_, y_, r_);                                                 \
        if (w_.ok) {                                                                                  \
            const int ny_ = w_.cy1 - w_.cy0 + 1, ncells_ = (w_.cx1 - w_.cx0 + 1) * ny_;               \
            for (int cb_ = 0; cb_ < ncells_; cb_ += 64 * PW_CPL) {                                    \
                int kb_[PW_CPL], ke_[PW_CPL];                                                         \
                _Pragma("unroll") for (int u_ = 0; u_ < PW_CPL; ++u_) {                               \
                    const int cc_ = cb_ + u_ * 64 + lane;                                             \
                    kb_[u_] = 0;                                                                      \
                    ke_[u_] = 0;                                                                      \
                    if (cc_ < ncells_) {                                                              \
                        const int cxo_ = cc_ / ny_;                                                   \
                        const int cell_ = (w_.cx0 + cxo_) * J.rows + (w_.cy0 + cc_ - cxo_ * ny_);     \
                        kb_[u_] = J.cell_ptr[cell_];                                                  \
                        ke_[u_] = J.cell_ptr[cell_ + 1];                                              \
                    }                                                                                 \
                }                                                                                     \
                _Pragma("unroll") for (int u_ = 0; u_ < PW_CPL; ++u_) {                               \
                    const int c = cb_ + u_ * 64 + lane;                                               \
                    for (int k_ = kb_[u_]; k_ < ke_[u_]; ++k_) {                                      \
                        const int4 e_ = J.cell_ent[k_];                                               \
                        const int idx = e_.x;                                                         \
                        const float fx_ = __int_as_float(e_.y), fy_ = __int_as_float(e_.z);           \
                        const float s_ = __int_as_float(e_.w);                                        \
                        if (s_ < mn_ || s_ > mx_) continue;                                           \
                        if (!(fabsf(fx_ - x_) < r_ && fabsf(fy_ - y_) < r_)) continue;                \
                        if (sg_) {                                                                    \
                            const float ur_ = J.u_right[idx];                                         \
                            if (ur_ > 0.0f && fabsf(qur_ - ur_) > qer_) continue;                      \
                        }                                                                             \
                        const int kpos = k_ - kb_[u_], epos = k_;                                     \
                        (void)epos, (void)kpos, (void)c;                                              \
                        VISIT                                                                         \
                    }                                                                                 \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    }

// The ranking kernels walk the window DENSELY: the candidates of a chunk of cells (64 x PW_CPL cells) are first listed in LDS - every lane
// appends the entry ranges of its cells at the offset a wave scan of the counts gives it - and then dealt to the lanes round robin.  In the
// per-lane form above the wavefront runs as many iterations as its busiest lane has candidates, each a dependent entry -> descriptor load
// pair with most lanes idle (SearchForInitialization, 100-pixel windows: ~20 iterations of ~1.5 us; PMC: 114 vector-memory instructions
// per live wavefront); here it runs ceil(candidates / 64) of them with all lanes loading.  A chunk with more than PW_LIST candidates
// (never at the reference's 64 x 48 grid and feature budgets) falls back to the per-lane walk for that chunk.
#define PW_LIST 256
#define PROJ_WAVE_WINDOW_DENSE(J, q, lane, S_LIST, VISIT)                                             \
    {                                                                                                 \
        const float x_ = J.qu[q], y_ = J.qv[q], r_ = J.qr[q], mn_ = J.qmin[q], mx_ = J.qmax[q];       \
        const bool sg_ = J.stereo_gate != 0;                                                          \
        const float qur_ = sg_ ? J.q_ur[q] : 0.0f, qer_ = sg_ ? J.q_er[q] : 0.0f;                      \
        const Window w_ = proj_window(J, x_, y_, r_);                                                 \
        if (w_.ok) {                                                                                  \
            const int ny_ = w_.cy1 - w_.cy0 + 1, ncells_ = (w_.cx1 - w_.cx0 + 1) * ny_;               \
            for (int cb_ = 0; cb_ < ncells_; cb_ += 64 * PW_CPL) {                                    \
                int kb_[PW_CPL], ke_[PW_CPL];                                                         \
                int cnt_ = 0;                                                                         \
                _Pragma("unroll") for (int u_ = 0; u_ < PW_CPL; ++u_) {                               \
                    const int cc_ = cb_ + u_ * 64 + lane;                                             \
                    kb_[u_] = 0;                                                                      \
                    ke_[u_] = 0;                                                                      \
                    if (cc_ < ncells_) {                                                              \
                        const int cxo_ = cc_ / ny_;                                                   \
                        const int cell_ = (w_.cx0 + cxo_) * J.rows + (w_.cy0 + cc_ - cxo_ * ny_);     \
                        kb_[u_] = J.cell_ptr[cell_];                                                  \
                        ke_[u_] = J.cell_ptr[cell_ + 1];                                              \
                    }                                                                                 \
                    cnt_ += ke_[u_] - kb_[u_];                                                        \
                }                                                                                     \
                const int incl_ = afv_wave_incl_scan(cnt_);                                           \
                const int total_ = __builtin_amdgcn_readlane(incl_, 63);                              \
                if (total_ == 0) continue;                                                            \
                if (total_ <= PW_LIST) {                                                              \
                    int at_ = incl_ - cnt_;                                                           \
                    _Pragma("unroll") for (int u_ = 0; u_ < PW_CPL; ++u_)                             \
                        for (int k_ = kb_[u_]; k_ < ke_[u_]; ++k_)                                    \
                            S_LIST[at_++] = make_int2(k_, ((cb_ + u_ * 64 + lane) << 16) | (k_ - kb_[u_])); \
                    WAVE_LDS_SYNC();                                                                  \
                    for (int i_ = lane; i_ < total_; i_ += 64) {                                      \
                        const int2 ck_ = S_LIST[i_];                                                  \
                        const int4 e_ = J.cell_ent[ck_.x];                                            \
                        const int idx = e_.x;                                                         \
                        const float fx_ = __int_as_float(e_.y), fy_ = __int_as_float(e_.z);           \
                        const float s_ = __int_as_float(e_.w);                                        \
                        if (s_ < mn_ || s_ > mx_) continue;                                           \
                        if (!(fabsf(fx_ - x_) < r_ && fabsf(fy_ - y_) < r_)) continue;                \
                        if (sg_) {                                                                    \
                            const float ur_ = J.u_right[idx];                                         \
                            if (ur_ > 0.0f && fabsf(qur_ - ur_) > qer_) continue;                      \
                        }                                                                             \
                        const int c = (int)((unsigned)ck_.y >> 16), kpos = ck_.y & 0xffff, epos = ck_.x; \
                        (void)epos, (void)kpos, (void)c;                                              \
                        VISIT                                                                         \
                    }                                                                                 \
                    WAVE_LDS_SYNC();                                                                  \
                } else {                                                                              \
                    _Pragma("unroll") for (int u_ = 0; u_ < PW_CPL; ++u_) {                           \
                        const int c = cb_ + u_ * 64 + lane;                                           \
                        for (int k_ = kb_[u_]; k_ < ke_[u_]; ++k_) {                                  \
                            const int4 e_ = J.cell_ent[k_];                                           \
                            const int idx = e_.x;                                                     \
                            const float fx_ = __int_as_float(e_.y), fy_ = __int_as_float(e_.z);       \
                            const float s_ = __int_as_float(e_.w);                                    \
                            if (s_ < mn_ || s_ > mx_) continue;                                       \
                            if (!(fabsf(fx_ - x_) < r_ && fabsf(fy_ - y_) < r_)) continue;            \
                            if (sg_) {                                                                \
                                const float ur_ = J.u_right[idx];                                     \
                                if (ur_ > 0.0f && fabsf(qur_ - ur_) > qer_) continue;                  \
                            }                                                                         \
                            const int kpos = k_ - kb_[u_], epos = k_;                                 \
                            (void)epos, (void)kpos, (void)c;                                          \
                            VISIT                                                                     \
                        }                                                                             \
                    }                                                                                 \
                }                                                                                     \
            }                                                                                         \
        }                                                                                             \
    }

__device__ __forceinline__ unsigned long long make_key(int d, int c, int kpos, int idx) {
    return ((unsigned long long)d << 48) | ((unsigned long long)(c & 0xffff) << 32) | ((unsigned long long)(kpos & 0xffff) << 16) |
           (unsigned)idx;
}

// Float descriptors (round 6; W == 0 in the templates below).  FeatureMatcher::DescriptorDistance dispatches on DescriptorType
// (FeatureMatcher.cc:1508-1531): SIFT128 / SURF64 / KAZE64 / R2D2 ... return cv::norm(a, b, NORM_L2SQR) narrowed to Descriptor_Distance_Type =
// float (Feature_sift128.cpp:132-134, Types.h:127).  The distance is evaluated as k_match_l2.hip does (float differences, squares and 4-way
// partial sums in double, one rounding to float at the end: the summation order is part of the result), and a non-negative float's bit
// pattern orders like the number, so the candidate key is  distance bits << 32 | entry position << 16 | feature:  the position of the
// candidate in cell_ent is monotonic in the reference's visiting order (cells in ix-outer / iy-inner order ARE ascending cell indices, and
// the entries are stored cell by cell), < 8192, and breaks distance ties exactly as (window cell rank, position in cell) does for the
// binary keys.  Rows are J.fdim floats (a multiple of 4, 16-byte aligned); only the ordered-walk engines (REC 0 / 2) read these keys.
__device__ __forceinline__ float proj_l2sqr(const float *__restrict__ a, const float *__restrict__ b, int dim) {
    double s = 0;
    for (int i = 0; i < dim; i += 4) {
        const float4 x = *reinterpret_cast<const float4 *>(a + i), y = *reinterpret_cast<const float4 *>(b + i);
        const double v0 = (double)(x.x - y.x), v1 = (double)(x.y - y.y), v2 = (double)(x.z - y.z), v3 = (double)(x.w - y.w);
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    return (float)s;
}
__device__ __forceinline__ const float *proj_frow(const DevProjJob &J, int idx) { return reinterpret_cast<const float *>(J.fdesc) + (size_t)idx * J.fdim; }
__device__ __forceinline__ const float *proj_qrow(const DevProjJob &J, int q) { return reinterpret_cast<const float *>(J.qdesc) + (size_t)q * J.fdim; }
__device__ __forceinline__ unsigned long long make_key_f32(float d, int epos, int idx) {
    return ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned long long)(epos & 0xffff) << 16) | (unsigned)idx;
}
// the distance a key carries, as the type the reference computes with for that descriptor kind (int widened to float at the comparisons)
template <int W>
__device__ __forceinline__ auto key_dist_of(unsigned long long k) {
    if constexpr (W == 0) return __uint_as_float((unsigned)(k >> 32));
    else return key_dist(k);
}

// ---------------- phase 1: K best keys per query, one wave per query ----------------
// REC selects what the ordered phase reads:
//   0  projection, ordered walk (k_proj_resolve): 64-byte record = 4 keys | 4 aux | #candidates | "occupies"
//   1  projection, fixed point (k_proj_resolve_wg): 32-byte record = 4 x (distance << 16 | feature) | #candidates | "occupies" | 0 | 0
//   2  initialization, ordered walk (k_init_resolve): IK 64-bit keys + ncand
//   3  initialization, fixed point (k_init_resolve_wg): IK x (distance << 16 | feature) + ncand
// Features that are occupied before the search starts (F.pts[i] with observations, :108-110, :1361-1363) never become free again: they
// are dropped here, so the ordered phase only deals with what the queries of THIS call take from each other.
#define PROJ_NO_KEY32 0xffffffffu
template <int W, int K, int REC>
__device__ void topk_query(const DevProjJob &J, int q, int lane, int2 *s_list /* this wavefront's PW_LIST candidate slots in LDS */) {
    unsigned long long k[K];
#pragma unroll
    for (int s = 0; s < K; ++s) k[s] = P_NO_KEY;
    int visited = 0;
    if constexpr (W == 0) {
        if constexpr (REC == 3) return;  // SearchForInitialization's fixed point packs distances in 16 bits: float jobs take its ordered walk (REC 2)
        if (!J.qvalid || J.qvalid[q]) {
            const float *qrow = proj_qrow(J, q);
            PROJ_WAVE_WINDOW_DENSE(J, q, lane, s_list, {
                if (REC < 2 && J.occupied && J.occupied[idx]) continue;
                unsigned long long key = make_key_f32(proj_l2sqr(qrow, proj_frow(J, idx), J.fdim), epos, idx);
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
    } else if (!J.qvalid || J.qvalid[q]) {
        uint32_t qd[W];
        {
            const uint4 *qp = reinterpret_cast<const uint4 *>(J.qdesc + (size_t)q * W);
#pragma unroll
            for (int i = 0; i < W / 4; ++i) {
                const uint4 t = qp[i];
                qd[4 * i] = t.x, qd[4 * i + 1] = t.y, qd[4 * i + 2] = t.z, qd[4 * i + 3] = t.w;
            }
        }
        PROJ_WAVE_WINDOW_DENSE(J, q, lane, s_list, {
            if (REC < 2 && J.occupied && J.occupied[idx]) continue;
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
    visited = wave_sum_i32(visited);
    // K extraction rounds: the wave minimum of the lanes' heads (keys are unique: the feature is part of the key), the holder pops
    unsigned long long mine = P_NO_KEY;  // lane s ends up holding the query's s-th best key
#pragma unroll
    for (int s = 0; s < K; ++s) {
        const unsigned long long m = wave_min_u64(k[0]);
        if (lane == s) mine = m;
        if (k[0] == m && m != P_NO_KEY) {
#pragma unroll
            for (int t = 0; t + 1 < K; ++t) k[t] = k[t + 1];
            k[K - 1] = P_NO_KEY;
        }
    }
    if (REC == 0) {
        // a 64-byte record per query so that the ordered walk never touches global memory on its fast path:
        // 4 keys | 4 x (candidate size [mode 0] or rotation bin [mode 1]) | #candidates | "occupies" flag
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
    } else if (REC == 1) {
        uint32_t *rec = reinterpret_cast<uint32_t *>(J.keys) + (size_t)q * 8;
        if constexpr (W == 0) {
            // float distances: the same 32 bytes hold 4 x distance bits | #candidates | "occupies" | the four features as 16-bit halves
            // (0xffff = no key; a feature index is < 8192)
            if (lane < PK) rec[lane] = (uint32_t)(mine >> 32);
            const uint32_t f = mine == P_NO_KEY ? 0xffffu : (uint32_t)key_idx(mine);
            const uint32_t f_next = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane + 1) & 63) * 4, (int)f);
            if (lane == 0 || lane == 2) rec[6 + (lane >> 1)] = f | (f_next << 16);
        } else {
            if (lane < PK) rec[lane] = mine == P_NO_KEY ? PROJ_NO_KEY32 : (((uint32_t)key_dist(mine) << 16) | (uint32_t)key_idx(mine));
        }
        if (lane == PK) rec[4] = (uint32_t)visited;
        if (lane == PK + 1) rec[5] = (!J.qocc || J.qocc[q]) ? 1u : 0u;
    } else if (REC == 2) {
        if (lane < K) J.keys[(size_t)q * K + lane] = mine;
        if (lane == 0) J.ncand[q] = visited;
    } else {
        uint32_t *rec = reinterpret_cast<uint32_t *>(J.keys) + (size_t)q * K;
        if (lane < K) rec[lane] = mine == P_NO_KEY ? PROJ_NO_KEY32 : (((uint32_t)key_dist(mine) << 16) | (uint32_t)key_idx(mine));
        if (lane == 0) J.ncand[q] = visited;
    }
}

template <int K, int REC>
__global__ __launch_bounds__(PT) void k_proj_topk(const DevProjJob *__restrict__ jobs) {
    __shared__ int2 s_list[PT / 64][PW_LIST];
    const DevProjJob J = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, q = blockIdx.x * (PT / 64) + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (q >= J.nq) return;
    if (J.fdim) topk_query<0, K, REC>(J, q, lane, s_list[threadIdx.x >> 6]);
    else if (J.words == 8) topk_query<8, K, REC>(J, q, lane, s_list[threadIdx.x >> 6]);
    else topk_query<16, K, REC>(J, q, lane, s_list[threadIdx.x >> 6]);
}
// one job, its record a kernel argument: a search against a resident frame uploads nothing ahead of the launch - the queries are read
// straight from the caller's pinned staging arena (a few KB over the link, once)
template <int K, int REC>
__global__ __launch_bounds__(PT) void k_proj_topk1(const DevProjJob J) {
    __shared__ int2 s_list[PT / 64][PW_LIST];
    const int lane = threadIdx.x & 63, q = blockIdx.x * (PT / 64) + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (q >= J.nq) return;
    if (J.fdim) topk_query<0, K, REC>(J, q, lane, s_list[threadIdx.x >> 6]);
    else if (J.words == 8) topk_query<8, K, REC>(J, q, lane, s_list[threadIdx.x >> 6]);
    else topk_query<16, K, REC>(J, q, lane, s_list[threadIdx.x >> 6]);
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
        int e0 = -1, e1 = -1;
        decltype(key_dist_of<W>(0ull)) d0 = 0, d1 = 0;  // int (Hamming) or float (L2^2)
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
                            d0 = key_dist_of<W>(k[s]);
                            a0 = aux[s];
                            if (J.mode == 1) {  // best only
                                open = false;
                                exhausted = false;
                            }
                        } else {
                            e1 = idx;
                            d1 = key_dist_of<W>(k[s]);
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
            unsigned long long k0 = P_NO_KEY, k1 = P_NO_KEY;
            if constexpr (W == 0) {
                const float *qrow = proj_qrow(J, q0);
                PROJ_WAVE_WINDOW(J, q0, lane, {
                    if ((s_occ[idx >> 5] >> (idx & 31)) & 1u) continue;
                    const unsigned long long key = make_key_f32(proj_l2sqr(qrow, proj_frow(J, idx), J.fdim), epos, idx);
                    if (key < k0) {
                        k1 = k0;
                        k0 = key;
                    } else if (key < k1) {
                        k1 = key;
                    }
                })
            } else {
                uint32_t qd[W];
#pragma unroll
                for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q0 * W + i];
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
            }
            const unsigned long long g0 = wave_min_u64(k0);
            const unsigned long long g1 = wave_min_u64(k0 == g0 ? k1 : k0);
            if (g0 != P_NO_KEY) {
                const float best = (float)key_dist_of<W>(g0);
                const int bidx = key_idx(g0);
                bool ok = best <= J.th;
                if (ok && J.mode == 0 && g1 != P_NO_KEY) {
                    const float best2 = (float)key_dist_of<W>(g1), bsz = J.size[bidx], bsz2 = J.size[key_idx(g1)];
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
        dropped = wave_sum_i32(dropped);
        nm -= dropped;
    }
#ifdef AFV_PROJ_STATS
    if (lane == 0) printf("proj_resolve: mode %d nq %d rounds %d rescans %d matches %d\n", J.mode, J.nq, st_rounds, st_rescans, nm);
#endif
    if (lane == 0) *J.nmatches = nm;
}

__global__ __launch_bounds__(PT) void k_proj_resolve(const DevProjJob *__restrict__ jobs, int stage_cap) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.fdim) proj_resolve<0>(J, stage_cap);
    else if (J.words == 8) proj_resolve<8>(J, stage_cap);
    else proj_resolve<16>(J, stage_cap);
}

// ---------------- phase 2, workgroup form (round 5): ONE fixed point over all live queries of a job, 1024 threads ----------------
// The ordered walk above replays the queries 64 at a time on one wavefront: a job of 1000 queries costs 16+ rounds of dependent LDS
// round trips behind workgroup-scope fences (37 us measured in round 2).  The argument of k_match_resolve_wg (k_match.hip) carries over:
// query i's decision is a function f of the features WANTED by the occupying queries before it - want_i = f({want_j : j < i, j
// occupies}) - so any assignment that satisfies all these equations IS the sequential outcome (induction on i), and iterating "every
// query re-evaluates f against the current claims" reaches it after as many passes as the longest chain of queries competing for a
// feature + 1.
//   * "feature c is taken for query i" = claim[c] < i, claim[c] = smallest live index of an OCCUPYING query that wants c (atomic min);
//     three rotating LDS arrays (read / write / clear): one barrier per pass, which also carries the "did anything change" vote;
//   * features occupied before the call never appear (phase 1 drops them);
//   * a query whose four keys are used up (and whose window holds more) needs the exact rescan of its window, meaningful once every
//     query before it is final: after convergence the waiting queries are rescanned, one per wavefront (16 per step), against the claims
//     of the queries before them; the answers are adopted in order up to and including the first that takes a feature, those queries
//     are pinned, the iteration continues.  Two shortcuts keep rescans rare: the best free key already above TH_HIGH is final, and in
//     the local-map mode a best key that passes the ratio test against the LAST key passes it against anything outside the list;
//   * F.pts[c] ends up as the LAST query that took c (a query without observations does not block later ones, :108-110): atomic max over
//     the final wants.
#define PW_T 1024
#define PW_NW (PW_T / 64)
// "did any thread change something in this pass": ONE barrier.  Three rotating flags (the pass that writes flag p % 3 clears the one the
// pass after next will use): __syncthreads_or goes through the device library's workgroup reduction (an LDS round plus two barriers).
__device__ __forceinline__ bool wg_any_changed(bool changed, int pass, int *s_vote) {
    if (__ballot(changed) && (threadIdx.x & 63) == 0) s_vote[pass % 3] = 1;
    __syncthreads();
    const bool any = s_vote[pass % 3] != 0;
    if (threadIdx.x == 0) s_vote[(pass + 2) % 3] = 0;
    return any;
}
#define PW_INF 0x7fffffff
#define PW_WLIST 128
#define PW_GUARD (-0x7fffffff)  // *nmatches when the pass guard trips (never observed; the host turns it into AFV_EHIP)

static inline size_t proj_wg_lds_bytes(int n, int nq, bool float_rows = false) {
    const size_t nr = ((size_t)n + 63) & ~(size_t)63, qr = ((size_t)nq + 63) & ~(size_t)63;
    return 3 * nr * 4 + qr * 16 /*keys*/ + qr * 4 /*meta*/ + qr * 4 /*query angle*/ + 3 * qr * 2 /*w1, w2, pin*/ + qr * 2 /*live*/ + qr /*flag*/ + 64 +
           (float_rows ? qr * 8 /*the keys' features*/ : 0);
}

// one live query against the claims in R: the feature it accepts (-1: none) and whether it needs the exact rescan
// (W == 0, float descriptors: kk = the four distances' bits, feats = the four features as 16-bit halves, 0xffff = no key)
template <int W>
__device__ __forceinline__ void proj_eval(const DevProjJob &J, const int4 kk, const int2 feats, int meta, const int *R, int li, int &want, bool &rescan) {
    const unsigned keys[PK] = {(unsigned)kk.x, (unsigned)kk.y, (unsigned)kk.z, (unsigned)kk.w};
    const bool complete = (meta & 0x7fffffff) <= PK;  // the key list holds the whole window
    int fi[PK];
    bool has[PK];
    if constexpr (W == 0) {
        fi[0] = feats.x & 0xffff, fi[1] = (int)((unsigned)feats.x >> 16), fi[2] = feats.y & 0xffff, fi[3] = (int)((unsigned)feats.y >> 16);
#pragma unroll
        for (int s = 0; s < PK; ++s) has[s] = fi[s] != 0xffff;
    } else {
#pragma unroll
        for (int s = 0; s < PK; ++s) {
            has[s] = keys[s] != PROJ_NO_KEY32;
            fi[s] = (int)(keys[s] & 0xffffu);
        }
    }
    int cl[PK];
#pragma unroll
    for (int s = 0; s < PK; ++s) cl[s] = has[s] ? R[fi[s]] : -1;  // four LDS reads in flight together
    using dist_t = std::conditional_t<W == 0, float, int>;
    auto dist_of = [&](int s) -> dist_t {
        if constexpr (W == 0) return __uint_as_float(keys[s]);
        else return (int)(keys[s] >> 16);
    };
    int e0 = -1, e1 = -1;
    dist_t d0 = 0, d1 = 0;
#pragma unroll
    for (int s = 0; s < PK; ++s) {
        const bool fr = has[s] && !(cl[s] < li);
        const int idx = fi[s];
        const dist_t d = dist_of(s);
        if (fr && e0 < 0) {
            e0 = idx;
            d0 = d;
        } else if (fr && e1 < 0) {
            e1 = idx;
            d1 = d;
        }
    }
    const float d_last = (float)dist_of(PK - 1);  // incomplete lists are full: every candidate outside is at least this far
    want = -1;
    rescan = false;
    if (e0 < 0) {
        rescan = !complete && d_last <= J.th;
    } else if (!((float)d0 <= J.th)) {
        // the best free candidate fails TH_HIGH: final
    } else if (J.mode == 1) {
        want = e0;  // best only (:1379-1393)
    } else if (e1 >= 0) {
        bool rej = false;
        if ((float)d0 > J.ratio * (float)d1) {  // FeatureMatcher.cc:142-148
            const float bs = J.size[e0], bs2 = J.size[e1];
            rej = (bs / bs2 < J.tol) && (bs / bs2 > J.inv_tol) && (bs2 > 0.0f);
        }
        want = rej ? -1 : e0;
    } else if (complete) {
        want = e0;  // no second candidate at all: bestSize2 = -1, the ratio test is skipped
    } else if (J.ratio >= 0.0f && !((float)d0 > J.ratio * d_last)) {
        want = e0;  // ratio * d2 >= ratio * d_last >= d0 for every candidate outside the list: the test cannot reject
    } else {
        rescan = true;
    }
}

template <int W>
__device__ void proj_resolve_wg(const DevProjJob &J) {
    extern __shared__ __attribute__((aligned(16))) char pw_smem[];
    const int nr = (J.n + 63) & ~63, qr = (J.nq + 63) & ~63;
    int *s_claim = reinterpret_cast<int *>(pw_smem);                         // three arrays of nr
    int4 *s_keys = reinterpret_cast<int4 *>(s_claim + 3 * nr);               // per live query
    int *s_meta = reinterpret_cast<int *>(s_keys + qr);                       // #candidates | occupies << 31
    float *s_qang = reinterpret_cast<float *>(s_meta + qr);                   // the query's angle (orientation check), fetched with its keys
    short *s_w1 = reinterpret_cast<short *>(s_qang + qr), *s_w2 = s_w1 + qr;  // a live query's want after the last / the last but one pass
    short *s_pin = s_w2 + qr;                                                 // the answer of its rescan
    unsigned short *s_live = reinterpret_cast<unsigned short *>(s_pin + qr);  // live index -> query
    uint8_t *s_flag = reinterpret_cast<uint8_t *>(s_live + qr);               // 1 = asked for a rescan in the last pass, 2 = pinned by a rescan
    int2 *s_feat = reinterpret_cast<int2 *>(s_flag + qr + 64 - 8);            // float jobs only: the keys' features (qr and the 64 spare bytes keep it 8-byte aligned)
    __shared__ int s_hist[32];
    __shared__ int s_nm, s_drop[3], s_first, s_part[PW_NW], s_cntw[PW_NW], s_guard, s_vote[3];
    __shared__ unsigned short s_wlist[PW_WLIST];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 3) s_vote[tid] = 0;
#ifdef AFV_PROJ_STATS
    const long long st_t0 = wall_clock64();
    long long st_t1 = 0, st_t2 = 0, st_tr = 0;
    int st_conv = 0, st_steps = 0, st_adopt = 0, st_took = 0;
#endif
    for (int i = tid; i < 3 * nr; i += PW_T) s_claim[i] = PW_INF;
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) s_guard = 0;
    // live queries (valid, at least one candidate, best key within TH_HIGH) IN QUERY ORDER
    const int4 *grec = reinterpret_cast<const int4 *>(J.keys);
    int nlive = 0;
    for (int q0 = 0; q0 < J.nq; q0 += PW_T) {
        const int q = q0 + tid;
        int4 ka = make_int4(-1, -1, -1, -1), kb = make_int4(0, 0, 0, 0);
        float qa = 0.0f;
        if (q < J.nq) {
            ka = grec[2 * q];
            kb = grec[2 * q + 1];
            if (J.mode == 1 && J.check_ori) qa = J.qangle[q];
        }
        bool live;
        if constexpr (W == 0) live = q < J.nq && (kb.z & 0xffff) != 0xffff && __int_as_float(ka.x) <= J.th;
        else live = (unsigned)ka.x != PROJ_NO_KEY32 && (float)((unsigned)ka.x >> 16) <= J.th;
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_cntw[wv] = __popcll(m);
        __syncthreads();
        int off = nlive, tot = 0;
#pragma unroll
        for (int w = 0; w < PW_NW; ++w) {
            const int cw = s_cntw[w];
            off += w < wv ? cw : 0;
            tot += cw;
        }
        if (live) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_live[slot] = (unsigned short)q;
            s_keys[slot] = ka;
            if constexpr (W == 0) s_feat[slot] = make_int2(kb.z, kb.w);
            s_qang[slot] = qa;
            s_meta[slot] = (kb.x & 0x7fffffff) | (kb.y ? (int)0x80000000 : 0);
            s_w1[slot] = -1;
            s_w2[slot] = -1;
            s_pin[slot] = -1;
            s_flag[slot] = 0;
        }
        nlive += tot;
        __syncthreads();
    }
#ifdef AFV_PROJ_STATS
    st_t1 = wall_clock64();
#endif
    // ---- the fixed point ----
    int pass = 0;
    const int pass_limit = J.pass_cap > 0 ? J.pass_cap : 3 * nlive + 64;  // every pass finalises at least one more query or a rescan does: a guard
    bool done = nlive == 0;
    while (!done) {
        if (pass >= pass_limit) {
            if (tid == 0) s_guard = 1;
            break;
        }
        int *R = s_claim + (pass % 3) * nr, *Wc = s_claim + ((pass + 1) % 3) * nr, *Z = s_claim + ((pass + 2) % 3) * nr;
        bool changed = false;
        for (int li = tid; li < nlive; li += PW_T) {
            const int w1 = s_w1[li], w2 = s_w2[li];
            const int flag = s_flag[li];
            const int meta = s_meta[li];
            int want = -1;
            bool rescan = false;
            if (flag & 2) {
                want = s_pin[li];  // pinned by its rescan: the query asserts that answer in every pass
            } else {
                int2 ft = make_int2(0, 0);
                if constexpr (W == 0) ft = s_feat[li];
                proj_eval<W>(J, s_keys[li], ft, meta, R, li, want, rescan);
                s_flag[li] = rescan ? 1 : 0;
            }
            if (want >= 0 && meta < 0) atomicMin(&Wc[want], li);  // only a map point with observations blocks later queries
            if (w2 >= 0) Z[w2] = PW_INF;  // what this query (maybe) put into Z two passes ago: Z is empty before it is written again
            s_w2[li] = (short)w1;
            s_w1[li] = (short)want;
            changed = changed || want != w1 || (rescan != ((flag & 1) != 0));
        }
        if (wg_any_changed(changed, pass++, s_vote)) continue;
        // converged: Wc holds the claims of the final wants (so far).  The queries that asked for a rescan, in order
        {
            const bool waits = tid < nlive && (s_flag[tid] & 3) == 1;  // the first 1024 live queries are looked at per cycle
            const unsigned long long bal = __ballot(waits);
            if (lane == 0) s_cntw[wv] = __popcll(bal);
            __syncthreads();
            int off = 0, run = 0;
#pragma unroll
            for (int w = 0; w < PW_NW; ++w) {
                const int cw = s_cntw[w];
                off += w < wv ? cw : 0;
                run += cw;
            }
            if (waits) {
                const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (slot < PW_WLIST) s_wlist[slot] = (unsigned short)tid;
            }
            if (tid == 0) s_first = run;
            __syncthreads();
        }
        const int nwait = s_first;
        int nw = min(nwait, PW_WLIST);
        if (nw == 0 && nlive > PW_T) {  // jobs above 1024 live queries: the waiting ones behind the first 1024, one at a time
            __syncthreads();
            if (tid == 0) s_first = PW_INF;
            __syncthreads();
            int mine = PW_INF;
            for (int li = PW_T + tid; li < nlive; li += PW_T)
                if ((s_flag[li] & 3) == 1) mine = min(mine, li);
            if (mine != PW_INF) atomicMin(&s_first, mine);
            __syncthreads();
            if (s_first != PW_INF) {
                if (tid == 0) s_wlist[0] = (unsigned short)s_first;
                nw = 1;
            }
            __syncthreads();
        }
        if (nw == 0) break;
#ifdef AFV_PROJ_STATS
        ++st_conv;
        const long long st_r0 = wall_clock64();
#endif
        // exact rescans, one waiting query per wavefront (lanes over its window's cells), each against the claims of the queries BEFORE
        // it; adopted in order up to and including the first that takes a feature
        bool took = false;
        for (int g0 = 0; g0 < nw && !took; g0 += PW_NW) {
            const int g = g0 + wv;
            int wr = -1;
            if (g < nw) {
                const int li = s_wlist[g], q0 = s_live[li];
                unsigned long long k0 = P_NO_KEY, k1 = P_NO_KEY;
                if constexpr (W == 0) {
                    const float *qrow = proj_qrow(J, q0);
                    PROJ_WAVE_WINDOW(J, q0, lane, {
                        if (J.occupied && J.occupied[idx]) continue;
                        if (Wc[idx] < li) continue;
                        const unsigned long long key = make_key_f32(proj_l2sqr(qrow, proj_frow(J, idx), J.fdim), epos, idx);
                        if (key < k0) {
                            k1 = k0;
                            k0 = key;
                        } else if (key < k1) {
                            k1 = key;
                        }
                    })
                } else {
                    uint32_t qd[W];
                    {
                        const uint4 *qp = reinterpret_cast<const uint4 *>(J.qdesc + (size_t)q0 * W);
#pragma unroll
                        for (int i = 0; i < W / 4; ++i) {
                            const uint4 t = qp[i];
                            qd[4 * i] = t.x, qd[4 * i + 1] = t.y, qd[4 * i + 2] = t.z, qd[4 * i + 3] = t.w;
                        }
                    }
                    PROJ_WAVE_WINDOW(J, q0, lane, {
                        if (J.occupied && J.occupied[idx]) continue;
                        if (Wc[idx] < li) continue;
                        const unsigned long long key = make_key(proj_hamming<W>(qd, J.fdesc + (size_t)idx * W), c, kpos, idx);
                        if (key < k0) {
                            k1 = k0;
                            k0 = key;
                        } else if (key < k1) {
                            k1 = key;
                        }
                    })
                }
                const unsigned long long b0 = wave_min_u64(k0);
                const unsigned long long b1 = wave_min_u64(k0 == b0 ? k1 : k0);
                if (b0 != P_NO_KEY) {
                    const float best = (float)key_dist_of<W>(b0);
                    const int bidx = key_idx(b0);
                    bool ok = best <= J.th;
                    if (ok && J.mode == 0 && b1 != P_NO_KEY) {
                        const float best2 = (float)key_dist_of<W>(b1), bsz = J.size[bidx], bsz2 = J.size[key_idx(b1)];
                        if ((bsz / bsz2 < J.tol) && (bsz / bsz2 > J.inv_tol) && (bsz2 > 0.0f) && (best > J.ratio * best2)) ok = false;
                    }
                    if (ok) wr = bidx;
                }
            }
            if (lane == 0) s_part[wv] = wr;
            __syncthreads();
            int nadopt = 0;
#pragma unroll
            for (int w = 0; w < PW_NW; ++w) {
                if (g0 + w < nw && !took) {
                    ++nadopt;
                    took = s_part[w] >= 0;
                }
            }
            if (tid < nadopt) {
                const int r = s_wlist[g0 + tid];
                s_flag[r] = 2;
                s_pin[r] = (short)s_part[tid];
            }
#ifdef AFV_PROJ_STATS
            ++st_steps;
            st_adopt += nadopt;
            st_took += took ? 1 : 0;
#endif
            __syncthreads();
        }
#ifdef AFV_PROJ_STATS
        st_tr += wall_clock64() - st_r0;
#endif
        // a taken feature enters the claims with the next pass (the pinned query's want changes from -1); if nothing was taken and every
        // waiting query was looked at, the converged state is the final one
        if (!took && nwait <= PW_WLIST && nlive <= PW_T) break;
    }
    __syncthreads();
#ifdef AFV_PROJ_STATS
    st_t2 = wall_clock64();
#endif
    // ---- F.pts, count ----
    int *A = s_claim;  // assign[feature] = last query that took it
    for (int i = tid; i < nr; i += PW_T) A[i] = -1;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    int cnt = 0;
    const bool ori = J.mode == 1 && J.check_ori;
    for (int li = tid; li < nlive; li += PW_T) {
        const int w = (s_flag[li] & 2) ? (int)s_pin[li] : (int)s_w1[li];
        s_w1[li] = (short)w;
        if (w >= 0) {
            const int q = s_live[li];
            atomicMax(&A[w], q);
            ++cnt;
            if (ori) {  // updateRotationHistogram(rotHist, bestIdx2, LastFrame.mvKeysUn[i], CurrentFrame.mvKeysUn[bestIdx2]) (:1384-1385)
                const int bin = proj_rotation_bin(s_qang[li], J.angle[w]);
                s_flag[li] = (uint8_t)bin;
                atomicAdd(&s_hist[bin], 1);
            }
        }
    }
    cnt = afv_wave_incl_scan(cnt);
    if (lane == 63 && cnt) atomicAdd(&s_nm, cnt);
    __syncthreads();
    if (ori) {
        // filterMatchesWithOrientation (Pt flavour, FeatureMatcher.cc:1601-1613): every accepted entry of a losing bin clears F.pts
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
        const int i1 = s_drop[0], i2 = s_drop[1], i3 = s_drop[2];
        int dropped = 0;
        for (int li = tid; li < nlive; li += PW_T) {
            const int w = s_w1[li];
            if (w >= 0) {
                const int b = s_flag[li];
                if (b != i1 && b != i2 && b != i3) {
                    A[w] = -1;
                    ++dropped;
                }
            }
        }
        if (dropped) atomicSub(&s_nm, dropped);
        __syncthreads();
    }
    for (int i = tid; i < J.n; i += PW_T) J.assign[i] = A[i];
    if (tid == 0) *J.nmatches = s_guard ? PW_GUARD : s_nm;
#ifdef AFV_PROJ_STATS
    if (tid == 0)
        printf("proj_resolve_wg: mode %d nq %d nlive %d passes %d convergences %d rescan_steps %d adopted %d took %d | x10ns: setup %lld fixedpoint %lld (rescans %lld) tail %lld\n",
               J.mode, J.nq, nlive, pass, st_conv, st_steps, st_adopt, st_took, st_t1 - st_t0, st_t2 - st_t1, st_tr, wall_clock64() - st_t2);
#endif
}

__global__ __launch_bounds__(PW_T) void k_proj_resolve_wg(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.fdim) proj_resolve_wg<0>(J);
    else if (J.words == 8) proj_resolve_wg<8>(J);
    else proj_resolve_wg<16>(J);
}
__global__ __launch_bounds__(PW_T) void k_proj_resolve_wg1(const DevProjJob J) {
    if (J.fdim) proj_resolve_wg<0>(J);
    else if (J.words == 8) proj_resolve_wg<8>(J);
    else proj_resolve_wg<16>(J);
}

// ---------------- Fuse / SearchBySim3: independent queries, one wave each; first minimum in visiting order (:905) ----------------
template <int W>
__device__ void fuse_query(const DevProjJob &J, int q, int lane, int2 *s_list) {
    unsigned long long k0 = P_NO_KEY;
    if (!J.qvalid || J.qvalid[q]) {
        uint32_t qd[W == 0 ? 1 : W];
        const float *qrow = nullptr;
        if constexpr (W == 0) {
            qrow = proj_qrow(J, q);
            (void)qd;
        } else {
            (void)qrow;
#pragma unroll
            for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
        }
        const float u = J.qu[q], v = J.qv[q];
        const float qur = J.u_right ? J.q_ur[q] : 0.0f;
        PROJ_WAVE_WINDOW_DENSE(J, q, lane, s_list, {
            if (J.inf) {  // reprojection gate of Fuse (:876-900); absent in Fuse(Sim3) / SearchBySim3
                const float ex = u - fx_;
                const float ey = v - fy_;
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
            unsigned long long key;
            if constexpr (W == 0) key = make_key_f32(proj_l2sqr(qrow, proj_frow(J, idx), J.fdim), epos, idx);
            else key = make_key(proj_hamming<W>(qd, J.fdesc + (size_t)idx * W), c, kpos, idx);
            k0 = key < k0 ? key : k0;
        })
    }
    const unsigned long long g0 = wave_min_u64(k0);
    if (lane == 0) J.assign[q] = (g0 != P_NO_KEY && (float)key_dist_of<W>(g0) <= J.th) ? key_idx(g0) : -1;
}

__global__ __launch_bounds__(PT) void k_match_fuse(const DevProjJob *__restrict__ jobs) {
    __shared__ int2 s_list[PT / 64][PW_LIST];
    const DevProjJob J = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63, q = blockIdx.x * (PT / 64) + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (q >= J.nq) return;
    if (J.fdim) fuse_query<0>(J, q, lane, s_list[threadIdx.x >> 6]);
    else if (J.words == 8) fuse_query<8>(J, q, lane, s_list[threadIdx.x >> 6]);
    else fuse_query<16>(J, q, lane, s_list[threadIdx.x >> 6]);
}
__global__ __launch_bounds__(PT) void k_match_fuse1(const DevProjJob J) {
    __shared__ int2 s_list[PT / 64][PW_LIST];
    const int lane = threadIdx.x & 63, q = blockIdx.x * (PT / 64) + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (q >= J.nq) return;
    if (J.fdim) fuse_query<0>(J, q, lane, s_list[threadIdx.x >> 6]);
    else if (J.words == 8) fuse_query<8>(J, q, lane, s_list[threadIdx.x >> 6]);
    else fuse_query<16>(J, q, lane, s_list[threadIdx.x >> 6]);
}

// ---------------- SearchForInitialization (FeatureMatcher.cc:399-557, active part :480-556) ----------------
// queries = level-0 features of F1 searched in a fixed window around vbPrevMatched in F2.  Inherently ordered: a
// candidate is skipped when an earlier query already matched it at a distance <= the current one (:513), and a later
// query steals the feature (:531-535).  k_proj_topk<IK> ranks every query's window in parallel; one wave then walks the
// queries in order.  The gate only ever removes candidates, so best / second are the first two ungated keys; when the
// IK keys run out before two are found (and the window holds more) the window is rescanned exactly.
template <int W>
__device__ void init_resolve(const DevProjJob &J, unsigned char *s_tables /* 6 x P_MAX_FEATS bytes, 16-byte aligned */, int *s_hist) {
    // binary: holder as int, its distance as u16 (48 KB); float distances need 32 bits, so the holder shrinks to u16 (nq <= 65535: 0xffff = none).
    // (One buffer in the kernel, typed here: static arrays inside this template would be laid out once PER INSTANTIATION - 3 x 48 KB.)
    using m21_t = std::conditional_t<W == 0, unsigned short, int>;
    using md_t = std::conditional_t<W == 0, float, unsigned short>;
    constexpr m21_t M21_NONE = (m21_t)-1;
    md_t *s_mdist = reinterpret_cast<md_t *>(s_tables + (W == 0 ? 0 : P_MAX_FEATS * 4));
    m21_t *s_m21 = reinterpret_cast<m21_t *>(s_tables + (W == 0 ? P_MAX_FEATS * 4 : 0));
    const int lane = threadIdx.x;
    for (int i = lane; i < J.n; i += 64) {
        s_m21[i] = M21_NONE;
        if constexpr (W == 0) s_mdist[i] = 3.402823466e+38f;  // vMatchedDistance starts at highestPossibleDistance = numeric_limits<float>::max() (:485)
        else s_mdist[i] = 0xffff;
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
            bool okk;  // gate (:513)
            if constexpr (W == 0) okk = key != P_NO_KEY && !(s_mdist[ki] <= key_dist_of<0>(key));
            else okk = key != P_NO_KEY && !((int)s_mdist[ki] <= key_dist(key));
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
                unsigned long long k0 = P_NO_KEY, k1 = P_NO_KEY;
                if constexpr (W == 0) {
                    const float *qrow = proj_qrow(J, q);
                    PROJ_WAVE_WINDOW(J, q, lane, {
                        const float d = proj_l2sqr(qrow, proj_frow(J, idx), J.fdim);
                        if (s_mdist[idx] <= d) continue;
                        const unsigned long long kk = make_key_f32(d, epos, idx);
                        if (kk < k0) {
                            k1 = k0;
                            k0 = kk;
                        } else if (kk < k1) {
                            k1 = kk;
                        }
                    })
                } else {
                    uint32_t qd[W];
#pragma unroll
                    for (int i = 0; i < W; ++i) qd[i] = J.qdesc[(size_t)q * W + i];
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
                }
                g0 = wave_min_u64(k0);
                g1 = wave_min_u64(k0 == g0 ? k1 : k0);
            }
            if (g0 == P_NO_KEY) continue;
            const float best = (float)key_dist_of<W>(g0);
            const float best2 = g1 == P_NO_KEY ? 3.402823466e+38f : (float)key_dist_of<W>(g1);
            const int bidx = key_idx(g0);
            if (best <= J.th && best < best2 * J.ratio) {  // :527-529
                if (lane == 0) {
                    const m21_t prev = s_m21[bidx];
                    if (prev != M21_NONE) J.assign[prev] = -1;  // stolen (:531-535)
                    J.assign[q] = bidx;
                    s_m21[bidx] = (m21_t)q;
                    s_mdist[bidx] = (md_t)key_dist_of<W>(g0);
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
    cnt = wave_sum_i32(cnt);
    if (lane == 0) *J.nmatches = cnt;
}

__global__ __launch_bounds__(64) void k_init_resolve(const DevProjJob *__restrict__ jobs) {
    __shared__ __attribute__((aligned(16))) unsigned char s_tables[P_MAX_FEATS * 6];
    __shared__ int s_hist[32];
    const DevProjJob J = jobs[blockIdx.x];
    if (J.fdim) init_resolve<0>(J, s_tables, s_hist);
    else if (J.words == 8) init_resolve<8>(J, s_tables, s_hist);
    else init_resolve<16>(J, s_tables, s_hist);
}

// ---------------- SearchForInitialization, workgroup form (round 5) ----------------
// The same fixed point for the ordered loop of :480-556.  What query i may take depends on the queries before it through
// vMatchedDistance: candidate f at distance d is skipped when an EARLIER query holds f at a distance <= d (:513); an accepted match
// steals f from its previous holder (:531-535), so the holders of a feature, in query order, have strictly decreasing distances and
// the last one keeps it.  want_i = f({(want_j, d_j) : j < i}) again, so any assignment satisfying all equations is the sequential
// outcome.  Per feature three rotating LDS words describe the queries that currently want it:
//   E = min(live index << 16 | distance)   the earliest wanter          M = min(distance << 16 | live index)   the closest wanter
//   C = how many
// gate(i, f, d) = "some wanter j < i of f has d_j <= d" follows exactly from them in all but one constellation (an earlier wanter
// farther than d, the closest wanter not earlier than i, three or more wanters), which falls back to a scan of the wants of the
// queries before i (double-buffered by pass parity, so the scan reads the previous pass consistently).
#define IW_T 1024
#define IW_NW (IW_T / 64)
#define IW_INF 0x7fffffff
#define IW_WLIST 128
static inline size_t init_wg_lds_bytes(int n, int nq) {
    const size_t nr = ((size_t)n + 63) & ~(size_t)63, qr = ((size_t)nq + 63) & ~(size_t)63;
    return 9 * nr * 4 + qr * 32 /*keys*/ + qr * 4 /*ncand*/ + 4 * qr * 2 /*want, dist x 2 buffers*/ + 2 * qr * 2 /*pin*/ + qr * 2 /*live*/ + qr /*flag*/ + 64;
}

struct InitState {
    const int *E, *M, *C;          // the arrays of the previous pass
    const short *want, *dist;      // the wants / distances of the previous pass, by live index
};
__device__ __forceinline__ bool init_gate(const InitState &S, int li, int f, int d) {
    const int e = S.E[f];
    if (e == IW_INF) return false;
    if ((e >> 16) >= li) return false;                 // nobody before li wants f
    if ((e & 0xffff) <= d) return true;                // the earliest wanter already holds it at <= d
    const int m = S.M[f];
    if ((m & 0xffff) < li) return (m >> 16) <= d;      // the closest wanter of all is before li
    if (S.C[f] == 2) return false;                     // two wanters: only the earliest is before li, and it is farther than d
    int mind = IW_INF;                                 // exact: the closest wanter among the queries before li
    for (int j = 0; j < li; ++j)
        if (S.want[j] == f) mind = min(mind, (int)S.dist[j]);
    return mind <= d;
}

template <int W>
__device__ void init_resolve_wg(const DevProjJob &J) {
    extern __shared__ __attribute__((aligned(16))) char iw_smem[];
    const int nr = (J.n + 63) & ~63, qr = (J.nq + 63) & ~63;
    int *s_E = reinterpret_cast<int *>(iw_smem), *s_M = s_E + 3 * nr, *s_C = s_M + 3 * nr;
    int4 *s_keys = reinterpret_cast<int4 *>(s_C + 3 * nr);  // two int4 per live query
    int *s_ncand = reinterpret_cast<int *>(s_keys + 2 * qr);
    short *s_want = reinterpret_cast<short *>(s_ncand + qr);  // [2][qr] by pass parity
    short *s_dist = s_want + 2 * qr;                            // [2][qr]
    short *s_pin = s_dist + 2 * qr, *s_pind = s_pin + qr;
    unsigned short *s_live = reinterpret_cast<unsigned short *>(s_pind + qr);
    uint8_t *s_flag = reinterpret_cast<uint8_t *>(s_live + qr);
    __shared__ int s_hist[32];
    __shared__ int s_nm, s_drop[3], s_first, s_part[IW_NW], s_partd[IW_NW], s_cntw[IW_NW], s_guard, s_vote[3];
    __shared__ unsigned short s_wlist[IW_WLIST];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 3) s_vote[tid] = 0;
    for (int i = tid; i < 9 * nr; i += IW_T) s_E[i] = (i >= 6 * nr) ? 0 : IW_INF;
    for (int q = tid; q < J.nq; q += IW_T) J.assign[q] = -1;
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) s_guard = 0;
    const int4 *grec = reinterpret_cast<const int4 *>(J.keys);
    int nlive = 0;
    for (int q0 = 0; q0 < J.nq; q0 += IW_T) {
        const int q = q0 + tid;
        int4 ka = make_int4(-1, -1, -1, -1), kb = ka;
        int nc = 0;
        if (q < J.nq && (!J.qvalid || J.qvalid[q])) {
            ka = grec[2 * q];
            kb = grec[2 * q + 1];
            nc = J.ncand[q];
        }
        const bool live = (unsigned)ka.x != PROJ_NO_KEY32 && (float)((unsigned)ka.x >> 16) <= J.th;
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_cntw[wv] = __popcll(m);
        __syncthreads();
        int off = nlive, tot = 0;
#pragma unroll
        for (int w = 0; w < IW_NW; ++w) {
            const int cw = s_cntw[w];
            off += w < wv ? cw : 0;
            tot += cw;
        }
        if (live) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_live[slot] = (unsigned short)q;
            s_keys[2 * slot] = ka;
            s_keys[2 * slot + 1] = kb;
            s_ncand[slot] = nc;
            s_want[slot] = -1;
            s_want[qr + slot] = -1;
            s_dist[slot] = 0;
            s_dist[qr + slot] = 0;
            s_pin[slot] = -1;
            s_pind[slot] = 0;
            s_flag[slot] = 0;
        }
        nlive += tot;
        __syncthreads();
    }
    int pass = 0;
    const int pass_limit = J.pass_cap > 0 ? J.pass_cap : 3 * nlive + 64;
    bool done = nlive == 0;
    while (!done) {
        if (pass >= pass_limit) {
            if (tid == 0) s_guard = 1;
            break;
        }
        const int ir = pass % 3, iw = (pass + 1) % 3, iz = (pass + 2) % 3;
        const int cur = pass & 1, nxt = cur ^ 1;
        InitState S{s_E + ir * nr, s_M + ir * nr, s_C + ir * nr, s_want + cur * qr, s_dist + cur * qr};
        int *WE = s_E + iw * nr, *WM = s_M + iw * nr, *WC = s_C + iw * nr;
        bool changed = false;
        for (int li = tid; li < nlive; li += IW_T) {
            const int w1 = S.want[li], w2 = s_want[nxt * qr + li];
            const int flag = s_flag[li];
            int want = -1, dw = 0;
            bool rescan = false;
            if (flag & 2) {
                want = s_pin[li];
                dw = s_pind[li];
            } else {
                const int4 ka = s_keys[2 * li], kb = s_keys[2 * li + 1];
                const unsigned keys[IK] = {(unsigned)ka.x, (unsigned)ka.y, (unsigned)ka.z, (unsigned)ka.w,
                                           (unsigned)kb.x, (unsigned)kb.y, (unsigned)kb.z, (unsigned)kb.w};
                const bool complete = s_ncand[li] <= IK;
                int b0 = -1, d0 = 0, d1 = -1;
#pragma unroll
                for (int s = 0; s < IK; ++s) {
                    if (keys[s] != PROJ_NO_KEY32 && d1 < 0) {
                        const int f = (int)(keys[s] & 0xffffu), d = (int)(keys[s] >> 16);
                        if (!init_gate(S, li, f, d)) {
                            if (b0 < 0) {
                                b0 = f;
                                d0 = d;
                            } else {
                                d1 = d;
                            }
                        }
                    }
                }
                const float d_last = (float)(keys[IK - 1] >> 16);
                if (b0 < 0) {
                    rescan = !complete && d_last <= J.th;
                } else if (!((float)d0 <= J.th)) {
                    // final: everything else is farther
                } else if (d1 >= 0) {
                    if ((float)d0 < (float)d1 * J.ratio) want = b0;  // :527-529
                } else if (complete) {
                    if ((float)d0 < 3.402823466e+38f * J.ratio) want = b0;
                } else if (J.ratio >= 0.0f && (float)d0 < d_last * J.ratio) {
                    want = b0;  // the second candidate, wherever it is, is at least as far as the last key
                } else {
                    rescan = true;
                }
                dw = d0;
                s_flag[li] = rescan ? 1 : 0;
            }
            if (want >= 0) {
                atomicMin(&WE[want], (li << 16) | dw);
                atomicMin(&WM[want], (dw << 16) | li);
                atomicAdd(&WC[want], 1);
            }
            if (w2 >= 0) {
                s_E[iz * nr + w2] = IW_INF;
                s_M[iz * nr + w2] = IW_INF;
                s_C[iz * nr + w2] = 0;
            }
            const int dprev = S.dist[li];
            s_want[nxt * qr + li] = (short)want;
            s_dist[nxt * qr + li] = (short)dw;
            changed = changed || want != w1 || (want >= 0 && dw != dprev) || (rescan != ((flag & 1) != 0));
        }
        if (wg_any_changed(changed, pass++, s_vote)) continue;
        {
            const bool waits = tid < nlive && (s_flag[tid] & 3) == 1;
            const unsigned long long bal = __ballot(waits);
            if (lane == 0) s_cntw[wv] = __popcll(bal);
            __syncthreads();
            int off = 0, run = 0;
#pragma unroll
            for (int w = 0; w < IW_NW; ++w) {
                const int cw = s_cntw[w];
                off += w < wv ? cw : 0;
                run += cw;
            }
            if (waits) {
                const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (slot < IW_WLIST) s_wlist[slot] = (unsigned short)tid;
            }
            if (tid == 0) s_first = run;
            __syncthreads();
        }
        const int nwait = s_first;
        int nw = min(nwait, IW_WLIST);
        if (nw == 0 && nlive > IW_T) {
            __syncthreads();
            if (tid == 0) s_first = IW_INF;
            __syncthreads();
            int mine = IW_INF;
            for (int li = IW_T + tid; li < nlive; li += IW_T)
                if ((s_flag[li] & 3) == 1) mine = min(mine, li);
            if (mine != IW_INF) atomicMin(&s_first, mine);
            __syncthreads();
            if (s_first != IW_INF) {
                if (tid == 0) s_wlist[0] = (unsigned short)s_first;
                nw = 1;
            }
            __syncthreads();
        }
        if (nw == 0) break;
        // the state the rescans read: the arrays written by the pass that just converged, the wants it produced
        InitState F{s_E + iw * nr, s_M + iw * nr, s_C + iw * nr, s_want + nxt * qr, s_dist + nxt * qr};
        bool took = false;
        for (int g0 = 0; g0 < nw && !took; g0 += IW_NW) {
            const int g = g0 + wv;
            int wr = -1, wd = 0;
            if (g < nw) {
                const int li = s_wlist[g], q0 = s_live[li];
                uint32_t qd[W];
                {
                    const uint4 *qp = reinterpret_cast<const uint4 *>(J.qdesc + (size_t)q0 * W);
#pragma unroll
                    for (int i = 0; i < W / 4; ++i) {
                        const uint4 t = qp[i];
                        qd[4 * i] = t.x, qd[4 * i + 1] = t.y, qd[4 * i + 2] = t.z, qd[4 * i + 3] = t.w;
                    }
                }
                unsigned long long k0 = P_NO_KEY, k1 = P_NO_KEY;
                PROJ_WAVE_WINDOW(J, q0, lane, {
                    const int d = proj_hamming<W>(qd, J.fdesc + (size_t)idx * W);
                    if (init_gate(F, li, idx, d)) continue;
                    const unsigned long long kk = make_key(d, c, kpos, idx);
                    if (kk < k0) {
                        k1 = k0;
                        k0 = kk;
                    } else if (kk < k1) {
                        k1 = kk;
                    }
                })
                const unsigned long long b0 = wave_min_u64(k0);
                const unsigned long long b1 = wave_min_u64(k0 == b0 ? k1 : k0);
                if (b0 != P_NO_KEY) {
                    const float best = (float)key_dist(b0);
                    const float best2 = b1 == P_NO_KEY ? 3.402823466e+38f : (float)key_dist(b1);
                    if (best <= J.th && best < best2 * J.ratio) {
                        wr = key_idx(b0);
                        wd = key_dist(b0);
                    }
                }
            }
            if (lane == 0) {
                s_part[wv] = wr;
                s_partd[wv] = wd;
            }
            __syncthreads();
            int nadopt = 0;
#pragma unroll
            for (int w = 0; w < IW_NW; ++w) {
                if (g0 + w < nw && !took) {
                    ++nadopt;
                    took = s_part[w] >= 0;
                }
            }
            if (tid < nadopt) {
                const int r = s_wlist[g0 + tid];
                s_flag[r] = 2;
                s_pin[r] = (short)s_part[tid];
                s_pind[r] = (short)s_partd[tid];
            }
            __syncthreads();
        }
        if (!took && nwait <= IW_WLIST && nlive <= IW_T) break;
    }
    __syncthreads();
    // ---- matches: a query keeps its feature iff it is the closest (= last) of the feature's holders ----
    const int fin = pass % 3, fb = pass & 1;  // arrays / wants written by the last pass
    const int *FM = s_M + fin * nr;
    const short *fw = s_want + fb * qr;
    if (tid == 0) s_nm = 0;
    __syncthreads();
    int cnt = 0;
    for (int li = tid; li < nlive; li += IW_T) {
        const int w = (s_flag[li] & 2) ? (int)s_pin[li] : (int)fw[li];
        int keep = -1, bin = 31;
        if (w >= 0) {
            const int q = s_live[li];
            // a pinned answer of the last adoption round that took nothing is -1; one that took something was followed by a pass
            if ((FM[w] & 0xffff) == li) keep = w;
            if (J.check_ori) {
                bin = proj_rotation_bin(J.qangle[q], J.angle[w]);  // F1 keypoint first (:543); stolen matches stay in the histogram
                atomicAdd(&s_hist[bin], 1);
            }
        }
        s_pin[li] = (short)keep;
        s_flag[li] = (uint8_t)bin;
    }
    __syncthreads();
    if (J.check_ori && tid == 0) {
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
    for (int li = tid; li < nlive; li += IW_T) {
        int keep = s_pin[li];
        if (keep >= 0 && J.check_ori) {
            const int b = s_flag[li];
            if (b != s_drop[0] && b != s_drop[1] && b != s_drop[2]) keep = -1;  // filterMatchesWithOrientation, int flavour (:1615-1629)
        }
        if (keep >= 0) {
            J.assign[s_live[li]] = keep;
            ++cnt;
        }
    }
    cnt = afv_wave_incl_scan(cnt);
    if (lane == 63 && cnt) atomicAdd(&s_nm, cnt);
    __syncthreads();
    if (tid == 0) *J.nmatches = s_guard ? PW_GUARD : s_nm;
}

__global__ __launch_bounds__(IW_T) void k_init_resolve_wg(const DevProjJob *__restrict__ jobs) {
    const DevProjJob J = jobs[blockIdx.x];
    if (J.words == 8) init_resolve_wg<8>(J);
    else init_resolve_wg<16>(J);
}
__global__ __launch_bounds__(IW_T) void k_init_resolve_wg1(const DevProjJob J) {
    if (J.words == 8) init_resolve_wg<8>(J);
    else init_resolve_wg<16>(J);
}

// ---------------- ranking + ordered phase in ONE launch (a single job whose record is the kernel argument) ----------------
// A search against a resident frame is two dependent launches of a few microseconds each; the second one's dispatch and prologue are as long
// as its work.  Here the ranking runs on 1024-thread workgroups (16 queries each, a wavefront per query as in k_proj_topk), and the
// workgroup that finishes LAST (one ticket per launch; cdna_hip_programming.md Guideline 16, the hand-off of k_match_topk_mfma: every
// wave's stores drained by the barrier -> ONE lane: agent-scope release, ticket; the last arriver: ONE agent-scope acquire -> barrier
// -> plain loads) goes on as the fixed-point workgroup.  The ticket is zero at rest: the last arriver re-arms it.
template <int KIND>  // 0 = projection (PK keys, REC 1), 1 = initialization (IK keys, REC 3)
__global__ __launch_bounds__(PW_T) void k_proj_search1(const DevProjJob J, int *__restrict__ ticket) {
    __shared__ int2 s_list[PW_NW][PW_LIST];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int q = (int)blockIdx.x * PW_NW + wv;
    if (q < J.nq) {  // wave-uniform
        if (KIND == 0) {  // (binary rows only: float jobs take the two-launch form - their instantiations would push this kernel into spilling)
            if (J.words == 8) topk_query<8, PK, 1>(J, q, lane, s_list[wv]);
            else topk_query<16, PK, 1>(J, q, lane, s_list[wv]);
        } else {
            if (J.words == 8) topk_query<8, IK, 3>(J, q, lane, s_list[wv]);
            else topk_query<16, IK, 3>(J, q, lane, s_list[wv]);
        }
    }
    __shared__ int s_last;
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // restated where the compiler cannot drop it (ROCm 7.2)
        const int old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (int)gridDim.x - 1;
        if (last) {
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    if (KIND == 0) {
        if (J.words == 8) proj_resolve_wg<8>(J);
        else proj_resolve_wg<16>(J);
    } else {
        if (J.words == 8) init_resolve_wg<8>(J);
        else init_resolve_wg<16>(J);
    }
}

// the workgroup engines need more LDS than the 64 KB a kernel gets by default: raised once per device (afv_create), checked
extern "C" int afv_project_prepare(void) {
    // the one-launch search keeps 32 KB of static LDS for the ranking phase's candidate lists (16 wavefronts x PW_LIST x 8 bytes): its
    // dynamic share is what is left of the 160 KB of a CU, and that is the budget every engine is held to
    const int want = 150 * 1024 - (int)sizeof(int2) * PW_NW * PW_LIST;
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve_wg), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_init_resolve_wg), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_resolve_wg1), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_init_resolve_wg1), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_search1<0>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_proj_search1<1>), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    if (!ok) (void)hipGetLastError();
    // dynamic bytes a job may ask for (the kernels' static arrays take about 1 KB more); without the raised limit: what every kernel gets
    return ok ? want - 2048 : 62 * 1024 - (int)sizeof(int2) * PW_NW * PW_LIST;
}
extern "C" size_t afv_project_wg_lds(int kind_init, int n, int nq, int float_rows) {
    return kind_init ? init_wg_lds_bytes(n, nq) : proj_wg_lds_bytes(n, nq, float_rows != 0);
}

// `one` != nullptr: a single job whose record travels as the kernel argument (the fixed-point engines and Fuse); else `jobs` is the
// device array of njobs records
extern "C" void afv_launch_match_init(const DevProjJob *jobs, int njobs, int max_nq, size_t wg_lds, const DevProjJob *one, int *ticket,
                                      hipStream_t stream) {
    const dim3 tg((max_nq + PT / 64 - 1) / (PT / 64), njobs);
    if (wg_lds && one && ticket) {  // ranking + ordered phase in one launch
        hipLaunchKernelGGL(k_proj_search1<1>, dim3(std::max((max_nq + PW_NW - 1) / PW_NW, 1)), dim3(PW_T), wg_lds, stream, *one, ticket);
        return;
    }
    if (wg_lds && one) {
        if (max_nq > 0) hipLaunchKernelGGL((k_proj_topk1<IK, 3>), tg, dim3(PT), 0, stream, *one);
        hipLaunchKernelGGL(k_init_resolve_wg1, dim3(1), dim3(IW_T), wg_lds, stream, *one);
    } else if (wg_lds) {
        if (max_nq > 0) hipLaunchKernelGGL((k_proj_topk<IK, 3>), tg, dim3(PT), 0, stream, jobs);
        hipLaunchKernelGGL(k_init_resolve_wg, dim3(njobs), dim3(IW_T), wg_lds, stream, jobs);
    } else {
        if (max_nq > 0) hipLaunchKernelGGL((k_proj_topk<IK, 2>), tg, dim3(PT), 0, stream, jobs);
        hipLaunchKernelGGL(k_init_resolve, dim3(njobs), dim3(64), 0, stream, jobs);
    }
}

extern "C" void afv_launch_match_fuse(const DevProjJob *jobs, int njobs, int max_nq, const DevProjJob *one, hipStream_t stream) {
    if (max_nq <= 0) return;
    const dim3 tg((max_nq + PT / 64 - 1) / (PT / 64), njobs);
    if (one) hipLaunchKernelGGL(k_match_fuse1, tg, dim3(PT), 0, stream, *one);
    else hipLaunchKernelGGL(k_match_fuse, tg, dim3(PT), 0, stream, jobs);
}

// wg_lds != 0: the fixed-point engine with that much dynamic LDS (the largest job's tables); 0: the ordered walk
extern "C" void afv_launch_match_projection(const DevProjJob *jobs, int njobs, int max_nq, size_t wg_lds, const DevProjJob *one, int *ticket,
                                            hipStream_t stream) {
    const dim3 tg((max_nq + PT / 64 - 1) / (PT / 64), njobs);
    if (wg_lds && one && ticket) {  // ranking + ordered phase in one launch
        hipLaunchKernelGGL(k_proj_search1<0>, dim3(std::max((max_nq + PW_NW - 1) / PW_NW, 1)), dim3(PW_T), wg_lds, stream, *one, ticket);
        return;
    }
    if (wg_lds && one) {
        if (max_nq > 0) hipLaunchKernelGGL((k_proj_topk1<PK, 1>), tg, dim3(PT), 0, stream, *one);
        hipLaunchKernelGGL(k_proj_resolve_wg1, dim3(1), dim3(PW_T), wg_lds, stream, *one);
        return;
    }
    if (wg_lds) {
        if (max_nq > 0) hipLaunchKernelGGL((k_proj_topk<PK, 1>), tg, dim3(PT), 0, stream, jobs);
        hipLaunchKernelGGL(k_proj_resolve_wg, dim3(njobs), dim3(PW_T), wg_lds, stream, jobs);
        return;
    }
    if (max_nq > 0) hipLaunchKernelGGL((k_proj_topk<PK, 0>), tg, dim3(PT), 0, stream, jobs);
    // stage the query records in LDS when the largest job fits (64 B per query next to the 33.9 KB of tables; 160 KB per CU)
    int stage_cap = max_nq;
    if ((size_t)PR_LDS_FIXED + (size_t)stage_cap * PR_REC_BYTES > 128 * 1024) stage_cap = 0;
    hipLaunchKernelGGL(k_proj_resolve, dim3(njobs), dim3(PT), PR_LDS_FIXED + (size_t)stage_cap * PR_REC_BYTES, stream, jobs, stage_cap);
}
