// k_match.hip — M1..M6, M8: Hamming / L2 descriptor matchers.
//
// Replaces FeatureMatcher::SearchByBoW(KF,KF) (FeatureMatcher.cc:561-660), SearchByBoW(KF,Frame) (:186-283),
// SearchForTriangulation (:662-790) with CheckDistEpipolarLine (:165-182), the rotation-histogram filter
// (:1579-1668) and DescriptorDistance_orb32 / _sift128 (Feature_orb32.cpp:67-84, Feature_sift128.cpp:132-134).
//
// SearchByBoW is greedy: row idx1 may only take a column that no earlier row has taken (vbMatched2 / the
// vpMapPointMatches test).  Distance of a 256-bit pair = 8 xor + 8 v_bcnt-accumulate (the popcount north_star asks for).
// Three kernels share that arithmetic and the reference's float predicates (strict / non-strict threshold, nnratio
// product, rotation histogram):
//   * k_match_topk + k_match_resolve — brute force over whole descriptor sets (pipeline path and plain host jobs):
//     parallel top-4 per row, then the ordered greedy walk in speculative 64-row rounds;
//   * k_match_bow_seg (+ k_match_bow_finish) — BoW-guided jobs, one wavefront per shared vocabulary node;
//   * k_match_bow — generic ordered workgroup-per-job kernel (jobs with validity masks / KF-Frame mode and one node).
// SearchForTriangulation (k_match_tri) has no cross-row dependency: every wavefront takes its own rows.
#include <algorithm>

#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch
#include "afv_jobs.h"


#include <type_traits>

#define MT 256
#define NO_KEY 0x7fffffff
#define MAX_SIDE 8192  // features per side a job may hold (LDS bitset + bin table)

// LDS hand-off between lanes of ONE wavefront: DS operations of a wave execute in order, only the compiler has to be
// kept from moving reads above writes
#define WAVE_LDS_SYNC()                                        \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); \
    } while (0)

// The same hand-off when nothing but LDS traffic of THIS wavefront has to be ordered: a workgroup-scope release also drains the
// vector-memory counter, i.e. it waits for every global store / load the wavefront still has in flight (about 2 us per round of the
// ordered resolve walk, measured) — wavefront scope keeps the compiler from reordering and costs nothing at run time.
#define WAVE_LDS_ONLY_SYNC()                                   \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)



__device__ __forceinline__ int rotation_bin(float a1, float a2) {
    // FeatureMatcher.cc:1587-1599, rotFactor = 1/30 (:1579-1585)
    const float rot_factor = 1.0f / 30.0f;
    float rot = a1 - a2;
    if (rot < 0.0f) rot += 360.0f;
    int bin = (int)roundf(rot * rot_factor);
    if (bin == 30) bin = 0;
    return bin;
}

// merge two (best key, second distance) summaries.  key = dist << 16 | position
__device__ __forceinline__ void merge_best(int &k, int &s, int k2, int s2) {
    if (k2 < k) {
        s = min(s2, k >> 16);
        k = k2;
    } else {
        s = min(s, k2 >> 16);
    }
}

// wave-wide merge_best over the 64 lanes on DPP (row shifts, then the row totals carried with row_bcast15 / row_bcast31: no LDS
// crossbar); lanes without a source merge the identity.  Every lane receives the result.
__device__ __forceinline__ void wave_merge_best(int &k, int &s) {
    const int IDK = 0x7fffffff, IDS = 0x7fffffff >> 16;
#define AFV_MB_STEP(ctrl, rmask)                                                               \
    {                                                                                          \
        const int k2 = __builtin_amdgcn_update_dpp(IDK, k, ctrl, rmask, 0xf, false);           \
        const int s2 = __builtin_amdgcn_update_dpp(IDS, s, ctrl, rmask, 0xf, false);           \
        merge_best(k, s, k2, s2);                                                              \
    }
    AFV_MB_STEP(0x111, 0xf)  // row_shr:1
    AFV_MB_STEP(0x112, 0xf)  // row_shr:2
    AFV_MB_STEP(0x114, 0xf)  // row_shr:4
    AFV_MB_STEP(0x118, 0xf)  // row_shr:8
    AFV_MB_STEP(0x142, 0xa)  // row_bcast:15 into rows 1, 3
    AFV_MB_STEP(0x143, 0xc)  // row_bcast:31 into rows 2, 3
#undef AFV_MB_STEP
    k = __builtin_amdgcn_readlane(k, 63);
    s = __builtin_amdgcn_readlane(s, 63);
}

__device__ __forceinline__ int med3_i32(int a, int b, int c) {  // v_med3_i32 (no clang builtin for the integer form)
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int W>
__device__ __forceinline__ int hamming_words(const uint32_t *a_regs, const uint32_t *b) {
    int d = 0;
#pragma unroll
    for (int i = 0; i < W; ++i) d += __popc(a_regs[i] ^ b[i]);
    return d;
}

// ---------------- float descriptors inside the BoW-guided matchers (round 6; W == 0 in the templates below) ----------------
// FeatureMatcher::DescriptorDistance dispatches on DescriptorType (FeatureMatcher.cc:1508-1531, called from SearchByBoW :236 / :616 and
// SearchForTriangulation :734); for SIFT128 / SURF64 / KAZE64 / R2D2 it is cv::norm(a, b, NORM_L2SQR) narrowed to float
// (Feature_sift128.cpp:132-134, Types.h:127).  A non-negative float's bits order like the number, so (best key, second distance) =
// (distance bits << 32 | position, distance bits) merge exactly like the integer pair of the binary path.
__device__ __forceinline__ float l2sqr(const float *a, const float *b, int n) {
    // normL2Sqr<float,double>: float differences, double squares, 4-way partial sums
    double s = 0;
    int i = 0;
    for (; i <= n - 4; i += 4) {
        const double v0 = (double)(a[i] - b[i]), v1 = (double)(a[i + 1] - b[i + 1]), v2 = (double)(a[i + 2] - b[i + 2]),
                     v3 = (double)(a[i + 3] - b[i + 3]);
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    }
    for (; i < n; ++i) {
        const double v = (double)(a[i] - b[i]);
        s += v * v;
    }
    return (float)s;
}
__device__ __forceinline__ void merge_best(unsigned long long &k, unsigned &s, unsigned long long k2, unsigned s2) {
    if (k2 < k) {
        s = min(s2, (unsigned)(k >> 32));
        k = k2;
    } else {
        s = min(s, (unsigned)(k2 >> 32));
    }
}
template <int W>
struct BowKey {  // binary: distance << 16 | position in an int, second-best distance in an int
    using key_t = int;
    using sec_t = int;
    static constexpr key_t none = NO_KEY;
    static constexpr sec_t none2 = NO_KEY >> 16;
    static __device__ __forceinline__ key_t make(int d, int b) { return (d << 16) | b; }
    static __device__ __forceinline__ float dist(key_t k) { return (float)(k >> 16); }
    static __device__ __forceinline__ float second(sec_t s) { return s == none2 ? 3.402823466e+38f : (float)s; }
};
template <>
struct BowKey<0> {  // float: distance bits << 32 | position, second-best distance bits
    using key_t = unsigned long long;
    using sec_t = unsigned;
    static constexpr key_t none = ~0ull;
    static constexpr sec_t none2 = ~0u;
    static __device__ __forceinline__ key_t make(float d, int b) { return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)b; }
    static __device__ __forceinline__ float dist(key_t k) { return __uint_as_float((unsigned)(k >> 32)); }
    static __device__ __forceinline__ float second(sec_t s) { return s == none2 ? 3.402823466e+38f : __uint_as_float(s); }
};
__device__ __forceinline__ const float *frow(const uint32_t *base, int idx, int dim) { return reinterpret_cast<const float *>(base) + (size_t)idx * dim; }

// core of M2 / M3 for one job, executed by a whole workgroup.  s_* are LDS scratch.
template <int W>
__device__ void match_bow_job(const DevMatchJob &J, uint32_t *s_matched, uint8_t *s_bin, int *s_red, int *s_hist) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool kf_frame = J.mode == AFV_MATCH_KF_FRAME;
    const int nout = kf_frame ? J.n2 : J.n1;
    for (int i = tid; i < nout; i += MT) J.out[i] = -1;
    for (int i = tid; i < (J.n2 + 31) / 32; i += MT) s_matched[i] = 0;
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) s_red[16] = 0;
    __syncthreads();

    for (int sg = 0; sg < J.nseg; ++sg) {
        const Seg S = J.segs[sg];
        for (int a = 0; a < S.n1; ++a) {
            const int idx1 = J.idx1 ? J.idx1[S.s1 + a] : S.s1 + a;
            if (J.valid1 && !J.valid1[idx1]) continue;  // uniform
            using BK = BowKey<W>;
            uint32_t q[W == 0 ? 1 : W];
            if constexpr (W != 0) {
#pragma unroll
                for (int i = 0; i < W; ++i) q[i] = J.d1[(size_t)idx1 * W + i];
            }
            typename BK::key_t k = BK::none;
            typename BK::sec_t s = BK::none2;
            for (int b = tid; b < S.n2; b += MT) {
                const int idx2 = J.idx2 ? J.idx2[S.s2 + b] : S.s2 + b;
                if ((s_matched[idx2 >> 5] >> (idx2 & 31)) & 1u) continue;
                if (!kf_frame && J.valid2 && !J.valid2[idx2]) continue;
                if constexpr (W == 0) {
                    (void)q;
                    merge_best(k, s, BK::make(l2sqr(frow(J.d1, idx1, J.fdim), frow(J.d2, idx2, J.fdim), J.fdim), b), BK::none2);
                } else {
                    const int d = hamming_words<W>(q, J.d2 + (size_t)idx2 * W);
                    merge_best(k, s, (d << 16) | b, NO_KEY >> 16);
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const typename BK::key_t k2 = __shfl_xor(k, m, 64);
                const typename BK::sec_t s2 = __shfl_xor(s, m, 64);
                merge_best(k, s, k2, s2);
            }
            // (binary: slots 2 wv, 2 wv + 1; float: the key's two halves and the second distance in slots 20 + 3 wv ..)
            if (lane == 0) {
                if constexpr (W == 0) {
                    s_red[20 + wv * 3] = (int)(unsigned)k;
                    s_red[21 + wv * 3] = (int)(unsigned)(k >> 32);
                    s_red[22 + wv * 3] = (int)s;
                } else {
                    s_red[wv * 2] = k;
                    s_red[wv * 2 + 1] = s;
                }
            }
            __syncthreads();
            if (tid == 0) {
                typename BK::key_t K;
                typename BK::sec_t Sx;
                if constexpr (W == 0) {
                    K = ((unsigned long long)(unsigned)s_red[21] << 32) | (unsigned)s_red[20];
                    Sx = (unsigned)s_red[22];
                    for (int w = 1; w < MT / 64; ++w)
                        merge_best(K, Sx, ((unsigned long long)(unsigned)s_red[21 + 3 * w] << 32) | (unsigned)s_red[20 + 3 * w], (unsigned)s_red[22 + 3 * w]);
                } else {
                    K = s_red[0], Sx = s_red[1];
                    for (int w = 1; w < MT / 64; ++w) merge_best(K, Sx, s_red[2 * w], s_red[2 * w + 1]);
                }
                if (K != BK::none) {
                    const float best1 = BK::dist(K);
                    const float best2 = BK::second(Sx);
                    const bool under = kf_frame ? (best1 <= J.th) : (best1 < J.th);  // FeatureMatcher.cc:250 / :630
                    if (under && best1 < J.ratio * best2) {                            // :252 / :632
                        const int b = (int)(K & 0xffff);
                        const int idx2 = J.idx2 ? J.idx2[S.s2 + b] : S.s2 + b;
                        const int key = kf_frame ? idx2 : idx1;
                        J.out[key] = kf_frame ? idx1 : idx2;
                        s_matched[idx2 >> 5] |= 1u << (idx2 & 31);
                        s_red[16]++;
                        if (J.check_ori) {
                            const int bin = rotation_bin(J.ang1[(size_t)idx1 * J.ang_stride], J.ang2[(size_t)idx2 * J.ang_stride]);
                            s_bin[key] = (uint8_t)bin;
                            s_hist[bin]++;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    // M6: keep only the three dominant rotation bins (computeThreeMaxima :1631-1668)
    if (J.check_ori) {
        if (tid == 0) {
            int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
            for (int i = 0; i < 30; ++i) {
                const int sz = s_hist[i];
                if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
                else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
                else if (sz > max3) { max3 = sz; i3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
            s_red[8] = i1; s_red[9] = i2; s_red[10] = i3;
        }
        __syncthreads();
        const int i1 = s_red[8], i2 = s_red[9], i3 = s_red[10];
        int dropped = 0;
        for (int i = tid; i < nout; i += MT) {
            if (J.out[i] >= 0) {
                const int b = s_bin[i];
                if (b != i1 && b != i2 && b != i3) {
                    J.out[i] = -1;
                    ++dropped;
                }
            }
        }
        if (dropped) atomicSub(&s_red[16], dropped);
        __syncthreads();
    }
    if (tid == 0) *J.nmatches = s_red[16];
}

__global__ __launch_bounds__(MT) void k_match_bow(const DevMatchJob *__restrict__ jobs) {
    __shared__ uint32_t s_matched[MAX_SIDE / 32];
    __shared__ uint8_t s_bin[MAX_SIDE];
    __shared__ int s_red[32];
    __shared__ int s_hist[32];
    const DevMatchJob J = jobs[blockIdx.x];
    if (J.fdim) match_bow_job<0>(J, s_matched, s_bin, s_red, s_hist);
    else if (J.words == 8) match_bow_job<8>(J, s_matched, s_bin, s_red, s_hist);
    else match_bow_job<16>(J, s_matched, s_bin, s_red, s_hist);
}

// ---------------- BoW-guided jobs: one WAVEFRONT per (job, shared node) ----------------
// vbMatched2 / vpMapPointMatches only couple rows inside ONE vocabulary node (a feature belongs to exactly one node),
// so the nodes of a job are independent: each wavefront walks the rows of its node in the reference's order, the 64
// lanes scan the node's columns, and the "taken" flags live in a per-wave LDS bitset indexed by the column's position
// inside the node.  Orientation bins go to a per-job histogram (global atomics); k_match_bow_finish applies M6.

// A node of at most 64 x 64 features (all of them at the shipped vocabulary: ~10 features per node of depth 2) lives in REGISTERS for the
// whole walk: lane a holds row a's descriptor, lane b column b's; a row's descriptor reaches the columns through eight v_readlane, the
// (best, second) pair through the DPP reduction of the pair matcher, the "taken" flags are one 64-bit scalar.  The walk itself touches
// no memory: the loop below cost four dependent memory round trips and a store drain PER ROW in the general form (50 us for one pair of
// frames, measured in round 5: a node with 12 rows = 12 x 4 us); results leave once, after the walk.
template <int W>
__device__ __forceinline__ void bow_segment_small(const DevMatchJob &J, const Seg S, int *hist, uint8_t *bins) {
    const int lane = threadIdx.x & 63;
    const bool kf_frame = J.mode == AFV_MATCH_KF_FRAME;
    const int idx1 = lane < S.n1 ? (J.idx1 ? J.idx1[S.s1 + lane] : S.s1 + lane) : -1;
    const int idx2 = lane < S.n2 ? (J.idx2 ? J.idx2[S.s2 + lane] : S.s2 + lane) : -1;
    const bool v1 = idx1 >= 0 && !(J.valid1 && !J.valid1[idx1]);
    const bool v2 = idx2 >= 0 && !(!kf_frame && J.valid2 && !J.valid2[idx2]);
    uint32_t rd[W], cd[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        rd[i] = idx1 >= 0 ? J.d1[(size_t)idx1 * W + i] : 0u;
        cd[i] = idx2 >= 0 ? J.d2[(size_t)idx2 * W + i] : 0u;
    }
    float a1 = 0.0f, a2 = 0.0f;
    if (J.check_ori) {
        if (idx1 >= 0) a1 = J.ang1[(size_t)idx1 * J.ang_stride];
        if (idx2 >= 0) a2 = J.ang2[(size_t)idx2 * J.ang_stride];
    }
    unsigned long long taken = ~__ballot(v2);  // columns that can never be taken count as taken
    const unsigned long long rows = __ballot(v1);
    int mine = -1, nm = 0;
    for (int a = 0; a < S.n1; ++a) {
        if (!((rows >> a) & 1ull)) continue;  // uniform
        int d = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) d += __popc((uint32_t)__builtin_amdgcn_readlane((int)rd[i], a) ^ cd[i]);
        int k = ((taken >> lane) & 1ull) ? NO_KEY : ((d << 16) | lane), s2 = NO_KEY >> 16;
        wave_merge_best(k, s2);
        if (k == NO_KEY) continue;
        const float best1 = (float)(k >> 16);
        const float best2 = (s2 == (NO_KEY >> 16)) ? 3.402823466e+38f : (float)s2;
        const bool under = kf_frame ? (best1 <= J.th) : (best1 < J.th);  // FeatureMatcher.cc:250 / :630
        if (under && best1 < J.ratio * best2) {                            // :252 / :632
            const int b = k & 0xffff;
            taken |= 1ull << b;
            if (lane == a) mine = b;
            ++nm;
        }
    }
    // results: row a's lane fetches its column's feature index (and angle) from the column's lane
    const int src = (mine >= 0 ? mine : 0) * 4;
    const int m_idx2 = __builtin_amdgcn_ds_bpermute(src, idx2);
    const float m_a2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, a2)));
    if (mine >= 0) {
        const int key = kf_frame ? m_idx2 : idx1;
        J.out[key] = kf_frame ? idx1 : m_idx2;
        if (J.check_ori) {
            const int bin = rotation_bin(a1, m_a2);
            bins[key] = (uint8_t)bin;
            atomicAdd(&hist[bin], 1);
        }
    }
    if (lane == 0 && nm) atomicAdd(J.nmatches, nm);
}

template <int W>
__device__ void bow_segment(const DevMatchJob &J, const Seg S, uint32_t *s_taken, int *hist, uint8_t *bins) {
    if constexpr (W != 0) {
        if (S.n1 <= 64 && S.n2 <= 64) {  // wave-uniform
            bow_segment_small<W>(J, S, hist, bins);
            return;
        }
    }
    using BK = BowKey<W>;
    const int lane = threadIdx.x & 63;
    const bool kf_frame = J.mode == AFV_MATCH_KF_FRAME;
    for (int i = lane; i < (S.n2 + 31) / 32; i += 64) s_taken[i] = 0;
    WAVE_LDS_SYNC();
    int nm = 0;
    for (int a = 0; a < S.n1; ++a) {
        const int idx1 = J.idx1 ? J.idx1[S.s1 + a] : S.s1 + a;
        if (J.valid1 && !J.valid1[idx1]) continue;  // uniform
        uint32_t q[W == 0 ? 1 : W];
        if constexpr (W != 0) {
#pragma unroll
            for (int i = 0; i < W; ++i) q[i] = J.d1[(size_t)idx1 * W + i];
        }
        typename BK::key_t k = BK::none;
        typename BK::sec_t s = BK::none2;
        for (int b = lane; b < S.n2; b += 64) {
            if ((s_taken[b >> 5] >> (b & 31)) & 1u) continue;
            const int idx2 = J.idx2 ? J.idx2[S.s2 + b] : S.s2 + b;
            if (!kf_frame && J.valid2 && !J.valid2[idx2]) continue;
            if constexpr (W == 0) {
                (void)q;
                merge_best(k, s, BK::make(l2sqr(frow(J.d1, idx1, J.fdim), frow(J.d2, idx2, J.fdim), J.fdim), b), BK::none2);
            } else {
                const int d = hamming_words<W>(q, J.d2 + (size_t)idx2 * W);
                merge_best(k, s, (d << 16) | b, NO_KEY >> 16);
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const typename BK::key_t k2 = __shfl_xor(k, m, 64);
            const typename BK::sec_t s2 = __shfl_xor(s, m, 64);
            merge_best(k, s, k2, s2);
        }
        if (k == BK::none) continue;
        const float best1 = BK::dist(k);
        const float best2 = BK::second(s);
        const bool under = kf_frame ? (best1 <= J.th) : (best1 < J.th);  // FeatureMatcher.cc:250 / :630
        if (under && best1 < J.ratio * best2) {                            // :252 / :632
            const int b = (int)(k & 0xffff);
            if (lane == 0) {
                const int idx2 = J.idx2 ? J.idx2[S.s2 + b] : S.s2 + b;
                const int key = kf_frame ? idx2 : idx1;
                J.out[key] = kf_frame ? idx1 : idx2;
                s_taken[b >> 5] |= 1u << (b & 31);
                if (J.check_ori) {
                    const int bin = rotation_bin(J.ang1[(size_t)idx1 * J.ang_stride], J.ang2[(size_t)idx2 * J.ang_stride]);
                    bins[key] = (uint8_t)bin;
                    atomicAdd(&hist[bin], 1);
                }
            }
            ++nm;
            WAVE_LDS_SYNC();
        }
    }
    if (lane == 0 && nm) atomicAdd(J.nmatches, nm);
}

// hist: [njobs][32] ints, bins: per job a byte per output slot (offsets in bin_off)
__global__ __launch_bounds__(MT) void k_match_bow_seg(const DevMatchJob *__restrict__ jobs, const SegTask *__restrict__ tasks,
                                                      int ntasks, int *__restrict__ hist, uint8_t *__restrict__ bins,
                                                      const int *__restrict__ bin_off) {
    __shared__ uint32_t s_taken[MT / 64][MAX_SIDE / 32];
    const int wv = threadIdx.x >> 6;
    const int t = blockIdx.x * (MT / 64) + wv;
    if (t >= ntasks) return;  // wave-uniform; no workgroup barrier below
    const SegTask T = tasks[t];
    const DevMatchJob J = jobs[T.job];
    const Seg S = J.segs[T.seg];
    if (J.fdim) bow_segment<0>(J, S, s_taken[wv], hist + T.job * 32, bins + bin_off[T.job]);
    else if (J.words == 8) bow_segment<8>(J, S, s_taken[wv], hist + T.job * 32, bins + bin_off[T.job]);
    else bow_segment<16>(J, S, s_taken[wv], hist + T.job * 32, bins + bin_off[T.job]);
}

// M6 for the per-node kernel: keep only the three dominant rotation bins (computeThreeMaxima :1631-1668)
__global__ __launch_bounds__(MT) void k_match_bow_finish(const DevMatchJob *__restrict__ jobs, const int *__restrict__ hist,
                                                         const uint8_t *__restrict__ bins, const int *__restrict__ bin_off) {
    __shared__ int s_i[3], s_drop;
    const DevMatchJob J = jobs[blockIdx.x];
    if (!J.check_ori) return;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int *h = hist + blockIdx.x * 32;
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < 30; ++i) {
            const int sz = h[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; i3 = i2; i2 = i1; i1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; i3 = i2; i2 = i; }
            else if (sz > max3) { max3 = sz; i3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
        s_i[0] = i1; s_i[1] = i2; s_i[2] = i3;
        s_drop = 0;
    }
    __syncthreads();
    const int nout = J.mode == AFV_MATCH_KF_FRAME ? J.n2 : J.n1;
    const uint8_t *bj = bins + bin_off[blockIdx.x];
    int dropped = 0;
    for (int i = tid; i < nout; i += MT) {
        if (J.out[i] >= 0) {
            const int b = bj[i];
            if (b != s_i[0] && b != s_i[1] && b != s_i[2]) {
                J.out[i] = -1;
                ++dropped;
            }
        }
    }
    if (dropped) atomicAdd(&s_drop, dropped);
    __syncthreads();
    if (tid == 0 && s_drop) *J.nmatches -= s_drop;
}

// ---------------- brute-force pairs, two-phase form (bench / pipeline path) ----------------
// SearchByBoW's greedy rule only ever removes columns.  So the K best columns of a row, ordered by
// (distance, position), contain the row's best and second-best UNMATCHED column unless more than K-2 of them have been
// taken by earlier rows.  Phase 1 (embarrassingly parallel, the 1e6 popcount distances per pair) computes each row's
// top-4 keys (a record of 2 x int4 per row, see k_match_resolve) with one LANE per row: the column descriptors are broadcast from LDS, no cross-lane traffic at all.
// Phase 2 walks the rows in the reference's order on one wavefront per pair, consuming those 4 keys; a row whose best
// distance already fails TH_LOW is skipped (it can never match), and a row that runs out of keys falls back to an exact
// wave-wide rescan of the unmatched columns.  Results are identical to the sequential algorithm.
#define TOPK 4
#define COL_TILE 1024

__global__ __launch_bounds__(MT) void k_match_topk(const uint8_t *__restrict__ desc, const int *__restrict__ nset, int cap,
                                                   const int *__restrict__ pair_a, const int *__restrict__ pair_b,
                                                   int4 *__restrict__ topk, int pair_base) {
    __shared__ __attribute__((aligned(16))) uint32_t s_cols[COL_TILE * 8];
    const int p = pair_base + blockIdx.y;
    const int a = pair_a[p], b = pair_b[p];
    const int n1 = min(nset[a], cap), n2 = min(nset[b], cap);
    const int row = blockIdx.x * MT + threadIdx.x;
    if (blockIdx.x * MT >= n1) return;  // uniform
    const uint4 *qa = reinterpret_cast<const uint4 *>(desc + ((size_t)a * cap + min(row, n1 - 1)) * 32);
    const uint4 q0 = qa[0], q1 = qa[1];
    int k[TOPK] = {NO_KEY, NO_KEY, NO_KEY, NO_KEY};
    const uint4 *cb = reinterpret_cast<const uint4 *>(desc + (size_t)b * cap * 32);
    for (int c0 = 0; c0 < n2; c0 += COL_TILE) {
        const int nc = min(COL_TILE, n2 - c0);
        __syncthreads();
        for (int i = threadIdx.x; i < nc * 2; i += MT) reinterpret_cast<uint4 *>(s_cols)[i] = cb[(size_t)c0 * 2 + i];
        __syncthreads();
        for (int j = 0; j < nc; ++j) {
            const uint4 c_lo = reinterpret_cast<const uint4 *>(s_cols)[2 * j], c_hi = reinterpret_cast<const uint4 *>(s_cols)[2 * j + 1];
            int d = __popc(q0.x ^ c_lo.x) + __popc(q0.y ^ c_lo.y) + __popc(q0.z ^ c_lo.z) + __popc(q0.w ^ c_lo.w);
            d += __popc(q1.x ^ c_hi.x) + __popc(q1.y ^ c_hi.y) + __popc(q1.z ^ c_hi.z) + __popc(q1.w ^ c_hi.w);
            const int key = (d << 16) | (c0 + j);
            // sorted insert into k[0] <= k[1] <= k[2] <= k[3], the largest of the five falls out: the new k[i] is the median of
            // (k[i-1], k[i], key) — one v_min + three v_med3 per column
            const int n3 = med3_i32(k[2], k[3], key), n2 = med3_i32(k[1], k[2], key);
            const int n1 = med3_i32(k[0], k[1], key);
            k[0] = min(k[0], key); k[1] = n1; k[2] = n2; k[3] = n3;
        }
    }
    if (row < n1) {  // record = 7 key slots + the number of EXACT leading keys (this engine: four)
        topk[((size_t)p * cap + row) * 2] = make_int4(k[0], k[1], k[2], k[3]);
        topk[((size_t)p * cap + row) * 2 + 1] = make_int4(NO_KEY, NO_KEY, NO_KEY, TOPK);
    }
}

#define PAIR_MAX_SIDE 4096  // rows / columns per set in the pairs path (16-bit row / column indices)
#define PAIR_KEYS_LDS 1024  // the key records of at most this many live rows are staged in LDS
#define RKEYS 7             // key slots of a record (int4 x 2: seven keys + the exact-prefix length)

// LDS of one k_match_resolve workgroup, sized by the per-set capacity of the launch (23 KB at cap = 1024, so several pairs share a
// CU: the ordered walk is one wavefront deep and latency-bound, what a batch costs is set by how many walks run at once)
#define PAIR_COLS_LDS 1024  // sets up to this capacity also keep the column descriptors in LDS (exact rescans of the walk): 56 KB in all,
                            // below the 64 KB a launch may ask for without raising the function's dynamic-LDS limit
static inline size_t resolve_lds_bytes(int cap, bool stage_cols) {
    const size_t c = ((size_t)cap + 63) & ~(size_t)63;
    return std::min<size_t>(c, PAIR_KEYS_LDS) * 32 /*key records*/ + c * 4 /*claim*/ + c * 2 /*live*/ + c /*bin*/ + c / 8 /*matched*/ +
           (stage_cols ? c * 32 + 16 : 0) /*columns, 16-byte aligned*/;
}

__global__ __launch_bounds__(MT, 2) void k_match_resolve(const uint8_t *__restrict__ desc, const float *__restrict__ ang, int ang_stride,
                                                      const int *__restrict__ nset, int cap, const int *__restrict__ pair_a,
                                                      const int *__restrict__ pair_b, const int4 *__restrict__ topk, float th,
                                                      float ratio, int check_ori, int *__restrict__ match,
                                                      int *__restrict__ nmatches, int pair_base, int stage_cols) {
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int capr = (cap + 63) & ~63;
    int4 *s_keys = reinterpret_cast<int4 *>(s_dyn);  // key records (2 x int4) of the live rows (first PAIR_KEYS_LDS of them)
    int *s_claim = reinterpret_cast<int *>(s_keys + 2 * min(capr, PAIR_KEYS_LDS));
    unsigned short *s_live = reinterpret_cast<unsigned short *>(s_claim + capr);
    uint8_t *s_bin = reinterpret_cast<uint8_t *>(s_live + capr);
    uint32_t *s_matched = reinterpret_cast<uint32_t *>(s_bin + capr);
    const bool cols_in_lds = stage_cols != 0;
    uint32_t *s_cols = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(s_matched + capr / 32) + 15) & ~(uintptr_t)15);
    __shared__ int s_cols_ready;
#ifdef AFV_RESOLVE_STATS
    __shared__ long long s_t[2];
#endif
    __shared__ int s_hist[32];
    __shared__ int s_wave[8];
    __shared__ int s_nm, s_drop[3];
    const int p = pair_base + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef AFV_RESOLVE_STATS
    const long long st_k0 = wall_clock64();
#endif
    const int a = pair_a[p], b = pair_b[p];
    const int n1 = min(nset[a], cap), n2 = min(nset[b], cap);
    const int4 *tk = topk + (size_t)p * cap * 2;
    int *out = match + (size_t)p * cap;
    for (int i = tid; i < cap; i += MT) out[i] = -1;
    for (int i = tid; i < (n2 + 31) / 32; i += MT) s_matched[i] = 0;
    for (int i = tid; i < n2; i += MT) s_claim[i] = 0x7fffffff;
    if (tid < 32) s_hist[tid] = 0;
    if (tid == 0) s_cols_ready = 0;
    // rows whose best distance fails TH_LOW can never match: compact the others IN ROW ORDER, keys staged in LDS
    int nlive = 0;
    for (int i0 = 0; i0 < n1; i0 += MT) {
        const int i = i0 + tid;
        int4 t4 = make_int4(NO_KEY, NO_KEY, NO_KEY, NO_KEY), t8 = make_int4(NO_KEY, NO_KEY, NO_KEY, TOPK);
        if (i < n1) {
            t4 = tk[2 * i];
            t8 = tk[2 * i + 1];
        }
        const bool live = t4.x != NO_KEY && (float)(t4.x >> 16) < th;
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = nlive;
        for (int w = 0; w < wv; ++w) off += s_wave[w];
        if (live) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_live[slot] = (unsigned short)i;
            if (slot < PAIR_KEYS_LDS) {
                s_keys[2 * slot] = t4;
                s_keys[2 * slot + 1] = t8;
            }
        }
        nlive += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const uint32_t *d1 = reinterpret_cast<const uint32_t *>(desc + (size_t)a * cap * 32);
    const uint32_t *d2 = reinterpret_cast<const uint32_t *>(desc + (size_t)b * cap * 32);
#ifdef AFV_RESOLVE_STATS
    if (tid == 0) s_t[0] = wall_clock64();
#endif
    if (wv != 0) {
        // The ordered walk below is one wavefront deep.  While it runs, the other three wavefronts copy the column descriptors into
        // LDS (when the launch reserved room for them) so that the exact rescans of the walk read LDS instead of L2; a flag in
        // LDS, not a barrier, hands the copy over.
        if (cols_in_lds && nlive > 0) {
            uint4 *sc = reinterpret_cast<uint4 *>(s_cols);
            const uint4 *gc = reinterpret_cast<const uint4 *>(d2);
            // all loads of a thread in flight together (<= 11 x 16 B at PAIR_COLS_LDS = 1024): two round trips instead of eleven
            static_assert(PAIR_COLS_LDS * 2 <= 11 * (MT - 64), "column copy: eleven 16-byte loads per thread");
            const int last = n2 * 2 - 1, i0 = tid - 64;
#define AFV_COL(k) const uint4 v##k = gc[min(i0 + (k) * (MT - 64), last)];
            AFV_COL(0) AFV_COL(1) AFV_COL(2) AFV_COL(3) AFV_COL(4) AFV_COL(5) AFV_COL(6) AFV_COL(7) AFV_COL(8) AFV_COL(9) AFV_COL(10)
#undef AFV_COL
#define AFV_COL(k) if (i0 + (k) * (MT - 64) <= last) sc[i0 + (k) * (MT - 64)] = v##k;
            AFV_COL(0) AFV_COL(1) AFV_COL(2) AFV_COL(3) AFV_COL(4) AFV_COL(5) AFV_COL(6) AFV_COL(7) AFV_COL(8) AFV_COL(9) AFV_COL(10)
#undef AFV_COL
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) atomicAdd(&s_cols_ready, 1);
        }
    } else {
        // Ordered walk, 64 live rows per round, solved as a FIXED POINT instead of "commit the clean prefix, replay the rest":
        // every lane evaluates ITS row against the matched set plus the columns claimed by EARLIER lanes of the round (claim =
        // LDS atomic min of the lane index), claims are rebuilt, and the evaluation repeats until no lane changes its claim.  Lane k
        // depends only on lanes < k, so lane k is final after at most k + 1 passes (in practice the depth of the longest chain of
        // rows competing for a column: 2..4 with the same corner detected on several pyramid levels) and the fixed point IS the
        // outcome of the sequential loop (FeatureMatcher.cc:587-641).  A row whose four keys are used up needs the exact rescan of
        // the unmatched columns: the round is cut there, the rows before it are committed, the rescan runs, the walk goes on.
        int nm = 0;
        int pos = 0;
        bool cols_ready = false;
#ifdef AFV_RESOLVE_STATS
        int st_rounds = 0, st_rescans = 0, st_iters = 0, st_racc = 0;
        long long st_pre = 0, st_pass = 0, st_commit = 0, st_resc = 0, st_mark = 0, st_r1 = 0, st_r2 = 0, st_r3 = 0;
        const long long st_t0 = wall_clock64();
#endif
        while (pos < nlive) {
#ifdef AFV_RESOLVE_STATS
            ++st_rounds;
            st_mark = wall_clock64();
#endif
            const int li = pos + lane;
            const bool act = li < nlive;
            const int row = act ? s_live[li] : 0;
            int4 t4 = make_int4(NO_KEY, NO_KEY, NO_KEY, NO_KEY), t8 = make_int4(NO_KEY, NO_KEY, NO_KEY, TOPK);
            if (act) {
                t4 = li < PAIR_KEYS_LDS ? s_keys[2 * li] : tk[2 * row];
                t8 = li < PAIR_KEYS_LDS ? s_keys[2 * li + 1] : tk[2 * row + 1];
            }
            // the first nk keys are the row's nk nearest columns EXACTLY (4 from the popcount engine; 4..7 from the matrix-core engine,
            // whose two lane halves each keep a top-4 over half of the columns: the merged list is exact up to the first half's 4th key)
            const int keys[RKEYS] = {t4.x, t4.y, t4.z, t4.w, t8.x, t8.y, t8.z};
            const int nk = min(max(t8.w, 1), RKEYS);
            int last_key = keys[0];
#pragma unroll
            for (int q = 1; q < RKEYS; ++q) last_key = q < nk ? keys[q] : last_key;
            const float last_d = (float)(last_key >> 16);  // every column outside the list is at least this far
            // every lane fetches its row's descriptor now: if the round is cut at this lane, the rescan needs it, and the round trip
            // hides behind the fixed-point passes
            const uint4 *qp = reinterpret_cast<const uint4 *>(d1 + (size_t)row * 8);
            const uint4 qlo = qp[0], qhi = qp[1];
            // the matched set does not change inside a round: one look per key.  v[q] = the key while it can still be chosen, else
            // "none"; cq[q] = its column (0 for a slot that holds no key: a harmless claim lookup)
            int v[RKEYS], cq[RKEYS];
            bool complete = false;  // a NO_KEY inside the exact prefix: the row has fewer than nk columns, the list is all there is
#pragma unroll
            for (int q = 0; q < RKEYS; ++q) {
                const bool has = q < nk && keys[q] != NO_KEY;
                complete = complete || (q < nk && keys[q] == NO_KEY);
                cq[q] = has ? (keys[q] & 0xffff) : 0;
                const bool gone = has && ((s_matched[cq[q] >> 5] >> (cq[q] & 31)) & 1u);
                v[q] = (has && !gone) ? keys[q] : NO_KEY;
            }
            int type = 0, e0 = -1, my_claim = -1;
#ifdef AFV_RESOLVE_STATS
            if (v[0] == 12345 && lane == 99) st_pre = 1;  // keep the loads above the timer
            { const long long t = wall_clock64(); st_pre += t - st_mark; st_mark = t; }
#endif
            for (int pass = 0; pass < 66; ++pass) {
#ifdef AFV_RESOLVE_STATS
                ++st_iters;
#endif
                // branch-free: the keys are unique and sorted, so best / second-best available = the two smallest of the surviving slots
                int cl[RKEYS], a[RKEYS];
#pragma unroll
                for (int q = 0; q < RKEYS; ++q) cl[q] = s_claim[cq[q]];  // seven LDS reads in flight together
#pragma unroll
                for (int q = 0; q < RKEYS; ++q) a[q] = (cl[q] < lane) ? NO_KEY : v[q];
                const int best = act ? min(min(min(a[0], a[1]), min(a[2], a[3])), min(min(a[4], a[5]), a[6])) : NO_KEY;
#pragma unroll
                for (int q = 0; q < RKEYS; ++q) a[q] = a[q] > best ? a[q] : NO_KEY;
                const int second_key = min(min(min(a[0], a[1]), min(a[2], a[3])), min(min(a[4], a[5]), a[6]));
                const int second = second_key == NO_KEY ? -1 : (second_key >> 16);
                // the walk through the list ends early at the second available key or at the end of a complete list; otherwise it ran
                // out of exact keys
                const bool exhausted = act && second < 0 && !complete;
                e0 = best == NO_KEY ? -1 : (best & 0xffff);
                type = 0;  // 0 = no match, 1 = accept column e0, 2 = exact rescan needed
                if (act) {
                    if (best != NO_KEY && !((float)(best >> 16) < th)) {
                        // the best unmatched column already fails TH_LOW: final whatever happens to the set
                    } else if (exhausted && n2 > nk) {
                        // the (second-)best unmatched column lies beyond the exact keys: it is at least as far as the last of them, so
                        // the ratio test is already decided when it passes against that lower bound, and a row without any key left
                        // cannot match when even that bound fails TH_LOW.
                        if (best != NO_KEY) type = ((float)(best >> 16) < ratio * last_d) ? 1 : 2;
                        else type = (last_d < th) ? 2 : 0;
                    } else if (best != NO_KEY) {
                        const float best1 = (float)(best >> 16);
                        const float best2 = second < 0 ? 3.402823466e+38f : (float)second;
                        type = (best1 < th && best1 < ratio * best2) ? 1 : 0;  // FeatureMatcher.cc:630,632
                    }
                }
                const int want = type == 1 ? e0 : -1;
                if (!__ballot(want != my_claim)) break;  // fixed point
                // rebuild the claims: release all, then claim all (two lanes may hold the same column: the lower one must survive)
                if (my_claim >= 0) s_claim[my_claim] = 0x7fffffff;
                WAVE_LDS_ONLY_SYNC();
                if (want >= 0) atomicMin(&s_claim[want], lane);
                my_claim = want;
                WAVE_LDS_ONLY_SYNC();
            }
#ifdef AFV_RESOLVE_STATS
            { const long long t = wall_clock64(); st_pass += t - st_mark; st_mark = t; }
#endif
            // The converged round holds for every lane UP TO the first one that needs the exact rescan - and, when that rescan ends in
            // "no match" (it almost always does: on overlapping video frames 0-1 of a pair's 8-30 rescans take a column), beyond it:
            // the fixed point already treated that lane as taking nothing.  So the lanes before a rescan lane are committed, the
            // rescan runs against the then exact matched set, and only a rescan that TAKES a column cuts the round (round 4: before,
            // every rescan cut it and paid a new round set-up + two passes: 25 of them cost a slow pair 50 us).
            unsigned long long sm = __ballot(type == 2);
            int done = 0;  // lanes [0, done) of the round are settled
            while (true) {
                const int stop = sm ? (int)__builtin_ctzll(sm) : 64;
                const bool commit = type == 1 && lane >= done && lane < stop;
                if (commit) {
                    out[row] = e0;
                    atomicOr(&s_matched[e0 >> 5], 1u << (e0 & 31));
                }
                nm += __popcll(__ballot(commit));
                done = stop;
                if (stop == 64 || pos + stop >= nlive) break;
                WAVE_LDS_ONLY_SYNC();
                // row `pos + stop` needs the exact rescan of the unmatched columns: whole wave, columns from LDS once the copy has landed
#ifdef AFV_RESOLVE_STATS
                ++st_rescans;
#endif
                if (cols_in_lds && !cols_ready) {
                    while (__hip_atomic_load(&s_cols_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < MT / 64 - 1) __builtin_amdgcn_s_sleep(1);
                    cols_ready = true;
                }
                const uint4 *cb = cols_in_lds ? reinterpret_cast<const uint4 *>(s_cols) : reinterpret_cast<const uint4 *>(d2);
                const int i = __builtin_amdgcn_readlane(row, stop);
                const uint32_t qv[8] = {qlo.x, qlo.y, qlo.z, qlo.w, qhi.x, qhi.y, qhi.z, qhi.w};
                uint32_t q[8];
#pragma unroll
                for (int w = 0; w < 8; ++w) q[w] = (uint32_t)__builtin_amdgcn_readlane((int)qv[w], stop);
                int k = NO_KEY, s2nd = NO_KEY >> 16;
                // four columns per lane and step, branch-free: the eight 16-byte loads and the four matched-bit words are in flight
                // together, a taken or out-of-range column enters as the "no column" key.  k = best key, s2nd = second-best distance
                // (invariant s2nd >= k >> 16): inserting a key is  s2nd = min(s2nd, max(k, key) >> 16), k = min(k, key).
                for (int c0 = lane; c0 < n2; c0 += 256) {
                    uint4 lo[4], hi[4];
                    uint32_t mw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = min(c0 + 64 * u, n2 - 1);
                        lo[u] = cb[2 * c];
                        hi[u] = cb[2 * c + 1];
                        mw[u] = s_matched[c >> 5];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + 64 * u;
                        const int d = __popc(q[0] ^ lo[u].x) + __popc(q[1] ^ lo[u].y) + __popc(q[2] ^ lo[u].z) + __popc(q[3] ^ lo[u].w) +
                                      __popc(q[4] ^ hi[u].x) + __popc(q[5] ^ hi[u].y) + __popc(q[6] ^ hi[u].z) + __popc(q[7] ^ hi[u].w);
                        const bool usable = c < n2 && !((mw[u] >> (c & 31)) & 1u);
                        const int key = usable ? ((d << 16) | c) : NO_KEY;
                        s2nd = min(s2nd, max(k, key) >> 16);
                        k = min(k, key);
                    }
                }
                wave_merge_best(k, s2nd);
                bool took = false;
                if (k != NO_KEY) {
                    const float best1 = (float)(k >> 16);
                    const float best2 = (s2nd == (NO_KEY >> 16)) ? 3.402823466e+38f : (float)s2nd;
                    if (best1 < th && best1 < ratio * best2) {
                        const int bcol = k & 0xffff;
                        if (lane == 0) {
                            out[i] = bcol;
                            s_matched[bcol >> 5] |= 1u << (bcol & 31);
                        }
                        ++nm;
                        took = true;
#ifdef AFV_RESOLVE_STATS
                        ++st_racc;
#endif
                    }
                }
                done = stop + 1;
                if (took) break;  // the lanes behind it were evaluated without this column taken: new round from there
                sm &= ~(1ull << stop);
            }
            if (my_claim >= 0) s_claim[my_claim] = 0x7fffffff;  // release every claim of this round
            WAVE_LDS_ONLY_SYNC();
            pos += done;
#ifdef AFV_RESOLVE_STATS
            { const long long t = wall_clock64(); st_commit += t - st_mark; st_mark = t; }
#endif
        }
        if (lane == 0) s_nm = nm;
#ifdef AFV_RESOLVE_STATS
        if (lane == 0) s_t[1] = wall_clock64();
#endif
#ifdef AFV_RESOLVE_STATS
        if (AFV_RESOLVE_STATS == 3 && lane == 0) printf("pair %d nlive %d rounds %d passes %d rescans %d accepting %d matches %d walk_us %lld\n", p, nlive, st_rounds, st_iters, st_rescans, st_racc, nm, (wall_clock64() - st_t0) / 100);
        if (AFV_RESOLVE_STATS == 1 && lane == 0 && p <= 2) printf("resolve pair %d: n1 %d nlive %d rounds %d passes %d rescans %d matches %d walk %lld us (prologue %lld passes %lld commit %lld rescan %lld = wait %lld scan %lld reduce %lld)\n", p, n1, nlive, st_rounds, st_iters, st_rescans, nm, (wall_clock64() - st_t0) / 100, st_pre / 100, st_pass / 100, st_commit / 100, st_resc / 100, st_r1 / 100, st_r2 / 100, st_r3 / 100);
#endif
    }
    __syncthreads();
    if (check_ori) {
        // rotation histogram of the accepted matches (FeatureMatcher.cc:1587-1599): it never influences the walk, so
        // it is built afterwards by the whole workgroup
        for (int i = tid; i < n1; i += MT) {
            const int c = out[i];
            if (c >= 0) {
                const int bin = rotation_bin(ang[((size_t)a * cap + i) * ang_stride], ang[((size_t)b * cap + c) * ang_stride]);
                s_bin[i] = (uint8_t)bin;
                atomicAdd(&s_hist[bin], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {  // computeThreeMaxima (FeatureMatcher.cc:1631-1668)
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
        for (int i = tid; i < n1; i += MT) {
            if (out[i] >= 0) {
                const int bb = s_bin[i];
                if (bb != i1 && bb != i2 && bb != i3) { out[i] = -1; ++dropped; }
            }
        }
        if (dropped) atomicSub(&s_nm, dropped);
        __syncthreads();
    }
    if (tid == 0) nmatches[p] = s_nm;
#ifdef AFV_RESOLVE_STATS
    if (AFV_RESOLVE_STATS == 2 && tid == 0 && p <= 2)
        printf("resolve pair %d: whole workgroup %lld us (set-up + compaction %lld, walk %lld, histogram %lld)\n", p, (wall_clock64() - st_k0) / 100,
               (s_t[0] - st_k0) / 100, (s_t[1] - s_t[0]) / 100, (wall_clock64() - s_t[1]) / 100);
#endif
}

// ---------------- phase 2, workgroup form (round 4): ONE fixed point over all live rows, 1024 threads ----------------
// k_match_resolve above walks the live rows 64 at a time on one wavefront: a pair of consecutive video frames (~900 live rows) costs
// 15-20 rounds of ~2 us of dependent LDS round trips, and a pair of unrelated frames still 11 rounds although hardly a row matches.
// The fixed-point argument of that walk does not need the rounds: row i's decision is a function f of the columns WANTED by the rows
// before it, want_i = f({want_j : j < i}), so any assignment that satisfies all these equations IS the sequential outcome (induction
// on i), and iterating "every row re-evaluates f against the current claims" reaches it after as many passes as the longest chain of
// rows competing for a column (2..4 in practice) + 1.  Here all live rows do that at once, a thread per row (a pass of four rows per
// thread on 256 threads cost 4-5 us of dependent LDS round trips; one row per thread: about 1.5):
//   * claims live in THREE rotating LDS arrays (claim[c] = smallest live index that wants column c, by atomic min): a pass reads the
//     array written by the previous pass, writes the next one and clears its own entries of the third - ONE barrier per pass, which
//     also carries the "did anything change" vote;
//   * a row whose exact keys are used up needs the exact rescan of the free columns, which is only meaningful once every row before it
//     is final: after convergence the waiting rows are rescanned, up to 32 at a time (two per wavefront), against the claims of the
//     rows before them, the answers up to the first that takes a column are pinned, and the iteration continues;
//   * there is no matched-set bitmap: "column c is taken for row i" is claim[c] < i.
#define RW_INF 0x7fffffff
#define AFV_RESOLVE_GUARD (-0x7fffffff)  // nmatches of a pair whose fixed point hit its pass guard
#define RW_WLIST 128  // waiting rows looked at per convergence
#define RW_RPW 2      // waiting rows a wavefront rescans together (they share the column loads): 32 per step
#define RWT 1024      // threads: one per live row
#define RW_NW (RWT / 64)
static inline size_t resolve_wg_lds_bytes(int cap, bool stage_cols) {
    const size_t c = ((size_t)cap + 63) & ~(size_t)63;
    return std::min<size_t>(c, PAIR_KEYS_LDS) * 32 /*key records*/ + 3 * c * 4 /*claims*/ + c * 4 /*matches*/ + 2 * c * 2 /*wants of the last two passes*/ +
           c * 2 /*live*/ + c /*bin*/ + c /*flags*/ + (stage_cols ? c * 32 + 16 : 0) /*columns, 16-byte aligned*/;
}

// one row against the claims in R: the column it accepts (-1: none) and whether it needs the exact rescan.  Same decisions as the walk of
// k_match_resolve (FeatureMatcher.cc:587-641), "taken" = claimed by an earlier live row
__device__ __forceinline__ void resolve_eval(const int4 t4, const int4 t8, const int *R, int li, int n2, float th, float ratio, int &want, bool &rescan) {
    const int keys[RKEYS] = {t4.x, t4.y, t4.z, t4.w, t8.x, t8.y, t8.z};
    const int nk = min(max(t8.w, 1), RKEYS);
    int last_key = keys[0];
#pragma unroll
    for (int q = 1; q < RKEYS; ++q) last_key = q < nk ? keys[q] : last_key;
    const float last_d = (float)(last_key >> 16);  // every column outside the list is at least this far
    int v[RKEYS], cq[RKEYS];
    bool complete = false;  // a NO_KEY inside the exact prefix: the row has fewer than nk columns, the list is all there is
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) {
        const bool has = q < nk && keys[q] != NO_KEY;
        complete = complete || (q < nk && keys[q] == NO_KEY);
        cq[q] = has ? (keys[q] & 0xffff) : 0;
        v[q] = has ? keys[q] : NO_KEY;
    }
    int cl[RKEYS], a[RKEYS];
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) cl[q] = R[cq[q]];  // seven LDS reads in flight together
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) a[q] = (cl[q] < li) ? NO_KEY : v[q];
    const int best = min(min(min(a[0], a[1]), min(a[2], a[3])), min(min(a[4], a[5]), a[6]));
#pragma unroll
    for (int q = 0; q < RKEYS; ++q) a[q] = a[q] > best ? a[q] : NO_KEY;
    const int second_key = min(min(min(a[0], a[1]), min(a[2], a[3])), min(min(a[4], a[5]), a[6]));
    const int second = second_key == NO_KEY ? -1 : (second_key >> 16);
    const bool exhausted = second < 0 && !complete;
    const int e0 = best == NO_KEY ? -1 : (best & 0xffff);
    int type = 0;  // 0 = no match, 1 = accept column e0, 2 = exact rescan needed
    if (best != NO_KEY && !((float)(best >> 16) < th)) {
        // the best free column already fails TH_LOW: final whatever happens to the set
    } else if (exhausted && n2 > nk) {
        if (best != NO_KEY) type = ((float)(best >> 16) < ratio * last_d) ? 1 : 2;
        else type = (last_d < th) ? 2 : 0;
    } else if (best != NO_KEY) {
        const float best1 = (float)(best >> 16);
        const float best2 = second < 0 ? 3.402823466e+38f : (float)second;
        type = (best1 < th && best1 < ratio * best2) ? 1 : 0;  // FeatureMatcher.cc:630,632
    }
    want = type == 1 ? e0 : -1;
    rescan = type == 2;
}

__global__ __launch_bounds__(RWT) void k_match_resolve_wg(const uint8_t *__restrict__ desc, const float *__restrict__ ang, int ang_stride,
                                                         const int *__restrict__ nset, int cap, const int *__restrict__ pair_a,
                                                         const int *__restrict__ pair_b, const int4 *__restrict__ topk, float th,
                                                         float ratio, int check_ori, int *__restrict__ match,
                                                         int *__restrict__ nmatches, int pair_base, int stage_cols, int pass_cap) {
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    const int capr = (cap + 63) & ~63;
    int4 *s_keys = reinterpret_cast<int4 *>(s_dyn);  // key records (2 x int4) of the live rows (first PAIR_KEYS_LDS of them)
    int *s_claim = reinterpret_cast<int *>(s_keys + 2 * min(capr, PAIR_KEYS_LDS));  // three arrays of capr
    int *s_out = s_claim + 3 * capr;
    short *s_w1 = reinterpret_cast<short *>(s_out + capr), *s_w2 = s_w1 + capr;  // a live row's want after the last / the last but one pass
    unsigned short *s_live = reinterpret_cast<unsigned short *>(s_w2 + capr);
    uint8_t *s_bin = reinterpret_cast<uint8_t *>(s_live + capr);
    uint8_t *s_flag = s_bin + capr;  // per live row: 1 = asked for a rescan in the last pass, 2 = pinned by a rescan
    uint32_t *s_cols = reinterpret_cast<uint32_t *>((reinterpret_cast<uintptr_t>(s_flag + capr) + 15) & ~(uintptr_t)15);
    __shared__ int s_hist[32];
    __shared__ int s_nm, s_drop[3], s_first, s_part[RW_RPW * RW_NW], s_cntw[RW_NW];
    __shared__ unsigned short s_wlist[RW_WLIST];  // live indices of the rows waiting for a rescan, in row order
    const int p = pair_base + blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int a = pair_a[p], b = pair_b[p];
    const int n1 = min(nset[a], cap), n2 = min(nset[b], cap);
    const int4 *tk = topk + (size_t)p * cap * 2;
    int *out = match + (size_t)p * cap;
    for (int i = tid; i < capr; i += RWT) s_out[i] = -1;
    for (int i = tid; i < 3 * capr; i += RWT) s_claim[i] = RW_INF;
    if (tid < 32) s_hist[tid] = 0;
    // rows whose best distance fails TH_LOW can never match: compact the others IN ROW ORDER, keys staged in LDS.  A thread per row (sets of
    // up to 1024 rows - the usual case - in ONE step: one L2 round trip for the records, the 16 wavefronts' live counts meet in LDS behind
    // one barrier, every thread places its row from them)
    int nlive = 0;
    for (int i0 = 0; i0 < n1; i0 += RWT) {
        const int i = i0 + tid;
        int4 t4 = make_int4(NO_KEY, NO_KEY, NO_KEY, NO_KEY), t8 = make_int4(NO_KEY, NO_KEY, NO_KEY, TOPK);
        if (i < n1) {
            t4 = tk[2 * i];
            t8 = tk[2 * i + 1];
        }
        const bool live = t4.x != NO_KEY && (float)(t4.x >> 16) < th;
        const unsigned long long m = __ballot(live);
        if (lane == 0) s_cntw[wv] = __popcll(m);
        __syncthreads();
        int off = nlive, tot = 0;
#pragma unroll
        for (int w = 0; w < RW_NW; ++w) {
            const int cw = s_cntw[w];
            off += w < wv ? cw : 0;
            tot += cw;
        }
        if (live) {
            const int slot = off + __popcll(m & ((1ull << lane) - 1ull));
            s_live[slot] = (unsigned short)i;
            s_w1[slot] = -1;
            s_w2[slot] = -1;
            s_flag[slot] = 0;
            if (slot < PAIR_KEYS_LDS) {
                s_keys[2 * slot] = t4;
                s_keys[2 * slot + 1] = t8;
            }
        }
        nlive += tot;
        __syncthreads();
    }
    const uint32_t *d1 = reinterpret_cast<const uint32_t *>(desc + (size_t)a * cap * 32);
    const uint32_t *d2 = reinterpret_cast<const uint32_t *>(desc + (size_t)b * cap * 32);
#if defined(AFV_RESOLVE_STATS) && AFV_RESOLVE_STATS == 4
    const long long wg_t0 = wall_clock64();
    long long wg_tresc = 0;
    int wg_steps = 0, wg_resc = 0, wg_took = 0, wg_conv = 0;
#endif
    // ---- the fixed point ----
    bool cols_ready = false;
    int pass = 0;
    // every pass finalises at least one more row or rescans one: a guard, never reached - if it ever is, the pair is REPORTED
    // (nmatches = AFV_RESOLVE_GUARD, which the host entry points turn into AFV_EHIP), not returned half settled
    const int pass_limit = pass_cap > 0 ? pass_cap : 3 * nlive + 64;
    bool guard_hit = false;
    while (nlive > 0) {
        if (pass >= pass_limit) {
            guard_hit = true;
            break;
        }
        int *R = s_claim + (pass % 3) * capr, *W = s_claim + ((pass + 1) % 3) * capr, *Z = s_claim + ((pass + 2) % 3) * capr;
        bool changed = false;
        for (int li = tid; li < nlive; li += RWT) {
            const int w1 = s_w1[li], w2 = s_w2[li];
            const int flag = s_flag[li];
            int want = 0;
            bool rescan = false;
            if (flag & 2) {
                want = s_out[s_live[li]];  // pinned by its rescan: the row asserts that answer in every pass
            } else {
                int4 t4, t8;
                if (li < PAIR_KEYS_LDS) {
                    t4 = s_keys[2 * li];
                    t8 = s_keys[2 * li + 1];
                } else {
                    const int row = s_live[li];
                    t4 = tk[2 * row];
                    t8 = tk[2 * row + 1];
                }
                resolve_eval(t4, t8, R, li, n2, th, ratio, want, rescan);
                s_flag[li] = rescan ? 1 : 0;
            }
            if (want >= 0) atomicMin(&W[want], li);
            if (w2 >= 0) Z[w2] = RW_INF;  // what this row put into Z two passes ago (every row that did clears it: the array is empty before it is written again)
            s_w2[li] = (short)w1;
            s_w1[li] = (short)want;
            changed = changed || want != w1 || (rescan != ((flag & 1) != 0));
        }
        ++pass;
        if (__syncthreads_or(changed ? 1 : 0)) continue;
        // converged: W holds the claims of the final wants (so far).  The rows that asked for a rescan, in row order (s_wlist): live row
        // t is held by thread t, so a ballot + the wavefronts' counts place them
        {
            const bool waits = tid < nlive && (s_flag[tid] & 3) == 1;  // the first 1024 live rows are looked at per cycle
            const unsigned long long bal = __ballot(waits);
            if (lane == 0) s_cntw[wv] = __popcll(bal);
            __syncthreads();
            int off = 0, run = 0;
#pragma unroll
            for (int w = 0; w < RW_NW; ++w) {
                const int cw = s_cntw[w];
                off += w < wv ? cw : 0;
                run += cw;
            }
            if (waits) {
                const int slot = off + __popcll(bal & ((1ull << lane) - 1ull));
                if (slot < RW_WLIST) s_wlist[slot] = (unsigned short)tid;
            }
            if (tid == 0) s_first = run;
            __syncthreads();
        }
        int nw = min(s_first, RW_WLIST);
#if defined(AFV_RESOLVE_STATS) && AFV_RESOLVE_STATS == 4
        ++wg_conv;
        const long long wg_r0 = wall_clock64();
#endif
        if (nw == 0 && nlive > RWT) {  // sets above 1024 live rows: the rows behind the first 1024, one at a time (rare, slow, exact)
            if (tid == 0) s_first = RW_INF;
            __syncthreads();
            int mine = RW_INF;
            for (int li = RWT + tid; li < nlive; li += RWT)
                if ((s_flag[li] & 3) == 1) mine = min(mine, li);
            if (mine != RW_INF) atomicMin(&s_first, mine);
            __syncthreads();
            if (s_first != RW_INF) {
                if (tid == 0) s_wlist[0] = (unsigned short)s_first;
                nw = 1;
            }
            __syncthreads();
        }
        if (nw == 0) break;
        if (stage_cols && !cols_ready) {  // first rescan of this pair: park the column descriptors in LDS
            const uint4 *gc = reinterpret_cast<const uint4 *>(d2);
            uint4 *sc = reinterpret_cast<uint4 *>(s_cols);
            for (int i = tid; i < n2 * 2; i += RWT) sc[i] = gc[i];
            cols_ready = true;
            __syncthreads();
        }
        const uint4 *cb = (stage_cols && cols_ready) ? reinterpret_cast<const uint4 *>(s_cols) : reinterpret_cast<const uint4 *>(d2);
        // Exact rescans, RW_RPW waiting rows per wavefront and step (16 at a time), each over the columns no EARLIER row claims; the
        // rows of a wavefront share the column loads.  A row's answer is final when every row before it is final: the rows before the
        // first waiting row are (converged, none of them waits), and a rescan that ends in "no match" - nearly all do: 0-2 of the 6-30
        // of a pair of consecutive frames take a column - changes nothing for anybody behind it.  So the answers are adopted in row
        // order up to and including the first one that TAKES a column; the rows behind that one are rescanned again after the
        // iteration has settled with the new claim.  (Four rows per step, one per wavefront, left the slowest of 128 concurrent pairs
        // at 8 steps of two barriers each: that tail, not the median, is what a batch waits for.)
        bool took = false;
        for (int g0 = 0; g0 < nw && !took; g0 += RW_RPW * RW_NW) {
            int rr[RW_RPW], kk[RW_RPW], s2[RW_RPW], wr[RW_RPW];
            uint32_t q[RW_RPW][8];
            const int gb = g0 + wv * RW_RPW;
#pragma unroll
            for (int j = 0; j < RW_RPW; ++j) {
                rr[j] = s_wlist[min(gb + j, nw - 1)];
                const uint4 *qp = reinterpret_cast<const uint4 *>(d1 + (size_t)s_live[rr[j]] * 8);
                const uint4 qlo = qp[0], qhi = qp[1];
                q[j][0] = qlo.x, q[j][1] = qlo.y, q[j][2] = qlo.z, q[j][3] = qlo.w, q[j][4] = qhi.x, q[j][5] = qhi.y, q[j][6] = qhi.z, q[j][7] = qhi.w;
                kk[j] = NO_KEY;
                s2[j] = NO_KEY >> 16;
                wr[j] = -1;
            }
            if (gb < nw) {
                for (int c0 = lane; c0 < n2; c0 += 128) {
                    uint4 lo[2], hi[2];
                    int cl[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = min(c0 + 64 * u, n2 - 1);
                        lo[u] = cb[2 * c];
                        hi[u] = cb[2 * c + 1];
                        cl[u] = W[c];
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = c0 + 64 * u;
#pragma unroll
                        for (int j = 0; j < RW_RPW; ++j) {
                            const int d = __popc(q[j][0] ^ lo[u].x) + __popc(q[j][1] ^ lo[u].y) + __popc(q[j][2] ^ lo[u].z) + __popc(q[j][3] ^ lo[u].w) +
                                          __popc(q[j][4] ^ hi[u].x) + __popc(q[j][5] ^ hi[u].y) + __popc(q[j][6] ^ hi[u].z) + __popc(q[j][7] ^ hi[u].w);
                            const bool usable = c < n2 && !(cl[u] < rr[j]);
                            const int key = usable ? ((d << 16) | c) : NO_KEY;
                            s2[j] = min(s2[j], max(kk[j], key) >> 16);
                            kk[j] = min(kk[j], key);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < RW_RPW; ++j) {
                    wave_merge_best(kk[j], s2[j]);
                    if (kk[j] != NO_KEY) {
                        const float best1 = (float)(kk[j] >> 16);
                        const float best2 = (s2[j] == (NO_KEY >> 16)) ? 3.402823466e+38f : (float)s2[j];
                        if (best1 < th && best1 < ratio * best2) wr[j] = kk[j] & 0xffff;
                    }
                }
            }
            if (lane < RW_RPW) {
                int v = wr[0];
#pragma unroll
                for (int j = 1; j < RW_RPW; ++j) v = lane == j ? wr[j] : v;
                s_part[wv * RW_RPW + lane] = v;
            }
            __syncthreads();
            // adopt in row order (every thread computes the same verdict)
            int nadopt = 0;
#pragma unroll
            for (int w = 0; w < RW_RPW * RW_NW; ++w) {
                if (g0 + w < nw && !took) {
                    ++nadopt;
                    took = s_part[w] >= 0;
                }
            }
#if defined(AFV_RESOLVE_STATS) && AFV_RESOLVE_STATS == 4
            ++wg_steps; wg_resc += nadopt; wg_took += took ? 1 : 0;
#endif
            if (tid < nadopt) {
                const int r = s_wlist[g0 + tid];
                s_flag[r] = 2;  // pinned: from now on the row asserts this answer in every pass (its want so far was -1)
                s_out[s_live[r]] = s_part[tid];
            }
            __syncthreads();
        }
        // a taken column enters the claims with the next pass (the pinned row's want changes from -1), the rows behind it re-evaluate;
        // if nothing was taken and every waiting row was looked at, the converged state is the final one
#if defined(AFV_RESOLVE_STATS) && AFV_RESOLVE_STATS == 4
        wg_tresc += wall_clock64() - wg_r0;
#endif
        if (!took && s_first <= RW_WLIST && nlive <= RWT) break;
    }
#if defined(AFV_RESOLVE_STATS) && AFV_RESOLVE_STATS == 4
    if (tid == 0) printf("wgpair %d nlive %d passes %d convergences %d rescan_steps %d rescans %d took %d fixedpoint_x10ns %lld rescan_x10ns %lld\n", p, nlive, pass, wg_conv, wg_steps, wg_resc, wg_took, wall_clock64() - wg_t0, wg_tresc);
#endif
    // ---- matches, count ----
    {
        int cnt = 0;
        for (int li = tid; li < nlive; li += RWT) {
            const int w = s_w1[li];
            if (w >= 0) {
                s_out[s_live[li]] = w;
                ++cnt;
            }
        }
        if (tid == 0) s_nm = 0;
        __syncthreads();
        cnt = afv_wave_incl_scan(cnt);
        if (lane == 63 && cnt) atomicAdd(&s_nm, cnt);
        __syncthreads();
    }
    if (check_ori) {
        // rotation histogram of the accepted matches (FeatureMatcher.cc:1587-1599): it never influences the walk
        for (int i = tid; i < n1; i += RWT) {
            const int c = s_out[i];
            if (c >= 0) {
                const int bin = rotation_bin(ang[((size_t)a * cap + i) * ang_stride], ang[((size_t)b * cap + c) * ang_stride]);
                s_bin[i] = (uint8_t)bin;
                atomicAdd(&s_hist[bin], 1);
            }
        }
        __syncthreads();
        if (tid == 0) {  // computeThreeMaxima (FeatureMatcher.cc:1631-1668)
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
        for (int i = tid; i < n1; i += RWT) {
            if (s_out[i] >= 0) {
                const int bb = s_bin[i];
                if (bb != i1 && bb != i2 && bb != i3) { s_out[i] = -1; ++dropped; }
            }
        }
        if (dropped) atomicSub(&s_nm, dropped);
        __syncthreads();
    }
    for (int i = tid; i < cap; i += RWT) out[i] = s_out[i];
    if (tid == 0) nmatches[p] = guard_hit ? AFV_RESOLVE_GUARD : s_nm;
}

// ---------------- M4: SearchForTriangulation ----------------

// One thread per KF1 feature (row_seg = the shared node it belongs to, -1 if none): rows are independent here (vbMatched2 is
// never set, FeatureMatcher.cc:681,724), so the whole job runs in one pass; the threads of a wave mostly sit in the same node
// and read the same KF2 candidates (uniform loads).  Equal distances: the LAST candidate in the node's order wins (:736).
template <int W>
__device__ int tri_row(const DevTriJob &T, int idx1) {
    const DevMatchJob &J = T.m;
    const int sg = T.row_seg[idx1];
    if (sg < 0) return -1;
    if (J.valid1 && J.valid1[idx1]) return -1;  // already has a MapPoint (:699-703)
    const bool stereo1 = T.u_right1 && T.u_right1[idx1] >= 0.0f;  // :705
    if (T.only_stereo && !stereo1) return -1;                      // :707-709
    const Seg S = J.segs[sg];
    uint32_t q[W == 0 ? 1 : W];
    if constexpr (W != 0) {
#pragma unroll
        for (int i = 0; i < W; ++i) q[i] = J.d1[(size_t)idx1 * W + i];
    }
    const float kx = T.x1[idx1], ky = T.y1[idx1];
    // epipolar line in image 2: l = x1' F12 (:168-170)
    const float la = kx * T.F[0] + ky * T.F[3] + T.F[6];
    const float lb = kx * T.F[1] + ky * T.F[4] + T.F[7];
    const float lc = kx * T.F[2] + ky * T.F[5] + T.F[8];
    const float den = la * la + lb * lb;
    if (den == 0) return -1;
    std::conditional_t<W == 0, float, int> best;  // bestDist starts at TH_LOW (:714): the test below against J.th says the same
    if constexpr (W == 0) best = 3.402823466e+38f;
    else best = 0x7fffffff;
    int best_idx = -1;
    for (int b = 0; b < S.n2; ++b) {
        const int idx2 = J.idx2 ? J.idx2[S.s2 + b] : S.s2 + b;
        if (J.valid2 && J.valid2[idx2]) continue;
        const bool stereo2 = T.u_right2 && T.u_right2[idx2] >= 0.0f;  // :727
        if (T.only_stereo && !stereo2) continue;                      // :729-731
        decltype(best) d;
        if constexpr (W == 0) {
            (void)q;
            d = l2sqr(frow(J.d1, idx1, J.fdim), frow(J.d2, idx2, J.fdim), J.fdim);
        } else {
            d = hamming_words<W>(q, J.d2 + (size_t)idx2 * W);
        }
        if ((float)d > J.th || d > best) continue;
        const float x2 = T.x2[idx2], y2 = T.y2[idx2], sg2 = T.sigma2_2[idx2];
        if (!stereo1 && !stereo2) {
            const float dex = T.ex - x2, dey = T.ey - y2;
            if (dex * dex + dey * dey < 100.0f * sqrtf(sg2)) continue;  // too close to the epipole (:741-748)
        }
        const float num = la * x2 + lb * y2 + lc;
        const float dsqr = num * num / den;
        if (!(dsqr < 3.84f * sg2)) continue;  // CheckDistEpipolarLine (:172-181)
        best = d;
        best_idx = idx2;
    }
    return best_idx;
}

__global__ __launch_bounds__(MT) void k_match_tri(const DevTriJob *__restrict__ jobs) {
    const DevTriJob &T = jobs[blockIdx.y];
    const int i = blockIdx.x * MT + threadIdx.x;
    int r = -1;
    if (i < T.m.n1) {
        r = T.m.fdim ? tri_row<0>(T, i) : (T.m.words == 8 ? tri_row<8>(T, i) : tri_row<16>(T, i));
        T.m.out[i] = r;
    }
    const int cnt = __popcll(__ballot(r >= 0));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(T.m.nmatches, cnt);  // the counter arrives zeroed with the staging blob
}

// ---------------- M8: float descriptors, L2^2 (cv::norm NORM_L2SQR semantics): l2sqr() above ----------------
__device__ __forceinline__ void merge_best_f(float &d, int &p, float &s, float d2, int p2, float s2) {
    if (d2 < d || (d2 == d && p2 < p)) {
        s = fminf(s2, d);
        d = d2;
        p = p2;
    } else {
        s = fminf(s, d2);
    }
}

__global__ __launch_bounds__(MT) void k_match_l2(const float *__restrict__ d1, int n1, const float *__restrict__ d2, int n2,
                                                 int dim, const uint8_t *__restrict__ valid1,
                                                 const uint8_t *__restrict__ valid2, float th, float ratio,
                                                 int *__restrict__ out, int *__restrict__ nmatches) {
    __shared__ uint32_t s_matched[MAX_SIDE / 32];
    __shared__ float s_d[4], s_s[4];
    __shared__ int s_p[4], s_n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float FMAX = 3.402823466e+38f;
    for (int i = tid; i < (n2 + 31) / 32; i += MT) s_matched[i] = 0;
    if (tid == 0) s_n = 0;
    __syncthreads();
    for (int i = 0; i < n1; ++i) {
        if (tid == 0) out[i] = -1;
        if (valid1 && !valid1[i]) continue;
        float bd = FMAX, bs = FMAX;
        int bp = 0x7fffffff;
        for (int k = tid; k < n2; k += MT) {
            if ((s_matched[k >> 5] >> (k & 31)) & 1u) continue;
            if (valid2 && !valid2[k]) continue;
            const float d = l2sqr(d1 + (size_t)i * dim, d2 + (size_t)k * dim, dim);
            merge_best_f(bd, bp, bs, d, k, FMAX);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float d2_ = __shfl_xor(bd, m, 64), s2_ = __shfl_xor(bs, m, 64);
            const int p2_ = __shfl_xor(bp, m, 64);
            merge_best_f(bd, bp, bs, d2_, p2_, s2_);
        }
        if (lane == 0) { s_d[wv] = bd; s_p[wv] = bp; s_s[wv] = bs; }
        __syncthreads();
        if (tid == 0) {
            float D = s_d[0], Sx = s_s[0];
            int P = s_p[0];
            for (int w = 1; w < 4; ++w) merge_best_f(D, P, Sx, s_d[w], s_p[w], s_s[w]);
            if (P != 0x7fffffff && D < th && D < ratio * Sx) {
                out[i] = P;
                s_matched[P >> 5] |= 1u << (P & 31);
                s_n++;
            }
        }
        __syncthreads();
    }
    if (tid == 0) *nmatches = s_n;
}

extern "C" void afv_launch_match_bow(const DevMatchJob *jobs, int njobs, hipStream_t stream) {
    hipLaunchKernelGGL(k_match_bow, dim3(njobs), dim3(MT), 0, stream, jobs);
}
extern "C" void afv_launch_match_bow_seg(const DevMatchJob *jobs, int njobs, const void *tasks, int ntasks, int *hist, uint8_t *bins,
                                         const int *bin_off, int any_ori, hipStream_t stream) {
    hipLaunchKernelGGL(k_match_bow_seg, dim3((ntasks + MT / 64 - 1) / (MT / 64)), dim3(MT), 0, stream, jobs,
                       reinterpret_cast<const SegTask *>(tasks), ntasks, hist, bins, bin_off);
    if (any_ori) hipLaunchKernelGGL(k_match_bow_finish, dim3(njobs), dim3(MT), 0, stream, jobs, hist, bins, bin_off);
}
// phase 1 (VALU-bound: the 8 xor + 8 v_bcnt per descriptor pair) and phase 2 (latency-bound ordered resolve) are launched
// separately so that the runtime can time them apart.
extern "C" void afv_launch_match_topk_mfma(const uint8_t *desc, const int *nset, int cap, const int *pa, const int *pb, int npairs,
                                           void *topk_scratch, int pair_base, int nslices, void *slice_scratch, void *tickets, hipStream_t stream);
// engine: AFV_MATCH_ENGINE_MFMA (default: the exact i8 contraction of k_match_mfma.hip; its keys hold the column in 13 bits) or
// AFV_MATCH_ENGINE_POPCOUNT (k_match_topk below).  Both write the same top-4 keys.
// Column slices of phase 1 (the small-batch path; only the matrix-core engine deals its column tiles out).  A sliced launch needs, beside
// the key records (2 x int4 per row), `slice_scratch` = 2 x int4 per row AND slice, and `tickets` = one int per pair and row tile of
// 256 rows, ZERO before the first launch (the kernel re-arms them), both indexed by the global pair number like the records.
extern "C" int afv_match_topk_slices(int cap, int engine, int want) {
    return (engine == AFV_MATCH_ENGINE_MFMA && cap < 8192) ? std::max(1, std::min(want, 16)) : 1;
}
extern "C" void afv_launch_match_topk(const uint8_t *desc, const int *nset, int cap, const int *pa, const int *pb, int npairs,
                                      void *topk_scratch, int pair_base, int engine, int nslices, void *slice_scratch, int *tickets,
                                      hipStream_t stream) {
    if (engine == AFV_MATCH_ENGINE_MFMA && cap < 8192) {
        afv_launch_match_topk_mfma(desc, nset, cap, pa, pb, npairs, topk_scratch, pair_base, nslices, slice_scratch, tickets, stream);
        return;
    }
    int4 *topk = reinterpret_cast<int4 *>(topk_scratch);
    hipLaunchKernelGGL(k_match_topk, dim3((cap + MT - 1) / MT, npairs), dim3(MT), 0, stream, desc, nset, cap, pa, pb, topk, pair_base);
}
// ang: keypoint angles in degrees, element (set, i) at ang[(set * cap + i) * ang_stride] (stride 7 = afv_keypoint::angle)
// test hook (process-wide, 0 = off): caps the passes of the fixed-point engines (k_match_resolve_wg, k_proj_resolve_wg, k_init_resolve_wg) so
// that a test can drive them into their guard and check that the call reports it instead of returning a half-settled assignment
// ---------------- MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:279-349), batched over map points ----------------
// Per map point: the N x N Hamming distances of its N observed descriptors, per row the median = the floor(0.5 (N - 1))-th smallest
// entry (the row sorted, MapPoint.cc:329-331; the self-distance 0 is part of the row), the row with the least median wins, first one on
// ties (strict <, :333).  One wavefront per map point, lane = row (rows in chunks of 64); nothing is sorted or stored: the m-th smallest of a
// row of integers in [0, 8 * bytes] is found by bisection on the VALUE - count(d < mid) against m - recomputing the row's distances in each
// of the <= 10 steps (N = 20 observations: 10 x 20 x 8 xor + popcount per lane; the other rows' descriptors are the same address in all lanes).
template <int W>
__global__ __launch_bounds__(256) void k_distinctive(const uint32_t *__restrict__ desc, const int *__restrict__ set_ptr, int nsets, int max_dist,
                                                     int *__restrict__ best_idx, int *__restrict__ best_median) {
    const int s = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= nsets) return;  // wave-uniform
    const int b = set_ptr[s], n = set_ptr[s + 1] - b;
    if (n <= 0) {
        if (lane == 0) {
            best_idx[s] = -1;
            best_median[s] = 0;
        }
        return;
    }
    const int m = (n - 1) >> 1;  // vDists[0.5 * (N - 1)]
    unsigned best = 0xffffffffu;  // median << 16 | row
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        uint32_t q[W];
#pragma unroll
        for (int w = 0; w < W; ++w) q[w] = desc[(size_t)(b + min(i, n - 1)) * W + w];
        // smallest v with count(d <= v) > m  <=>  the m-th smallest (0-based) value
        int lo = 0, hi = max_dist;  // answer in [lo, hi]
        while (__any(lo < hi)) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;  // #{j : d(i, j) <= mid}
            for (int j = 0; j < n; ++j) {
                const uint32_t *r = desc + (size_t)(b + j) * W;
                int d = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) d += __popc(q[w] ^ r[w]);
                cnt += d <= mid;
            }
            if (lo < hi) {
                if (cnt > m) hi = mid;
                else lo = mid + 1;
            }
        }
        if (i < n) best = min(best, ((unsigned)lo << 16) | (unsigned)i);
    }
    // wave minimum of (median, row): the least median, the first row on ties
    best = min(best, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)best, 0x111, 0xf, 0xf, false));
    best = min(best, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)best, 0x112, 0xf, 0xf, false));
    best = min(best, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)best, 0x114, 0xf, 0xf, false));
    best = min(best, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)best, 0x118, 0xf, 0xf, false));
    best = min(best, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)best, 0x142, 0xa, 0xf, false));
    best = min(best, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)best, 0x143, 0xc, 0xf, false));
    const unsigned g = (unsigned)__builtin_amdgcn_readlane((int)best, 63);
    if (lane == 0) {
        best_idx[s] = (int)(g & 0xffffu);
        best_median[s] = (int)(g >> 16);
    }
}

// The same for FLOAT descriptors (DescriptorDistance = L2^2 as a float, FeatureMatcher.cc:1508-1531): the m-th smallest of a row is found by
// bisection on the BIT PATTERN of the distances (non-negative floats order like their bits: <= 31 steps).  A map point with at most 64
// observations keeps its rows' distances in LDS (row i = lane, column-major: conflict-free) and pays the 4 x dim double operations per pair
// once; a larger one recomputes them in every step.
#define DF_CACHE 64
__global__ __launch_bounds__(256) void k_distinctive_f32(const float *__restrict__ desc, int dim, const int *__restrict__ set_ptr, int nsets,
                                                         int *__restrict__ best_idx, float *__restrict__ best_median) {
    __shared__ unsigned s_d[4][DF_CACHE][64];  // [wavefront][j][lane = row i]
    const int wv = (int)threadIdx.x >> 6;
    const int s = (int)blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (s >= nsets) return;  // wave-uniform
    const int b = set_ptr[s], n = set_ptr[s + 1] - b;
    if (n <= 0) {
        if (lane == 0) {
            best_idx[s] = -1;
            best_median[s] = 0.0f;
        }
        return;
    }
    const int m = (n - 1) >> 1;  // vDists[0.5 * (N - 1)]
    const bool cached = n <= DF_CACHE;  // wave-uniform
    unsigned best_med = 0xffffffffu, best_row = 0xffffffffu;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const float *q = desc + (size_t)(b + min(i, n - 1)) * dim;
        if (cached)
            for (int j = 0; j < n; ++j) s_d[wv][j][lane] = __float_as_uint(l2sqr(q, desc + (size_t)(b + j) * dim, dim));  // (the diagonal: 0)
        unsigned lo = 0u, hi = 0x7f800000u;  // smallest key v with #{j : key(d(i, j)) <= v} > m  <=>  the m-th smallest (0-based) distance
        while (__any(lo < hi)) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            int cnt = 0;
            if (cached)
                for (int j = 0; j < n; ++j) cnt += s_d[wv][j][lane] <= mid;
            else
                for (int j = 0; j < n; ++j) cnt += __float_as_uint(l2sqr(q, desc + (size_t)(b + j) * dim, dim)) <= mid;
            if (lo < hi) {
                if (cnt > m) hi = mid;
                else lo = mid + 1u;
            }
        }
        if (i < n && lo < best_med) {  // strict: the first row keeps a tie (rows ascend with i0 inside a lane)
            best_med = lo;
            best_row = (unsigned)i;
        }
    }
    // the least median over the lanes, then the first row that has it
    auto wave_min = [](unsigned v) {
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x111, 0xf, 0xf, false));
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x112, 0xf, 0xf, false));
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x114, 0xf, 0xf, false));
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x118, 0xf, 0xf, false));
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x142, 0xa, 0xf, false));
        v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x143, 0xc, 0xf, false));
        return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    };
    const unsigned gm = wave_min(best_med);
    const unsigned gr = wave_min(best_med == gm ? best_row : 0xffffffffu);
    if (lane == 0) {
        best_idx[s] = (int)gr;
        best_median[s] = __uint_as_float(gm);
    }
}
extern "C" void afv_launch_distinctive_f32(const float *desc, int dim, const int *set_ptr, int nsets, int *best_idx, float *best_median, hipStream_t stream) {
    if (nsets <= 0) return;
    hipLaunchKernelGGL(k_distinctive_f32, dim3((nsets + 3) / 4), dim3(256), 0, stream, desc, dim, set_ptr, nsets, best_idx, best_median);
}

extern "C" void afv_launch_distinctive(const uint32_t *desc, const int *set_ptr, int nsets, int words, int desc_bytes, int *best_idx, int *best_median,
                                       hipStream_t stream) {
    if (nsets <= 0) return;
    const dim3 grid((nsets + 3) / 4);
    if (words == 8) hipLaunchKernelGGL(k_distinctive<8>, grid, dim3(256), 0, stream, desc, set_ptr, nsets, 8 * desc_bytes, best_idx, best_median);
    else hipLaunchKernelGGL(k_distinctive<16>, grid, dim3(256), 0, stream, desc, set_ptr, nsets, 8 * desc_bytes, best_idx, best_median);
}

extern "C" int afv_debug_pass_cap = 0;

// once per context (afv_create, on the context's device): both ordered-phase kernels may ask for more dynamic LDS than the default 64 KB
extern "C" int afv_match_prepare(void) {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_match_resolve_wg), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_match_resolve), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess && ok;
    if (!ok) (void)hipGetLastError();
    return ok ? 1 : 0;
}

extern "C" void afv_launch_match_resolve(const uint8_t *desc, const float *ang, int ang_stride, const int *nset, int cap, const int *pa,
                                         const int *pb, int npairs, float th, float ratio, int check_ori, int *match, int *nmatches,
                                         const void *topk_scratch, int pair_base, int engine, hipStream_t stream) {
    const int4 *topk = reinterpret_cast<const int4 *>(topk_scratch);
    if (engine == 1) {  // workgroup-wide fixed point (round 4); columns are parked in LDS only for batches, and only once a pair needs a rescan
        const bool stage = cap <= PAIR_COLS_LDS && npairs > 8;
        const size_t lds_wg = resolve_wg_lds_bytes(cap, stage);  // (the raised LDS limit: afv_match_prepare at afv_create)
        hipLaunchKernelGGL(k_match_resolve_wg, dim3(npairs), dim3(RWT), lds_wg, stream, desc, ang, ang_stride, nset, cap, pa, pb, topk, th, ratio,
                           check_ori, match, nmatches, pair_base, stage ? 1 : 0, afv_debug_pass_cap);
        return;
    }
    // the column descriptors ride in LDS (for the exact rescans) when they fit and the launch is a batch; a handful of pairs (the
    // single-frame plugin path) runs leaner: 39 KB instead of 71 KB, rescans through L2
    const bool stage_cols = cap <= PAIR_COLS_LDS && npairs > 8;
    const size_t lds = resolve_lds_bytes(cap, stage_cols);  // 71 KB with the columns: above the default 64 KB limit (afv_match_prepare)
    hipLaunchKernelGGL(k_match_resolve, dim3(npairs), dim3(MT), lds, stream, desc, ang, ang_stride, nset, cap, pa, pb, topk, th, ratio,
                       check_ori, match, nmatches, pair_base, stage_cols ? 1 : 0);
}
extern "C" void afv_launch_match_tri(const DevTriJob *jobs, int njobs, int max_n1, hipStream_t stream) {
    if (max_n1 > 0) hipLaunchKernelGGL(k_match_tri, dim3((max_n1 + MT - 1) / MT, njobs), dim3(MT), 0, stream, jobs);
}
extern "C" void afv_launch_match_l2(const float *d1, int n1, const float *d2, int n2, int dim, const uint8_t *v1,
                                    const uint8_t *v2, float th, float ratio, int *out, int *nmatches, hipStream_t stream) {
    hipLaunchKernelGGL(k_match_l2, dim3(1), dim3(MT), 0, stream, d1, n1, d2, n2, dim, v1, v2, th, ratio, out, nmatches);
}
