// k_select.hip — E4b + E7: retainBest on the Harris response and DistributeOctTree, one workgroup per (frame, level).
//
// Replaces KeyPointsFilter::retainBest(quota) on the Harris response (inside cv::ORB::detect, Feature_orb32.cpp:34; the
// retainBest(2*quota) on the FAST score before it is k_harris.hip's k_retain_score) and FeatureExtractor::DistributeOctTree
// (ORBextractor.cc:239-458, called through filterKeypoints_notScaled FeatureExtractor.cpp:276-284).
//
// Everything here is a SET operation plus a deterministic sequential algorithm, re-expressed so that 256 lanes can
// run it in lock step:
//   * retainBest = "keep everything >= the k-th largest key": k-th largest by a 3-pass 11/11/10-bit radix select on the
//     order-preserving integer image of the float response.
//   * DistributeOctTree is a level-synchronous quadtree build.  std::list order is fully determined by creation
//     order (children are always push_front'ed, nodes never move), so the list is kept as a dense array in list
//     order and every round recomputes it with prefix sums:
//       phase A (ORBextractor.cc:292-367): every node with >1 point splits; new list = children of the split nodes
//         in reverse processing order (n4,n3,n2,n1 each), then the untouched nodes in their old order;
//       phase B (:370-431): nodes created in the previous round are processed largest-first (ties: later-created
//         first == smaller list index) until the node count reaches N; the stop point is found with a prefix sum
//         over the rank-ordered size deltas.
//     Termination tests (:366-369, :427-430) are evaluated on the same quantities as the reference.
//   * the survivor of each node is the max-response point, ties -> smallest raster index (== "first max wins" on
//     raster-ordered input, :446-453), found with one 64-bit LDS atomic max per point.
#include <algorithm>

#include "afv_device.h"
#include "afv_runtime.h"  // the launchers below are declared there: a signature that drifts is a compile error, not a silent ABI mismatch

// Two instantiations: ST = 256 threads per workgroup for batches (what a batch costs is set by how many (frame, level) workgroups a CU
// holds: 29 KB of LDS, <= 96 VGPRs), ST = 1024 for the small-batch path (one frame: the eight workgroups of a frame are alone on the
// chip and the kernel's time is the critical path of level 0 - four times the lanes shorten every per-thread loop: 5 instead of 20
// candidates, 3 instead of 9 survivors per thread).
#define SEL_CAND 5120  // candidates kept in registers: the fast path covers n <= ST * CPT = 5120
template <int ST> struct SelCfg {
    static constexpr int CPT = SEL_CAND / ST;             // candidates a thread keeps in registers
    static constexpr int PPT = ST == 256 ? 9 : 3;         // retainBest survivors a thread keeps in registers over the quadtree rounds
    static constexpr int KEPT_LDS = ST * PPT;             // ... and how many of them (position + node label) LDS holds (else global scratch)
    static constexpr int NW = ST / 64;
};
#define SEL_TMP 32    // ints of small shared state: [0..15] counters / results, [16..31] per-wavefront partial sums
// The kernel is a chain of short dependent phases (latency-, not throughput-bound): what a batch costs is set by how many
// (frame, level) workgroups a CU holds at once, so the LDS footprint (~29 KB at M = 256 -> 5 workgroups per CU) and the
// register budget (<= 96 VGPRs, 5 waves per SIMD) are the tuning parameters here.

// quadtree region of the LDS layout (see the kernel); never smaller than the 2048-bin histogram that shares it
// (the 1024-thread instantiation finds the retainBest threshold with ONE 4096-bin pass + an exact ranking of the threshold bin's keys: its
// histogram (16 KB) and key list (SEL_EXACT u32) share the region)
#define SEL_EXACT 1024  // keys of the threshold bin the exact ranking takes (more: the three radix passes)
__host__ __device__ constexpr size_t afv_select_tree_bytes(int M, bool wide = false) {
    const size_t tree = (size_t)M * 60, need = wide ? (size_t)4096 * 4 + SEL_EXACT * 4 : (size_t)8192;
    return tree > need ? tree : need;
}

struct Rect16 {
    short x0, y0, x1, y1;
};

__device__ __forceinline__ uint32_t float_key(float r) {
    const uint32_t u = __float_as_uint(r);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ---- block-wide helpers (256 threads = 4 waves) ----
__device__ __forceinline__ int wave_incl_scan(int v) { return afv_wave_incl_scan(v); }

// exclusive prefix sum of arr[0..n) in place; returns the total.  n <= ST * 16.  `tmp` = SEL_TMP ints of LDS.
template <int ST>
__device__ int block_excl_scan(int *arr, int n, int *tmp) {
    int *ws = tmp + 16;
    const int per = (n + ST - 1) / ST;
    const int b = threadIdx.x * per, e = min(b + per, n);
    int local = 0;
    for (int i = b; i < e; ++i) local += arr[i];
    const int incl = wave_incl_scan(local);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) ws[w] = incl;
    __syncthreads();
    int base = incl - local, total = 0;
#pragma unroll
    for (int k = 0; k < ST / 64; ++k) {
        const int v = ws[k];
        base += k < w ? v : 0;
        total += v;
    }
    for (int i = b; i < e; ++i) {
        const int v = arr[i];
        arr[i] = base;
        base += v;
    }
    __syncthreads();
    return total;
}

// two exclusive prefix sums in ONE pass: a[0..na) and u[0..nu) (na <= nu), both with totals < 65536 so that the pair rides one packed
// wave scan.  Returns (total of a) | (total of u) << 16.  `tmp` = 4 ints of LDS.  One barrier inside, one at the end.
template <int ST>
__device__ int block_excl_scan2(int *a, int na, int *u, int nu, int *tmp) {
    int *ws = tmp + 16;
    const int per = (nu + ST - 1) / ST;
    const int b = threadIdx.x * per, e = min(b + per, nu), ea = min(e, na);
    int local = 0;
    for (int i = b; i < ea; ++i) local += a[i];
    for (int i = b; i < e; ++i) local += u[i] << 16;
    const int incl = wave_incl_scan(local);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) ws[w] = incl;
    __syncthreads();
    int base = incl - local, total = 0;
#pragma unroll
    for (int k = 0; k < ST / 64; ++k) {
        const int v = ws[k];
        base += k < w ? v : 0;
        total += v;
    }
    int ba = base & 0xffff, bu = (int)((unsigned)base >> 16);
    for (int i = b; i < ea; ++i) {
        const int v = a[i];
        a[i] = ba;
        ba += v;
    }
    for (int i = b; i < e; ++i) {
        const int v = u[i];
        u[i] = bu;
        bu += v;
    }
    __syncthreads();
    return total;
}

// k-th largest over a histogram hist[0..nbins): returns the bin b such that sum(hist[b+1..]) < k <= sum(hist[b..]),
// and *above = sum(hist[b+1..]).  hist is destroyed.  Requires sum(hist) >= k >= 1.
template <int ST>
__device__ int block_kth_from_top(int *hist, int nbins, int k, int *above, int *tmp) {
    int *ws = tmp + 16;
    // suffix sums via a prefix scan of the reversed index space
    const int per = (nbins + ST - 1) / ST;
    const int b = threadIdx.x * per, e = min(b + per, nbins);
    int local = 0;
    for (int i = b; i < e; ++i) local += hist[nbins - 1 - i];
    const int incl = wave_incl_scan(local);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) ws[w] = incl;
    if (threadIdx.x == 0) tmp[4] = -1;
    __syncthreads();
    int base = incl - local;
#pragma unroll
    for (int q = 0; q < ST / 64; ++q) base += q < w ? ws[q] : 0;
    // this thread's chunk (in reversed order) covers cumulative counts (base, base+local]
    if (base < k && k <= base + local) {
        int cum = base;
        for (int i = b; i < e; ++i) {
            const int h = hist[nbins - 1 - i];
            if (cum < k && k <= cum + h) {
                tmp[4] = nbins - 1 - i;
                tmp[5] = cum;
            }
            cum += h;
        }
    }
    __syncthreads();
    const int bin = tmp[4];
    *above = tmp[5];
    __syncthreads();
    return bin;
}

__device__ __forceinline__ int point_quadrant(uint32_t xy, float scale, const Rect16 r) {
    // ExtractorNode::DivideNode (ORBextractor.cc:181-224): halfX = ceil((UR.x-UL.x)/2), float compares
    const float px = (float)(xy & 4095u) * scale, py = (float)(xy >> 12) * scale;
    const int hx = (r.x1 - r.x0 + 1) >> 1, hy = (r.y1 - r.y0 + 1) >> 1;
    const float mx = (float)(r.x0 + hx), my = (float)(r.y0 + hy);
    // n1: x<mx,y<my   n2: x>=mx,y<my   n3: x<mx,y>=my   n4: x>=mx,y>=my
    return (px < mx ? 0 : 1) + (py < my ? 0 : 2);
}

__device__ __forceinline__ Rect16 child_rect(const Rect16 r, int q) {
    const int hx = (r.x1 - r.x0 + 1) >> 1, hy = (r.y1 - r.y0 + 1) >> 1;
    Rect16 c;
    c.x0 = (q & 1) ? (short)(r.x0 + hx) : r.x0;
    c.x1 = (q & 1) ? r.x1 : (short)(r.x0 + hx);
    c.y0 = (q & 2) ? (short)(r.y0 + hy) : r.y0;
    c.y1 = (q & 2) ? r.y1 : (short)(r.y0 + hy);
    return c;
}

// dynamic LDS layout, M = max nodes (multiple of 64):
//   quadtree region (60 * M bytes): Rect16 rect[2][M]; int cnt[2][M]; int child[M*4] (the survivor keys best[M] reuse it after the
//     last round); int aux[M], aux2[M], unt[M]; uint16 remap[M*4]
//   the retainBest histograms hist[2048] live on top of that region (dead before the first node is created)
//   int tmp[SEL_TMP]; kept_xy u32[KEPT_LDS]; kept_node u16[KEPT_LDS]        (the responses of the survivors stay in global scratch)
template <int ST>
__device__ __forceinline__ void select_quadtree_body(const Geo *__restrict__ geo_p, const uint32_t *__restrict__ cand_packed,
                                                       const float *__restrict__ cand_resp,
                                                       const int *__restrict__ cand_count, uint32_t *__restrict__ kept_xy,
                                                       float *__restrict__ kept_resp, uint16_t *__restrict__ kept_node,
                                                       SelPoint *__restrict__ sel, int *__restrict__ sel_count, int M, int frame_base,
                                                       int total_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CPT = SelCfg<ST>::CPT, KEPT_LDS = SelCfg<ST>::KEPT_LDS;
    int *hist = reinterpret_cast<int *>(smem);
    Rect16 *rect0 = reinterpret_cast<Rect16 *>(smem);
    Rect16 *rect1 = rect0 + M;
    int *cnt0 = reinterpret_cast<int *>(rect1 + M);
    int *cnt1 = cnt0 + M;
    int *child = cnt1 + M;
    unsigned long long *best = reinterpret_cast<unsigned long long *>(child);  // 8-byte aligned: 24 * M bytes in
    int *aux = child + 4 * M;
    int *aux2 = aux + M;
    int *unt = aux2 + M;
    uint16_t *remap = reinterpret_cast<uint16_t *>(unt + M);
    int *tmp = reinterpret_cast<int *>(smem + afv_select_tree_bytes(M, ST == 1024));

#ifdef AFV_SELECT_STATS
    const long long st0 = wall_clock64();
    long long st1 = 0, st2 = 0, st3 = 0;
    int st_rounds = 0, st_b = 0;
    long long st_r[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_occ = 0;
#endif
    const Geo &geo = *geo_p;
    // XCD-aware placement: the 8 level-workgroups of a frame run on one XCD (they read what that frame's FAST tiles wrote)
    const int work = afv_xcd_remap(blockIdx.x, total_blocks);
    if (work >= total_blocks) return;
    const int l = work % geo.nlevels, f = frame_base + work / geo.nlevels;
    const LevelGeo &L = geo.lv[l];
    const size_t base = L.cand_off + (size_t)f * L.cand_frame_stride;
    const uint32_t *cp = cand_packed + base;
    const float *cr = cand_resp + base;
    uint32_t *kxy = kept_xy + base;  // re-pointed to LDS below when the survivors fit
    float *kr = kept_resp + base;
    uint16_t *kn = kept_node + base;
    uint32_t *lds_kxy = reinterpret_cast<uint32_t *>(tmp + SEL_TMP);
    uint16_t *lds_kn = reinterpret_cast<uint16_t *>(lds_kxy + KEPT_LDS);
    const int n = min(cand_count[f * AFV_MAX_LEVELS + l], L.cand_cap);
    const int tid = threadIdx.x, lane = tid & 63;
    const float scale = L.scale;
    const int N = L.quota;

    // Fast path: every thread keeps its share of the candidate list (items tid, tid+256, ...) in registers, so the
    // histogram / radix / count / compaction passes below never go back to memory.  Lists longer than ST*CPT (only
    // noise-like images at level 0) stream from global memory instead.
    const bool reg = n <= ST * CPT;
    uint32_t cpk[CPT];
    float crs[CPT];
    if (reg) {
#pragma unroll
        for (int k_ = 0; k_ < CPT; ++k_) {
            const int i = k_ * ST + tid;
            cpk[k_] = 0;
            crs[k_] = 0.f;
            if (k_ * ST < n && i < n) {
                cpk[k_] = cp[i];
                crs[k_] = cr[i];
            }
        }
    }
#define FOR_CAND(BODY)                                          \
    if (reg) {                                                  \
        _Pragma("unroll") for (int k_ = 0; k_ < CPT; ++k_) {    \
            if (k_ * ST >= n) break;                            \
            if (k_ * ST + tid < n) {                            \
                const float r = crs[k_];                        \
                BODY                                            \
            }                                                   \
        }                                                       \
    } else {                                                    \
        for (int i_ = tid; i_ < n; i_ += ST) {                  \
            const float r = cr[i_];                             \
            BODY                                                \
        }                                                       \
    }

    // ---------------- E4b: threshold on the Harris response (retainBest on the score already happened: k_harris.hip) ----------------
    const int n1 = n;
    uint32_t T2 = 0;
    int known_survivors = -1;  // how many candidates pass T2, when the selection below already knows (saves the counting pass)
    bool have_T2 = false;
    if (ST == 1024 && n1 > L.cv_quota) {  // uniform
        // The one-frame instantiation (round 6): what this kernel costs is its chain of workgroup-wide phases (16 wavefronts: ~0.8 us per
        // barrier and dependent LDS round trip), and the three radix passes were eighteen of them.  ONE pass over the top 12 key bits (sign,
        // exponent, 3 mantissa bits: 4096 bins) finds the bin that holds the k-th largest key; that bin's keys (a few dozen of some thousand
        // responses) are listed and ranked exactly - a key x is the k-th largest iff  #{y > x} < k <= #{y >= x}  among them - and the number
        // of survivors (keys >= T2, ties kept: KeyPointsFilter::retainBest) falls out of the same counts.  A bin with more than SEL_EXACT
        // keys (flat images: thousands of equal responses) takes the radix passes below.
        const int k = L.cv_quota;
        uint32_t *klist = reinterpret_cast<uint32_t *>(hist + 4096);
        for (int i = tid; i < 4096; i += ST) hist[i] = 0;
        if (tid == 0) tmp[6] = 0;
        __syncthreads();
        FOR_CAND({ atomicAdd(&hist[float_key(r) >> 20], 1); })
        __syncthreads();
        int above;
        const int bin = block_kth_from_top<ST>(hist, 4096, k, &above, tmp);
        const int in_bin = hist[bin], kk = k - above;  // the kk-th largest key of the bin is the threshold
        if (in_bin <= SEL_EXACT) {  // uniform
            FOR_CAND({
                const uint32_t key = float_key(r);
                if ((int)(key >> 20) == bin) klist[atomicAdd(&tmp[6], 1)] = key;
            })
            __syncthreads();
            if (tid < in_bin) {
                const uint32_t x = klist[tid];
                int g = 0, ge = 0;
                for (int j = 0; j < in_bin; ++j) {  // broadcast reads: every lane the same word
                    const uint32_t y = klist[j];
                    g += y > x;
                    ge += y >= x;
                }
                if (g < kk && kk <= ge) {  // every key equal to the threshold says the same
                    tmp[7] = (int)x;
                    tmp[9] = above + ge;
                }
            }
            __syncthreads();
            T2 = (uint32_t)tmp[7];
            known_survivors = tmp[9];  // (tmp[7] / tmp[9] are next written behind the compaction's barrier)
            have_T2 = true;
        }
    }
    if (n1 > L.cv_quota && !have_T2) {  // uniform
        int k = L.cv_quota;
        uint32_t prefix = 0, mask = 0;
        const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
        for (int p = 0; p < 3; ++p) {
            const int nb = 1 << bits[p];
            const int sh = shifts[p];
            for (int i = tid; i < nb; i += ST) hist[i] = 0;
            __syncthreads();
            FOR_CAND({
                const uint32_t key = float_key(r);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> sh) & (nb - 1)], 1);
            })
            __syncthreads();
            int above;
            const int bin = block_kth_from_top<ST>(hist, nb, k, &above, tmp);
            k -= above;
            prefix |= (uint32_t)bin << sh;
            mask |= (uint32_t)(nb - 1) << sh;
        }
        T2 = prefix;
    }
#ifdef AFV_SELECT_STATS
    st1 = wall_clock64();
#endif
    // number of survivors decides where they live during the quadtree rounds
    int n_surv = known_survivors;
    if (n1 <= L.cv_quota) n_surv = n1;  // nothing is cut
    if (n_surv < 0) {  // uniform: the radix passes do not count
        if (tid == 0) tmp[9] = 0;
        __syncthreads();
        {
            int c = 0;
            FOR_CAND(c += (float_key(r) >= T2);)
            c = wave_incl_scan(c);
            if (lane == 63) atomicAdd(&tmp[9], c);
        }
        __syncthreads();
        n_surv = tmp[9];
        __syncthreads();
    }
    if (n_surv <= KEPT_LDS) {
        kxy = lds_kxy;
        kn = lds_kn;
    }

    // ---------------- compaction of the survivors + root assignment ----------------
    const int n_ini = geo.n_ini;
    if (tid < 16) aux[tid] = 0;  // points per root
    if (tid == 0) tmp[8] = 0;
    __syncthreads();
    // wave-aggregated, order-free compaction of one candidate per lane
    auto emit = [&](bool keep, uint32_t e, float r) {
        const unsigned long long m = __ballot(keep);
        int wbase = 0;
        if (lane == 0 && m) wbase = atomicAdd(&tmp[8], __popcll(m));
        wbase = __shfl(wbase, 0, 64);
        if (keep) {
            const int slot = wbase + __popcll(m & ((1ull << lane) - 1ull));
            const uint32_t xy = e & 0x00ffffffu;
            kxy[slot] = xy;
            kr[slot] = r;
            int root = 0;
            if (n_ini > 1) {
                root = (int)(((float)(xy & 4095u) * scale) / geo.h_x);  // vpIniNodes[kp.pt.x/hX]
                root = min(root, n_ini - 1);
                atomicAdd(&aux[root], 1);
            }
            kn[slot] = (uint16_t)root;
        }
    };
    if (reg) {
#pragma unroll
        for (int k_ = 0; k_ < CPT; ++k_) {
            if (k_ * ST >= n) break;
            const bool in = k_ * ST + tid < n;
            emit(in && (float_key(crs[k_]) >= T2), cpk[k_], crs[k_]);
        }
    } else {
        for (int i0 = 0; i0 < n; i0 += ST) {
            const int i = i0 + tid;
            uint32_t e = 0;
            float r = 0.f;
            if (i < n) {
                e = cp[i];
                r = cr[i];
            }
            emit(i < n && (float_key(r) >= T2), e, r);
        }
    }
    __syncthreads();
    const int m2 = tmp[8];
    __threadfence_block();

    // initial list (ORBextractor.cc:243-283): non-empty roots in order
    if (tid == 0) {
        int sz = 0;
        for (int i = 0; i < n_ini; ++i) {
            const int c = (n_ini > 1) ? aux[i] : m2;
            if (c > 0) {
                Rect16 r;
                r.x0 = (short)(int)(geo.h_x * (float)i);
                r.x1 = (short)(int)(geo.h_x * (float)(i + 1));
                r.y0 = 0;
                r.y1 = (short)geo.height;
                rect0[sz] = r;
                cnt0[sz] = c;
                aux2[i] = sz;
                ++sz;
            } else {
                aux2[i] = 0;
            }
        }
        tmp[9] = sz;
    }
    __syncthreads();
    if (n_ini > 1) {
        for (int p = tid; p < m2; p += ST) kn[p] = (uint16_t)aux2[kn[p]];
    }
    int size = tmp[9];
    // the fast-forward below counts points per grid cell at depths 1..ff_dmax under each root (S_d = roots * 4^d cells, S_dmax <= M):
    // counters by natural path index ((root * 4 + q1) * 4 + q2) * 4 + q3.  C2 sits in `aux` (the root counts in it were consumed
    // before the barrier above), C3 and C1 borrow `remap` (only used inside a round: 2 M ints)
    int ff_dmax = 0;
    for (int d = 1, S = size * 4; d <= 3 && S <= M && m2 > 0 && m2 <= SelCfg<ST>::KEPT_LDS; ++d, S *= 4) ff_dmax = d;
    const int ff_SM = size * (ff_dmax == 0 ? 1 : ff_dmax == 1 ? 4 : ff_dmax == 2 ? 16 : 64);
    int *C1 = reinterpret_cast<int *>(remap) + M, *C2 = aux, *C3 = reinterpret_cast<int *>(remap);
    if (ff_dmax >= 1) for (int i = tid; i < size * 4; i += ST) C1[i] = 0;
    if (ff_dmax >= 2) for (int i = tid; i < size * 16; i += ST) C2[i] = 0;
    if (ff_dmax >= 3) for (int i = tid; i < size * 64; i += ST) C3[i] = 0;
    // every round starts with child[0 .. 4*size) and its n_expand slot cleared (the previous round does it; this is round 0's - wide
    // enough for whatever list the fast-forward hands over)
    for (int i = tid; i < ff_SM * 4; i += ST) child[i] = 0;
    if (tid == 0) {
        tmp[12] = tmp[13] = tmp[10] = 0;
        tmp[11] = 0x7fffffff;
    }
    __syncthreads();

#ifdef AFV_SELECT_STATS
    st2 = wall_clock64();
#endif
    Rect16 *rc = rect0, *rn = rect1;
    int *cc = cnt0, *cn = cnt1;
    bool finish = (m2 == 0);
    bool phase_b = false;
    // the usual case (survivors fit the LDS arrays): every thread keeps its <= PPT points (position, node label) in registers for
    // all rounds; kn[] is written back once after the last round
    constexpr int PPT = SelCfg<ST>::PPT;
    const bool small = m2 > 0 && m2 <= KEPT_LDS;
    int nd_[PPT];
    uint32_t xy_[PPT];
    if (small) {
#pragma unroll
        for (int k_ = 0; k_ < PPT; ++k_) {
            const int p = min(tid + k_ * ST, m2 - 1);  // clamped: the tail threads carry a duplicate they never count
            nd_[k_] = kn[p];
            xy_[k_] = kxy[p];
        }
    }
    int par = 0;  // n_expand slot of this round: tmp[12 + par]

    // ---------------- fast-forward through the first rounds (round 4) ----------------
    // While phase A lasts, EVERY node with more than one point splits, so as long as no node holds a single point the list after round
    // d is the set of non-empty cells of the 4^d grid below each root (cells cut by DivideNode's own halving rule), in an order that
    // follows from "children of later-processed nodes first, n4 before n1": list index j_d = (S_(d-1) - 1 - j_(d-1)) * 4 + (3 - q_d),
    // j_0 = the root's index, S_d = roots * 4^d.  A point finds its cell at depths 1..3 by arithmetic on its root's box (no list, no
    // barrier), three LDS atomics count the cells' points, and the loop's own bookkeeping - sizes, nodes that can still split, the
    // termination and phase-B tests of every round (ORBextractor.cc:366-370) - is evaluated on those counts.  The deepest round D
    // for which the loop would still have been in phase A with every node splitting is adopted as the loop's starting state (list,
    // counts, boxes, labels): three rounds of ~3.5 us of dependent LDS phases become one pass.  Anything else (a single-point node, an
    // early stop) falls back to the loop from wherever the fast-forward is still exact - round 0 at worst.
    if (small && !finish && ff_dmax >= 1) {
        __shared__ int s_ff[4];  // wave 0's verdict: D, size of the list after round D, finished, phase B
        const int nroots = size, dmax = ff_dmax;
        uint32_t paths[PPT];  // natural path index at depth dmax, and the digits q1..q3 in bits 24..29
#pragma unroll
        for (int k_ = 0; k_ < PPT; ++k_) {
            paths[k_] = 0;
            if (tid + k_ * ST < m2) {
                Rect16 r = rc[nd_[k_]];
                int pth = nd_[k_];
                uint32_t dig = 0;
                for (int d = 1; d <= dmax; ++d) {
                    const int q = point_quadrant(xy_[k_], scale, r);
                    r = child_rect(r, q);
                    pth = pth * 4 + q;
                    dig |= (uint32_t)q << (22 + 2 * d);
                    atomicAdd(d == 1 ? &C1[pth] : d == 2 ? &C2[pth] : &C3[pth], 1);
                }
                paths[k_] = (uint32_t)pth | dig;
            }
        }
        __syncthreads();
        // list order -> natural path at depth D (and back: the digit flips are involutions)
        auto nat_of = [&](int j, int D, int &root) {
            int q[3] = {0, 0, 0};
            int S = nroots * (D == 1 ? 1 : D == 2 ? 4 : 16);
            for (int d = D; d >= 1; --d) {
                q[d - 1] = 3 - (j & 3);
                j = S - 1 - (j >> 2);
                S >>= 2;
            }
            root = j;
            int nat = j;
            for (int d = 0; d < D; ++d) nat = nat * 4 + q[d];
            return nat;
        };
        if (tid < 64) {
            // ONE wavefront replays the loop's bookkeeping on the counts (a workgroup-wide phase costs ~0.8 us of barrier + dependent LDS
            // round trips whatever it does; this is a few hundred counters): per depth the non-empty cells and the cells that can still
            // split, the loop's decisions round by round, then the list positions of the adopted depth by a wave-wide prefix sum
            int D = 0, sz_prev = nroots;
            bool fin = false, pb = false;
            int nx_prev = __ballot(tid < nroots && cc[min(tid, nroots - 1)] < 2) ? -1 : nroots;  // every root must split in round 1 (<= 16 roots)
            for (int d = 1, S = nroots * 4; d <= dmax; ++d, S *= 4) {
                if (fin || pb || nx_prev != sz_prev) break;  // the loop would not run round d as an all-splitting phase-A round
                const int *C = d == 1 ? C1 : d == 2 ? C2 : C3;
                int ne = 0, nx = 0;
                for (int i = tid; i < S; i += 64) {
                    const int c = C[i];
                    ne += c > 0;
                    nx += c > 1;
                }
                const int packed = __builtin_amdgcn_readlane(afv_wave_incl_scan(ne | (nx << 16)), 63);
                const int sz = packed & 0xffff;
                nx = packed >> 16;
                D = d;
                fin = sz >= N || sz == sz_prev;
                pb = !fin && sz + nx * 3 > N;
                sz_prev = sz;
                nx_prev = nx;
            }
            int new_size = 0;
            if (D >= 1) {
                const int SD = nroots * (D == 1 ? 4 : D == 2 ? 16 : 64);
                const int *C = D == 1 ? C1 : D == 2 ? C2 : C3;
                const int per = (SD + 63) / 64, b = tid * per, e = min(b + per, SD);
                int local = 0;
                for (int j = b; j < e; ++j) {
                    int root;
                    local += C[nat_of(j, D, root)] > 0 ? 1 : 0;
                }
                const int incl = afv_wave_incl_scan(local);
                new_size = __builtin_amdgcn_readlane(incl, 63);
                int pos = incl - local;
                for (int j = b; j < e; ++j) {
                    int root;
                    unt[j] = pos;  // list position of slot j (meaningful where the cell is non-empty)
                    pos += C[nat_of(j, D, root)] > 0 ? 1 : 0;
                }
            }
            if (tid == 0) {
                s_ff[0] = D;
                s_ff[1] = new_size;
                s_ff[2] = fin ? 1 : 0;
                s_ff[3] = pb ? 1 : 0;
            }
        }
        __syncthreads();
        const int D = s_ff[0];
        if (D >= 1) {
            const int SD = nroots * (D == 1 ? 4 : D == 2 ? 16 : 64);
            const int *C = D == 1 ? C1 : D == 2 ? C2 : C3;
            for (int j = tid; j < SD; j += ST) {
                int root;
                const int nat = nat_of(j, D, root);
                const int c = C[nat];
                if (c > 0) {
                    Rect16 r = rc[root];
                    for (int d = D - 1; d >= 0; --d) r = child_rect(r, (nat >> (2 * d)) & 3);
                    rn[unt[j]] = r;
                    cn[unt[j]] = c;
                }
            }
            // labels: a point's list slot from its digits
#pragma unroll
            for (int k_ = 0; k_ < PPT; ++k_) {
                if (tid + k_ * ST < m2) {
                    int nat = (int)(paths[k_] & 0xffffffu);
                    for (int d = dmax; d > D; --d) nat >>= 2;  // the path was taken down to dmax
                    int j = nat >> (2 * D), S = nroots;        // the root
                    for (int d = 1; d <= D; ++d) {
                        j = (S - 1 - j) * 4 + (3 - (int)((paths[k_] >> (22 + 2 * d)) & 3u));
                        S *= 4;
                    }
                    nd_[k_] = unt[j];
                }
            }
            size = s_ff[1];
            {
                Rect16 *t = rc; rc = rn; rn = t;
                int *u = cc; cc = cn; cn = u;
            }
            finish = s_ff[2] != 0;
            phase_b = s_ff[3] != 0;
        }
        __syncthreads();
    }

    // Barriers per round: occupancy | order | (scan: 2) | new list | relabel.  Everything a later phase reads is written at
    // least one barrier earlier; the clears for the NEXT round ride the relabel phase, which touches neither array.
    while (!finish) {
#ifdef AFV_SELECT_STATS
        const long long r0 = wall_clock64();
        ++st_rounds;
        st_b += phase_b;
#endif
        const int prev_size = size;
        // 1. child occupancy of every node that may split this round
        // each point's quadrant is computed ONCE per round (here) and reused when the points are relabelled in step 4: a thread keeps
        // the quadrants of its <= PPT points in a register while the survivors live in LDS (the usual case)
        uint32_t quads = 0;  // 2 bits per point of this thread
        if (small) {
            // staged so that the PPT node lookups are in flight together (one LDS round trip per stage, not per point)
            int c_[PPT];
            Rect16 r_[PPT];
#pragma unroll
            for (int k_ = 0; k_ < PPT; ++k_) {
                c_[k_] = cc[nd_[k_]];
                r_[k_] = rc[nd_[k_]];
            }
#pragma unroll
            for (int k_ = 0; k_ < PPT; ++k_) {
                if (tid + k_ * ST < m2 && c_[k_] > 1) {
                    const int q = point_quadrant(xy_[k_], scale, r_[k_]);
                    quads |= (uint32_t)q << (2 * k_);
                    atomicAdd(&child[nd_[k_] * 4 + q], 1);
                }
            }
        } else {
            for (int p = tid; p < m2; p += ST) {
                const int nd = kn[p];
                if (cc[nd] > 1) atomicAdd(&child[nd * 4 + point_quadrant(kxy[p], scale, rc[nd])], 1);
            }
        }
        __syncthreads();
#ifdef AFV_SELECT_STATS
        st_occ += wall_clock64() - r0;
#endif
        // 2. processing order.  aux[key] = number of non-empty children of the node processed key-th (0 if that
        //    node does not split); aux2[i] = key of node i or -1.
        int nproc;  // number of processing slots
        if (!phase_b) {
            for (int i = tid; i < size; i += ST) {
                int ne = 0;
                if (cc[i] > 1) ne = (child[4 * i] > 0) + (child[4 * i + 1] > 0) + (child[4 * i + 2] > 0) + (child[4 * i + 3] > 0);
                aux[i] = ne;
                aux2[i] = (cc[i] > 1) ? i : -1;
                unt[i] = (cc[i] > 1) ? 0 : 1;
            }
            nproc = size;
            __syncthreads();
        } else {
            // rank among expandable nodes by (count desc, list index asc) == sort ascending by (size, pointer) walked
            // from the back (ORBextractor.cc:381-382) with pointer ties resolved by creation order.  The ranks are a
            // permutation of 0..E-1, so aux[0..E) is fully rewritten here: aux[rank] = nodes gained by that split.
            // tmp[10] (E) and tmp[11] (stop rank) were reset in the previous round's relabel phase.
            int ecount = 0;
            // RL lanes per node share the O(size) rank count (the 1024-thread instantiation: 4 lanes, each a quarter of the list; the
            // partial ranks meet by two DPP quad shuffles)
            constexpr int RL = ST / 256;
            const int sub = tid & (RL - 1);
            for (int i0 = 0; i0 < size; i0 += ST / RL) {  // uniform trip count: the quad shuffles need whole quads
                const int i = i0 + tid / RL;
                int key = -1;
                const int ci = i < size ? cc[i] : 0;
                // (count, -index) as one key: node j goes first iff kj > ki.  Four nodes per LDS read (cnt arrays are
                // 16-byte aligned, M is a multiple of 64); slots past `size` are masked.
                const int ki = (ci << 12) | (4095 - i);
                int r = 0;
                if (ci > 1) {
#pragma unroll 4
                    for (int j = 4 * sub; j < size; j += 4 * RL) {
                        const int4 c4 = *reinterpret_cast<const int4 *>(cc + j);
                        const int k0 = (c4.x << 12) | (4095 - j), k1 = (c4.y << 12) | (4094 - j);
                        const int k2 = (c4.z << 12) | (4093 - j), k3 = (c4.w << 12) | (4092 - j);
                        r += (c4.x > 1 && k0 > ki) + (j + 1 < size && c4.y > 1 && k1 > ki) + (j + 2 < size && c4.z > 1 && k2 > ki) +
                             (j + 3 < size && c4.w > 1 && k3 > ki);
                    }
                }
                if (RL == 4) {
                    r += __builtin_amdgcn_update_dpp(0, r, 0xB1 /*quad_perm:[1,0,3,2]*/, 0xf, 0xf, true);
                    r += __builtin_amdgcn_update_dpp(0, r, 0x4E /*quad_perm:[2,3,0,1]*/, 0xf, 0xf, true);
                }
                if (ci > 1 && sub == 0) {
                    key = r;
                    ++ecount;
                    aux[key] = (child[4 * i] > 0) + (child[4 * i + 1] > 0) + (child[4 * i + 2] > 0) + (child[4 * i + 3] > 0) - 1;
                }
                if (i < size && sub == 0) aux2[i] = key;
            }
            ecount = wave_incl_scan(ecount);
            if (lane == 63) atomicAdd(&tmp[10], ecount);
            __syncthreads();
            const int E = tmp[10];
            block_excl_scan<ST>(aux, E, tmp);  // aux[r] = sum of deltas of ranks < r  (deltas >= 0: monotone)
            // stop rank r*: smallest r whose split lifts the node count to >= N (ORBextractor.cc:424-425); the
            // count after rank r is prev_size + aux[r+1]; if no rank reaches N every node is processed
            for (int r = tid; r + 1 < E; r += ST)
                if (prev_size + aux[r + 1] >= N) atomicMin(&tmp[11], r);
            __syncthreads();
            const int rstar = min(tmp[11], E - 1);
            // rebuild aux as "non-empty children by processing slot" for the ranks <= r* (later slots are never read)
            for (int i = tid; i < size; i += ST) {
                int key = aux2[i];
                if (key > rstar) key = -1;
                aux2[i] = key;
                unt[i] = (key < 0) ? 1 : 0;
                if (key >= 0)
                    aux[key] = (child[4 * i] > 0) + (child[4 * i + 1] > 0) + (child[4 * i + 2] > 0) + (child[4 * i + 3] > 0);
            }
            nproc = rstar + 1;
            __syncthreads();
        }
        // 3. suffix sums over the processing order: children of later-processed nodes come first in the new list
        // child (node i, quadrant q) -> position (total - aux[key] - ne(i)) + #non-empty children with quadrant > q
        // untouched node i (unt[i] = 1, set with aux2 above) -> total_children + rank among untouched nodes
        const int totals = block_excl_scan2<ST>(aux, nproc, unt, size, tmp);  // aux[key] = children of keys < key
        const int total_children = totals & 0xffff, untouched = (int)((unsigned)totals >> 16);
        const int new_size = total_children + untouched;
        int n_expand_local = 0;
        for (int i = tid; i < size; i += ST) {
            const int key = aux2[i];
            if (key < 0) {
                const int pos = total_children + unt[i];
                rn[pos] = rc[i];
                cn[pos] = cc[i];
                remap[4 * i] = (uint16_t)pos;
            } else {
                const int c0 = child[4 * i], c1 = child[4 * i + 1], c2 = child[4 * i + 2], c3 = child[4 * i + 3];
                const int ne = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                int pos = total_children - aux[key] - ne;  // first slot of this node's children (n4 first)
                const Rect16 r = rc[i];
                const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (cs[q] > 0) {
                        rn[pos] = child_rect(r, q);
                        cn[pos] = cs[q];
                        remap[4 * i + q] = (uint16_t)pos;
                        n_expand_local += (cs[q] > 1);
                        ++pos;
                    }
                }
            }
        }
        n_expand_local = wave_incl_scan(n_expand_local);
        if (lane == 63) atomicAdd(&tmp[12 + par], n_expand_local);
        __syncthreads();
        // 4. relabel the points; clear the next round's occupancy counters and n_expand slot
        for (int i = tid; i < new_size * 4; i += ST) child[i] = 0;
        if (tid == 0) {
            tmp[12 + (par ^ 1)] = 0;
            tmp[10] = 0;
            tmp[11] = 0x7fffffff;
        }
        if (small) {
            int key_[PPT];
#pragma unroll
            for (int k_ = 0; k_ < PPT; ++k_) key_[k_] = aux2[nd_[k_]];
#pragma unroll
            for (int k_ = 0; k_ < PPT; ++k_) {
                const int q = (key_[k_] >= 0) ? (int)((quads >> (2 * k_)) & 3u) : 0;  // aux2 >= 0 implies the node had > 1 point
                // the tail threads' clamped duplicates keep their (in-range, never counted) label: their quadrant bits were never
                // set, and remap[4 * node] of a split node whose first quadrant is empty is not written this round
                if (tid + k_ * ST < m2) nd_[k_] = remap[4 * nd_[k_] + q];
            }
        } else {
            for (int p = tid; p < m2; p += ST) {
                const int nd = kn[p];
                const int q = (aux2[nd] >= 0) ? point_quadrant(kxy[p], scale, rc[nd]) : 0;
                kn[p] = remap[4 * nd + q];
            }
        }
        __syncthreads();
        const int n_expand = tmp[12 + par];
        par ^= 1;
        size = new_size;
        {
            Rect16 *t = rc; rc = rn; rn = t;
            int *u = cc; cc = cn; cn = u;
        }
        // 5. termination (ORBextractor.cc:366-370, :427-430)
        if (size >= N || size == prev_size) finish = true;
        else if (!phase_b && size + n_expand * 3 > N) phase_b = true;
#ifdef AFV_SELECT_STATS
        if (st_rounds <= 8) st_r[st_rounds - 1] = wall_clock64() - r0;
#endif
    }

    if (small) {
#pragma unroll
        for (int k_ = 0; k_ < PPT; ++k_)
            if (tid + k_ * ST < m2) kn[tid + k_ * ST] = (uint16_t)nd_[k_];
    }
#ifdef AFV_SELECT_STATS
    st3 = wall_clock64();
#endif
    // ---------------- survivor of each node ----------------
    for (int i = tid; i < size; i += ST) best[i] = 0ull;
    __syncthreads();
    const int lw = L.w;
    for (int p = tid; p < m2; p += ST) {
        const uint32_t xy = kxy[p];
        const uint32_t raster = (xy >> 12) * (uint32_t)lw + (xy & 4095u);
        const unsigned long long key = ((unsigned long long)float_key(kr[p]) << 32) | (unsigned long long)(0xffffffffu - raster);
        atomicMax(&best[kn[p]], key);
    }
    __syncthreads();
    SelPoint *out = sel + (size_t)f * geo.sel_per_frame + L.sel_base;
    const int nout = min(size, L.sel_cap);
    for (int i = tid; i < nout; i += ST) {
        const unsigned long long key = best[i];
        const uint32_t raster = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
        SelPoint s;
        s.y = (uint16_t)(raster / (uint32_t)lw);
        s.x = (uint16_t)(raster - (uint32_t)s.y * (uint32_t)lw);
        s.response = key_float((uint32_t)(key >> 32));
        out[i] = s;
    }
    if (tid == 0) sel_count[f * AFV_MAX_LEVELS + l] = nout;
#ifdef AFV_SELECT_STATS
    if (tid == 0 && f == 0)
        printf("select level %d: n %d kept %d nodes %d rounds %d (phase B %d) | us: load + radix %lld, count + compaction + roots %lld, quadtree %lld, survivors %lld\n", l, n, m2, size,
               st_rounds, st_b, (st1 - st0) / 100, (st2 - st1) / 100, (st3 - st2) / 100, (wall_clock64() - st3) / 100),
        printf("   level %d rounds (x10 ns): %lld %lld %lld %lld %lld | occupancy total %lld\n", l, st_r[0], st_r[1], st_r[2], st_r[3], st_r[4], st_occ);
#endif
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_select_quadtree(const Geo *__restrict__ geo_p, const uint32_t *__restrict__ cand_packed,
                                                       const float *__restrict__ cand_resp,
                                                       const int *__restrict__ cand_count, uint32_t *__restrict__ kept_xy,
                                                       float *__restrict__ kept_resp, uint16_t *__restrict__ kept_node,
                                                       SelPoint *__restrict__ sel, int *__restrict__ sel_count, int M, int frame_base,
                                                       int total_blocks) {
    select_quadtree_body<256>(geo_p, cand_packed, cand_resp, cand_count, kept_xy, kept_resp, kept_node, sel, sel_count, M, frame_base, total_blocks);
}
// the small-batch instantiation (see the top of the file)
__global__ __launch_bounds__(1024) void k_select_quadtree_wide(const Geo *__restrict__ geo_p, const uint32_t *__restrict__ cand_packed,
                                                               const float *__restrict__ cand_resp, const int *__restrict__ cand_count,
                                                               uint32_t *__restrict__ kept_xy, float *__restrict__ kept_resp,
                                                               uint16_t *__restrict__ kept_node, SelPoint *__restrict__ sel,
                                                               int *__restrict__ sel_count, int M, int frame_base, int total_blocks) {
    select_quadtree_body<1024>(geo_p, cand_packed, cand_resp, cand_count, kept_xy, kept_resp, kept_node, sel, sel_count, M, frame_base, total_blocks);
}

static size_t select_lds_bytes(int M, int kept, bool wide) { return afv_select_tree_bytes(M, wide) + SEL_TMP * 4 + (size_t)kept * 6 /*kept xy, node*/; }
extern "C" size_t afv_select_lds_bytes(int M) { return std::max(select_lds_bytes(M, SelCfg<256>::KEPT_LDS, false), select_lds_bytes(M, SelCfg<1024>::KEPT_LDS, true)); }

// once per context (afv_create, on the context's device): the wide instantiation may need more dynamic LDS than the 64 KB a kernel gets
// by default.  false: the attribute call failed and the tables do not fit without it - the caller keeps to the 256-thread kernel.
extern "C" int afv_select_prepare(int M) {
    const size_t lds = select_lds_bytes(M, SelCfg<1024>::KEPT_LDS, true);
    if (lds <= 64 * 1024) return 1;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_select_quadtree_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return 1;
}

extern "C" void afv_launch_select(const Geo *geo_dev, int nlevels, const uint32_t *cand_packed, const float *cand_resp,
                                  const int *cand_count, uint32_t *kept_xy, float *kept_resp, uint16_t *kept_node,
                                  SelPoint *sel, int *sel_count, int M, int frame_base, int nframes, int wide, hipStream_t stream) {
    const int total = nlevels * nframes;
    dim3 grid((total + 7) / 8 * 8);
    if (wide) {
        const size_t lds = select_lds_bytes(M, SelCfg<1024>::KEPT_LDS, true);  // above 64 KB: afv_select_prepare raised the limit at afv_create
        hipLaunchKernelGGL(k_select_quadtree_wide, grid, dim3(1024), lds, stream, geo_dev, cand_packed, cand_resp, cand_count, kept_xy, kept_resp,
                           kept_node, sel, sel_count, M, frame_base, total);
        return;
    }
    const size_t lds = select_lds_bytes(M, SelCfg<256>::KEPT_LDS, false);
    hipLaunchKernelGGL(k_select_quadtree, grid, dim3(256), lds, stream, geo_dev, cand_packed, cand_resp,
                       cand_count, kept_xy, kept_resp, kept_node, sel, sel_count, M, frame_base, total);
}
