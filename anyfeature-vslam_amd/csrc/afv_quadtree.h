// afv_quadtree.h — FeatureExtractor::DistributeOctTree (reference src/ORBextractor.cc:239-458, DivideNode :181-237) as a
// level-synchronous build for one 1024-thread workgroup, over an abstract point set (float level-0 coordinates).
//
// Same algorithm as the ORB path's k_select.hip (see its header for the derivation: std::list order is determined by creation
// order, so the list is a dense array rebuilt with prefix sums; phase A splits every multi-point node, phase B splits
// largest-first until the quota is reached), factored out so that the AKAZE plugin can run it on sub-pixel keypoints
// (filterKeypoints_notScaled, reference src/FeatureExtractor.cpp:276-284, called from Feature_akaze61.cpp:63-65).
#ifndef AFV_QUADTREE_H
#define AFV_QUADTREE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define QT_T 1024
#define QT_NW (QT_T / 64)
#define QT_PPT 8  // points a thread keeps in registers (coordinates + node label): sets of up to 8192 points never re-read memory

struct QtRect {
    short x0, y0, x1, y1;
};

__device__ __forceinline__ uint32_t qt_float_key(float r) {  // order-preserving integer image of a float
    const uint32_t b = __float_as_uint(r);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ int qt_wave_incl_scan(int v) { return afv_wave_incl_scan(v); }

// exclusive prefix sum of arr[0..n) in place; returns the total.  `tmp` = QT_NW ints of LDS (the scalar slots live behind them).
__device__ inline int qt_block_excl_scan(int *arr, int n, int *tmp) {
    const int per = (n + QT_T - 1) / QT_T;
    const int b = threadIdx.x * per, e = min(b + per, n);
    int local = 0;
    for (int i = b; i < e; ++i) local += arr[i];
    const int incl = qt_wave_incl_scan(local);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) tmp[w] = incl;
    __syncthreads();
    int base = incl - local;
    for (int k = 0; k < w; ++k) base += tmp[k];
    int total = 0;
#pragma unroll
    for (int k = 0; k < QT_NW; ++k) total += tmp[k];
    for (int i = b; i < e; ++i) {
        const int v = arr[i];
        arr[i] = base;
        base += v;
    }
    __syncthreads();
    return total;
}

__device__ __forceinline__ int qt_quadrant(float px, float py, const QtRect r) {
    // ExtractorNode::DivideNode (ORBextractor.cc:181-224): halfX = ceil((UR.x-UL.x)/2), float compares
    const int hx = (r.x1 - r.x0 + 1) >> 1, hy = (r.y1 - r.y0 + 1) >> 1;
    const float mx = (float)(r.x0 + hx), my = (float)(r.y0 + hy);
    return (px < mx ? 0 : 1) + (py < my ? 0 : 2);  // n1: x<mx,y<my  n2: x>=mx,y<my  n3: x<mx,y>=my  n4: x>=mx,y>=my
}

__device__ __forceinline__ QtRect qt_child_rect(const QtRect r, int q) {
    const int hx = (r.x1 - r.x0 + 1) >> 1, hy = (r.y1 - r.y0 + 1) >> 1;
    QtRect c;
    c.x0 = (q & 1) ? (short)(r.x0 + hx) : r.x0;
    c.x1 = (q & 1) ? r.x1 : (short)(r.x0 + hx);
    c.y0 = (q & 2) ? (short)(r.y0 + hy) : r.y0;
    c.y1 = (q & 2) ? r.y1 : (short)(r.y0 + hy);
    return c;
}

// LDS scratch for M = max nodes: see qt_lds_bytes
struct QtScratch {
    QtRect *rect0, *rect1;
    int *cnt0, *cnt1, *child, *aux, *aux2, *scan, *tmp;
    uint16_t *remap;
};

__host__ __device__ inline size_t qt_lds_bytes(int M) {
    return (size_t)M * 8 * 2 /*rect*/ + (size_t)M * 4 * 2 /*cnt*/ + (size_t)M * 16 /*child*/ + (size_t)M * 4 * 3 /*aux, aux2, scan*/ + 128 /*tmp: QT_NW wave sums + scalar slots*/ +
           (size_t)M * 8 /*remap*/;
}

__device__ inline QtScratch qt_carve(char *smem, int M) {
    QtScratch S;
    S.rect0 = reinterpret_cast<QtRect *>(smem);
    S.rect1 = S.rect0 + M;
    S.cnt0 = reinterpret_cast<int *>(S.rect1 + M);
    S.cnt1 = S.cnt0 + M;
    S.child = S.cnt1 + M;
    S.aux = S.child + 4 * M;
    S.aux2 = S.aux + M;
    S.scan = S.aux2 + M;
    S.tmp = S.scan + M;
    S.remap = reinterpret_cast<uint16_t *>(S.tmp + 32);
    return S;
}

// Builds the tree over points p = 0..m2-1 (P.x(p), P.y(p) in level-0 pixels).  kn[p] (any memory) receives the index, in
// std::list order, of the node that holds point p; the return value is the number of nodes.  N = quota; n_ini / h_x /
// height describe the root boxes (ORBextractor.cc:243-283).  All QT_T threads must call it.
// REG: a thread keeps the coordinates and the node label of its <= QT_PPT points in registers for the whole build (the two passes over
// the points per round were two dependent memory round trips per point and pass: 240 us for the 7 000 points of one AKAZE level); sets
// beyond QT_T x QT_PPT points walk memory as before.
#define QT_FOR_POINTS(BODY)                                           \
    if (REG) {                                                        \
        _Pragma("unroll") for (int k_ = 0; k_ < QT_PPT; ++k_) {       \
            if (k_ * QT_T + tid < m2) {                               \
                const float px = rx[k_], py = ry[k_];                 \
                int nd = rn_[k_];                                     \
                BODY                                                  \
                rn_[k_] = nd;                                         \
            }                                                         \
        }                                                             \
    } else {                                                          \
        for (int p_ = tid; p_ < m2; p_ += QT_T) {                     \
            const float px = P.x(p_), py = P.y(p_);                   \
            int nd = kn[p_];                                          \
            BODY                                                      \
            kn[p_] = (uint16_t)nd;                                    \
        }                                                             \
    }
template <class Pts, bool REG>
__device__ int qt_build_impl(const Pts &P, int m2, int N, int n_ini, float h_x, int height, uint16_t *kn, const QtScratch &S) {
    const int tid = threadIdx.x, lane = tid & 63;
    QtRect *rect0 = S.rect0, *rect1 = S.rect1;
    int *cnt0 = S.cnt0, *cnt1 = S.cnt1, *child = S.child, *aux = S.aux, *aux2 = S.aux2, *scan = S.scan, *tmp = S.tmp;
    uint16_t *remap = S.remap;
    // root assignment: vpIniNodes[kp.pt.x / hX]
    if (tid < 16) aux[tid] = 0;
    __syncthreads();
    float rx[QT_PPT], ry[QT_PPT];
    int rn_[QT_PPT];
    if (REG) {
#pragma unroll
        for (int k_ = 0; k_ < QT_PPT; ++k_) {
            const int p = min(k_ * QT_T + tid, max(m2 - 1, 0));
            rx[k_] = m2 > 0 ? P.x(p) : 0.0f;
            ry[k_] = m2 > 0 ? P.y(p) : 0.0f;
            rn_[k_] = 0;
        }
    }
    QT_FOR_POINTS({
        (void)py;
        int root = 0;
        if (n_ini > 1) {
            root = min((int)(px / h_x), n_ini - 1);
            atomicAdd(&aux[root], 1);
        }
        nd = root;
    })
    __syncthreads();
    if (tid == 0) {
        int sz = 0;
        for (int i = 0; i < n_ini; ++i) {
            const int c = (n_ini > 1) ? aux[i] : m2;
            if (c > 0) {
                QtRect r;
                r.x0 = (short)(int)(h_x * (float)i);
                r.x1 = (short)(int)(h_x * (float)(i + 1));
                r.y0 = 0;
                r.y1 = (short)height;
                rect0[sz] = r;
                cnt0[sz] = c;
                aux2[i] = sz;
                ++sz;
            } else {
                aux2[i] = 0;
            }
        }
        tmp[QT_NW + 0] = sz;
    }
    __syncthreads();
    if (n_ini > 1) {
        QT_FOR_POINTS({ (void)px; (void)py; nd = aux2[nd]; })
    }
    int size = tmp[QT_NW + 0];
    __syncthreads();

    QtRect *rc = rect0, *rn = rect1;
    int *cc = cnt0, *cn = cnt1;
    bool finish = (m2 == 0);
    bool phase_b = false;
    while (!finish) {
        const int prev_size = size;
        // 1. child occupancy of every node that may split this round
        for (int i = tid; i < size * 4; i += QT_T) child[i] = 0;
        __syncthreads();
        QT_FOR_POINTS({
            if (cc[nd] > 1) atomicAdd(&child[nd * 4 + qt_quadrant(px, py, rc[nd])], 1);
        })
        __syncthreads();
        // 2. processing order: aux[key] = non-empty children of the node processed key-th; aux2[i] = key of node i or -1
        int nproc;
        if (!phase_b) {
            for (int i = tid; i < size; i += QT_T) {
                int ne = 0;
                if (cc[i] > 1) ne = (child[4 * i] > 0) + (child[4 * i + 1] > 0) + (child[4 * i + 2] > 0) + (child[4 * i + 3] > 0);
                aux[i] = ne;
                aux2[i] = (cc[i] > 1) ? i : -1;
            }
            nproc = size;
            __syncthreads();
        } else {
            // rank among expandable nodes by (count desc, list index asc) == ascending sort of (size, pointer) walked from the
            // back (ORBextractor.cc:381-382), pointer ties resolved by creation order
            if (tid == 0) tmp[QT_NW + 1] = 0;
            for (int i = tid; i < size; i += QT_T) aux[i] = 0;
            __syncthreads();
            int ecount = 0;
            for (int i = tid; i < size; i += QT_T) {
                int key = -1;
                const int ci = cc[i];
                if (ci > 1) {
                    int r = 0;
                    for (int j = 0; j < size; ++j) {
                        const int cj = cc[j];
                        r += (cj > 1) && (cj > ci || (cj == ci && j < i));
                    }
                    key = r;
                    ++ecount;
                }
                aux2[i] = key;
            }
            ecount = qt_wave_incl_scan(ecount);
            if (lane == 63) atomicAdd(&tmp[QT_NW + 1], ecount);
            __syncthreads();
            const int E = tmp[QT_NW + 1];
            for (int i = tid; i < size; i += QT_T) {
                const int key = aux2[i];
                if (key >= 0) aux[key] = (child[4 * i] > 0) + (child[4 * i + 1] > 0) + (child[4 * i + 2] > 0) + (child[4 * i + 3] > 0) - 1;
            }
            __syncthreads();
            qt_block_excl_scan(aux, E, tmp);  // aux[r] = sum of deltas of ranks < r
            if (tid == 0) tmp[QT_NW + 2] = E - 1;
            __syncthreads();
            // stop rank: smallest r whose split lifts the node count to >= N (ORBextractor.cc:424-425)
            for (int r = tid; r + 1 < E; r += QT_T)
                if (prev_size + aux[r + 1] >= N) atomicMin(&tmp[QT_NW + 2], r);
            __syncthreads();
            const int rstar = tmp[QT_NW + 2];
            __syncthreads();
            for (int i = tid; i < E; i += QT_T) aux[i] = 0;
            __syncthreads();
            for (int i = tid; i < size; i += QT_T) {
                int key = aux2[i];
                if (key > rstar) key = -1;
                aux2[i] = key;
                if (key >= 0) aux[key] = (child[4 * i] > 0) + (child[4 * i + 1] > 0) + (child[4 * i + 2] > 0) + (child[4 * i + 3] > 0);
            }
            nproc = rstar + 1;
            __syncthreads();
        }
        // 3. children of later-processed nodes come first in the new list; untouched nodes follow in their old order
        const int total_children = qt_block_excl_scan(aux, nproc, tmp);
        for (int i = tid; i < size; i += QT_T) scan[i] = (aux2[i] < 0) ? 1 : 0;
        __syncthreads();
        const int untouched = qt_block_excl_scan(scan, size, tmp);
        const int new_size = total_children + untouched;
        int n_expand_local = 0;
        for (int i = tid; i < size; i += QT_T) {
            const int key = aux2[i];
            if (key < 0) {
                const int pos = total_children + scan[i];
                rn[pos] = rc[i];
                cn[pos] = cc[i];
                remap[4 * i] = (uint16_t)pos;
            } else {
                const int c0 = child[4 * i], c1 = child[4 * i + 1], c2 = child[4 * i + 2], c3 = child[4 * i + 3];
                const int ne = (c0 > 0) + (c1 > 0) + (c2 > 0) + (c3 > 0);
                int pos = total_children - aux[key] - ne;  // first slot of this node's children (n4 first)
                const QtRect r = rc[i];
                const int cs[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (cs[q] > 0) {
                        rn[pos] = qt_child_rect(r, q);
                        cn[pos] = cs[q];
                        remap[4 * i + q] = (uint16_t)pos;
                        n_expand_local += (cs[q] > 1);
                        ++pos;
                    }
                }
            }
        }
        n_expand_local = qt_wave_incl_scan(n_expand_local);
        if (tid == 0) tmp[QT_NW + 3] = 0;
        __syncthreads();
        if (lane == 63) atomicAdd(&tmp[QT_NW + 3], n_expand_local);
        // 4. relabel the points
        QT_FOR_POINTS({
            const int q = (aux2[nd] >= 0) ? qt_quadrant(px, py, rc[nd]) : 0;
            nd = remap[4 * nd + q];
        })
        __syncthreads();
        const int n_expand = tmp[QT_NW + 3];
        size = new_size;
        {
            QtRect *t = rc; rc = rn; rn = t;
            int *u = cc; cc = cn; cn = u;
        }
        // 5. termination (ORBextractor.cc:366-370, :427-430)
        if (size >= N || size == prev_size) finish = true;
        else if (!phase_b && size + n_expand * 3 > N) phase_b = true;
        __syncthreads();
    }
    if (REG) {  // the labels leave the registers once
#pragma unroll
        for (int k_ = 0; k_ < QT_PPT; ++k_)
            if (k_ * QT_T + tid < m2) kn[k_ * QT_T + tid] = (uint16_t)rn_[k_];
    }
    __syncthreads();
    return size;
}
#undef QT_FOR_POINTS

template <class Pts>
__device__ int qt_build(const Pts &P, int m2, int N, int n_ini, float h_x, int height, uint16_t *kn, const QtScratch &S) {
    if (m2 <= QT_T * QT_PPT) return qt_build_impl<Pts, true>(P, m2, N, n_ini, h_x, height, kn, S);  // uniform
    return qt_build_impl<Pts, false>(P, m2, N, n_ini, h_x, height, kn, S);
}
#endif
