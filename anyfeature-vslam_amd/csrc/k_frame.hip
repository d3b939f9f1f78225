// k_frame.hip — the device-resident Frame (round 5): what Frame::Frame does between the extractor call and the first matcher
// (src/Frame.cc:171-223), on the device, so that nothing of a frame is uploaded again after its image.
//
//   k_frame_grid      per frame (one 1024-thread workgroup): optional phase 1 derives the per-feature arrays the matchers read - mvKeysUn
//                     x / y (a copy of mvKeys when mDistCoef[0] == 0, Frame.cc:405-409), angle, keyPtsSize / Sigma2 / Inf from the octave
//                     (FeatureExtractor.cpp:132-172), mvuRight = -1 (Frame.cc:197) - from the keypoints the describe kernel just wrote;
//                     phase 2 builds the grid of Frame::AssignFeaturesToGrid (Frame.cc:225-240) with PosInGrid's expression
//                     (Frame.cc:384-394): cell = round((x - mnMinX) * mfGridElementWidthInv) (ix * rows + iy), ascending feature index
//                     inside a cell - the order vCell is filled in and GetFeaturesInArea (Frame.cc:333-382) visits.
//   k_frame_gather    query descriptors by reference: rows (slot, idx) of a keyframe table -> the query block of a projection search
//   k_featvec_build   DBoW2::FeatureVector of a frame from the per-descriptor node ids of k_bow_transform: feature indices sorted by
//                     (node id, feature index), stopped words left out - the CSR body the BoW-guided matchers index (FeatureMatcher.cc:205-276)
//   k_table_promote   KeyFrame::KeyFrame(Frame&) (KeyFrame.cc:36-60): the frame's arrays into a slot of the keyframe table
#include "afv_device.h"
#include "afv_runtime.h"
#include "afv_jobs.h"

#define FG_T 1024
#define FG_NW (FG_T / 64)
#define FG_NOCELL 0xffffu
#define FG_TAB_MAX 4096  // keys (cells / nodes) the per-wave count table covers: 16 bytes each

// ---------------- stable counting sort of n features by a small key ----------------
// Both orders this file produces - the features of a grid cell by ascending index (the order vCell is filled in, Frame.cc:225-240, and
// GetFeaturesInArea visits, :333-382) and the features of a FeatureVector node by ascending index (FeatureVector::addFeature is called
// for i = 0, 1, ...) - are "feature i goes to start[key_i] + #{j < i : key_j == key_i}".  Chunks of 1024 features (a thread each):
//   * inside a wavefront the rank among equal keys comes from ballots over the distinct keys of the wave (a loop of at most 64, a few
//     scalar instructions per round);
//   * across the 16 wavefronts a table of 16 BYTES per key holds every wave's count of that key: one 16-byte LDS read per feature gives
//     the number of equal keys in the waves before it (v_dot4 over the masked bytes);
//   * start[] (the exclusive prefix of the key histogram) runs on from chunk to chunk.
// O(n) LDS traffic.  (The first version ranked by comparing against ALL earlier features - n^2 / 2 broadcast reads on one CU: 20 us for
// 1000 features, measured; this form: see DESIGN.)  Keys above FG_TAB_MAX (grids beyond 4096 cells, node levels wider than that) take
// the quadratic path below.
__device__ __forceinline__ int fg_bytes_before(const uint4 row, int wv) {  // sum of bytes [0, wv) of a 16-byte row
    const unsigned w[4] = {row.x, row.y, row.z, row.w};
    unsigned sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nb = min(max(wv - 4 * q, 0), 4);
        const unsigned m = nb >= 4 ? 0xffffffffu : ((1u << (8 * nb)) - 1u);
        sum = __builtin_amdgcn_udot4(w[q] & m, 0x01010101u, sum, false);
    }
    return (int)sum;
}

// s_key[i] (u16, FG_NOCELL = not placed) for i < n; s_start[key]: exclusive prefix of the histogram (advanced as the chunks go by);
// s_tab: nkeys x 16 bytes.  emit(i, pos) for every placed feature.  All 1024 threads call this.
// The caller zeroes s_tab before its last barrier ahead of the call (the first chunk then starts without one).
// A wave's count of a key is accumulated with ONE LDS atomic per feature (add 1 to the wave's byte of the key's row: at most 64 per byte,
// no carry); the rank among equal keys of the same wave - by lane, i.e. by feature index - is 0 for every key the wave holds once
// (read back from the row: LDS operations of a wave complete in order), and only lanes that share a key with another lane of their wave
// enter the ballot loop (grid cells: almost never; FeatureVector nodes: a few dozen rounds).
template <class Emit>
__device__ __forceinline__ void fg_stable_place(const unsigned short *s_key, int n, int nkeys, int *s_start, uint4 *s_tab, Emit emit) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned *tab32 = reinterpret_cast<unsigned *>(s_tab);
    const int sh = 8 * (wv & 3), wd = wv >> 2;
    for (int c0 = 0; c0 < n; c0 += FG_T) {
        const int i = c0 + tid;
        const unsigned key = i < n ? (unsigned)s_key[i] : FG_NOCELL;
        int rank = 0;
        if (key != FG_NOCELL) atomicAdd(&tab32[key * 4 + wd], 1u << sh);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int cnt = key != FG_NOCELL ? (int)((tab32[key * 4 + wd] >> sh) & 0xffu) : 0;
        unsigned long long todo = __ballot(cnt > 1);
        while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            const unsigned kc = (unsigned)__builtin_amdgcn_readlane((int)key, l);
            const unsigned long long m = __ballot(key == kc);
            if (key == kc) rank = __popcll(m & ((1ull << lane) - 1ull));
            todo &= ~m;
        }
        __syncthreads();
        if (key != FG_NOCELL) emit(i, s_start[key] + fg_bytes_before(s_tab[key], wv) + rank);
        if (c0 + FG_T >= n) break;  // the usual frame: one chunk, one barrier
        __syncthreads();
        if (key != FG_NOCELL && rank == 0) atomicAdd(&s_start[key], cnt);
        for (int k = tid; k < nkeys; k += FG_T) s_tab[k] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
}

// exclusive prefix of s_cnt[0 .. ncell) in place (+ the total at [ncell]); out (global, may be null) receives the same ncell + 1 values
__device__ __forceinline__ void fg_exclusive_scan(int *s_cnt, int ncell, int *out, int *s_wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (ncell + FG_T - 1) / FG_T;
    const int c0 = min(tid * per, ncell), c1 = min(c0 + per, ncell);
    int run = 0;
    for (int c = c0; c < c1; ++c) run += s_cnt[c];
    const int incl = afv_wave_incl_scan(run);
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int base = incl - run;
#pragma unroll
    for (int w = 0; w < FG_NW; ++w) base += w < wv ? s_wsum[w] : 0;
    for (int c = c0; c < c1; ++c) {
        const int v = s_cnt[c];
        s_cnt[c] = base;
        if (out) out[c] = base;
        base += v;
    }
    if (tid == FG_T - 1) {
        s_cnt[ncell] = base;
        if (out) out[ncell] = base;
    }
    __syncthreads();
}

// ---------------- k_frame_grid ----------------
// LDS: s_cnt[ncell + 1] (histogram, then the exclusive prefix = cell_ptr) | s_cell[n8] u16 (cell of every feature, FG_NOCELL = outside) |
// the 16-byte-per-cell wave table (grids up to FG_TAB_MAX cells)
__device__ __forceinline__ void frame_grid_body(const DevGridJob &J) {
    extern __shared__ __attribute__((aligned(16))) char fg_smem[];
    __shared__ int s_wsum[FG_NW];
    const int tid = threadIdx.x;
    // the first 1024 features (all of them for the usual frame) stay in registers from the keypoint load to the grid entry: their
    // position / size is never read back from memory.  The keypoint load does not wait for the count (slots below cap are readable).
    afv_keypoint k0{};
    if (J.kps && tid < J.cap) k0 = J.kps[tid];
    int n = J.n_ptr ? *J.n_ptr : J.n;
    n = min(max(n, 0), J.cap);
    float xr = 0.0f, yr = 0.0f, sr = 0.0f;
    // ---- phase 1: per-feature arrays from the keypoints ----
    if (J.kps) {
        for (int i = tid; i < n; i += FG_T) {
            const afv_keypoint k = i == tid ? k0 : J.kps[i];
            float xx, yy, ss;
            if (J.copy_xy) {
                xx = k.x;
                yy = k.y;
                J.x[i] = xx;
                J.y[i] = yy;
            } else {
                xx = J.x[i];
                yy = J.y[i];
            }
            J.angle[i] = k.angle;
            if (J.use_tab) {
                const int o = min(max(k.octave, 0), AFV_MAX_LEVELS - 1);
                ss = J.tab_size[o];
                J.size[i] = ss;
                J.sigma2[i] = J.tab_sigma2[o];
                J.inf[i] = J.tab_inf[o];
            } else {
                ss = J.size[i];
                const float s2 = ss * ss;  // keyPtsSigma2 = size^2, keyPtsInf = 1 / size^2 (FeatureExtractor.cpp:160-170)
                J.sigma2[i] = s2;
                J.inf[i] = 1.0f / s2;
            }
            if (J.fill_mono) J.u_right[i] = -1.0f;
            if (J.oct0) J.oct0[i] = k.octave == 0;  // SearchForInitialization only looks at level-0 keypoints of F1 (:485-489)
            if (i == tid) {
                xr = xx;
                yr = yy;
                sr = ss;
            }
        }
    } else if (J.cell_ptr && tid < n) {
        xr = J.x[tid];
        yr = J.y[tid];
        sr = J.size[tid];
    }
    if (!J.cell_ptr) return;
    const int ncell = J.cols * J.rows;
    const int n8 = (n + 7) & ~7;
    int *s_cnt = reinterpret_cast<int *>(fg_smem);
    unsigned short *s_cell = reinterpret_cast<unsigned short *>(fg_smem + (((size_t)ncell + 1) * 4 + 15) / 16 * 16);
    uint4 *s_tab = reinterpret_cast<uint4 *>(fg_smem + (((size_t)ncell + 1) * 4 + 15) / 16 * 16 + (((size_t)J.cap + 7) & ~(size_t)7) * 2);
    for (int c = tid; c <= ncell; c += FG_T) s_cnt[c] = 0;
    if (ncell <= FG_TAB_MAX)
        for (int c = tid; c < ncell; c += FG_T) s_tab[c] = make_uint4(0, 0, 0, 0);
    __syncthreads();  // (features beyond the first 1024 are read back by the thread that wrote them)
    for (int i = tid; i < n8; i += FG_T) {
        unsigned cell = FG_NOCELL;
        if (i < n) {
            const float xx = i == tid ? xr : J.x[i], yy = i == tid ? yr : J.y[i];
            const int px = (int)roundf((xx - J.min_x) * J.inv_w), py = (int)roundf((yy - J.min_y) * J.inv_h);  // PosInGrid
            if (!(px < 0 || px >= J.cols || py < 0 || py >= J.rows)) {
                cell = (unsigned)(px * J.rows + py);
                atomicAdd(&s_cnt[cell], 1);
            }
        }
        s_cell[i] = (unsigned short)cell;
    }
    __syncthreads();
    fg_exclusive_scan(s_cnt, ncell, J.cell_ptr, s_wsum);
    auto emit = [&](int i, int pos) {
        const bool mine = i == tid;
        J.cell_ent[pos] = make_int4(i, __float_as_int(mine ? xr : J.x[i]), __float_as_int(mine ? yr : J.y[i]), __float_as_int(mine ? sr : J.size[i]));
    };
    if (ncell <= FG_TAB_MAX) {
        fg_stable_place(s_cell, n, ncell, s_cnt, s_tab, emit);
        return;
    }
    // quadratic placement (grids beyond FG_TAB_MAX cells): rank = #{j < i : cell_j == cell_i}, eight cells per 16-byte LDS read
    for (int i = tid; i < n; i += FG_T) {
        const unsigned cell = s_cell[i];
        if (cell == FG_NOCELL) continue;
        int rank = 0;
        const uint4 *sc = reinterpret_cast<const uint4 *>(s_cell);
        const unsigned pat = cell | (cell << 16);
        const int full = i >> 3;  // groups of eight cells entirely below i
        for (int g = 0; g < full; ++g) {
            const uint4 v = sc[g];
            const unsigned w[4] = {v.x ^ pat, v.y ^ pat, v.z ^ pat, v.w ^ pat};
#pragma unroll
            for (int q = 0; q < 4; ++q) rank += ((w[q] & 0xffffu) == 0) + ((w[q] >> 16) == 0);
        }
        for (int j = full << 3; j < i; ++j) rank += s_cell[j] == cell;
        emit(i, s_cnt[cell] + rank);
    }
}

// many jobs, records in device memory (the host-array entry points: one job per search job) / one job, record as kernel argument (a resident
// frame: nothing to upload ahead of the launch)
extern "C" __global__ __launch_bounds__(FG_T) void k_frame_grid(const DevGridJob *__restrict__ jobs) { frame_grid_body(jobs[blockIdx.x]); }
extern "C" __global__ __launch_bounds__(FG_T) void k_frame_grid1(const DevGridJob J) { frame_grid_body(J); }

extern "C" size_t afv_frame_grid_lds(int cols, int rows, int cap) {
    const size_t ncell = (size_t)cols * rows;
    return ((ncell + 1) * 4 + 15) / 16 * 16 + (((size_t)cap + 7) & ~(size_t)7) * 2 + (ncell <= FG_TAB_MAX ? ncell * 16 : 0) + 16;
}

extern "C" void afv_launch_frame_grid(const DevGridJob *jobs, int njobs, size_t lds_bytes, hipStream_t stream) {
    if (njobs > 0) hipLaunchKernelGGL(k_frame_grid, dim3(njobs), dim3(FG_T), lds_bytes, stream, jobs);
}
extern "C" void afv_launch_frame_grid1(const DevGridJob *job, size_t lds_bytes, hipStream_t stream) {
    hipLaunchKernelGGL(k_frame_grid1, dim3(1), dim3(FG_T), lds_bytes, stream, *job);
}

// ---------------- k_frame_gather: descriptor rows by reference ----------------
// out[q] = table[(slot[q] * cap + idx[q])] (32 bytes = 2 x uint4; eight queries per 16 threads would not matter at this size: a thread per
// half row); a reference outside the table yields a zero row and raises *bad (may be null: the host checked the references)
extern "C" __global__ __launch_bounds__(256) void k_frame_gather(const uint8_t *__restrict__ table, const int *__restrict__ nset, int nsets, int cap,
                                                              const int *__restrict__ slot, const int *__restrict__ idx, int nq,
                                                              uint4 *__restrict__ out, int *__restrict__ bad) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int q = t >> 1, h = t & 1;
    if (q >= nq) return;
    const int s = slot[q], i = idx[q];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (s >= 0 && s < nsets && i >= 0 && i < min(nset[s], cap)) v = reinterpret_cast<const uint4 *>(table + ((size_t)s * cap + i) * 32)[h];
    else if (h == 0 && bad) atomicOr(bad, 1);
    out[2 * q + h] = v;
}

extern "C" void afv_launch_frame_gather(const uint8_t *table, const int *nset, int nsets, int cap, const int *slot, const int *idx, int nq,
                                        void *out, int *bad, hipStream_t stream) {
    if (nq > 0) hipLaunchKernelGGL(k_frame_gather, dim3((2 * nq + 255) / 256), dim3(256), 0, stream, table, nset, nsets, cap, slot, idx, nq,
                                   reinterpret_cast<uint4 *>(out), bad);
}

// ---------------- k_featvec_build ----------------
// FeatureVector::addFeature is called for i = 0, 1, ... (DBoW2 transform), so a node's list is ascending and the map iterates nodes in
// ascending id: the CSR body is the kept features sorted by (node id, feature index).  The descent (k_bow.hip) reports, next to the node
// id, the node's RANK among the nodes of its depth (ascending DBoW2 id, precomputed with the device image of the tree): sorting by that
// dense rank is sorting by node id, and the rank is small (<= k^(L - levelsup): 100 for the shipped vocabulary), so the body is one
// stable counting sort (fg_stable_place).  Levels wider than FG_TAB_MAX nodes rank by comparison (quadratic, toy trees only).
// Also mirrors leaf / node id / rank to the host arrays of the call when those are device-visible (h_* may be null).
extern "C" __global__ __launch_bounds__(FG_T) void k_featvec_build(const int *__restrict__ leaf, const int *__restrict__ nid, const int *__restrict__ dense,
                                                               int n, int cap, int width, const uint8_t *__restrict__ stopped,
                                                               int *__restrict__ seg_idx, int *__restrict__ n_kept, int *__restrict__ h_leaf,
                                                               int *__restrict__ h_nid, int *__restrict__ h_dense) {
    extern __shared__ __attribute__((aligned(16))) char fv_smem[];
    __shared__ int s_wsum[FG_NW];
    const int tid = threadIdx.x;
    n = min(max(n, 0), cap);
    const int n8 = (n + 7) & ~7;
    if (width <= FG_TAB_MAX) {
        int *s_cnt = reinterpret_cast<int *>(fv_smem);
        unsigned short *s_key = reinterpret_cast<unsigned short *>(fv_smem + (((size_t)width + 1) * 4 + 15) / 16 * 16);
        uint4 *s_tab = reinterpret_cast<uint4 *>(fv_smem + (((size_t)width + 1) * 4 + 15) / 16 * 16 + (((size_t)cap + 7) & ~(size_t)7) * 2);
        for (int c = tid; c <= width; c += FG_T) s_cnt[c] = 0;
        for (int c = tid; c < width; c += FG_T) s_tab[c] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        for (int i = tid; i < n8; i += FG_T) {
            unsigned key = FG_NOCELL;
            if (i < n) {
                const int lf = leaf[i], nd = nid[i], dr = dense[i];
                if (h_leaf) h_leaf[i] = lf;
                if (h_nid) h_nid[i] = nd;
                if (h_dense) h_dense[i] = dr;
                if (!(stopped && stopped[lf]) && dr >= 0 && dr < width) {
                    key = (unsigned)dr;
                    atomicAdd(&s_cnt[key], 1);
                }
            }
            s_key[i] = (unsigned short)key;
        }
        __syncthreads();
        fg_exclusive_scan(s_cnt, width, nullptr, s_wsum);
        if (tid == 0) *n_kept = s_cnt[width];
        fg_stable_place(s_key, n, width, s_cnt, s_tab, [&](int i, int pos) { seg_idx[pos] = i; });
        return;
    }
    // wide levels: rank by counting (node id, index) pairs below; node ids of the kept features in LDS (INT_MAX: a stopped word)
    int *s_nid = reinterpret_cast<int *>(fv_smem);
    __shared__ int s_kept;
    if (tid == 0) s_kept = 0;
    __syncthreads();
    const int n4 = (n + 3) & ~3;
    int kept = 0;
    for (int i = tid; i < n4; i += FG_T) {
        int v = 0x7fffffff;
        if (i < n) {
            const int lf = leaf[i], nd = nid[i];
            if (h_leaf) h_leaf[i] = lf;
            if (h_nid) h_nid[i] = nd;
            if (h_dense) h_dense[i] = dense[i];
            if (!(stopped && stopped[lf])) {
                v = nd;
                ++kept;
            }
        }
        s_nid[i] = v;
    }
    if (kept) atomicAdd(&s_kept, kept);
    __syncthreads();
    for (int i = tid; i < n; i += FG_T) {
        const int mine = s_nid[i];
        if (mine == 0x7fffffff) continue;
        int rank = 0;
        const int4 *sv = reinterpret_cast<const int4 *>(s_nid);
        const int full = i >> 2;
        for (int g = 0; g < full; ++g) {  // j < i: (nid_j, j) < (nid_i, i) <=> nid_j <= nid_i
            const int4 v = sv[g];
            rank += (v.x <= mine) + (v.y <= mine) + (v.z <= mine) + (v.w <= mine);
        }
        for (int j = full << 2; j < i; ++j) rank += s_nid[j] <= mine;
        for (int j = i + 1; j < ((i + 4) & ~3) && j < n4; ++j) rank += s_nid[j] < mine;
        for (int g = full + 1; g < (n4 >> 2); ++g) {  // j > i: strictly smaller node id
            const int4 v = sv[g];
            rank += (v.x < mine) + (v.y < mine) + (v.z < mine) + (v.w < mine);
        }
        seg_idx[rank] = i;
    }
    if (tid == 0) *n_kept = s_kept;
}

// once per context (afv_create): a 64 x 48 grid with 2000 features already asks for more than the 64 KB of dynamic LDS a kernel gets by
// default.  Returns the bytes a launch may ask for (without this the launch failed and the grid stayed the zeroed memset).
extern "C" int afv_frame_prepare(void) {
    const int want = 150 * 1024;
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_frame_grid), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_frame_grid1), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    ok = hipFuncSetAttribute(reinterpret_cast<const void *>(k_featvec_build), hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess && ok;
    if (!ok) (void)hipGetLastError();
    return ok ? want - 1024 : 63 * 1024;  // (the kernels' static arrays take a few hundred bytes)
}
extern "C" size_t afv_featvec_build_lds(int cap, int width) {
    const size_t cap8 = ((size_t)cap + 7) & ~(size_t)7;
    return width <= FG_TAB_MAX ? (((size_t)width + 1) * 4 + 15) / 16 * 16 + cap8 * 2 + (size_t)width * 16 + 16 : cap8 * 4 + 16;
}

extern "C" void afv_launch_featvec_build(const int *leaf, const int *nid, const int *dense, int n, int cap, int width, const uint8_t *stopped,
                                         int *seg_idx, int *n_kept, int *h_leaf, int *h_nid, int *h_dense, hipStream_t stream) {
    const size_t lds = afv_featvec_build_lds(cap, width);
    hipLaunchKernelGGL(k_featvec_build, dim3(1), dim3(FG_T), lds, stream, leaf, nid, dense, n, cap, width, stopped, seg_idx, n_kept, h_leaf, h_nid,
                       h_dense);
}

// ---------------- k_table_promote ----------------
// the frame's arrays into slot `set` of the table planes (descriptors 32 B rows, angle, x / y / sigma2 / mvuRight, FeatureVector body,
// validity = 1, count): one launch instead of nine small copies
extern "C" __global__ __launch_bounds__(256) void k_table_promote(PromoteArgs A) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 2 * A.n) A.t_desc[t] = A.f_desc[t];
    if (t < A.n) {
        A.t_angle[t] = A.f_angle[t];
        if (A.t_x) {
            A.t_x[t] = A.f_x[t];
            A.t_y[t] = A.f_y[t];
            A.t_sigma2[t] = A.f_sigma2[t];
            A.t_ur[t] = A.f_ur[t];
        }
    }
    if (A.t_idx && A.f_seg_idx && t < A.nkept) A.t_idx[t] = A.f_seg_idx[t];
    if (A.t_valid && t < A.cap) A.t_valid[t] = 1;
    if (t == 0) *A.t_n = A.n;
}

extern "C" void afv_launch_table_promote(const void *args, int n, int cap, hipStream_t stream) {
    const PromoteArgs &A = *reinterpret_cast<const PromoteArgs *>(args);
    const int work = std::max(std::max(2 * n, cap), 1);
    hipLaunchKernelGGL(k_table_promote, dim3((work + 255) / 256), dim3(256), 0, stream, A);
}
