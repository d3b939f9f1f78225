// afv_runtime.h — host-side internals shared by the translation units behind the C-ABI (afv_api.hip, afv_comm.hip):
// the context, the pinned staging arena (Blob) and the kernel launcher prototypes.  Not part of the public interface.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "afv_device.h"
#include "afv_jobs.h"

// ---- kernel launchers (k_*.hip) ----
extern "C" void afv_launch_resize(const uint8_t *src, int sw, int sh, int spitch, size_t sframe, uint8_t *dst, int dw, int dh,
                                  int dpitch, size_t dframe, const short2 *xt, const short2 *yt, int frame_base, int nframes, int *zero_counts,
                                  int n_zero, int *zero_one, hipStream_t stream);
extern "C" int afv_resize_window_ok(int sw, int sh, int dw, int dh);
extern "C" void afv_launch_pyramid_fused(const FrameSrc *src0, uint8_t *pyr, const PyrFuseArgs *args, size_t lds_bytes, int frame_base, int nframes,
                                         hipStream_t stream);
extern "C" int afv_pyramid_fused_prepare(size_t lds_bytes);
extern "C" void afv_launch_fast_nms(const Geo *geo, int total_tiles, const FrameSrc *src0, const uint8_t *pyr, uint32_t *cand_packed,
                                    int *cand_count, int frame_base, int nframes, hipStream_t stream);
extern "C" size_t afv_harris_queue_per_frame(const Geo *g);
extern "C" void afv_launch_retain_harris(const Geo *geo_dev, int nlevels, const FrameSrc *src0, const uint8_t *pyr,
                                         const uint32_t *cand_packed, const int *cand_count, uint32_t *l1, int *l1_count,
                                         float *l1_resp, uint2 *queue, int *queue_n, int frame_base, int nframes, int small, hipStream_t stream);
extern "C" size_t afv_select_lds_bytes(int M);
extern "C" void afv_launch_select(const Geo *geo_dev, int nlevels, const uint32_t *cand_packed, const float *cand_resp,
                                  const int *cand_count, uint32_t *kept_xy, float *kept_resp, uint16_t *kept_node, SelPoint *sel,
                                  int *sel_count, int M, int frame_base, int nframes, int wide, hipStream_t stream);
extern "C" int afv_describe_blocks_per_frame(const Geo *g);
// second destination of the describe kernel's outputs (afv_frame_extract: the frame's device arrays next to the pinned host copy); all null = none
struct DescribeMirror {
    afv_keypoint *kps;
    uint8_t *desc;
    int *n;
};
extern "C" void afv_launch_describe(const Geo *geo_dev, int blocks_per_frame, const FrameSrc *src0, const uint8_t *pyr,
                                    const SelPoint *sel, const int *sel_count, afv_keypoint *kps, uint8_t *desc,
                                    int cap_per_frame, int *n_out, int *status, int frame_base, int nframes, const DescribeMirror *mirror,
                                    hipStream_t stream);
extern "C" void afv_launch_describe_given(const Geo *geo_dev, const FrameSrc *src0, const uint8_t *pyr, const afv_keypoint *given, int n, uint8_t *desc,
                                          int frame, hipStream_t stream);
extern "C" void afv_launch_blur_level(const uint8_t *img, int w, int h, int pitch, uint8_t *out, hipStream_t stream);

extern "C" void afv_launch_match_bow(const DevMatchJob *jobs, int njobs, hipStream_t stream);
extern "C" void afv_launch_match_bow_seg(const DevMatchJob *jobs, int njobs, const void *tasks, int ntasks, int *hist, uint8_t *bins,
                                         const int *bin_off, int any_ori, hipStream_t stream);
extern "C" int afv_match_topk_slices(int cap, int engine, int want);
extern "C" void afv_launch_match_topk(const uint8_t *desc, const int *nset, int cap, const int *pa, const int *pb, int npairs,
                                      void *topk_scratch, int pair_base, int engine, int nslices, void *slice_scratch, int *tickets,
                                      hipStream_t stream);
extern "C" void afv_launch_match_resolve(const uint8_t *desc, const float *ang, int ang_stride, const int *nset, int cap, const int *pa,
                                         const int *pb, int npairs, float th, float ratio, int check_ori, int *match, int *nmatches,
                                         const void *topk_scratch, int pair_base, int engine, hipStream_t stream);
extern "C" void afv_launch_match_tri(const DevTriJob *jobs, int njobs, int max_n1, hipStream_t stream);
extern "C" void afv_launch_match_l2(const float *d1, int n1, const float *d2, int n2, int dim, const uint8_t *v1,
                                    const uint8_t *v2, float th, float ratio, int *out, int *nmatches, hipStream_t stream);

extern "C" void afv_launch_distinctive_f32(const float *desc, int dim, const int *set_ptr, int nsets, int *best_idx, float *best_median, hipStream_t stream);
extern "C" void afv_launch_distinctive(const uint32_t *desc, const int *set_ptr, int nsets, int words, int desc_bytes, int *best_idx, int *best_median,
                                       hipStream_t stream);
extern "C" void afv_launch_match_projection(const DevProjJob *jobs, int njobs, int max_nq, size_t wg_lds, const DevProjJob *one, int *ticket,
                                            hipStream_t stream);
extern "C" int afv_project_prepare(void);
extern "C" int afv_match_prepare(void);
extern "C" int afv_select_prepare(int M);
extern "C" int afv_debug_pass_cap;
extern "C" size_t afv_project_wg_lds(int kind_init, int n, int nq, int float_rows);
extern "C" size_t afv_frame_grid_lds(int cols, int rows, int cap);
extern "C" int afv_frame_prepare(void);
extern "C" size_t afv_featvec_build_lds(int cap, int width);
extern "C" void afv_launch_frame_grid(const DevGridJob *jobs, int njobs, size_t lds_bytes, hipStream_t stream);
extern "C" void afv_launch_frame_grid1(const DevGridJob *job, size_t lds_bytes, hipStream_t stream);
extern "C" void afv_launch_frame_gather(const uint8_t *table, const int *nset, int nsets, int cap, const int *slot, const int *idx, int nq,
                                        void *out, int *bad, hipStream_t stream);
extern "C" void afv_launch_featvec_build(const int *leaf, const int *nid, const int *dense, int n, int cap, int width, const uint8_t *stopped,
                                         int *seg_idx, int *n_kept, int *h_leaf, int *h_nid, int *h_dense, hipStream_t stream);
extern "C" void afv_launch_table_promote(const void *args, int n, int cap, hipStream_t stream);
extern "C" void afv_launch_match_fuse(const DevProjJob *jobs, int njobs, int max_nq, const DevProjJob *one, hipStream_t stream);
extern "C" size_t afv_match_l2_scratch_bytes(int n1, int n2, int *ntiles_out, int *cols_per_tile_out);
extern "C" int afv_launch_match_l2_tiled(const float *d1, int n1, const float *d2, int n2, int dim, const uint8_t *v1, const uint8_t *v2,
                                         float th, float ratio, int *out, int *nmatches, void *scratch, int ntiles, int cols_per_tile,
                                         hipStream_t stream);
extern "C" int afv_launch_match_l2_pairs(const float *desc, const int *nset, int cap, int dim, const int *pa, const int *pb, int npairs,
                                         int pair_base, float th, float ratio, int *out, int *nmatches, void *scratch, hipStream_t stream);
extern "C" void afv_launch_match_init(const DevProjJob *jobs, int njobs, int max_nq, size_t wg_lds, const DevProjJob *one, int *ticket,
                                      hipStream_t stream);

extern "C" void afv_launch_bow_transform(const DevVocab *v, const uint32_t *desc, int n, int levelsup, int *leaf_node,
                                         int *node_at_level, int *rank_at_level, hipStream_t stream);
extern "C" int afv_launch_bow_transform_f32(const DevVocab *v, const float *desc, int n, int dim, int levelsup, int *leaf_node, int *node_at_level,
                                            int *rank_at_level, hipStream_t stream);
struct afv_vocab {
    DevVocab dev{};
    int desc_bytes = 32;
    int float_dim = 0;           // > 0: node descriptors are float_dim floats (afv_vocab_create_f32), `desc_bytes` = 4 * float_dim
    void *d_rec = nullptr;       // breadth-first records (k_bow.hip)
    uint8_t *d_stopped = nullptr;
    std::vector<uint8_t> h_stopped;  // host copy of the stop list (empty: none)
    std::vector<int> depth_width;    // nodes per depth (index = depth, 0 = the root)
};

// the device-resident Frame (afv_frame.hip)
struct afv_frame {
    afv_ctx *c = nullptr;
    afv_frame_params p{};
    int cap = 0, n = 0;
    int desc_bytes = 32, words = 8;  // descriptor size and dwords per (zero-padded) device row: 8 up to 32 bytes, 16 up to 64
    int float_dim = 0;               // > 0: the rows are float_dim floats (desc_bytes = 4 * float_dim, words = float_dim): L2^2 distances
    float inv_w = 0, inv_h = 0;
    bool has_features = false, has_grid = false, has_fv = false;
    // device arrays, one allocation
    uint8_t *d_block = nullptr;
    afv_keypoint *d_kps = nullptr;
    uint8_t *d_desc = nullptr;
    float *d_x = nullptr, *d_y = nullptr, *d_size = nullptr, *d_angle = nullptr, *d_sigma2 = nullptr, *d_inf = nullptr, *d_ur = nullptr;
    int *d_n = nullptr, *d_cell_ptr = nullptr, *d_leaf = nullptr, *d_nid = nullptr, *d_dense = nullptr, *d_seg_idx = nullptr, *d_nkept = nullptr;
    int4 *d_cell_ent = nullptr;
    uint8_t *d_oct0 = nullptr;         // octave == 0 per feature (the query filter of SearchForInitialization)
    // host side of the FeatureVector: node structure for the merge-join
    std::vector<int32_t> fv_node_id, fv_seg_ptr;
    int fv_total = 0;
};

#define AFV_MAX_SIDE 8192

struct afv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // second lane for split batches (latency-bound kernels overlap VALU-bound ones)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream_copy = nullptr;                         // copy lane of the host-buffer batch pipeline (created on first use)
    std::vector<hipEvent_t> pipe_ev;                           // 3 events per chunk in flight
    int pipe_chunk = 64;                                       // frames per pipeline chunk
    int pipe_ahead = 8;                                        // uploads run this many chunks ahead of the compute
    int split_min_frames = 64;     // batches of at least this many frames / pairs are split over the two streams
    int match_engine = AFV_MATCH_ENGINE_MFMA;  // phase 1 of the brute-force pair matcher; afv_set_match_engine
    int resolve_engine = 2;        // phase 2: 0 = ordered walk on one wavefront (64-row rounds), 1 = workgroup-wide fixed point, 2 = by call size
    int resolve_wg_max_pairs = 256; // ... 2: calls of at most this many pairs take the fixed point; afv_set_match_resolve
    int proj_engine = 2;           // ordered phase of the projection searches: 0 = ordered walk, 1 = workgroup fixed point, 2 = 1 when it fits
    int proj_wg_lds_max = 0;       // dynamic LDS bytes the workgroup engines may use (0: unavailable); afv_project_prepare at afv_create
    int frame_lds_max = 0;         // dynamic LDS bytes k_frame_grid / k_featvec_build may use; afv_frame_prepare at afv_create
    int *d_proj_ticket = nullptr;  // hand-off ticket of the one-launch search (k_proj_search1), zero at rest
    int proj_fuse = 1;             // 1: single-job searches rank and resolve in one launch (afv_set_projection_fuse)
    std::vector<afv_frame *> frames;  // frames alive on this context (destroyed with it)
    int split_chunks = 0;          // ... into this many chunks (alternating streams); 0 = about 85 frames each; afv_set_split_chunks
    // small-batch ("latency") path: kernels shaped for one or a few frames; afv_set_small_batch_path
    int small_mode = 1;            // 0 = never, 1 = batches of at most small_max_frames, 2 = always
    int small_max_frames = 4;
    int pf_tw = 16, pf_th = 8;     // top-level tile of the one-launch pyramid (k_pyramid_fused): 204 workgroups for one 640 x 480 frame - the
                                   // kernel's level loop is bound by the vector ALU of the CUs it runs on (32 x 16 tiles: 54 CUs, 7.7 us of levels)
    bool pf_ok = false;            // the current geometry has a one-launch pyramid (else: level-by-level launches)
    PyrFuseArgs pf{};
    size_t pf_lds = 0;
    uint8_t *d_pf_blob = nullptr;  // device image of the plan (afv_device.h)
    size_t pf_blob_cap = 0;
    afv_orb_params p{};
    Geo geo{};          // current geometry (host copy)
    Geo cap_geo{};      // geometry of (max_width, max_height): sizes every allocation
    Geo *d_geo = nullptr;
    bool geo_valid = false;
    short2 *d_tab = nullptr;  // resize tables, all levels
    size_t tab_off_x[AFV_MAX_LEVELS]{}, tab_off_y[AFV_MAX_LEVELS]{};
    size_t tab_elems = 0;
    uint8_t *d_pyr = nullptr;
    uint32_t *d_cand_packed = nullptr, *d_kept_xy = nullptr;
    uint32_t *d_l1 = nullptr;          // per (frame, level): the candidates that survive retainBest on the FAST score ...
    float *d_l1_resp = nullptr;        // ... and their Harris responses
    int *d_l1_count = nullptr;
    uint2 *d_hq = nullptr;             // Harris work queue, `hq_per_frame` items per frame; a launch over the frames
    int *d_hq_n = nullptr;             // [f0, f0 + nf) owns the slice starting at frame f0 and the counter d_hq_n[f0]
    size_t hq_per_frame = 0;
    float *d_kept_resp = nullptr;
    uint16_t *d_kept_node = nullptr;
    int *d_cand_count = nullptr, *d_sel_count = nullptr;
    SelPoint *d_sel = nullptr;
    int select_M = 64;
    bool select_wide_ok = true;   // the 1024-thread quadtree kernel got its LDS (afv_select_prepare at afv_create)
    // staging of the host-pointer entry points
    uint8_t *d_frames = nullptr;
    size_t frames_pitch = 0, frames_stride = 0;
    uint8_t *d_out_block = nullptr;  // [n][kps][desc] of the host-pointer calls in one allocation; the three pointers below point into it
    size_t out_kps_off = 0, out_desc_off = 0, out_bytes = 0;
    afv_keypoint *d_kps = nullptr;
    uint8_t *d_desc = nullptr;
    int *d_n = nullptr, *d_status = nullptr;
    int stage_cap = 0;
    // matcher staging (grow only)
    uint8_t *d_match = nullptr;
    size_t match_bytes = 0;
    uint8_t *h_stage = nullptr;  // pinned host image of d_match (matcher staging both ways), grow-only
    size_t stage_bytes = 0;
    bool stage_pinned = false;
    void *d_topk = nullptr;  // [npairs][cap] 2 x int4: key record per row (seven (distance, column) keys + the exact-prefix length)
    size_t topk_bytes = 0;
    // column-sliced phase 1 (small-batch path): per-slice records [npairs][cap][nslices] 2 x int4 and the row-tile tickets (zero at rest)
    void *d_l2_scratch = nullptr;  // key records of the float-descriptor pair matcher (afv_match_l2_pairs_device), its own buffer
    size_t l2_bytes = 0;
    int l2_chunk_pairs = 2048;     // pairs per launch of that matcher; afv_set_l2_chunk_pairs
    void *d_slice = nullptr;
    size_t slice_bytes = 0;
    int *d_tickets = nullptr;
    size_t tickets_n = 0;
    // last extraction (debug getters)
    FrameSrc last_src{};
    int last_nframes = 0;
    std::string last_error;
    // live stage timing
    bool prof = false;             // the CURRENT call is timed (set per entry-point call from prof_every and the call counters)
    int prof_every = 0;            // 0 = profiling off; n = time the stages of every n-th extraction / pair-match call
    unsigned prof_tick_extract = 0, prof_tick_match = 0;
    std::vector<hipEvent_t> prof_ev[AFV_NUM_STAGES];  // pairs (begin, end)
    size_t prof_used[AFV_NUM_STAGES]{};
    int prof_launches[AFV_NUM_STAGES]{};
    float prof_ms[AFV_NUM_STAGES]{};
    long long prof_units[AFV_NUM_STAGES]{};  // frames (pairs for the match stage) covered by the timed launches
};

struct StageTimer {  // RAII: record begin/end events around one stage on the launch stream
    afv_ctx *c;
    int stage;
    hipStream_t s;
    hipEvent_t e1 = nullptr;
    StageTimer(afv_ctx *c_, int stage_, hipStream_t s_, int units = 0) : c(c_), stage(stage_), s(s_) {
        if (!c->prof) return;
        c->prof_units[stage] += units;
        auto &v = c->prof_ev[stage];
        size_t &u = c->prof_used[stage];
        if (u + 2 > v.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            v.push_back(a);
            v.push_back(b);
        }
        (void)hipEventRecord(v[u], s);
        e1 = v[u + 1];
        u += 2;
    }
    ~StageTimer() {
        if (e1) (void)hipEventRecord(e1, s);
    }
};


#define HIPCHK(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? AFV_ENOMEM : AFV_EHIP;                                \
        }                                                                                            \
    } while (0)

// Fills and device-to-device copies that a later launch depends on.  hipMemset / hipMemcpy(device -> device) run on the NULL stream and
// may return before they have run; the contexts' streams are created non-blocking, so nothing orders a launch on them behind such an
// operation (round 6: a fresh context's first sliced match could meet ticket words the fill had not reached yet - or have its counts
// wiped by a fill that ran late).  These run on the context's own stream and the host waits for them.
static inline hipError_t afv_fill(afv_ctx *c, void *p, int v, size_t n) {
    if (!n) return hipSuccess;
    const hipError_t e = hipMemsetAsync(p, v, n, c->stream);
    return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}
static inline hipError_t afv_copy_dd(afv_ctx *c, void *dst, const void *src, size_t n) {
    if (!n) return hipSuccess;
    const hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDefault, c->stream);
    return e == hipSuccess ? hipStreamSynchronize(c->stream) : e;
}

static inline int cv_round(float v) { return (int)lrintf(v); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// the C-ABI never throws: host allocation failures inside the matcher entry points become AFV_ENOMEM
template <class F>
static inline int guarded(afv_ctx *c, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        if (c) c->last_error = "out of host memory";
        return AFV_ENOMEM;
    } catch (...) {
        if (c) c->last_error = "unexpected exception";
        return AFV_EHIP;
    }
}

// versioned job arrays (include/afv_hip.h, "JOB RECORDS THAT CARRY struct_size"): the caller's array, whatever layout it was compiled
// against, copied into the current one; fields its layout does not have are zero.  false = AFV_EINVAL (missing / inconsistent / absurd
// struct_size)
template <class T>
static inline bool afv_load_jobs(const T *jobs, int njobs, size_t first_layout_bytes, std::vector<T> &out) {
    if (!jobs || njobs < 1) return false;
    const uint8_t *base = reinterpret_cast<const uint8_t *>(jobs);
    uint32_t ss;
    std::memcpy(&ss, base, sizeof(ss));
    if (ss < first_layout_bytes || ss > 4 * sizeof(T) || (ss & 3)) return false;
    out.resize((size_t)njobs);
    for (int i = 0; i < njobs; ++i) {
        const uint8_t *p = base + (size_t)i * ss;
        uint32_t si;
        std::memcpy(&si, p, sizeof(si));
        if (si != ss) return false;
        std::memset(static_cast<void *>(&out[i]), 0, sizeof(T));
        std::memcpy(static_cast<void *>(&out[i]), p, std::min<size_t>(ss, sizeof(T)));
        out[i].struct_size = (uint32_t)sizeof(T);
    }
    return true;
}

// ---- matcher staging ----
// Host image of the device staging buffer.  It lives in the context's pinned arena, so the one H2D copy of a call and the
// D2H copies of its results are true async DMA transfers (no pageable bounce inside the runtime); results land at the same
// offsets in the arena and are handed to the caller's arrays after the stream sync.
struct HostImage {
    afv_ctx *c;
    size_t n = 0;
    uint8_t *data() { return c->h_stage; }
    size_t size() const { return n; }
    void resize(size_t m, bool zero) {
        if (m > c->stage_bytes) {
            const size_t want = align_up(m + m / 2, 1 << 20);
            uint8_t *np = nullptr;
            bool pinned = hipHostMalloc(reinterpret_cast<void **>(&np), want, hipHostMallocDefault) == hipSuccess && np;
            if (!pinned) {
                (void)hipGetLastError();
                np = static_cast<uint8_t *>(std::malloc(want));
                if (!np) throw std::bad_alloc();
            }
            if (n) std::memcpy(np, c->h_stage, n);
            if (c->h_stage) {
                if (c->stage_pinned) (void)hipHostFree(c->h_stage);
                else std::free(c->h_stage);
            }
            c->h_stage = np;
            c->stage_bytes = want;
            c->stage_pinned = pinned;
        }
        if (zero && m > n) std::memset(c->h_stage + n, 0, m - n);
        n = m;
    }
};

struct Blob {
    HostImage h;
    struct Pending { void *dst; size_t off, bytes; };
    std::vector<Pending> pending;
    explicit Blob(afv_ctx *c) : h{c} {}
    size_t put(const void *src, size_t bytes, size_t align = 16) {
        const size_t off = align_up(h.size(), align);
        h.resize(off + bytes, src == nullptr);
        if (src && bytes) std::memcpy(h.data() + off, src, bytes);
        return off;
    }
    size_t reserve(size_t bytes, size_t align = 16) {  // zero-filled (counters, histograms, padded rows rely on it)
        const size_t off = align_up(h.size(), align);
        h.resize(off + bytes, true);
        return off;
    }
    size_t reserve_scratch(size_t bytes, size_t align = 16) {  // device-only scratch: never copied, never filled
        const size_t off = align_up(h.size(), align);
        h.resize(off + bytes, false);
        return off;
    }
    // queue a device -> caller copy of [off, off + bytes): DMA into the arena now, memcpy to dst in finish()
    hipError_t fetch(void *dst, size_t off, size_t bytes, hipStream_t s) {
        if (!bytes) return hipSuccess;
        pending.push_back(Pending{dst, off, bytes});
        return hipMemcpyAsync(h.data() + off, h.c->d_match + off, bytes, hipMemcpyDeviceToHost, s);
    }
    void finish() {
        for (const Pending &p : pending) std::memcpy(p.dst, h.data() + p.off, p.bytes);
        pending.clear();
    }
};

// scratch of a column-sliced phase 1 over `npairs` pairs (grow-only; growing implies a device sync, tickets are zeroed once)
static inline int ensure_slice_scratch(afv_ctx *c, int npairs, int cap, int nslices) {
    if (nslices <= 1) return AFV_OK;
    const size_t need = (size_t)npairs * cap * 32 * nslices, nt = (size_t)npairs * ((cap + 255) / 256);
    if (need > c->slice_bytes || nt > c->tickets_n) HIPCHK(c, hipDeviceSynchronize());
    if (need > c->slice_bytes) {
        if (c->d_slice) (void)hipFree(c->d_slice);
        c->d_slice = nullptr;
        c->slice_bytes = 0;
        HIPCHK(c, hipMalloc(&c->d_slice, need));
        c->slice_bytes = need;
    }
    if (nt > c->tickets_n) {
        if (c->d_tickets) (void)hipFree(c->d_tickets);
        c->d_tickets = nullptr;
        c->tickets_n = 0;
        HIPCHK(c, hipMalloc(&c->d_tickets, nt * sizeof(int)));
        HIPCHK(c, afv_fill(c, c->d_tickets, 0, nt * sizeof(int)));
        c->tickets_n = nt;
    }
    return AFV_OK;
}

static inline int ensure_match_buffer(afv_ctx *c, size_t bytes) {
    if (bytes <= c->match_bytes) return AFV_OK;
    if (c->d_match) (void)hipFree(c->d_match);
    c->d_match = nullptr;
    c->match_bytes = 0;
    const size_t want = align_up(bytes + bytes / 2, 1 << 20);
    HIPCHK(c, hipMalloc(&c->d_match, want));
    c->match_bytes = want;
    return AFV_OK;
}

// descriptors -> rows of `words` dwords (zero padded)
static inline size_t put_desc(Blob &b, const uint8_t *d, int n, int desc_bytes, int words) {
    const size_t off = b.reserve((size_t)std::max(n, 1) * words * 4);
    for (int i = 0; i < n; ++i) {
        uint8_t *row = b.h.data() + off + (size_t)i * words * 4;
        std::memset(row, 0, (size_t)words * 4);
        std::memcpy(row, d + (size_t)i * desc_bytes, (size_t)desc_bytes);
    }
    return off;
}

// ---- the keyframe table (afv_comm.hip; afv_frame.hip gathers query descriptors from it) ----
struct HostFeatVec {  // host copy of one keyframe's FeatureVector (node ids, CSR pointers, feature indices)
    std::vector<int32_t> node_id, seg_ptr, seg_idx;
};

struct afv_table {
    afv_ctx *c = nullptr;
    int nsets = 0, cap = 0;
    uint8_t *d_desc = nullptr;  // [nsets][cap][32]
    float *d_angle = nullptr;   // [nsets][cap]
    int32_t *d_n = nullptr;     // [nsets]
    int32_t *d_idx = nullptr;   // [nsets][cap] FeatureVector feature indices in node order (afv_table_set_featvec), lazily allocated
    float *d_geo = nullptr;     // [4][nsets][cap]: x, y, sigma2, mvuRight (afv_table_set_geometry / _u_right; -1 = monocular), lazily allocated
    uint8_t *d_valid = nullptr; // [nsets][cap] "map point exists && !isBad()" (afv_table_set_valid), lazily allocated, default 1
    std::vector<int32_t> h_n;
    std::vector<HostFeatVec> fv;
    std::vector<uint8_t> has_fv, has_geo;  // per set: afv_table_set_featvec / afv_table_set_geometry called since the last afv_table_set
    std::vector<uint8_t> fv_body_on_device;  // per set: the FeatureVector body came from a frame (afv_table_set_from_frame): no host copy yet
    // grow-only device buffers of the pair entry points + their pinned host image
    int32_t *d_pairs = nullptr;  // [2][pair_cap]
    int32_t *d_out = nullptr;    // [pair_cap][cap]
    int32_t *d_nm = nullptr;     // [pair_cap]
    int32_t *h_pin = nullptr;    // pinned: [2][pair_cap] pairs, then [pair_cap] counts, then [pair_cap][cap] matches
    int pair_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// ---- shared between afv_api.hip and afv_comm.hip ----
int afv_match_pairs_core(afv_ctx *c, const uint8_t *d_desc, const float *d_ang, int ang_stride, const int32_t *d_n, int cap,
                         const int32_t *d_pair_a, const int32_t *d_pair_b, int npairs, float th_low, float nnratio,
                         int check_orientation, int32_t *d_match, int32_t *d_nmatches, hipStream_t s);
void afv_shared_segments(const afv_match_job &j, std::vector<Seg> &segs);
int afv_check_resolve_guard(afv_ctx *c, const int32_t *nmatches, int n);
void afv_table_release_all(afv_ctx *c);  // afv_destroy: tables / communicators still alive die with their context
void afv_frame_release_all(afv_ctx *c);  // ... and so do its frames
int afv_frame_after_extract(afv_frame *f, hipStream_t s);  // afv_frame.hip: k_frame_grid behind the describe kernel of afv_frame_extract
int afv_extract_into_frame(afv_ctx *c, afv_frame *f, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps,
                           uint8_t *desc32, int cap, int *n_out);  // afv_api.hip: afv_orb_extract with the frame as second destination
// the host side of E12 (FeatureExtractor.cpp:132-172): keyPtsSize of octave `o` as afv_orb_size_sigma computes it
float afv_size_of_octave(const afv_ctx *c, int octave);
// projection searches over a feature side that is already on the device (afv_frame.hip) share the staging / launch code of the host-pointer
// entry points (afv_api.hip)
struct ProjFeatureSide {      // device pointers of the feature side; null fdesc = stage it from the job's host arrays
    const uint32_t *fdesc = nullptr;
    int n = 0, words = 8;
    const float *x = nullptr, *y = nullptr, *size = nullptr, *angle = nullptr, *inf = nullptr, *u_right = nullptr;
    const int *cell_ptr = nullptr;
    const int4 *cell_ent = nullptr;
    const uint32_t *qdesc_dev = nullptr;  // queries' descriptors already on the device (another frame's rows)
    const afv_table *qref_table = nullptr;  // ... or rows (slot, idx) of a keyframe table, gathered on the device (host arrays of nq ints)
    const int32_t *qref_slot = nullptr, *qref_idx = nullptr;
    const float *qangle_dev = nullptr;
    const uint8_t *qvalid_dev = nullptr;
};
enum { AFV_KIND_PROJ = 0, AFV_KIND_FUSE = 1, AFV_KIND_INIT = 2 };
int afv_match_projection_core(afv_ctx *c, const afv_proj_job *jobs, int njobs, int32_t *assign, int32_t *nmatches, int kind,
                              const ProjFeatureSide *dev_side);
