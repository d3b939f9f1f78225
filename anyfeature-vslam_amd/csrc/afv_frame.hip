// afv_frame.hip — host side of the device-resident Frame (include/afv_hip.h, "the device-resident Frame"; kernels: k_frame.hip).
//
// Reference object: Frame (src/Frame.cc:171-223).  What the constructor computes after the extractor call - mvKeysUn (:403-433), the
// per-feature scale data (keyPtsSize / Sigma2 / Inf, FeatureExtractor.cpp:132-172), the 64 x 48 grid (:225-240) - and what the tracking
// thread then asks of it - GetFeaturesInArea inside the projection searches (Tracking.cc:747,753,1026), ComputeBoW (:397-401) ahead of
// SearchByBoW (Tracking.cc:626-629), the copy into a KeyFrame (KeyFrame.cc:36-60) - happens here against arrays that never leave HBM.
#include <mutex>
#include <unordered_set>

#include "afv_runtime.h"

// Frames alive in the process.  A Frame of the host commonly outlives the extractor (the tracker's last / initial frames at shutdown,
// adapter ~DeviceFrame): afv_frame_destroy after afv_destroy must be a no-op, and it can only know by looking the POINTER up - the
// object, and the context it names, are gone by then (reading f->c there was a use-after-free).
static std::mutex g_frames_mutex;
static std::unordered_set<const afv_frame *> g_live_frames;

static void frame_free(afv_frame *f) {
    if (!f) return;
    {
        std::lock_guard<std::mutex> lk(g_frames_mutex);
        g_live_frames.erase(f);
    }
    if (f->c) (void)hipSetDevice(f->c->device);
    if (f->d_block) (void)hipFree(f->d_block);
    delete f;
}

void afv_frame_release_all(afv_ctx *c) {
    std::vector<afv_frame *> mine;
    mine.swap(c->frames);
    for (afv_frame *f : mine) frame_free(f);
}

extern "C" int afv_frame_create(afv_ctx *c, const afv_frame_params *params, afv_frame **out) {
    if (!c || !params || !out) return AFV_EINVAL;
    *out = nullptr;
    if (params->struct_size < offsetof(afv_frame_params, cap) + sizeof(int32_t) || params->struct_size > 4 * sizeof(afv_frame_params)) return AFV_EINVAL;
    afv_frame_params p{};
    std::memcpy(&p, params, std::min<size_t>(params->struct_size, sizeof(p)));
    if (p.grid_cols < 1 || p.grid_rows < 1 || (long)p.grid_cols * p.grid_rows > 8192) return AFV_EINVAL;
    if (!(p.max_x > p.min_x) || !(p.max_y > p.min_y)) return AFV_EINVAL;
    if (p.float_dim != 0 && (p.float_dim < 4 || p.float_dim > 1024 || (p.float_dim & 3))) return AFV_EINVAL;
    const int desc_bytes = p.float_dim ? 4 * p.float_dim : (p.desc_bytes == 0 ? AFV_DESC_BYTES : p.desc_bytes);  // (callers of the round-5 layout: the fields read 0)
    if (!p.float_dim && (desc_bytes < 1 || desc_bytes > 64)) return AFV_EINVAL;
    const int words = p.float_dim ? p.float_dim : (desc_bytes <= 32 ? 8 : 16);
    const int cap = p.cap > 0 ? p.cap : c->stage_cap;
    if (cap < 1 || cap > AFV_MAX_SIDE) return AFV_EINVAL;
    // the grid and the FeatureVector body are each built by ONE workgroup in LDS: what does not fit is refused here, not at the first launch
    if (afv_frame_grid_lds(p.grid_cols, p.grid_rows, cap) > (size_t)c->frame_lds_max || afv_featvec_build_lds(cap, 1) > (size_t)c->frame_lds_max) {
        c->last_error = "afv_frame_create: grid_cols x grid_rows x cap does not fit the LDS of one workgroup";
        return AFV_EUNSUPPORTED;
    }
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        afv_frame *f = new (std::nothrow) afv_frame();
        if (!f) return AFV_ENOMEM;
        f->c = c;
        f->p = p;
        f->cap = cap;
        f->desc_bytes = desc_bytes;
        f->words = words;
        f->float_dim = p.float_dim;
        // Frame.cc:201-202: mfGridElementWidthInv = FRAME_GRID_COLS / (mnMaxX - mnMinX), same for the height (float arithmetic)
        f->inv_w = static_cast<float>(p.grid_cols) / static_cast<float>(p.max_x - p.min_x);
        f->inv_h = static_cast<float>(p.grid_rows) / static_cast<float>(p.max_y - p.min_y);
        const size_t ncell = (size_t)p.grid_cols * p.grid_rows;
        size_t off = 0;
        auto take = [&](size_t bytes) {
            const size_t o = off;
            off = align_up(off + bytes, 256);
            return o;
        };
        const size_t o_kps = take((size_t)cap * sizeof(afv_keypoint)), o_desc = take((size_t)cap * words * 4);
        size_t o_f[7];
        for (size_t &o : o_f) o = take((size_t)cap * 4);
        const size_t o_n = take(16), o_cptr = take((ncell + 1) * 4), o_cent = take((size_t)cap * 16);
        const size_t o_leaf = take((size_t)cap * 4), o_nid = take((size_t)cap * 4), o_seg = take((size_t)cap * 4), o_oct = take((size_t)cap);
        const size_t o_dense = take((size_t)cap * 4);
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&f->d_block), off);
        if (e == hipSuccess) e = hipMemsetAsync(f->d_block, 0, off, c->stream);
        if (e != hipSuccess) {
            c->last_error = std::string("afv_frame_create: ") + hipGetErrorString(e);
            frame_free(f);
            return e == hipErrorOutOfMemory ? AFV_ENOMEM : AFV_EHIP;
        }
        uint8_t *B = f->d_block;
        f->d_kps = reinterpret_cast<afv_keypoint *>(B + o_kps);
        f->d_desc = B + o_desc;
        float **fp[7] = {&f->d_x, &f->d_y, &f->d_size, &f->d_angle, &f->d_sigma2, &f->d_inf, &f->d_ur};
        for (int i = 0; i < 7; ++i) *fp[i] = reinterpret_cast<float *>(B + o_f[i]);
        f->d_n = reinterpret_cast<int *>(B + o_n);
        f->d_nkept = f->d_n + 1;
        f->d_cell_ptr = reinterpret_cast<int *>(B + o_cptr);
        f->d_cell_ent = reinterpret_cast<int4 *>(B + o_cent);
        f->d_leaf = reinterpret_cast<int *>(B + o_leaf);
        f->d_nid = reinterpret_cast<int *>(B + o_nid);
        f->d_seg_idx = reinterpret_cast<int *>(B + o_seg);
        f->d_oct0 = B + o_oct;
        f->d_dense = reinterpret_cast<int *>(B + o_dense);
        try {
            c->frames.push_back(f);
            std::lock_guard<std::mutex> lk(g_frames_mutex);
            g_live_frames.insert(f);
        } catch (...) {
            auto it = std::find(c->frames.begin(), c->frames.end(), f);
            if (it != c->frames.end()) c->frames.erase(it);
            frame_free(f);
            return AFV_ENOMEM;
        }
        *out = f;
        return AFV_OK;
    });
}

extern "C" void afv_frame_destroy(afv_frame *f) {
    if (!f) return;
    {
        std::lock_guard<std::mutex> lk(g_frames_mutex);
        if (!g_live_frames.count(f)) return;  // released with its context (afv_destroy): the pointer is dead, nothing of it may be read
    }
    afv_ctx *c = f->c;
    auto it = std::find(c->frames.begin(), c->frames.end(), f);
    if (it == c->frames.end()) return;  // already released with its context
    c->frames.erase(it);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    frame_free(f);
}

// the job record of k_frame_grid for this frame; `soa`: derive the per-feature arrays from d_kps first; `grid`: build the grid
static int frame_launch_grid(afv_frame *f, bool soa, bool copy_xy, bool use_tab, bool fill_mono, bool grid, bool n_on_device, hipStream_t s) {
    afv_ctx *c = f->c;
    DevGridJob g{};
    g.n_ptr = n_on_device ? f->d_n : nullptr;
    g.n = f->n;
    g.cap = f->cap;
    g.kps = soa ? f->d_kps : nullptr;
    g.copy_xy = copy_xy;
    g.use_tab = use_tab;
    g.fill_mono = fill_mono;
    for (int o = 0; o < AFV_MAX_LEVELS; ++o) {
        const float sz = afv_size_of_octave(c, o), s2 = sz * sz;  // afv_orb_size_sigma's arithmetic, per octave
        g.tab_size[o] = sz;
        g.tab_sigma2[o] = s2;
        g.tab_inf[o] = 1.0f / s2;
    }
    g.x = f->d_x; g.y = f->d_y; g.size = f->d_size; g.angle = f->d_angle; g.sigma2 = f->d_sigma2; g.inf = f->d_inf; g.u_right = f->d_ur;
    g.oct0 = f->d_oct0;
    g.min_x = f->p.min_x; g.min_y = f->p.min_y; g.inv_w = f->inv_w; g.inv_h = f->inv_h; g.cols = f->p.grid_cols; g.rows = f->p.grid_rows;
    g.cell_ptr = grid ? f->d_cell_ptr : nullptr;
    g.cell_ent = f->d_cell_ent;
    afv_launch_frame_grid1(&g, afv_frame_grid_lds(g.cols, g.rows, f->cap), s);  // the record is a kernel argument: nothing to upload
    HIPCHK(c, hipGetLastError());
    return AFV_OK;
}

// called by extract_one (afv_api.hip) right after the describe kernel was enqueued with the frame as second destination
int afv_frame_after_extract(afv_frame *f, hipStream_t s) {
    f->has_features = true;
    f->has_fv = false;
    f->has_grid = false;
    const int rc = frame_launch_grid(f, true, !f->p.distorted, true, true, !f->p.distorted, true, s);
    if (rc) return rc;
    f->has_grid = !f->p.distorted;
    return AFV_OK;
}

extern "C" int afv_frame_extract(afv_frame *f, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps, uint8_t *desc32,
                                 int cap, int *n_out) {
    if (!f || !gray || stride_bytes < width) return AFV_EINVAL;
    if ((kps || desc32) && (!kps || !desc32 || !n_out || cap < 1)) return AFV_EINVAL;
    if (f->desc_bytes != AFV_DESC_BYTES) return AFV_EUNSUPPORTED;  // this is the ORB32 extractor
    return afv_extract_into_frame(f->c, f, gray, width, height, stride_bytes, kps, desc32, kps ? cap : 0x7fffffff, n_out);
}

extern "C" int afv_frame_set_features(afv_frame *f, const afv_keypoint *kps, const uint8_t *desc32, int n, const float *size, const float *u_right) {
    if (!f || n < 0 || n > f->cap || (n > 0 && (!kps || !desc32))) return AFV_EINVAL;
    afv_ctx *c = f->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        hipStream_t s = c->stream;
        // pageable sources: hipMemcpyAsync stages them before it returns (the caller's arrays may die after the call)
        if (n) {
            HIPCHK(c, hipMemcpyAsync(f->d_kps, kps, (size_t)n * sizeof(afv_keypoint), hipMemcpyHostToDevice, s));
            const size_t row = (size_t)f->words * 4;
            if ((size_t)f->desc_bytes == row) {
                HIPCHK(c, hipMemcpyAsync(f->d_desc, desc32, (size_t)n * row, hipMemcpyHostToDevice, s));
            } else {  // rows of desc_bytes -> zero-padded device rows (the distance over the padded dwords is the distance over the bytes)
                HostImage arena{c};
                arena.resize((size_t)n * row, false);
                uint8_t *hb = arena.data();
                for (int i = 0; i < n; ++i) {
                    std::memcpy(hb + (size_t)i * row, desc32 + (size_t)i * f->desc_bytes, (size_t)f->desc_bytes);
                    std::memset(hb + (size_t)i * row + f->desc_bytes, 0, row - (size_t)f->desc_bytes);
                }
                HIPCHK(c, hipMemcpyAsync(f->d_desc, hb, (size_t)n * row, hipMemcpyHostToDevice, s));
            }
            if (size) HIPCHK(c, hipMemcpyAsync(f->d_size, size, (size_t)n * 4, hipMemcpyHostToDevice, s));
            if (u_right) HIPCHK(c, hipMemcpyAsync(f->d_ur, u_right, (size_t)n * 4, hipMemcpyHostToDevice, s));
        }
        HIPCHK(c, hipMemcpyAsync(f->d_n, &n, sizeof(int), hipMemcpyHostToDevice, s));
        f->n = n;
        f->has_features = true;
        f->has_fv = false;
        f->has_grid = false;
        const int rc = frame_launch_grid(f, true, !f->p.distorted, size == nullptr, u_right == nullptr, !f->p.distorted, false, s);
        if (rc) return rc;
        f->has_grid = !f->p.distorted;
        HIPCHK(c, hipStreamSynchronize(s));  // the sources were pageable: nothing of the caller's may still be in flight
        return AFV_OK;
    });
}

extern "C" int afv_frame_set_undistorted(afv_frame *f, const float *x, const float *y) {
    if (!f || !f->has_features || !f->p.distorted || (f->n > 0 && (!x || !y))) return AFV_EINVAL;
    afv_ctx *c = f->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        hipStream_t s = c->stream;
        if (f->n) {
            HIPCHK(c, hipMemcpyAsync(f->d_x, x, (size_t)f->n * 4, hipMemcpyHostToDevice, s));
            HIPCHK(c, hipMemcpyAsync(f->d_y, y, (size_t)f->n * 4, hipMemcpyHostToDevice, s));
        }
        const int rc = frame_launch_grid(f, false, false, false, false, true, false, s);
        if (rc) return rc;
        f->has_grid = true;
        HIPCHK(c, hipStreamSynchronize(s));
        return AFV_OK;
    });
}

extern "C" int afv_frame_count(const afv_frame *f) { return f ? f->n : AFV_EINVAL; }

extern "C" int afv_frame_device_ptrs(afv_frame *f, afv_keypoint **d_kps, uint8_t **d_desc, float **d_x, float **d_y, float **d_size, float **d_angle,
                                     int32_t **d_n) {
    if (!f) return AFV_EINVAL;
    if (d_kps) *d_kps = f->d_kps;
    if (d_desc) *d_desc = f->d_desc;
    if (d_x) *d_x = f->d_x;
    if (d_y) *d_y = f->d_y;
    if (d_size) *d_size = f->d_size;
    if (d_angle) *d_angle = f->d_angle;
    if (d_n) *d_n = f->d_n;
    return AFV_OK;
}

extern "C" int afv_frame_get_grid(afv_frame *f, int32_t *cell_ptr, int32_t *cell_idx) {
    if (!f || !f->has_grid) return AFV_EINVAL;
    afv_ctx *c = f->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const size_t ncell = (size_t)f->p.grid_cols * f->p.grid_rows;
        std::vector<int32_t> ptr(ncell + 1);
        HIPCHK(c, hipMemcpy(ptr.data(), f->d_cell_ptr, (ncell + 1) * 4, hipMemcpyDeviceToHost));
        if (cell_ptr) std::memcpy(cell_ptr, ptr.data(), (ncell + 1) * 4);
        const int total = ptr[ncell];
        if (cell_idx && total > 0) {
            std::vector<int32_t> ent((size_t)total * 4);
            HIPCHK(c, hipMemcpy(ent.data(), f->d_cell_ent, (size_t)total * 16, hipMemcpyDeviceToHost));
            for (int i = 0; i < total; ++i) cell_idx[i] = ent[(size_t)i * 4];
        }
        return AFV_OK;
    });
}

// ---- Frame::ComputeBoW ----
extern "C" int afv_frame_bow_transform(afv_frame *f, const afv_vocab *v, int levelsup, int32_t *leaf_node, int32_t *node_at_level, int32_t *nnodes_out) {
    if (!f || !v || !f->has_features) return AFV_EINVAL;
    if (v->float_dim != f->float_dim || v->dev.words != f->words || v->desc_bytes != f->desc_bytes) return AFV_EUNSUPPORTED;  // a vocabulary of another descriptor kind / size
    afv_ctx *c = f->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        hipStream_t s = c->stream;
        const int n = f->n;
        f->has_fv = false;
        f->fv_node_id.clear();
        f->fv_seg_ptr.clear();
        f->fv_total = 0;
        if (nnodes_out) *nnodes_out = 0;
        if (n == 0) {
            f->has_fv = true;
            return AFV_OK;
        }
        // results for the host: leaf / node ids / sort keys land in the pinned arena, written by the kernel that builds the body
        const size_t row = align_up((size_t)n * 4, 256);
        HostImage arena{c};
        arena.resize(3 * row + 256, false);
        int *h_leaf = reinterpret_cast<int *>(arena.data()), *h_nid = reinterpret_cast<int *>(arena.data() + row);
        int *h_dense = reinterpret_cast<int *>(arena.data() + 2 * row), *h_kept = reinterpret_cast<int *>(arena.data() + 3 * row);
        const bool zc = c->stage_pinned;
        // sort key of a feature: 0 = the root, 1 + rank of its node among the nodes of depth L - levelsup (k_bow.hip)
        const int nid_level = v->dev.L - levelsup;
        const int width = (nid_level > 0 && (size_t)nid_level < v->depth_width.size()) ? v->depth_width[(size_t)nid_level] + 1 : 1;
        if (afv_featvec_build_lds(f->cap, width) > (size_t)c->frame_lds_max) {
            c->last_error = "afv_frame_bow_transform: the node level is too wide for the LDS of one workgroup";
            return AFV_EUNSUPPORTED;
        }
        if (f->float_dim) {
            if (!afv_launch_bow_transform_f32(&v->dev, reinterpret_cast<const float *>(f->d_desc), n, f->float_dim, levelsup, f->d_leaf, f->d_nid, f->d_dense, s)) {
                c->last_error = "afv_frame_bow_transform: float descriptors of this dimension have no descent kernel (64, 128, 256)";
                return AFV_EUNSUPPORTED;
            }
        } else {
            afv_launch_bow_transform(&v->dev, reinterpret_cast<const uint32_t *>(f->d_desc), n, levelsup, f->d_leaf, f->d_nid, f->d_dense, s);
        }
        afv_launch_featvec_build(f->d_leaf, f->d_nid, f->d_dense, n, f->cap, width, v->dev.stopped, f->d_seg_idx, zc ? h_kept : f->d_nkept,
                                 zc ? h_leaf : nullptr, zc ? h_nid : nullptr, zc ? h_dense : nullptr, s);
        HIPCHK(c, hipGetLastError());
        if (!zc) {
            HIPCHK(c, hipMemcpyAsync(h_leaf, f->d_leaf, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(h_nid, f->d_nid, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(h_dense, f->d_dense, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(c, hipMemcpyAsync(h_kept, f->d_nkept, sizeof(int), hipMemcpyDeviceToHost, s));
        }
        HIPCHK(c, hipStreamSynchronize(s));
        if (leaf_node) std::memcpy(leaf_node, h_leaf, (size_t)n * 4);
        if (node_at_level) std::memcpy(node_at_level, h_nid, (size_t)n * 4);
        // node structure of the FeatureVector (what the merge-join walks, FeatureMatcher.cc:205-276): distinct node ids ascending and
        // their segment sizes - a histogram over the same sort keys the device placed the body by
        const int kept = *h_kept;
        std::vector<int32_t> cnt((size_t)width, 0), node_of((size_t)width, 0);
        int host_kept = 0;
        for (int i = 0; i < n; ++i) {
            const int lf = h_leaf[i], key = h_dense[i];
            if (lf < 0 || lf >= v->dev.nnodes || key < 0 || key >= width) return AFV_EHIP;
            if (!v->h_stopped.empty() && v->h_stopped[(size_t)lf]) continue;
            ++cnt[(size_t)key];
            node_of[(size_t)key] = h_nid[i];
            ++host_kept;
        }
        if (host_kept != kept) {
            c->last_error = "afv_frame_bow_transform: host and device disagree on the stopped words";
            return AFV_EHIP;
        }
        for (int k = 0; k < width; ++k) {
            if (!cnt[(size_t)k]) continue;
            if (f->fv_node_id.empty()) f->fv_seg_ptr.push_back(0);
            f->fv_node_id.push_back(node_of[(size_t)k]);
            f->fv_seg_ptr.push_back(f->fv_seg_ptr.back() + cnt[(size_t)k]);
        }
        f->fv_total = kept;
        f->has_fv = true;
        if (nnodes_out) *nnodes_out = (int32_t)f->fv_node_id.size();
        return AFV_OK;
    });
}

extern "C" int afv_frame_get_featvec(afv_frame *f, int32_t *node_id, int32_t *seg_ptr, int32_t *seg_idx) {
    if (!f || !f->has_fv) return AFV_EINVAL;
    afv_ctx *c = f->c;
    const size_t nn = f->fv_node_id.size();
    if (node_id && nn) std::memcpy(node_id, f->fv_node_id.data(), nn * 4);
    if (seg_ptr) {
        if (nn) std::memcpy(seg_ptr, f->fv_seg_ptr.data(), (nn + 1) * 4);
        else seg_ptr[0] = 0;
    }
    if (seg_idx && f->fv_total > 0) {
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(seg_idx, f->d_seg_idx, (size_t)f->fv_total * 4, hipMemcpyDeviceToHost));
    }
    return AFV_OK;
}

// ---- the projection searches against a resident frame ----
static void frame_side(const afv_frame *f, ProjFeatureSide &S) {
    S.fdesc = reinterpret_cast<const uint32_t *>(f->d_desc);
    S.n = f->n;
    S.words = f->words;
    S.x = f->d_x; S.y = f->d_y; S.size = f->d_size; S.angle = f->d_angle; S.inf = f->d_inf; S.u_right = f->d_ur;
    S.cell_ptr = f->d_cell_ptr;
    S.cell_ent = f->d_cell_ent;
}

static int frame_proj_job(const afv_frame *f, const afv_proj_queries &q, afv_proj_job &j) {
    j = afv_proj_job{};
    j.struct_size = sizeof(afv_proj_job);
    j.n = f->n;
    j.desc_bytes = f->desc_bytes;
    j.float_dim = f->float_dim;
    j.min_x = f->p.min_x; j.min_y = f->p.min_y; j.grid_inv_w = f->inv_w; j.grid_inv_h = f->inv_h;
    j.grid_cols = f->p.grid_cols; j.grid_rows = f->p.grid_rows;
    j.occupied = q.occupied;
    j.nq = q.nq;
    j.qdesc = q.qdesc; j.qvalid = q.qvalid;
    j.qu = q.qu; j.qv = q.qv; j.qr = q.qr; j.qmin_size = q.qmin_size; j.qmax_size = q.qmax_size;
    j.qangle = q.qangle; j.qoccupies = q.qoccupies;
    j.th_high = q.th_high; j.nnratio = q.nnratio;
    j.size_tol = f->c->p.scale_factor;              // Frame.cc:73: sizeTolerance = extractor->GetScaleFactor()
    j.inv_size_tol = 1.0f / j.size_tol;             // Frame.cc:74
    j.check_orientation = q.check_orientation; j.mode = q.mode;
    j.q_ur = q.q_ur; j.q_er_max = q.q_er_max;
    return AFV_OK;
}

static int frame_match(afv_frame *f, const afv_proj_queries *caller_q, int kind, int use_inf_gate, int32_t *out, int32_t *nm) {
    if (!f || !caller_q || !out || !nm) return AFV_EINVAL;
    if (!f->has_features || !f->has_grid) return AFV_EINVAL;
    afv_ctx *c = f->c;
    return guarded(c, [&]() -> int {
        const uint32_t ss = caller_q->struct_size;
        if (ss < offsetof(afv_proj_queries, qref_table) || ss > 4 * sizeof(afv_proj_queries)) return AFV_EINVAL;
        afv_proj_queries q{};
        std::memcpy(&q, caller_q, std::min<size_t>(ss, sizeof(q)));
        if (q.nq < 0 || q.nq > 65535) return AFV_EINVAL;
        if (q.nq > 0 && q.desc_bytes != f->desc_bytes && !q.qref_table) return AFV_EINVAL;
        if (q.qref_table && f->desc_bytes != AFV_DESC_BYTES) return AFV_EUNSUPPORTED;  // the keyframe table holds 32-byte rows
        ProjFeatureSide S;
        frame_side(f, S);
        if (kind == AFV_KIND_FUSE && !use_inf_gate) S.inf = nullptr;
        HIPCHK(c, hipSetDevice(c->device));
        if (q.qref_table && q.nq > 0 && !q.qdesc) {  // MapPoint descriptors as rows of a keyframe table: gathered on the device
            S.qref_table = q.qref_table;
            S.qref_slot = q.qref_slot;
            S.qref_idx = q.qref_idx;
        }
        afv_proj_job j;
        frame_proj_job(f, q, j);
        return afv_match_projection_core(c, &j, 1, out, nm, kind, &S);
    });
}

extern "C" int afv_frame_match_projection(afv_frame *f, const afv_proj_queries *q, int32_t *assign, int32_t *nmatches) {
    return frame_match(f, q, AFV_KIND_PROJ, 0, assign, nmatches);
}
extern "C" int afv_frame_match_fuse(afv_frame *f, const afv_proj_queries *q, int use_inf_gate, int32_t *best, int32_t *nfound) {
    return frame_match(f, q, AFV_KIND_FUSE, use_inf_gate, best, nfound);
}

// SearchForInitialization(F1, F2, ...): the queries are F1's features - descriptors, angles and the octave-0 filter (:485-489) are read from
// F1's device arrays; vbPrevMatched and three constant per-query arrays (window radius, size band 0 .. F1.maxKeyPtSize) are all that travels.
extern "C" int afv_frame_match_initialization(afv_frame *f1, afv_frame *f2, const float *prev_x, const float *prev_y, float window_size, float th_low,
                                              float nnratio, int check_orientation, int32_t *match12, int32_t *nmatches) {
    if (!f1 || !f2 || !match12 || !nmatches || f1->c != f2->c || f1->desc_bytes != f2->desc_bytes) return AFV_EINVAL;
    if (!f1->has_features || !f2->has_features || !f2->has_grid) return AFV_EINVAL;
    if (f1->n > 0 && (!prev_x || !prev_y)) return AFV_EINVAL;
    afv_ctx *c = f1->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        const int n1 = f1->n;
        // F1.maxKeyPtSize = featureExtractor->GetMaxKeyPtSize() (Frame.cc:251) = the settings constant scaleFactorOrb^(nOctavesOrb - 1) =
        // 1.2^7 (FeatureExtractor.cpp:52-54), whatever the detector's own pyramid (the largest size OF THE PYRAMID differs
        // from it for non-default scale factors / level counts)
        const float max_size = powf(1.2f, float(8 - 1.0));
        std::vector<float> r((size_t)std::max(n1, 1), window_size), mn((size_t)std::max(n1, 1), 0.0f), mx((size_t)std::max(n1, 1), max_size);
        ProjFeatureSide S;
        S.fdesc = reinterpret_cast<const uint32_t *>(f2->d_desc);
        S.n = f2->n;
        S.words = f2->words;
        S.x = f2->d_x; S.y = f2->d_y; S.size = f2->d_size; S.angle = f2->d_angle;
        S.cell_ptr = f2->d_cell_ptr; S.cell_ent = f2->d_cell_ent;
        S.qdesc_dev = reinterpret_cast<const uint32_t *>(f1->d_desc);
        S.qangle_dev = f1->d_angle;
        S.qvalid_dev = f1->d_oct0;  // level1 > 0 -> skipped (:487-489); the mask was written when F1 was extracted
        afv_proj_job j{};
        j.struct_size = sizeof(afv_proj_job);
        j.n = f2->n;
        j.desc_bytes = f2->desc_bytes;
        j.float_dim = f2->float_dim;
        j.min_x = f2->p.min_x; j.min_y = f2->p.min_y; j.grid_inv_w = f2->inv_w; j.grid_inv_h = f2->inv_h;
        j.grid_cols = f2->p.grid_cols; j.grid_rows = f2->p.grid_rows;
        j.nq = n1;
        j.qu = prev_x; j.qv = prev_y; j.qr = r.data(); j.qmin_size = mn.data(); j.qmax_size = mx.data();
        j.th_high = th_low; j.nnratio = nnratio;
        j.size_tol = c->p.scale_factor; j.inv_size_tol = 1.0f / j.size_tol;
        j.check_orientation = check_orientation;
        return afv_match_projection_core(c, &j, 1, match12, nmatches, AFV_KIND_INIT, &S);
    });
}
