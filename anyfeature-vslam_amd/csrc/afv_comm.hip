// afv_comm.hip — keyframe descriptor table resident in HBM and its multi-GPU replication (BASELINE.json configs[3]).
//
// Reference call shape: LoopClosing::ComputeSim3 (src/LoopClosing.cc:255-281) runs SearchByBoW(KF,KF) once per loop
// candidate returned by KeyFrameDatabase::DetectLoopCandidates (src/KeyFrameDatabase.cc:76-197); Tracking::Relocalization
// (src/Tracking.cc:1162-1182) does the same per relocalisation candidate; LocalMapping::CreateNewMapPoints
// (src/LocalMapping.cc:238-297) runs SearchForTriangulation against <= 20 neighbours.  Every call re-reads descriptors
// that are const after keyframe construction (include/KeyFrame.h:190).  Here they are uploaded ONCE into a table
// [nsets][cap][32] and batches of (a, b) jobs run against it; only pair lists go in and match vectors come out.
//
// Multi-GPU: one process per GPU; the table is replicated with ONE ncclBroadcast per array (RCCL over
// xGMI), jobs are block-partitioned (afv_shard_range), no other data-path collective exists.  RCCL is resolved at run
// time (dlopen / already-loaded copy), so libafv_hip.so carries no link dependency on it and single-GPU hosts never
// touch it.
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// hosts without the RCCL development headers: the few declarations this file needs (the library is resolved with dlopen at run time)
extern "C" {
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1 } ncclDataType_t;
}
#endif

#include <mutex>

#include "afv_runtime.h"

// ------------------------------------------------------------------------------------------------------------------
// table
// ------------------------------------------------------------------------------------------------------------------
struct afv_comm {
    afv_ctx *c = nullptr;
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0;
};

static std::mutex g_reg_mutex;
static std::vector<afv_table *> g_tables;
static std::vector<afv_comm *> g_comms;

static void table_free(afv_table *t) {
    if (!t) return;
    if (t->c) (void)hipSetDevice(t->c->device);
    void *ptrs[] = {t->d_desc, t->d_angle, t->d_n, t->d_idx, t->d_geo, t->d_valid, t->d_pairs, t->d_out, t->d_nm};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (t->h_pin) (void)hipHostFree(t->h_pin);
    if (t->ev0) (void)hipEventDestroy(t->ev0);
    if (t->ev1) (void)hipEventDestroy(t->ev1);
    delete t;
}

extern "C" int afv_table_create(afv_ctx *c, int nsets, int cap, afv_table **out) {
    if (!c || !out || nsets < 1 || cap < 1 || cap > 4096) return AFV_EINVAL;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    afv_table *t = new (std::nothrow) afv_table();
    if (!t) return AFV_ENOMEM;
    t->c = c;
    t->nsets = nsets;
    t->cap = cap;
    hipError_t e = hipMalloc(&t->d_desc, (size_t)nsets * cap * 32);
    if (e == hipSuccess) e = hipMalloc(&t->d_angle, (size_t)nsets * cap * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&t->d_n, (size_t)nsets * sizeof(int32_t));
    if (e == hipSuccess) e = afv_fill(c, t->d_n, 0, (size_t)nsets * sizeof(int32_t));
    if (e == hipSuccess) e = afv_fill(c, t->d_angle, 0, (size_t)nsets * cap * sizeof(float));
    if (e == hipSuccess) e = hipEventCreate(&t->ev0);
    if (e == hipSuccess) e = hipEventCreate(&t->ev1);
    if (e != hipSuccess) {
        c->last_error = std::string("afv_table_create: ") + hipGetErrorString(e);
        table_free(t);
        return e == hipErrorOutOfMemory ? AFV_ENOMEM : AFV_EHIP;
    }
    try {
        t->h_n.assign((size_t)nsets, 0);
        t->fv.resize((size_t)nsets);
        t->has_fv.assign((size_t)nsets, 0);
        t->has_geo.assign((size_t)nsets, 0);
        t->fv_body_on_device.assign((size_t)nsets, 0);
        std::lock_guard<std::mutex> g(g_reg_mutex);
        g_tables.push_back(t);
    } catch (...) {
        table_free(t);
        return AFV_ENOMEM;
    }
    *out = t;
    return AFV_OK;
}

extern "C" void afv_table_destroy(afv_table *t) {
    if (!t) return;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        auto it = std::find(g_tables.begin(), g_tables.end(), t);
        if (it == g_tables.end()) return;  // already released with its context
        g_tables.erase(it);
    }
    if (t->c) {
        (void)hipSetDevice(t->c->device);
        (void)hipStreamSynchronize(t->c->stream);
        (void)hipStreamSynchronize(t->c->stream2);
    }
    table_free(t);
}

extern "C" int afv_table_set(afv_table *t, int set, const uint8_t *desc32, const float *angles, int n) {
    if (!t || set < 0 || set >= t->nsets || n < 0 || n > t->cap || (n > 0 && !desc32)) return AFV_EINVAL;
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    if (n) HIPCHK(c, hipMemcpyAsync(t->d_desc + (size_t)set * t->cap * 32, desc32, (size_t)n * 32, hipMemcpyHostToDevice, c->stream));
    if (n && angles)
        HIPCHK(c, hipMemcpyAsync(t->d_angle + (size_t)set * t->cap, angles, (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    else if (n)
        HIPCHK(c, hipMemsetAsync(t->d_angle + (size_t)set * t->cap, 0, (size_t)n * sizeof(float), c->stream));
    t->h_n[set] = n;
    // a recycled slot must not inherit anything of its previous occupant: FeatureVector (indices may be out of range), "map point
    // exists" mask (unset = all valid), geometry (afv_table_match_triangulation refuses the slot until it is set again)
    t->fv[set] = HostFeatVec();
    t->has_fv[set] = 0;
    t->fv_body_on_device[set] = 0;
    t->has_geo[set] = 0;
    if (t->d_valid) HIPCHK(c, hipMemsetAsync(t->d_valid + (size_t)set * t->cap, 1, (size_t)t->cap, c->stream));
    HIPCHK(c, hipMemcpyAsync(t->d_n + set, &t->h_n[set], sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AFV_OK;
}

extern "C" int afv_table_set_featvec(afv_table *t, int set, const int32_t *node_id, const int32_t *seg_ptr, const int32_t *seg_idx,
                                     int nnodes) {
    if (!t || set < 0 || set >= t->nsets || nnodes < 0 || (nnodes > 0 && (!node_id || !seg_ptr || !seg_idx))) return AFV_EINVAL;
    afv_ctx *c = t->c;
    const int n = t->h_n[set];
    // same checks as the host-job path (validate_job): ascending ids, monotone pointers, indices in range, one node per feature
    int total = 0;
    if (nnodes > 0) {
        if (seg_ptr[0] != 0) return AFV_EINVAL;
        for (int i = 0; i < nnodes; ++i) {
            if (seg_ptr[i + 1] < seg_ptr[i]) return AFV_EINVAL;
            if (i > 0 && node_id[i] <= node_id[i - 1]) return AFV_EINVAL;
        }
        total = seg_ptr[nnodes];
        if (total > n) return AFV_EINVAL;
        for (int i = 0; i < total; ++i)
            if (seg_idx[i] < 0 || seg_idx[i] >= n) return AFV_EINVAL;
    }
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        if (!t->d_idx) HIPCHK(c, hipMalloc(&t->d_idx, (size_t)t->nsets * t->cap * sizeof(int32_t)));
        HostFeatVec &f = t->fv[set];
        f.node_id.assign(node_id, node_id + nnodes);
        f.seg_ptr.assign(seg_ptr, seg_ptr + (nnodes ? nnodes + 1 : 0));
        f.seg_idx.assign(seg_idx, seg_idx + total);
        t->fv_body_on_device[set] = 0;
        if (total) HIPCHK(c, hipMemcpy(t->d_idx + (size_t)set * t->cap, seg_idx, (size_t)total * sizeof(int32_t), hipMemcpyHostToDevice));
        t->has_fv[set] = 1;
        return AFV_OK;
    });
}

extern "C" int afv_table_set_geometry(afv_table *t, int set, const float *x, const float *y, const float *sigma2) {
    if (!t || set < 0 || set >= t->nsets || !x || !y || !sigma2) return AFV_EINVAL;
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t plane = (size_t)t->nsets * t->cap;
    if (!t->d_geo) HIPCHK(c, hipMalloc(&t->d_geo, 4 * plane * sizeof(float)));
    const int n = t->h_n[set];
    t->has_geo[set] = 1;
    if (n == 0) return AFV_OK;
    const std::vector<float> mono((size_t)n, -1.0f);  // a keyframe is monocular until afv_table_set_u_right says otherwise
    const float *src[4] = {x, y, sigma2, mono.data()};
    for (int k = 0; k < 4; ++k)
        HIPCHK(c, hipMemcpy(t->d_geo + k * plane + (size_t)set * t->cap, src[k], (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return AFV_OK;
}

extern "C" int afv_table_set_u_right(afv_table *t, int set, const float *u_right) {
    if (!t || set < 0 || set >= t->nsets || !u_right) return AFV_EINVAL;
    if (!t->d_geo || !t->has_geo[set]) return AFV_EINVAL;  // after afv_table_set_geometry of the same slot
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t plane = (size_t)t->nsets * t->cap;
    const int n = t->h_n[set];
    if (n > 0) HIPCHK(c, hipMemcpy(t->d_geo + 3 * plane + (size_t)set * t->cap, u_right, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    return AFV_OK;
}

extern "C" int afv_table_set_valid(afv_table *t, int set, const uint8_t *valid) {
    if (!t || set < 0 || set >= t->nsets) return AFV_EINVAL;
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    if (!t->d_valid) {
        if (!valid) return AFV_OK;  // nothing was ever restricted
        HIPCHK(c, hipMalloc(&t->d_valid, (size_t)t->nsets * t->cap));
        HIPCHK(c, afv_fill(c, t->d_valid, 1, (size_t)t->nsets * t->cap));
    }
    const int n = t->h_n[set];
    if (valid) {
        if (n) HIPCHK(c, hipMemcpy(t->d_valid + (size_t)set * t->cap, valid, (size_t)n, hipMemcpyHostToDevice));
    } else {
        HIPCHK(c, afv_fill(c, t->d_valid + (size_t)set * t->cap, 1, (size_t)t->cap));
    }
    return AFV_OK;
}

extern "C" int afv_table_device_ptrs(afv_table *t, uint8_t **d_desc, float **d_angle, int32_t **d_n) {
    if (!t) return AFV_EINVAL;
    if (d_desc) *d_desc = t->d_desc;
    if (d_angle) *d_angle = t->d_angle;
    if (d_n) *d_n = t->d_n;
    return AFV_OK;
}

extern "C" int afv_table_sync_counts(afv_table *t) {  // after a broadcast / external device-side write of d_n
    if (!t) return AFV_EINVAL;
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<int32_t> fresh((size_t)t->nsets);
    HIPCHK(c, hipMemcpy(fresh.data(), t->d_n, (size_t)t->nsets * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int s = 0; s < t->nsets; ++s) {
        const int v = std::min(std::max(fresh[s], 0), t->cap);
        if (v < t->h_n[s] && t->has_fv[s]) {
            // the set shrank under a stored FeatureVector: indices >= v would address rows that no longer exist
            bool stale = t->fv_body_on_device[s] != 0;  // (a promoted frame's body has no host copy to check: dropped with the shrink)
            for (int32_t i : t->fv[s].seg_idx) stale |= i >= v;
            if (stale) {
                t->fv[s] = HostFeatVec();
                t->has_fv[s] = 0;
                t->fv_body_on_device[s] = 0;
            }
        }
        t->h_n[s] = v;
    }
    return AFV_OK;
}

// ---- replica image: everything a replica needs besides the device planes (host-side FeatureVector structure + per-set flags).
// Layout (int32): [nsets] then per set {has_fv, has_geo, nnodes, node_id[nnodes], seg_ptr[nnodes + 1] (absent when nnodes == 0)}.
// The feature indices themselves travel with the d_idx plane. ----
static void table_pack_meta(const afv_table *t, std::vector<int32_t> &blob) {
    blob.clear();
    blob.push_back(t->nsets);
    for (int s = 0; s < t->nsets; ++s) {
        const HostFeatVec &f = t->fv[s];
        blob.push_back(t->has_fv[s]);
        blob.push_back(t->has_geo[s]);
        blob.push_back((int32_t)f.node_id.size());
        blob.insert(blob.end(), f.node_id.begin(), f.node_id.end());
        blob.insert(blob.end(), f.seg_ptr.begin(), f.seg_ptr.end());
    }
}

// rebuilds fv[] / flags of `t` from a replica image; the d_idx plane and the counts (h_n) must already be in place
static int table_unpack_meta(afv_table *t, const int32_t *blob, size_t len) {
    afv_ctx *c = t->c;
    if (len < 1 || blob[0] != t->nsets) return AFV_EINVAL;
    size_t pos = 1;
    std::vector<int32_t> idx_row((size_t)t->cap);
    for (int s = 0; s < t->nsets; ++s) {
        if (pos + 3 > len) return AFV_EINVAL;
        const int has_fv = blob[pos], has_geo = blob[pos + 1], nnodes = blob[pos + 2];
        pos += 3;
        if (nnodes < 0 || pos + (size_t)nnodes + (nnodes ? (size_t)nnodes + 1 : 0) > len) return AFV_EINVAL;
        HostFeatVec f;
        f.node_id.assign(blob + pos, blob + pos + nnodes);
        pos += (size_t)nnodes;
        if (nnodes) {
            f.seg_ptr.assign(blob + pos, blob + pos + nnodes + 1);
            pos += (size_t)nnodes + 1;
            const int total = f.seg_ptr[nnodes];
            if (total < 0 || total > t->h_n[s] || !t->d_idx) return AFV_EINVAL;
            if (total) {
                HIPCHK(c, hipMemcpy(idx_row.data(), t->d_idx + (size_t)s * t->cap, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
                for (int i = 0; i < total; ++i)
                    if (idx_row[i] < 0 || idx_row[i] >= t->h_n[s]) return AFV_EINVAL;
                f.seg_idx.assign(idx_row.begin(), idx_row.begin() + total);
            }
        }
        t->fv[s] = std::move(f);
        t->fv_body_on_device[s] = 0;
        t->has_fv[s] = has_fv != 0;
        t->has_geo[s] = has_geo != 0;
    }
    return pos == len ? AFV_OK : AFV_EINVAL;
}

extern "C" int afv_table_clone(const afv_table *src, afv_table *dst) {
    if (!src || !dst || src == dst || src->nsets != dst->nsets || src->cap != dst->cap) return AFV_EINVAL;
    afv_ctx *c = dst->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(src->c->device));
        HIPCHK(c, hipStreamSynchronize(src->c->stream));
        HIPCHK(c, hipSetDevice(c->device));
        const size_t plane = (size_t)dst->nsets * dst->cap;
        if (src->d_idx && !dst->d_idx) HIPCHK(c, hipMalloc(&dst->d_idx, plane * sizeof(int32_t)));
        if (src->d_geo && !dst->d_geo) HIPCHK(c, hipMalloc(&dst->d_geo, 4 * plane * sizeof(float)));
        if (src->d_valid && !dst->d_valid) HIPCHK(c, hipMalloc(&dst->d_valid, plane));
        HIPCHK(c, afv_copy_dd(c, dst->d_desc, src->d_desc, plane * 32));
        HIPCHK(c, afv_copy_dd(c, dst->d_angle, src->d_angle, plane * sizeof(float)));
        HIPCHK(c, afv_copy_dd(c, dst->d_n, src->d_n, (size_t)dst->nsets * sizeof(int32_t)));
        if (src->d_idx) HIPCHK(c, afv_copy_dd(c, dst->d_idx, src->d_idx, plane * sizeof(int32_t)));
        if (src->d_geo) HIPCHK(c, afv_copy_dd(c, dst->d_geo, src->d_geo, 4 * plane * sizeof(float)));
        if (src->d_valid) HIPCHK(c, afv_copy_dd(c, dst->d_valid, src->d_valid, plane));
        else if (dst->d_valid) HIPCHK(c, afv_fill(c, dst->d_valid, 1, plane));
        for (int s = 0; s < dst->nsets; ++s) {  // nothing of the destination's previous content survives
            dst->fv[s] = HostFeatVec();
            dst->has_fv[s] = dst->has_geo[s] = 0;
            dst->fv_body_on_device[s] = 0;
        }
        int rc = afv_table_sync_counts(dst);
        if (rc) return rc;
        std::vector<int32_t> blob;
        table_pack_meta(src, blob);
        return table_unpack_meta(dst, blob.data(), blob.size());  // the code path every receiver of afv_table_broadcast runs
    });
}

static int table_reserve_pairs(afv_table *t, int npairs) {
    afv_ctx *c = t->c;
    if (npairs <= t->pair_cap) return AFV_OK;
    HIPCHK(c, hipDeviceSynchronize());
    if (t->d_pairs) (void)hipFree(t->d_pairs);
    if (t->d_out) (void)hipFree(t->d_out);
    if (t->d_nm) (void)hipFree(t->d_nm);
    if (t->h_pin) (void)hipHostFree(t->h_pin);
    t->d_pairs = t->d_out = t->d_nm = nullptr;
    t->h_pin = nullptr;
    t->pair_cap = 0;
    const int want = npairs + npairs / 4;
    HIPCHK(c, hipMalloc(&t->d_pairs, (size_t)want * 2 * sizeof(int32_t)));
    HIPCHK(c, hipMalloc(&t->d_out, (size_t)want * t->cap * sizeof(int32_t)));
    HIPCHK(c, hipMalloc(&t->d_nm, (size_t)want * sizeof(int32_t)));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&t->h_pin), ((size_t)want * 3 + (size_t)want * t->cap) * sizeof(int32_t), hipHostMallocDefault));
    t->pair_cap = want;
    return AFV_OK;
}

static int check_pairs(const afv_table *t, const int32_t *pa, const int32_t *pb, int npairs) {
    for (int i = 0; i < npairs; ++i)
        if (pa[i] < 0 || pa[i] >= t->nsets || pb[i] < 0 || pb[i] >= t->nsets) return AFV_EINVAL;
    return AFV_OK;
}

extern "C" int afv_table_match_pairs_device(afv_table *t, const int32_t *d_pair_a, const int32_t *d_pair_b, int npairs, float th_low,
                                            float nnratio, int check_orientation, int32_t *d_match12, int32_t *d_nmatches, void *stream) {
    if (!t || !d_pair_a || !d_pair_b || npairs < 1 || !d_match12 || !d_nmatches) return AFV_EINVAL;
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    return afv_match_pairs_core(c, t->d_desc, t->d_angle, 1, t->d_n, t->cap, d_pair_a, d_pair_b, npairs, th_low, nnratio,
                                check_orientation, d_match12, d_nmatches, stream ? (hipStream_t)stream : c->stream);
}

extern "C" int afv_table_match_pairs(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, int npairs, float th_low, float nnratio,
                                     int check_orientation, int32_t *match12, int32_t *nmatches) {
    if (!t || !pair_a || !pair_b || npairs < 1 || !nmatches) return AFV_EINVAL;
    if (check_pairs(t, pair_a, pair_b, npairs)) return AFV_EINVAL;
    afv_ctx *c = t->c;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = table_reserve_pairs(t, npairs);
    if (rc) return rc;
    int32_t *hp = t->h_pin, *h_nm = hp + 2 * (size_t)t->pair_cap, *h_out = h_nm + t->pair_cap;
    std::memcpy(hp, pair_a, (size_t)npairs * sizeof(int32_t));
    std::memcpy(hp + t->pair_cap, pair_b, (size_t)npairs * sizeof(int32_t));
    HIPCHK(c, hipMemcpyAsync(t->d_pairs, hp, (size_t)t->pair_cap * 2 * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    rc = afv_match_pairs_core(c, t->d_desc, t->d_angle, 1, t->d_n, t->cap, t->d_pairs, t->d_pairs + t->pair_cap, npairs, th_low, nnratio,
                              check_orientation, t->d_out, t->d_nm, c->stream);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(h_nm, t->d_nm, (size_t)npairs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (match12)
        HIPCHK(c, hipMemcpyAsync(h_out, t->d_out, (size_t)npairs * t->cap * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(nmatches, h_nm, (size_t)npairs * sizeof(int32_t));
    if (match12) std::memcpy(match12, h_out, (size_t)npairs * t->cap * sizeof(int32_t));
    return afv_check_resolve_guard(c, nmatches, npairs);
}

// ---- BoW-guided / triangulation batches over the table: the kernels of k_match.hip with job records that point into
// the table; per call only the merge-joined segment lists (host work, FeatureMatcher.cc:205-276) are uploaded ----
static void join_featvecs(const HostFeatVec &A, const HostFeatVec &B, std::vector<Seg> &segs) {
    size_t a = 0, b = 0;
    while (a < A.node_id.size() && b < B.node_id.size()) {
        if (A.node_id[a] == B.node_id[b]) {
            segs.push_back(Seg{A.seg_ptr[a], A.seg_ptr[a + 1] - A.seg_ptr[a], B.seg_ptr[b], B.seg_ptr[b + 1] - B.seg_ptr[b]});
            ++a;
            ++b;
        } else if (A.node_id[a] < B.node_id[b]) {
            ++a;
        } else {
            ++b;
        }
    }
}

static int table_match_bow_impl(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, int npairs, float th_low, float nnratio,
                                int check_orientation, int32_t *match12, int32_t *nmatches) {
    afv_ctx *c = t->c;
    if (!t->d_idx) return AFV_EINVAL;  // no FeatureVector was ever stored
    for (int p = 0; p < npairs; ++p)
        for (int s : {pair_a[p], pair_b[p]})
            if (t->h_n[s] > 0 && !t->has_fv[s]) {  // "no shared node" must not be confused with "FeatureVector never stored"
                c->last_error = "afv_table_match_bow: set " + std::to_string(s) + " holds features but no FeatureVector (afv_table_set_featvec)";
                return AFV_EINVAL;
            }
    HIPCHK(c, hipSetDevice(c->device));
    const int cap = t->cap;
    Blob b(c);
    std::vector<Seg> segs;
    std::vector<SegTask> tasks;
    std::vector<int> seg_first((size_t)npairs + 1, 0);
    for (int p = 0; p < npairs; ++p) {
        const size_t before = segs.size();
        join_featvecs(t->fv[pair_a[p]], t->fv[pair_b[p]], segs);
        for (size_t s = before; s < segs.size(); ++s) tasks.push_back(SegTask{p, (int)(s - before)});
        seg_first[p + 1] = (int)segs.size();
    }
    const size_t segs_off = b.put(segs.data(), segs.size() * sizeof(Seg));
    const size_t tasks_off = b.put(tasks.data(), tasks.size() * sizeof(SegTask));
    const size_t jobs_off = b.reserve((size_t)npairs * sizeof(DevMatchJob));
    const size_t binoff_off = b.reserve((size_t)npairs * sizeof(int));
    const size_t hist_off = b.reserve((size_t)npairs * 32 * sizeof(int));
    const size_t nm_off = b.reserve((size_t)npairs * sizeof(int));
    const size_t in_bytes = b.h.size();
    const size_t out_off = b.reserve_scratch((size_t)npairs * cap * sizeof(int));
    const size_t bins_off = b.reserve_scratch((size_t)npairs * cap);
    int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    DevMatchJob *J = reinterpret_cast<DevMatchJob *>(b.h.data() + jobs_off);
    int *bin_off = reinterpret_cast<int *>(b.h.data() + binoff_off);
    for (int p = 0; p < npairs; ++p) {
        const int a = pair_a[p], bb = pair_b[p];
        DevMatchJob &d = J[p];
        d.d1 = reinterpret_cast<const uint32_t *>(t->d_desc + (size_t)a * cap * 32);
        d.d2 = reinterpret_cast<const uint32_t *>(t->d_desc + (size_t)bb * cap * 32);
        d.n1 = t->h_n[a];
        d.n2 = t->h_n[bb];
        d.words = 8;
        d.segs = reinterpret_cast<const Seg *>(c->d_match + segs_off) + seg_first[p];
        d.nseg = seg_first[p + 1] - seg_first[p];
        d.idx1 = t->d_idx + (size_t)a * cap;
        d.idx2 = t->d_idx + (size_t)bb * cap;
        d.valid1 = t->d_valid ? t->d_valid + (size_t)a * cap : nullptr;   // FeatureMatcher.cc:593-597 / :609-613
        d.valid2 = t->d_valid ? t->d_valid + (size_t)bb * cap : nullptr;
        d.ang1 = t->d_angle + (size_t)a * cap;
        d.ang2 = t->d_angle + (size_t)bb * cap;
        d.ang_stride = 1;
        d.th = th_low;
        d.ratio = nnratio;
        d.check_ori = check_orientation != 0;
        d.mode = AFV_MATCH_KF_KF;
        d.out = reinterpret_cast<int *>(c->d_match + out_off) + (size_t)p * cap;
        d.nmatches = reinterpret_cast<int *>(c->d_match + nm_off) + p;
        bin_off[p] = p * cap;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_match + out_off, 0xff, (size_t)npairs * cap * sizeof(int), c->stream));  // -1
    if (!tasks.empty())
        afv_launch_match_bow_seg(reinterpret_cast<const DevMatchJob *>(c->d_match + jobs_off), npairs, c->d_match + tasks_off,
                                 (int)tasks.size(), reinterpret_cast<int *>(c->d_match + hist_off), c->d_match + bins_off,
                                 reinterpret_cast<const int *>(c->d_match + binoff_off), check_orientation ? 1 : 0, c->stream);
    HIPCHK(c, hipGetLastError());
    if (match12) HIPCHK(c, b.fetch(match12, out_off, (size_t)npairs * cap * sizeof(int), c->stream));
    HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)npairs * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}

extern "C" int afv_table_match_bow(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, int npairs, float th_low, float nnratio,
                                   int check_orientation, int32_t *match12, int32_t *nmatches) {
    if (!t || !pair_a || !pair_b || npairs < 1 || !nmatches) return AFV_EINVAL;
    if (check_pairs(t, pair_a, pair_b, npairs)) return AFV_EINVAL;
    return guarded(t->c, [&] { return table_match_bow_impl(t, pair_a, pair_b, npairs, th_low, nnratio, check_orientation, match12, nmatches); });
}

// Relocalisation batch: SearchByBoW(KF, Frame) (FeatureMatcher.cc:186-283) of ONE frame against `nslots` candidate keyframes of the
// table (Tracking::Relocalization, Tracking.cc:1162,1182: a loop over the candidates of DetectRelocalizationCandidates).  The frame
// travels once (descriptors, angles, FeatureVector feature indices); per candidate only the merge-join of the two FeatureVectors (host,
// a few hundred ints).  M3 rules: validity on the keyframe side only (:216-222), a frame feature that already has a match is skipped
// (:232), accept best <= TH_LOW (:250), rotation histogram keyed by the frame feature (:259).
// `fr` != null: the frame side is a resident afv_frame (descriptors, angles and the FeatureVector body are on the device already; its node
// structure on the host side of the handle) and F only carries n
static int table_match_bow_frame_impl(afv_table *t, const int32_t *slots, int nslots, const afv_frame_view *F, float th_low, float nnratio,
                                      int check_orientation, int32_t *match_f, int32_t *nmatches, const afv_frame *fr = nullptr) {
    afv_ctx *c = t->c;
    if (!t->d_idx) return AFV_EINVAL;  // no FeatureVector was ever stored
    const int nf = F->n, cap = t->cap;
    for (int p = 0; p < nslots; ++p) {
        const int s = slots[p];
        if (s < 0 || s >= t->nsets) return AFV_EINVAL;
        if (t->h_n[s] > 0 && !t->has_fv[s]) {
            c->last_error = "afv_table_match_bow_frame: set " + std::to_string(s) + " holds features but no FeatureVector (afv_table_set_featvec)";
            return AFV_EINVAL;
        }
    }
    // the frame's FeatureVector: ascending node ids, indices inside the frame
    HostFeatVec FV;
    if (fr) {
        FV.node_id = fr->fv_node_id;
        FV.seg_ptr = fr->fv_seg_ptr;
    } else if (F->nnodes > 0) {
        FV.node_id.assign(F->node_id, F->node_id + F->nnodes);
        FV.seg_ptr.assign(F->seg_ptr, F->seg_ptr + F->nnodes + 1);
        const int total = FV.seg_ptr[F->nnodes];
        if (FV.seg_ptr[0] != 0 || total < 0 || total > nf) return AFV_EINVAL;
        for (int k = 0; k < F->nnodes; ++k)
            if (FV.seg_ptr[k + 1] < FV.seg_ptr[k] || (k > 0 && FV.node_id[k] <= FV.node_id[k - 1])) return AFV_EINVAL;
        for (int i = 0; i < total; ++i)
            if (F->seg_idx[i] < 0 || F->seg_idx[i] >= nf) return AFV_EINVAL;
    }
    HIPCHK(c, hipSetDevice(c->device));
    Blob b(c);
    std::vector<Seg> segs;
    std::vector<SegTask> tasks;
    std::vector<int> seg_first((size_t)nslots + 1, 0);
    for (int p = 0; p < nslots; ++p) {
        const size_t before = segs.size();
        join_featvecs(t->fv[slots[p]], FV, segs);
        for (size_t s = before; s < segs.size(); ++s) tasks.push_back(SegTask{p, (int)(s - before)});
        seg_first[p + 1] = (int)segs.size();
    }
    const int nfe = std::max(nf, 1);
    const size_t fdesc_off = fr ? 0 : b.put(F->desc32, (size_t)nf * 32);
    const size_t fang_off = (!fr && check_orientation && nf) ? b.put(F->angle, (size_t)nf * sizeof(float)) : 0;
    const size_t fidx_off = fr ? 0 : b.put(F->nnodes > 0 ? F->seg_idx : nullptr, (size_t)(F->nnodes > 0 ? FV.seg_ptr[F->nnodes] : 0) * sizeof(int32_t));
    const size_t segs_off = b.put(segs.data(), segs.size() * sizeof(Seg));
    const size_t tasks_off = b.put(tasks.data(), tasks.size() * sizeof(SegTask));
    const size_t jobs_off = b.reserve((size_t)nslots * sizeof(DevMatchJob));
    const size_t binoff_off = b.reserve((size_t)nslots * sizeof(int));
    const size_t hist_off = b.reserve((size_t)nslots * 32 * sizeof(int));
    const size_t nm_off = b.reserve((size_t)nslots * sizeof(int));
    const size_t in_bytes = b.h.size();
    const size_t out_off = b.reserve_scratch((size_t)nslots * nfe * sizeof(int));
    const size_t bins_off = b.reserve_scratch((size_t)nslots * nfe);
    int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    DevMatchJob *J = reinterpret_cast<DevMatchJob *>(b.h.data() + jobs_off);
    int *bin_off = reinterpret_cast<int *>(b.h.data() + binoff_off);
    for (int p = 0; p < nslots; ++p) {
        const int a = slots[p];
        DevMatchJob &d = J[p];
        d.d1 = reinterpret_cast<const uint32_t *>(t->d_desc + (size_t)a * cap * 32);
        d.d2 = fr ? reinterpret_cast<const uint32_t *>(fr->d_desc) : reinterpret_cast<const uint32_t *>(c->d_match + fdesc_off);
        d.n1 = t->h_n[a];
        d.n2 = nf;
        d.words = 8;
        d.segs = reinterpret_cast<const Seg *>(c->d_match + segs_off) + seg_first[p];
        d.nseg = seg_first[p + 1] - seg_first[p];
        d.idx1 = t->d_idx + (size_t)a * cap;
        d.idx2 = fr ? fr->d_seg_idx : reinterpret_cast<const int *>(c->d_match + fidx_off);
        d.valid1 = t->d_valid ? t->d_valid + (size_t)a * cap : nullptr;  // FeatureMatcher.cc:216-222
        d.valid2 = nullptr;
        d.ang1 = t->d_angle + (size_t)a * cap;
        d.ang2 = fr ? fr->d_angle : reinterpret_cast<const float *>(c->d_match + fang_off);
        d.ang_stride = 1;
        d.th = th_low;
        d.ratio = nnratio;
        d.check_ori = check_orientation != 0;
        d.mode = AFV_MATCH_KF_FRAME;
        d.out = reinterpret_cast<int *>(c->d_match + out_off) + (size_t)p * nfe;
        d.nmatches = reinterpret_cast<int *>(c->d_match + nm_off) + p;
        bin_off[p] = p * nfe;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_match + out_off, 0xff, (size_t)nslots * nfe * sizeof(int), c->stream));  // -1
    if (!tasks.empty())
        afv_launch_match_bow_seg(reinterpret_cast<const DevMatchJob *>(c->d_match + jobs_off), nslots, c->d_match + tasks_off,
                                 (int)tasks.size(), reinterpret_cast<int *>(c->d_match + hist_off), c->d_match + bins_off,
                                 reinterpret_cast<const int *>(c->d_match + binoff_off), check_orientation ? 1 : 0, c->stream);
    HIPCHK(c, hipGetLastError());
    if (match_f && nf) HIPCHK(c, b.fetch(match_f, out_off, (size_t)nslots * nf * sizeof(int), c->stream));
    HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)nslots * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}

extern "C" int afv_table_match_bow_frame(afv_table *t, const int32_t *slots, int nslots, const afv_frame_view *frame, float th_low,
                                         float nnratio, int check_orientation, int32_t *match_f, int32_t *nmatches) {
    if (!t || !slots || nslots < 1 || !frame || !nmatches) return AFV_EINVAL;
    if (frame->n < 0 || frame->n > AFV_MAX_SIDE || (frame->n > 0 && !frame->desc32) || frame->nnodes < 0) return AFV_EINVAL;
    if (frame->nnodes > 0 && (!frame->node_id || !frame->seg_ptr || !frame->seg_idx)) return AFV_EINVAL;
    if (check_orientation && frame->n > 0 && !frame->angle) return AFV_EINVAL;
    return guarded(t->c, [&] { return table_match_bow_frame_impl(t, slots, nslots, frame, th_low, nnratio, check_orientation, match_f, nmatches); });
}

extern "C" int afv_table_match_bow_frame_h(afv_table *t, const int32_t *slots, int nslots, afv_frame *f, float th_low, float nnratio,
                                           int check_orientation, int32_t *match_f, int32_t *nmatches) {
    if (!t || !slots || nslots < 1 || !f || !nmatches) return AFV_EINVAL;
    if (f->c != t->c || !f->has_features || !f->has_fv) return AFV_EINVAL;  // afv_frame_bow_transform first
    if (f->desc_bytes != AFV_DESC_BYTES) return AFV_EUNSUPPORTED;           // the table holds 32-byte rows
    afv_frame_view view{};
    view.n = f->n;
    return guarded(t->c, [&] { return table_match_bow_frame_impl(t, slots, nslots, &view, th_low, nnratio, check_orientation, match_f, nmatches, f); });
}

// KeyFrame::KeyFrame(Frame &F, ...) (src/KeyFrame.cc:36-60) on the device: one kernel copies the frame's arrays into the slot's rows of the
// table planes; the FeatureVector's node structure goes host to host
extern "C" int afv_table_set_from_frame(afv_table *t, int slot, afv_frame *f) {
    if (!t || !f || slot < 0 || slot >= t->nsets || f->c != t->c || !f->has_features) return AFV_EINVAL;
    if (f->desc_bytes != AFV_DESC_BYTES) return AFV_EUNSUPPORTED;  // the table holds 32-byte rows
    if (f->n > t->cap) return AFV_ECAPACITY;
    afv_ctx *c = t->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        const size_t plane = (size_t)t->nsets * t->cap;
        if (!t->d_geo) {  // a promoted frame brings its geometry: the planes exist from the first promotion on
            HIPCHK(c, hipMalloc(&t->d_geo, 4 * plane * sizeof(float)));
        }
        if (f->has_fv && !t->d_idx) HIPCHK(c, hipMalloc(&t->d_idx, plane * sizeof(int32_t)));
        PromoteArgs A{};
        A.f_desc = reinterpret_cast<const uint4 *>(f->d_desc);
        A.f_angle = f->d_angle; A.f_x = f->d_x; A.f_y = f->d_y; A.f_sigma2 = f->d_sigma2; A.f_ur = f->d_ur;
        A.f_seg_idx = f->has_fv ? f->d_seg_idx : nullptr;
        A.t_desc = reinterpret_cast<uint4 *>(t->d_desc + (size_t)slot * t->cap * 32);
        A.t_angle = t->d_angle + (size_t)slot * t->cap;
        A.t_x = t->d_geo + (size_t)slot * t->cap;
        A.t_y = t->d_geo + plane + (size_t)slot * t->cap;
        A.t_sigma2 = t->d_geo + 2 * plane + (size_t)slot * t->cap;
        A.t_ur = t->d_geo + 3 * plane + (size_t)slot * t->cap;
        A.t_idx = (f->has_fv && t->d_idx) ? t->d_idx + (size_t)slot * t->cap : nullptr;
        A.t_valid = t->d_valid ? t->d_valid + (size_t)slot * t->cap : nullptr;
        A.t_n = t->d_n + slot;
        A.n = f->n; A.cap = t->cap; A.nkept = f->has_fv ? f->fv_total : 0;
        afv_launch_table_promote(&A, f->n, t->cap, c->stream);
        HIPCHK(c, hipGetLastError());
        t->h_n[slot] = f->n;
        t->fv[slot] = HostFeatVec();
        t->has_fv[slot] = 0;
        if (f->has_fv) {
            t->fv[slot].node_id = f->fv_node_id;
            t->fv[slot].seg_ptr = f->fv_seg_ptr;
            // the body lives on the device only; the host copy is fetched on demand by the few paths that read it (triangulation row map)
            t->fv[slot].seg_idx.clear();
            t->fv_body_on_device[slot] = 1;
            t->has_fv[slot] = 1;
        }
        t->has_geo[slot] = 1;
        return AFV_OK;
    });
}

// host copy of a slot's FeatureVector body when it was promoted from a frame (device to device): fetched once, on first need
static int table_fetch_fv_body(afv_table *t, int slot) {
    if (!t->fv_body_on_device[slot]) return AFV_OK;
    afv_ctx *c = t->c;
    HostFeatVec &f = t->fv[slot];
    const int total = f.seg_ptr.empty() ? 0 : f.seg_ptr.back();
    f.seg_idx.assign((size_t)total, 0);
    if (total) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(f.seg_idx.data(), t->d_idx + (size_t)slot * t->cap, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
    t->fv_body_on_device[slot] = 0;
    return AFV_OK;
}

static int table_match_tri_impl(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, const afv_table_tri_job *caller_geo, int npairs,
                                int32_t *match12, int32_t *nmatches) {
    afv_ctx *c = t->c;
    std::vector<afv_table_tri_job> loaded;
    if (!afv_load_jobs(caller_geo, npairs, offsetof(afv_table_tri_job, only_stereo), loaded)) return AFV_EINVAL;
    const afv_table_tri_job *geo = loaded.data();
    for (int p = 0; p < npairs; ++p) {
        if (geo[p].only_stereo != 0 && geo[p].only_stereo != 1) return AFV_EINVAL;
        const int rcf = table_fetch_fv_body(t, pair_a[p]);
        if (rcf) return rcf;
    }
    if (!t->d_idx || !t->d_geo) return AFV_EINVAL;
    for (int p = 0; p < npairs; ++p)
        for (int s : {pair_a[p], pair_b[p]})
            if (t->h_n[s] > 0 && (!t->has_fv[s] || !t->has_geo[s])) {
                c->last_error = "afv_table_match_triangulation: set " + std::to_string(s) + " lacks its FeatureVector or geometry";
                return AFV_EINVAL;
            }
    HIPCHK(c, hipSetDevice(c->device));
    const int cap = t->cap;
    const size_t plane = (size_t)t->nsets * cap;
    Blob b(c);
    std::vector<Seg> segs;
    std::vector<int> seg_first((size_t)npairs + 1, 0);
    std::vector<size_t> rowseg_off((size_t)npairs), m1_off((size_t)npairs, 0), m2_off((size_t)npairs, 0);
    std::vector<int> row_seg;
    for (int p = 0; p < npairs; ++p) {
        const HostFeatVec &A = t->fv[pair_a[p]];
        const size_t before = segs.size();
        join_featvecs(A, t->fv[pair_b[p]], segs);
        seg_first[p + 1] = (int)segs.size();
        const int n1 = t->h_n[pair_a[p]], n2 = t->h_n[pair_b[p]];
        row_seg.assign((size_t)std::max(n1, 1), -1);  // feature -> shared node (a feature sits in exactly one node)
        for (size_t s = before; s < segs.size(); ++s)
            for (int r = 0; r < segs[s].n1; ++r) {
                const int fi = A.seg_idx[segs[s].s1 + r];
                if (fi < 0 || fi >= n1) return AFV_EINVAL;  // cannot happen while afv_table_sync_counts drops stale FeatureVectors
                row_seg[fi] = (int)(s - before);
            }
        rowseg_off[p] = b.put(row_seg.data(), row_seg.size() * sizeof(int));
        if (geo[p].has_mp1 && n1) m1_off[p] = b.put(geo[p].has_mp1, (size_t)n1);
        if (geo[p].has_mp2 && n2) m2_off[p] = b.put(geo[p].has_mp2, (size_t)n2);
    }
    const size_t segs_off = b.put(segs.data(), segs.size() * sizeof(Seg));
    const size_t jobs_off = b.reserve((size_t)npairs * sizeof(DevTriJob));
    const size_t nm_off = b.reserve((size_t)npairs * sizeof(int));
    const size_t in_bytes = b.h.size();
    const size_t out_off = b.reserve_scratch((size_t)npairs * cap * sizeof(int));
    int rc = ensure_match_buffer(c, b.h.size());
    if (rc) return rc;
    DevTriJob *J = reinterpret_cast<DevTriJob *>(b.h.data() + jobs_off);
    int max_n1 = 0;
    for (int p = 0; p < npairs; ++p) {
        const int a = pair_a[p], bb = pair_b[p];
        DevTriJob &T = J[p];
        DevMatchJob &d = T.m;
        d.d1 = reinterpret_cast<const uint32_t *>(t->d_desc + (size_t)a * cap * 32);
        d.d2 = reinterpret_cast<const uint32_t *>(t->d_desc + (size_t)bb * cap * 32);
        d.n1 = t->h_n[a];
        d.n2 = t->h_n[bb];
        max_n1 = std::max(max_n1, d.n1);
        d.words = 8;
        d.segs = reinterpret_cast<const Seg *>(c->d_match + segs_off) + seg_first[p];
        d.nseg = seg_first[p + 1] - seg_first[p];
        d.idx1 = t->d_idx + (size_t)a * cap;
        d.idx2 = t->d_idx + (size_t)bb * cap;
        d.valid1 = (geo[p].has_mp1 && d.n1) ? c->d_match + m1_off[p] : nullptr;
        d.valid2 = (geo[p].has_mp2 && d.n2) ? c->d_match + m2_off[p] : nullptr;
        d.ang1 = d.ang2 = nullptr;
        d.ang_stride = 1;
        d.th = geo[p].th_low;
        d.ratio = 0.f;
        d.check_ori = 0;
        d.mode = AFV_MATCH_KF_KF;
        d.out = reinterpret_cast<int *>(c->d_match + out_off) + (size_t)p * cap;
        d.nmatches = reinterpret_cast<int *>(c->d_match + nm_off) + p;
        T.x1 = t->d_geo + (size_t)a * cap;
        T.y1 = t->d_geo + plane + (size_t)a * cap;
        T.x2 = t->d_geo + (size_t)bb * cap;
        T.y2 = t->d_geo + plane + (size_t)bb * cap;
        T.sigma2_2 = t->d_geo + 2 * plane + (size_t)bb * cap;
        std::memcpy(T.F, geo[p].F12, sizeof(T.F));
        T.ex = geo[p].ex;
        T.ey = geo[p].ey;
        T.row_seg = reinterpret_cast<const int *>(c->d_match + rowseg_off[p]);
        T.u_right1 = t->d_geo + 3 * plane + (size_t)a * cap;  // -1 everywhere for a monocular keyframe (afv_table_set_geometry)
        T.u_right2 = t->d_geo + 3 * plane + (size_t)bb * cap;
        T.only_stereo = geo[p].only_stereo != 0;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_match, b.h.data(), in_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_match + out_off, 0xff, (size_t)npairs * cap * sizeof(int), c->stream));
    afv_launch_match_tri(reinterpret_cast<const DevTriJob *>(c->d_match + jobs_off), npairs, max_n1, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, b.fetch(match12, out_off, (size_t)npairs * cap * sizeof(int), c->stream));
    HIPCHK(c, b.fetch(nmatches, nm_off, (size_t)npairs * sizeof(int), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.finish();
    return AFV_OK;
}

extern "C" int afv_table_match_triangulation(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, const afv_table_tri_job *geo,
                                             int npairs, int32_t *match12, int32_t *nmatches) {
    if (!t || !pair_a || !pair_b || !geo || npairs < 1 || !match12 || !nmatches) return AFV_EINVAL;
    if (check_pairs(t, pair_a, pair_b, npairs)) return AFV_EINVAL;
    return guarded(t->c, [&] { return table_match_tri_impl(t, pair_a, pair_b, geo, npairs, match12, nmatches); });
}

// ------------------------------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

const RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // 1. a copy some other component (PyTorch) already mapped; 2. the system RCCL
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        const char *paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : paths)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) return;
#define AFV_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name))
        AFV_SYM(GetUniqueId, "ncclGetUniqueId");
        AFV_SYM(CommInitRank, "ncclCommInitRank");
        AFV_SYM(CommDestroy, "ncclCommDestroy");
        AFV_SYM(Broadcast, "ncclBroadcast");
        AFV_SYM(AllGather, "ncclAllGather");
        AFV_SYM(GetErrorString, "ncclGetErrorString");
#undef AFV_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Broadcast && api.AllGather;
    });
    return api;
}
}  // namespace

#define NCCLCHK(ctx, call)                                                                                  \
    do {                                                                                                    \
        ncclResult_t r_ = (call);                                                                           \
        if (r_ != ncclSuccess) {                                                                            \
            (ctx)->last_error = std::string(#call) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"); \
            return AFV_EHIP;                                                                                \
        }                                                                                                   \
    } while (0)

extern "C" int afv_comm_unique_id(uint8_t id[AFV_COMM_ID_BYTES]) {
    static_assert(AFV_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) return AFV_EINVAL;
    if (!rccl().ok) return AFV_EUNSUPPORTED;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return AFV_EHIP;
    std::memcpy(id, u.internal, AFV_COMM_ID_BYTES);
    return AFV_OK;
}

extern "C" int afv_comm_create(afv_ctx *c, const uint8_t id[AFV_COMM_ID_BYTES], int nranks, int rank, afv_comm **out) {
    if (!c || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return AFV_EINVAL;
    *out = nullptr;
    if (!rccl().ok) {
        c->last_error = "RCCL (librccl.so.1) not found";
        return AFV_EUNSUPPORTED;
    }
    HIPCHK(c, hipSetDevice(c->device));
    afv_comm *m = new (std::nothrow) afv_comm();
    if (!m) return AFV_ENOMEM;
    m->c = c;
    m->nranks = nranks;
    m->rank = rank;
    ncclUniqueId u;
    std::memcpy(u.internal, id, AFV_COMM_ID_BYTES);
    const ncclResult_t r = rccl().CommInitRank(&m->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        c->last_error = std::string("ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
        delete m;
        return AFV_EHIP;
    }
    try {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        g_comms.push_back(m);
    } catch (...) {
        (void)rccl().CommDestroy(m->comm);
        delete m;
        return AFV_ENOMEM;
    }
    *out = m;
    return AFV_OK;
}

extern "C" void afv_comm_destroy(afv_comm *m) {
    if (!m) return;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        auto it = std::find(g_comms.begin(), g_comms.end(), m);
        if (it == g_comms.end()) return;
        g_comms.erase(it);
    }
    if (m->c) {
        (void)hipSetDevice(m->c->device);
        (void)hipStreamSynchronize(m->c->stream);
    }
    if (m->comm) (void)rccl().CommDestroy(m->comm);
    delete m;
}

extern "C" int afv_comm_rank(const afv_comm *m) { return m ? m->rank : AFV_EINVAL; }
extern "C" int afv_comm_size(const afv_comm *m) { return m ? m->nranks : AFV_EINVAL; }

extern "C" int afv_comm_broadcast(afv_comm *m, void *d_buf, size_t bytes, int root, void *stream) {
    if (!m || (!d_buf && bytes) || root < 0 || root >= m->nranks) return AFV_EINVAL;
    if (!bytes) return AFV_OK;
    afv_ctx *c = m->c;
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl().Broadcast(d_buf, d_buf, bytes, ncclUint8, root, m->comm, stream ? (hipStream_t)stream : c->stream));
    return AFV_OK;
}

extern "C" int afv_comm_allgather(afv_comm *m, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream) {
    if (!m || !d_send || !d_recv) return AFV_EINVAL;
    if (!bytes_per_rank) return AFV_OK;
    afv_ctx *c = m->c;
    HIPCHK(c, hipSetDevice(c->device));
    NCCLCHK(c, rccl().AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, m->comm, stream ? (hipStream_t)stream : c->stream));
    return AFV_OK;
}

extern "C" int afv_table_broadcast(afv_comm *m, afv_table *t, int root, float *elapsed_ms) {
    if (!m || !t || m->c != t->c || root < 0 || root >= m->nranks) return AFV_EINVAL;
    afv_ctx *c = t->c;
    return guarded(c, [&]() -> int {
        HIPCHK(c, hipSetDevice(c->device));
        // what the root holds: optional planes (ranks allocate them on demand so the buffers exist everywhere) and the length of its
        // replica image (host-side FeatureVector structure + per-set flags)
        std::vector<int32_t> blob;
        if (m->rank == root) table_pack_meta(t, blob);
        int32_t flags[4] = {t->d_idx != nullptr, t->d_geo != nullptr, t->d_valid != nullptr, (int32_t)blob.size()};
        int32_t *d_flags = nullptr;
        HIPCHK(c, hipMalloc(&d_flags, sizeof(flags)));
        hipError_t e = hipMemcpyAsync(d_flags, flags, sizeof(flags), hipMemcpyHostToDevice, c->stream);
        int rc = e == hipSuccess ? afv_comm_broadcast(m, d_flags, sizeof(flags), root, c->stream) : AFV_EHIP;
        if (rc == AFV_OK) e = hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream);
        if (rc == AFV_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_flags);
        if (rc) return rc;
        HIPCHK(c, e);
        if (flags[3] < 1) return AFV_EINVAL;
        const size_t plane = (size_t)t->nsets * t->cap;
        if (flags[0] && !t->d_idx) HIPCHK(c, hipMalloc(&t->d_idx, plane * sizeof(int32_t)));
        if (flags[1] && !t->d_geo) HIPCHK(c, hipMalloc(&t->d_geo, 4 * plane * sizeof(float)));
        if (flags[2] && !t->d_valid) HIPCHK(c, hipMalloc(&t->d_valid, plane));
        int32_t *d_meta = nullptr;
        HIPCHK(c, hipMalloc(&d_meta, (size_t)flags[3] * sizeof(int32_t)));
        struct Free { void *p; ~Free() { (void)hipFree(p); } } free_meta{d_meta};
        if (m->rank == root) HIPCHK(c, hipMemcpyAsync(d_meta, blob.data(), blob.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(t->ev0, c->stream));
        rc = afv_comm_broadcast(m, t->d_desc, plane * 32, root, c->stream);
        if (!rc) rc = afv_comm_broadcast(m, t->d_angle, plane * sizeof(float), root, c->stream);
        if (!rc) rc = afv_comm_broadcast(m, t->d_n, (size_t)t->nsets * sizeof(int32_t), root, c->stream);
        if (!rc && flags[0]) rc = afv_comm_broadcast(m, t->d_idx, plane * sizeof(int32_t), root, c->stream);
        if (!rc && flags[1]) rc = afv_comm_broadcast(m, t->d_geo, 4 * plane * sizeof(float), root, c->stream);
        if (!rc && flags[2]) rc = afv_comm_broadcast(m, t->d_valid, plane, root, c->stream);
        if (!rc) rc = afv_comm_broadcast(m, d_meta, (size_t)flags[3] * sizeof(int32_t), root, c->stream);
        if (rc) {
            (void)hipStreamSynchronize(c->stream);
            return rc;
        }
        HIPCHK(c, hipEventRecord(t->ev1, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (elapsed_ms) HIPCHK(c, hipEventElapsedTime(elapsed_ms, t->ev0, t->ev1));
        if (m->rank == root) return afv_table_sync_counts(t);
        // receivers: nothing of the previous content survives; counts first, then the FeatureVectors against them
        for (int s = 0; s < t->nsets; ++s) {
            t->fv[s] = HostFeatVec();
            t->has_fv[s] = t->has_geo[s] = 0;
        }
        if (!flags[2] && t->d_valid) HIPCHK(c, afv_fill(c, t->d_valid, 1, plane));
        rc = afv_table_sync_counts(t);
        if (rc) return rc;
        blob.resize((size_t)flags[3]);
        HIPCHK(c, hipMemcpy(blob.data(), d_meta, blob.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        return table_unpack_meta(t, blob.data(), blob.size());
    });
}

extern "C" void afv_shard_range(long n_units, int rank, int nranks, long *lo, long *hi) {
    if (nranks < 1) nranks = 1;
    rank = std::min(std::max(rank, 0), nranks - 1);
    const long base = n_units / nranks, rem = n_units % nranks;
    const long l = rank * base + std::min<long>(rank, rem);
    if (lo) *lo = l;
    if (hi) *hi = l + base + (rank < rem ? 1 : 0);
}

// afv_destroy: whatever the caller forgot to release dies with the context
void afv_table_release_all(afv_ctx *c) {
    std::vector<afv_table *> ts;
    std::vector<afv_comm *> ms;
    {
        std::lock_guard<std::mutex> g(g_reg_mutex);
        for (auto it = g_tables.begin(); it != g_tables.end();)
            if ((*it)->c == c) { ts.push_back(*it); it = g_tables.erase(it); } else ++it;
        for (auto it = g_comms.begin(); it != g_comms.end();)
            if ((*it)->c == c) { ms.push_back(*it); it = g_comms.erase(it); } else ++it;
    }
    for (afv_table *t : ts) table_free(t);
    for (afv_comm *m : ms) {
        if (m->comm && rccl().ok) (void)rccl().CommDestroy(m->comm);
        delete m;
    }
}
