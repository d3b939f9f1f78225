// akaze_api.hip — host runtime of the AKAZE61 path (include/afv_akaze.h): evolution plan, HBM layout, stage enqueue.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/afv_akaze.h"
#include "../../include/afv_hip.h"
#include "akz_jobs.h"

#define AKD_ENTRY_CAP 65535   // keypoints per frame after detection
#define AKD_SLOT_CAP 131072   // slot space of the ordered suppression: one slot per candidate (all levels of a frame)

struct afv_akaze {
    int device = 0;
    afv_akaze_params prm{};
    afv_akaze_plan plan{};  // for the current frame size
    hipStream_t stream = nullptr;
    std::string last_error;
    // HBM: per level 3 planes x max_batch frames (level 0: Lsmooth aliases Lt); scratch at level-0 size.  The first derivatives have no
    // planes (round 5): k_akz_dhess keeps them in registers, the descriptor stage evaluates them where it samples (k_akaze_desc.hip)
    float *lt[AFV_AKZ_MAX_LEVELS] = {}, *lsm[AFV_AKZ_MAX_LEVELS] = {}, *ldet[AFV_AKZ_MAX_LEVELS] = {};
    float *flow = nullptr, *pong = nullptr, *half = nullptr;
    float *d_taps = nullptr;  // gauss_soffset[32] | gauss_one[8]
    unsigned int *d_hmax = nullptr;
    int *d_hist = nullptr;
    float *d_kcontrast = nullptr;
    uint8_t *d_gray = nullptr;
    size_t gray_bytes = 0;
    int cur_w = 0, cur_h = 0, cur_frames = 0;
    // detection
    AkdParams dp{};
    AkdState ds{};
    float *d_cand_resp = nullptr;
    unsigned long long *d_mask = nullptr;  // [frame][rows][32] candidate bitmap words
    int *d_row_count = nullptr, *d_row_start = nullptr, *d_cand = nullptr, *d_cand_count = nullptr, *d_kp_count = nullptr, *d_status = nullptr;
    afv_keypoint *d_kps = nullptr;
    size_t cand_stride_max = 0, rows_stride_max = 0, grid_cells_max = 0, grid_elems_max = 0, cells_bytes = 0;
    bool have_scale_space = false, have_keypoints = false, have_descriptors = false;
    // plugin tail: quadtree filter + descriptors
    int *d_lvl_idx = nullptr, *d_sel = nullptr, *d_sel_count = nullptr, *d_out_count = nullptr;
    uint16_t *d_lvl_node = nullptr;
    afv_keypoint *d_out_kps = nullptr;
    uint8_t *d_out_desc = nullptr;
    int sel_cap = 0, out_cap = 0, qt_M = 0;
    int quota[16] = {};
    bool step_by_step = false;  // test hook: one kernel per FED step instead of the fused level kernel
    int suppress_mode = 2;      // ordered duplicate suppression: 0 = speculative rounds, 1 = fixed point, 2 = fixed point, falling back to 0
                                // for a batch in which a candidate has more than AKF_K earlier in-range candidates (or the pass bound is hit)
    bool detect_fallback = false;  // this scale space already needed the fallback
    bool profiling = false;
    hipEvent_t ev[3] = {};
    float ms_ss = 0, ms_hess = 0;
    int launches = 0;
    std::vector<void *> allocs;
    // pinned staging of afv_akaze_extract (grow-only): [frames in][status][counts][keypoints][descriptors, 64-byte rows]
    uint8_t *h_pin = nullptr;
    size_t pin_bytes = 0;
};

#define AKZ_HIPCHK(a, expr)                                                               \
    do {                                                                                  \
        const hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                           \
            (a)->last_error = std::string(#expr) + ": " + hipGetErrorString(e_);          \
            return e_ == hipErrorOutOfMemory ? AFV_ENOMEM : AFV_EHIP;                     \
        }                                                                                 \
    } while (0)

static inline int f_round(float x) { return (int)(x + 0.5f); }

// ---- FED time steps (libAKAZE fed.cpp) ----
static bool fed_is_prime(int n) {
    if (n <= 1) return false;
    if (n == 2 || n == 3 || n == 5 || n == 7) return true;
    if (n % 2 == 0 || n % 3 == 0 || n % 5 == 0 || n % 7 == 0) return false;
    const int upper = (int)(std::sqrt((double)n + 1.0));
    for (int d = 11; d <= upper; d += 2)
        if (n % d == 0) return false;
    return true;
}
static int fed_tau(float T, float tau_max, float *tau) {
    const int n = (int)(ceilf(sqrtf(3.0f * T / tau_max + 0.25f) - 0.5f - 1.0e-8f) + 0.5f);
    if (n > AFV_AKZ_MAX_FED) return -1;
    if (n <= 0) return 0;
    const float scale = 3.0f * T / (tau_max * (float)(n * (n + 1)));
    float tauh[AFV_AKZ_MAX_FED];
    const float c = 1.0f / (4.0f * (float)n + 2.0f), d = scale * tau_max / 2.0f;
    for (int k = 0; k < n; ++k) {
        const float hcos = cosf(3.14159265358979323846f * (2.0f * (float)k + 1.0f) * c);
        tauh[k] = d / (hcos * hcos);
    }
    const int kappa = n / 2;
    int prime = n + 1;
    while (!fed_is_prime(prime)) prime++;
    for (int k = 0, l = 0; l < n; ++k, ++l) {
        int index;
        while ((index = ((k + 1) * kappa) % prime - 1) >= n) k++;
        tau[l] = tauh[index];
    }
    return n;
}
static int gauss_ksize(float sigma) {
    int ks = (int)ceilf(2.0f * (1.0f + (sigma - 0.8f) / 0.3f));
    if ((ks % 2) == 0) ks += 1;
    return ks;
}
static void gauss_taps(float sigma, int n, float *k) {  // cv::getGaussianKernel(n, sigma, CV_32F), sigma > 0
    const double s = (double)sigma, scale2x = -0.5 / (s * s);
    double t[64], sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        t[i] = std::exp(scale2x * x * x);
        sum += t[i];
    }
    for (int i = 0; i < n; ++i) k[i] = (float)(t[i] / sum);
}

extern "C" void afv_akaze_default_params(afv_akaze_params *p) {
    if (!p) return;
    p->omax = 2; p->nsublevels = 4; p->soffset = 1.6f; p->derivative_factor = 1.5f;
    p->dthreshold = 0.0005f; p->min_dthreshold = 0.00001f; p->kcontrast_percentile = 0.7f; p->kcontrast_nbins = 300;
    p->max_width = 1280; p->max_height = 720; p->max_batch = 1;
    p->nfeatures = 1000; p->scale_factor = 1.1892f;  // Tracking.cc:1515-1520, settings/akaze61_settings.yaml:7
}

extern "C" int afv_akaze_plan_for(const afv_akaze_params *o, int w, int h, afv_akaze_plan *p) {
    if (!o || !p || w < 16 || h < 16 || o->omax < 1 || o->nsublevels < 1) return AFV_EINVAL;
    std::memset(p, 0, sizeof *p);
    p->w = w; p->h = h;
    int n = 0;
    for (int i = 0; i < o->omax; ++i) {
        const float rfactor = 1.0f / powf(2.0f, (float)i);
        const int lh = (int)((float)h * rfactor), lw = (int)((float)w * rfactor);
        if ((lw < 80 || lh < 40) && i != 0) break;
        for (int j = 0; j < o->nsublevels; ++j) {
            if (n >= AFV_AKZ_MAX_LEVELS) return AFV_EINVAL;
            afv_akaze_level &L = p->lv[n++];
            L.w = lw; L.h = lh; L.octave = i; L.sublevel = j;
            L.esigma = o->soffset * powf(2.0f, (float)j / (float)o->nsublevels + (float)i);
            L.etime = 0.5f * (L.esigma * L.esigma);
            L.sigma_size = f_round(L.esigma * o->derivative_factor / powf(2.0f, (float)i));
            if (L.sigma_size < 2 || L.sigma_size > 8) return AFV_EUNSUPPORTED;  // sparse-tap Scharr path (k_akz_deriv1)
        }
    }
    p->nlevels = n;
    for (int i = 1; i < n; ++i) {
        const int ns = fed_tau(p->lv[i].etime - p->lv[i - 1].etime, 0.25f, p->lv[i].tau);
        if (ns < 0) return AFV_EUNSUPPORTED;
        p->lv[i].nsteps = ns;
        if (p->lv[i].octave > p->lv[i - 1].octave && ((p->lv[i - 1].w & 1) || (p->lv[i - 1].h & 1))) return AFV_EUNSUPPORTED;  // exact 2x INTER_AREA only
    }
    p->ksize_soffset = gauss_ksize(o->soffset);
    p->ksize_one = gauss_ksize(1.0f);
    if (p->ksize_soffset > 13 || p->ksize_one > 7) return AFV_EUNSUPPORTED;
    gauss_taps(o->soffset, p->ksize_soffset, p->gauss_soffset);
    gauss_taps(1.0f, p->ksize_one, p->gauss_one);
    return AFV_OK;
}

extern "C" const char *afv_akaze_last_error(const afv_akaze *a) { return a ? a->last_error.c_str() : "null context"; }

extern "C" void afv_akaze_destroy(afv_akaze *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    if (a->stream) (void)hipStreamSynchronize(a->stream);
    for (void *p : a->allocs)
        if (p) (void)hipFree(p);
    if (a->d_gray) (void)hipFree(a->d_gray);
    if (a->h_pin) (void)hipHostFree(a->h_pin);
    for (hipEvent_t e : a->ev)
        if (e) (void)hipEventDestroy(e);
    if (a->stream) (void)hipStreamDestroy(a->stream);
    delete a;
}

template <class T>
static int akz_alloc(afv_akaze *a, T **p, size_t count) {
    void *v = nullptr;
    AKZ_HIPCHK(a, hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)));
    a->allocs.push_back(v);
    *p = static_cast<T *>(v);
    return AFV_OK;
}

// Grids of the ordered suppression (k_akaze_detect.hip, AkdLevel): level c's entries are searched with the radii of levels c-1
// (upper-level filter), c and c+1, so its cell edge is twice the largest of those (a disc then overlaps at most 2 x 2 cells);
// a list can never hold more elements than there are strict 3 x 3 maxima of level c inside one cell.
static int akz_grid_geometry(const afv_akaze_plan &P, float derivative_factor, AkdLevel *lv, int *gcells, int *gelems, int *lds_bytes) {
    int cells = 0, elems = 0, lds = 0, prev_cells = 0;
    for (int i = 0; i < P.nlevels; ++i) {
        const float r = P.lv[i].esigma * derivative_factor;
        const float rn = i + 1 < P.nlevels ? P.lv[i + 1].esigma * derivative_factor : r;
        const float edge = 2.0f * std::max(r, rn) * 1.0002f + 0.01f;
        const float ratio = powf(2.0f, (float)P.lv[i].octave);
        AkdLevel &L = lv[i];
        L.ginv = 1.0f / edge;
        L.gw = (int)((float)P.w * L.ginv) + 1;
        L.gh = (int)((float)P.h * L.ginv) + 1;
        const int side = (int)(edge / ratio) + 2;  // pixel positions of this level along one cell edge
        L.gcap = ((side + 1) / 2) * ((side + 1) / 2);
        if (L.gw >= 65536 || L.gh >= 16384 || L.gcap > 255) return AFV_EUNSUPPORTED;  // akd_box packing, u8 list lengths
        L.gcell_off = cells;
        L.gelem_off = elems;
        const int nc = L.gw * L.gh;
        lds = std::max(lds, ((nc + 15) & ~15) + ((prev_cells + 15) & ~15));
        prev_cells = nc;
        cells += nc;
        elems += nc * L.gcap;
    }
    *gcells = cells; *gelems = elems; *lds_bytes = lds;
    return AFV_OK;
}

extern "C" int afv_akaze_create(int device, const afv_akaze_params *prm, afv_akaze **out) {
    if (!prm || !out) return AFV_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AFV_ENODEV;  // no CPU fallback: fail loudly
    if (device < 0 || device >= ndev) return AFV_ENODEV;
    if (prm->max_batch < 1 || prm->max_width < 80 || prm->max_height < 40 || prm->nfeatures < 1 || prm->nfeatures > 8000 ||
        !(prm->scale_factor > 1.0f))
        return AFV_EINVAL;
    afv_akaze_plan plan;
    int rc = afv_akaze_plan_for(prm, prm->max_width, prm->max_height, &plan);
    if (rc) return rc;
    afv_akaze *a = new (std::nothrow) afv_akaze();
    if (!a) return AFV_ENOMEM;
    a->device = device;
    a->prm = *prm;
    a->plan = plan;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) != hipSuccess) {
        delete a;
        return AFV_EHIP;
    }
    const size_t B = (size_t)prm->max_batch;
    for (int i = 0; i < plan.nlevels && rc == AFV_OK; ++i) {
        const size_t n = (size_t)plan.lv[i].w * plan.lv[i].h * B;
        rc = akz_alloc(a, &a->lt[i], n);
        if (rc == AFV_OK) {
            if (i == 0) a->lsm[0] = a->lt[0];  // evolution_[0].Lt.copyTo(evolution_[0].Lsmooth)
            else rc = akz_alloc(a, &a->lsm[i], n);
        }
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ldet[i], n);
    }
    const size_t n0 = (size_t)prm->max_width * prm->max_height * B;
    if (rc == AFV_OK) rc = akz_alloc(a, &a->flow, n0);
    if (rc == AFV_OK) rc = akz_alloc(a, &a->pong, n0);
    if (rc == AFV_OK) rc = akz_alloc(a, &a->half, n0 / 4 + B);
    if (rc == AFV_OK) rc = akz_alloc(a, &a->d_taps, 40);
    // maxima and histograms of the contrast percentile in one block: one memset per call clears both
    if (rc == AFV_OK) rc = akz_alloc(a, &a->d_hmax, B + B * (size_t)(prm->kcontrast_nbins + 1));
    if (rc == AFV_OK) a->d_hist = reinterpret_cast<int *>(a->d_hmax + B);
    if (rc == AFV_OK) rc = akz_alloc(a, &a->d_kcontrast, B);
    {   // detection buffers sized for the largest frame
        size_t cands = 0, rows = 0;
        for (int i = 0; i < plan.nlevels; ++i) {
            cands += (size_t)plan.lv[i].w * plan.lv[i].h / 8 + 64;
            rows += (size_t)plan.lv[i].h;
        }
        a->cand_stride_max = cands;
        a->rows_stride_max = rows;
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_mask, rows * 32 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_row_start, rows * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_cand, cands * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_cand_resp, cands * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_cand_count, 16 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_kp_count, B);

        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_kps, (size_t)AKD_ENTRY_CAP * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.entry, (size_t)AKD_SLOT_CAP * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.keep, (size_t)AKD_SLOT_CAP * B);
        // one cell grid per level (the levels of a frame are suppressed as a pipeline), sized for the largest frame
        // one cell grid per level (the levels of a frame are suppressed as a pipeline), sized for the largest frame
        AkdLevel glv[16];
        int gcells = 0, gelems = 0, glds = 0;
        if (rc == AFV_OK) rc = akz_grid_geometry(plan, prm->derivative_factor, glv, &gcells, &gelems, &glds);
        a->grid_cells_max = (size_t)gcells;
        a->grid_elems_max = (size_t)gelems;
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.cells, (size_t)gelems * B);
        a->cells_bytes = (size_t)gelems * B * sizeof(uint4);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.gcnt, (size_t)gcells * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.ticket, (size_t)8 + 16 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.used, (size_t)16 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.chunk_cnt, (size_t)(AKD_SLOT_CAP / 1024) * B);
        // fixed-point engine
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.fp_nbr, (size_t)AKD_SLOT_CAP * AKF_K * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.fp_state, (size_t)AKD_SLOT_CAP * 2 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.fp_active, (size_t)AKD_SLOT_CAP * 2 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.fp_ctl, (size_t)AKF_CTL * B + 1);  // + the status word: one memset per detection
        if (rc == AFV_OK) a->d_status = a->ds.fp_ctl + (size_t)AKF_CTL * B;
        if (rc == AFV_OK) rc = akz_alloc(a, &a->ds.wpre, rows * 32 * B);
    }
    {   // quadtree quotas (FeatureExtractor.cpp:97-108) for nfeatures / scaleFactor / nlevels of the akaze61 settings
        const int nl = plan.nlevels;
        const float factor = 1.0f / prm->scale_factor;
        float desired = (float)prm->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nl));
        int sum = 0, qmax = 0;
        for (int l = 0; l < nl - 1; ++l) {
            a->quota[l] = (int)lrintf(desired);  // cvRound
            sum += a->quota[l];
            desired *= factor;
        }
        a->quota[nl - 1] = std::max(prm->nfeatures - sum, 0);
        for (int l = 0; l < nl; ++l) qmax = std::max(qmax, a->quota[l]);
        a->sel_cap = qmax + 3;
        a->out_cap = prm->nfeatures + 3 * nl;
        a->qt_M = (std::max(qmax + 8, 4 * 16 + 8) + 63) / 64 * 64;
        if (afv_akz_select_lds_bytes(a->qt_M) > 150 * 1024) rc = AFV_EUNSUPPORTED;
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_lvl_idx, (size_t)nl * AKD_ENTRY_CAP * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_lvl_node, (size_t)nl * AKD_ENTRY_CAP * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_sel, (size_t)nl * a->sel_cap * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_sel_count, 16 * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_out_count, B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_out_kps, (size_t)a->out_cap * B);
        if (rc == AFV_OK) rc = akz_alloc(a, &a->d_out_desc, (size_t)a->out_cap * 64 * B);
    }
    for (hipEvent_t &e : a->ev)
        if (rc == AFV_OK && hipEventCreate(&e) != hipSuccess) rc = AFV_EHIP;
    if (rc != AFV_OK) {
        afv_akaze_destroy(a);
        return rc;
    }
    *out = a;
    return AFV_OK;
}

static int akz_enqueue(afv_akaze *a, const uint8_t *d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    if (w != a->cur_w || h != a->cur_h) {
        afv_akaze_plan plan;
        const int rc = afv_akaze_plan_for(&a->prm, w, h, &plan);
        if (rc) return rc;
        a->plan = plan;
        a->cur_w = w;
        a->cur_h = h;
        float taps[40] = {};
        std::memcpy(taps, plan.gauss_soffset, sizeof(float) * 32);
        std::memcpy(taps + 32, plan.gauss_one, sizeof(float) * 8);
        AKZ_HIPCHK(a, hipMemcpyAsync(a->d_taps, taps, sizeof taps, hipMemcpyHostToDevice, a->stream));
        AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));  // `taps` is a stack buffer
    }
    a->cur_frames = nframes;
    a->have_scale_space = true;
    a->have_keypoints = false;
    a->detect_fallback = false;
    const afv_akaze_plan &P = a->plan;
    hipStream_t st = a->stream;
    const int nb = a->prm.kcontrast_nbins;
    if (a->profiling) AKZ_HIPCHK(a, hipEventRecord(a->ev[0], st));
    // level 0: Lt = GaussianBlur(convert(gray), soffset); contrast factor from the sigma = 1 smoothed image
    if (afv_akz_launch_gauss(d_gray, 1, stride, frame_stride, w, h, nframes, a->d_taps, P.ksize_soffset, a->lt[0], st)) return AFV_EUNSUPPORTED;
    // the sigma = 1 image only feeds the gradient magnitude: with the usual 5-tap Gaussian it is formed inside that kernel
    const bool fused_contrast = !a->step_by_step && P.ksize_one == 5;
    if (!fused_contrast && afv_akz_launch_gauss(d_gray, 1, stride, frame_stride, w, h, nframes, a->d_taps + 32, P.ksize_one, a->pong, st))
        return AFV_EUNSUPPORTED;
    AKZ_HIPCHK(a, hipMemsetAsync(a->d_hmax, 0, ((size_t)a->prm.max_batch + (size_t)nframes * (nb + 1)) * sizeof(int), st));  // d_hist follows d_hmax
    afv_akz_launch_kcontrast(fused_contrast ? nullptr : a->pong, d_gray, stride, (size_t)frame_stride, a->d_taps + 32, w, h, nframes, a->flow, a->d_hmax,
                             a->d_hist, nb, a->prm.kcontrast_percentile, a->d_kcontrast, st);
    for (int i = 1; i < P.nlevels; ++i) {
        const afv_akaze_level &L = P.lv[i], &Q = P.lv[i - 1];
        const float *src = a->lt[i - 1];
        if (L.octave > Q.octave) {
            afv_akz_launch_halfsample(a->lt[i - 1], Q.w, Q.h, a->half, L.w, L.h, nframes, st);
            src = a->half;
        }
        // Lsmooth + conductivity + the whole FED cycle of this level in one kernel (the usual case: 5-tap Gaussian, <= AKZ_FED_MAX steps)
        if (!a->step_by_step && P.ksize_one == 5 &&
            afv_akz_launch_fed_gauss(src, a->lsm[i], a->d_taps + 32, L.w, L.h, nframes, a->d_kcontrast, L.octave, L.nsteps, L.tau, a->lt[i], st))
            continue;
        // step by step (the test reference, longer FED cycles, degenerate sizes): Gaussian, conductivity, one kernel per FED step
        if (afv_akz_launch_gauss(src, 0, L.w, (size_t)L.w * L.h, L.w, L.h, nframes, a->d_taps + 32, P.ksize_one, a->lsm[i], st))
            return AFV_EUNSUPPORTED;
        afv_akz_launch_flow(a->lsm[i], L.w, L.h, nframes, a->d_kcontrast, L.octave, a->flow, st);
        // FED cycle, ping-pong so that the last step lands in Lt of this level
        const float *cur = src;
        for (int j = 0; j < L.nsteps; ++j) {
            float *dst = ((L.nsteps - 1 - j) % 2 == 0) ? a->lt[i] : a->pong;
            afv_akz_launch_nld_step(cur, a->flow, L.w, L.h, nframes, L.tau[j], dst, st);
            cur = dst;
        }
        if (L.nsteps == 0) AKZ_HIPCHK(a, hipMemcpyAsync(a->lt[i], src, (size_t)L.w * L.h * nframes * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (a->profiling) AKZ_HIPCHK(a, hipEventRecord(a->ev[1], st));
    for (int i = 0; i < P.nlevels; ++i) {
        const afv_akaze_level &L = P.lv[i];
        // the first derivatives stay in the kernel's registers; the two-kernel form (test reference, other sigma sizes) hands them over
        // through the scratch planes, which are free once the scale space is built
        const bool fused = !a->step_by_step && L.sigma_size >= 2 && L.sigma_size <= 4;
        if (afv_akz_launch_hessian(a->lsm[i], L.w, L.h, nframes, L.sigma_size, fused ? 0 : 1, fused ? nullptr : a->pong, fused ? nullptr : a->flow,
                                   a->ldet[i], st))
            return AFV_EUNSUPPORTED;
    }
    if (a->profiling) {
        AKZ_HIPCHK(a, hipEventRecord(a->ev[2], st));
        AKZ_HIPCHK(a, hipEventSynchronize(a->ev[2]));
        float m0 = 0, m1 = 0;
        AKZ_HIPCHK(a, hipEventElapsedTime(&m0, a->ev[0], a->ev[1]));
        AKZ_HIPCHK(a, hipEventElapsedTime(&m1, a->ev[1], a->ev[2]));
        a->ms_ss += m0;
        a->ms_hess += m1;
        a->launches += 1;
    }
    AKZ_HIPCHK(a, hipGetLastError());
    return AFV_OK;
}

static int akz_check(afv_akaze *a, const uint8_t *gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    if (!a || !gray || nframes < 1 || nframes > a->prm.max_batch || w < 80 || h < 40 || w > a->prm.max_width || h > a->prm.max_height ||
        stride < w || (nframes > 1 && frame_stride < (size_t)stride * (h - 1) + w) || (size_t)w * h > (size_t)a->prm.max_width * a->prm.max_height)
        return AFV_EINVAL;
    return AFV_OK;
}

extern "C" int afv_akaze_scale_space_device(afv_akaze *a, const uint8_t *d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    const int rc = akz_check(a, d_gray, nframes, w, h, stride, frame_stride);
    if (rc) return rc;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    return akz_enqueue(a, d_gray, nframes, w, h, stride, frame_stride);
}

extern "C" int afv_akaze_scale_space(afv_akaze *a, const uint8_t *gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    int rc = akz_check(a, gray, nframes, w, h, stride, frame_stride);
    if (rc) return rc;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    const size_t need = (size_t)nframes * w * h;
    if (need > a->gray_bytes) {
        AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
        if (a->d_gray) (void)hipFree(a->d_gray);
        a->d_gray = nullptr;
        a->gray_bytes = 0;
        AKZ_HIPCHK(a, hipMalloc(reinterpret_cast<void **>(&a->d_gray), need));
        a->gray_bytes = need;
    }
    for (int f = 0; f < nframes; ++f)
        AKZ_HIPCHK(a, hipMemcpy2DAsync(a->d_gray + (size_t)f * w * h, (size_t)w, gray + (size_t)f * frame_stride, (size_t)stride, (size_t)w,
                                       (size_t)h, hipMemcpyHostToDevice, a->stream));
    rc = akz_enqueue(a, a->d_gray, nframes, w, h, w, (size_t)w * h);
    if (rc) return rc;
    AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
    return AFV_OK;
}

extern "C" int afv_akaze_synchronize(afv_akaze *a) {
    if (!a) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
    return AFV_OK;
}

extern "C" int afv_akaze_get_plane(afv_akaze *a, int frame, int level, int which, float *out) {
    if (!a || !out || frame < 0 || frame >= a->cur_frames || level < 0 || level >= a->plan.nlevels || a->cur_w == 0) return AFV_EINVAL;
    const afv_akaze_level &L = a->plan.lv[level];
    const float *base = nullptr;
    switch (which) {
        case AFV_AKZ_LT: base = a->lt[level]; break;
        case AFV_AKZ_LSMOOTH: base = a->lsm[level]; break;
        case AFV_AKZ_LX: base = a->pong; break;  // evaluated below
        case AFV_AKZ_LY: base = a->flow; break;
        case AFV_AKZ_LDET: base = a->ldet[level]; break;
        default: return AFV_EINVAL;
    }
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    size_t off = (size_t)frame * L.w * L.h;
    if (which == AFV_AKZ_LX || which == AFV_AKZ_LY) {
        // no derivative planes exist: this frame's are produced now, by the kernel the pipeline ran (with its first-derivative stores
        // switched on; the determinant it rewrites is the one that is there), into the scratch planes
        if (afv_akz_launch_hessian(a->lsm[level] + off, L.w, L.h, 1, L.sigma_size, a->step_by_step ? 1 : 0, a->pong, a->flow, a->ldet[level] + off,
                                   a->stream))
            return AFV_EUNSUPPORTED;
        off = 0;
    }
    AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
    AKZ_HIPCHK(a, hipMemcpy(out, base + off, (size_t)L.w * L.h * sizeof(float), hipMemcpyDeviceToHost));
    if (which == AFV_AKZ_LX || which == AFV_AKZ_LY) {  // the device planes are unscaled: Lx *= sigma_size as upstream does in place
        const float fs = (float)L.sigma_size;
        for (size_t i = 0, n = (size_t)L.w * L.h; i < n; ++i) out[i] = out[i] * fs;
    }
    return AFV_OK;
}

extern "C" int afv_akaze_get_kcontrast(afv_akaze *a, int frame, float *out) {
    if (!a || !out || frame < 0 || frame >= a->cur_frames) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
    AKZ_HIPCHK(a, hipMemcpy(out, a->d_kcontrast + frame, sizeof(float), hipMemcpyDeviceToHost));
    return AFV_OK;
}

extern "C" int afv_akaze_profile_enable(afv_akaze *a, int on) {
    if (!a) return AFV_EINVAL;
    a->profiling = on != 0;
    a->ms_ss = a->ms_hess = 0;
    a->launches = 0;
    return AFV_OK;
}

extern "C" int afv_akaze_profile_read(afv_akaze *a, float *ms_scale_space, float *ms_hessian, int *launches) {
    if (!a) return AFV_EINVAL;
    if (ms_scale_space) *ms_scale_space = a->ms_ss;
    if (ms_hessian) *ms_hessian = a->ms_hess;
    if (launches) *launches = a->launches;
    return AFV_OK;
}

// ---- Feature_Detection ----
static int akz_detect_enqueue(afv_akaze *a) {
    if (!a->have_scale_space) return AFV_EINVAL;
    const afv_akaze_plan &P = a->plan;
    AkdParams &D = a->dp;
    D = AkdParams{};
    D.nlevels = P.nlevels; D.W = P.w; D.H = P.h;
    D.dthreshold = a->prm.dthreshold; D.min_dthreshold = a->prm.min_dthreshold;
    int coff = 0, roff = 0;
    for (int i = 0; i < P.nlevels; ++i) {
        AkdLevel &L = D.lv[i];
        L.w = P.lv[i].w; L.h = P.lv[i].h; L.octave = P.lv[i].octave;
        L.psize = P.lv[i].esigma * a->prm.derivative_factor;
        L.ratio = powf(2.0f, (float)P.lv[i].octave);
        L.sigma_size = (int)(L.psize / L.ratio + 0.5f);
        L.ldet = a->ldet[i];
        L.cand_off = coff; L.cand_cap = L.w * L.h / 8 + 64; L.row_off = roff;
        coff += L.cand_cap; roff += L.h;
    }
    D.cand_stride = coff; D.rows_stride = roff;
    if ((size_t)coff > a->cand_stride_max || (size_t)roff > a->rows_stride_max) return AFV_EINVAL;
    if (P.w > 2048) return AFV_EUNSUPPORTED;
    {
        const int rc = akz_grid_geometry(P, a->prm.derivative_factor, D.lv, &D.gcells, &D.gelems, &D.lds_bytes);
        if (rc) return rc;
        if ((size_t)D.gcells > a->grid_cells_max || (size_t)D.gelems > a->grid_elems_max || D.lds_bytes > 52 * 1024) return AFV_EUNSUPPORTED;  // + ~10 KB of static LDS in k_akz_suppress (256-candidate rounds): 64 KB in all
    }
    D.entry_cap = AKD_SLOT_CAP; D.kp_cap = AKD_ENTRY_CAP;
    hipStream_t st = a->stream;
    AKZ_HIPCHK(a, hipMemsetAsync(a->ds.fp_ctl, 0, ((size_t)AKF_CTL * a->prm.max_batch + 1) * sizeof(int), st));  // fixed-point control block + d_status
    // list elements carry a 14-bit launch epoch (k_akaze_detect.hip); the grids are wiped whenever it starts over
    if (a->ds.epoch == 0 || a->ds.epoch >= 0x3fffu) {
        AKZ_HIPCHK(a, hipMemsetAsync(a->ds.cells, 0, a->cells_bytes, st));
        a->ds.epoch = 0;
    }
    ++a->ds.epoch;
    const int engine = (a->suppress_mode == 0 || a->detect_fallback) ? 0 : 1;
    afv_akz_launch_candidates(&D, a->cur_frames, a->d_mask, a->d_row_start, a->ds.wpre, a->d_cand, a->d_cand_resp, a->d_cand_count, a->d_status, st);
    afv_akz_launch_suppress(&D, &a->ds, a->cur_frames, engine, a->d_mask, a->d_cand, a->d_cand_resp, a->d_cand_count, a->d_row_start, a->d_kps,
                            a->d_kp_count, a->d_status, st);
    AKZ_HIPCHK(a, hipGetLastError());
    a->have_keypoints = true;
    a->have_descriptors = false;
    return AFV_OK;
}

extern "C" int afv_akaze_detect(afv_akaze *a) {
    if (!a) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    return akz_detect_enqueue(a);
}

static const char *akz_status_text(int st) {
    return st == 1 ? "candidate capacity exceeded" : st == 2 ? "grid cell capacity exceeded" : st == 3 ? "keypoint list capacity exceeded"
           : st == 6 ? "fixed-point suppression: more than AKF_K earlier in-range candidates" : st == 7 ? "fixed-point suppression: pass bound hit"
           : "output capacity exceeded";
}
static int akz_describe_enqueue(afv_akaze *a);
static int akz_status(afv_akaze *a) {
    int st = 0;
    AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
    AKZ_HIPCHK(a, hipMemcpy(&st, a->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if ((st == 6 || st == 7) && a->suppress_mode == 2 && !a->detect_fallback) {
        // the fixed-point engine gave up on this batch: the same scale space goes through the ordered-rounds engine
        const bool desc = a->have_descriptors;
        a->detect_fallback = true;
        int rc = akz_detect_enqueue(a);
        if (rc == AFV_OK && desc) rc = akz_describe_enqueue(a);
        if (rc) return rc;
        AKZ_HIPCHK(a, hipStreamSynchronize(a->stream));
        AKZ_HIPCHK(a, hipMemcpy(&st, a->d_status, sizeof(int), hipMemcpyDeviceToHost));
    }
    if (st == 5) {  // not a capacity problem: an earlier ticket holder of the level pipeline did not move for ~1 s (k_akaze_detect.hip)
        a->last_error = "level pipeline stalled (suppression): the device was preempted or is being profiled; repeat the call";
        return AFV_ETIMEOUT;
    }
    if (st) {
        a->last_error = akz_status_text(st);
        return AFV_ECAPACITY;
    }
    return AFV_OK;
}

extern "C" int afv_akaze_get_candidates(afv_akaze *a, int frame, int level, int32_t *out_idx, int cap, int *n_out) {
    if (!a || !n_out || frame < 0 || frame >= a->cur_frames || level < 0 || level >= a->plan.nlevels || !a->have_keypoints) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    const int rc = akz_status(a);
    if (rc) return rc;
    int n = 0;
    AKZ_HIPCHK(a, hipMemcpy(&n, a->d_cand_count + frame * 16 + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (out_idx && n > 0) {
        if (n > cap) return AFV_ECAPACITY;
        AKZ_HIPCHK(a, hipMemcpy(out_idx, a->d_cand + (size_t)frame * a->dp.cand_stride + a->dp.lv[level].cand_off, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return AFV_OK;
}

extern "C" int afv_akaze_get_keypoints(afv_akaze *a, int frame, afv_keypoint *out, int cap, int *n_out) {
    if (!a || !n_out || frame < 0 || frame >= a->cur_frames || !a->have_keypoints) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    const int rc = akz_status(a);
    if (rc) return rc;
    int n = 0;
    AKZ_HIPCHK(a, hipMemcpy(&n, a->d_kp_count + frame, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (out && n > 0) {
        if (n > cap) return AFV_ECAPACITY;
        AKZ_HIPCHK(a, hipMemcpy(out, a->d_kps + (size_t)frame * AKD_ENTRY_CAP, (size_t)n * sizeof(afv_keypoint), hipMemcpyDeviceToHost));
    }
    return AFV_OK;
}

// ---- plugin tail: filterKeypoints (quadtree per level) + computeDescriptors + mergeKeypointLevels ----
static int akz_describe_enqueue(afv_akaze *a) {
    if (!a->have_keypoints) return AFV_EINVAL;
    const afv_akaze_plan &P = a->plan;
    AksParams S{};
    S.nlevels = P.nlevels; S.W = P.w; S.H = P.h;
    S.n_ini = (int)roundf((float)P.w / (float)P.h);  // ORBextractor.cc:243
    if (S.n_ini < 1 || S.n_ini > 16) return AFV_EUNSUPPORTED;
    S.h_x = (float)P.w / (float)S.n_ini;
    for (int l = 0; l < P.nlevels; ++l) S.quota[l] = a->quota[l];
    S.kp_cap = AKD_ENTRY_CAP; S.sel_cap = a->sel_cap; S.out_cap = a->out_cap; S.M = a->qt_M;
    AkdDescParams D{};
    D.nlevels = P.nlevels; D.kp_cap = AKD_ENTRY_CAP; D.sel_cap = a->sel_cap; D.out_cap = a->out_cap; D.desc_pitch = 64;
    for (int l = 0; l < P.nlevels; ++l) D.lv[l] = AkdLevelPlanes{a->lt[l], a->lsm[l], P.lv[l].w, P.lv[l].h, P.lv[l].octave, P.lv[l].sigma_size, (float)P.lv[l].sigma_size};
    hipStream_t st = a->stream;
    afv_akz_launch_select(&S, a->cur_frames, a->d_kps, a->d_kp_count, a->d_lvl_idx, a->d_lvl_node, a->d_sel, a->d_sel_count, st);
    afv_akz_launch_describe(&D, a->cur_frames, a->out_cap, a->d_kps, a->d_sel, a->d_sel_count, a->d_out_kps, a->d_out_desc, a->d_out_count,
                            a->d_status, st);
    AKZ_HIPCHK(a, hipGetLastError());
    a->have_descriptors = true;
    return AFV_OK;
}

extern "C" int afv_akaze_describe(afv_akaze *a) {
    if (!a) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    return akz_describe_enqueue(a);
}

extern "C" int afv_akaze_get_features(afv_akaze *a, int frame, afv_keypoint *kps, uint8_t *desc61, int cap, int *n_out) {
    if (!a || !n_out || frame < 0 || frame >= a->cur_frames || !a->have_descriptors) return AFV_EINVAL;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    const int rc = akz_status(a);
    if (rc) return rc;
    int n = 0;
    AKZ_HIPCHK(a, hipMemcpy(&n, a->d_out_count + frame, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > 0 && (kps || desc61)) {
        if (n > cap) return AFV_ECAPACITY;
        if (kps) AKZ_HIPCHK(a, hipMemcpy(kps, a->d_out_kps + (size_t)frame * a->out_cap, (size_t)n * sizeof(afv_keypoint), hipMemcpyDeviceToHost));
        if (desc61)
            AKZ_HIPCHK(a, hipMemcpy2D(desc61, 61, a->d_out_desc + (size_t)frame * a->out_cap * 64, 64, 61, (size_t)n, hipMemcpyDeviceToHost));
    }
    return AFV_OK;
}

// FeatureExtractor_akaze61::detectAndCompute for a batch of host frames (Feature_akaze61.cpp:17-24 after initializeExtractor).
// The plugin calls it with ONE frame (FeatureExtractor.cpp:111-121): the whole call is one upload, the kernel chain, four read-backs and ONE
// wait - frames go through a pinned arena (a pageable 1280 x 720 image costs one memcpy, not a staged 2-D copy), the results (status,
// counts, keypoints, 64-byte descriptor rows) come back into it asynchronously and are unpacked on the host.  (Until round 4 the call
// waited after the scale space, then read status / count / keypoints / descriptors with four blocking copies, the last one a 2-D copy of
// 61-byte rows into pageable memory: 6.7 ms per frame around 1.2 ms of kernels.)
static bool akz_is_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

extern "C" int afv_akaze_extract(afv_akaze *a, const uint8_t *gray, int nframes, int w, int h, int stride, size_t frame_stride,
                                 afv_keypoint *kps, uint8_t *desc61, int cap_per_frame, int32_t *n_out) {
    if (!a || !n_out || !kps || !desc61 || cap_per_frame < 1) return AFV_EINVAL;
    int rc = akz_check(a, gray, nframes, w, h, stride, frame_stride);
    if (rc) return rc;
    AKZ_HIPCHK(a, hipSetDevice(a->device));
    hipStream_t st = a->stream;
    const size_t img = (size_t)w * h, in_bytes = (size_t)nframes * img;
    if (in_bytes > a->gray_bytes) {
        AKZ_HIPCHK(a, hipStreamSynchronize(st));
        if (a->d_gray) (void)hipFree(a->d_gray);
        a->d_gray = nullptr;
        a->gray_bytes = 0;
        AKZ_HIPCHK(a, hipMalloc(reinterpret_cast<void **>(&a->d_gray), in_bytes));
        a->gray_bytes = in_bytes;
    }
    const size_t oc = (size_t)a->out_cap;
    const size_t off_st = (in_bytes + 255) & ~(size_t)255, off_n = off_st + 256, off_k = off_n + (((size_t)nframes * 4 + 255) & ~(size_t)255);
    const size_t off_d = off_k + (((size_t)nframes * oc * sizeof(afv_keypoint) + 255) & ~(size_t)255), need = off_d + (size_t)nframes * oc * 64;
    if (need > a->pin_bytes) {
        AKZ_HIPCHK(a, hipStreamSynchronize(st));
        if (a->h_pin) (void)hipHostFree(a->h_pin);
        a->h_pin = nullptr;
        a->pin_bytes = 0;
        AKZ_HIPCHK(a, hipHostMalloc(reinterpret_cast<void **>(&a->h_pin), need, hipHostMallocDefault));
        a->pin_bytes = need;
    }
    // upload: page-locked caller frames are DMA'd in place, pageable ones take one memcpy into the arena
    const bool pinned_in = akz_is_pinned(gray) && akz_is_pinned(gray + (size_t)(nframes - 1) * frame_stride + (size_t)(h - 1) * stride + w - 1);
    if (pinned_in) {
        for (int f = 0; f < nframes; ++f)
            AKZ_HIPCHK(a, hipMemcpy2DAsync(a->d_gray + (size_t)f * img, (size_t)w, gray + (size_t)f * frame_stride, (size_t)stride, (size_t)w, (size_t)h,
                                           hipMemcpyHostToDevice, st));
    } else {
        for (int f = 0; f < nframes; ++f) {
            const uint8_t *src = gray + (size_t)f * frame_stride;
            uint8_t *dst = a->h_pin + (size_t)f * img;
            if (stride == w) std::memcpy(dst, src, img);
            else
                for (int y = 0; y < h; ++y) std::memcpy(dst + (size_t)y * w, src + (size_t)y * stride, (size_t)w);
        }
        AKZ_HIPCHK(a, hipMemcpyAsync(a->d_gray, a->h_pin, in_bytes, hipMemcpyHostToDevice, st));
    }
    rc = akz_enqueue(a, a->d_gray, nframes, w, h, w, img);
    if (rc) return rc;
    // a stalled level pipeline (AFV_ETIMEOUT) is repeated, twice at most: the scale space is still there.  The re-run through the ordered
    // rounds after the fixed point gave up is NOT one of those repeats (they shared one budget of three, and a loop that ended on a
    // `continue` copied the rejected run's results out).
    bool settled = false;
    for (int timeouts = 0, fallbacks = 0; timeouts < 3 && fallbacks < 2;) {
        rc = akz_detect_enqueue(a);
        if (rc) return rc;
        rc = akz_describe_enqueue(a);
        if (rc) return rc;
        AKZ_HIPCHK(a, hipMemcpyAsync(a->h_pin + off_st, a->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
        AKZ_HIPCHK(a, hipMemcpyAsync(a->h_pin + off_n, a->d_out_count, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost, st));
        AKZ_HIPCHK(a, hipMemcpyAsync(a->h_pin + off_k, a->d_out_kps, (size_t)nframes * oc * sizeof(afv_keypoint), hipMemcpyDeviceToHost, st));
        AKZ_HIPCHK(a, hipMemcpyAsync(a->h_pin + off_d, a->d_out_desc, (size_t)nframes * oc * 64, hipMemcpyDeviceToHost, st));
        AKZ_HIPCHK(a, hipStreamSynchronize(st));
        const int status = *reinterpret_cast<const int *>(a->h_pin + off_st);
        if (status == 5) {
            a->last_error = "level pipeline stalled (suppression): the device was preempted or is being profiled; repeat the call";
            rc = AFV_ETIMEOUT;
            ++timeouts;
            continue;
        }
        if ((status == 6 || status == 7) && a->suppress_mode == 2 && !a->detect_fallback) {
            a->detect_fallback = true;  // the fixed-point engine gave up on this batch: once more through the ordered rounds
            a->last_error = akz_status_text(status);
            rc = AFV_ECAPACITY;
            ++fallbacks;
            continue;
        }
        if (status) {
            a->last_error = akz_status_text(status);
            rc = AFV_ECAPACITY;
        } else {
            rc = AFV_OK;
        }
        settled = true;
        break;
    }
    if (!settled || (rc != AFV_OK && rc != AFV_ECAPACITY)) {  // no run was accepted: nothing of the arena is a result
        for (int f = 0; f < nframes; ++f) n_out[f] = 0;
        return rc;
    }
    const int *cnt = reinterpret_cast<const int *>(a->h_pin + off_n);
    for (int f = 0; f < nframes; ++f) {
        int n = std::min(std::max(cnt[f], 0), (int)oc);
        n_out[f] = n;
        if (n > cap_per_frame) {
            n = cap_per_frame;
            if (rc == AFV_OK) rc = AFV_ECAPACITY;
        }
        std::memcpy(kps + (size_t)f * cap_per_frame, a->h_pin + off_k + (size_t)f * oc * sizeof(afv_keypoint), (size_t)n * sizeof(afv_keypoint));
        const uint8_t *rows = a->h_pin + off_d + (size_t)f * oc * 64;
        uint8_t *out = desc61 + (size_t)f * cap_per_frame * 61;
        for (int i = 0; i < n; ++i) std::memcpy(out + (size_t)i * 61, rows + (size_t)i * 64, 61);
    }
    return rc;
}

extern "C" int afv_akaze_extract_device(afv_akaze *a, const uint8_t *d_gray, int nframes, int w, int h, int stride, size_t frame_stride) {
    int rc = afv_akaze_scale_space_device(a, d_gray, nframes, w, h, stride, frame_stride);
    if (rc) return rc;
    rc = akz_detect_enqueue(a);
    if (rc) return rc;
    return akz_describe_enqueue(a);
}

extern "C" int afv_akaze_get_quotas(const afv_akaze *a, int32_t *quota16) {
    if (!a || !quota16) return AFV_EINVAL;
    for (int l = 0; l < 16; ++l) quota16[l] = a->quota[l];
    return AFV_OK;
}

// 0 = ordered speculative rounds (k_akz_suppress), 1 = fixed point (k_akz_fp_*), 2 = fixed point with fallback (default); pass_cap > 0
// bounds the fixed point's passes (test hook: 1 forces the fallback / the error)
extern "C" int afv_akaze_set_suppress_engine(afv_akaze *a, int mode, int pass_cap) {
    if (!a || mode < 0 || mode > 2) return AFV_EINVAL;
    a->suppress_mode = mode;
    a->ds.fp_pass_cap = pass_cap > 0 ? pass_cap : 0;
    return AFV_OK;
}

// test hook: the fixed-point engine gives up (status 6) on a candidate with more than `cap` earlier in-range candidates (0: the built-in 16)
extern "C" int afv_akaze_debug_neighbour_cap(afv_akaze *a, int cap) {
    if (!a || cap < 0) return AFV_EINVAL;
    a->ds.fp_nbr_cap = cap;
    return AFV_OK;
}

// test hook: 1 = run pm_g2 and every FED step as its own kernel (the reference structure), 0 = fused level kernel (default)
extern "C" int afv_akaze_set_step_by_step(afv_akaze *a, int on) {
    if (!a) return AFV_EINVAL;
    a->step_by_step = on != 0;
    return AFV_OK;
}
