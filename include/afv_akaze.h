/* afv_akaze.h — C-ABI of the AKAZE61 path, same library as afv_hip.h (libafv_hip.so).
 *
 * Replaces what FeatureExtractor_akaze61 (reference src/Feature_akaze61.cpp:9-61) obtains from the un-vendored libAKAZE fork
 * `fontan::akaze` (environment.yml:33): libAKAZE::AKAZE(options), Create_Nonlinear_Scale_Space, Feature_Detection,
 * Compute_Descriptors.  PARITY UNPINNED against that fork; the arithmetic is upstream libAKAZE 1.5 as restated in
 * oracle/akaze.c, and the HIP kernels are bit-exact against that restatement. */
#ifndef AFV_AKAZE_H
#define AFV_AKAZE_H
#include <stddef.h>
#include <stdint.h>

#include "afv_hip.h" /* afv_keypoint, error codes */
#ifdef __cplusplus
extern "C" {
#endif

#define AFV_AKZ_MAX_LEVELS 16
#define AFV_AKZ_MAX_FED 32

typedef struct afv_akaze afv_akaze;

/* AKAZEOptions as FeatureExtractor_akaze61's constructor fills them (Feature_akaze61.cpp:9-15) + capacities */
typedef struct {
    int32_t omax, nsublevels;       /* numOctaves / 4, numOctaves / 2 -> 2, 4 */
    float soffset, derivative_factor;
    float dthreshold, min_dthreshold; /* settings detectionTh = 0.0005 (settings/akaze61_settings.yaml:8) */
    float kcontrast_percentile;
    int32_t kcontrast_nbins;
    int32_t max_width, max_height, max_batch;
    int32_t nfeatures;              /* quadtree budget (Tracking.cc:1515-1520: 1000) */
    float scale_factor;             /* settings scaleFactor used for the per-level quotas: 1.1892 (akaze61_settings.yaml:7) */
} afv_akaze_params;

typedef struct {
    int32_t w, h, octave, sublevel, sigma_size;
    float esigma, etime;
    int32_t nsteps;
    float tau[AFV_AKZ_MAX_FED];
} afv_akaze_level;

typedef struct {
    int32_t nlevels, w, h;
    afv_akaze_level lv[AFV_AKZ_MAX_LEVELS];
    float gauss_soffset[32]; int32_t ksize_soffset;
    float gauss_one[8];      int32_t ksize_one;
} afv_akaze_plan;

enum { AFV_AKZ_LT = 0, AFV_AKZ_LSMOOTH = 1, AFV_AKZ_LX = 2, AFV_AKZ_LY = 3, AFV_AKZ_LDET = 4 };

void afv_akaze_default_params(afv_akaze_params *p);                 /* the reference's akaze61 configuration, 1280 x 720, batch 1 */
int afv_akaze_create(int device, const afv_akaze_params *p, afv_akaze **out);
void afv_akaze_destroy(afv_akaze *a);
const char *afv_akaze_last_error(const afv_akaze *a);
/* AKAZE::Allocate_Memory_Evolution + FED time steps for a w x h image (host side, no GPU work) */
int afv_akaze_plan_for(const afv_akaze_params *p, int w, int h, afv_akaze_plan *out);

/* Create_Nonlinear_Scale_Space + Compute_Determinant_Hessian_Response for a batch of gray frames (host pointers:
 * frame f starts at gray + f * frame_stride).  Results stay on the device for detection / description. */
int afv_akaze_scale_space(afv_akaze *a, const uint8_t *gray, int nframes, int w, int h, int stride, size_t frame_stride);
/* same with frames already in HBM; nothing is synchronised */
int afv_akaze_scale_space_device(afv_akaze *a, const uint8_t *d_gray, int nframes, int w, int h, int stride, size_t frame_stride);
int afv_akaze_synchronize(afv_akaze *a);

/* Feature_Detection on the scale space built by the last afv_akaze_scale_space* call: Find_Scale_Space_Extrema (ordered
 * duplicate suppression, upper-level filter) + Do_Subpixel_Refinement; keypoints stay on the device (asynchronous).
 * Capacities per frame: w * h / 8 + 64 extrema candidates per level, 131 072 candidates over all levels (one list slot each),
 * 65 535 keypoints; beyond that the getters return AFV_ECAPACITY (afv_akaze_last_error names the limit) - never a silently
 * truncated result. */
int afv_akaze_detect(afv_akaze *a);
/* keypoints of one frame in libAKAZE's output order: pt in level-0 pixels, size = 2 * esigma * derivative_factor, angle 0,
 * response = |Ldet|, octave, class_id = evolution level (what FeatureExtractor_akaze61::GetKeypointOctave reads).
 * out == NULL returns only the count. */
int afv_akaze_get_keypoints(afv_akaze *a, int frame, afv_keypoint *out, int cap, int *n_out);
/* raster-ordered local-maximum candidates of one level (index = y * w + x), for stage-by-stage tests */
int afv_akaze_get_candidates(afv_akaze *a, int frame, int level, int32_t *out_idx, int cap, int *n_out);

/* plugin tail on the detected keypoints: bucket by class_id, DistributeOctTree per level with the quotas of
 * FeatureExtractor.cpp:97-108 (filterKeypoints_notScaled, FeatureExtractor.cpp:276-284), Compute_Main_Orientation +
 * MLDB-486 descriptors, levels merged in ascending order (mergeKeypointLevels, FeatureExtractor.cpp:296-308).  Asynchronous. */
int afv_akaze_describe(afv_akaze *a);
/* final keypoints (angle in radians as libAKAZE leaves it) and N x 61 descriptors of one frame; NULL pointers skip a part */
int afv_akaze_get_features(afv_akaze *a, int frame, afv_keypoint *kps, uint8_t *desc61, int cap, int *n_out);
/* FeatureExtractor_akaze61::detectAndCompute for a batch: scale space + detect + describe + copy out.
 * kps / desc61: nframes x cap_per_frame (x 61 bytes); n_out[nframes]. */
int afv_akaze_extract(afv_akaze *a, const uint8_t *gray, int nframes, int w, int h, int stride, size_t frame_stride, afv_keypoint *kps,
                      uint8_t *desc61, int cap_per_frame, int32_t *n_out);
/* same, frames in HBM, everything stays on the device (read back with afv_akaze_get_features); asynchronous */
int afv_akaze_extract_device(afv_akaze *a, const uint8_t *d_gray, int nframes, int w, int h, int stride, size_t frame_stride);
int afv_akaze_get_quotas(const afv_akaze *a, int32_t *quota16);

/* test / inspection access (synchronises).  LX / LY have no planes on the device (the pipeline keeps them in registers): they are
 * produced for the requested frame by the pipeline's own kernel at the time of the call. */
int afv_akaze_get_plane(afv_akaze *a, int frame, int level, int which, float *out);
int afv_akaze_get_kcontrast(afv_akaze *a, int frame, float *out);
/* 1 = conductivity and every FED step as separate kernels (upstream's structure), 0 = one fused kernel per level (default);
 * both produce identical planes (tests/test_gpu_akaze.py) */
int afv_akaze_set_step_by_step(afv_akaze *a, int on);
/* The ordered duplicate suppression of Find_Scale_Space_Extrema has two engines with identical results: 0 = the loop replayed in
 * speculative rounds, 1 = a fixed point over all candidates of a frame (faster; gives up with AFV_ECAPACITY when a candidate has more
 * than 16 earlier candidates inside its radius), 2 = engine 1, and engine 0 for a batch it gives up on (default).  pass_cap > 0 bounds
 * the fixed point's passes (test hook). */
int afv_akaze_set_suppress_engine(afv_akaze *a, int mode, int pass_cap);
/* test hook: engine 1 gives up on a candidate with more than `cap` earlier in-range candidates (0 = the built-in 16) */
int afv_akaze_debug_neighbour_cap(afv_akaze *a, int cap);
/* per-stage timing like afv_profile_*: stage 0 scale space, 1 hessian */
int afv_akaze_profile_enable(afv_akaze *a, int on);
int afv_akaze_profile_read(afv_akaze *a, float *ms_scale_space, float *ms_hessian, int *launches);

#ifdef __cplusplus
}
#endif
#endif
