/* oracle/akaze.h — CPU restatement of the AKAZE61 path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED.  The reference drives `fontan::akaze`, an un-vendored fork of libAKAZE (environment.yml:33), through
 * Feature_akaze61.cpp:9-61 (AKAZEOptions: omax = numOctaves/4 = 2, nsublevels = numOctaves/2 = 4, dthreshold = detectionTh =
 * 0.0005; Create_Nonlinear_Scale_Space / Feature_Detection / Compute_Descriptors; keypoint.class_id = evolution level).  The
 * fork is absent, so this file restates upstream libAKAZE 1.5 (P. F. Alcantarilla, "Fast Explicit Diffusion for Accelerated
 * Features in Nonlinear Scale Spaces", BMVC 2013) from the published algorithm: PM-G2 conductivity, FED cycles, Scharr
 * derivatives, determinant-of-Hessian extrema, MLDB-486.  OpenCV pieces it calls (GaussianBlur, Scharr, sepFilter2D, resize
 * INTER_AREA) are restated with an explicit float operation order; every float expression below is evaluated with one rounding
 * per operator (-ffp-contract=off) and the HIP kernels use the same expressions, so GPU == oracle bit for bit. */
#ifndef AFVO_AKAZE_H
#define AFVO_AKAZE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AKZ_MAX_LEVELS 16
#define AKZ_MAX_FED 32

typedef struct {
    int32_t omax, nsublevels;      /* Feature_akaze61.cpp:12-13 */
    float soffset;                 /* 1.6 */
    float derivative_factor;       /* 1.5 */
    float dthreshold;              /* Feature_akaze61.cpp:14 (settings detectionTh) */
    float min_dthreshold;          /* 0.00001 */
    float kcontrast_percentile;    /* 0.7 */
    int32_t kcontrast_nbins;       /* 300 */
} akz_options;

typedef struct {
    int32_t w, h, octave, sublevel, sigma_size;
    float esigma, etime;
    int32_t nsteps;                /* FED cycle that produces this level from the previous one (0 for level 0) */
    float tau[AKZ_MAX_FED];
} akz_level_info;

typedef struct {
    int32_t nlevels, w, h;
    akz_level_info lv[AKZ_MAX_LEVELS];
    float gauss_soffset[32]; int32_t ksize_soffset;   /* GaussianBlur taps for sigma = soffset */
    float gauss_one[8];      int32_t ksize_one;       /* ... for sigma = 1 */
} akz_plan;

void akz_default_options(akz_options *o);
/* AKAZE::Allocate_Memory_Evolution + the FED time steps (fed_tau_by_process_time, tau_max 0.25, reordered) + Gaussian taps.
 * Host-side double/float math; the GPU side consumes the very same plan, so none of it is a parity surface. */
int akz_make_plan(const akz_options *o, int w, int h, akz_plan *p);

/* level planes of one frame, all float, row-major w x h of the level */
typedef struct {
    float *Lt, *Lsmooth, *Lx, *Ly, *Ldet;
} akz_planes;

/* Create_Nonlinear_Scale_Space: gray u8 -> Lt / Lsmooth of every level; returns the contrast factor of level 0 in *k0 */
void akz_scale_space(const akz_plan *p, const akz_options *o, const uint8_t *gray, int stride, akz_planes *lv, float *k0);
/* pieces (exposed for stage-by-stage parity tests) */
void akz_convert(const uint8_t *gray, int stride, int w, int h, float *dst);
void akz_gauss(const float *src, int w, int h, const float *taps, int ksize, float *dst);          /* BORDER_REPLICATE */
float akz_kcontrast(const float *img, int w, int h, const akz_plan *p, const akz_options *o);
void akz_halfsample(const float *src, int w, int h, float *dst, int dw, int dh);                    /* INTER_AREA, exact 2x */
void akz_flow_g2(const float *Lsmooth, int w, int h, float k, float *flow);                          /* Scharr + pm_g2 */
void akz_nld_step(const float *Lt, const float *flow, int w, int h, float tau, float *out);          /* nld_step_scalar */
/* Compute_Multiscale_Derivatives + Compute_Determinant_Hessian_Response for one level */
void akz_hessian(const float *Lsmooth, int w, int h, int sigma_size, float *Lx, float *Ly, float *Ldet);

/* ---- Feature_Detection ---- */
typedef struct { float x, y, size, angle, response; int32_t octave, class_id; } akz_keypoint; /* = cv::KeyPoint */
/* local-maximum candidates of one level in raster order (Find_Scale_Space_Extrema's inner test incl. the descriptor-border test);
 * out_xy[i] = y * w + x; returns the count (<= cap) */
int akz_level_candidates(const akz_plan *p, const akz_options *o, int level, const float *Ldet, int32_t *out_idx, int cap);
/* Find_Scale_Space_Extrema: ordered duplicate suppression across the same / lower level, then the upper-level filter.
 * Keypoints come out in kpts_aux slot order with pt in level-0 pixels, size = esigma * derivative_factor, class_id = level. */
int akz_find_extrema(const akz_plan *p, const akz_options *o, const akz_planes *lv, akz_keypoint *out, int cap);
/* Do_Subpixel_Refinement: 2x2 solve on Ldet; drops points that move more than one pixel; size *= 2; returns the new count */
int akz_subpixel(const akz_plan *p, const akz_planes *lv, akz_keypoint *kpts, int n);

/* ---- Compute_Descriptors: Compute_Main_Orientation + Get_MLDB_Full_Descriptor (486 bits = 61 bytes, 3 channels) ----
 * atanf / cos / sin are replaced by explicit double-precision algorithms shared with the HIP kernels (libm and the device
 * library do not agree in the last bit); sample coordinates are clamped to the level (upstream reads a few pixels outside
 * the image for diagonal orientations near the border rule's limit). */
float akz_get_angle(float x, float y);
void akz_main_orientation(const akz_plan *p, const akz_planes *lv, akz_keypoint *k);
void akz_mldb(const akz_plan *p, const akz_planes *lv, const akz_keypoint *k, uint8_t *desc61);
void akz_compute_descriptors(const akz_plan *p, const akz_planes *lv, akz_keypoint *kpts, int n, uint8_t *desc /* n x 61 */);

#ifdef __cplusplus
}
#endif
#endif
