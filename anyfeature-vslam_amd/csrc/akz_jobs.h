// akz_jobs.h - the parameter / state records of the AKAZE kernels and their launchers: ONE definition / declaration for the kernels
// (k_akaze.hip, k_akaze_detect.hip, k_akaze_desc.hip) and the runtime that drives them (akaze_api.hip): a record or a signature that drifts
// is a compile error, not a silent mismatch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/afv_hip.h"  // afv_keypoint

#define AKS_MAX_LEVELS 16

struct AkdLevel {
    int w, h, octave, sigma_size;
    float psize, ratio;      // esigma * derivative_factor, 2^octave
    const float *ldet;       // [frame][h][w]
    int cand_off;            // offset of this level's candidate slice inside a frame's candidate array
    int cand_cap;
    int row_off;             // offset of this level's rows inside a frame's row-count array
    // uniform grid over the entries of this level (level-0 pixel coordinates).  The cell edge is at least twice the largest
    // radius the grid is ever searched with (this level's and the next one's), so a search disc overlaps at most 2 x 2 cells;
    // gcap = the number of strict 3 x 3 maxima that fit into a cell = the longest a cell list can get.
    float ginv;              // 1 / cell edge
    int gw, gh, gcap;
    int gcell_off, gelem_off;  // offsets of this level's cells / list elements inside a frame's arrays
};

struct AkdParams {
    int nlevels, W, H;
    float dthreshold, min_dthreshold;
    AkdLevel lv[16];
    int cand_stride;   // candidates per frame (all levels)
    int rows_stride;   // rows per frame (all levels)
    int gcells, gelems;  // cells / list elements per frame (all levels)
    int lds_bytes;       // list lengths (u8) of a level's own grid + length hints of the grid below, largest level pair
    int entry_cap, kp_cap;
};

// state of the ordered suppression (k_akz_suppress and the refinement kernels; the hand-off protocol is described in k_akaze_detect.hip)
struct AkdState {
    float4 *entry;            // [frame][entry_cap], slot-indexed: {x, y, response, level (integer bits)} - one 16-byte store per commit
    uint4 *cells;             // [frame][gelems] {x, y, response, tag}; level c's lists start at lv[c].gelem_off
    int *gcnt;                // [frame][gcells] published list lengths (hints, see above)
    int *ticket;              // [8] per-XCD ticket counters, then [frame][16] committed candidates per level (all zeroed before the launch)
    int *used;                // [frame][16] slots used per level
    int *chunk_cnt;           // [frame][AKD_CHUNKS] refined keypoints per 1024-slot chunk
    unsigned char *keep;      // [frame][entry_cap]  (set to 1 before the launch; the upper-level filter clears)
    unsigned int epoch;       // 1 .. AKD_EPOCH_MAX, changes with every launch
    // fixed-point engine (round 5, k_akz_fp_*): everything indexed by a candidate's position in upstream's loop order
    // (gid = candidates of the levels below + index inside its level), [frame][entry_cap]
    int *fp_nbr;              // x AKF_K: the EARLIER candidates (same level / level below) inside the candidate's radius
    int4 *fp_state;           // x 2: {what the candidate does: AKF_APPEND, AKF_DROP or the gid of the holder it replaces; its entry's slot = root
                              //  gid; response; -} {stamped minima of the candidates that replace it: even passes, odd passes; -; -}
    int4 *fp_active;          // x 2: the candidates with at least one such neighbour (any order): {gid, count, response, act} {root, n0, n1, n2}
    int *fp_ctl;              // [frame][AKF_CTL]: how many of those per level [16] | the pass that found the frame converged | per pass: changed something
    unsigned short *wpre;     // [frame][rows_stride][AKD_MAXCHUNKS] candidates of the row in front of the 64-column word (k_akz_cand_emit)
    int fp_pass_cap;
    int fp_nbr_cap;           // 0 = AKF_K; smaller: test hook that makes the engine give up (status 6) on ordinary frames
};
#define AKF_K 16
#define AKF_PASSES 16  // pass launches enqueued per call (the bench frames and band-limited noise need 8; a launch for a converged frame costs ~5 us)
#define AKF_CTL (17 + AKF_PASSES + 3)
#define AKF_APPEND (-1)
#define AKF_DROP (-2)
#define AKF_NONE 0x7fffffff

struct AksParams {
    int nlevels, W, H, n_ini;
    float h_x;
    int quota[AKS_MAX_LEVELS];
    int kp_cap;    // detected keypoints per frame (input stride)
    int sel_cap;   // selected per (frame, level)
    int out_cap;   // final keypoints per frame
    int M;         // quadtree node capacity
};

struct AkdLevelPlanes {
    const float *lt, *lsm;  // [frame][h][w]: Lt and Lsmooth of the level.  The first derivatives are NOT stored (round 5): the descriptor
                            // stage evaluates them from Lsmooth at its ~550 sample positions per keypoint with k_akz_deriv1's expressions
    int w, h, octave;
    int s;                  // sigma_size (tap distance of the derivative filters)
    float fs;               // (float)sigma_size: Lx = lx * fs, Ly = ly * fs (the in-place scaling of Compute_Multiscale_Derivatives)
};
struct AkdDescParams {
    int nlevels, kp_cap, sel_cap, out_cap, desc_pitch;
    AkdLevelPlanes lv[AKS_MAX_LEVELS];
};

// ---- launchers (defined next to their kernels) ----
extern "C" int afv_akz_launch_gauss(const void *src, int is_u8, int src_stride, size_t src_frame_stride, int w, int h, int nframes,
                                    const float *taps, int ksize, float *dst, hipStream_t st);
extern "C" void afv_akz_launch_kcontrast(const float *gsm, const uint8_t *gray, int src_stride, size_t src_frame_stride, const float *taps, int w,
                                         int h, int nframes, float *modg, unsigned int *hmax_bits, int *hist, int nbins, float perc,
                                         float *kcontrast, hipStream_t st);
extern "C" void afv_akz_launch_halfsample(const float *src, int w, int h, float *dst, int dw, int dh, int nframes, hipStream_t st);
extern "C" void afv_akz_launch_flow(const float *lsm, int w, int h, int nframes, const float *kcontrast, int octave, float *flow,
                                    hipStream_t st);
extern "C" void afv_akz_launch_nld_step(const float *Lt, const float *flow, int w, int h, int nframes, float tau, float *out, hipStream_t st);
extern "C" int afv_akz_launch_fed_gauss(const float *Lt_in, float *lsm, const float *taps, int w, int h, int nframes, const float *kcontrast,
                                        int octave, int nsteps, const float *tau, float *Lt_out, hipStream_t st);
extern "C" int afv_akz_launch_hessian(const float *lsm, int w, int h, int nframes, int s, int two_kernels, float *dx, float *dy,
                                      float *Ldet, hipStream_t st);

extern "C" void afv_akz_launch_candidates(const AkdParams *P, int nframes, unsigned long long *mask, int *row_start, unsigned short *wpre, int *cand,
                                          float *cand_resp, int *cand_count, int *status, hipStream_t st);
// engine: 0 = ordered speculative rounds (k_akz_suppress), 1 = fixed point (k_akz_fp_build + k_akz_fp_resolve); identical results
extern "C" void afv_akz_launch_suppress(const AkdParams *P, const AkdState *S, int nframes, int engine, const unsigned long long *mask, const int *cand,
                                        const float *cand_resp, const int *cand_count, const int *row_start, afv_keypoint *kps, int *kp_count,
                                        int *status, hipStream_t st);
extern "C" size_t afv_akz_select_lds_bytes(int M);
extern "C" void afv_akz_launch_select(const AksParams *P, int nframes, const afv_keypoint *kps, const int *kp_count, int *lvl_idx,
                                      uint16_t *lvl_node, int *sel, int *sel_count, hipStream_t st);
extern "C" void afv_akz_launch_describe(const AkdDescParams *P, int nframes, int max_out, const afv_keypoint *kps, const int *sel,
                                        const int *sel_count, afv_keypoint *out_kps, uint8_t *out_desc, int *out_count, int *status,
                                        hipStream_t st);
