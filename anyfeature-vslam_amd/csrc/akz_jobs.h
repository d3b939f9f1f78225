// akz_jobs.h - the parameter / state records of the AKAZE detection and description kernels: ONE definition for the kernels
// (k_akaze_detect.hip, k_akaze_desc.hip) and the runtime that fills them (akaze_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
};

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
    const float *lt, *lx, *ly;  // [frame][h][w]; lx / ly hold the UNSCALED first derivatives
    int w, h, octave;
    float fs;                   // sigma_size: Lx = lx * fs, Ly = ly * fs (the in-place scaling of Compute_Multiscale_Derivatives)
};
struct AkdDescParams {
    int nlevels, kp_cap, sel_cap, out_cap, desc_pitch;
    AkdLevelPlanes lv[AKS_MAX_LEVELS];
};
