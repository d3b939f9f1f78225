// afv_jobs.h - the job records the host stages for the matcher / vocabulary kernels: ONE definition for the kernels (k_match.hip,
// k_project.hip, k_bow.hip) and the runtime (afv_api.hip, afv_comm.hip) that fills them.  (Until round 4 every record was written down
// twice, kernel side and host side, and kept equal by hand.)
#pragma once
#include <stdint.h>

struct Seg {
    int s1, n1, s2, n2;  // ranges into idx1/idx2 (or identity when the idx pointer is null)
};

struct DevMatchJob {
    const uint32_t *d1;
    const uint32_t *d2;
    int n1, n2, words;  // words per descriptor (8 for ORB32)
    const Seg *segs;
    int nseg;
    const int *idx1;
    const int *idx2;
    const uint8_t *valid1;
    const uint8_t *valid2;
    const float *ang1;
    const float *ang2;
    int ang_stride;  // in floats (1 for plain arrays, 7 for afv_keypoint::angle)
    float th, ratio;
    int check_ori, mode;
    int *out;
    int *nmatches;
};

struct SegTask {
    int job, seg;
};

// ---------------- M4: SearchForTriangulation ----------------
struct DevTriJob {
    DevMatchJob m;  // valid1/valid2 = "has a map point" => skip
    const float *x1, *y1, *x2, *y2, *sigma2_2;
    float F[9];
    float ex, ey;
    const int *row_seg;  // [n1] index of the shared node holding the feature, -1 = none
    const float *u_right1, *u_right2;  // mvuRight of either keyframe (NULL: monocular)
    int only_stereo;                   // bOnlyStereo
};

// ---------------- projection-guided searches (k_project.hip) ----------------
struct DevProjJob {
    const uint32_t *fdesc;
    int n, words;
    const float *x, *y, *size, *angle;
    const uint8_t *occupied;
    const float *inf;
    float min_x, min_y, inv_w, inv_h;
    int cols, rows;
    const int *cell_ptr, *cell_idx;  // grid CSR, cell = ix * rows + iy, ascending feature index inside a cell
    int nq;
    const uint32_t *qdesc;
    const uint8_t *qvalid;
    const float *qu, *qv, *qr, *qmin, *qmax, *qangle;
    const uint8_t *qocc;
    float th, ratio, tol, inv_tol;
    int check_ori, mode;
    unsigned long long *keys;  // projection: [nq] 64-byte records (see topk_query); initialization: [nq][IK] keys
    int *ncand;                // [nq] candidates inside the window (geometry only)
    int *orilist;              // [nq][2] accepted (slot, rotation bin) pairs
    int *assign;               // [n] (projection) or [nq] (fuse, initialization)
    int *nmatches;
    const float *u_right, *q_ur, *q_er;  // stereo: mvuRight of the features, projected right coordinate / gate of the queries (NULL: mono)
    int stereo_gate;                     // the projection searches skip features with u_right > 0 and |q_ur - u_right| > q_er (:114-119, :1367-1372)
};

// ---------------- BoW quantisation (k_bow.hip) ----------------
struct DevVocab {
    int k, L, nnodes, words;  // words = dwords per node descriptor
    const int *child_ptr, *child_idx;
    const uint32_t *desc;
};
