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
    int fdim;  // > 0: float descriptors - d1 / d2 are rows of fdim floats, the distance is L2^2 as cv::norm evaluates it; 0: binary, `words` dwords
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
    const int *cell_ptr;             // grid CSR, cell = ix * rows + iy, ascending feature index inside a cell
    const int4 *cell_ent;            // ... its entries: {feature index, x bits, y bits, keyPtsSize bits} - what GetFeaturesInArea filters on,
                                     // in ONE 16-byte load per candidate (built on the device by k_frame_grid)
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
    int pass_cap;                        // test hook: cap on the passes of the fixed-point engines (0 = their own guard)
    int stereo_gate;                     // the projection searches skip features with u_right > 0 and |q_ur - u_right| > q_er (:114-119, :1367-1372)
    int fdim;                            // > 0: float descriptors - fdesc / qdesc are rows of fdim floats, the distance is L2^2 (k_project.hip); 0: binary, `words` dwords
};

// ---------------- the device-resident Frame (k_frame.hip) ----------------
// One job of k_frame_grid: derive the per-feature arrays of a Frame from its keypoints (optional) and build the grid of
// Frame::AssignFeaturesToGrid (Frame.cc:225-240) on the device.
struct DevGridJob {
    const int *n_ptr;        // the feature count lives on the device (the extraction that runs ahead on the stream writes it) ...
    int n;                   // ... or is known to the host (n_ptr == nullptr)
    int cap;
    // phase 1 (kps != nullptr): keypoints -> mvKeysUn x / y (copy_xy), angle, keyPtsSize / Sigma2 / Inf from the octave (use_tab) or from
    // `size` as it stands, mvuRight = -1 (fill_mono)
    const afv_keypoint *kps;
    int copy_xy, use_tab, fill_mono;
    float tab_size[AFV_MAX_LEVELS], tab_sigma2[AFV_MAX_LEVELS], tab_inf[AFV_MAX_LEVELS];
    float *x, *y, *size, *angle, *sigma2, *inf, *u_right;
    uint8_t *oct0;           // [cap] octave == 0 (may be null)
    // phase 2 (cell_ptr != nullptr): the grid
    float min_x, min_y, inv_w, inv_h;
    int cols, rows;
    int *cell_ptr;           // [cols * rows + 1]
    int4 *cell_ent;          // [cap]
};

// KeyFrame::KeyFrame(Frame&) on the device (k_table_promote): the frame's arrays into a slot of the keyframe table
struct PromoteArgs {
    const uint4 *f_desc;
    const float *f_angle, *f_x, *f_y, *f_sigma2, *f_ur;
    const int *f_seg_idx;   // may be null (no FeatureVector yet)
    uint4 *t_desc;
    float *t_angle, *t_x, *t_y, *t_sigma2, *t_ur;  // geometry planes may be null (table without geometry)
    int *t_idx;             // may be null
    uint8_t *t_valid;       // may be null
    int *t_n;
    int n, cap, nkept;
};

// ---------------- BoW quantisation (k_bow.hip) ----------------
// Device image of the vocabulary tree: nodes renumbered breadth first so that the children of a node are CONSECUTIVE records; a record =
// the node's descriptor (words dwords) followed by {first child record, number of children, DBoW2 node id, rank of that id among the
// nodes of the same depth}.  One load per child brings
// everything the next level needs: the descent is one dependent memory round trip per level.
struct DevVocab {
    int k, L, nnodes, words;  // words = dwords per node descriptor
    int rec_dwords;           // words + 4
    const uint32_t *rec;      // [nnodes][rec_dwords], record 0 = the root
    const uint8_t *stopped;   // [nnodes] by DBoW2 node id: the word's weight is not > 0 (nullptr: none)
};
