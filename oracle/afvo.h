/*
 * afvo — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A dependency-free, single-threaded C restatement of the reference's default ORB32 front end
 * (AnyFeature-VSLAM, /root/reference) and of its BoW-guided / brute-force Hamming matchers.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (anyfeature-vslam_amd/) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED.  The reference has no tests and no golden vectors and cannot be
 * built here: every translation unit needs OpenCV / Eigen / DBoW2 headers that are absent from the
 * image and are not vendored (Feature_orb32.cpp:21-53 calls cv::ORB for pyramid, FAST, Harris, IC
 * angle, blur and rBRIEF).  The OpenCV stages are therefore restated from the published OpenCV 4.x
 * algorithm (modules/features2d/src/orb.cpp, fast.cpp, imgproc resize/filter) and anchored on the
 * reference's own call sites; the in-repo stages (quadtree, quotas, merge, sizes, matchers,
 * rotation histogram, Hamming distance) follow the cited reference lines.  Known-answer values that
 * CAN be derived from the reference's text are pinned in tests/test_oracle_kat.py.
 *
 * Deviations (both documented in DESIGN.md):
 *  - cv::KeyPointsFilter::retainBest leaves the survivors in std::nth_element order, which is
 *    libstdc++-specific; the oracle keeps the same SET (everything >= the k-th best response, ties
 *    kept) in raster order and breaks response ties inside a quadtree cell by raster index.
 *  - DistributeOctTree sorts (size, ExtractorNode*) pairs (ORBextractor.cc:381): ties are ordered by
 *    heap address, i.e. allocator-dependent.  The oracle orders ties by creation sequence
 *    (later-created node first), which is what a monotonically growing heap gives.
 */
#ifndef AFVO_H
#define AFVO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AFVO_MAX_LEVELS 16
#define AFVO_BORDER 23 /* cv::ORB apron: max(edgeThreshold=0, ceil(15*sqrt2)=22, 9/2)+1 */

/* bit-compatible with cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id) */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} afvo_keypoint;

typedef struct {
    int32_t nfeatures;      /* Tracking.cc:1515-1520 -> 1000 @640x480 */
    int32_t nlevels;        /* settings/orb32_settings.yaml:6  -> 8   */
    float scale_factor;     /* settings/orb32_settings.yaml:7  -> 1.2 */
    int32_t fast_threshold; /* int(detectTh) Feature_orb32.cpp:30 -> 20 */
} afvo_params;

/* FAST candidate (level coordinates) with the integer Harris sums north_star calls "integer Harris scores" */
typedef struct {
    int32_t x, y, level, fast_score;
    int32_t ha, hb, hc; /* sum Ix^2, sum Iy^2, sum IxIy over the 7x7 block */
    float response;     /* Harris response */
} afvo_candidate;

/* ---- tables / geometry ---- */
void afvo_level_geometry(int w, int h, int nlevels, float scale_factor, int *lw, int *lh, float *lscale);
void afvo_quotas_extractor(int nfeatures, int nlevels, float scale_factor, int *q); /* FeatureExtractor.cpp:97-108 */
void afvo_quotas_cvorb(int nfeatures, int nlevels, float scale_factor, int *q);     /* cv::ORB computeKeyPoints  */
void afvo_umax(int *umax16);                                                        /* 16 entries, half patch 15 */
void afvo_gauss7_taps(int *taps7);                                                  /* round(256*g), sigma 2     */
const int8_t *afvo_brief_pattern(void);                                             /* 1024 int8                */
float afvo_fast_atan2(float y, float x);
void afvo_sincos_deg(float angle_deg, float *cos_out, float *sin_out);
float afvo_keypoint_size(int octave, float scale_factor);                           /* Feature_orb32.cpp:59-61  */
void afvo_size_sigma(const afvo_keypoint *kps, int n, float scale_factor, float *size, float *sigma2, float *inf);

/* ---- image stages (all u8, row-major, explicit strides) ---- */
void afvo_resize_linear_exact(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride);
void afvo_make_border101(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int border); /* dst: (w+2b)x(h+2b), stride w+2b */
int afvo_fast9_16(const uint8_t *img, int w, int h, int stride, int threshold, int32_t *xs, int32_t *ys, int32_t *scores, int cap);
void afvo_fast_score_map(const uint8_t *img, int w, int h, int stride, int threshold, uint8_t *score); /* pre-NMS, stride w */
void afvo_harris_sums(const uint8_t *bordered, int bstride, int x, int y, int *a, int *b, int *c); /* (x,y) in un-bordered coords */
float afvo_harris_response(int a, int b, int c);
float afvo_ic_angle(const uint8_t *bordered, int bstride, int x, int y);
void afvo_gaussian_blur7(const uint8_t *bordered, int w, int h, int bstride, uint8_t *dst, int dstride); /* reads apron, writes w x h */
void afvo_brief_descriptor(const uint8_t *bordered_blurred, int bstride, int cx, int cy, float angle_deg, uint8_t *desc32);

/* ---- keypoint selection ---- */
/* keep every element whose key >= k-th largest key (ties kept), order preserved; returns new count */
int afvo_retain_best_mask(const float *resp, int n, int k, uint8_t *keep);
/* DistributeOctTree (ORBextractor.cc:239-458) over level-0 coordinates; out_idx receives indices into the input in
   list order; returns how many. */
int afvo_quadtree(const float *px, const float *py, const float *resp, const int64_t *tiebreak, int n,
                  int min_x, int max_x, int min_y, int max_y, int N, int32_t *out_idx, int cap);

/* ---- full extraction (Feature_orb32.cpp:11-18 + FeatureExtractor.cpp:111-129) ---- */
/* variant 0 = de-duplicated (8 level builds, 8 blurs); variant 1 = reference-faithful call pattern
   (1 detect pyramid + 8 compute() calls that rebuild and blur levels 0..L).  Same outputs. */
int afvo_orb_extract(const afvo_params *p, const uint8_t *gray, int w, int h, int stride, int variant,
                     afvo_keypoint *kps, uint8_t *desc32, int cap, int *n_out);
/* computeDescriptors alone (Feature_orb32.cpp:42-53 = cv::ORB::compute at caller-given keypoints) */
int afvo_orb_compute(const afvo_params *p, const uint8_t *gray, int w, int h, int stride, const afvo_keypoint *kps, int n, uint8_t *desc32);
/* intermediate products for stage-level parity tests */
typedef struct {
    int nlevels;
    int lw[AFVO_MAX_LEVELS], lh[AFVO_MAX_LEVELS];
    float lscale[AFVO_MAX_LEVELS];
    uint8_t *level[AFVO_MAX_LEVELS];   /* un-bordered, stride lw */
    uint8_t *blurred[AFVO_MAX_LEVELS]; /* un-bordered, stride lw */
    afvo_candidate *cand;              /* all FAST candidates after NMS (raster order per level, levels ascending) */
    int ncand;
    uint8_t *keep1;                    /* survived retainBest(2*quota) on FAST score */
    uint8_t *keep2;                    /* survived retainBest(quota) on Harris response */
    int t_counts[AFVO_MAX_LEVELS];     /* per-level quadtree survivors */
} afvo_trace;
int afvo_orb_extract_trace(const afvo_params *p, const uint8_t *gray, int w, int h, int stride, afvo_trace *tr,
                           afvo_keypoint *kps, uint8_t *desc32, int cap, int *n_out);
void afvo_trace_free(afvo_trace *tr);

/* ---- matching ---- */
int afvo_hamming256(const uint8_t *a, const uint8_t *b);       /* Feature_orb32.cpp:67-84 (SWAR form) */
int afvo_hamming_bytes(const uint8_t *a, const uint8_t *b, int nbytes); /* Feature_akaze61.cpp:75-77 */
float afvo_l2sqr(const float *a, const float *b, int dim);     /* Feature_sift128.cpp:132-134 */

typedef struct {
    const uint8_t *desc1; int32_t n1;
    const uint8_t *desc2; int32_t n2;
    int32_t desc_bytes;            /* 32 (ORB), 61 (AKAZE) ... */
    /* BoW node segments, CSR; nnodes==0 => brute force (one node holding 0..n-1 on both sides) */
    const int32_t *node_id1; const int32_t *seg_ptr1; const int32_t *seg_idx1; int32_t nnodes1;
    const int32_t *node_id2; const int32_t *seg_ptr2; const int32_t *seg_idx2; int32_t nnodes2;
    const uint8_t *valid1; const uint8_t *valid2; /* NULL => all valid */
    const float *angle1; const float *angle2;     /* degrees, needed iff check_orientation */
    float th_low; float nnratio; int32_t check_orientation;
    int32_t float_dim;             /* > 0: desc1 / desc2 are rows of float_dim floats, the distance is afvo_l2sqr (DescriptorDistance
                                      dispatches on DescriptorType, FeatureMatcher.cc:1508-1531); 0: binary rows of desc_bytes */
} afvo_bow_job;

/* M2: SearchByBoW(KF,KF) FeatureMatcher.cc:561-660. match12[n1] = idx2 or -1. returns nmatches */
int afvo_search_by_bow_kf_kf(const afvo_bow_job *j, int32_t *match12);
/* M3: SearchByBoW(KF,Frame) FeatureMatcher.cc:186-283. side 1 = KF, side 2 = Frame. matchF[n2] = idxKF or -1 */
int afvo_search_by_bow_kf_frame(const afvo_bow_job *j, int32_t *matchF);

typedef struct {
    afvo_bow_job bow;              /* valid1/valid2 here mean "already HAS a map point" => skip */
    const float *x1, *y1, *x2, *y2; /* mvKeysUn */
    const float *sigma2_2;          /* GetKeyPt1DSigma2 of KF2 */
    float F12[9];                   /* row-major */
    float ex, ey;                   /* epipole in image 2 */
    /* stereo (FeatureMatcher.cc:705-709, :727-731, :741): mvuRight of either keyframe (>= 0: the keypoint has a right match); NULL =
       monocular keyframe.  only_stereo = bOnlyStereo. */
    const float *u_right1, *u_right2;
    int32_t only_stereo;
} afvo_tri_job;
/* M4: SearchForTriangulation FeatureMatcher.cc:662-790. match12[n1] */
int afvo_search_for_triangulation(const afvo_tri_job *j, int32_t *match12);

/* float-descriptor brute force / BoW (M8 distance inside M2 control flow) */
typedef struct {
    const float *desc1; int32_t n1; const float *desc2; int32_t n2; int32_t dim;
    const uint8_t *valid1; const uint8_t *valid2;
    float th_low; float nnratio;
} afvo_l2_job;
int afvo_match_l2_bruteforce(const afvo_l2_job *j, int32_t *match12);

/* ---- projection-guided matching core (grid window + Hamming) ----
 * Flat restatement of the matching loops of SearchByProjection(F, localMapPoints) (FeatureMatcher.cc:73-154, mode 0) and
 * SearchByProjection(CurrentFrame, LastFrame) (:1291-1402, mode 1) over Frame::GetFeaturesInArea (Frame.cc:333-382)
 * and the 64x48 grid of Frame::AssignFeaturesToGrid / PosInGrid (Frame.cc:225-240, :383-394).  The projection itself
 * (pose * point, radius from viewing angle / keypoint size) is evaluated by the caller exactly as the reference does and
 * arrives as (u, v, r, min_size, max_size) per query. */
typedef struct {
    const uint8_t *desc; int32_t n; int32_t desc_bytes;   /* frame features */
    const float *x, *y, *size, *angle;                    /* mvKeysUn pt / keyPtsSize / mvKeysUn angle */
    const uint8_t *occupied;                              /* F.pts[i] && NumberOfObservations() > 0; NULL = none */
    const float *inf;                                     /* KeyFrame::GetKeyPt1DInf(i) (Fuse only) */
    float min_x, min_y, grid_inv_w, grid_inv_h;           /* mnMinX, mnMinY, mfGridElementWidthInv/HeightInv */
    int32_t grid_cols, grid_rows;                         /* 64, 48 */
    int32_t nq;                                           /* queries in the reference's iteration order */
    const uint8_t *qdesc; const uint8_t *qvalid;
    const float *qu, *qv, *qr, *qmin_size, *qmax_size, *qangle;
    const uint8_t *qoccupies;                             /* assigned point has observations (> 0); NULL = yes */
    float th_high, nnratio, size_tol, inv_size_tol;
    int32_t check_orientation, mode;                      /* mode 0 = local map, 1 = last frame */
    /* stereo (NULL u_right = monocular): u_right[i] = mvuRight of feature i; q_ur[q] = the query's projected right coordinate
       (pMP->mTrackProjXR :116, u - mbf * invzc :1369, ur of Fuse :885); q_er_max[q] = the gate of modes 0 / 1 (r * pMP->trackSigma
       :117, radius :1371).  Modes 0 / 1 skip a feature with u_right > 0 whose |q_ur - u_right| exceeds the gate; Fuse replaces the
       2-dof gate (5.99) by the 3-dof one (7.8) for features with u_right >= 0 (:880-894). */
    const float *u_right; const float *q_ur; const float *q_er_max;
    /* float descriptors (FeatureMatcher::DescriptorDistance dispatches on DescriptorType, FeatureMatcher.cc:1508-1531: SIFT128, SURF64,
       KAZE64, R2D2 ... = cv::norm(a, b, NORM_L2SQR), Feature_sift128.cpp:132-134): float_dim > 0 = desc / qdesc point to rows of float_dim
       floats (desc_bytes is ignored), the distance is afvo_l2sqr; 0 = binary rows of desc_bytes */
    int32_t float_dim;
} afvo_proj_job;
int afvo_match_projection(const afvo_proj_job *j, int32_t *assign /* [n]: query index or -1 */);
/* matching core of FeatureMatcher::Fuse(pKF, vpMapPoints, th) (FeatureMatcher.cc:794-940): per map point the most
 * similar keypoint in the window that lies in the predicted size band and passes the 5.99 reprojection gate; best[q] = feature
 * index or -1 (bestDist > TH_LOW).  Independent per point: the map surgery (:918-936) stays with the caller.  returns #found */
int afvo_match_fuse(const afvo_proj_job *j, int32_t *best /* [nq] */);   /* inf == NULL: no gate = Fuse(Sim3) core (:942-1064) */
/* SearchBySim3 (FeatureMatcher.cc:1066-1287): two gate-less directed searches + agreement; match12[j12->nq] */
int afvo_match_sim3(const afvo_proj_job *j12, const afvo_proj_job *j21, int32_t *match12);
/* SearchForInitialization (FeatureMatcher.cc:480-556): ordered, with distance-gated skipping and stealing; match12[nq] */
int afvo_match_initialization(const afvo_proj_job *j, int32_t *match12);

/* ---- BoW quantisation (DBoW2 TemplatedVocabulary::transform(features, v, fv, levelsup), called from
 * Vocabulary.cpp:156-206 with levelsup = 4).  DBoW2 is an empty submodule: restated from upstream DBoW2 (parity unpinned).
 * Tree in CSR form: children of node i are child_idx[child_ptr[i] .. child_ptr[i+1]) in DBoW2's order, node 0 = root,
 * a node without children is a word.  Per feature: greedy descent, first minimal Hamming distance wins (strict <);
 * word[i] = node id of the reached leaf, node_at_level[i] = node id met at depth L - levelsup (0 if that depth is <= 0 or
 * the leaf is reached earlier). */
typedef struct {
    int32_t k, L, nnodes;
    const int32_t *child_ptr, *child_idx;
    const uint8_t *desc; int32_t desc_bytes;
} afvo_vocab;
void afvo_bow_transform(const afvo_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *leaf_node, int32_t *node_at_level);
int afvo_distinctive_descriptor(const uint8_t *desc, int n, int desc_bytes, int *median_out);  /* MapPoint.cc:279-349 */
int afvo_distinctive_descriptor_f32(const float *desc, int n, int dim, float *median_out);        /* ... on float descriptors (L2^2) */
void afvo_bow_transform_f32(const afvo_vocab *v, const float *desc, int n, int levelsup, int32_t *leaf_node, int32_t *node_at_level);

/* M6 pieces, exposed for KATs */
int afvo_rotation_bin(float a1, float a2); /* FeatureMatcher.cc:1587-1599 */
void afvo_three_maxima(const int *hist_sizes, int L, int *i1, int *i2, int *i3); /* :1631-1668 */

#ifdef __cplusplus
}
#endif
#endif
