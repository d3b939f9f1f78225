/*
 * afv_hip.h — C-ABI of libafv_hip.so: MI355X (gfx950) ORB32 extraction + descriptor matching.
 *
 * Drop-in boundary for AnyFeature-VSLAM's front end.  Every entry point is plain C:
 * pointers, sizes, POD structs; returns 0 on success or a negative AFV_E* code; never throws, never
 * terminates the process.  One afv_ctx owns one HIP stream and its scratch buffers: use one context per
 * calling thread (Tracking / LocalMapping / LoopClosing each call matchers concurrently in the reference,
 * FeatureMatcher.cc §3.2) — a context is NOT re-entrant, different contexts are independent.
 *
 * Reference interfaces replaced (file:line under the reference tree):
 *   afv_orb_extract*            FeatureExtractor::operator() 6/3-arg (src/FeatureExtractor.cpp:111-129) ->
 *                               FeatureExtractor_orb32::detectAndCompute (src/Feature_orb32.cpp:11-18):
 *                               detectKeypoints(:26-40, cv::ORB::detect) + filterKeypoints(:63-65 ->
 *                               DistributeOctTree src/ORBextractor.cc:239-458) + computeDescriptors(:42-53,
 *                               cv::ORB::compute per level) + mergeKeypointLevels (FeatureExtractor.cpp:296-308)
 *   afv_orb_size_sigma          computeSize / computeSigma (src/FeatureExtractor.cpp:132-172)
 *   afv_match_bow*              FeatureMatcher::SearchByBoW(KF,KF) (src/FeatureMatcher.cc:561-660) and
 *                               SearchByBoW(KF,Frame) (:186-283), incl. the rotation histogram (:1579-1668);
 *                               with no node segments = the brute-force form (north_star "Hamming brute force")
 *   afv_match_triangulation*    FeatureMatcher::SearchForTriangulation (:662-790) + CheckDistEpipolarLine (:165-182)
 *   afv_match_l2                DescriptorDistance_sift128 (src/Feature_sift128.cpp:132-134) inside the
 *                               SearchByBoW(KF,KF) control flow (config #3)
 *   afv_hamming256              DescriptorDistance_orb32 (src/Feature_orb32.cpp:67-84)
 */
#ifndef AFV_HIP_H
#define AFV_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AFV_MAX_LEVELS 8
#define AFV_DESC_BYTES 32

enum {
    AFV_OK = 0,
    AFV_EINVAL = -1,     /* bad argument (null pointer, size out of range for the context) */
    AFV_ENODEV = -2,     /* no usable HIP device / device ordinal out of range */
    AFV_ENOMEM = -3,     /* device or host allocation failed */
    AFV_EHIP = -4,       /* a HIP runtime call or kernel failed; see afv_last_error() */
    AFV_ECAPACITY = -5,  /* caller-provided output capacity too small; outputs truncated */
    AFV_EUNSUPPORTED = -6,
    AFV_ETIMEOUT = -7    /* a device-side wait gave up (a level pipeline stalled for ~1 s: preemption, a debugger, a profiler); nothing
                            about the inputs is wrong - the call can be repeated (afv_akaze_extract does so once by itself) */
};

/* ABI revision of this header.  It goes up whenever a record, an argument list or a limit changes incompatibly; a host built against
 * another revision must not call into the library (check once: afv_abi_version() == AFV_ABI_VERSION).
 *   5  (round 5) afv_proj_job / afv_tri_job / afv_table_tri_job start with struct_size; frame grids hold at most 8192 cells (was 65536)
 *   6  (round 6) afv_orb_detect / afv_orb_compute; afv_frame_params.desc_bytes; afv_vocab_create_f32 / afv_bow_transform_f32; afv_proj_job.float_dim; grids / frames whose one-workgroup build does not fit the
 *      LDS are refused with AFV_EUNSUPPORTED at creation instead of failing at the first launch */
#define AFV_ABI_VERSION 6
int afv_abi_version(void);

typedef struct afv_ctx afv_ctx;

/* bit-compatible with cv::KeyPoint {Point2f pt; float size, angle, response; int octave, class_id;} */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} afv_keypoint;

typedef struct {
    int32_t nfeatures;      /* quadtree budget, Tracking.cc:1515-1520 (1000 @ 640x480); 1..4000 */
    int32_t nlevels;        /* FeatureExtractor.numOctaves (settings/orb32_settings.yaml:6); 1..8 */
    float scale_factor;     /* FeatureExtractor.scaleFactor (:7) */
    int32_t fast_threshold; /* int(FeatureExtractor.detectionTh) (:8, Feature_orb32.cpp:30) */
    int32_t max_width;      /* largest frame the context must accept (<= 4096) */
    int32_t max_height;
    int32_t max_batch;      /* frames per afv_orb_extract_batch* call */
} afv_orb_params;

void afv_default_orb_params(afv_orb_params *p); /* ORB32 defaults: 1000, 8, 1.2f, 20, 640x480, batch 1 */

int afv_create(int device_ordinal, const afv_orb_params *params, afv_ctx **out);
void afv_destroy(afv_ctx *ctx);
const char *afv_strerror(int code);
const char *afv_last_error(const afv_ctx *ctx); /* text of the last failing HIP call, "" if none */
int afv_max_keypoints_per_frame(const afv_ctx *ctx); /* sum over levels of max(quota_l + 2, 4 * nIni): safe `cap` */
void *afv_stream(afv_ctx *ctx); /* the context's hipStream_t (for callers that enqueue their own copies) */

/* ---- extraction: host-buffer plugin path (one frame; synchronous) ---- */
int afv_orb_extract(afv_ctx *ctx, const uint8_t *gray, int width, int height, int stride_bytes,
                    afv_keypoint *kps, uint8_t *desc32, int cap, int *n_out);

/* ---- the two halves of the plugin call on their own (FeatureExtractor::detectKeypoints / computeDescriptors, include/FeatureExtractor.h:123-124,
 * src/Feature_orb32.cpp:26-53): a host that calls the virtuals separately - a vocabulary builder that keeps keypoints and recomputes
 * descriptors at them - binds to these.  afv_orb_detect followed by afv_orb_compute on its keypoints gives afv_orb_extract's outputs bit for bit.
 *   afv_orb_detect   detectKeypoints + filterKeypoints: cv::ORB::detect (FAST + NMS, retainBest x 2, Harris response, IC angle, pt scaled to
 *                    level 0; Feature_orb32.cpp:26-40) and DistributeOctTree per level (:63-65, FeatureExtractor.cpp:276-284), merged in
 *                    ascending level order: keypoints with angle and response, no descriptors.
 *   afv_orb_compute  computeDescriptors: cv::ORB::compute (:42-53) for n caller-given keypoints - each described in its own octave at
 *                    cvRound(pt / scale) with its own angle, on the pyramid rebuilt from `gray` (cv::ORB::compute rebuilds levels 0 .. max
 *                    octave of its keypoints).  desc32[n][32] in the order of kps.  Keypoints whose octave is not a level of the context
 *                    or whose centre falls off their level image: AFV_EINVAL, nothing is written. */
int afv_orb_detect(afv_ctx *ctx, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps, int cap, int *n_out);
int afv_orb_compute(afv_ctx *ctx, const uint8_t *gray, int width, int height, int stride_bytes, const afv_keypoint *kps, int n,
                    uint8_t *desc32);

/* ---- extraction: host-buffer batch (vocabulary builder shape, createVocabulary.cpp:161-174).  frames[f] = row-major gray image
 * with stride_bytes between rows; outputs kps[nframes][cap_per_frame], desc32[nframes][cap_per_frame][32], n_out[nframes].
 * Batches of at least two pipeline chunks are software-pipelined (upload of the next chunks / compute / download of the
 * finished chunk overlap).  Page-locked caller buffers (hipHostMalloc, hipHostRegister) are DMA'd in place: the whole rows
 * kps[f][0..cap) / desc32[f][0..cap) are then overwritten and entries beyond n_out[f] are unspecified.  All or nothing: the frames
 * count as page-locked only if the first and last frame of every chunk are (mixing pinned and pageable frames inside a chunk is not
 * supported), the outputs only if both arrays are page-locked end to end.  Pageable buffers are staged through the context's pinned
 * arena — correct, but about 3x slower over PCIe.  On any error no transfer is left in flight. ---- */
int afv_orb_extract_batch(afv_ctx *ctx, const uint8_t *const *frames, int nframes, int width, int height,
                          int stride_bytes, afv_keypoint *kps, uint8_t *desc32, int cap_per_frame, int *n_out);

/* ---- extraction: device-resident batch (benchmark / pipelines).  All pointers are DEVICE pointers;
 * frames are [nframes][height][stride_bytes] with frame_stride_bytes between frames (4-byte aligned base and
 * strides); outputs kps[nframes][cap_per_frame], desc32[nframes][cap_per_frame][32], n_out[nframes].
 * Enqueued on `stream` (hipStream_t); asynchronous — the caller synchronises.  STREAM RULE (all *_device entry points):
 * NULL selects the context's own non-blocking stream (afv_stream), which is NOT ordered with any stream of the caller —
 * not even HIP's null stream: either pass one of your own (non-null) streams, or order afv_stream(ctx) with events
 * (hipEventRecord + hipStreamWaitEvent both ways) or synchronise it.  The Python mirror does the event bridging when torch's
 * current stream is the default stream (_lib.launch_ordered).
 * status_out (device int32, may be NULL) receives 0 or AFV_ECAPACITY. ---- */
int afv_orb_extract_batch_device(afv_ctx *ctx, const uint8_t *d_frames, int nframes, int width, int height,
                                 int stride_bytes, size_t frame_stride_bytes, afv_keypoint *d_kps,
                                 uint8_t *d_desc32, int cap_per_frame, int32_t *d_n_out, int32_t *d_status_out,
                                 void *stream);

/* E12: size_i = normalised powf(scale, octave); sigma2_i = size^2; inf_i = 1/size^2 (host arithmetic) */
int afv_orb_size_sigma(const afv_ctx *ctx, const afv_keypoint *kps, int n, float *size, float *sigma2, float *inf);

/* ---- matching ---- */
enum { AFV_MATCH_KF_KF = 0, AFV_MATCH_KF_FRAME = 1,
       /* OR-ed into afv_match_job.mode (round 6): the descriptors are FLOAT rows - desc1 / desc2 point to n x (desc_bytes / 4) floats,
          desc_bytes = 4 * dim (a multiple of 16, <= 4096) - and the distance is cv::norm(a, b, NORM_L2SQR) narrowed to float, what
          FeatureMatcher::DescriptorDistance returns for SIFT128 / SURF64 / KAZE64 / R2D2 (FeatureMatcher.cc:1508-1531,
          Feature_sift128.cpp:132-134), evaluated like afv_match_l2.  afv_match_bow and afv_match_triangulation (bow.mode) take it;
          th_low / nnratio apply to the float distances as they stand. */
       AFV_MATCH_FLOAT32 = 0x100 };

typedef struct {
    const uint8_t *desc1; int32_t n1;   /* side 1 (KF1 / KF), n1 x desc_bytes, row-major */
    const uint8_t *desc2; int32_t n2;   /* side 2 (KF2 / Frame) */
    int32_t desc_bytes;                 /* 32 for ORB; any multiple of 4 up to 64 (AKAZE61 rows padded to 64 with zeros) */
    /* DBoW2::FeatureVector of each side as CSR over ascending node ids; nnodes == 0 on either side => brute
       force: a single node holding 0..n-1 on both sides */
    const int32_t *node_id1; const int32_t *seg_ptr1; const int32_t *seg_idx1; int32_t nnodes1;
    const int32_t *node_id2; const int32_t *seg_ptr2; const int32_t *seg_idx2; int32_t nnodes2;
    const uint8_t *valid1; const uint8_t *valid2; /* map point exists && !isBad(); NULL => all valid.
                                                     KF_FRAME ignores valid2 (FeatureMatcher.cc:216-232) */
    const float *angle1; const float *angle2;     /* keypoint angles in degrees; required iff check_orientation */
    float th_low;                       /* FeatureMatcher::TH_LOW (matchingTh, 75 for ORB) */
    float nnratio;                      /* mfNNratio */
    int32_t check_orientation;          /* mbCheckOrientation */
    int32_t mode;                       /* AFV_MATCH_KF_KF: out[n1] = idx2|-1, accept best < th_low;
                                           AFV_MATCH_KF_FRAME: out[n2] = idx1|-1, accept best <= th_low */
} afv_match_job;

/* host pointers; jobs are staged, matched on the GPU and copied back.  out = concatenation over jobs of
   int32[n1] (KF_KF) or int32[n2] (KF_FRAME); nmatches[njobs]. */
int afv_match_bow(afv_ctx *ctx, const afv_match_job *jobs, int njobs, int32_t *out, int32_t *nmatches);

/* JOB RECORDS THAT CARRY struct_size (afv_tri_job, afv_proj_job, afv_table_tri_job): set the FIRST field to sizeof(the struct you were
 * compiled against) in every job of the array.  The runtime uses it as the array stride and reads only the fields it covers; fields
 * beyond it are taken as zero (= the monocular call).  A value below the record's first published layout, above 4 x the current one,
 * or different between the jobs of one call is AFV_EINVAL - so a caller that forgot to fill it (0, stack garbage) is rejected instead of
 * having uninitialised tail pointers dereferenced.  Records grow only at the end. */
typedef struct {
    uint32_t struct_size;            /* sizeof(afv_tri_job) */
    afv_match_job bow;               /* valid1/valid2 mean "already HAS a map point" => skipped; nnratio / check_orientation and
                                        mode (but for its AFV_MATCH_FLOAT32 bit) ignored */
    const float *x1, *y1, *x2, *y2; /* mvKeysUn */
    const float *sigma2_2;           /* KeyFrame::GetKeyPt1DSigma2 of side 2 */
    float F12[9];                    /* row-major fundamental matrix */
    float ex, ey;                    /* epipole of camera 1 in image 2 */
    /* stereo keyframes (FeatureMatcher.cc:705-709, :727-731, :741): KeyFrame::mvuRight of either side (>= 0: the keypoint has a
       right-image match); NULL = monocular.  only_stereo = bOnlyStereo: features without a right match are skipped on both sides;
       the epipole-distance test (:741-748) only applies when NEITHER feature is stereo.  A zero-initialised tail = the mono call. */
    const float *u_right1, *u_right2;
    int32_t only_stereo;
} afv_tri_job;
int afv_match_triangulation(afv_ctx *ctx, const afv_tri_job *jobs, int njobs, int32_t *match12, int32_t *nmatches);

/* device-resident brute-force batch: pair p matches descriptor set a[p] (side 1) against set b[p] (side 2) of a
   table d_desc[nsets][cap][32] with per-set counts d_n[nsets] and angles d_kps (afv_keypoint, may be NULL when
   !check_orientation).  d_match[npairs][cap] (idx2 | -1), d_nmatches[npairs].  SearchByBoW(KF,KF) semantics. */
/* (d_nmatches[p] == -0x7fffffff marks a pair whose ordered phase gave up at its pass guard - never observed; the host-result entry
   points turn it into AFV_EHIP.  At most ONE call that uses the column-sliced phase 1 (calls of at most afv_set_small_batch_path's
   max_frames pairs) may be in flight per context: its tickets and slice records are per context, not per stream.) */
int afv_match_bruteforce_pairs_device(afv_ctx *ctx, const uint8_t *d_desc, const afv_keypoint *d_kps,
                                      const int32_t *d_n, int nsets, int cap, const int32_t *d_pair_a,
                                      const int32_t *d_pair_b, int npairs, float th_low, float nnratio,
                                      int check_orientation, int32_t *d_match, int32_t *d_nmatches, void *stream);

/* ---- keyframe descriptor table resident in HBM + multi-GPU replication (BASELINE.json configs[3]) ----
 * The reference matches a query keyframe against every loop / relocalisation candidate, one SearchByBoW call per
 * candidate (LoopClosing::ComputeSim3 src/LoopClosing.cc:255-281 over the candidates of KeyFrameDatabase::
 * DetectLoopCandidates src/KeyFrameDatabase.cc:76-197; Tracking::Relocalization src/Tracking.cc:1162-1182).  Here the
 * keyframes' descriptors (KeyFrame::mDescriptors, const after construction, include/KeyFrame.h:190), keypoint angles
 * (mvKeysUn[i].angle) and optionally their DBoW2::FeatureVector (mFeatVec, KeyFrame.h:194) live in ONE device table
 * of `nsets` slots x `cap` features; batches of (slot a, slot b) pair jobs are matched without any descriptor leaving
 * HBM.  With several GPUs (one process per GPU) the table is built on one rank and replicated with ONE RCCL broadcast
 * per array over xGMI (afv_table_broadcast); every rank then matches its own share of the jobs against its replica.
 * A table belongs to the context it was created on and is destroyed before it. */
typedef struct afv_table afv_table;
int afv_table_create(afv_ctx *ctx, int nsets, int cap /* <= 4096 */, afv_table **out);
void afv_table_destroy(afv_table *t);
/* upload keyframe `set`: n x 32-byte descriptors, angles[n] in degrees (NULL = zeros).  Host pointers; synchronous. */
int afv_table_set(afv_table *t, int set, const uint8_t *desc32, const float *angles, int n);
/* the keyframe's FeatureVector as CSR over ascending node ids (as in afv_match_job); needed by afv_table_match_bow only.
 * The node ids / segment sizes stay on the host too (the merge-join of two FeatureVectors, FeatureMatcher.cc:205-276,
 * is host work: <= a few hundred ints per pair), the feature indices go to the device. */
int afv_table_set_featvec(afv_table *t, int set, const int32_t *node_id, const int32_t *seg_ptr, const int32_t *seg_idx,
                          int nnodes);
/* per-feature validity of keyframe `set` for afv_table_match_bow: valid[i] = map point exists && !isBad() (FeatureMatcher.cc:593-597,
 * :609-613; it changes as the map evolves, so the host re-uploads the n bytes when it does).  NULL = every feature valid (the
 * default).  The brute-force pair entry points take every feature as valid. */
int afv_table_set_valid(afv_table *t, int set, const uint8_t *valid);
/* device views for zero-copy callers: d_desc[nsets][cap][32], d_angle[nsets][cap] (float), d_n[nsets] (int32) */
int afv_table_device_ptrs(afv_table *t, uint8_t **d_desc, float **d_angle, int32_t **d_n);
/* brute-force SearchByBoW(KF,KF) (FeatureMatcher.cc:561-660 with one node holding everything) of npairs (a, b) slot
 * pairs.  pair arrays are HOST int32; outputs are HOST arrays: match12[npairs][cap] (may be NULL: counts only) and
 * nmatches[npairs].  The descriptors never leave the device; only the pair list goes in and the results come out. */
int afv_table_match_pairs(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, int npairs, float th_low, float nnratio,
                          int check_orientation, int32_t *match12, int32_t *nmatches);
/* same, fully device-resident and asynchronous on `stream` (see afv_orb_extract_batch_device for the stream rule):
 * d_pair_a / d_pair_b / d_match12[npairs][cap] / d_nmatches[npairs] are DEVICE pointers */
int afv_table_match_pairs_device(afv_table *t, const int32_t *d_pair_a, const int32_t *d_pair_b, int npairs, float th_low,
                                 float nnratio, int check_orientation, int32_t *d_match12, int32_t *d_nmatches, void *stream);
/* BoW-guided SearchByBoW(KF,KF) (FeatureMatcher.cc:561-660: merge-join of the two FeatureVectors, per shared node the
 * ordered greedy scan) of npairs slot pairs whose FeatureVectors were stored with afv_table_set_featvec.  Host pair
 * arrays in, host results out (match12 may be NULL); descriptors and feature indices stay in HBM. */
int afv_table_match_bow(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, int npairs, float th_low, float nnratio,
                        int check_orientation, int32_t *match12, int32_t *nmatches);
/* Relocalisation batch: SearchByBoW(KF, Frame) (FeatureMatcher.cc:186-283) of ONE frame against `nslots` candidate keyframes of the
 * table - what Tracking::Relocalization does in a loop over the candidates of DetectRelocalizationCandidates (src/Tracking.cc:1162,
 * 1182).  The frame is uploaded once (descriptors, angles, the feature indices of its FeatureVector); every candidate is one job of the
 * same launch pair.  Rules of the reference: validity (afv_table_set_valid) on the keyframe side only (:216-222), a frame feature that
 * already holds a match is skipped (:232), accept best <= th_low (:250), rotation histogram keyed by the frame feature (:259).
 * Host pointers.  match_f[nslots][frame->n] = index of the matched keyframe feature per FRAME feature or -1 (may be NULL: counts
 * only); nmatches[nslots]. */
typedef struct {
    const uint8_t *desc32; int32_t n;   /* the frame's n x 32-byte descriptors (Frame::mDescriptors) */
    const float *angle;                 /* mvKeysUn[i].angle in degrees; required iff check_orientation */
    /* Frame::mFeatVec as CSR over ascending node ids (as in afv_match_job); nnodes == 0: the frame shares no node with anybody */
    const int32_t *node_id; const int32_t *seg_ptr; const int32_t *seg_idx; int32_t nnodes;
} afv_frame_view;
int afv_table_match_bow_frame(afv_table *t, const int32_t *slots, int nslots, const afv_frame_view *frame, float th_low, float nnratio,
                              int check_orientation, int32_t *match_f, int32_t *nmatches);
/* SearchForTriangulation (FeatureMatcher.cc:662-790, stereo branches included: afv_table_set_u_right) of npairs slot pairs
 * over the stored FeatureVectors and the per-keyframe geometry stored with afv_table_set_geometry (mvKeysUn[i].pt and KeyFrame::GetKeyPt1DSigma2(i)): what
 * LocalMapping::CreateNewMapPoints does against <= 20 neighbours (src/LocalMapping.cc:238-297).  Per pair only the
 * fundamental matrix, the epipole and the "already has a map point" masks travel. */
int afv_table_set_geometry(afv_table *t, int set, const float *x, const float *y, const float *sigma2);
/* KeyFrame::mvuRight of a stereo keyframe (>= 0: the keypoint has a right-image match; FeatureMatcher.cc:705, :727), after
 * afv_table_set_geometry of the same slot, which marks every feature monocular (-1). */
int afv_table_set_u_right(afv_table *t, int set, const float *u_right);
/* afv_table_set (a recycled slot) forgets the slot's FeatureVector, geometry and validity mask: afv_table_match_bow /
 * afv_table_match_triangulation return AFV_EINVAL for a slot that holds features but lacks what the call needs, instead of
 * answering "no matches". */
typedef struct {
    uint32_t struct_size;    /* sizeof(afv_table_tri_job) (see afv_tri_job) */
    float F12[9];            /* row-major fundamental matrix (as afv_tri_job) */
    float ex, ey;            /* epipole of camera a in image b */
    const uint8_t *has_mp1;  /* [n_a] feature already has a map point => skipped (NULL = none has) */
    const uint8_t *has_mp2;  /* [n_b] */
    float th_low;            /* FeatureMatcher::TH_LOW */
    int32_t only_stereo;     /* bOnlyStereo (FeatureMatcher.cc:707-709, :729-731); 0 for monocular maps */
} afv_table_tri_job;
int afv_table_match_triangulation(afv_table *t, const int32_t *pair_a, const int32_t *pair_b, const afv_table_tri_job *geo, int npairs,
                                  int32_t *match12 /*[npairs][cap]: idx_b | -1*/, int32_t *nmatches);
/* refresh the host copy of the per-slot counts after the caller wrote d_n on the device itself */
int afv_table_sync_counts(afv_table *t);

/* RCCL communicator (one process per GPU).  No RCCL symbol is linked: the library is looked up at run time
 * (an RCCL already loaded into the process, e.g. by PyTorch, is reused; otherwise librccl.so.1 is opened). */
typedef struct afv_comm afv_comm;
#define AFV_COMM_ID_BYTES 128
/* rank 0 creates the id and hands it to the other ranks through any host channel (file, socket, MPI, torch store) */
int afv_comm_unique_id(uint8_t id[AFV_COMM_ID_BYTES]);
int afv_comm_create(afv_ctx *ctx, const uint8_t id[AFV_COMM_ID_BYTES], int nranks, int rank, afv_comm **out);
void afv_comm_destroy(afv_comm *comm);
int afv_comm_rank(const afv_comm *comm);
int afv_comm_size(const afv_comm *comm);
/* in-place broadcast of `bytes` at device pointer `d_buf` from rank `root` (ncclBroadcast, uint8), asynchronous on
 * `stream` (NULL = the context's stream) */
int afv_comm_broadcast(afv_comm *comm, void *d_buf, size_t bytes, int root, void *stream);
/* every rank contributes bytes_per_rank at d_send; d_recv receives nranks * bytes_per_rank in rank order */
int afv_comm_allgather(afv_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
/* replicate the WHOLE table from `root`: descriptors, angles, counts, and — when the root holds them — FeatureVectors (the
 * feature-index plane plus the host-side node structure), geometry and validity planes: one broadcast per array and one for the
 * "replica image" (node ids / CSR pointers / per-set flags), then a stream synchronisation.  After the call every rank answers
 * afv_table_match_pairs, afv_table_match_bow and afv_table_match_triangulation exactly as the root does; whatever a receiver held
 * before is replaced.  elapsed_ms (may be NULL) = device time of the broadcasts (hipEvents on the context's stream). */
int afv_table_broadcast(afv_comm *comm, afv_table *t, int root, float *elapsed_ms);
/* copy everything `src` holds into `dst` (same nsets / cap; may live on another context or device) through the same replica image
 * and rebuild step the receivers of afv_table_broadcast run — a replica without a communicator (e.g. one table per calling thread) */
int afv_table_clone(const afv_table *src, afv_table *dst);
/* block partition of n_units over the ranks: unit u belongs to the rank whose [lo, hi) holds it */
void afv_shard_range(long n_units, int rank, int nranks, long *lo, long *hi);

/* float descriptors, L2^2 distance (config #3): brute force with SearchByBoW(KF,KF) control flow */
int afv_match_l2(afv_ctx *ctx, const float *desc1, int n1, const float *desc2, int n2, int dim,
                 const uint8_t *valid1, const uint8_t *valid2, float th_low, float nnratio, int32_t *match12,
                 int32_t *nmatches);
/* the same matcher over a device-resident table of float descriptors, many (set a, set b) pair jobs per call (the float counterpart of
 * afv_match_bruteforce_pairs_device): d_desc[(set * cap + row) * dim], d_n[set] rows per set, dim 64 or 128; d_match[pair * cap + row] =
 * index into set b or -1 (rows past the set's count: -1), d_nmatches[pair].  Asynchronous on `stream` (NULL: the context's stream). */
int afv_match_l2_pairs_device(afv_ctx *ctx, const float *d_desc, const int32_t *d_n, int cap, int dim, const int32_t *d_pair_a,
                              const int32_t *d_pair_b, int npairs, float th_low, float nnratio, int32_t *d_match, int32_t *d_nmatches,
                              void *stream);

/* ---- projection-guided matching core (grid window + Hamming) ----
 * Matching loops of FeatureMatcher::SearchByProjection(F, localMapPoints) (src/FeatureMatcher.cc:73-154, mode
 * AFV_PROJ_LOCALMAP) and SearchByProjection(CurrentFrame, LastFrame) (:1291-1402, AFV_PROJ_LASTFRAME, mono) incl.
 * Frame::GetFeaturesInArea (src/Frame.cc:333-382) over the grid of AssignFeaturesToGrid (Frame.cc:225-240).  The caller
 * evaluates the projection as the reference does and passes, per query in the reference's iteration order:
 * (u, v) = projected position, r = window radius (:91 / :1343), [min_size, max_size] = admissible keyPtsSize band. */
enum { AFV_PROJ_LOCALMAP = 0, AFV_PROJ_LASTFRAME = 1 };
typedef struct {
    uint32_t struct_size;                               /* sizeof(afv_proj_job) (see afv_tri_job) */
    const uint8_t *desc; int32_t n; int32_t desc_bytes; /* frame features F.mDescriptors (n <= 8192) */
    const float *x; const float *y;                     /* F.mvKeysUn[i].pt */
    const float *size;                                  /* F.keyPtsSize[i] */
    const float *angle;                                 /* F.mvKeysUn[i].angle (LASTFRAME + check_orientation) */
    const uint8_t *occupied;                            /* F.pts[i] && F.pts[i]->NumberOfObservations() > 0; NULL = none */
    const float *inf;                                   /* KeyFrame::GetKeyPt1DInf(i): afv_match_fuse only */
    float min_x, min_y, grid_inv_w, grid_inv_h;         /* mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv */
    int32_t grid_cols, grid_rows;                       /* FRAME_GRID_COLS 64, FRAME_GRID_ROWS 48 (Frame.h:40-41); cols * rows <= 8192 */
    int32_t nq;                                         /* queries: map points / last-frame keypoints */
    const uint8_t *qdesc; const uint8_t *qvalid;        /* pMP->GetDescriptor(); pMP && in view && !isBad (NULL = all) */
    const float *qu; const float *qv; const float *qr; const float *qmin_size; const float *qmax_size;
    const float *qangle;                                /* LastFrame.mvKeysUn[i].angle (LASTFRAME + check_orientation) */
    const uint8_t *qoccupies;                           /* pMP->NumberOfObservations() > 0 (NULL = yes) */
    float th_high, nnratio, size_tol, inv_size_tol;     /* TH_HIGH, mfNNratio, F.sizeTolerance, F.invSizeTolerance */
    int32_t check_orientation, mode;
    /* stereo frames / keyframes (NULL u_right = monocular; a zero-initialised tail = the mono call):
       u_right[i] = mvuRight of feature i; q_ur[q] = the query's projected right-image coordinate (pMP->mTrackProjXR :116,
       u - mbf * invzc :1369, ur of Fuse :885); q_er_max[q] = the gate of afv_match_projection (r * pMP->trackSigma :117 in
       AFV_PROJ_LOCALMAP, the window radius :1371 in AFV_PROJ_LASTFRAME).
       afv_match_projection skips a feature with u_right > 0 whose |q_ur - u_right| exceeds q_er_max (:114-119, :1367-1372);
       afv_match_fuse replaces the 2-dof gate e2 * inf <= 5.99 by the 3-dof one (ex^2 + ey^2 + er^2) * inf <= 7.8 for features with
       u_right >= 0 (:880-894).  afv_match_sim3 / afv_match_initialization have no stereo branch and ignore the three fields. */
    const float *u_right; const float *q_ur; const float *q_er_max;
    /* (ABI 6, appended: struct_size tells) float descriptors.  FeatureMatcher::DescriptorDistance dispatches on DescriptorType (FeatureMatcher.cc:1508-1531) and returns
       Descriptor_Distance_Type = float (Types.h:127): for SIFT128 / SURF64 / KAZE64 / R2D2 / any non-binary feature it is
       cv::norm(a, b, NORM_L2SQR) (Feature_sift128.cpp:132-134).  float_dim > 0: desc and qdesc point to rows of float_dim floats
       (a multiple of 4, <= 1024; desc_bytes is ignored), distances are evaluated like afv_match_l2 (float differences, squares and 4-way
       partial sums in double, narrowed once), th_high / nnratio apply to them as they stand; 0 (what a caller compiled against an older
       header gets) = binary rows of desc_bytes.  All four searches below accept it; afv_match_projection runs float jobs on either engine of
       its ordered phase (afv_set_projection_resolve), afv_match_initialization always on the ordered walk (its fixed point packs
       distances in 16 bits). */
    int32_t float_dim;
} afv_proj_job;
/* assign = concatenation over jobs of int32[n]: index of the query assigned to feature i (F.pts[i] = pMP) or -1 */
int afv_match_projection(afv_ctx *ctx, const afv_proj_job *jobs, int njobs, int32_t *assign, int32_t *nmatches);
/* matching core of FeatureMatcher::Fuse(pKF, vpMapPoints, th) (src/FeatureMatcher.cc:794-940) over
 * KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:613-652): per map point the most similar keypoint of the window that lies in
 * the size band [qmin_size, qmax_size] (= predictedSize / sizeTolerance .. * sizeTolerance, :871-873) and passes the
 * reprojection gate e2 * inf <= 5.99 (:897); th_high carries TH_LOW (:915).  Map points are independent here; the map surgery
 * (:918-936) stays with the caller.  best = concatenation over jobs of int32[nq] (feature index or -1); nfound[njobs]. */
int afv_match_fuse(afv_ctx *ctx, const afv_proj_job *jobs, int njobs, int32_t *best, int32_t *nfound);
/* inf == NULL in a job drops the reprojection gate: that is the matching core of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)
 * (src/FeatureMatcher.cc:942-1064).  The remaining SearchByProjection flavours need no extra entry point:
 *   - SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (:287-397): afv_match_projection, mode AFV_PROJ_LASTFRAME,
 *     check_orientation 0, occupied = vpMatched[i] != NULL, qmin/qmax = predictedSize / , * sizeTolerance, th_high = TH_LOW;
 *   - SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, useHigh) (:1404-1506, relocalisation): the same mode with
 *     check_orientation = mbCheckOrientation, qangle = pKF->mvKeysUn[i].angle, occupied = CurrentFrame.pts[i] != NULL,
 *     th_high = descDistTh_{low,high}_reloc. */

/* SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) (src/FeatureMatcher.cc:1066-1287): j12 = map points of KF1
 * (query q = KF1 feature index; q_valid = has a good, not already matched map point that projects into KF2) searched among
 * KF2's features, j21 the reverse; both are gate-less best-only searches with th_high = TH_HIGH (:1171, :1254); the
 * agreement check (:1268-1284) runs here too.  Requires j12->nq == j21->n and j21->nq == j12->n.
 * match12[j12->nq] = KF2 feature index or -1. */
int afv_match_sim3(afv_ctx *ctx, const afv_proj_job *j12, const afv_proj_job *j21, int32_t *match12, int32_t *nfound);

/* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (src/FeatureMatcher.cc:399-557; active code
 * :480-556): queries = F1's features (q_valid = octave 0, q_u/q_v = vbPrevMatched, q_radius = windowSize,
 * q_size_min = 0, q_size_max = F1.maxKeyPtSize, q_angle = F1 keypoint angle), features = F2 with its grid, th_high = TH_LOW,
 * nnratio, check_orientation.  Ordered: a feature already matched at a distance <= the current one is skipped (:513) and a
 * later query steals it otherwise (:531-535).  match12 = concatenation over jobs of int32[nq]; the caller refreshes
 * vbPrevMatched (:551-553). */
int afv_match_initialization(afv_ctx *ctx, const afv_proj_job *jobs, int njobs, int32_t *match12, int32_t *nmatches);

/* ---- BoW quantisation ----
 * DBoW2 TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) as called by Vocabulary::transform
 * (src/Vocabulary.cpp:156-206, levelsup = 4; Frame::ComputeBoW src/Frame.cc:397-401, KeyFrame::ComputeBoW
 * src/KeyFrame.cc:65-73).  The tree is uploaded once; per descriptor the kernel descends it (k-way Hamming argmin per level,
 * first minimum wins) and returns the leaf node and the node met at depth L - levelsup, from which the host builds the
 * FeatureVector (node -> ascending feature indices) that afv_match_bow consumes and the BowVector (word weights). */
typedef struct afv_vocab afv_vocab;
/* children of node i = child_idx[child_ptr[i] .. child_ptr[i+1]) in DBoW2 order; node 0 = root; desc = nnodes x desc_bytes */
int afv_vocab_create(afv_ctx *ctx, int k, int L, int nnodes, const int32_t *child_ptr, const int32_t *child_idx,
                     const uint8_t *desc, int desc_bytes, afv_vocab **out);
void afv_vocab_destroy(afv_ctx *ctx, afv_vocab *v);
int afv_bow_transform(afv_ctx *ctx, const afv_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *leaf_node,
                      int32_t *node_at_level);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:279-349) for a batch of map points - the one M1 (DescriptorDistance) call site
 * of the reference that had no device form (LocalMapping calls it per created / fused point).  Map point s observes the
 * descriptors desc[set_ptr[s] .. set_ptr[s + 1]) (rows of desc_bytes, host memory, <= 65535 per point); best_idx[s] = the row (index
 * INSIDE the set) with the least median distance to the others - the median as the reference takes it: the floor(0.5 (N - 1))-th entry
 * of the sorted row of the N x N matrix, the zero self-distance included; the first row wins ties - or -1 for an empty set;
 * best_median[s] (may be NULL) = that median.  Binary descriptors. */
int afv_distinctive_descriptors(afv_ctx *ctx, const uint8_t *desc, int desc_bytes, const int32_t *set_ptr, int nsets, int32_t *best_idx,
                                int32_t *best_median);

/* ... and for float descriptors (DescriptorDistance = cv::norm(a, b, NORM_L2SQR) as a float, like afv_match_l2): desc = rows of `dim` floats,
 * best_median (may be NULL) = the winning row's median distance. */
int afv_distinctive_descriptors_f32(afv_ctx *ctx, const float *desc, int dim, const int32_t *set_ptr, int nsets, int32_t *best_idx,
                                    float *best_median);

/* The float cases of Vocabulary::transform (src/Vocabulary.cpp:158-187: SIFT128, SURF64, KAZE64, R2D2, any non-binary feature): node
 * descriptors and features are `dim` floats (64, 128 or 256), the distance at a node is DBoW2's float-descriptor distance - squared
 * differences evaluated in float, accumulated in double in index order (upstream FSurf64::distance; the reference's fork with the classes
 * it instantiates is an empty submodule: parity unpinned) - first minimum wins.  Same tree arguments as afv_vocab_create; destroyed with
 * afv_vocab_destroy; afv_bow_transform refuses a float vocabulary and afv_bow_transform_f32 a binary one; afv_frame_bow_transform wants the
 * vocabulary of the frame's own descriptor kind and size (a float frame - afv_frame_params.float_dim - a float vocabulary of that dimension). */
int afv_vocab_create_f32(afv_ctx *ctx, int k, int L, int nnodes, const int32_t *child_ptr, const int32_t *child_idx, const float *desc, int dim,
                         afv_vocab **out);
int afv_bow_transform_f32(afv_ctx *ctx, const afv_vocab *v, const float *desc, int n, int levelsup, int32_t *leaf_node, int32_t *node_at_level);

/* words the vocabulary stops (DBoW2 transform: a word whose weight is not > 0 enters neither the BowVector nor the FeatureVector):
 * stopped[nnodes] != 0 marks them; NULL = none (the default).  Only the device-built FeatureVector of afv_frame_bow_transform reads it. */
int afv_vocab_set_stopped(afv_ctx *ctx, afv_vocab *v, const uint8_t *stopped);

/* ---- the device-resident Frame (round 5) ----
 * In the reference a Frame is built once (src/Frame.cc:171-223: ExtractFeatures :242-259 -> UndistortKeyPoints :403-433 ->
 * AssignFeaturesToGrid :225-240) and then consumed by every matcher of that tracking step: SearchByProjection(cur, last)
 * (src/Tracking.cc:747,753), SearchByProjection(F, local map points) (:1026), Frame::ComputeBoW (Frame.cc:397-401) -> SearchByBoW(KF, F)
 * (Tracking.cc:626-629), SearchForInitialization, and - if promoted - copied into a KeyFrame (src/KeyFrame.cc:36).  An afv_frame is that
 * object on the device: afv_frame_extract runs the extraction of afv_orb_extract and KEEPS keypoints and descriptors in HBM, derives
 * mvKeysUn / keyPtsSize / keyPtsSigma2 / keyPtsInf per feature and builds the 64 x 48 grid there (PosInGrid's expression, Frame.cc:384-394:
 * cell = round((x - mnMinX) * mfGridElementWidthInv), ascending feature index inside a cell); the consumers below read it in place.
 * After the image nothing of the frame is uploaded again: a projection search sends its queries, a BoW search nothing at all.
 * A frame belongs to the context it was created on (one calling thread at a time, like the context) and dies with it. */
typedef struct afv_frame afv_frame;
typedef struct {
    uint32_t struct_size;             /* sizeof(afv_frame_params) */
    float min_x, min_y, max_x, max_y; /* mnMinX, mnMinY, mnMaxX, mnMaxY (Frame::ComputeImageBounds, Frame.cc:435-466) */
    int32_t grid_cols, grid_rows;     /* FRAME_GRID_COLS 64, FRAME_GRID_ROWS 48 (Frame.h:40-41); cols * rows <= 8192 */
    int32_t distorted;                /* 0: mDistCoef[0] == 0, mvKeysUn = mvKeys (Frame.cc:405-409) and the grid is built inside
                                         afv_frame_extract; 1: the caller undistorts (cv::undistortPoints of N points is host work) and
                                         hands mvKeysUn over with afv_frame_set_undistorted, which builds the grid */
    int32_t cap;                      /* most features the frame can hold; 0 = afv_max_keypoints_per_frame(ctx); <= 8192 */
    int32_t desc_bytes;               /* (ABI 6) bytes of one binary descriptor, 1 .. 64: 32 ORB (0 = 32), 61 AKAZE, 48 BRISK ... - the reference's
                                         matchers dispatch on DescriptorType (FeatureMatcher.cc:1508-1531).  Rows live zero-padded to 8 or 16
                                         dwords in HBM.  A frame that is not 32-byte is filled with afv_frame_set_features (afv_frame_extract
                                         is the ORB32 extractor; the keyframe TABLE holds 32-byte rows only: afv_table_set_from_frame /
                                         afv_table_match_bow_frame_h answer AFV_EUNSUPPORTED for it) and serves afv_frame_bow_transform with a
                                         vocabulary of the same descriptor size, afv_frame_match_projection / _fuse / _initialization */
    int32_t float_dim;                /* (ABI 6, appended) > 0: the frame holds FLOAT descriptors of float_dim floats (a multiple of 4, <= 1024:
                                         SIFT128, SURF64, KAZE64, R2D2 ...; desc_bytes is ignored): afv_frame_set_features takes the rows as
                                         n x float_dim floats behind its uint8_t pointer, afv_frame_bow_transform wants a float vocabulary of
                                         that dimension (afv_vocab_create_f32), the projection searches and SearchForInitialization use
                                         L2^2 distances (afv_proj_job.float_dim; afv_proj_queries.qdesc = nq x float_dim floats,
                                         desc_bytes = 4 * float_dim).  Like every frame that is not 32-byte it stays out of the keyframe
                                         table: SearchByBoW(KF, F) on float rows is afv_match_bow with AFV_MATCH_FLOAT32 and the frame's
                                         FeatureVector (afv_frame_get_featvec) */
} afv_frame_params;
int afv_frame_create(afv_ctx *ctx, const afv_frame_params *params, afv_frame **out);
void afv_frame_destroy(afv_frame *f);
/* FeatureExtractor::operator() into the frame: host outputs exactly as afv_orb_extract (kps / desc32 / n_out; the three may all be NULL
 * when the caller wants the device copy only) and the device-resident frame described above.  Synchronous. */
int afv_frame_extract(afv_frame *f, const uint8_t *gray, int width, int height, int stride_bytes, afv_keypoint *kps, uint8_t *desc32,
                      int cap, int *n_out);
/* a frame whose features were produced elsewhere (a stereo rig, another extractor, a test): n keypoints (cv::KeyPoint layout; pt =
 * mvKeysUn unless `distorted`), n descriptor rows of the frame's kind (desc_bytes bytes each; float_dim floats each for a float frame), per-feature keyPtsSize (NULL = E12 from the octave as afv_orb_size_sigma) and
 * mvuRight (NULL = monocular: -1).  Host pointers; asynchronous on the context's stream (the arrays are copied before the call returns). */
int afv_frame_set_features(afv_frame *f, const afv_keypoint *kps, const uint8_t *desc32, int n, const float *size, const float *u_right);
/* mvKeysUn of a `distorted` frame: x[n], y[n] (host); builds the grid.  Asynchronous on the context's stream. */
int afv_frame_set_undistorted(afv_frame *f, const float *x, const float *y);
int afv_frame_count(const afv_frame *f); /* N of the last afv_frame_extract / afv_frame_set_features */
/* device views (valid until the frame is destroyed; contents change with the next extract): any may be NULL.
 * d_kps[cap] afv_keypoint (mvKeys), d_desc[cap][row] (row = 32 bytes; 64 for binary rows above 32 bytes, zero-padded; 4 * float_dim for a float frame), d_xy = {x[cap], y[cap]} of mvKeysUn, d_size[cap] keyPtsSize, d_angle[cap],
 * d_n = the count */
int afv_frame_device_ptrs(afv_frame *f, afv_keypoint **d_kps, uint8_t **d_desc, float **d_x, float **d_y, float **d_size, float **d_angle,
                          int32_t **d_n);
/* parity / debugging: the grid as CSR over cells (cell = ix * grid_rows + iy): cell_ptr[cols * rows + 1], cell_idx[n] */
int afv_frame_get_grid(afv_frame *f, int32_t *cell_ptr, int32_t *cell_idx);
/* Frame::ComputeBoW (Frame.cc:397-401): the tree descent of afv_bow_transform over the frame's own descriptors.  leaf_node[n] /
 * node_at_level[n] (host, may be NULL) feed the caller's BowVector; the FeatureVector (node -> ascending feature indices, stopped words
 * left out) is built ON THE DEVICE and stays with the frame for afv_table_match_bow_frame_h / afv_table_set_from_frame; its node
 * structure (what the merge-join of two FeatureVectors walks, FeatureMatcher.cc:205-276) is kept on the host side of the handle.
 * nnodes_out (may be NULL) = number of distinct nodes.  Synchronous. */
int afv_frame_bow_transform(afv_frame *f, const afv_vocab *v, int levelsup, int32_t *leaf_node, int32_t *node_at_level, int32_t *nnodes_out);
/* the FeatureVector of the last afv_frame_bow_transform as CSR (host copies; any may be NULL): node_id[nnodes], seg_ptr[nnodes + 1],
 * seg_idx[seg_ptr[nnodes]] */
int afv_frame_get_featvec(afv_frame *f, int32_t *node_id, int32_t *seg_ptr, int32_t *seg_idx);
/* the queries of a projection search (everything afv_proj_job carries on the query side), host pointers */
typedef struct {
    uint32_t struct_size;                               /* sizeof(afv_proj_queries) */
    int32_t nq;
    const uint8_t *qdesc; int32_t desc_bytes;           /* pMP->GetDescriptor(), nq x desc_bytes (32) */
    const uint8_t *qvalid;                              /* NULL = all */
    const float *qu; const float *qv; const float *qr; const float *qmin_size; const float *qmax_size;
    const float *qangle;                                /* AFV_PROJ_LASTFRAME / initialization with check_orientation */
    const uint8_t *qoccupies;                           /* NULL = yes */
    const float *q_ur; const float *q_er_max;           /* stereo frames only (see afv_proj_job) */
    const uint8_t *occupied;                            /* F.pts[i] && F.pts[i]->NumberOfObservations() > 0 per FRAME feature; NULL = none:
                                                           the one piece of map state the searches read from the frame side */
    float th_high, nnratio;                             /* TH_HIGH (or the threshold of the flavour), mfNNratio */
    int32_t check_orientation, mode;                    /* mode: AFV_PROJ_LOCALMAP / AFV_PROJ_LASTFRAME */
    /* the queries' descriptors by reference instead of by value: a map point's descriptor is a row of the keyframe that observed it
     * (MapPoint::ComputeDistinctDescriptors copies pKF->mDescriptors.row(idx)), so when that keyframe sits in an afv_table the row is
     * gathered on the device: qref_table + qref_slot[nq] + qref_idx[nq] (qdesc NULL).  8 bytes per query instead of 32. */
    afv_table *qref_table; const int32_t *qref_slot; const int32_t *qref_idx;
} afv_proj_queries;
/* SearchByProjection(F, vpMapPoints, th) (FeatureMatcher.cc:73-154) / SearchByProjection(CurrentFrame, LastFrame, th, mono) (:1291-1402)
 * and the flavours listed under afv_match_projection, against the frame's resident features and grid; size_tol = the frame's
 * sizeTolerance (the context's scale_factor, Frame.cc:73-74).  assign[n] (query index now in F.pts[i] | -1), *nmatches.  Synchronous. */
int afv_frame_match_projection(afv_frame *f, const afv_proj_queries *q, int32_t *assign, int32_t *nmatches);
/* matching core of Fuse(pKF, vpMapPoints, th) with the frame in the role of the keyframe (KeyFrame::GetFeaturesInArea reads the grid the
 * KeyFrame copied from its Frame, KeyFrame.cc:36,613-652): best[nq] (feature index | -1), *nfound.  use_inf_gate 0 = the Sim3 flavour. */
int afv_frame_match_fuse(afv_frame *f, const afv_proj_queries *q, int use_inf_gate, int32_t *best, int32_t *nfound);
/* SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (FeatureMatcher.cc:399-557) between two resident frames: the
 * queries are F1's own features (descriptor, angle, octave-0 filter: all on the device), prev_x / prev_y = vbPrevMatched (host, n1
 * floats each).  match12[n1], *nmatches.  Nothing but the 8 n1 bytes of vbPrevMatched is uploaded. */
int afv_frame_match_initialization(afv_frame *f1, afv_frame *f2, const float *prev_x, const float *prev_y, float window_size, float th_low,
                                   float nnratio, int check_orientation, int32_t *match12, int32_t *nmatches);
/* KeyFrame::KeyFrame(Frame &F, ...) (src/KeyFrame.cc:36-60): the frame's descriptors, angles, mvKeysUn, sigma2, mvuRight and - when
 * afv_frame_bow_transform ran - its FeatureVector become keyframe `slot` of the table, device to device (the node structure host to
 * host).  The slot's validity mask is reset to "all valid" like afv_table_set does.  Asynchronous on the context's stream. */
int afv_table_set_from_frame(afv_table *t, int slot, afv_frame *f);
/* afv_table_match_bow_frame with the frame side taken from a resident frame (afv_frame_bow_transform must have run): SearchByBoW(KF, F)
 * of TrackReferenceKeyFrame (Tracking.cc:626-629, one slot) and of Relocalization (:1162-1182, the candidates). */
int afv_table_match_bow_frame_h(afv_table *t, const int32_t *slots, int nslots, afv_frame *f, float th_low, float nnratio,
                                int check_orientation, int32_t *match_f, int32_t *nmatches);
/* engine of the ordered phase of the projection searches / SearchForInitialization (identical results):
 *   1: one fixed point over all live queries of a job on a 1024-thread workgroup (round 5)   0: the ordered walk of rounds 1-4 on one
 *   wavefront   2 (default): 1 whenever the job's tables fit the workgroup's LDS, else 0   3: as 1, but ranking and ordered phase as two
 *   launches even for a single job on a resident frame (by default those run in ONE launch: the last ranking workgroup goes on as the
 *   fixed-point workgroup) */
int afv_set_projection_resolve(afv_ctx *ctx, int engine);

/* DescriptorDistance_orb32 on the host (utility for adapters / tests) */
int afv_hamming256(const uint8_t *a, const uint8_t *b);

/* ---- live stage timing: when enabled, every afv_orb_extract_batch_device / afv_match_bruteforce_pairs_device / afv_table_match_pairs* call
 * brackets each kernel stage with hipEvents recorded on the stream the kernels are launched on.
 * afv_profile_read waits for the recorded events and returns, per stage, the number of launches and the summed
 * duration in milliseconds since the last afv_profile_enable(ctx, 1).  Stages: ---- */
enum { AFV_STAGE_PYRAMID = 0, AFV_STAGE_FAST_NMS = 1 /* k_fast_nms */, AFV_STAGE_SELECT = 2, AFV_STAGE_DESCRIBE = 3,
       AFV_STAGE_MATCH = 4 /* k_match_topk: the xor + popcount phase */, AFV_STAGE_MATCH_RESOLVE = 5 /* ordered greedy resolve */,
       AFV_STAGE_HARRIS = 6 /* k_retain_score + k_harris: retainBest on the score, Harris response of the survivors */,
       AFV_NUM_STAGES = 7 };
/* The stage list grows between releases: size the arrays passed to afv_profile_read with afv_num_stages() (or with the AFV_NUM_STAGES
 * of the header the caller was compiled against, after checking that afv_num_stages() is not larger). */
int afv_num_stages(void);
int afv_profile_enable(afv_ctx *ctx, int enable); /* 0 = off; n >= 1 = time the stages of every n-th extraction / pair-match call
                                                     * (1 = every call; the event pairs cost ~3 % of a batch step, sampling keeps that
                                                     * out of a timed run); resets the accumulated figures */
int afv_profile_read(afv_ctx *ctx, int32_t *launches /*[afv_num_stages()]*/, float *total_ms /*[afv_num_stages()]*/,
                     int64_t *units /*[afv_num_stages()], frames (pairs for MATCH) covered by those launches; may be NULL*/);
/* batches of at least `min_frames` frames (pairs) are split over the context's two streams so that latency-bound kernels of
 * one half overlap the VALU-bound ones of the other (default 64; 0x7fffffff disables the split) */
int afv_set_split_threshold(afv_ctx *ctx, int min_frames);
int afv_set_split_chunks(afv_ctx *ctx, int chunks); /* ... into this many chunks alternating over the two streams (2..64; 0 = automatic,
                                                      * chunks of about 85 frames: the default) */
/* afv_match_l2_pairs_device processes its jobs in launches of at most `pairs` jobs (default 2048; the key scratch of a launch is
 * pairs x cap x 32 bytes and is reused by the next launch on the same stream) */
int afv_set_l2_chunk_pairs(afv_ctx *ctx, int pairs);
/* Phase 2 of the brute-force pair matchers (the ordered greedy assignment of SearchByBoW, FeatureMatcher.cc:587-641); identical results:
 *   1: one fixed point over all live rows of a pair, a thread per row, claims in rotating LDS arrays (one barrier per pass): half the
 *      latency of a pair, more total work
 *   0: the ordered walk of rounds 2-3: one wavefront, 64 rows per round: what a batch that fills the chip runs fastest with
 *   2 (default): 1 for calls of at most 256 pairs (every pair then has a CU to itself), else 0 */
int afv_set_match_resolve(afv_ctx *ctx, int engine);
/* The small-batch ("latency") path.  Tracking extracts ONE frame per call (src/Frame.cc:186 -> src/FeatureExtractor.cpp:111-121) and
 * matches it against ONE other frame: at that size the batch kernels are a chain of dependent launches, each a few microseconds of
 * ramp and cache boundary around a fraction of a microsecond of work.  Calls of at most `max_frames` frames / pairs (default 4) run
 * kernels shaped for that case instead - the whole pyramid in one launch, retainBest + Harris in one launch, a quadtree workgroup of
 * 1024 threads, the brute-force distance phase split over column slices - with bit-identical results.
 *   mode 0 = never, 1 = calls of at most max_frames frames (default), 2 = always (parity tests); max_frames 0 keeps the current value */
int afv_set_small_batch_path(afv_ctx *ctx, int mode, int max_frames);
/* Phase 1 of the brute-force pair matchers (afv_match_bruteforce_pairs_device, afv_table_match_pairs*, plain SearchByBoW jobs without
 * a FeatureVector): every row's four nearest columns.  Both engines produce identical keys, hence identical match vectors.
 *   AFV_MATCH_ENGINE_MFMA (default): Hamming distance as an exact i8 x i8 -> i32 contraction on the matrix cores
 *   AFV_MATCH_ENGINE_POPCOUNT:       8 xor + 8 v_bcnt per descriptor pair on the vector ALU */
enum { AFV_MATCH_ENGINE_POPCOUNT = 0, AFV_MATCH_ENGINE_MFMA = 1 };
int afv_set_match_engine(afv_ctx *ctx, int engine);
/* afv_orb_extract_batch pipelines H2D / compute / D2H over chunks of `frames` frames with the uploads `chunks_ahead` chunks ahead of
 * the compute (defaults 64 and 8; batches below two chunks run as one) */
int afv_set_pipeline_chunk(afv_ctx *ctx, int frames, int chunks_ahead);

/* ---- stage-level introspection of the LAST afv_orb_extract* call (parity tests / profiling) ---- */
typedef struct {
    int32_t nlevels, width, height;
    int32_t lw[AFV_MAX_LEVELS], lh[AFV_MAX_LEVELS];
    float lscale[AFV_MAX_LEVELS];
    int32_t quota[AFV_MAX_LEVELS];    /* mnFeaturesPerLevel */
    int32_t cv_quota[AFV_MAX_LEVELS]; /* cv::ORB nfeaturesPerLevel for nfeatures*10 */
    int32_t cand_cap[AFV_MAX_LEVELS];
} afv_geometry;
int afv_get_geometry(const afv_ctx *ctx, afv_geometry *g);
/* copy pyramid level `level` of frame `frame` to host (tightly packed lw x lh) */
int afv_debug_get_level(afv_ctx *ctx, int frame, int level, uint8_t *out);
/* FAST+NMS candidates of (frame, level), unordered: packed[i] = x | y<<12 | score<<24, response[i] = Harris */
int afv_debug_get_candidates(afv_ctx *ctx, int frame, int level, uint32_t *packed, float *response, int cap, int *n_out);
/* keypoints chosen by the quadtree for (frame, level) in list order: x,y level coordinates */
int afv_debug_get_selected(afv_ctx *ctx, int frame, int level, int32_t *x, int32_t *y, float *response, int cap, int *n_out);
/* standalone 7x7 Gaussian blur of a level (E9) — same arithmetic as the fused describe kernel */
int afv_debug_blur_level(afv_ctx *ctx, int frame, int level, uint8_t *out);
/* host-only (no device, no context): the work plan of the one-launch pyramid for a geometry - per level and per tile index of the top
 * level the (need.lo, need.hi, own.lo, own.hi) ranges in x (first [nlevels][ntx] quadruples) and y ([nlevels][nty]), the resize
 * coefficient tables (pairs (source offset, weight of the right tap) per level: x then y, levels 1..) and, in info[4 + 6 * 8]:
 * nlevels, ntx, nty, LDS bytes, then per level w, h, LDS pitch, log2 dword slots per row, x-table offset, y-table offset (in pairs).
 * tests/test_host_logic.py replays the kernel's data flow on the CPU from exactly these numbers. */
int afv_debug_pyramid_plan(const afv_orb_params *params, int width, int height, int tile_w, int tile_h, int16_t *regions,
                           int regions_cap, int32_t *info, int16_t *tables, int tables_cap);

#ifdef __cplusplus
}
#endif
#endif
