// afv_adapter.hpp — header-only C++17 host adapter over the C-ABI (include/afv_hip.h).
//
// It mirrors the reference's plugin classes so that a host which HAS OpenCV can drop the GPU path in by compiling
// this header with -DAFV_WITH_OPENCV (then afv::KeyPoint = cv::KeyPoint, afv::Mat8 = cv::Mat) — see INTEGRATION.md for
// the subclass that plugs into Tracking.cc:1523-1552.  Without OpenCV (this repository's build container) the same
// code works on two layout-compatible PODs, which is what adapter_selftest.cpp compiles and runs.
//
// Reference interfaces mirrored (file:line in the reference tree):
//   FeatureExtractorSettings        include/FeatureExtractor.h:23-66, src/FeatureExtractor.cpp:21-56
//   FeatureExtractor::operator()    include/FeatureExtractor.h:76-93, src/FeatureExtractor.cpp:111-129
//   FeatureExtractor_orb32          include/Feature_orb32.h:12-33, src/Feature_orb32.cpp:11-65
//   FeatureMatcher::SearchByBoW x2  include/FeatureMatcher.h:59-60, src/FeatureMatcher.cc:186-283, 561-660
//   FeatureMatcher::SearchForTriangulation  include/FeatureMatcher.h:66-67, src/FeatureMatcher.cc:662-790
//   Frame (the part the front end reads)  include/Frame.h, src/Frame.cc:171-240, 333-433 -> DeviceFrame (afv_frame_*)
//   FeatureMatcher::SearchByProjection x4 / Fuse x2 / SearchBySim3 / SearchForInitialization (matching cores)
//                                   include/FeatureMatcher.h:47-82, src/FeatureMatcher.cc:73-154, 287-397, 399-557, 794-1064,
//                                   1066-1287, 1291-1506 — the projection geometry stays with the caller, as in the reference
// The reference binary is vslamlab_anyfeature_mono; the stereo branches of the matchers (FeatureMatcher.cc:114-119, 705-747, 880-894,
// 1367-1372) are served through the optional mvuRight members below, mvImagePyramid through ImagePyramid().  What stays with the
// reference is Frame::ComputeStereoMatches itself (Frame.cc:465-645), which is Frame code, not plugin code.
// Error behaviour follows the reference: no exceptions cross the boundary; an empty image leaves the outputs
// untouched (ORBextractor.cc:570-571); an unrecoverable device error terminates (cf. Feature_sift128.cpp:61).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <map>
#include <memory>
#include <utility>
#include <vector>

#include <cstring>

#include "afv_akaze.h"
#include "afv_hip.h"

#ifdef AFV_WITH_OPENCV
#include <opencv2/core.hpp>
#endif

namespace afv {

#ifdef AFV_WITH_OPENCV
using KeyPoint = cv::KeyPoint;
static_assert(sizeof(cv::KeyPoint) == sizeof(afv_keypoint), "cv::KeyPoint layout changed");
#else
struct KeyPoint {  // same member order and size as cv::KeyPoint
    struct { float x, y; } pt;
    float size, angle, response;
    int octave, class_id;
};
struct Mat8 {  // minimal stand-in for a continuous CV_8UC1 cv::Mat
    int rows = 0, cols = 0;
    std::vector<uint8_t> data;
    void create(int r, int c) { rows = r; cols = c; data.assign((size_t)r * c, 0); }
    uint8_t *ptr(int r = 0) { return data.data() + (size_t)r * cols; }
    const uint8_t *ptr(int r = 0) const { return data.data() + (size_t)r * cols; }
    bool empty() const { return rows == 0 || cols == 0; }
    size_t step() const { return (size_t)cols; }
};
struct Image {  // include/Image.h:13-29: the extractor reads grayImg only
    Mat8 grayImg;
};
#endif
static_assert(sizeof(KeyPoint) == sizeof(afv_keypoint), "KeyPoint must be bit-compatible with afv_keypoint");

struct Mat2f { float m[2][2]; };  // Eigen::Matrix<float,2,2> (Types.h:139), row/column symmetric here

struct FeatureExtractorSettings {  // FeatureExtractor.h:23-66
    static inline int numOctaves0 = 8;
    static inline float scaleFactor0 = 1.2f;
    static inline float th0 = 20.0f;
    float scaleFactor = scaleFactor0;
    int nOctaves = numOctaves0;
    float detectTh = th0;
    bool ON_automaticTuning = true;
    static float GetDetectorNominalScaleFactor() { return scaleFactor0; }
    static int GetDetectorNominalNumOctaves() { return numOctaves0; }
    static float GetDetectorNominalThreshold() { return th0; }
};

[[noreturn]] inline void fatal(const char *what, int rc, afv_ctx *ctx) {
    std::fprintf(stderr, "afv: %s failed: %s (%s)\n", what, afv_strerror(rc), ctx ? afv_last_error(ctx) : "");
    std::terminate();
}

// drop-in for FeatureExtractor_orb32: same call operators, GPU behind them
class FeatureExtractor_orb32_hip {
  public:
    std::shared_ptr<FeatureExtractorSettings> settings;

    FeatureExtractor_orb32_hip(const int &nfeatures_, std::shared_ptr<FeatureExtractorSettings> &settings_, int device = 0,
                               int max_width = 640, int max_height = 480)
        : settings(settings_), nfeatures(nfeatures_) {
        afv_orb_params p;
        afv_default_orb_params(&p);
        p.nfeatures = nfeatures_;
        p.nlevels = settings->nOctaves;
        p.scale_factor = settings->scaleFactor;
        p.fast_threshold = int(settings->detectTh);  // Feature_orb32.cpp:30 setFastThreshold(int(detectTh))
        p.max_width = max_width;
        p.max_height = max_height;
        const int rc = afv_create(device, &p, &ctx);
        if (rc != AFV_OK) fatal("afv_create", rc, nullptr);
        cap = afv_max_keypoints_per_frame(ctx);
        mvScaleFactor.resize(settings->nOctaves);  // FeatureExtractor.cpp:77-85
        mvScaleFactor[0] = 1.0f;
        for (int i = 1; i < settings->nOctaves; i++) mvScaleFactor[i] = mvScaleFactor[i - 1] * settings->scaleFactor;
    }
    ~FeatureExtractor_orb32_hip() { afv_destroy(ctx); }
    FeatureExtractor_orb32_hip(const FeatureExtractor_orb32_hip &) = delete;
    FeatureExtractor_orb32_hip &operator=(const FeatureExtractor_orb32_hip &) = delete;

    // 6-argument operator() (FeatureExtractor.cpp:111-121)
    template <class ImageT, class MatT>
    void operator()(const ImageT &img, std::vector<KeyPoint> &keypoints, MatT &descriptors, std::vector<Mat2f> &keyPtsSigma2,
                    std::vector<Mat2f> &keyPtsInf, std::vector<float> &keyPtsSize) {
        (*this)(img, keypoints, descriptors);
        const int n = (int)keypoints.size();
        keyPtsSize.assign(n, 0.f);
        std::vector<float> s2(n), inf(n);
        const int rc = afv_orb_size_sigma(ctx, reinterpret_cast<const afv_keypoint *>(keypoints.data()), n, keyPtsSize.data(),
                                          s2.data(), inf.data());
        if (rc != AFV_OK) fatal("afv_orb_size_sigma", rc, ctx);
        keyPtsSigma2.clear();
        keyPtsInf.clear();
        for (int i = 0; i < n; ++i) {  // computeSigma(SIZE): sigma^2 * I, 1/sigma^2 * I (FeatureExtractor.cpp:159-170)
            keyPtsSigma2.push_back(Mat2f{{{s2[i], 0.f}, {0.f, s2[i]}}});
            keyPtsInf.push_back(Mat2f{{{inf[i], 0.f}, {0.f, inf[i]}}});
        }
    }

    // 3-argument operator() (FeatureExtractor.cpp:123-129)
    template <class ImageT, class MatT>
    void operator()(const ImageT &img, std::vector<KeyPoint> &keypoints, MatT &descriptors) {
        if (settings->ON_automaticTuning) {  // automaticTuning (FeatureExtractor.cpp:195-274): detectTh = th0, once
            settings->detectTh = FeatureExtractorSettings::GetDetectorNominalThreshold();
            settings->ON_automaticTuning = false;
        }
        detectAndCompute(img, keypoints, descriptors);
    }

    // Feature_orb32.cpp:11-18: detect -> quadtree -> describe -> merge, all on the GPU
    template <class ImageT, class MatT>
    void detectAndCompute(const ImageT &img, std::vector<KeyPoint> &keypoints, MatT &descriptors) {
        const auto &g = img.grayImg;
        if (g.empty()) return;  // outputs untouched (ORBextractor.cc:570-571)
        std::vector<KeyPoint> kps((size_t)cap);
        std::vector<uint8_t> desc((size_t)cap * AFV_DESC_BYTES);
        int n = 0;
        const int rc = afv_orb_extract(ctx, g.ptr(0), g.cols, g.rows, (int)row_step(g), reinterpret_cast<afv_keypoint *>(kps.data()),
                                       desc.data(), cap, &n);
        if (rc != AFV_OK) fatal("afv_orb_extract", rc, ctx);
        kps.resize((size_t)n);
        keypoints.swap(kps);  // mergeKeypointLevels clears and refills (FeatureExtractor.cpp:300-307)
        fill_descriptors(descriptors, desc.data(), n);
    }

    // The three virtuals detectAndCompute is composed of (FeatureExtractor.h:123-128), for a host that calls them one by one - a vocabulary
    // builder that keeps keypoints and recomputes descriptors at them (cv::ORB::compute semantics).  detectKeypoints here already
    // includes the quadtree of filterKeypoints (the device never ships cv::ORB::detect's 10x candidate set to the host), so the
    // filterKeypoints override is the identity; detectKeypoints -> filterKeypoints -> computeDescriptors -> mergeKeypointLevels gives
    // detectAndCompute's outputs bit for bit (tests/test_gpu_extract.py::test_plugin_virtuals_one_by_one, adapter_selftest).
    template <class ImageT>
    void detectKeypoints(std::map<int, std::vector<KeyPoint>> &keypoints_level, const ImageT &img, const float & /*detectTh*/, const int & /*nOctaves*/) const {
        const auto &g = img.grayImg;
        if (g.empty()) return;
        std::vector<KeyPoint> kps((size_t)cap);
        int n = 0;
        const int rc = afv_orb_detect(ctx, g.ptr(0), g.cols, g.rows, (int)row_step(g), reinterpret_cast<afv_keypoint *>(kps.data()), cap, &n);
        if (rc != AFV_OK) fatal("afv_orb_detect", rc, ctx);
        for (int i = 0; i < n; ++i) keypoints_level[kps[(size_t)i].octave].push_back(kps[(size_t)i]);  // GetKeypointOctave (Feature_orb32.cpp:55-57)
    }
    template <class MatT>
    void filterKeypoints(std::map<int, std::vector<KeyPoint>> & /*keypoints_level*/, const MatT & /*image*/, const MatT & /*mask*/) const {}
    // descriptors_level[level]: n x 32 rows in the order of keypoints_level[level] (one cv::ORB::compute per level in the reference,
    // Feature_orb32.cpp:49-50; ONE device call here)
    template <class ImageT, class MatT>
    void computeDescriptors(std::map<int, MatT> &descriptors_level, std::map<int, std::vector<KeyPoint>> &keypoints_level, const ImageT &img) const {
        const auto &g = img.grayImg;
        std::vector<KeyPoint> all;
        for (auto &lk : keypoints_level) all.insert(all.end(), lk.second.begin(), lk.second.end());
        std::vector<uint8_t> desc(all.size() * AFV_DESC_BYTES);
        const int rc = afv_orb_compute(ctx, g.ptr(0), g.cols, g.rows, (int)row_step(g), reinterpret_cast<const afv_keypoint *>(all.data()), (int)all.size(),
                                       desc.data());
        if (rc != AFV_OK) fatal("afv_orb_compute", rc, ctx);
        size_t o = 0;
        for (auto &lk : keypoints_level) {
            fill_descriptors(descriptors_level[lk.first], desc.data() + o * AFV_DESC_BYTES, (int)lk.second.size());
            o += lk.second.size();
        }
    }

    // FeatureExtractor::mvImagePyramid (FeatureExtractor.h:142; read by Frame::ComputeStereoMatches, Frame.cc:475,568): the levels of
    // the LAST extracted frame, copied from the device on demand (the mono entry point never asks).  MatT: cv::Mat (CV_8UC1) or Mat8.
    template <class MatT>
    void ImagePyramid(std::vector<MatT> &pyramid) {
        afv_geometry g;
        int rc = afv_get_geometry(ctx, &g);
        if (rc != AFV_OK) fatal("afv_get_geometry", rc, ctx);
        pyramid.resize((size_t)g.nlevels);
        std::vector<uint8_t> tight;
        for (int l = 0; l < g.nlevels; ++l) {
            tight.resize((size_t)g.lw[l] * g.lh[l]);
            rc = afv_debug_get_level(ctx, 0, l, tight.data());
            if (rc != AFV_OK) fatal("afv_debug_get_level", rc, ctx);
#ifdef AFV_WITH_OPENCV
            pyramid[(size_t)l].create(g.lh[l], g.lw[l], CV_8U);
#else
            pyramid[(size_t)l].create(g.lh[l], g.lw[l]);
#endif
            for (int y = 0; y < g.lh[l]; ++y) std::copy(tight.begin() + (size_t)y * g.lw[l], tight.begin() + (size_t)(y + 1) * g.lw[l], pyramid[(size_t)l].ptr(y));
        }
    }

    int GetLevels() { return settings->nOctaves; }
    float GetScaleFactor() { return settings->scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    int GetKeypointOctave(const KeyPoint &kp) const { return kp.octave; }                                  // Feature_orb32.cpp:55-57
    float GetKeypointSize(const KeyPoint &kp) const { return powf(settings->scaleFactor, float(kp.octave)); }  // :59-61
    afv_ctx *context() { return ctx; }

  private:
    template <class M> static size_t row_step(const M &m) {
#ifdef AFV_WITH_OPENCV
        return (size_t)m.step;
#else
        return m.step();
#endif
    }
    template <class M> static void fill_descriptors(M &m, const uint8_t *src, int n) {
#ifdef AFV_WITH_OPENCV
        m.create(n, AFV_DESC_BYTES, CV_8U);
#else
        m.create(n, AFV_DESC_BYTES);
#endif
        for (int i = 0; i < n; ++i) std::copy(src + (size_t)i * AFV_DESC_BYTES, src + (size_t)(i + 1) * AFV_DESC_BYTES, m.ptr(i));
    }
    int nfeatures;
    afv_ctx *ctx = nullptr;
    int cap = 0;
    std::vector<float> mvScaleFactor;
};

// ---- the device-resident Frame (include/afv_hip.h afv_frame_*; reference object: Frame, src/Frame.cc:171-240) ----
// One per Frame the tracker keeps alive (currentFrame, lastFrame, the initial frame): Frame::Frame calls Extract() instead of the plain
// operator(), and from then on SearchByProjection / Fuse / SearchForInitialization / ComputeBoW / SearchByBoW(KF, F) and the promotion to
// a KeyFrame read the frame where the extraction left it ("Tracking through a resident frame").
class DeviceFrame {
  public:
    // mnMinX .. mnMaxY: Frame::ComputeImageBounds (Frame.cc:435-466); distorted: mDistCoef.at<float>(0) != 0 (Frame.cc:405)
    DeviceFrame(afv_ctx *ctx_, float mnMinX, float mnMinY, float mnMaxX, float mnMaxY, bool distorted = false, int grid_cols = 64, int grid_rows = 48)
        : ctx(ctx_) {
        afv_frame_params p{};
        p.struct_size = sizeof(p);
        p.min_x = mnMinX; p.min_y = mnMinY; p.max_x = mnMaxX; p.max_y = mnMaxY;
        p.grid_cols = grid_cols; p.grid_rows = grid_rows;
        p.distorted = distorted ? 1 : 0;
        const int rc = afv_frame_create(ctx, &p, &f);
        if (rc != AFV_OK) fatal("afv_frame_create", rc, ctx);
    }
    ~DeviceFrame() { afv_frame_destroy(f); }
    DeviceFrame(const DeviceFrame &) = delete;
    DeviceFrame &operator=(const DeviceFrame &) = delete;

    // Frame::ExtractFeatures (Frame.cc:242-259): FeatureExtractor::operator() 3-argument form into the resident frame; the host vectors
    // are filled exactly as by FeatureExtractor_orb32_hip::detectAndCompute
    template <class ImageT, class MatT>
    void Extract(const ImageT &img, std::vector<KeyPoint> &keypoints, MatT &descriptors) {
        const auto &g = img.grayImg;
        if (g.empty()) return;
        const int cap = afv_max_keypoints_per_frame(ctx);
        std::vector<KeyPoint> kps((size_t)cap);
        std::vector<uint8_t> desc((size_t)cap * AFV_DESC_BYTES);
        int n = 0;
#ifdef AFV_WITH_OPENCV
        const size_t step = (size_t)g.step;
#else
        const size_t step = g.step();
#endif
        const int rc = afv_frame_extract(f, g.ptr(0), g.cols, g.rows, (int)step, reinterpret_cast<afv_keypoint *>(kps.data()), desc.data(), cap, &n);
        if (rc != AFV_OK) fatal("afv_frame_extract", rc, ctx);
        kps.resize((size_t)n);
        keypoints.swap(kps);
#ifdef AFV_WITH_OPENCV
        descriptors.create(n, AFV_DESC_BYTES, CV_8U);
#else
        descriptors.create(n, AFV_DESC_BYTES);
#endif
        for (int i = 0; i < n; ++i) std::copy(desc.data() + (size_t)i * AFV_DESC_BYTES, desc.data() + (size_t)(i + 1) * AFV_DESC_BYTES, descriptors.ptr(i));
    }
    // Frame::UndistortKeyPoints for a distorted camera: mvKeysUn after cv::undistortPoints (Frame.cc:411-432); builds the grid
    void SetUndistorted(const std::vector<KeyPoint> &mvKeysUn) {
        std::vector<float> x(mvKeysUn.size()), y(mvKeysUn.size());
        for (size_t i = 0; i < mvKeysUn.size(); ++i) {
            x[i] = mvKeysUn[i].pt.x;
            y[i] = mvKeysUn[i].pt.y;
        }
        const int rc = afv_frame_set_undistorted(f, x.data(), y.data());
        if (rc != AFV_OK) fatal("afv_frame_set_undistorted", rc, ctx);
    }
    int N() const { return afv_frame_count(f); }
    afv_frame *handle() { return f; }
    afv_ctx *context() { return ctx; }

  private:
    afv_ctx *ctx;
    afv_frame *f = nullptr;
};

// ---- matcher side: flat view of what FeatureMatcher reads from KeyFrame / Frame ----
using FeatureVector = std::map<unsigned, std::vector<unsigned>>;  // DBoW2::FeatureVector (node id -> feature indices)

// The reference's matchers dispatch on the descriptor type (FeatureMatcher::DescriptorDistance, FeatureMatcher.cc:1508-1531): the views
// below carry it as `desc_bytes` (binary rows: 32 ORB, 61 AKAZE, 48 BRISK ..., Hamming) or `float_dim` > 0 (float rows - SIFT128, SURF64,
// KAZE64, R2D2 ...: `descriptors` then points to N x float_dim floats, mDescriptors.ptr<float>(), and the distance is
// cv::norm(a, b, NORM_L2SQR) as a float, Feature_sift128.cpp:132-134).  Both sides of a call must agree.
struct FeatureView {
    const uint8_t *descriptors = nullptr;  // N x desc_bytes (or N x float_dim floats), continuous (KeyFrame::mDescriptors)
    int N = 0;
    int desc_bytes = AFV_DESC_BYTES, float_dim = 0;
    const FeatureVector *featVec = nullptr;  // nullptr => brute force
    const uint8_t *valid = nullptr;          // map point exists && !isBad() (triangulation: has a map point)
    const float *angles = nullptr;           // mvKeysUn[i].angle
    const float *x = nullptr, *y = nullptr;  // mvKeysUn[i].pt       (triangulation)
    const float *sigma2 = nullptr;           // GetKeyPt1DSigma2(i)  (triangulation)
    const float *mvuRight = nullptr;         // KeyFrame::mvuRight (stereo keyframes; nullptr: monocular)  (triangulation)
};

// the queries' descriptors named as rows (slot, idx) of a keyframe table instead of carried by value (afv_proj_queries::qref_*)
struct TableRefs {
    afv_table *table = nullptr;
    const int32_t *slot = nullptr, *idx = nullptr;
};

class FeatureMatcherHip {
  public:
    static inline float TH_LOW = 0.0f, TH_HIGH = 0.0f;  // FeatureMatcher.cc:56-59
    static void setDescriptorDistanceThresholds(float matchingTh) { TH_LOW = TH_HIGH = matchingTh; }  // :1533-1545

    FeatureMatcherHip(afv_ctx *ctx_, float nnratio = 0.6f, bool checkOri = true) : ctx(ctx_), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

    // SearchByBoW(KF1, KF2, vpMatches12) (FeatureMatcher.cc:561-660): matches12[i] = index in KF2 or -1
    int SearchByBoW(const FeatureView &kf1, const FeatureView &kf2, std::vector<int> &matches12) {
        return run(kf1, kf2, AFV_MATCH_KF_KF, matches12);
    }
    // SearchByBoW(KF, Frame, vpMapPointMatches) (:186-283): matchesF[idxF] = index in KF or -1
    int SearchByBoW_Frame(const FeatureView &kf, const FeatureView &frame, std::vector<int> &matchesF) {
        return run(kf, frame, AFV_MATCH_KF_FRAME, matchesF);
    }
    // SearchForTriangulation (:662-790; stereo branches :705-709, :727-731, :741 through FeatureView::mvuRight): vMatchedPairs ascending idx1
    int SearchForTriangulation(const FeatureView &kf1, const FeatureView &kf2, const float F12[9], float ex, float ey,
                               std::vector<std::pair<size_t, size_t>> &vMatchedPairs, bool bOnlyStereo = false) {
        Csr c1(kf1), c2(kf2);
        afv_tri_job t{};
        t.struct_size = sizeof(t);
        fill(t.bow, kf1, kf2, c1, c2, AFV_MATCH_KF_KF);
        t.x1 = kf1.x; t.y1 = kf1.y; t.x2 = kf2.x; t.y2 = kf2.y; t.sigma2_2 = kf2.sigma2;
        for (int i = 0; i < 9; ++i) t.F12[i] = F12[i];
        t.ex = ex; t.ey = ey;
        t.u_right1 = kf1.mvuRight; t.u_right2 = kf2.mvuRight; t.only_stereo = bOnlyStereo ? 1 : 0;
        std::vector<int32_t> m((size_t)std::max(kf1.N, 1), -1);
        int32_t n = 0;
        const int rc = afv_match_triangulation(ctx, &t, 1, m.data(), &n);
        if (rc != AFV_OK) fatal("afv_match_triangulation", rc, ctx);
        vMatchedPairs.clear();
        for (int i = 0; i < kf1.N; ++i)
            if (m[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m[i]);
        return n;
    }

    // ---- projection-guided searches.  The caller evaluates the projections exactly as the reference does
    // and hands over, per query in the reference's iteration order, (u, v, r, size band); see include/afv_hip.h ----
    struct FrameGridView {  // what the matchers read from a Frame / KeyFrame (Frame.cc:100-101, Frame.h:40-41)
        const uint8_t *descriptors = nullptr; int N = 0;
        int desc_bytes = AFV_DESC_BYTES, float_dim = 0;  // descriptor kind (see FeatureView); the queries carry the same kind
        const float *x = nullptr, *y = nullptr;  // mvKeysUn[i].pt
        const float *size = nullptr;             // keyPtsSize[i]
        const float *angle = nullptr;            // mvKeysUn[i].angle
        const uint8_t *occupied = nullptr;       // pts[i] (&& observations > 0 where the reference asks) ; nullptr = none
        const float *inf = nullptr;              // GetKeyPt1DInf(i) (Fuse only)
        const float *mvuRight = nullptr;         // stereo frames / keyframes (nullptr: monocular): FeatureMatcher.cc:114, :880, :1367
        float mnMinX = 0.f, mnMinY = 0.f, mfGridElementWidthInv = 0.f, mfGridElementHeightInv = 0.f;
        int grid_cols = 64, grid_rows = 48;      // FRAME_GRID_COLS / ROWS
        float sizeTolerance = 1.2f, invSizeTolerance = 1.0f / 1.2f;  // Frame.cc:73-74
    };
    struct ProjectionQueries {
        const uint8_t *descriptors = nullptr; int n = 0;  // pMP->GetDescriptor() / LastFrame.mDescriptors
        const float *u = nullptr, *v = nullptr, *r = nullptr, *min_size = nullptr, *max_size = nullptr;
        const uint8_t *valid = nullptr;      // nullptr = all
        const float *angle = nullptr;        // last-frame style searches with orientation check
        const uint8_t *occupies = nullptr;   // pMP->NumberOfObservations() > 0 (nullptr = yes)
        // stereo: projected right-image coordinate (pMP->mTrackProjXR :116 / u - mbf * invzc :1369 / ur :885) and the gate on
        // |ur - mvuRight| of the two SearchByProjection flavours that have one (r * pMP->trackSigma :117 / the window radius :1371)
        const float *ur = nullptr, *er_max = nullptr;
    };
    // SearchByProjection(F, vpMapPoints, radiusTh) (FeatureMatcher.cc:73-154): assign[i] = query stored in F.pts[i] or -1
    int SearchByProjection(const FrameGridView &F, const ProjectionQueries &q, std::vector<int> &assign) {
        return projection(F, q, TH_HIGH, AFV_PROJ_LOCALMAP, false, assign);
    }
    // SearchByProjection(CurrentFrame, LastFrame, radiusTh) (:1291-1402)
    int SearchByProjection_LastFrame(const FrameGridView &F, const ProjectionQueries &q, std::vector<int> &assign) {
        return projection(F, q, TH_HIGH, AFV_PROJ_LASTFRAME, mbCheckOrientation, assign);
    }
    // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, radiusTh, useHighMatchingThreshold) (:1404-1506, relocalisation)
    int SearchByProjection_Reloc(const FrameGridView &F, const ProjectionQueries &q, float th, std::vector<int> &assign) {
        return projection(F, q, th, AFV_PROJ_LASTFRAME, mbCheckOrientation, assign, false);
    }
    // SearchByProjection(pKF, Scw, vpPoints, vpMatched, radiusTh) (:287-397, loop closing)
    int SearchByProjection_Sim3(const FrameGridView &KF, const ProjectionQueries &q, std::vector<int> &assign) {
        return projection(KF, q, TH_LOW, AFV_PROJ_LASTFRAME, false, assign, false);
    }
    // Fuse(pKF, vpMapPoints, radiusTh) (:794-940): best[q] = keypoint index or -1; with KF.inf == nullptr it is
    // Fuse(pKF, Scw, vpPoints, radiusTh, vpReplacePoint) (:942-1064)
    int Fuse(const FrameGridView &KF, const ProjectionQueries &q, std::vector<int> &best) {
        afv_proj_job j = proj_job(KF, q);
        j.th_high = TH_LOW;
        best.assign((size_t)std::max(q.n, 1), -1);
        int32_t n = 0;
        const int rc = afv_match_fuse(ctx, &j, 1, best.data(), &n);
        if (rc != AFV_OK) fatal("afv_match_fuse", rc, ctx);
        best.resize((size_t)q.n);
        return n;
    }
    // SearchBySim3(pKF1, pKF2, vpMatches12, ...) (:1066-1287): q1 = KF1's map points projected into KF2, q2 the reverse
    int SearchBySim3(const FrameGridView &KF1, const ProjectionQueries &q1, const FrameGridView &KF2, const ProjectionQueries &q2,
                     std::vector<int> &matches12) {
        afv_proj_job j12 = proj_job(KF2, q1), j21 = proj_job(KF1, q2);
        j12.th_high = j21.th_high = TH_HIGH;
        matches12.assign((size_t)std::max(q1.n, 1), -1);
        int32_t n = 0;
        const int rc = afv_match_sim3(ctx, &j12, &j21, matches12.data(), &n);
        if (rc != AFV_OK) fatal("afv_match_sim3", rc, ctx);
        matches12.resize((size_t)q1.n);
        return n;
    }
    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:399-557): q = F1's features
    int SearchForInitialization(const FrameGridView &F2, const ProjectionQueries &q, std::vector<int> &vnMatches12) {
        afv_proj_job j = proj_job(F2, q);
        j.th_high = TH_LOW;
        j.check_orientation = mbCheckOrientation;
        vnMatches12.assign((size_t)std::max(q.n, 1), -1);
        int32_t n = 0;
        const int rc = afv_match_initialization(ctx, &j, 1, vnMatches12.data(), &n);
        if (rc != AFV_OK) fatal("afv_match_initialization", rc, ctx);
        vnMatches12.resize((size_t)q.n);
        return n;
    }

    // ---- the same searches against a resident frame: nothing of the frame is uploaded, only the queries travel (or, with q_table /
    // q_slot / q_idx, only 8 bytes per query: a map point's descriptor named as the keyframe row MapPoint::ComputeDistinctDescriptors
    // copied it from) ----
    int SearchByProjection(DeviceFrame &F, const ProjectionQueries &q, const uint8_t *occupied, std::vector<int> &assign, TableRefs refs = TableRefs()) {
        return frame_projection(F, q, occupied, TH_HIGH, AFV_PROJ_LOCALMAP, false, refs, assign);
    }
    int SearchByProjection_LastFrame(DeviceFrame &F, const ProjectionQueries &q, const uint8_t *occupied, std::vector<int> &assign,
                                     TableRefs refs = TableRefs()) {
        return frame_projection(F, q, occupied, TH_HIGH, AFV_PROJ_LASTFRAME, mbCheckOrientation, refs, assign);
    }
    int SearchByProjection_Reloc(DeviceFrame &F, const ProjectionQueries &q, const uint8_t *occupied, float th, std::vector<int> &assign) {
        ProjectionQueries mono = q;
        mono.ur = mono.er_max = nullptr;  // no mvuRight branch in :1404-1506
        return frame_projection(F, mono, occupied, th, AFV_PROJ_LASTFRAME, mbCheckOrientation, TableRefs(), assign);
    }
    int Fuse(DeviceFrame &KF, const ProjectionQueries &q, bool reprojection_gate, std::vector<int> &best) {
        afv_proj_queries Q = frame_queries(q, nullptr, TH_LOW, 0, false, TableRefs());
        best.assign((size_t)std::max(q.n, 1), -1);
        int32_t n = 0;
        const int rc = afv_frame_match_fuse(KF.handle(), &Q, reprojection_gate ? 1 : 0, best.data(), &n);
        if (rc != AFV_OK) fatal("afv_frame_match_fuse", rc, ctx);
        best.resize((size_t)q.n);
        return n;
    }
    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:399-557) between two resident frames; vbPrevMatched is
    // refreshed like :551-553 when F2's undistorted keypoints are given
    int SearchForInitialization(DeviceFrame &F1, DeviceFrame &F2, std::vector<float> &prev_x, std::vector<float> &prev_y, int windowSize,
                                std::vector<int> &vnMatches12, const std::vector<KeyPoint> *F2_mvKeysUn = nullptr) {
        const int n1 = F1.N();
        vnMatches12.assign((size_t)std::max(n1, 1), -1);
        int32_t n = 0;
        const int rc = afv_frame_match_initialization(F1.handle(), F2.handle(), prev_x.data(), prev_y.data(), (float)windowSize, TH_LOW, mfNNratio,
                                                      mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n);
        if (rc != AFV_OK) fatal("afv_frame_match_initialization", rc, ctx);
        vnMatches12.resize((size_t)n1);
        if (F2_mvKeysUn)
            for (int i = 0; i < n1; ++i)
                if (vnMatches12[i] >= 0) {
                    prev_x[i] = (*F2_mvKeysUn)[(size_t)vnMatches12[i]].pt.x;
                    prev_y[i] = (*F2_mvKeysUn)[(size_t)vnMatches12[i]].pt.y;
                }
        return n;
    }

  private:
    afv_proj_queries frame_queries(const ProjectionQueries &q, const uint8_t *occupied, float th, int mode, bool ori, TableRefs refs) const {
        afv_proj_queries Q{};
        Q.struct_size = sizeof(Q);
        Q.nq = q.n; Q.qdesc = refs.table ? nullptr : q.descriptors; Q.desc_bytes = AFV_DESC_BYTES;
        Q.qvalid = q.valid; Q.qu = q.u; Q.qv = q.v; Q.qr = q.r; Q.qmin_size = q.min_size; Q.qmax_size = q.max_size;
        Q.qangle = q.angle; Q.qoccupies = q.occupies; Q.q_ur = q.ur; Q.q_er_max = q.er_max;
        Q.occupied = occupied;
        Q.th_high = th; Q.nnratio = mfNNratio; Q.check_orientation = ori ? 1 : 0; Q.mode = mode;
        Q.qref_table = refs.table; Q.qref_slot = refs.slot; Q.qref_idx = refs.idx;
        return Q;
    }
    int frame_projection(DeviceFrame &F, const ProjectionQueries &q, const uint8_t *occupied, float th, int mode, bool ori, TableRefs refs,
                         std::vector<int> &assign) {
        afv_proj_queries Q = frame_queries(q, occupied, th, mode, ori, refs);
        const int n = F.N();
        assign.assign((size_t)std::max(n, 1), -1);
        int32_t nm = 0;
        const int rc = afv_frame_match_projection(F.handle(), &Q, assign.data(), &nm);
        if (rc != AFV_OK) fatal("afv_frame_match_projection", rc, ctx);
        assign.resize((size_t)n);
        return nm;
    }
    afv_proj_job proj_job(const FrameGridView &F, const ProjectionQueries &q) const {
        afv_proj_job j{};
        j.struct_size = sizeof(j);
        j.desc = F.descriptors; j.n = F.N; j.desc_bytes = F.float_dim ? 4 * F.float_dim : F.desc_bytes; j.float_dim = F.float_dim;
        j.x = F.x; j.y = F.y; j.size = F.size; j.angle = F.angle; j.occupied = F.occupied; j.inf = F.inf;
        j.min_x = F.mnMinX; j.min_y = F.mnMinY; j.grid_inv_w = F.mfGridElementWidthInv; j.grid_inv_h = F.mfGridElementHeightInv;
        j.grid_cols = F.grid_cols; j.grid_rows = F.grid_rows;
        j.nq = q.n; j.qdesc = q.descriptors; j.qvalid = q.valid; j.qu = q.u; j.qv = q.v; j.qr = q.r;
        j.qmin_size = q.min_size; j.qmax_size = q.max_size; j.qangle = q.angle; j.qoccupies = q.occupies;
        j.nnratio = mfNNratio; j.size_tol = F.sizeTolerance; j.inv_size_tol = F.invSizeTolerance;
        j.u_right = F.mvuRight; j.q_ur = q.ur; j.q_er_max = q.er_max;
        return j;
    }
    // stereo = false: the relocalisation / Sim3 flavours share the entry point but have no mvuRight branch (:287-397, :1404-1506)
    int projection(const FrameGridView &F, const ProjectionQueries &q, float th, int mode, bool ori, std::vector<int> &assign, bool stereo = true) {
        afv_proj_job j = proj_job(F, q);
        if (!stereo) j.u_right = nullptr;
        j.th_high = th; j.mode = mode; j.check_orientation = ori;
        assign.assign((size_t)std::max(F.N, 1), -1);
        int32_t n = 0;
        const int rc = afv_match_projection(ctx, &j, 1, assign.data(), &n);
        if (rc != AFV_OK) fatal("afv_match_projection", rc, ctx);
        assign.resize((size_t)F.N);
        return n;
    }
    struct Csr {
        std::vector<int32_t> id, ptr, idx;
        explicit Csr(const FeatureView &v) {
            if (!v.featVec) return;
            ptr.push_back(0);
            for (const auto &kv : *v.featVec) {
                id.push_back((int32_t)kv.first);
                for (unsigned f : kv.second) idx.push_back((int32_t)f);
                ptr.push_back((int32_t)idx.size());
            }
        }
    };
    void fill(afv_match_job &j, const FeatureView &a, const FeatureView &b, const Csr &ca, const Csr &cb, int mode) const {
        j.desc1 = a.descriptors; j.n1 = a.N; j.desc2 = b.descriptors; j.n2 = b.N;
        if (a.float_dim != b.float_dim || a.desc_bytes != b.desc_bytes) fatal("FeatureMatcherHip: the two sides carry different descriptor kinds", AFV_EINVAL, ctx);
        j.desc_bytes = a.float_dim ? 4 * a.float_dim : a.desc_bytes;
        if (a.float_dim) mode |= AFV_MATCH_FLOAT32;
        j.node_id1 = ca.id.data(); j.seg_ptr1 = ca.ptr.data(); j.seg_idx1 = ca.idx.data(); j.nnodes1 = (int32_t)ca.id.size();
        j.node_id2 = cb.id.data(); j.seg_ptr2 = cb.ptr.data(); j.seg_idx2 = cb.idx.data(); j.nnodes2 = (int32_t)cb.id.size();
        j.valid1 = a.valid; j.valid2 = b.valid; j.angle1 = a.angles; j.angle2 = b.angles;
        j.th_low = TH_LOW; j.nnratio = mfNNratio; j.check_orientation = mbCheckOrientation && a.angles && b.angles; j.mode = mode;
    }
    int run(const FeatureView &a, const FeatureView &b, int mode, std::vector<int> &out) {
        Csr ca(a), cb(b);
        afv_match_job j{};
        fill(j, a, b, ca, cb, mode);
        const int nout = mode == AFV_MATCH_KF_FRAME ? b.N : a.N;
        out.assign((size_t)std::max(nout, 1), -1);
        int32_t n = 0;
        const int rc = afv_match_bow(ctx, &j, 1, out.data(), &n);
        if (rc != AFV_OK) fatal("afv_match_bow", rc, ctx);
        out.resize((size_t)nout);
        return n;
    }
    afv_ctx *ctx;
    float mfNNratio;
    bool mbCheckOrientation;
};


// ---- AKAZE61 plugin (reference include/Feature_akaze61.h, src/Feature_akaze61.cpp) ----
class FeatureExtractor_akaze61_hip {
  public:
    FeatureExtractor_akaze61_hip(const int &nfeatures_, std::shared_ptr<FeatureExtractorSettings> &settings_, int device = 0,
                                 int max_width = 1280, int max_height = 720)
        : settings(settings_), nfeatures(nfeatures_) {
        afv_akaze_params p;
        afv_akaze_default_params(&p);
        p.omax = FeatureExtractorSettings::numOctaves0 / 4;        // Feature_akaze61.cpp:12
        p.nsublevels = FeatureExtractorSettings::numOctaves0 / 2;  // :13
        p.dthreshold = settings->detectTh;                         // :14
        p.nfeatures = nfeatures;
        p.scale_factor = FeatureExtractorSettings::scaleFactor0;
        p.max_width = max_width; p.max_height = max_height; p.max_batch = 1;
        const int rc = afv_akaze_create(device, &p, &akz);
        if (rc != AFV_OK) fatal("afv_akaze_create", rc, nullptr);
        cap = nfeatures + 64;
        kbuf.resize((size_t)cap);
        dbuf.resize((size_t)cap * 61);
    }
    ~FeatureExtractor_akaze61_hip() { afv_akaze_destroy(akz); }
    FeatureExtractor_akaze61_hip(const FeatureExtractor_akaze61_hip &) = delete;
    FeatureExtractor_akaze61_hip &operator=(const FeatureExtractor_akaze61_hip &) = delete;

    // detectAndCompute (Feature_akaze61.cpp:17-24): initializeExtractor + detectKeypoints + filterKeypoints + computeDescriptors +
    // mergeKeypointLevels in one call; descriptors = N x 61 CV_8U
    template <class ImageT, class MatT>
    void detectAndCompute(const ImageT &img, std::vector<KeyPoint> &keypoints, MatT &descriptors) {
        const auto &g = img.grayImg;
        if (g.empty()) return;
        int32_t n = 0;
        const int rc = afv_akaze_extract(akz, g.ptr(), 1, g.cols, g.rows, (int)row_step(g), 0, kbuf.data(), dbuf.data(), cap, &n);
        if (rc == AFV_ECAPACITY) {
            // a device-side capacity was exceeded on a very busy frame: the outputs are a truncated but valid result
            // (the reference has no such limit; losing the surplus beats terminating the SLAM process)
            std::fprintf(stderr, "afv: afv_akaze_extract: %s (%s) - keeping %d keypoints\n", afv_strerror(rc), afv_akaze_last_error(akz), (int)n);
            if (n < 0 || n > cap) n = 0;
        } else if (rc != AFV_OK) {
            std::fprintf(stderr, "afv: afv_akaze_extract failed: %s (%s)\n", afv_strerror(rc), afv_akaze_last_error(akz));
            std::terminate();
        }
        keypoints.resize((size_t)n);
        static_assert(sizeof(KeyPoint) == sizeof(afv_keypoint), "KeyPoint layout");
        if (n) std::memcpy(static_cast<void *>(keypoints.data()), kbuf.data(), (size_t)n * sizeof(afv_keypoint));
#ifdef AFV_WITH_OPENCV
        descriptors.create(n, 61, CV_8U);
#else
        descriptors.create(n, 61);
#endif
        for (int i = 0; i < n; ++i) std::memcpy(descriptors.ptr(i), dbuf.data() + (size_t)i * 61, 61);
    }
    int GetKeypointOctave(const KeyPoint &kp) const { return kp.class_id; }  // Feature_akaze61.cpp:55-57
    float GetKeypointSize(const KeyPoint &kp) const { return powf(FeatureExtractorSettings::scaleFactor0, float(GetKeypointOctave(kp))); }  // :59-61
    std::shared_ptr<FeatureExtractorSettings> settings;

  private:
    template <class M> static size_t row_step(const M &m) {  // ROI / non-continuous images: the true row stride
#ifdef AFV_WITH_OPENCV
        return (size_t)m.step;
#else
        return m.step();
#endif
    }
    int nfeatures, cap = 0;
    afv_akaze *akz = nullptr;
    std::vector<afv_keypoint> kbuf;
    std::vector<uint8_t> dbuf;
};

// ---- Vocabulary::transform on the GPU (reference include/Vocabulary.h, src/Vocabulary.cpp:156-206) ----
using BowVector = std::map<unsigned, double>;  // DBoW2::BowVector (word id -> weight)

class VocabularyHip {
  public:
    // nodes in DBoW2 id order (node 0 = root): parent id, leaf flag, 32-byte descriptor, weight — what loadFromTextFile reads
    VocabularyHip(afv_ctx *ctx_, int k_, int L_, const std::vector<int> &parent, const std::vector<uint8_t> &is_leaf,
                  const std::vector<uint8_t> &node_desc, const std::vector<double> &weight_)
        : ctx(ctx_), k(k_), L(L_), weight(weight_) {
        const int n = (int)parent.size();
        std::vector<int32_t> child_ptr((size_t)n + 1, 0), child_idx((size_t)std::max(n - 1, 1));
        for (int i = 1; i < n; ++i) child_ptr[(size_t)parent[i] + 1]++;
        for (int i = 0; i < n; ++i) child_ptr[i + 1] += child_ptr[i];
        std::vector<int32_t> fill(child_ptr.begin(), child_ptr.end() - 1);
        for (int i = 1; i < n; ++i) child_idx[(size_t)fill[parent[i]]++] = i;  // children keep their id order
        word_id.assign((size_t)n, -1);
        int words = 0;
        for (int i = 0; i < n; ++i)
            if (is_leaf[i]) word_id[i] = words++;
        int rc = afv_vocab_create(ctx, k, L, n, child_ptr.data(), child_idx.data(), node_desc.data(), 32, &voc);
        if (rc != AFV_OK) fatal("afv_vocab_create", rc, ctx);
        std::vector<uint8_t> stopped((size_t)n, 0);  // DBoW2 transform: a word enters the vectors only if (w > 0)
        bool any = false;
        for (int i = 0; i < n; ++i)
            if (is_leaf[i] && !(weight[(size_t)i] > 0)) stopped[(size_t)i] = 1, any = true;
        if (any) {
            rc = afv_vocab_set_stopped(ctx, voc, stopped.data());
            if (rc != AFV_OK) fatal("afv_vocab_set_stopped", rc, ctx);
        }
    }
    ~VocabularyHip() { afv_vocab_destroy(ctx, voc); }
    VocabularyHip(const VocabularyHip &) = delete;
    VocabularyHip &operator=(const VocabularyHip &) = delete;

    // transform(mDescriptors, mBowVec, mFeatVec) with levelsup = 4 (Vocabulary.cpp:156-206)
    void transform(const uint8_t *descriptors, int n, BowVector &bow, FeatureVector &fv, int levelsup = 4) {
        bow.clear();
        fv.clear();
        std::vector<int32_t> leaf((size_t)std::max(n, 1)), nid((size_t)std::max(n, 1));
        const int rc = afv_bow_transform(ctx, voc, descriptors, n, levelsup, leaf.data(), nid.data());
        if (rc != AFV_OK) fatal("afv_bow_transform", rc, ctx);
        for (int i = 0; i < n; ++i) {
            const double w = weight[(size_t)leaf[i]];
            if (w > 0) {
                bow[(unsigned)word_id[(size_t)leaf[i]]] += w;  // BowVector::addWeight
                fv[(unsigned)nid[i]].push_back((unsigned)i);    // FeatureVector::addFeature
            }
        }
        double s = 0;
        for (const auto &kv : bow) s += std::fabs(kv.second);
        if (s > 0)
            for (auto &kv : bow) kv.second /= s;  // BowVector::normalize(L1)
    }

    // Frame::ComputeBoW (Frame.cc:397-401) on a resident frame: mBowVec / mFeatVec for the host's own use (KeyFrameDatabase scoring);
    // the FeatureVector also stays on the device with the frame (SearchByBoW(KF, F) through afv_table_match_bow_frame_h, promotion)
    void transform(DeviceFrame &F, BowVector &bow, FeatureVector &fv, int levelsup = 4) {
        bow.clear();
        fv.clear();
        const int n = F.N();
        std::vector<int32_t> leaf((size_t)std::max(n, 1)), nid((size_t)std::max(n, 1));
        const int rc = afv_frame_bow_transform(F.handle(), voc, levelsup, leaf.data(), nid.data(), nullptr);
        if (rc != AFV_OK) fatal("afv_frame_bow_transform", rc, ctx);
        for (int i = 0; i < n; ++i) {
            const double w = weight[(size_t)leaf[i]];
            if (w > 0) {
                bow[(unsigned)word_id[(size_t)leaf[i]]] += w;
                fv[(unsigned)nid[i]].push_back((unsigned)i);
            }
        }
        double s = 0;
        for (const auto &kv : bow) s += std::fabs(kv.second);
        if (s > 0)
            for (auto &kv : bow) kv.second /= s;
    }
    afv_vocab *handle() { return voc; }

  private:
    afv_ctx *ctx;
    int k, L;
    std::vector<double> weight;
    std::vector<int32_t> word_id;
    afv_vocab *voc = nullptr;
};

}  // namespace afv
