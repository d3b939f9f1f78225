"""Host-side mirror of the reference's AKAZE61 plugin (include/Feature_akaze61.h, src/Feature_akaze61.cpp) over the C-ABI of
include/afv_akaze.h.  Nothing here computes on the CPU: without the HIP library / a GPU the constructors raise."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ptr

MAX_LEVELS, MAX_FED = 16, 32
LT, LSMOOTH, LX, LY, LDET = 0, 1, 2, 3, 4


class AkazeParams(C.Structure):
    _fields_ = [("omax", C.c_int32), ("nsublevels", C.c_int32), ("soffset", C.c_float), ("derivative_factor", C.c_float),
                ("dthreshold", C.c_float), ("min_dthreshold", C.c_float), ("kcontrast_percentile", C.c_float),
                ("kcontrast_nbins", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32), ("max_batch", C.c_int32),
                ("nfeatures", C.c_int32), ("scale_factor", C.c_float)]


class AkazeLevel(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("octave", C.c_int32), ("sublevel", C.c_int32), ("sigma_size", C.c_int32),
                ("esigma", C.c_float), ("etime", C.c_float), ("nsteps", C.c_int32), ("tau", C.c_float * MAX_FED)]


class AkazePlan(C.Structure):
    _fields_ = [("nlevels", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("lv", AkazeLevel * MAX_LEVELS),
                ("gauss_soffset", C.c_float * 32), ("ksize_soffset", C.c_int32), ("gauss_one", C.c_float * 8), ("ksize_one", C.c_int32)]


def default_params(num_octaves=8, detection_th=0.0005, max_width=1280, max_height=720, max_batch=1, nfeatures=1000, scale_factor=1.1892):
    """AKAZEOptions as FeatureExtractor_akaze61's constructor sets them (Feature_akaze61.cpp:9-15) from
    settings/akaze61_settings.yaml (numOctaves 8, detectionTh 0.0005)"""
    p = AkazeParams()
    _lib.load().afv_akaze_default_params(C.byref(p))
    p.omax = num_octaves // 4
    p.nsublevels = num_octaves // 2
    p.dthreshold = detection_th
    p.max_width, p.max_height, p.max_batch = max_width, max_height, max_batch
    p.nfeatures, p.scale_factor = nfeatures, scale_factor
    return p


def plan_for(params, w, h):
    plan = AkazePlan()
    rc = _lib.load().afv_akaze_plan_for(C.byref(params), int(w), int(h), C.byref(plan))
    if rc:
        raise ValueError("afv_akaze_plan_for: %s" % _lib.strerror(rc))
    return plan


class AkazeContext:
    def __init__(self, params=None, device=0):
        self.lib = _lib.load()
        self.params = params or default_params()
        h = C.c_void_p()
        rc = self.lib.afv_akaze_create(int(device), C.byref(self.params), C.byref(h))
        if rc:
            raise RuntimeError("afv_akaze_create failed: %s (the AKAZE path has no CPU fallback)" % _lib.strerror(rc))
        self.handle = h
        self.plan = None

    def close(self):
        if self.handle:
            self.lib.afv_akaze_destroy(self.handle)
            self.handle = None

    def check(self, rc, what):
        if rc:
            raise RuntimeError("%s: %s (%s)" % (what, _lib.strerror(rc), self.lib.afv_akaze_last_error(self.handle).decode()))

    def scale_space(self, frames):
        """frames: (H, W) or (B, H, W) uint8 host array"""
        frames = np.ascontiguousarray(frames, np.uint8)
        if frames.ndim == 2:
            frames = frames[None]
        b, h, w = frames.shape
        self.check(self.lib.afv_akaze_scale_space(self.handle, ptr(frames), b, w, h, w, w * h), "afv_akaze_scale_space")
        self.plan = plan_for(self.params, w, h)
        return self.plan

    def scale_space_device(self, frames_t):
        """frames_t: torch uint8 CUDA tensor (B, H, W), contiguous; asynchronous"""
        b, h, w = frames_t.shape
        self.check(self.lib.afv_akaze_scale_space_device(self.handle, C.c_void_p(frames_t.data_ptr()), b, w, h, w, w * h),
                   "afv_akaze_scale_space_device")
        if self.plan is None or (self.plan.w, self.plan.h) != (w, h):
            self.plan = plan_for(self.params, w, h)

    def synchronize(self):
        self.check(self.lib.afv_akaze_synchronize(self.handle), "afv_akaze_synchronize")

    def plane(self, frame, level, which):
        L = self.plan.lv[level]
        out = np.empty((L.h, L.w), np.float32)
        self.check(self.lib.afv_akaze_get_plane(self.handle, frame, level, which, ptr(out)), "afv_akaze_get_plane")
        return out

    def kcontrast(self, frame=0):
        v = C.c_float()
        self.check(self.lib.afv_akaze_get_kcontrast(self.handle, frame, C.byref(v)), "afv_akaze_get_kcontrast")
        return float(v.value)

    def detect(self):
        """Feature_Detection on the current scale space (asynchronous)"""
        self.check(self.lib.afv_akaze_detect(self.handle), "afv_akaze_detect")

    def candidates(self, frame, level):
        n = C.c_int()
        self.check(self.lib.afv_akaze_get_candidates(self.handle, frame, level, None, 0, C.byref(n)), "afv_akaze_get_candidates")
        out = np.zeros(max(n.value, 1), np.int32)
        self.check(self.lib.afv_akaze_get_candidates(self.handle, frame, level, ptr(out), len(out), C.byref(n)), "afv_akaze_get_candidates")
        return out[:n.value].copy()

    def keypoints(self, frame=0):
        from .extractor import KP_DTYPE
        n = C.c_int()
        self.check(self.lib.afv_akaze_get_keypoints(self.handle, frame, None, 0, C.byref(n)), "afv_akaze_get_keypoints")
        out = np.zeros(max(n.value, 1), KP_DTYPE)
        self.check(self.lib.afv_akaze_get_keypoints(self.handle, frame, ptr(out), len(out), C.byref(n)), "afv_akaze_get_keypoints")
        return out[:n.value].copy()

    def describe(self):
        self.check(self.lib.afv_akaze_describe(self.handle), "afv_akaze_describe")

    def features(self, frame=0):
        from .extractor import KP_DTYPE
        n = C.c_int()
        self.check(self.lib.afv_akaze_get_features(self.handle, frame, None, None, 0, C.byref(n)), "afv_akaze_get_features")
        kps = np.zeros(max(n.value, 1), KP_DTYPE); desc = np.zeros((max(n.value, 1), 61), np.uint8)
        self.check(self.lib.afv_akaze_get_features(self.handle, frame, ptr(kps), ptr(desc), len(kps), C.byref(n)), "afv_akaze_get_features")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def quotas(self):
        q = np.zeros(16, np.int32)
        self.check(self.lib.afv_akaze_get_quotas(self.handle, ptr(q)), "afv_akaze_get_quotas")
        return q

    def extract(self, frames):
        """FeatureExtractor_akaze61::detectAndCompute for (H, W) or (B, H, W) uint8 frames -> list of (keypoints, N x 61 descriptors)"""
        from .extractor import KP_DTYPE
        frames = np.ascontiguousarray(frames, np.uint8)
        single = frames.ndim == 2
        if single:
            frames = frames[None]
        b, h, w = frames.shape
        cap = self.params.nfeatures + 3 * 16
        kps = np.zeros((b, cap), KP_DTYPE); desc = np.zeros((b, cap, 61), np.uint8); n = np.zeros(b, np.int32)
        self.check(self.lib.afv_akaze_extract(self.handle, ptr(frames), b, w, h, w, w * h, ptr(kps), ptr(desc), cap, ptr(n)), "afv_akaze_extract")
        self.plan = plan_for(self.params, w, h)
        out = [(kps[f, :n[f]].copy(), desc[f, :n[f]].copy()) for f in range(b)]
        return out[0] if single else out

    def extract_device(self, frames_t):
        b, h, w = frames_t.shape
        self.check(self.lib.afv_akaze_extract_device(self.handle, C.c_void_p(frames_t.data_ptr()), b, w, h, w, w * h), "afv_akaze_extract_device")
        if self.plan is None or (self.plan.w, self.plan.h) != (w, h):
            self.plan = plan_for(self.params, w, h)

    def set_step_by_step(self, on=True):
        self.check(self.lib.afv_akaze_set_step_by_step(self.handle, int(on)), "afv_akaze_set_step_by_step")

    def set_suppress_engine(self, mode=2, pass_cap=0):
        """ordered duplicate suppression: 0 = speculative rounds, 1 = fixed point, 2 = fixed point with fallback (default)"""
        self.check(self.lib.afv_akaze_set_suppress_engine(self.handle, int(mode), int(pass_cap)), "afv_akaze_set_suppress_engine")

    def debug_neighbour_cap(self, cap=0):
        self.check(self.lib.afv_akaze_debug_neighbour_cap(self.handle, int(cap)), "afv_akaze_debug_neighbour_cap")

    def profile_enable(self, on=True):
        self.check(self.lib.afv_akaze_profile_enable(self.handle, int(on)), "afv_akaze_profile_enable")

    def profile_read(self):
        a, b, n = C.c_float(), C.c_float(), C.c_int()
        self.check(self.lib.afv_akaze_profile_read(self.handle, C.byref(a), C.byref(b), C.byref(n)), "afv_akaze_profile_read")
        return {"scale_space_ms": a.value, "hessian_ms": b.value, "launches": n.value}


class FeatureExtractor_akaze61:
    """Mirror of ANYFEATURE_VSLAM::FeatureExtractor_akaze61 (include/Feature_akaze61.h, src/Feature_akaze61.cpp): same
    constructor arguments and method names; detectAndCompute runs initializeExtractor (scale space), detectKeypoints,
    filterKeypoints and computeDescriptors on the GPU in one call."""

    def __init__(self, nfeatures, settings=None, device=0, max_width=1280, max_height=720, max_batch=1):
        from .extractor import FeatureExtractorSettings
        self.settings = settings or FeatureExtractorSettings()
        self.nfeatures = int(nfeatures)
        s = self.settings
        # Feature_akaze61.cpp:11-14: omax = nominal octaves / 4, nsublevels = nominal octaves / 2, dthreshold = detectTh
        self.params = default_params(num_octaves=s.GetDetectorNominalNumOctaves(), detection_th=float(s.detectTh), max_width=max_width,
                                     max_height=max_height, max_batch=max_batch, nfeatures=self.nfeatures,
                                     scale_factor=float(s.GetDetectorNominalScaleFactor()))
        self.ctx = AkazeContext(self.params, device)

    def detectAndCompute(self, gray):
        """-> (keypoints [KP_DTYPE: pt in level-0 pixels, size, angle (radians), response, octave, class_id = level], N x 61 u8)"""
        return self.ctx.extract(gray)

    def __call__(self, gray):
        return self.detectAndCompute(gray)

    @staticmethod
    def GetKeypointOctave(keypoint):
        return int(keypoint["class_id"])              # Feature_akaze61.cpp:55-57

    def GetKeypointSize(self, keypoint):
        # Feature_akaze61.cpp:59-61: powf(GetDetectorNominalScaleFactor(), octave)
        return float(np.float32(self.settings.GetDetectorNominalScaleFactor()) ** np.float32(self.GetKeypointOctave(keypoint)))

    def GetLevels(self):
        return self.settings.nOctaves

    def close(self):
        self.ctx.close()
