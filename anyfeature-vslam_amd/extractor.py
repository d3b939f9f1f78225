"""Host-side mirror of the reference's extractor plugin interface on top of the C-ABI.

Reference:  include/FeatureExtractor.h:68-161 (class FeatureExtractor, FeatureExtractorSettings :23-66),
            include/Feature_orb32.h, src/Feature_orb32.cpp, src/FeatureExtractor.cpp:74-172.
Same names, same argument meaning, same "callee fills the outputs" contract; every number is produced by
libafv_hip.so on the GPU (this file only marshals buffers).
"""
import ctypes as C
import enum

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, AfvError, ptr


class CovarianceMethod(enum.IntEnum):  # FeatureExtractor.h:18-21
    SIZE = 1
    NONE = 0


class FeatureExtractorSettings:
    """FeatureExtractor.h:23-66 / FeatureExtractor.cpp:21-56.  numOctaves0/scaleFactor0/th0 are process-wide statics in
    the reference (a second settings object built with file "none" inherits them, Tracking.cc:84); same here."""
    numOctaves0 = 8
    scaleFactor0 = 1.2
    th0 = 20.0

    def __init__(self, settings=None):
        """settings: None/"none" (inherit the statics) or a dict / YAML path holding FeatureExtractor.numOctaves,
        FeatureExtractor.scaleFactor, FeatureExtractor.detectionTh (settings/orb32_settings.yaml:6-8)."""
        if settings is not None and settings != "none":
            if isinstance(settings, str):
                import yaml
                with open(settings) as fh:
                    text = fh.read()
                if text.startswith("%YAML"):
                    text = text.split("\n", 1)[1]
                settings = yaml.safe_load(text)
            cls = FeatureExtractorSettings
            cls.numOctaves0 = int(settings["FeatureExtractor.numOctaves"])
            cls.scaleFactor0 = float(settings["FeatureExtractor.scaleFactor"])
            cls.th0 = float(settings["FeatureExtractor.detectionTh"])
        self.scaleFactor = self.GetDetectorNominalScaleFactor()
        self.nOctaves = self.GetDetectorNominalNumOctaves()
        self.detectTh = self.GetDetectorNominalThreshold()
        self.ON_automaticTuning = True
        self.iniThFAST = 20
        self.minThFAST = 7

    @classmethod
    def GetDetectorNominalScaleFactor(cls):
        return cls.scaleFactor0

    @classmethod
    def GetDetectorNominalNumOctaves(cls):
        return cls.numOctaves0

    @classmethod
    def GetDetectorNominalThreshold(cls):
        return cls.th0


class Context:
    """RAII wrapper of afv_ctx (one HIP stream + scratch; not re-entrant, one per calling thread)."""

    def __init__(self, nfeatures=1000, nlevels=8, scale_factor=1.2, fast_threshold=20, max_width=640, max_height=480,
                 max_batch=1, device=0):
        self.lib = _lib.load()
        self.params = _lib.OrbParams(nfeatures, nlevels, scale_factor, fast_threshold, max_width, max_height, max_batch)
        h = C.c_void_p()
        rc = self.lib.afv_create(device, C.byref(self.params), C.byref(h))
        if rc != 0:
            raise AfvError(rc, "afv_create")
        self.handle = h
        self.device = device
        self.cap = self.lib.afv_max_keypoints_per_frame(h)

    def close(self):
        if getattr(self, "handle", None):
            self._ext_stream = None  # torch wrapper of the context's stream (_lib.launch_ordered): dies with it
            self.lib.afv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc, what=""):
        if rc != 0:
            raise AfvError(rc, what + " " + self.lib.afv_last_error(self.handle).decode())

    def geometry(self):
        g = _lib.Geometry()
        self.check(self.lib.afv_get_geometry(self.handle, C.byref(g)))
        n = g.nlevels
        return {"nlevels": n, "width": g.width, "height": g.height, "lw": list(g.lw[:n]), "lh": list(g.lh[:n]),
                "lscale": list(g.lscale[:n]), "quota": list(g.quota[:n]), "cv_quota": list(g.cv_quota[:n]),
                "cand_cap": list(g.cand_cap[:n])}

    # ---- extraction ----
    def extract(self, gray, cap=None):
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = cap or self.cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        rc = self.lib.afv_orb_extract(self.handle, ptr(gray), w, h, gray.strides[0], ptr(kps), ptr(desc), cap, C.byref(n))
        self.check(rc, "afv_orb_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def detect(self, gray, cap=None):
        """afv_orb_detect: detectKeypoints + filterKeypoints (Feature_orb32.cpp:26-40, :63-65): the keypoints afv_orb_extract returns, no descriptors"""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        cap = cap or self.cap
        kps = np.zeros(cap, KP_DTYPE)
        n = C.c_int(0)
        self.check(self.lib.afv_orb_detect(self.handle, ptr(gray), w, h, gray.strides[0], ptr(kps), cap, C.byref(n)), "afv_orb_detect")
        return kps[:n.value].copy()

    def compute(self, gray, kps):
        """afv_orb_compute: computeDescriptors = cv::ORB::compute at the given keypoints (Feature_orb32.cpp:42-53)"""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        desc = np.zeros((len(kps), 32), np.uint8)
        self.check(self.lib.afv_orb_compute(self.handle, ptr(gray), w, h, gray.strides[0], ptr(kps), len(kps), ptr(desc)), "afv_orb_compute")
        return desc

    def extract_batch(self, frames, cap=None):
        frames = [np.ascontiguousarray(f, np.uint8) for f in frames]
        h, w = frames[0].shape
        nf = len(frames)
        cap = cap or self.cap
        kps = np.zeros((nf, cap), KP_DTYPE)
        desc = np.zeros((nf, cap, 32), np.uint8)
        n = np.zeros(nf, np.int32)
        arr = (C.c_void_p * nf)(*[f.ctypes.data for f in frames])
        rc = self.lib.afv_orb_extract_batch(self.handle, arr, nf, w, h, frames[0].strides[0], ptr(kps), ptr(desc), cap, ptr(n))
        self.check(rc, "afv_orb_extract_batch")
        return [(kps[i, :n[i]].copy(), desc[i, :n[i]].copy()) for i in range(nf)]

    def extract_batch_host(self, frames, kps, desc, n_out, cap=None):
        """afv_orb_extract_batch on caller-owned HOST buffers without any Python-side copies: frames uint8 [B,H,W] (numpy array or
        CPU torch tensor, rows contiguous), kps float32 [B,cap,7], desc uint8 [B,cap,32], n_out int32 [B].  Page-locked buffers
        (torch pin_memory) are DMA'd in place by the library's chunk pipeline; pageable ones go through its pinned arena."""
        def addr(a):
            return a.data_ptr() if hasattr(a, "data_ptr") else a.ctypes.data
        B, H, W = frames.shape
        cap = cap or kps.shape[1]
        row = frames.stride(1) if hasattr(frames, "stride") else frames.strides[1]
        fst = (frames.stride(0) if hasattr(frames, "stride") else frames.strides[0])
        base = addr(frames)
        arr = (C.c_void_p * B)(*[base + f * fst for f in range(B)])
        rc = self.lib.afv_orb_extract_batch(self.handle, arr, B, W, H, row, C.c_void_p(addr(kps)), C.c_void_p(addr(desc)), cap,
                                            C.c_void_p(addr(n_out)))
        self.check(rc, "afv_orb_extract_batch")

    def extract_batch_device(self, frames, kps=None, desc=None, n_out=None, status=None, cap=None, stream=None):
        """frames: CUDA uint8 tensor [B,H,W] (contiguous rows, 4-byte aligned strides).  Returns device tensors
        (kps float32 [B,cap,7] viewable as afv_keypoint, desc uint8 [B,cap,32], n int32 [B], status int32 [1]).
        Asynchronous on `stream` (default: torch's current stream)."""
        import torch
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3
        B, H, W = frames.shape
        cap = cap or self.cap
        dev = frames.device
        if kps is None:
            kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev)
        if desc is None:
            desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
        if n_out is None:
            n_out = torch.empty((B,), dtype=torch.int32, device=dev)
        if status is None:
            status = torch.empty((1,), dtype=torch.int32, device=dev)
        rc = _lib.launch_ordered(self, dev, stream, lambda s: self.lib.afv_orb_extract_batch_device(
            self.handle, frames.data_ptr(), B, W, H, frames.stride(1), frames.stride(0), kps.data_ptr(), desc.data_ptr(), cap,
            n_out.data_ptr(), status.data_ptr(), s), (frames, kps, desc, n_out, status))
        self.check(rc, "afv_orb_extract_batch_device")
        return kps, desc, n_out, status

    def size_sigma(self, kps):
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        n = len(kps)
        size = np.zeros(n, np.float32); s2 = np.zeros(n, np.float32); inf = np.zeros(n, np.float32)
        self.check(self.lib.afv_orb_size_sigma(self.handle, ptr(kps), n, ptr(size), ptr(s2), ptr(inf)))
        return size, s2, inf

    # ---- live stage timing (hipEvents on the launch stream) ----
    def profile_enable(self, enable=True, every=1):
        """every = n: only every n-th extraction / pair-match call records its stage events"""
        self.check(self.lib.afv_profile_enable(self.handle, int(every) if enable else 0))

    def profile_read(self):
        ns = max(int(self.lib.afv_num_stages()), len(_lib.STAGES))  # the library writes afv_num_stages() entries
        launches = np.zeros(ns, np.int32); ms = np.zeros(ns, np.float32)
        units = np.zeros(ns, np.int64)
        self.check(self.lib.afv_profile_read(self.handle, ptr(launches), ptr(ms), ptr(units)))
        return {name: {"launches": int(launches[i]), "total_ms": float(ms[i]), "units": int(units[i])}
                for i, name in enumerate(_lib.STAGES)}

    def set_split_threshold(self, min_frames):
        self.check(self.lib.afv_set_split_threshold(self.handle, int(min_frames)))

    def set_pipeline_chunk(self, frames, chunks_ahead=8):
        self.check(self.lib.afv_set_pipeline_chunk(self.handle, int(frames), int(chunks_ahead)))

    def set_match_engine(self, engine):
        """0 = popcount on the vector ALU, 1 = exact i8 contraction on the matrix cores (default); identical results"""
        self.check(self.lib.afv_set_match_engine(self.handle, int(engine)))

    def set_small_batch_path(self, mode, max_frames=0):
        """0 = never, 1 = calls of at most max_frames frames / pairs (default), 2 = always; identical results either way"""
        self.check(self.lib.afv_set_small_batch_path(self.handle, int(mode), int(max_frames)))

    def set_l2_chunk_pairs(self, pairs):
        self.check(self.lib.afv_set_l2_chunk_pairs(self.handle, int(pairs)))

    def set_match_resolve(self, engine):
        """phase 2 of the pair matchers: 1 = workgroup-wide fixed point, 0 = ordered walk on one wavefront, 2 = by call size (default); identical results"""
        self.check(self.lib.afv_set_match_resolve(self.handle, int(engine)))

    def set_split_chunks(self, chunks):
        self.check(self.lib.afv_set_split_chunks(self.handle, int(chunks)))

    @property
    def stream(self):
        """the context's own hipStream_t (what a NULL `stream` argument selects)"""
        return self.lib.afv_stream(self.handle)

    # ---- stage introspection (parity tests) ----
    def debug_level(self, frame, level):
        g = self.geometry()
        out = np.zeros((g["lh"][level], g["lw"][level]), np.uint8)
        self.check(self.lib.afv_debug_get_level(self.handle, frame, level, ptr(out)))
        return out

    def debug_blur_level(self, frame, level):
        g = self.geometry()
        out = np.zeros((g["lh"][level], g["lw"][level]), np.uint8)
        self.check(self.lib.afv_debug_blur_level(self.handle, frame, level, ptr(out)))
        return out

    def debug_candidates(self, frame, level):
        g = self.geometry()
        cap = g["cand_cap"][level]
        packed = np.zeros(cap, np.uint32); resp = np.zeros(cap, np.float32)
        n = C.c_int(0)
        self.check(self.lib.afv_debug_get_candidates(self.handle, frame, level, ptr(packed), ptr(resp), cap, C.byref(n)))
        packed = packed[:n.value]
        return (packed & 4095).astype(np.int32), ((packed >> 12) & 4095).astype(np.int32), (packed >> 24).astype(np.int32), \
            resp[:n.value].copy()

    def debug_selected(self, frame, level):
        g = self.geometry()
        cap = g["quota"][level] + 8
        x = np.zeros(cap, np.int32); y = np.zeros(cap, np.int32); r = np.zeros(cap, np.float32)
        n = C.c_int(0)
        self.check(self.lib.afv_debug_get_selected(self.handle, frame, level, ptr(x), ptr(y), ptr(r), cap, C.byref(n)))
        return x[:n.value].copy(), y[:n.value].copy(), r[:n.value].copy()


class FeatureExtractor_orb32:
    """Mirror of FeatureExtractor_orb32 (Feature_orb32.h:12-33) + the FeatureExtractor base (FeatureExtractor.h:68-161).

    ext = FeatureExtractor_orb32(nfeatures, settings)
    kps, desc, sigma2, inf, size = ext(gray)            # 6-arg operator() (FeatureExtractor.cpp:111-121)
    kps, desc = ext.detectAndCompute(gray)              # Feature_orb32.cpp:11-18
    """

    def __init__(self, nfeatures, settings=None, device=0, max_width=640, max_height=480, max_batch=1):
        self.settings = settings or FeatureExtractorSettings()
        self.nfeatures = int(nfeatures)
        s = self.settings
        self.ctx = Context(self.nfeatures, s.nOctaves, s.scaleFactor, int(s.detectTh), max_width, max_height, max_batch, device)
        # FeatureExtractor.cpp:77-95
        self.mvScaleFactor = [1.0]
        for _ in range(1, s.nOctaves):
            self.mvScaleFactor.append(float(np.float32(self.mvScaleFactor[-1]) * np.float32(s.scaleFactor)))
        self.mnFeaturesPerLevel = self.ctx.geometry()["quota"]

    # FeatureExtractor.h:95-99
    def GetLevels(self):
        return self.settings.nOctaves

    def GetScaleFactor(self):
        return self.settings.scaleFactor

    def GetScaleFactors(self):
        return list(self.mvScaleFactor)

    def GetKeypointOctave(self, kp):  # Feature_orb32.cpp:55-57
        return int(kp["octave"])

    def initializeExtractor(self, img):  # Feature_orb32.cpp:20-24 (cv::ORB::create; nothing to do on the GPU path)
        return None

    def automaticTuning(self, img):  # FeatureExtractor.cpp:195-274: detectTh = th0, then disables itself
        self.settings.detectTh = self.settings.GetDetectorNominalThreshold()
        self.settings.ON_automaticTuning = False

    # the three virtuals detectAndCompute is made of (FeatureExtractor.h:123-128): a host may call them one by one
    def detectKeypoints(self, gray):
        """-> {level: keypoints}; the quadtree (filterKeypoints) already ran on the device, see filterKeypoints"""
        kps = self.ctx.detect(gray)
        return {int(l): kps[kps["octave"] == l] for l in np.unique(kps["octave"])}

    def filterKeypoints(self, keypoints_level, gray=None, mask=None):
        """DistributeOctTree per level (Feature_orb32.cpp:63-65): done inside afv_orb_detect - the device never hands the 10x candidate set
        of cv::ORB::detect to the host - so this is the identity on detectKeypoints' output"""
        return keypoints_level

    def computeDescriptors(self, keypoints_level, gray):
        """-> {level: descriptors} (one cv::ORB::compute per level in the reference, Feature_orb32.cpp:49-50; one call here)"""
        levels = sorted(keypoints_level)
        allk = np.concatenate([keypoints_level[l] for l in levels]) if levels else np.zeros(0, KP_DTYPE)
        desc = self.ctx.compute(gray, allk)
        out, o = {}, 0
        for l in levels:
            out[l] = desc[o:o + len(keypoints_level[l])]
            o += len(keypoints_level[l])
        return out

    def detectAndCompute(self, gray):
        if gray is None or gray.size == 0:  # ORBextractor.cc:570-571: empty image -> outputs untouched
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        return self.ctx.extract(gray)

    def __call__(self, gray, with_covariance=True):
        self.initializeExtractor(gray)
        if self.settings.ON_automaticTuning:
            self.automaticTuning(gray)
        kps, desc = self.detectAndCompute(gray)
        if not with_covariance:  # 3-arg operator() (FeatureExtractor.cpp:123-129)
            return kps, desc
        size, s2, inf = self.ctx.size_sigma(kps)  # computeSize + computeSigma(SIZE)
        eye = np.eye(2, dtype=np.float32)
        return kps, desc, s2[:, None, None] * eye, inf[:, None, None] * eye, size
