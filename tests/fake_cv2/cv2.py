"""A stand-in for the `cv2` module, for ONE purpose: tests/test_opencv_pin.py drives tools/pin_against_opencv.py through it to prove
that the pinning kit calls OpenCV's Python API with the argument order and keywords the real module has, in the sequence the reference
uses (src/Feature_orb32.cpp:20-53), and that what it writes is what the consumer tests read.  Signatures below follow the OpenCV 4.x
Python bindings (positional order included: GaussianBlur takes `dst` BEFORE sigmaY); every function rejects arguments the real one
would reject.  The numbers come from the CPU oracle (test infrastructure), so the produced file is a stand-in and pins nothing.
Every call is appended to LOG."""
import numpy as np

__version__ = "4.x-fake"
INTER_LINEAR_EXACT = 5
BORDER_REFLECT_101 = 4
FastFeatureDetector_TYPE_9_16 = 2
FAST_FEATURE_DETECTOR_TYPE_9_16 = 2
AKAZE_DESCRIPTOR_MLDB = 5
KAZE_DIFF_PM_G2 = 1
LOG = []
_oracle = None


def _o():
    global _oracle
    if _oracle is None:
        import oracle
        oracle.lib()
        _oracle = oracle
    return _oracle


def _img(a, what):
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.ndim == 2 and a.flags["C_CONTIGUOUS"]):
        raise TypeError("%s: expected a contiguous 2-D uint8 array" % what)
    return a


def getBuildInformation():
    return "fake cv2 (tests/fake_cv2)"


class KeyPoint:
    def __init__(self, x, y, size, angle=-1.0, response=0.0, octave=0, class_id=-1):
        self.pt, self.size, self.angle, self.response, self.octave, self.class_id = (float(x), float(y)), float(size), float(angle), float(response), int(octave), int(class_id)


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=1):
    LOG.append(("resize", tuple(dsize), interpolation))
    if dst is not None or fx or fy or interpolation != INTER_LINEAR_EXACT:
        raise ValueError("the pinning kit must resize with INTER_LINEAR_EXACT to an explicit dsize")
    w, h = dsize
    return _o().resize_linear_exact(_img(src, "resize"), int(w), int(h))


def GaussianBlur(src, ksize, sigmaX, dst=None, sigmaY=0, borderType=BORDER_REFLECT_101):
    LOG.append(("GaussianBlur", tuple(ksize), sigmaX, sigmaY, borderType))
    if dst is not None:
        raise TypeError("GaussianBlur: Expected Ptr<cv::UMat> for argument 'dst'")  # what the real module answers to a number here
    if tuple(ksize) != (7, 7) or sigmaX != 2 or sigmaY != 2 or borderType != BORDER_REFLECT_101:
        raise ValueError("cv::ORB blurs with GaussianBlur(7x7, 2, 2, BORDER_REFLECT_101)")
    return _o().gaussian_blur7(_img(src, "GaussianBlur"))


class _Fast:
    def __init__(self, threshold, nms, type_):
        self.threshold, self.nms, self.type = threshold, nms, type_

    def detect(self, image, mask=None):
        LOG.append(("FastFeatureDetector.detect", image.shape))
        if mask is not None:
            raise ValueError("no mask")
        xs, ys, sc = _o().fast9_16(_img(image, "detect"), self.threshold)
        return tuple(KeyPoint(x, y, 7.0, -1.0, s) for x, y, s in zip(xs, ys, sc))


def FastFeatureDetector_create(threshold=10, nonmaxSuppression=True, type=FastFeatureDetector_TYPE_9_16):
    LOG.append(("FastFeatureDetector_create", threshold, nonmaxSuppression, type))
    if not nonmaxSuppression or type != FastFeatureDetector_TYPE_9_16:
        raise ValueError("cv::ORB runs FAST 9/16 with non-maximum suppression")
    return _Fast(int(threshold), nonmaxSuppression, type)


class _Orb:
    """cv::ORB as configured by Feature_orb32.cpp:20-24; detect / compute answered from the oracle's trace of the same frame"""

    def __init__(self):
        self.max_features, self.edge, self.fast, self.nlevels = 500, 31, 20, 8
        self._trace = None

    def setMaxFeatures(self, n):
        LOG.append(("ORB.setMaxFeatures", n))
        self.max_features = int(n)

    def setEdgeThreshold(self, n):
        LOG.append(("ORB.setEdgeThreshold", n))
        self.edge = int(n)

    def setFastThreshold(self, n):
        LOG.append(("ORB.setFastThreshold", n))
        if not isinstance(n, int):
            raise TypeError("setFastThreshold takes an int")
        self.fast = n

    def setNLevels(self, n):
        LOG.append(("ORB.setNLevels", n))
        self.nlevels = int(n)

    def detect(self, image, mask=None):
        LOG.append(("ORB.detect", image.shape, mask))
        if mask is not None:
            raise ValueError("no mask")
        o = _o()
        if self.max_features % 10 or self.edge != 0:
            raise ValueError("Feature_orb32.cpp:22-23: setMaxFeatures(nfeatures * 10), setEdgeThreshold(0)")
        _, _, tr = o.orb_extract_trace(_img(image, "detect"), o.default_params(self.max_features // 10, self.nlevels, 1.2, self.fast))
        self._trace = tr
        cand = tr["cand"][tr["keep2"]]
        sc = tr["lscale"]
        out = []
        for c in cand:
            l = int(c["level"])
            out.append(KeyPoint(np.float32(c["x"]) * np.float32(sc[l]), np.float32(c["y"]) * np.float32(sc[l]), 31 * sc[l],
                                o.ic_angle(tr["level"][l], c["x"], c["y"]), c["response"], l, -1))
        return tuple(out)

    def compute(self, image, keypoints, descriptors=None):
        LOG.append(("ORB.compute", image.shape, sorted({k.octave for k in keypoints}), len(keypoints)))
        if descriptors is not None or self._trace is None:
            raise ValueError("compute(image, keypoints) after detect")
        o, tr = _o(), self._trace
        desc = []
        for k in keypoints:
            l = k.octave
            inv = np.float32(1) / np.float32(tr["lscale"][l])
            desc.append(o.brief_descriptor(tr["level"][l], tr["blurred"][l], int(np.rint(np.float32(k.pt[0]) * inv)), int(np.rint(np.float32(k.pt[1]) * inv)),
                                           float(np.float32(k.angle))))
        return tuple(keypoints), (np.stack(desc) if desc else None)


def ORB_create(nfeatures=500, scaleFactor=1.2, nlevels=8, edgeThreshold=31, firstLevel=0, WTA_K=2, scoreType=0, patchSize=31, fastThreshold=20):
    LOG.append(("ORB_create", nfeatures, scaleFactor, nlevels, edgeThreshold, firstLevel, WTA_K, scoreType, patchSize, fastThreshold))
    return _Orb()


class _Akaze:
    def detectAndCompute(self, image, mask, descriptors=None, useProvidedKeypoints=False):
        LOG.append(("AKAZE.detectAndCompute", image.shape))
        return (), None


def AKAZE_create(descriptor_type=AKAZE_DESCRIPTOR_MLDB, descriptor_size=0, descriptor_channels=3, threshold=0.001, nOctaves=4, nOctaveLayers=4,
                 diffusivity=KAZE_DIFF_PM_G2):
    LOG.append(("AKAZE_create", descriptor_type, descriptor_size, descriptor_channels, threshold, nOctaves, nOctaveLayers, diffusivity))
    return _Akaze()
