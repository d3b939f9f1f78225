#!/usr/bin/env python3
"""Pinning kit for SURVEY.md 8c: dump what REAL OpenCV computes for the stages this repository restates from the published
algorithm (E2 pyramid, E3 FAST + NMS, E4/E5 retainBest + Harris, E6 IC angle, E9 blur, E10 rBRIEF; AKAZE as a softer check).

Needs only `cv2` (>= 4.5) and numpy — neither the GPU nor this package's library.  It can NOT run in the build image or on the
GPU box (no OpenCV there); run it on any machine that has OpenCV, from the repository root:

    python tools/pin_against_opencv.py            # writes tests/golden/opencv_<frame>.npz  (+ opencv_pyramids_<frame>.npz: the pyramid at
                                                  #   scale factors 1.1892 / 1.5 / 2.0 / 1.3, + opencv_akaze_<frame>.npz)

and commit the files.  tests/test_opencv_pin.py then activates: the CPU test compares oracle/ with them stage by stage and
names the FIRST stage that diverges; the `-m gpu` test does the same for the HIP path.  Until somebody runs this, parity of
those stages stays "unpinned" (DESIGN.md section 2).

OpenCV is driven exactly as the reference drives it (src/Feature_orb32.cpp:20-53):
    orb = cv::ORB::create(); setMaxFeatures(nfeatures * 10); setEdgeThreshold(0); setFastThreshold(int(20)); setNLevels(8)
    detect(gray) once, then compute(gray, keypoints_of_level_l) once per level.
The intermediate stages cv::ORB does not expose are reproduced with the public functions cv::ORB itself calls:
resize(INTER_LINEAR_EXACT) level by level, FastFeatureDetector(20, true, TYPE_9_16) per level, GaussianBlur(7x7, 2, 2,
BORDER_REFLECT_101) per level."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("AFV_PIN_OUT") or os.path.join(ROOT, "tests", "golden")   # where the files go (the override is for the self-test)
NFEATURES, NLEVELS, SCALE, FAST_TH = 1000, 8, 1.2, 20

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


# ---- the synthetic inputs of SURVEY.md 8d, restated here so the script has no dependency on the package ----
def lcg_states(seed, n):
    a = np.full(n, 1664525, dtype=np.uint32)
    with np.errstate(over="ignore"):
        A = np.multiply.accumulate(a, dtype=np.uint32)
        geo = np.concatenate([np.ones(1, np.uint32), A[:-1]])
        Cs = (np.add.accumulate(geo, dtype=np.uint32) * np.uint32(1013904223)).astype(np.uint32)
        return A * np.uint32(seed & 0xFFFFFFFF) + Cs


def corners_frame(seed, w=640, h=480, block=8):
    bw, bh = (w + block - 1) // block, (h + block - 1) // block
    st = lcg_states(seed, bw * bh + w * h)
    tiles = ((st[:bw * bh] >> np.uint32(8)) & np.uint32(255)).astype(np.int32).reshape(bh, bw)
    img = np.repeat(np.repeat(tiles, block, axis=0), block, axis=1)[:h, :w]
    p = np.pad(img, 1, mode="edge")
    s = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    img = (s + 4) // 9
    noise = ((st[bw * bh:] >> np.uint32(8)) % np.uint32(9)).astype(np.int32).reshape(h, w) - 4
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def noise_frame(seed, w=640, h=480):
    return ((lcg_states(seed, w * h) >> np.uint32(8)) & np.uint32(255)).astype(np.uint8).reshape(h, w)


def frames():
    out = {"corners1": corners_frame(1), "corners2": corners_frame(2), "noise3": noise_frame(3)}
    seq = os.path.join(OUT, "toy_seq_gray.npz")  # the five frames of the reference's docs/toy_sequence, committed as data
    toy = os.path.join(OUT, "toy_gray.npz")     # (its first frame)
    if os.path.exists(seq):
        for i, g in enumerate(np.load(seq)["gray"]):
            out["toy%d" % i] = g
    elif os.path.exists(toy):
        out["toy"] = np.load(toy)["gray"]
    return out


def kp_array(kps):
    a = np.zeros(len(kps), KP_DTYPE)
    for i, k in enumerate(kps):
        a[i] = (k.pt[0], k.pt[1], k.size, k.angle, k.response, k.octave, k.class_id)
    return a


def level_sizes(w, h):
    out = []
    for l in range(NLEVELS):
        s = np.float32(np.power(np.float64(np.float32(SCALE)), l))   # (float)pow(scaleFactor, level), scaleFactor held as double in cv::ORB
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def pin_orb(cv2, name, gray):
    d = {"gray": gray, "cv_version": np.array(cv2.__version__), "build": np.array(cv2.getBuildInformation()[:4000])}
    h, w = gray.shape
    # E2: level l = INTER_LINEAR_EXACT resize of level l-1 (orb.cpp builds the pyramid level from the previous level)
    levels = [gray]
    for (lw, lh) in level_sizes(w, h)[1:]:
        levels.append(cv2.resize(levels[-1], (lw, lh), interpolation=cv2.INTER_LINEAR_EXACT))
    fast = cv2.FastFeatureDetector_create(threshold=FAST_TH, nonmaxSuppression=True, type=cv2.FastFeatureDetector_TYPE_9_16)
    for l, im in enumerate(levels):
        d["level_%d" % l] = im
        d["level_crc_%d" % l] = np.array(zlib.crc32(im.tobytes()), np.uint32)
        # E9: what cv::ORB::compute applies to each level before sampling (the level ROI is blurred in place)
        # (sigmaY by keyword: the fourth POSITIONAL parameter of the Python binding is `dst`)
        d["blur_%d" % l] = cv2.GaussianBlur(im, (7, 7), sigmaX=2, sigmaY=2, borderType=cv2.BORDER_REFLECT_101)
        # E3: FAST-9/16 + NMS on the un-bordered level
        kf = fast.detect(im, None)
        d["fast_%d" % l] = np.array([(k.pt[0], k.pt[1], k.response) for k in kf], np.float32).reshape(-1, 3)
    # E1 + E4..E6: the detect call of Feature_orb32.cpp:34
    orb = cv2.ORB_create()
    orb.setMaxFeatures(NFEATURES * 10)
    orb.setEdgeThreshold(0)
    orb.setFastThreshold(int(FAST_TH))
    orb.setNLevels(NLEVELS)
    kps = orb.detect(gray, None)
    d["detect"] = kp_array(kps)
    # E8/E10: one compute call per level (Feature_orb32.cpp:42-53)
    by_level = {}
    for k in kps:
        by_level.setdefault(k.octave, []).append(k)
    for l, kl in sorted(by_level.items()):
        kout, desc = orb.compute(gray, kl)
        d["compute_kps_%d" % l] = kp_array(kout)
        d["compute_desc_%d" % l] = desc if desc is not None else np.zeros((0, 32), np.uint8)
    np.savez_compressed(os.path.join(OUT, "opencv_%s.npz" % name), **d)
    print("opencv_%s.npz: %d keypoints from detect, levels %s" % (name, len(kps), [im.shape for im in levels]))


# scale factors / level counts the parity tests started to rely on in round 4 (settings/*.yaml:6-7: 1.2 x 8 for ORB32, 1.1892 x 8 for the
# AKAZE61 quotas; 1.5 / 2.0 / 1.3 exercise the other branches of the INTER_LINEAR_EXACT coefficient rule: exact halves, long tap distances)
PYRAMIDS = ((1.1892, 8), (1.5, 5), (2.0, 4), (1.3, 6))


def level_sizes_for(w, h, scale, nlevels):
    out = []
    for l in range(nlevels):
        s = np.float32(np.power(np.float64(np.float32(scale)), l))
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def pin_pyramids(cv2, name, gray):
    """E2 at other scale factors: the pyramid cv::ORB would build (level l = INTER_LINEAR_EXACT resize of level l - 1)"""
    d = {"gray": gray, "cv_version": np.array(cv2.__version__)}
    h, w = gray.shape
    for k, (scale, nlevels) in enumerate(PYRAMIDS):
        d["pyr%d_scale" % k] = np.float32(scale)
        d["pyr%d_nlevels" % k] = np.int32(nlevels)
        im = gray
        for l, (lw, lh) in enumerate(level_sizes_for(w, h, scale, nlevels)):
            if l:
                im = cv2.resize(im, (lw, lh), interpolation=cv2.INTER_LINEAR_EXACT)
            d["pyr%d_level_%d" % (k, l)] = im
    np.savez_compressed(os.path.join(OUT, "opencv_pyramids_%s.npz" % name), **d)
    print("opencv_pyramids_%s.npz: %s" % (name, [(s, n) for s, n in PYRAMIDS]))


def pin_akaze(cv2, name, gray):
    """cv::AKAZE is OpenCV's port of libAKAZE, not the fork the reference links: a softer check (counts, overlap of positions)."""
    ak = cv2.AKAZE_create(descriptor_type=cv2.AKAZE_DESCRIPTOR_MLDB, descriptor_size=0, descriptor_channels=3, threshold=0.0005,
                          nOctaves=2, nOctaveLayers=4, diffusivity=cv2.KAZE_DIFF_PM_G2)
    kps, desc = ak.detectAndCompute(gray, None)
    np.savez_compressed(os.path.join(OUT, "opencv_akaze_%s.npz" % name), gray=gray, kps=kp_array(kps),
                        desc=desc if desc is not None else np.zeros((0, 61), np.uint8), cv_version=np.array(cv2.__version__))
    print("opencv_akaze_%s.npz: %d keypoints" % (name, len(kps)))


def main():
    try:
        import cv2
    except ImportError:
        sys.exit("OpenCV (cv2) is not installed here: run this script on a machine that has it (see the docstring)")
    os.makedirs(OUT, exist_ok=True)
    for name, gray in frames().items():
        pin_orb(cv2, name, np.ascontiguousarray(gray))
    pin_pyramids(cv2, "corners1", corners_frame(1))
    pin_pyramids(cv2, "noise3", noise_frame(3))
    pin_akaze(cv2, "corners1_720p", corners_frame(1, 1280, 720))


if __name__ == "__main__":
    main()
