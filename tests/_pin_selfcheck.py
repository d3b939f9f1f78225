"""Plumbing check of tests/test_opencv_pin.py WITHOUT OpenCV: writes tests/golden/opencv_selfcheck.npz in the format of
tools/pin_against_opencv.py but filled from the oracle, so the comparison code paths (CPU and -m gpu) can be exercised; the file is
a stand-in, pins nothing and must be deleted again:   python tests/_pin_selfcheck.py make | clean"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PATH = os.path.join(ROOT, "tests", "golden", "opencv_selfcheck.npz")
KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def make():
    import oracle
    oracle.lib()
    synth = importlib.import_module("anyfeature-vslam_amd.synth")
    gray = synth.corners_frame(1)
    _, _, tr = oracle.orb_extract_trace(gray)
    d = {"gray": gray, "cv_version": np.array("stand-in (oracle)")}
    nl = len(tr["level"])
    for l in range(nl):
        d["level_%d" % l] = tr["level"][l]
        d["blur_%d" % l] = tr["blurred"][l]
        xs, ys, sc = oracle.fast9_16(tr["level"][l], 20)
        d["fast_%d" % l] = np.stack([xs, ys, sc], 1).astype(np.float32)
    cand = tr["cand"][tr["keep2"]]
    sc = tr["lscale"]
    arr = np.zeros(len(cand), KP)
    for i, c in enumerate(cand):
        l = c["level"]
        arr[i] = (np.float32(c["x"]) * np.float32(sc[l]), np.float32(c["y"]) * np.float32(sc[l]), 31 * sc[l],
                  oracle.ic_angle(tr["level"][l], c["x"], c["y"]), c["response"], l, -1)
    d["detect"] = arr
    for l in range(nl):
        kl = arr[arr["octave"] == l]
        inv = np.float32(1) / np.float32(sc[l])
        d["compute_kps_%d" % l] = kl
        d["compute_desc_%d" % l] = np.stack([oracle.brief_descriptor(tr["level"][l], tr["blurred"][l], int(np.rint(q["x"] * inv)),
                                                                      int(np.rint(q["y"] * inv)), float(q["angle"])) for q in kl])
    np.savez_compressed(PATH, **d)


if __name__ == "__main__":
    if sys.argv[1:] == ["make"]:
        make()
    elif os.path.exists(PATH):
        os.remove(PATH)
