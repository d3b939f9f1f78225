#!/usr/bin/env python3
"""Generate tests/golden/float_matchers_expected.npz: the float-descriptor paths of the matchers (round 6) on the frames of the committed ORB32
fixture (orb32_expected.npz: keypoints and descriptors of corners_frame(1) and of the same frame shifted by 4 px).

Inputs  : the fixture's keypoints; float rows derived from its 32-byte descriptors by tests/_float_desc.py (first 128 bits as 0 / 0.75 plus a
          deterministic fraction per element - integer arithmetic and float32 products only).
Outputs : what the CPU oracle (oracle/, PARITY UNPINNED) answers for SearchByBoW(KF, KF), SearchByProjection(cur, last), SearchForInitialization and
          ComputeDistinctiveDescriptors with DescriptorDistance = L2^2 (Feature_sift128.cpp:132-134).  They pin the oracle against silent
          regressions and the HIP path on the GPU box independently of the oracle binary.  Run in the build container:
              python tests/golden/make_golden_float.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
afv = importlib.import_module("anyfeature-vslam_amd")
from _float_desc import floaten  # noqa: E402   (the oracle is imported by main() only: the tests reuse scene() / views() without it)

TH, DIM = 20.0, 128


def scene(gold):
    """the four cases as (inputs dict): everything a test needs beside the fixture's keypoints / descriptors"""
    k1, ks = gold["corners1_kps"], gold["shift4_kps"]
    f1, fs = floaten(gold["corners1_desc"], DIM, True), floaten(gold["shift4_desc"], DIM, True)
    tab = np.ones(8, np.float32)
    for i in range(1, 8):
        tab[i] = tab[i - 1] * np.float32(1.2)          # keyPtsSize per octave by float32 products (no libm)
    return k1, ks, f1, fs, tab[k1["octave"]], tab[ks["octave"]]


def views(k1, ks, f1, fs, z1, zs):
    F = afv.FrameGridView(f1, np.stack([k1["x"], k1["y"]], 1), z1, angles=k1["angle"])
    Q = afv.ProjectionQueries(fs, ks["x"] - np.float32(4), ks["y"], np.float32(15) * zs, zs / np.float32(1.2), zs * np.float32(1.2), angles=ks["angle"])
    n = len(ks)
    Qi = afv.ProjectionQueries(fs, ks["x"].copy(), ks["y"].copy(), np.full(n, 50.0, np.float32), np.zeros(n, np.float32),
                               np.full(n, z1.max(), np.float32), valid=(ks["octave"] == 0).astype(np.uint8), angles=ks["angle"])
    sets = [f1[7 * i:7 * i + 7 + (i % 3)] for i in range(12)]
    return F, Q, Qi, sets


def main():
    import oracle
    gold = np.load(os.path.join(HERE, "orb32_expected.npz"))
    k1, ks, f1, fs, z1, zs = scene(gold)
    F, Q, Qi, sets = views(k1, ks, f1, fs, z1, zs)
    store = {}
    m, n = oracle.search_by_bow_kf_kf(fs, f1, angle1=ks["angle"], angle2=k1["angle"], th_low=TH, nnratio=0.8, check_orientation=True)
    store["bow_match12"], store["bow_n"] = m, np.array([n], np.int32)
    a, n2 = oracle.match_projection(F, Q, th_high=TH, nnratio=0.9, check_orientation=True, last_frame=True)
    store["proj_assign"], store["proj_n"] = a, np.array([n2], np.int32)
    i12, n3 = oracle.match_initialization(F, Qi, th_low=TH, nnratio=0.9, check_orientation=True)
    store["init_match12"], store["init_n"] = i12, np.array([n3], np.int32)
    best = [oracle.distinctive_descriptor(s) for s in sets]
    store["dist_best"] = np.array([b[0] for b in best], np.int32)
    store["dist_median"] = np.array([b[1] for b in best], np.float32)
    store["row_crc"] = np.array([np.frombuffer(f1.tobytes(), np.uint8).astype(np.uint64).sum(), np.frombuffer(fs.tobytes(), np.uint8).astype(np.uint64).sum()], np.uint64)
    np.savez_compressed(os.path.join(HERE, "float_matchers_expected.npz"), **store)
    print("bow %d, projection %d, initialization %d matches; %d map points" % (n, n2, n3, len(sets)))
    print(os.path.getsize(os.path.join(HERE, "float_matchers_expected.npz")), "bytes")


if __name__ == "__main__":
    main()
