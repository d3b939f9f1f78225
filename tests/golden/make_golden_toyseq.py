#!/usr/bin/env python3
"""Generate tests/golden/toy_seq_gray.npz and toy_seq_expected.npz (run in the BUILD container only): all five frames of the
reference's toy sequence (/root/reference/docs/toy_sequence/rgb/*.png — data files, the input of BASELINE.json configs[0]),
converted to gray with OpenCV's cvtColor fixed-point weights (Image.cpp:30-53 calls cvtColor), and what the CPU oracle
produces for them: keypoints and descriptors per frame, and the brute-force SearchByBoW(KF,KF) matches of every frame against its
predecessor (TH_LOW 75, nnratio 0.6, orientation check: the pipeline step of bench.py).  PARITY UNPINNED, like every oracle
fixture here: the files pin the oracle against regressions and the HIP path independently of the oracle binary."""
import glob
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    from PIL import Image
    paths = sorted(glob.glob("/root/reference/docs/toy_sequence/rgb/*.png"))
    grays = []
    for p in paths:
        rgb = np.asarray(Image.open(p).convert("RGB")).astype(np.int64)
        y = (rgb[:, :, 0] * 4899 + rgb[:, :, 1] * 9617 + rgb[:, :, 2] * 1868 + 8192) >> 14
        grays.append(np.clip(y, 0, 255).astype(np.uint8))
    gray = np.stack(grays)
    np.savez_compressed(os.path.join(OUT, "toy_seq_gray.npz"), gray=gray, source=np.array([os.path.basename(p) for p in paths]))
    store = {}
    prev = None
    for i, g in enumerate(grays):
        kps, desc = oracle.orb_extract(g)
        store["kps_%d" % i] = kps
        store["desc_%d" % i] = desc
        if prev is not None:
            m, n = oracle.search_by_bow_kf_kf(desc, prev[1], angle1=kps["angle"], angle2=prev[0]["angle"], th_low=75.0, nnratio=0.6,
                                              check_orientation=True)
            store["match_%d" % i] = m
            store["nmatch_%d" % i] = np.array([n], np.int32)
        prev = (kps, desc)
    np.savez_compressed(os.path.join(OUT, "toy_seq_expected.npz"), **store)
    print("frames", len(grays), "keypoints", [len(store["kps_%d" % i]) for i in range(len(grays))], "matches",
          [int(store["nmatch_%d" % i][0]) for i in range(1, len(grays))])
    for f in ("toy_seq_gray.npz", "toy_seq_expected.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
