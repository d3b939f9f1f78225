#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ (run in the BUILD container only).

Inputs  : bit-reproducible LCG frames (anyfeature-vslam_amd/synth.py) and ONE real frame of the reference's toy
          sequence (/root/reference/docs/toy_sequence/rgb/*.png, a data file; converted to gray with OpenCV's
          cvtColor fixed-point weights Y = (R*4899 + G*9617 + B*1868 + 8192) >> 14 — Image.cpp:30-53 calls cvtColor).
Outputs : expected values produced by the CPU oracle (oracle/, PARITY UNPINNED: the reference itself cannot be built or
          run here and has no tests/golden vectors of its own).  They pin (a) the oracle against silent regressions and
          (b) the HIP path on the GPU box independently of the oracle binary.
"""
import glob
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib

afv_synth = importlib.import_module("anyfeature-vslam_amd.synth")
import oracle

OUT = os.path.dirname(os.path.abspath(__file__))


def toy_gray():
    from PIL import Image
    p = sorted(glob.glob("/root/reference/docs/toy_sequence/rgb/*.png"))[0]
    rgb = np.asarray(Image.open(p).convert("RGB")).astype(np.int64)
    y = (rgb[:, :, 0] * 4899 + rgb[:, :, 1] * 9617 + rgb[:, :, 2] * 1868 + 8192) >> 14
    return np.clip(y, 0, 255).astype(np.uint8), os.path.basename(p)


def record(name, img, store):
    kps, desc, tr = oracle.orb_extract_trace(img)
    store[name + "_kps"] = kps
    store[name + "_desc"] = desc
    store[name + "_level_crc"] = np.array([zlib.crc32(l.tobytes()) for l in tr["level"]], np.uint32)
    store[name + "_blur_crc"] = np.array([zlib.crc32(l.tobytes()) for l in tr["blurred"]], np.uint32)
    c = tr["cand"]
    store[name + "_ncand"] = np.array([(c["level"] == l).sum() for l in range(8)], np.int32)
    store[name + "_nkeep1"] = np.array([tr["keep1"][c["level"] == l].sum() for l in range(8)], np.int32)
    store[name + "_nkeep2"] = np.array([tr["keep2"][c["level"] == l].sum() for l in range(8)], np.int32)
    store[name + "_tcounts"] = np.array(tr["t_counts"], np.int32)
    # FAST candidate set checksum per level: crc over sorted (y, x, score)
    crcs = []
    for l in range(8):
        m = c["level"] == l
        a = np.stack([c["y"][m], c["x"][m], c["fast_score"][m]], 1).astype(np.int32)
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        crcs.append(zlib.crc32(a.tobytes()))
    store[name + "_cand_crc"] = np.array(crcs, np.uint32)
    return kps, desc


def main():
    store = {}
    gray, src = toy_gray()
    np.savez_compressed(os.path.join(OUT, "toy_gray.npz"), gray=gray, source=np.array(src))
    k_toy, d_toy = record("toy", gray, store)
    k1, d1 = record("corners1", afv_synth.corners_frame(1), store)
    k2, d2 = record("corners2", afv_synth.corners_frame(2), store)
    record("noise3", afv_synth.noise_frame(3), store)
    # matcher vectors: frame shifted by 4 px vs itself (real matches), brute force + orientation
    sh = np.roll(afv_synth.corners_frame(1), 4, axis=1)
    ks, ds = oracle.orb_extract(sh)
    m, n = oracle.search_by_bow_kf_kf(ds, d1, angle1=ks["angle"], angle2=k1["angle"], th_low=75.0, nnratio=0.6, check_orientation=True)
    store["shift4_match12"] = m
    store["shift4_nmatches"] = np.array([n], np.int32)
    store["shift4_kps"] = ks
    store["shift4_desc"] = ds
    np.savez_compressed(os.path.join(OUT, "orb32_expected.npz"), **store)
    print("toy: %d kps; corners1: %d; corners2: %d; shift4 matches: %d" % (len(k_toy), len(k1), len(k2), n))
    for f in ("toy_gray.npz", "orb32_expected.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
