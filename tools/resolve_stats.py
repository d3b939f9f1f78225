#!/usr/bin/env python3
"""k_match_resolve on overlapping frames: build the library with -DAFV_RESOLVE_STATS=1 (walk: rounds / passes / rescans / section times) or =2
(whole workgroup: set-up, walk, histogram; no printf inside the walk) — python tools/experiments.py build rs=AFV_RESOLVE_STATS=1 —
and run  python tools/resolve_stats.py anyfeature-vslam_amd/build_exp/libafv_rs.so  on the GPU box: the kernel prints rounds / passes /
rescans / walk time of pairs 1 and 2 (pair 1: frame 1 = frame 0 shifted by 3 px, pair 2: unrelated frames)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    importlib.import_module("anyfeature-vslam_amd._lib").use_library(sys.argv[1])
import bench  # noqa: E402

afv = importlib.import_module("anyfeature-vslam_amd")
print(bench.overlap_step(afv, 0, B=64, steps=1))
