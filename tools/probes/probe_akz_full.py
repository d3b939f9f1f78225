"""measurement: per-kernel times of the whole AKAZE step with a variant library: python probe_akz_full.py LIB [batch]"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] != "base":
    importlib.import_module("anyfeature-vslam_amd._lib").use_library(sys.argv[1])
afv = importlib.import_module("anyfeature-vslam_amd")
import bench
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
r = bench.akaze_step(afv, 0, B, 5) if hasattr(bench, "akaze_step") else None
print(sys.argv[1], r and {k: r[k] for k in ("ms_per_step", "frames_per_s", "scale_space_ms_per_step") if k in r})
