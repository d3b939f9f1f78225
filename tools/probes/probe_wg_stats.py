import importlib, os, sys, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
importlib.import_module("anyfeature-vslam_amd._lib").use_library(os.path.join(ROOT, "anyfeature-vslam_amd/build_exp/libafv_rs4.so"))
import bench
afv = importlib.import_module("anyfeature-vslam_amd")
print(bench.overlap_step(afv, 0, B=256, steps=1))
