"""round 6: per-level phase times of k_select_quadtree_wide (the -DAFV_SELECT_STATS build prints them): python tools/probes/select_stats.py LIB"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
afv = importlib.import_module("anyfeature-vslam_amd")
afv._lib.use_library(sys.argv[1])
ctx = afv.Context()
for name, img in (("corners", afv.synth.corners_frame(1)), ("toy", np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "toy_gray.npz"))["gray"])):
    ctx.extract(img)
    print("==", name, flush=True)
    ctx.extract(img)
