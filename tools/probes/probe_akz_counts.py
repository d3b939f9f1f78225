"""candidates per level entering the ordered suppression, and what leaves it, on the bench frames"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
afv = importlib.import_module("anyfeature-vslam_amd")
frames = afv.synth.corners_batch(1, 4, 1280, 720)
ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=4))
ctx.scale_space(frames); ctx.detect()
for f in range(2):
    c = [len(ctx.candidates(f, l)) for l in range(ctx.plan.nlevels)]
    k = ctx.keypoints(f)
    print("frame", f, "candidates per level", c, "sum", sum(c), "-> keypoints", len(k), "per level", np.bincount(k["octave"].astype(int) if "octave" in k.dtype.names else np.zeros(len(k), int)).tolist())
print([ (ctx.plan.lv[l].w, ctx.plan.lv[l].h, ctx.plan.lv[l].sigma_size, round(ctx.plan.lv[l].esigma,3)) for l in range(ctx.plan.nlevels)])
