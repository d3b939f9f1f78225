"""measurement: per-kernel times of the AKAZE scale space with a variant library: python probe_akz_fed.py LIB [batch]"""
import importlib, sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] != "base":
    importlib.import_module("anyfeature-vslam_amd._lib").use_library(sys.argv[1])
afv = importlib.import_module("anyfeature-vslam_amd")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=B))
frames = torch.from_numpy(afv.synth.corners_batch(1, B, 1280, 720)).cuda()
for _ in range(2):
    ctx.scale_space_device(frames)
ctx.synchronize()
t = time.perf_counter()
for _ in range(5):
    ctx.scale_space_device(frames)
ctx.synchronize()
print(sys.argv[1], "scale space ms/step", (time.perf_counter() - t) / 5 * 1e3)
