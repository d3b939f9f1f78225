"""AKAZE61 scale space + Hessian throughput (config #5: 1280 x 720), device-resident frames.
usage: python tools/bench_akaze.py [batch] [steps]"""
import importlib
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
afv = importlib.import_module("anyfeature-vslam_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 1280, 720
ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=B))
frames = torch.from_numpy(afv.synth.corners_batch(1, B, W, H)).cuda()
for _ in range(2):
    ctx.scale_space_device(frames)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.scale_space_device(frames)
ctx.synchronize()
dt = (time.perf_counter() - t0) / steps
plan = ctx.plan
px0 = W * H
# algorithmic HBM bytes per frame (each plane moved once per kernel that must touch it)
by = px0 + 4 * px0            # level 0: read u8, write Lt
by += px0 + 4 * px0 + 4 * px0 + 4 * px0   # k-percentile: read u8, write smoothed, read it, write |grad|, read |grad|
by += 4 * px0
for i in range(1, plan.nlevels):
    L, Q = plan.lv[i], plan.lv[i - 1]
    n = L.w * L.h
    if L.octave > Q.octave:
        by += 4 * Q.w * Q.h + 4 * n       # halfsample
    by += 8 * n                            # gauss: Lt -> Lsmooth
    by += 8 * n                            # flow
    by += L.nsteps * 12 * n                # FED steps: read Lt, flow; write Lt
for i in range(plan.nlevels):
    n = plan.lv[i].w * plan.lv[i].h
    by += 12 * n + 20 * n                  # deriv1: read 1 write 2; hessian: read 2 write 3
print("batch %d: %.3f ms per step, %.1f frames/s, algorithmic %.1f MB/frame -> %.0f GB/s (%.1f %% of 8 TB/s)" %
      (B, dt * 1e3, B / dt, by / 1e6, by * B / dt / 1e9, by * B / dt / 8e12 * 100))
ctx.profile_enable(True)
ctx.scale_space_device(frames)
print(ctx.profile_read())
