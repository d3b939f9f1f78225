"""AKAZE61 (config #5: 1280 x 720) throughput: scale space + detection + quadtree + MLDB descriptors, device-resident frames;
CPU oracle timed beside it on a bounded sample.
usage: python tools/bench_akaze.py [batch] [steps] [cpu_frames]"""
import importlib
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
afv = importlib.import_module("anyfeature-vslam_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cpu_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 0
W, H = 1280, 720
ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=B))
frames_h = afv.synth.corners_batch(1, B, W, H)
frames = torch.from_numpy(frames_h).cuda()
for _ in range(2):
    ctx.extract_device(frames)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.extract_device(frames)
ctx.synchronize()
dt = (time.perf_counter() - t0) / steps
nk = sum(len(ctx.features(f)[0]) for f in range(B))
det = sum(len(ctx.keypoints(f)) for f in range(B))
t0 = time.perf_counter()
for _ in range(steps):
    ctx.scale_space_device(frames)
ctx.synchronize()
dt_ss = (time.perf_counter() - t0) / steps
plan = ctx.plan
px0 = W * H
# Algorithmic HBM bytes per frame of scale space + Hessian, each datum moved once: level 0 reads the u8 frame twice (Gaussian
# and contrast percentile) and writes Lt; every further level reads the previous Lt, writes Lsmooth and Lt; the Hessian reads
# Lsmooth and writes Lx, Ly, Ldet.  (The kernel structure moves more: the per-kernel sum is reported as "kernel_bytes".)
strict = 2 * px0 + 4 * px0
kern = px0 + 4 * px0 + px0 + 4 * px0 + 4 * px0 + 4 * px0 + 4 * px0
for i in range(1, plan.nlevels):
    L, Q = plan.lv[i], plan.lv[i - 1]
    n = L.w * L.h
    strict += 4 * Q.w * Q.h if L.octave > Q.octave else 4 * n
    strict += 8 * n
    if L.octave > Q.octave:
        kern += 4 * Q.w * Q.h + 4 * n
    kern += 8 * n + 12 * n        # gauss (Lt -> Lsmooth), fused level kernel (Lt, Lsmooth -> Lt)
for i in range(plan.nlevels):
    n = plan.lv[i].w * plan.lv[i].h
    strict += 16 * n
    kern += 32 * n                # deriv1 (1 -> 2 planes) + hessian (2 -> 3 planes)
by = strict
out = {"workload": "AKAZE61 1280x720 synthetic corners frames, omax 2 x 4 sublevels, dthreshold 0.0005, 1000-feature quadtree, MLDB-486",
       "batch": B, "ms_per_step": dt * 1e3, "frames_per_s": B / dt, "keypoints_per_s": nk / dt, "detected_per_frame": det / B,
       "described_per_frame": nk / B, "scale_space_ms_per_step": dt_ss * 1e3,
       "scale_space_algorithmic_MB_per_frame": by / 1e6, "scale_space_kernel_MB_per_frame": kern / 1e6,
       "roofline": {"bound": "hbm", "kernel": "scale space + Hessian (k_akz_*)", "achieved": by * B / dt_ss / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": by * B / dt_ss / 8e12, "kernel_structure_GBps": kern * B / dt_ss / 1e9}}
if cpu_frames:
    from oracle import akaze_binding as ak
    from oracle import binding as ob
    op = ak.make_plan(W, H)
    q = ctx.quotas()
    t0 = time.perf_counter()
    tot = 0
    for f in range(cpu_frames):
        levels, _ = ak.full_evolution(frames_h[f], op)
        kp = ak.subpixel(op, levels, ak.find_extrema(op, levels))
        chosen = []
        for lvl in range(op.nlevels):
            idx = np.nonzero(kp["class_id"] == lvl)[0]
            if len(idx):
                chosen.append(idx[ob.quadtree(kp["x"][idx], kp["y"][idx], kp["response"][idx], int(q[lvl]), W, H, tiebreak=np.arange(len(idx)))])
        kk, dd = ak.compute_descriptors(op, levels, kp[np.concatenate(chosen)])
        tot += len(kk)
    ct = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": tot / ct, "unit": "keypoints/s", "cores": 1, "kind": "port", "ms_per_frame": ct / cpu_frames * 1e3,
                           "sample": "%d frames 1280x720 through oracle/akaze.c + oracle quadtree, single thread" % cpu_frames}
print(json.dumps(out))
