#!/usr/bin/env python3
"""Kernel experiments: build variant libraries (-DAFV_EXP=<n>) into gpurun_out-independent scratch dirs and time the
extraction stages of each on the GPU.  Usage:
   python tools/experiments.py build 0 1 2 3      (here, CPU)   -> anyfeature-vslam_amd/build_exp/libafv_exp<n>.so
   python tools/experiments.py run 0 1 2 3        (GPU box)     -> stage times per variant
"""
import importlib
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anyfeature-vslam_amd")
EXP = os.path.join(PKG, "build_exp")


def build(n):
    spec = importlib.util.spec_from_file_location("afv_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    os.makedirs(EXP, exist_ok=True)
    return mod.build(force=True, extra_flags=["-DAFV_EXP=%d" % n], out=os.path.join(EXP, "libafv_exp%d.so" % n),
                     objdir=os.path.join(EXP, "obj%d" % n))


def run_one(n, batch=256, steps=5):
    env = dict(os.environ, AFV_LIB_PATH=os.path.join(EXP, "libafv_exp%d.so" % n))
    code = r'''
import importlib, sys, json
sys.path.insert(0, %r)
import torch
afv = importlib.import_module("anyfeature-vslam_amd")
B = %d
ctx = afv.Context(max_batch=B)
ctx.set_split_threshold(1 << 30)  # one stream: clean per-kernel times
frames = torch.from_numpy(afv.synth.corners_batch(1, B)).cuda()
afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
m = afv.FeatureMatcher(0.6, True, ctx=ctx)
pa = torch.arange(B, dtype=torch.int32, device="cuda"); pb = (pa + B - 1) %% B
for _ in range(2):
    out = ctx.extract_batch_device(frames); m.match_pairs_device(out[1], out[0], out[2], pa, pb, th_low=75.0)
torch.cuda.synchronize()
ctx.profile_enable(True)
for _ in range(%d):
    out = ctx.extract_batch_device(frames); mm = m.match_pairs_device(out[1], out[0], out[2], pa, pb, th_low=75.0)
torch.cuda.synchronize()
st = ctx.profile_read()
print(json.dumps({k: round(v["total_ms"] / max(v["launches"], 1), 4) for k, v in st.items()}), int(out[2].sum().item()), "nm_sum", int(mm[1].sum().item()), "nm_max", int(mm[1].max().item()))
''' % (ROOT, batch, steps)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("exp %d:" % n, (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1])


if __name__ == "__main__":
    cmd, ids = sys.argv[1], [int(a) for a in sys.argv[2:]]
    for n in ids:
        if cmd == "build":
            print(build(n))
        else:
            run_one(n)
