#!/usr/bin/env python3
"""Kernel timeline of bench.py's overlap_match step (256 frames, every second (t, t-1) pair overlapping) under rocprofv3 --kernel-trace:
per kernel launch the start / end relative to the step, for the LAST step.  Usage (GPU box): python tools/overlap_trace.py [resolve_engine]"""
import csv
import glob
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def inner(engine):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    afv = importlib.import_module("anyfeature-vslam_amd")
    B, W, H = 256, 640, 480
    ctx = afv.Context(max_batch=B, device=0)
    ctx.set_match_resolve(engine)
    m = afv.FeatureMatcher(0.6, True, ctx=ctx)
    base = afv.synth.corners_batch(9001, B // 2, W, H)
    fr = np.empty((B, H, W), np.uint8)
    fr[0::2] = base
    fr[1::2] = np.roll(base, 3, axis=2)
    frames = torch.from_numpy(fr).cuda(0)
    cap = ctx.cap
    dev = frames.device
    kps = torch.empty((B, cap, 7), dtype=torch.float32, device=dev); desc = torch.empty((B, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.empty((B,), dtype=torch.int32, device=dev); st = torch.zeros((1,), dtype=torch.int32, device=dev)
    match = torch.empty((B, cap), dtype=torch.int32, device=dev); nm = torch.empty((B,), dtype=torch.int32, device=dev)
    pa = torch.arange(B, dtype=torch.int32, device=dev)
    pb = (pa + (B - 1)) % B
    side = torch.cuda.Stream(dev)
    for _ in range(5):
        with torch.cuda.stream(side):
            ctx.extract_batch_device(frames, kps, desc, n, st, cap)
            m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True, match=match, nmatches=nm)
        torch.cuda.synchronize()


def main():
    engine = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    out = os.path.join(ROOT, "gpurun_out", "overlap_trace")
    os.makedirs(out, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "-d", out, "-o", "ov", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "inner", str(engine)]
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    rows = list(csv.DictReader(open(sorted(glob.glob(os.path.join(out, "**", "ov_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
    ev = [e for e in ev if e[2].startswith("k_")]
    per = len(ev) // 5
    last = ev[-per:]
    t0 = last[0][0]
    print("engine", engine, "step %.1f us" % ((max(e[1] for e in last) - t0) / 1e3))
    for s, e, n in last:
        if "match" in n or "describe" in n:
            print("  %-28s %8.1f .. %8.1f  (%6.1f us)" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "inner":
        inner(int(sys.argv[2]))
    else:
        main()
