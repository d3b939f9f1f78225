#!/usr/bin/env python3
"""Kernel-level timeline of the per-frame plugin call (bench.py's single_frame.contexts_1 workload: one extract + match call per frame,
frame t against an UNRELATED frame t - 1, one context, one stream), under rocprofv3 --kernel-trace (on the GPU box).
Usage: python tools/single_frame_trace.py [small_path_mode]      -> gpurun_out/single_frame_trace/summary.json
With argument `inner` the script is the traced workload itself."""
import csv
import glob
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPS, WARM = 40, 8


def inner(mode):
    sys.path.insert(0, ROOT)
    import torch
    afv = importlib.import_module("anyfeature-vslam_amd")
    dev = torch.device("cuda", 0)
    frames_h = afv.synth.corners_batch(7001, 8, 640, 480)
    c = afv.Context(max_batch=2, device=0)
    c.set_small_batch_path(mode)
    m = afv.FeatureMatcher(0.6, True, ctx=c)
    cap = c.cap
    fr = torch.from_numpy(frames_h[:2].copy()).to(dev)
    kps = torch.empty((2, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.empty((2, cap, 32), dtype=torch.uint8, device=dev)
    n = torch.empty((2,), dtype=torch.int32, device=dev)
    s = torch.zeros((1,), dtype=torch.int32, device=dev)
    match = torch.empty((1, cap), dtype=torch.int32, device=dev)
    nm = torch.empty((1,), dtype=torch.int32, device=dev)
    pa = torch.tensor([1], dtype=torch.int32, device=dev)
    pb = torch.tensor([0], dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        c.extract_batch_device(fr[0:1], kps[0:1], desc[0:1], n[0:1], s, cap)
        for _ in range(WARM + REPS):
            c.extract_batch_device(fr[1:2], kps[1:2], desc[1:2], n[1:2], s, cap)
            m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True, match=match, nmatches=nm)
    torch.cuda.synchronize(dev)


def main():
    mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    out = os.path.join(ROOT, "gpurun_out", "single_frame_trace")
    os.makedirs(out, exist_ok=True)
    cmd = ["rocprofv3", "--kernel-trace", "-d", out, "-o", "sf", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "inner", str(mode)]
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    rows = list(csv.DictReader(open(sorted(glob.glob(os.path.join(out, "**", "sf_kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
    ev = [e for e in ev if e[2].startswith("k_")]
    per = len(ev) // (WARM + REPS + 1) if ev else 0
    # the timed calls: everything after the warm-up calls (the first call extracts only)
    first = next(i for i, e in enumerate(ev) if "match" in e[2])  # end of the first extraction
    per_call = [e for e in ev[first:]]
    ncall = sum(1 for e in per_call if e[2].startswith("k_match_resolve"))
    tail = per_call[len(per_call) - (len(per_call) // ncall) * REPS:] if ncall else per_call
    stats = {}
    for s_, e_, n_ in tail:
        a = stats.setdefault(n_, [0, 0.0])
        a[0] += 1
        a[1] += (e_ - s_) / 1e3
    t0, t1 = tail[0][0], max(e[1] for e in tail)
    busy = sum(e[1] - e[0] for e in tail)
    res = {"mode": mode, "calls": REPS, "wall_us_per_call": (t1 - t0) / 1e3 / REPS, "busy_us_per_call": busy / 1e3 / REPS,
           "kernels": {k: {"launches_per_call": c_ / REPS, "avg_us": us / c_} for k, (c_, us) in sorted(stats.items())}}
    json.dump(res, open(os.path.join(out, "summary_mode%d.json" % mode), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "inner":
        inner(int(sys.argv[2]))
    else:
        main()
