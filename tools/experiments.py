#!/usr/bin/env python3
"""Kernel experiments: build variant libraries (extra -D flags) next to the real one under anyfeature-vslam_amd/build_exp/ and
time / count them on the GPU box.  Variants may produce wrong results (e.g. AFV_FAST_STOP=n ends k_fast_nms after stage n to
attribute its cost) — this is measurement tooling, never the product.  Usage:
  python tools/experiments.py build NAME=FLAG[,FLAG...] ...      (here or on the box)
  python tools/experiments.py run NAME ...                        (on the box: bench --no-profile under rocprofv3 PMC)
build_exp/ is git-ignored; delete it after use (it travels to the GPU box)."""
import csv
import glob
import importlib.util
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "anyfeature-vslam_amd")
EXP = os.path.join(PKG, "build_exp")


def build(name, flags):
    spec = importlib.util.spec_from_file_location("afv_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = os.path.join(EXP, name)
    os.makedirs(d, exist_ok=True)
    return mod.build(extra_flags=flags, out=os.path.join(EXP, "libafv_%s.so" % name), objdir=d)


def run(name, batch=256):
    lib = os.path.join(EXP, "libafv_%s.so" % name) if name != "base" else os.path.join(PKG, "libafv_hip.so")
    out = os.path.join(ROOT, "gpurun_out", "exp", name)
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    counters = os.environ.get("AFV_EXP_PMC", "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU").split()
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", out, "-o", "pmc",
           "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(batch), "--steps", "2", "--warmup", "1", "--repeat", "1",
           "--cpu-frames", "0", "--no-profile", "--no-extras", "--lib", lib]
    subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    kern = os.environ.get("AFV_EXP_KERNEL", "k_fast_nms")
    res = {"name": name, "kernel": kern}
    acc, cnt = {}, {}
    for r in csv.DictReader(open(glob.glob(os.path.join(out, "**", "pmc_counter_collection.csv"), recursive=True)[0])):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            cnt[r["Counter_Name"]] = cnt.get(r["Counter_Name"], 0) + 1
    # ONE definition of "per frame" for every PMC tool of this repository (tools/pmc_per_frame.py, tools/pmc_tracking.py, this file): the sum of
    # a counter over ALL launches of the kernel in the run, divided by the frames the run processed (batch x (steps + warm-up) = batch x 3).
    # (Round 5 divided a launch's average by batch / 4 here - the split was two chunks of 128 frames, not four of 64: its figures were 2 x
    # those of pmc_per_frame.py for the same kernel and counter)
    frames_total = batch * 3
    launches = max(cnt.values()) if cnt else 1
    frames_per_launch = frames_total / launches
    for k in acc:
        res[k + "_per_frame"] = acc[k] / frames_total
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in
         csv.DictReader(open(glob.glob(os.path.join(out, "**", "pmc_kernel_trace.csv"), recursive=True)[0])) if kern in r["Kernel_Name"]]
    res["us_per_launch"] = sum(d) / len(d)
    res["frames_per_launch"] = frames_per_launch
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        for a in sys.argv[2:]:
            name, flags = a.split("=", 1)
            print(build(name, ["-D" + f for f in flags.split(",") if f]))
    else:
        for a in sys.argv[2:]:
            run(a)
