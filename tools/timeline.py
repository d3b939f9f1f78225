#!/usr/bin/env python3
"""Where does a step go?  Runs bench.py under rocprofv3 --kernel-trace (on the GPU box) and prints, for the timed steps, every kernel's
launch count / average / summed duration, the union of the busy intervals, the time two or more kernels overlap and the idle gaps.
Usage: python tools/timeline.py [bench args]      -> gpurun_out/timeline/summary.json"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = os.path.join(ROOT, "gpurun_out", "timeline")
    os.makedirs(out, exist_ok=True)
    steps = 6
    cmd = ["rocprofv3", "--kernel-trace", "-d", out, "-o", "tl", "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--steps", str(steps), "--warmup", "2", "--cpu-frames", "0", "--no-profile", "--no-extras"] + sys.argv[1:]
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    rows = list(csv.DictReader(open(glob.glob(os.path.join(out, "**", "tl_kernel_trace.csv"), recursive=True)[0])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)  # template kernels print as "void k_x<...>(...)"
    ev = [e for e in ev if e[2].startswith("k_")]
    per_step = len(ev) // (steps + 2)          # warm-up 2 + timed steps, every step launches the same kernels
    ev = ev[-per_step * steps:]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    stats = {}
    for s, e, n in ev:
        a = stats.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    # sweep for busy / overlap
    pts = sorted([(s, 1) for s, e, n in ev] + [(e, -1) for s, e, n in ev])
    depth, last, busy, multi = 0, t0, 0.0, 0.0
    for t, d in pts:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            multi += t - last
        depth += d
        last = t
    res = {"steps": steps, "wall_ms_per_step": (t1 - t0) / 1e6 / steps, "busy_ms_per_step": busy / 1e6 / steps,
           "overlap2_ms_per_step": multi / 1e6 / steps, "idle_ms_per_step": ((t1 - t0) - busy) / 1e6 / steps,
           "kernels": {n: {"launches_per_step": c / steps, "avg_us": us / c, "sum_ms_per_step": us / 1e3 / steps} for n, (c, us) in sorted(stats.items())}}
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
