#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats summary (…kernel_stats.csv) -> a small JSON stamped with the sha of the sources it ran on, so that
bench.py can quote a kernel's rocprof duration next to its own hipEvent figure and withhold it when the sources changed since.
  python tools/kernel_stats_json.py kernel_stats.csv out.json [frames_per_launch] [name filter,...]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_sha import csrc_sha  # noqa: E402


def short(name):
    n = name.replace("void ", "")
    return n.split("(")[0].strip()


def main():
    src, dst = sys.argv[1], sys.argv[2]
    fpl = float(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] not in ("", "-") else None
    filt = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    kernels = {}
    for r in csv.DictReader(open(src)):
        k = short(r["Name"])
        if filt and not any(f in k for f in filt):
            continue
        kernels[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
    out = {"csrc_sha": csrc_sha(), "source_csv": os.path.basename(src), "kernels": kernels}
    if fpl:
        out["frames_per_launch"] = fpl
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: round(v["avg_ns"] / 1e3, 2) for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
