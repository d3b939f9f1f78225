#!/usr/bin/env python3
"""Per-frame PMC figures from rocprofv3 counter_collection.csv files of `python bench.py --batch B --steps S --warmup W --cpu-frames 0`:
   python tools/pmc_per_frame.py B nsteps_total valu.csv fetch.csv write.csv  ->  JSON on stdout
nsteps_total = S + W (every step of the run is profiled)."""
import csv
import json
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    return acc


def main():
    B, nsteps = int(sys.argv[1]), int(sys.argv[2])
    acc = defaultdict(float)
    for p in sys.argv[3:]:
        for k, v in load(p).items():
            acc[k] += v
    frames = B * nsteps
    out = {"frames": frames, "kernels": {}}
    for (k, c), v in sorted(acc.items()):
        if not k.startswith("k_"):
            continue
        out["kernels"].setdefault(k, {})[c + "_per_frame"] = v / frames
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
