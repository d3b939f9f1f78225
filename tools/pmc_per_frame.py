#!/usr/bin/env python3
"""Per-frame PMC figures from rocprofv3 counter_collection.csv files of
   python bench.py --batch B --steps S --warmup W --cpu-frames 0 --no-profile --no-extras
Usage: python tools/pmc_per_frame.py B nsteps_total OUT_DIR fetch.csv write.csv valu.csv [calib_fetch.json] [calib_valu.json]
nsteps_total = S + W (every step of the run is profiled).  Writes OUT_DIR/traffic_pmc.json and OUT_DIR/valu_pmc.json, each stamped
with the sha of the sources it was collected on (tools/csrc_sha.py)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_sha import csrc_sha  # noqa: E402


def load(path):
    acc = defaultdict(float)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"])
    return acc


def main():
    B, nsteps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    fetch, write, valu = load(sys.argv[4]), load(sys.argv[5]), load(sys.argv[6])
    cal_f = json.load(open(sys.argv[7])) if len(sys.argv) > 7 and os.path.exists(sys.argv[7]) else None
    cal_v = json.load(open(sys.argv[8])) if len(sys.argv) > 8 and os.path.exists(sys.argv[8]) else None
    frames = B * nsteps
    sha = csrc_sha()
    # FETCH_SIZE / WRITE_SIZE are in KiB; their ratio to real bytes is calibrated on this box for the access widths used
    # (tools/calib_fetch.hip): hbm_bytes = FETCH_SIZE * 1024 / fetch_ratio + WRITE_SIZE * 1024 / write_ratio
    fr = cal_f["fetch_ratio_4B"] if cal_f else 0.5
    wr = cal_f["write_ratio_4B"] if cal_f else 1.0
    traffic = {"csrc_sha": sha, "batch": B, "steps_profiled": nsteps,
               "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --batch %d --steps %d "
                       "--warmup 1 --cpu-frames 0 --no-profile --no-extras; per FRAME.  Calibration (tools/calib_fetch.hip on the same box): "
                       "FETCH_SIZE*1024 = %.3f x bytes read, WRITE_SIZE*1024 = %.3f x bytes written => hbm_bytes = FETCH*1024/%.3f + WRITE*1024/%.3f "
                       "(MI355X_MICROARCH.md HBM section).  Fabric-side requests: Infinity-Cache hits included." % (B, nsteps - 1, fr, wr, fr, wr),
               "kernels": {}}
    for (k, c), v in sorted(fetch.items()):
        if c == "FETCH_SIZE":
            traffic["kernels"].setdefault(k, {})["FETCH_SIZE_KB_per_frame"] = v / frames
    for (k, c), v in sorted(write.items()):
        if c == "WRITE_SIZE":
            traffic["kernels"].setdefault(k, {})["WRITE_SIZE_KB_per_frame"] = v / frames
    for k, d in traffic["kernels"].items():
        d["hbm_bytes_per_frame"] = d.get("FETCH_SIZE_KB_per_frame", 0.0) * 1024 / fr + d.get("WRITE_SIZE_KB_per_frame", 0.0) * 1024 / wr
    json.dump(traffic, open(os.path.join(out, "traffic_pmc.json"), "w"), indent=1)
    peak = cal_v["valu_peak_winst_per_s"] if cal_v else 1024 * 2.4e9 / 4
    vj = {"csrc_sha": sha, "batch": B, "steps_profiled": nsteps,
          "note": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU -- the same bench command; wave-instructions per FRAME, summed over the launches of "
                  "a kernel.  valu_peak_winst_per_s = integer-VALU issue peak measured by tools/calib_valu.hip on the same box (calib_valu.json).",
          "valu_peak_winst_per_s": peak, "kernels": {}}
    tot = 0.0
    for (k, c), v in sorted(valu.items()):
        if c == "SQ_INSTS_VALU":
            vj["kernels"][k] = {"valu_winst_per_frame": v / frames}
            tot += v / frames
    vj["valu_winst_per_frame_total"] = tot
    json.dump(vj, open(os.path.join(out, "valu_pmc.json"), "w"), indent=1)
    print(json.dumps({"traffic": traffic["kernels"], "valu": vj["kernels"], "sha": sha}, indent=1))


if __name__ == "__main__":
    main()
