#!/usr/bin/env python3
"""per-launch counter means of the tracking-path kernels (projection searches, BoW descent, frame grid, ...) from two rocprofv3 --pmc
passes over tools/host_latency: python tools/pmc_tracking.py <sq_counters.csv> <l2_counters.csv> <out.json>"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_sha import csrc_sha  # noqa: E402

KEEP = ("k_proj", "k_init", "k_match_fuse", "k_bow", "k_frame", "k_featvec", "k_table_promote", "k_match_bow")


def short(name):
    return name.replace("void ", "").split("(")[0].strip()


def load(path, acc):
    if not path or not os.path.exists(path):
        return
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not any(k.startswith(p) for p in KEEP):
            continue
        d = acc.setdefault(k, {})
        e = d.setdefault(r["Counter_Name"], [0.0, 0])
        e[0] += float(r["Counter_Value"])
        e[1] += 1


def main():
    acc = {}
    load(sys.argv[1], acc)
    load(sys.argv[2] if len(sys.argv) > 2 else None, acc)
    out = {"csrc_sha": csrc_sha(), "per_launch": {}}
    for k, d in sorted(acc.items()):
        m = {c: v[0] / v[1] for c, v in d.items()}
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) > 0:
            m["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        if m.get("SQ_INSTS_LDS"):
            m["lds_conflict_cycles_per_inst"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_INSTS_LDS"]
        out["per_launch"][k] = {c: round(v, 4) for c, v in m.items()}
        print(k, json.dumps(out["per_launch"][k]))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
