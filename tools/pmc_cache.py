#!/usr/bin/env python3
"""L2 hit rate and LDS bank-conflict figures per kernel from rocprofv3 --pmc counter_collection.csv files (the LDS pass and the
TCC_HIT / TCC_MISS pass of tools/collect_profiles.sh): writes OUT_DIR/cache_pmc.json stamped with the sha of the sources it was
collected on (tools/csrc_sha.py).  bench.py reports it as the `cache` key (null + stale when the sources changed since).
Usage: python tools/pmc_cache.py OUT_DIR lds.csv l2.csv"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_sha import csrc_sha  # noqa: E402


def main():
    out = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(float))
    for p in sys.argv[2:]:
        if not os.path.exists(p):
            continue
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("k_"):
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {"csrc_sha": csrc_sha(), "kernels": {},
           "note": "rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum / --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT ... (separate passes) over "
                   "python bench.py --batch 256 --steps 3 --warmup 1 --no-extras; sums over the launches of a kernel.  l2_hit = TCC_HIT / "
                   "(TCC_HIT + TCC_MISS) (MI355X_MICROARCH.md, L2 section); lds_conflict_cycles_per_inst = SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS"}
    for k, c in sorted(acc.items()):
        e = {}
        if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
            e["l2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
            e["l2_requests"] = c["TCC_HIT_sum"] + c["TCC_MISS_sum"]
        if c.get("SQ_INSTS_LDS", 0) > 0:
            e["lds_insts"] = c["SQ_INSTS_LDS"]
            e["lds_conflict_cycles_per_inst"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_INSTS_LDS"]
            e["lds_wait_cycles_per_inst"] = c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_INSTS_LDS"]
        if e:
            res["kernels"][k] = e
    json.dump(res, open(os.path.join(out, "cache_pmc.json"), "w"), indent=1)
    print(json.dumps(res["kernels"], indent=1))


if __name__ == "__main__":
    main()
