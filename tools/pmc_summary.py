#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch."""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for p in paths:
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if not k.startswith("k_"):
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print("    %-28s n=%-3d mean=%.4g" % (c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main(sys.argv[1:])
