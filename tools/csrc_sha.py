#!/usr/bin/env python3
"""sha256 over the kernel / runtime sources the ORB32 / matcher profiles depend on (csrc/*.hip, *.h, *.inc, include/*.h and build.py, sorted
by name; the AKAZE-only sources — *akaze* — are left out: they never run in the default or the pairs10k workload).
Every profiles/r*/{traffic,valu}_pmc.json carries the value it was collected on; bench.py marks a figure "stale" when the
tree no longer matches."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha(root=ROOT):
    h = hashlib.sha256()
    files = []
    for pat in ("anyfeature-vslam_amd/csrc/*.hip", "anyfeature-vslam_amd/csrc/*.h", "anyfeature-vslam_amd/csrc/*.inc", "include/*.h"):
        files += glob.glob(os.path.join(root, pat))
    files = [f for f in files if "akaze" not in os.path.basename(f)]
    files.append(os.path.join(root, "anyfeature-vslam_amd", "build.py"))  # the flags are part of what the kernels are (round 6)
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
