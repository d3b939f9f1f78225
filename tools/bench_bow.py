"""BoW quantisation throughput: ORBvoc-shaped tree (k=10, L=6, 1.11 M nodes, 35.6 MB) and a batch of descriptors.
usage: python tools/bench_bow.py [n_desc]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import importlib

afv = importlib.import_module("anyfeature-vslam_amd")
from importlib import import_module

_lib = import_module("anyfeature-vslam_amd._lib")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1736 * 64
k, L = 10, 6
rng = np.random.default_rng(1)
counts = [k ** l for l in range(L + 1)]
nn = sum(counts)
parent = np.zeros(nn, np.int32)
start = np.cumsum([0] + counts)
for l in range(1, L + 1):
    parent[start[l]:start[l + 1]] = start[l - 1] + np.arange(counts[l]) // k
# hierarchical descriptors: child = parent with a few flipped bits, so descents are meaningful
desc = np.zeros((nn, 32), np.uint8)
for l in range(1, L + 1):
    p = desc[parent[start[l]:start[l + 1]]]
    flips = np.zeros((counts[l], 256), np.uint8)
    nflip = max(4, 48 >> (l - 1))
    idx = rng.integers(0, 256, (counts[l], nflip))
    np.put_along_axis(flips, idx, 1, axis=1)
    desc[start[l]:start[l + 1]] = p ^ np.packbits(flips, axis=1)
leaf = np.zeros(nn, bool); leaf[start[L]:] = True
ctx = afv.Context()
voc = afv.Vocabulary(k, L, parent, desc, np.ones(nn), leaf, ctx=ctx)
q = desc[rng.integers(start[L], nn, n)] ^ np.packbits((rng.random((n, 256)) < 0.05).astype(np.uint8), axis=1)
for rep in range(3):
    t0 = time.perf_counter()
    lf, nid = voc.transform_nodes(q, 4)
    dt = time.perf_counter() - t0
print("host API: n=%d  %.3f ms  (%.1f M desc/s incl. PCIe)" % (n, dt * 1e3, n / dt / 1e6))
ctx.profile_enable(True)
voc.transform_nodes(q, 4)
print(ctx.profile_read() if hasattr(ctx, "profile_read") else "")
