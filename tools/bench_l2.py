"""config #3 matcher: 1000 x 1000 x 128 f32 L2^2 brute force (afv_match_l2), host-API latency; kernel times via rocprofv3."""
import importlib
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, ".")
afv = importlib.import_module("anyfeature-vslam_amd")
s = afv.synth
n, dim = 1000, 128
a = (s.lcg_bytes(1, n * dim).reshape(n, dim).astype(np.float32)) ** 2
a /= np.linalg.norm(a, axis=1, keepdims=True)
noise = (s.lcg_bytes(2, n * dim).reshape(n, dim).astype(np.float32) - 128) / 2000.0
b = np.abs(a + noise).astype(np.float32)
b /= np.linalg.norm(b, axis=1, keepdims=True)
b = b[np.argsort(s.lcg_states(3, n), kind="stable")].copy()
ctx = afv.Context()
m = afv.FeatureMatcher(0.8, False, ctx=ctx)
for rep in range(5):
    t0 = time.perf_counter()
    got, gn = m.match_l2(a, b, 0.5, 0.8)
    dt = time.perf_counter() - t0
print("afv_match_l2 1000x1000x128: %.3f ms per call, %d matches" % (dt * 1e3, gn))
