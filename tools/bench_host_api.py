"""latency of the host-pointer entry points (what the reference's threads would call)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
afv = importlib.import_module("anyfeature-vslam_amd")
ctx = afv.Context()
s = afv.synth
img = s.corners_frame(1)
for _ in range(3): ctx.extract(img)
t = time.perf_counter(); N = 50
for _ in range(N): k, d = ctx.extract(img)
print("afv_orb_extract (1 frame, host buffers): %.3f ms" % ((time.perf_counter() - t) / N * 1e3))
k2, d2 = ctx.extract(np.roll(img, 4, axis=1))
afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
m = afv.FeatureMatcher(0.6, True, ctx=ctx)
def fv(n, nn, seed):
    node = s.lcg_states(seed, n) % nn
    return [(int(q), np.nonzero(node == q)[0].tolist()) for q in range(nn) if (node == q).any()]
v1 = afv.FeatureView(d, angles=k["angle"]); v2 = afv.FeatureView(d2, angles=k2["angle"])
b1 = afv.FeatureView(d, fv(len(d), 100, 3), angles=k["angle"], valid=np.ones(len(d), np.uint8))
b2 = afv.FeatureView(d2, fv(len(d2), 100, 4), angles=k2["angle"], valid=np.ones(len(d2), np.uint8))
for name, a, b in (("brute force 1000x1000", v1, v2), ("BoW 100 nodes", b1, b2)):
    for _ in range(3): m.SearchByBoW(a, b)
    t = time.perf_counter()
    for _ in range(N): r = m.SearchByBoW(a, b)
    print("SearchByBoW %-24s %.3f ms  (%d matches)" % (name, (time.perf_counter() - t) / N * 1e3, r[1]))
pairs = [(b1, b2)] * 20
for _ in range(2): m.SearchByBoW_batch(pairs)
t = time.perf_counter()
for _ in range(10): m.SearchByBoW_batch(pairs)
print("SearchByBoW_batch 20 BoW jobs: %.3f ms" % ((time.perf_counter() - t) / 10 * 1e3))
