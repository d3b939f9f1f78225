"""latency of the host-pointer entry points (what the reference's threads would call)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
afv = importlib.import_module("anyfeature-vslam_amd")


def med(fn, n=50, warm=3):
    """median wall time of one call in ms (the HIP runtime has occasional multi-ms housekeeping stalls: a mean would report those)"""
    for _ in range(warm):
        r = fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, r

ctx = afv.Context()
s = afv.synth
img = s.corners_frame(1)
N = 50
ms, (k, d) = med(lambda: ctx.extract(img))
print("afv_orb_extract (1 frame, host buffers): %.3f ms" % ms)
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
    ms, r = med(lambda: m.SearchByBoW(a, b))
    print("SearchByBoW %-24s %.3f ms  (%d matches)" % (name, ms, r[1]))
pairs = [(b1, b2)] * 20
ms, _ = med(lambda: m.SearchByBoW_batch(pairs), n=20)
print("SearchByBoW_batch 20 BoW jobs: %.3f ms" % ms)
# SearchForTriangulation: 100-node feature vectors, nothing has a map point yet, a plausible F12
has0 = np.zeros(len(d), np.uint8); has02 = np.zeros(len(d2), np.uint8)
p1 = np.stack([k["x"], k["y"]], 1); p2 = np.stack([k2["x"], k2["y"]], 1)
sz2, sg2, _ = ctx.size_sigma(k2)
t1 = afv.FeatureView(d, fv(len(d), 100, 3), has0, pts=p1)
t2 = afv.FeatureView(d2, fv(len(d2), 100, 4), has02, pts=p2, sigma2=sz2 * sz2)
F12 = np.array([[0, 0, 0], [0, 0, -1e-3], [0, 1e-3, 0]], np.float32)   # pure x translation: epipolar lines are rows
ms, r = med(lambda: m.SearchForTriangulation(t1, t2, F12, (1e6, 240.0)))
print("SearchForTriangulation 100 nodes: %.3f ms  (%d pairs)" % (ms, r[1]))
# projection searches
sz1, _, _ = ctx.size_sigma(k)
F = afv.FrameGridView(d, p1, sz1, angles=k["angle"])
Q = afv.ProjectionQueries(d2, k2["x"] - 4, k2["y"], 15.0 * sz2, sz2 / np.float32(1.2), sz2 * np.float32(1.2), angles=k2["angle"])
for name, kw in (("local map", {}), ("last frame", {"last_frame": True})):
    ms, r = med(lambda: m.SearchByProjection(F, Q, **kw))
    print("SearchByProjection %-12s %.3f ms  (%d matches)" % (name, ms, r[1]))
ms, r = med(lambda: m.Fuse_sim3(F, Q))
print("Fuse core: %.3f ms  (%d found)" % (ms, r[1]))
Q1 = afv.ProjectionQueries(d2, k2["x"], k2["y"], np.full(len(k2), 100.0, np.float32), np.zeros(len(k2), np.float32),
                           np.full(len(k2), 10.0, np.float32), valid=(k2["octave"] == 0).astype(np.uint8), angles=k2["angle"])
ms, r = med(lambda: m.SearchForInitialization(Q1, F))
print("SearchForInitialization (window 100): %.3f ms  (%d matches)" % (ms, r[1]))
voc = afv.Vocabulary.random(3, k=10, L=4, ctx=ctx)
ms, _ = med(lambda: voc.transform_nodes(d, 2))
print("Vocabulary descent (k=10, L=4, %d descriptors): %.3f ms" % (len(d), ms))
# AKAZE61 single frame (1280 x 720, host buffers)
actx = afv.AkazeContext(afv.akaze.default_params())
big = s.corners_batch(1, 1, 1280, 720)[0]
ms, r = med(lambda: actx.extract(big), n=10)
print("afv_akaze_extract (1 frame 1280x720, host buffers): %.3f ms  (%d keypoints)" % (ms, len(r[0])))
# relocalisation batch: ONE frame against 32 candidate keyframes of a device-resident table (Tracking::Relocalization, Tracking.cc:1162-1182)
tblm = importlib.import_module("anyfeature-vslam_amd.table")
tt, ta, tc = s.keyframe_table(32, 1000, seed=5)
table = tblm.DescriptorTable(ctx, 32, 1000)
for kf in range(32):
    table.set(kf, tt[kf], ta[kf])
    nodes = s.lcg_states(77, 1000) % 100
    f_ = [(int(q), np.nonzero(nodes == q)[0].tolist()) for q in range(100) if (nodes == q).any()]
    ids = np.array([q for q, _ in f_], np.int32); ptr_ = np.cumsum([0] + [len(v) for _, v in f_]).astype(np.int32)
    table.set_featvec(kf, ids, ptr_, np.array([x for _, v in f_ for x in v], np.int32))
fr = afv.FeatureView(s.perturbed_descriptors(tt[16].copy(), 99), f_, None, ta[16])
ms, r = med(lambda: table.match_bow_frame(np.arange(32, dtype=np.int32), fr, 75.0, 0.75, True), n=30)
print("afv_table_match_bow_frame, 1 frame x 32 keyframes (100 nodes): %.3f ms  (%d matches in all, %d against its own keyframe)" % (ms, int(r[1].sum()), int(r[1][16])))
