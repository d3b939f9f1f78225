"""-m gpu: SURVEY 8f rank 1 — projection-guided matching core (grid window + Hamming) vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(afv, gpu_ctx, seed, shift, radius_scale, perm=True):
    """frame features = keypoints of a frame; queries = keypoints of the same frame shifted by `shift` px, 'projected'
    back with that offset (what a motion model does), window r = radius_scale * size"""
    s = afv.synth
    img = s.corners_frame(seed)
    k1, d1 = gpu_ctx.extract(img)
    k2, d2 = gpu_ctx.extract(np.roll(img, shift, axis=1))
    size1, _, _ = gpu_ctx.size_sigma(k1)
    size2, _, _ = gpu_ctx.size_sigma(k2)
    occ = (s.lcg_bytes(seed + 9, len(k1)) < 30).astype(np.uint8)
    F = afv.FrameGridView(d1, np.stack([k1["x"], k1["y"]], 1), size1, angles=k1["angle"], occupied=occ)
    order = np.argsort(s.lcg_states(seed + 5, len(k2)), kind="stable") if perm else np.arange(len(k2))
    k2, d2, size2 = k2[order], d2[order], size2[order]
    u = k2["x"] - np.float32(shift) + ((s.lcg_states(seed + 6, len(k2)) % 5).astype(np.float32) - 2)
    v = k2["y"] + ((s.lcg_states(seed + 7, len(k2)) % 5).astype(np.float32) - 2)
    r = np.float32(radius_scale) * size2
    valid = (s.lcg_bytes(seed + 8, len(k2)) > 20).astype(np.uint8)
    occupies = (s.lcg_bytes(seed + 10, len(k2)) > 10).astype(np.uint8)
    Q = afv.ProjectionQueries(d2, u, v, r, size2 / np.float32(1.2), size2 * np.float32(1.2), valid=valid, angles=k2["angle"],
                              occupies=occupies)
    return F, Q


@pytest.mark.parametrize("seed,shift,rs", [(1, 4, 15.0), (2, 7, 40.0), (3, 0, 6.0), (4, 3, 120.0)])
def test_local_map_mode(afv, oracle, gpu_ctx, seed, shift, rs):
    F, Q = _scene(afv, gpu_ctx, seed, shift, rs)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.8, True, ctx=gpu_ctx)
    got, n = m.SearchByProjection(F, Q)
    want, wn = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    assert n == wn and np.array_equal(got, want)
    assert wn > 100


@pytest.mark.parametrize("ori", [False, True])
@pytest.mark.parametrize("seed,shift,rs", [(5, 5, 15.0), (6, 2, 60.0)])
def test_last_frame_mode(afv, oracle, gpu_ctx, seed, shift, rs, ori):
    F, Q = _scene(afv, gpu_ctx, seed, shift, rs)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.9, ori, ctx=gpu_ctx)
    got, n = m.SearchByProjection(F, Q, last_frame=True)
    want, wn = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.9, check_orientation=ori, last_frame=True)
    assert n == wn and np.array_equal(got, want)
    assert wn > 100


def test_projection_edge_cases(afv, oracle, gpu_ctx):
    F, Q = _scene(afv, gpu_ctx, 7, 4, 15.0)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.8, False, ctx=gpu_ctx)
    # windows entirely outside the image / zero radius / no queries / no features
    Q.u[:50] = -500.0
    Q.v[50:100] = 5000.0
    Q.r[100:150] = 0.0
    got, n = m.SearchByProjection(F, Q)
    want, wn = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    assert n == wn and np.array_equal(got, want)
    empty_q = afv.ProjectionQueries(np.zeros((0, 32), np.uint8), [], [], [], [], [])
    got, n = m.SearchByProjection(F, empty_q)
    assert n == 0 and np.all(got == -1)
    empty_f = afv.FrameGridView(np.zeros((0, 32), np.uint8), np.zeros((0, 2), np.float32), [])
    got, n = m.SearchByProjection(empty_f, Q)
    assert n == 0 and len(got) == 0
    # everything occupied: nothing can be assigned
    F.occupied[:] = 1
    got, n = m.SearchByProjection(F, Q)
    assert n == 0 and np.all(got == -1)


def test_projection_dense_cluster_forces_rescans(afv, oracle, gpu_ctx):
    """all queries aim at the same small window: heavy competition for a handful of features (claim / replay and the exact
    window rescan)"""
    s = afv.synth
    n, nq = 60, 400
    proto = s.random_descriptors(77, 6)
    d = proto[s.lcg_states(1, n) % 6].copy()
    d[np.arange(n), s.lcg_states(2, n) % 32] ^= 1
    pts = np.stack([300 + (s.lcg_states(3, n) % 40).astype(np.float32), 200 + (s.lcg_states(4, n) % 40).astype(np.float32)], 1)
    F = afv.FrameGridView(d, pts, np.ones(n, np.float32))
    qd = proto[s.lcg_states(5, nq) % 6].copy()
    qd[np.arange(nq), s.lcg_states(6, nq) % 32] ^= 2
    Q = afv.ProjectionQueries(qd, np.full(nq, 320.0), np.full(nq, 220.0), np.full(nq, 30.0), np.full(nq, 0.5), np.full(nq, 2.0))
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    for mode, ratio in ((False, 0.8), (False, 2.0), (True, 0.9)):
        m = afv.FeatureMatcher(ratio, False, ctx=gpu_ctx)
        got, nn = m.SearchByProjection(F, Q, last_frame=mode)
        want, wn = oracle.match_projection(F, Q, th_high=75.0, nnratio=ratio, last_frame=mode)
        assert nn == wn and np.array_equal(got, want), (mode, ratio)
    assert wn == n  # every feature ends up taken


@pytest.mark.parametrize("seed,shift,rs", [(11, 4, 15.0), (12, 1, 50.0)])
def test_fuse_core(afv, oracle, gpu_ctx, seed, shift, rs):
    """Fuse(pKF, vpMapPoints): independent map points, size band + 5.99 reprojection gate, first minimum wins"""
    F, Q = _scene(afv, gpu_ctx, seed, shift, rs)
    F.inf = np.ascontiguousarray(np.float32(0.2) / (F.sizes * F.sizes))   # GetKeyPt1DInf ~ 1/sigma^2 (loosened)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.6, True, ctx=gpu_ctx)
    got, n = m.Fuse(F, Q)
    want, wn = oracle.match_projection(F, Q, th_high=75.0, fuse=True)
    assert n == wn and np.array_equal(got, want)
    assert 50 < wn < Q.n  # the gate and the threshold both bite
