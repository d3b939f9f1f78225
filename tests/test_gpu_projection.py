"""-m gpu: SURVEY 8f rank 1 — projection-guided matching core (grid window + Hamming) vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["fixed_point", "ordered_walk"])
def proj_engine(request, gpu_ctx):
    """every case of this file through both engines of the ordered phase (afv_set_projection_resolve): the round-5 workgroup fixed
    point and the one-wavefront ordered walk of rounds 1-4 must give the same answers as the oracle"""
    gpu_ctx.check(gpu_ctx.lib.afv_set_projection_resolve(gpu_ctx.handle, request.param), "afv_set_projection_resolve")
    yield request.param
    gpu_ctx.lib.afv_set_projection_resolve(gpu_ctx.handle, 2)


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


# ---- stereo frames (FeatureMatcher.cc:114-119, :1367-1372, :880-894): mvuRight on the feature side, the projected right coordinate on
#      the query side ----
def _stereo(afv, F, Q, seed, gate_scale):
    """about half of the features get a right-image coordinate (mvuRight > 0, the others -1); a query's projected right coordinate is
    u minus a disparity, jittered so that the gate removes some candidates and keeps others"""
    s = afv.synth
    disp_f = (s.lcg_states(seed + 21, F.N) % 4000).astype(np.float32) / np.float32(100.0)
    has = s.lcg_bytes(seed + 22, F.N) > 110
    F.u_right = np.where(has, F.x - disp_f, np.float32(-1.0)).astype(np.float32)
    disp_q = (s.lcg_states(seed + 23, Q.n) % 4000).astype(np.float32) / np.float32(100.0)
    Q.ur = (Q.u - disp_q).astype(np.float32)
    Q.er_max = (np.float32(gate_scale) * Q.r).astype(np.float32)
    return F, Q


@pytest.mark.parametrize("last_frame", [False, True])
@pytest.mark.parametrize("seed,shift,rs,gate", [(31, 4, 15.0, 0.5), (32, 2, 60.0, 0.2), (33, 5, 25.0, 1.0)])
def test_projection_with_stereo_gate(afv, oracle, gpu_ctx, seed, shift, rs, gate, last_frame):
    F, Q = _stereo(afv, *_scene(afv, gpu_ctx, seed, shift, rs), seed, gate)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.85, True, ctx=gpu_ctx)
    got, n = m.SearchByProjection(F, Q, last_frame=last_frame)
    want, wn = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.85, check_orientation=last_frame, last_frame=last_frame)
    assert n == wn and np.array_equal(got, want)
    ur = F.u_right
    F.u_right = None  # the gate must have changed the outcome, or the test shows nothing
    mono, _ = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.85, check_orientation=last_frame, last_frame=last_frame)
    F.u_right = ur
    assert wn > 50 and not np.array_equal(mono, want)
    # the relocalisation / Sim3 flavours run through the same entry point but have no mvuRight branch: the wrapper drops it
    got2, n2 = m.SearchByProjection_sim3(F, Q)
    F.u_right = None
    want2, wn2 = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.85, check_orientation=False, last_frame=True)
    assert n2 == wn2 and np.array_equal(got2, want2)


@pytest.mark.parametrize("seed,shift,rs", [(41, 4, 15.0), (42, 1, 50.0)])
def test_fuse_core_with_stereo_keypoints(afv, oracle, gpu_ctx, seed, shift, rs):
    """Fuse(pKF, vpMapPoints) on a stereo keyframe: keypoints with mvuRight >= 0 pass the 3-dof gate (7.8), the others the 2-dof one"""
    F, Q = _stereo(afv, *_scene(afv, gpu_ctx, seed, shift, rs), seed, 1.0)
    F.inf = np.ascontiguousarray(np.float32(0.2) / (F.sizes * F.sizes))
    # right coordinates consistent with the projection up to a few pixels, so that the 3-dof gate sometimes holds and sometimes not
    Q.ur = (Q.u - np.float32(20.0) + (afv.synth.lcg_states(seed + 24, Q.n) % 9).astype(np.float32) - 4).astype(np.float32)
    F.u_right = np.where(F.u_right >= 0, F.x - np.float32(20.0), np.float32(-1.0)).astype(np.float32)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.6, True, ctx=gpu_ctx)
    got, n = m.Fuse(F, Q)
    want, wn = oracle.match_projection(F, Q, th_high=75.0, fuse=True)
    assert n == wn and np.array_equal(got, want)
    ur = F.u_right
    F.u_right = None
    mono, _ = oracle.match_projection(F, Q, th_high=75.0, fuse=True)
    F.u_right = ur
    assert 50 < wn < Q.n and not np.array_equal(mono, want)
    got3, n3 = m.Fuse_sim3(F, Q)  # no gate at all, stereo or not
    inf = F.inf
    F.inf = None
    want3, wn3 = oracle.match_projection(F, Q, th_high=75.0, fuse=True)
    F.inf = inf
    assert n3 == wn3 and np.array_equal(got3, want3)


# ---- the remaining rank-1 searches: relocalisation / Sim3 projection, Fuse(Sim3), SearchBySim3, SearchForInitialization ----
@pytest.mark.parametrize("use_high", [False, True])
def test_relocalisation_and_sim3_projection(afv, oracle, gpu_ctx, use_high):
    F, Q = _scene(afv, gpu_ctx, 21, 5, 30.0)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(60.0)
    m = afv.FeatureMatcher(0.9, True, ctx=gpu_ctx)
    afv.FeatureMatcher.descDistTh_high_reloc = 90.0
    th = 90.0 if use_high else 60.0
    got, n = m.SearchByProjection_reloc(F, Q, useHighMatchingThreshold=use_high)
    want, wn = oracle.match_projection(F, Q, th_high=th, nnratio=0.9, check_orientation=True, last_frame=True)
    assert n == wn and np.array_equal(got, want) and wn > 100
    got, n = m.SearchByProjection_sim3(F, Q)
    want, wn = oracle.match_projection(F, Q, th_high=60.0, nnratio=0.9, check_orientation=False, last_frame=True)
    assert n == wn and np.array_equal(got, want) and wn > 100
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)


def test_fuse_sim3_core(afv, oracle, gpu_ctx):
    F, Q = _scene(afv, gpu_ctx, 22, 4, 25.0)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.8, True, ctx=gpu_ctx)
    got, n = m.Fuse_sim3(F, Q)
    F.inf = None
    want, wn = oracle.match_projection(F, Q, th_high=75.0, fuse=True)
    assert n == wn and np.array_equal(got, want) and wn > 300


def test_search_by_sim3(afv, oracle, gpu_ctx):
    s = afv.synth
    img = s.corners_frame(23)
    k1, d1 = gpu_ctx.extract(img)
    k2, d2 = gpu_ctx.extract(np.roll(img, 6, axis=1))
    z1, _, _ = gpu_ctx.size_sigma(k1); z2, _, _ = gpu_ctx.size_sigma(k2)
    F1 = afv.FrameGridView(d1, np.stack([k1["x"], k1["y"]], 1), z1)
    F2 = afv.FrameGridView(d2, np.stack([k2["x"], k2["y"]], 1), z2)
    jit = lambda seed, n: (s.lcg_states(seed, n) % 7).astype(np.float32) - 3
    v1 = (s.lcg_bytes(31, len(k1)) > 40).astype(np.uint8); v2 = (s.lcg_bytes(32, len(k2)) > 40).astype(np.uint8)
    Q1 = afv.ProjectionQueries(d1, k1["x"] + 6 + jit(33, len(k1)), k1["y"] + jit(34, len(k1)), 12.0 * z1, z1 / np.float32(1.2),
                               z1 * np.float32(1.2), valid=v1)
    Q2 = afv.ProjectionQueries(d2, k2["x"] - 6 + jit(35, len(k2)), k2["y"] + jit(36, len(k2)), 12.0 * z2, z2 / np.float32(1.2),
                               z2 * np.float32(1.2), valid=v2)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.8, True, ctx=gpu_ctx)
    got, n = m.SearchBySim3(F1, Q1, F2, Q2)
    want, wn = oracle.match_sim3(F2, Q1, F1, Q2, th_high=75.0)
    assert n == wn and np.array_equal(got, want) and wn > 200
    assert np.all(got[v1 == 0] == -1)


@pytest.mark.parametrize("ori", [False, True])
@pytest.mark.parametrize("seed,shift,window", [(24, 6, 100.0), (25, 30, 40.0), (26, 0, 12.0)])
def test_search_for_initialization(afv, oracle, gpu_ctx, seed, shift, window, ori):
    s = afv.synth
    ctx2k = afv.Context(nfeatures=2000)
    img = s.corners_frame(seed)
    k1, d1 = ctx2k.extract(img)
    k2, d2 = ctx2k.extract(np.roll(img, shift, axis=1))
    z1, _, _ = ctx2k.size_sigma(k1); z2, _, _ = ctx2k.size_sigma(k2)
    F2 = afv.FrameGridView(d2, np.stack([k2["x"], k2["y"]], 1), z2, angles=k2["angle"])
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)          # vbPrevMatched starts as F1's own keypoints
    n1 = len(k1)
    Q1 = afv.ProjectionQueries(d1, prev[:, 0].copy(), prev[:, 1].copy(), np.full(n1, window, np.float32), np.zeros(n1, np.float32),
                               np.full(n1, z1.max(), np.float32), valid=(k1["octave"] == 0).astype(np.uint8), angles=k1["angle"])
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.9, ori, ctx=gpu_ctx)
    got, n = m.SearchForInitialization(Q1, F2, vbPrevMatched=prev)
    want, wn = oracle.match_initialization(F2, Q1, th_low=75.0, nnratio=0.9, check_orientation=ori)
    assert n == wn and np.array_equal(got, want)
    assert np.all(got[k1["octave"] != 0] == -1)
    if shift <= window:
        assert wn > 50
    mt = got >= 0
    assert np.array_equal(prev[mt, 0], F2.x[got[mt]]) and np.array_equal(prev[mt, 1], F2.y[got[mt]])
    # matches are one-to-one (stealing keeps vnMatches21 consistent)
    assert len(np.unique(got[mt])) == mt.sum()
    ctx2k.close()


def test_initialization_stealing_hand_case(afv, oracle, gpu_ctx):
    """three F1 features aim at the same F2 feature with decreasing distance: each later one steals it (:531-535); a fourth at
    a larger distance is skipped by the vMatchedDistance gate (:513) and falls back to its own second candidate."""
    d2 = np.zeros((2, 32), np.uint8); d2[1] = 0xFF
    F2 = afv.FrameGridView(d2, [[100.0, 100.0], [104.0, 100.0]], [1.0, 1.0], angles=[0.0, 0.0])
    d1 = np.zeros((4, 32), np.uint8)
    d1[0, :3] = 0xFF      # 24 bits from feature 0
    d1[1, :2] = 0xFF      # 16
    d1[2, :1] = 0xFF      # 8
    d1[3, :26] = 0xFF     # 208 from feature 0, 48 from feature 1 (gate hides feature 0 -> best = feature 1 alone)
    Q1 = afv.ProjectionQueries(d1, [100.0] * 4, [100.0] * 4, [20.0] * 4, [0.0] * 4, [10.0] * 4, valid=[1] * 4, angles=[0.0] * 4)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.9, False, ctx=gpu_ctx)
    got, n = m.SearchForInitialization(Q1, F2)
    want, wn = oracle.match_initialization(F2, Q1, th_low=75.0, nnratio=0.9, check_orientation=False)
    assert want.tolist() == [-1, -1, 0, 1] and wn == 2
    assert got.tolist() == want.tolist() and n == wn


def test_job_records_of_an_older_layout_and_without_a_size(afv, oracle, gpu_ctx):
    """a caller compiled against the layout BEFORE the stereo fields (struct_size = offsetof(u_right)) is served as the monocular call;
    a record whose struct_size is 0 / garbage is AFV_EINVAL instead of having its tail dereferenced"""
    import ctypes as C
    F, Q = _scene(afv, gpu_ctx, 3, 4, 15.0)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.8, False, ctx=gpu_ctx)
    j = m._proj_job(F, Q)
    j.th_high = 75.0; j.mode = 0
    # poison the tail, then declare the old size: the runtime must not look at it
    j.u_right = 0xdeadbeef; j.q_ur = 0xdeadbeef; j.q_er_max = 0xdeadbeef
    j.struct_size = afv._lib.ProjJob.u_right.offset
    out = np.full(F.N, -1, np.int32); nm = np.zeros(1, np.int32)
    jobs = (afv._lib.ProjJob * 1)(j)
    assert gpu_ctx.lib.afv_match_projection(gpu_ctx.handle, jobs, 1, out.ctypes.data, nm.ctypes.data) == 0
    want, wn = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    assert nm[0] == wn and np.array_equal(out, want)
    for bad in (0, 12, 7, 1 << 20):
        j.struct_size = bad
        jobs = (afv._lib.ProjJob * 1)(j)
        assert gpu_ctx.lib.afv_match_projection(gpu_ctx.handle, jobs, 1, out.ctypes.data, nm.ctypes.data) == afv._lib.EINVAL


# ---- descriptor-generic: the reference dispatches every matcher on DescriptorType (FeatureMatcher.cc:1508-1531, called from
#      :122,236,379,440,509,616,734,907,1036,1175,1253,1376,1481); 61 bytes = AKAZE61's MLDB, 48 = BRISK48, 20 = a short one ----
def _widen(d32, nbytes):
    """an nbytes-wide descriptor set with the neighbourhood structure of the 32-byte one: its bytes, then a rotated and inverted copy,
    cut to nbytes - distances between widened rows are about nbytes / 32 times those between the originals"""
    d32 = np.ascontiguousarray(d32, np.uint8)
    if nbytes == 32:
        return d32
    wide = np.concatenate([d32, np.roll(d32, 5, axis=1) ^ np.uint8(0x5A)], 1)
    return np.ascontiguousarray(wide[:, :nbytes])


def _widen_scene(F, Q, nbytes):
    F.descriptors = _widen(F.descriptors, nbytes)
    Q.descriptors = _widen(Q.descriptors, nbytes)
    return F, Q


def _all_searches(afv, oracle, gpu_ctx, conv, th):
    """every projection-guided search of the reference on descriptors converted by `conv` (from the extractor's 32-byte rows), against the
    oracle with the distance threshold `th`"""
    def conv_scene(F, Q):
        F.descriptors = conv(F.descriptors)
        Q.descriptors = conv(Q.descriptors)
        return F, Q

    afv.FeatureMatcher.setDescriptorDistanceThresholds(th)
    try:
        # SearchByProjection(F, local map) / (cur, last) with and without the orientation histogram
        F, Q = conv_scene(*_scene(afv, gpu_ctx, 51, 4, 15.0))
        m = afv.FeatureMatcher(0.8, True, ctx=gpu_ctx)
        got, n = m.SearchByProjection(F, Q)
        want, wn = oracle.match_projection(F, Q, th_high=th, nnratio=0.8)
        assert n == wn and np.array_equal(got, want) and wn > 100
        for ori in (False, True):
            m = afv.FeatureMatcher(0.9, ori, ctx=gpu_ctx)
            got, n = m.SearchByProjection(F, Q, last_frame=True)
            want, wn = oracle.match_projection(F, Q, th_high=th, nnratio=0.9, check_orientation=ori, last_frame=True)
            assert n == wn and np.array_equal(got, want) and wn > 100
        # stereo gate
        Fs, Qs = _stereo(afv, *conv_scene(*_scene(afv, gpu_ctx, 52, 3, 40.0)), 52, 0.5)
        m = afv.FeatureMatcher(0.85, True, ctx=gpu_ctx)
        got, n = m.SearchByProjection(Fs, Qs)
        want, wn = oracle.match_projection(Fs, Qs, th_high=th, nnratio=0.85)
        assert n == wn and np.array_equal(got, want) and wn > 50
        # Fuse (with its reprojection gate) and Fuse(Sim3)
        F, Q = conv_scene(*_scene(afv, gpu_ctx, 53, 4, 15.0))
        F.inf = np.ascontiguousarray(np.float32(0.2) / (F.sizes * F.sizes))
        m = afv.FeatureMatcher(0.6, True, ctx=gpu_ctx)
        got, n = m.Fuse(F, Q)
        want, wn = oracle.match_projection(F, Q, th_high=th, fuse=True)
        assert n == wn and np.array_equal(got, want) and wn > 50
        got, n = m.Fuse_sim3(F, Q)
        F.inf = None
        want, wn = oracle.match_projection(F, Q, th_high=th, fuse=True)
        assert n == wn and np.array_equal(got, want) and wn > 50
        # a dense cluster: the ordered phase's claim / rescan logic (the key lists run out, equal distances abound)
        s = afv.synth
        nf, nq = 60, 400
        proto = s.random_descriptors(77, 6)
        d = proto[s.lcg_states(1, nf) % 6].copy()
        d[np.arange(nf), s.lcg_states(2, nf) % 32] ^= 1
        pts = np.stack([300 + (s.lcg_states(3, nf) % 40).astype(np.float32), 200 + (s.lcg_states(4, nf) % 40).astype(np.float32)], 1)
        Fc = afv.FrameGridView(conv(d), pts, np.ones(nf, np.float32))
        qd = proto[s.lcg_states(5, nq) % 6].copy()
        qd[np.arange(nq), s.lcg_states(6, nq) % 32] ^= 2
        Qc = afv.ProjectionQueries(conv(qd), np.full(nq, 320.0), np.full(nq, 220.0), np.full(nq, 30.0), np.full(nq, 0.5), np.full(nq, 2.0))
        for mode, ratio in ((False, 0.8), (True, 0.9)):
            m = afv.FeatureMatcher(ratio, False, ctx=gpu_ctx)
            got, nn = m.SearchByProjection(Fc, Qc, last_frame=mode)
            want, wn = oracle.match_projection(Fc, Qc, th_high=th, nnratio=ratio, last_frame=mode)
            assert nn == wn and np.array_equal(got, want), (mode, ratio)
            assert wn > 10 or not mode  # (the local-map mode's ratio test rejects nearly everything inside a cluster)
        # SearchBySim3 and SearchForInitialization
        img = s.corners_frame(54)
        k1, d1 = gpu_ctx.extract(img)
        k2, d2 = gpu_ctx.extract(np.roll(img, 6, axis=1))
        d1, d2 = conv(d1), conv(d2)
        z1, _, _ = gpu_ctx.size_sigma(k1); z2, _, _ = gpu_ctx.size_sigma(k2)
        F1 = afv.FrameGridView(d1, np.stack([k1["x"], k1["y"]], 1), z1, angles=k1["angle"])
        F2 = afv.FrameGridView(d2, np.stack([k2["x"], k2["y"]], 1), z2, angles=k2["angle"])
        Q1 = afv.ProjectionQueries(d1, k1["x"] + 6, k1["y"], 12.0 * z1, z1 / np.float32(1.2), z1 * np.float32(1.2))
        Q2 = afv.ProjectionQueries(d2, k2["x"] - 6, k2["y"], 12.0 * z2, z2 / np.float32(1.2), z2 * np.float32(1.2))
        m = afv.FeatureMatcher(0.8, True, ctx=gpu_ctx)
        got, n = m.SearchBySim3(F1, Q1, F2, Q2)
        want, wn = oracle.match_sim3(F2, Q1, F1, Q2, th_high=th)
        assert n == wn and np.array_equal(got, want) and wn > 200
        prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
        n1 = len(k1)
        for window in (50.0, 100.0):  # the wider window: more than IK candidates per query, gated keys, exact rescans
            Qi = afv.ProjectionQueries(d1, prev[:, 0].copy(), prev[:, 1].copy(), np.full(n1, window, np.float32), np.zeros(n1, np.float32),
                                       np.full(n1, z1.max(), np.float32), valid=(k1["octave"] == 0).astype(np.uint8), angles=k1["angle"])
            m = afv.FeatureMatcher(0.9, True, ctx=gpu_ctx)
            got, n = m.SearchForInitialization(Qi, F2, vbPrevMatched=prev.copy())
            want, wn = oracle.match_initialization(F2, Qi, th_low=th, nnratio=0.9, check_orientation=True)
            assert n == wn and np.array_equal(got, want) and wn > 50
    finally:
        afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)


@pytest.mark.parametrize("nbytes", [61, 48, 20])
def test_projection_searches_are_descriptor_generic(afv, oracle, gpu_ctx, nbytes):
    _all_searches(afv, oracle, gpu_ctx, lambda d: _widen(d, nbytes), float(round(75.0 * nbytes / 32.0)))


# ---- float descriptors: DescriptorDistance_sift128 / _surf64 / _kaze64 / _r2d2_128 = cv::norm(a, b, NORM_L2SQR) as a float
#      (Feature_sift128.cpp:132-134; Descriptor_Distance_Type = float, Types.h:127) behind the same dispatch ----
from _float_desc import floaten as _floaten  # noqa: E402


@pytest.mark.parametrize("dim,real", [(128, False), (128, True), (64, True), (256, False)])
def test_projection_searches_with_float_descriptors(afv, oracle, gpu_ctx, dim, real):
    th = 75.0 * dim / 256.0 * (0.6 if real else 1.0)
    _all_searches(afv, oracle, gpu_ctx, lambda d: _floaten(d, dim, real), float(th))


def test_float_jobs_that_are_refused(afv, gpu_ctx):
    F, Q = _scene(afv, gpu_ctx, 61, 4, 15.0)
    Ff = _floaten(F.descriptors, 128, True)
    m = afv.FeatureMatcher(0.8, True, ctx=gpu_ctx)
    Q.descriptors = _floaten(Q.descriptors, 128, True)
    with pytest.raises(ValueError):  # binary frame, float queries
        m.SearchByProjection(F, Q)
    F.descriptors = np.ascontiguousarray(Ff[:, :126])  # not a multiple of 4
    Q.descriptors = np.ascontiguousarray(Q.descriptors[:, :126])
    with pytest.raises(afv._lib.AfvError):
        m.SearchByProjection(F, Q)


def test_float_descriptor_edge_cases(afv, oracle, gpu_ctx):
    """empty feature side, no queries, the smallest (4) and the largest (1024) row the float path takes, one feature shared by many queries"""
    s = afv.synth
    afv.FeatureMatcher.setDescriptorDistanceThresholds(1e9)
    try:
        m = afv.FeatureMatcher(0.9, False, ctx=gpu_ctx)
        for dim in (4, 1024):
            n, nq = 40, 90
            d = (s.lcg_bytes(1, n * dim).reshape(n, dim).astype(np.float32) - 128) / np.float32(64)
            q = (s.lcg_bytes(2, nq * dim).reshape(nq, dim).astype(np.float32) - 128) / np.float32(64)
            pts = np.stack([100 + (s.lcg_states(3, n) % 200).astype(np.float32), 100 + (s.lcg_states(4, n) % 200).astype(np.float32)], 1)
            F = afv.FrameGridView(d, pts, np.ones(n, np.float32), angles=np.zeros(n, np.float32))
            Q = afv.ProjectionQueries(q, 100 + (s.lcg_states(5, nq) % 200).astype(np.float32), 100 + (s.lcg_states(6, nq) % 200).astype(np.float32),
                                      np.full(nq, 60.0), np.full(nq, 0.5), np.full(nq, 2.0), angles=np.zeros(nq, np.float32))
            for last in (False, True):
                got, nn = m.SearchByProjection(F, Q, last_frame=last)
                want, wn = oracle.match_projection(F, Q, th_high=1e9, nnratio=0.9, last_frame=last)
                assert nn == wn and np.array_equal(got, want) and (wn > 5 or (dim == 1024 and not last)), (dim, last)  # (random 1024-float rows: the ratio test rejects all)
            got, nn = m.Fuse_sim3(F, Q)
            want, wn = oracle.match_projection(F, Q, th_high=1e9, fuse=True)
            assert nn == wn and np.array_equal(got, want)
            # no queries / no features
            Q0 = afv.ProjectionQueries(np.zeros((0, dim), np.float32), [], [], [], [], [])
            got, nn = m.SearchByProjection(F, Q0)
            assert nn == 0 and np.all(got == -1) and len(got) == n
            F0 = afv.FrameGridView(np.zeros((0, dim), np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.float32))
            got, nn = m.SearchByProjection(F0, Q)
            assert nn == 0 and len(got) == 0
            got, nn = m.SearchForInitialization(Q, F0)
            assert nn == 0 and np.all(got == -1) and len(got) == nq
        # one feature, many queries that all want it: the local-map flavour hands it to the first (later ones find it occupied), the
        # initialization search lets a closer later query steal it
        dim = 8
        F1 = afv.FrameGridView(np.zeros((1, dim), np.float32), np.float32([[200, 200]]), np.ones(1, np.float32), angles=np.zeros(1, np.float32))
        qd = np.zeros((5, dim), np.float32)
        qd[:, 0] = [3, 2, 2, 1, 4]
        Q5 = afv.ProjectionQueries(qd, np.full(5, 200.0), np.full(5, 200.0), np.full(5, 20.0), np.full(5, 0.5), np.full(5, 2.0), angles=np.zeros(5, np.float32))
        got, nn = m.SearchByProjection(F1, Q5)
        want, wn = oracle.match_projection(F1, Q5, th_high=1e9, nnratio=0.9)
        assert nn == wn == 1 and got.tolist() == want.tolist() == [0]
        got, nn = m.SearchForInitialization(Q5, F1)
        want, wn = oracle.match_initialization(F1, Q5, th_low=1e9, nnratio=0.9, check_orientation=False)
        assert nn == wn == 1 and got.tolist() == want.tolist() == [-1, -1, -1, 0, -1]   # 9, then 4 (steals), 4 (not < 4: gated), then 1 (steals), 16
    finally:
        afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
