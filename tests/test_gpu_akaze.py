"""-m gpu: SURVEY 8f rank 4 — AKAZE61 (config #5).  HIP kernels vs the CPU restatement in oracle/akaze.c, stage by stage and
end to end; bit-exact float parity (same expressions, one rounding per operator on both sides).  The libAKAZE fork the
reference links is absent: parity against it is unpinned (oracle/akaze.h)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def akz():
    from oracle import akaze_binding
    return akaze_binding


@pytest.fixture(autouse=True, params=[1, 0], ids=["fixed_point", "ordered_rounds"])
def suppress_engine(request, afv, monkeypatch):
    """every case runs through both engines of the ordered duplicate suppression (afv_akaze_set_suppress_engine): 1 = the fixed point
    WITHOUT its fallback, 0 = the speculative rounds of k_akz_suppress"""
    init = afv.akaze.AkazeContext.__init__

    def patched(self, *a, **k):
        init(self, *a, **k)
        self.set_suppress_engine(request.param)
    monkeypatch.setattr(afv.akaze.AkazeContext, "__init__", patched)
    return request.param


def _oracle_plan(akz, plan):
    """feed the oracle the product's own evolution plan (sizes, FED steps, Gaussian taps): the stage comparisons below then do
    not depend on host libm details; the plans themselves are compared in test_plan_matches_oracle"""
    p = akz.Plan()
    C.memmove(C.byref(p), C.byref(plan), C.sizeof(p))
    return p


def _frames(afv, w, h, seeds):
    return np.stack([afv.synth.corners_batch(s, 1, w, h)[0] for s in seeds])


def test_plan_matches_oracle(afv, akz):
    prm = afv.akaze.default_params()
    for (w, h) in ((1280, 720), (640, 480), (752, 480)):
        plan = afv.akaze.plan_for(prm, w, h)
        ref = akz.make_plan(w, h)
        assert C.sizeof(plan) == C.sizeof(ref)
        assert plan.nlevels == ref.nlevels == 8
        for i in range(plan.nlevels):
            a, b = plan.lv[i], ref.lv[i]
            assert (a.w, a.h, a.octave, a.sublevel, a.sigma_size, a.nsteps) == (b.w, b.h, b.octave, b.sublevel, b.sigma_size, b.nsteps)
            assert a.esigma == b.esigma and a.etime == b.etime
            assert list(a.tau)[:a.nsteps] == list(b.tau)[:b.nsteps]
        assert list(plan.gauss_soffset) == list(ref.gauss_soffset) and list(plan.gauss_one) == list(ref.gauss_one)
    # FED cycle lengths of the reference configuration and their defining property: the steps add up to the evolution time
    p = afv.akaze.plan_for(prm, 1280, 720)
    assert [p.lv[i].nsteps for i in range(8)] == [0, 3, 3, 4, 4, 5, 6, 7]
    for i in range(1, 8):
        assert abs(sum(list(p.lv[i].tau)[:p.lv[i].nsteps]) - (p.lv[i].etime - p.lv[i - 1].etime)) < 1e-5


@pytest.mark.parametrize("w,h", [(640, 480), (200, 120)])
def test_scale_space_stages_bit_exact(afv, akz, w, h):
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=2))
    frames = _frames(afv, w, h, (3, 4))
    plan = ctx.scale_space(frames)
    op = _oracle_plan(akz, plan)
    for f in range(2):
        levels, k0 = akz.scale_space(frames[f], op)
        assert ctx.kcontrast(f) == np.float32(k0)
        for i in range(plan.nlevels):
            assert np.array_equal(ctx.plane(f, i, afv.akaze.LSMOOTH), levels[i]["Lsmooth"]), (f, i, "Lsmooth")
            assert np.array_equal(ctx.plane(f, i, afv.akaze.LT), levels[i]["Lt"]), (f, i, "Lt")
            lx, ly, ldet = akz.hessian(levels[i]["Lsmooth"], plan.lv[i].sigma_size)
            assert np.array_equal(ctx.plane(f, i, afv.akaze.LX), lx), (f, i, "Lx")
            assert np.array_equal(ctx.plane(f, i, afv.akaze.LY), ly), (f, i, "Ly")
            assert np.array_equal(ctx.plane(f, i, afv.akaze.LDET), ldet), (f, i, "Ldet")
    ctx.close()


def test_conductivity_reciprocal_is_the_ieee_quotient():
    """k_akz_fed_gauss forms 1 / (1 + |grad|^2 / k^2) as v_rcp_f32 + one fused Newton step (csrc/akz_recip.h) instead of the division
    sequence; tools/akaze_recip_check (built by __graft_entry__.build()) compares it with 1.0f / d for every float in [1, 2^96)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "akaze_recip_check")
    if not os.path.exists(exe):  # normally built by __graft_entry__.build(); the same image has hipcc on the GPU box
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", exe + ".hip", "-o", exe],
                       check=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "recip_check ok" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("akz_recip_ge1 in 0") == 3, r.stdout


@pytest.mark.parametrize("w,h", [(700, 404), (97, 61), (80, 203), (83, 40)])
def test_fused_and_step_by_step_scale_space_agree(afv, w, h):
    """the fused level kernel (Lsmooth + conductivity + whole FED cycle in one launch) against one kernel per step, and the one-pass
    derivative / Hessian strip kernel against its two-kernel form, at sizes whose tiles are ragged on both axes - down to images smaller
    than one tile at octave 1 (41 x 20: a tile meets all four borders; the Lsmooth rows / columns just outside the image are patched in
    registers); 80 x 40 is the smallest frame the extractor accepts"""
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=1))
    frame = _frames(afv, w, h, (8,))
    plan = ctx.scale_space(frame)
    planes = (afv.akaze.LT, afv.akaze.LSMOOTH, afv.akaze.LX, afv.akaze.LY, afv.akaze.LDET)
    fused = [[ctx.plane(0, i, which) for which in planes] for i in range(plan.nlevels)]
    ctx.set_step_by_step(True)  # also: first derivatives and Hessian as two LDS-tiled kernels instead of the one-pass strip kernel
    ctx.scale_space(frame)
    for i in range(plan.nlevels):
        for k, which in enumerate(planes):
            assert np.array_equal(ctx.plane(0, i, which), fused[i][k]), (i, which)
    ctx.close()


def test_scale_space_properties_at_config5_size(afv, akz):
    """1280 x 720 (config #5): diffusion keeps the mean (zero-flux border) and obeys the maximum principle; a constant image
    stays constant with zero response"""
    ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=2))
    frames = _frames(afv, 1280, 720, (5,))
    frames = np.concatenate([frames, np.full((1, 720, 1280), 77, np.uint8)])
    plan = ctx.scale_space(frames)
    lt0 = ctx.plane(0, 0, afv.akaze.LT)
    for i in range(1, 4):
        lt = ctx.plane(0, i, afv.akaze.LT)
        assert abs(float(lt.mean(dtype=np.float64)) - float(lt0.mean(dtype=np.float64))) < 1e-5
        assert lt.min() >= lt0.min() - 1e-6 and lt.max() <= lt0.max() + 1e-6
        assert float(lt.var(dtype=np.float64)) < float(lt0.var(dtype=np.float64))
    for i in range(plan.nlevels):
        c = ctx.plane(1, i, afv.akaze.LT)
        assert np.all(np.abs(c - np.float32(77) * np.float32(1.0 / 255.0)) < 1e-6)
        assert np.all(np.abs(ctx.plane(1, i, afv.akaze.LDET)) < 1e-12)
    # one level of the big frame against the oracle (the CPU restatement needs ~0.4 s for it)
    op = _oracle_plan(akz, plan)
    levels, _ = akz.scale_space(frames[0], op)
    assert np.array_equal(ctx.plane(0, 7, afv.akaze.LT), levels[7]["Lt"])
    ctx.close()


@pytest.mark.parametrize("w,h,seeds", [(640, 480, (3, 9)), (320, 200, (4,))])
def test_detection_matches_oracle(afv, akz, w, h, seeds):
    """candidates in raster order per level, then Find_Scale_Space_Extrema's ordered suppression / upper-level filter and the
    sub-pixel refinement: same keypoints, same order, same floats"""
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=len(seeds)))
    frames = _frames(afv, w, h, seeds)
    plan = ctx.scale_space(frames)
    ctx.detect()
    op = _oracle_plan(akz, plan)
    for f in range(len(seeds)):
        levels, _ = akz.full_evolution(frames[f], op)
        for i in range(plan.nlevels):
            assert np.array_equal(ctx.candidates(f, i), akz.level_candidates(op, i, levels[i]["Ldet"])), (f, i)
        want = akz.subpixel(op, levels, akz.find_extrema(op, levels))
        got = ctx.keypoints(f)
        assert len(got) == len(want) and len(want) > 500
        for name in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(got[name], want[name]), (f, name)
    # the step-by-step scale space (separate Gaussian, one kernel per FED step, two derivative kernels) must lead to the same detection
    first = [[ctx.candidates(f, i) for i in range(plan.nlevels)] for f in range(len(seeds))]
    kps = [ctx.keypoints(f) for f in range(len(seeds))]
    ctx.set_step_by_step(True)
    ctx.scale_space(frames)
    ctx.detect()
    for f in range(len(seeds)):
        for i in range(plan.nlevels):
            assert np.array_equal(ctx.candidates(f, i), first[f][i]), (f, i)
        assert ctx.keypoints(f).tobytes() == kps[f].tobytes(), f
    ctx.close()


def test_level_pipeline_under_load(afv, akz, suppress_engine):
    """the eight levels of a frame are suppressed by eight workgroups that hand list state to each other (k_akaze_detect.hip):
    more workgroups than the chip holds at once, a chip that is busy with something else, repeated launches (the launch epoch in
    the list elements moves on) -> the same keypoints every time, and they are the oracle's"""
    import torch
    w, h, B = 640, 480, 72
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=B))
    frames = _frames(afv, w, h, tuple(range(100, 100 + B)))
    plan = ctx.scale_space(frames)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float16)
    runs = []
    for rep in range(3):
        if rep:
            with torch.cuda.stream(side):
                for _ in range(40):
                    a @ a
        ctx.detect()
        runs.append([ctx.keypoints(f) for f in range(B)])
        torch.cuda.synchronize()
    for rep in (1, 2):
        for f in range(B):
            assert runs[rep][f].tobytes() == runs[0][f].tobytes(), (rep, f)
    op = _oracle_plan(akz, plan)
    for f in (0, 35, B - 1):
        levels, _ = akz.full_evolution(frames[f], op)
        want = akz.subpixel(op, levels, akz.find_extrema(op, levels))
        got = runs[2][f]
        assert len(got) == len(want) and len(want) > 500
        for name in ("x", "y", "response", "class_id"):
            assert np.array_equal(got[name], want[name]), (f, name)
    ctx.close()


def test_detection_survives_the_epoch_wrap(afv, suppress_engine):
    """list elements of the suppression grids carry a 14-bit launch epoch; when it wraps the grids are wiped (akaze_api.hip).  More
    launches than that on one context: the keypoints never change"""
    if suppress_engine != 0:
        pytest.skip("the launch epoch belongs to the ordered-rounds engine")
    w, h = 160, 96
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=2))
    ctx.scale_space(_frames(afv, w, h, (21, 22)))
    ctx.detect()
    first = [ctx.keypoints(f).tobytes() for f in range(2)]
    assert len(first[0]) > 0
    for i in range(0x3fff + 40):
        ctx.detect()
        if i % 4096 == 0 or i > 0x3fff - 8:
            assert [ctx.keypoints(f).tobytes() for f in range(2)] == first, i
    ctx.close()


def _oracle_detect_and_compute(afv, akz, oracle, op, frame, quotas, w, h):
    """FeatureExtractor_akaze61::detectAndCompute restated with the oracle pieces: Feature_Detection, bucket by class_id,
    DistributeOctTree per level (oracle/afvo.c), Compute_Descriptors on the levels in ascending order"""
    levels, _ = akz.full_evolution(frame, op)
    kp = akz.subpixel(op, levels, akz.find_extrema(op, levels))
    chosen = []
    for lvl in range(op.nlevels):
        idx = np.nonzero(kp["class_id"] == lvl)[0]
        if len(idx) == 0:
            continue
        sel = oracle.quadtree(kp["x"][idx], kp["y"][idx], kp["response"][idx], int(quotas[lvl]), w, h, tiebreak=np.arange(len(idx)))
        chosen.append(idx[sel])
    chosen = np.concatenate(chosen) if chosen else np.zeros(0, np.int64)
    return akz.compute_descriptors(op, levels, kp[chosen])


@pytest.mark.parametrize("w,h,seeds,nfeatures", [(640, 480, (3, 9), 1000), (320, 200, (4,), 300), (640, 360, (6,), 1000)])
def test_detect_and_compute_matches_oracle(afv, akz, oracle, w, h, seeds, nfeatures):
    prm = afv.akaze.default_params(max_width=w, max_height=h, max_batch=len(seeds), nfeatures=nfeatures)
    ctx = afv.AkazeContext(prm)
    frames = _frames(afv, w, h, seeds)
    res = ctx.extract(frames)
    plan = ctx.plan
    op = _oracle_plan(akz, plan)
    quotas = ctx.quotas()
    assert quotas[:8].tolist() == oracle.quotas_extractor(nfeatures, 8, 1.1892).tolist()
    for f in range(len(seeds)):
        wk, wd = _oracle_detect_and_compute(afv, akz, oracle, op, frames[f], quotas, w, h)
        gk, gd = res[f]
        assert len(gk) == len(wk) and len(wk) > 0.5 * nfeatures
        for name in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(gk[name], wk[name]), (f, name)
        assert np.array_equal(gd, wd)
        assert np.all(np.diff(gk["class_id"]) >= 0)          # levels ascending
        assert gd.shape[1] == 61 and np.all(gd[:, 60] < 64)  # 486 bits: the top two bits of the last byte stay clear
    ctx.close()


def test_akaze_descriptors_feed_the_hamming_matcher(afv, oracle, gpu_ctx):
    """config #5 end to end: AKAZE61 features of two shifted frames through the 61-byte Hamming matcher (matchingTh 128,
    settings/akaze61_settings.yaml:11)"""
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=640, max_height=480, max_batch=2))
    img = afv.synth.corners_frame(11)
    (k1, d1), (k2, d2) = ctx.extract(np.stack([img, np.roll(img, 5, axis=1)]))
    afv.FeatureMatcher.setDescriptorDistanceThresholds(128.0)
    m = afv.FeatureMatcher(0.8, False, ctx=gpu_ctx)
    got, n = m.SearchByBoW(afv.FeatureView(d1), afv.FeatureView(d2))
    want, wn = oracle.search_by_bow_kf_kf(d1, d2, None, None, None, None, None, None, 128.0, 0.8, False)
    assert n == wn and np.array_equal(got, want) and wn > 200
    # most matches are the 5 px shift
    ok = got >= 0
    dx = k2["x"][got[ok]] - k1["x"][ok]
    assert np.mean(np.abs(dx - 5.0) < 1.5) > 0.8
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    ctx.close()


def test_hip_reproduces_golden_fixture(afv):
    """independent of the oracle binary: the committed expected outputs (tests/golden/make_golden_akaze.py)"""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "akaze61_expected.npz"))
    toy = np.load(os.path.join(gold, "toy_gray.npz"))["gray"]
    settings = afv.FeatureExtractorSettings({"FeatureExtractor.numOctaves": 8, "FeatureExtractor.scaleFactor": 1.1892,
                                             "FeatureExtractor.detectionTh": 0.0005})
    ext = afv.akaze.FeatureExtractor_akaze61(1000, settings, max_width=640, max_height=480)
    for name, img in (("toy", toy), ("corners5", afv.synth.corners_frame(5))):
        kps, desc = ext.detectAndCompute(img)
        assert np.array_equal(kps, g[name + "_kps"]) and np.array_equal(desc, g[name + "_desc"]), name
        assert ext.ctx.kcontrast(0) == g[name + "_k0"]
        assert len(ext.ctx.keypoints(0)) == int(g[name + "_ndetected"])
    k = kps[0]
    assert ext.GetKeypointOctave(k) == k["class_id"] and abs(ext.GetKeypointSize(k) - 1.1892 ** k["class_id"]) < 1e-5
    ext.close()
    afv.FeatureExtractorSettings({"FeatureExtractor.numOctaves": 8, "FeatureExtractor.scaleFactor": 1.2, "FeatureExtractor.detectionTh": 20.0})


@pytest.mark.parametrize("num_octaves,th,w,h,nfeatures", [(4, 0.001, 640, 480, 500), (8, 0.0002, 1280, 720, 1000)])
def test_other_settings_and_full_size(afv, akz, oracle, num_octaves, th, w, h, nfeatures):
    """other FeatureExtractor settings (numOctaves 4 -> omax 1 x 2 sublevels; a lower threshold) and one full config #5 frame
    (1280 x 720, two quadtree roots) through the whole plugin path"""
    prm = afv.akaze.default_params(num_octaves=num_octaves, detection_th=th, max_width=w, max_height=h, nfeatures=nfeatures)
    ctx = afv.AkazeContext(prm)
    frame = _frames(afv, w, h, (12,))[0]
    gk, gd = ctx.extract(frame)
    plan = ctx.plan
    assert plan.nlevels == (num_octaves // 4) * (num_octaves // 2)
    op = _oracle_plan(akz, plan)
    opts = akz.default_options()
    opts.omax, opts.nsublevels, opts.dthreshold = prm.omax, prm.nsublevels, prm.dthreshold
    levels, _ = akz.full_evolution(frame, op, opts)
    kp = akz.subpixel(op, levels, akz.find_extrema(op, levels, opts))
    quotas = ctx.quotas()
    assert quotas[:plan.nlevels].tolist() == oracle.quotas_extractor(nfeatures, plan.nlevels, 1.1892).tolist()
    chosen = []
    for lvl in range(op.nlevels):
        idx = np.nonzero(kp["class_id"] == lvl)[0]
        if len(idx):
            chosen.append(idx[oracle.quadtree(kp["x"][idx], kp["y"][idx], kp["response"][idx], int(quotas[lvl]), w, h, tiebreak=np.arange(len(idx)))])
    wk, wd = akz.compute_descriptors(op, levels, kp[np.concatenate(chosen)])
    assert len(gk) == len(wk) and len(wk) > 0.4 * nfeatures
    assert gk.tobytes() == wk.tobytes() and np.array_equal(gd, wd)
    ctx.close()


def test_one_frame_per_call_equals_the_batch_path(afv):
    """the plugin shape (FeatureExtractor.cpp:111-121: ONE frame per operator() call) against the batch call on the same frames"""
    w, h, seeds = 640, 480, (3, 9, 21)
    frames = _frames(afv, w, h, seeds)
    bctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=len(seeds)))
    batch = bctx.extract(frames)
    sctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=1))
    for f in range(len(seeds)):
        for rep in range(2):   # the second call of a frame reuses every buffer of the first
            k, d = sctx.extract(frames[f])
            assert k.tobytes() == batch[f][0].tobytes() and np.array_equal(d, batch[f][1]) and len(k) > 500
    # a strided view (a ROI of a larger image) and a page-locked source take the other upload branches
    import torch
    big = np.zeros((h + 8, w + 16), np.uint8)
    big[4:4 + h, 8:8 + w] = frames[0]
    roi = big[4:4 + h, 8:8 + w]
    kps = np.zeros(1064, afv.KP_DTYPE); desc = np.zeros((1064, 61), np.uint8); n = np.zeros(1, np.int32)
    rc = sctx.lib.afv_akaze_extract(sctx.handle, roi.ctypes.data, 1, w, h, big.strides[0], 0, kps.ctypes.data, desc.ctypes.data, 1064, n.ctypes.data)
    assert rc == 0 and kps[:n[0]].tobytes() == batch[0][0].tobytes() and np.array_equal(desc[:n[0]], batch[0][1])
    pinned = torch.from_numpy(frames[1].copy()).pin_memory()
    rc = sctx.lib.afv_akaze_extract(sctx.handle, pinned.data_ptr(), 1, w, h, w, 0, kps.ctypes.data, desc.ctypes.data, 1064, n.ctypes.data)
    assert rc == 0 and kps[:n[0]].tobytes() == batch[1][0].tobytes() and np.array_equal(desc[:n[0]], batch[1][1])
    bctx.close(); sctx.close()


def _noise_frame(seed, w, h, sigma):
    """band-limited noise: candidates everywhere instead of at the corners of a pattern (more, and differently clustered, neighbours)"""
    rng = np.random.default_rng(seed)
    k = np.exp(-0.5 * (np.arange(-8, 9) / sigma) ** 2)
    k /= k.sum()
    a = rng.random((h + 16, w + 16))
    a = np.apply_along_axis(lambda r: np.convolve(r, k, "valid"), 1, a)
    a = np.apply_along_axis(lambda c: np.convolve(c, k, "valid"), 0, a)
    a = (a - a.min()) / (a.max() - a.min())
    return (a * 2000.0 % 255.0).astype(np.uint8)


@pytest.mark.parametrize("sigma", [1.0, 2.0])
def test_suppression_on_dense_noise(afv, akz, sigma):
    """the ordered suppression on frames whose candidates crowd each other: both engines against the oracle's sequential loop"""
    w, h = 480, 360
    frames = np.stack([_noise_frame(5, w, h, sigma), _noise_frame(6, w, h, sigma)])
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=2))
    plan = ctx.scale_space(frames)
    ctx.detect()
    op = _oracle_plan(akz, plan)
    for f in range(2):
        levels, _ = akz.full_evolution(frames[f], op)
        want = akz.subpixel(op, levels, akz.find_extrema(op, levels))
        got = ctx.keypoints(f)
        assert len(got) == len(want) and len(want) > 1500, (len(got), len(want))
        for name in ("x", "y", "response", "class_id"):
            assert np.array_equal(got[name], want[name]), (f, name)
    ctx.close()


def test_fixed_point_gives_up_loudly_or_falls_back(afv, suppress_engine):
    """a pass bound of 1 cannot be enough: engine 1 reports it (AFV_ECAPACITY, never a wrong list), engine 2 repeats the batch through
    the ordered rounds and returns their keypoints"""
    if suppress_engine != 1:
        pytest.skip("one run is enough")
    w, h = 320, 200
    frames = _frames(afv, w, h, (4, 5))
    ctx = afv.AkazeContext(afv.akaze.default_params(max_width=w, max_height=h, max_batch=2))
    ctx.scale_space(frames)
    ctx.detect()
    want = [ctx.keypoints(f).tobytes() for f in range(2)]
    ctx.set_suppress_engine(1, pass_cap=1)
    ctx.detect()
    with pytest.raises(RuntimeError):
        ctx.keypoints(0)
    ctx.set_suppress_engine(2, pass_cap=1)
    ctx.scale_space(frames)
    ctx.detect()
    assert [ctx.keypoints(f).tobytes() for f in range(2)] == want
    # the other reason to give up: more earlier in-range candidates than a record holds (forced: at most one)
    ctx.set_suppress_engine(1)
    ctx.debug_neighbour_cap(1)
    ctx.scale_space(frames)
    ctx.detect()
    with pytest.raises(RuntimeError):
        ctx.keypoints(0)
    ctx.set_suppress_engine(2)
    ctx.scale_space(frames)
    ctx.detect()
    assert [ctx.keypoints(f).tobytes() for f in range(2)] == want
    ctx.debug_neighbour_cap(0)
    kps, desc = ctx.extract(frames)[0]
    ctx.set_suppress_engine(0)
    kps0, desc0 = ctx.extract(frames)[0]
    assert kps.tobytes() == kps0.tobytes() and np.array_equal(desc, desc0)
    ctx.close()
