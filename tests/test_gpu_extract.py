"""-m gpu: HIP extraction path vs the CPU oracle, stage by stage and end to end, through the C-ABI.

Bar: bit-exact for pixels, keypoint coordinates / octaves, integer FAST scores, Harris responses (one float expression
of the integer sums), IC angles and 256-bit descriptors.  Candidate lists leave the GPU unordered -> compared as sets.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(afv):
    s = afv.synth
    return {
        "corners1": s.corners_frame(1),
        "corners7": s.corners_frame(7),
        "noise": s.noise_frame(3),
        "constant": s.constant_frame(77),
        "ramp": s.ramp_frame(),
    }


@pytest.fixture(scope="module")
def frames(afv):
    return _frames(afv)


@pytest.fixture(scope="module", params=["auto", "batch-kernels", "small-batch-kernels"])
def gpu_ctx(afv, request):
    """every test of this module that takes the shared context runs three times: with the library's own choice (calls of <= 4 frames
    take the small-batch kernels: one-launch pyramid, retainBest + Harris in one launch, 1024-thread quadtree), with the batch kernels
    forced for every call, and with the small-batch kernels forced for every call (also the 5- and 6-frame batches)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = afv.Context(max_width=1280, max_height=720, max_batch=8)
    ctx.set_small_batch_path({"auto": 1, "batch-kernels": 0, "small-batch-kernels": 2}[request.param])
    yield ctx
    ctx.close()


def _cand_sets(ctx, oracle_trace, level):
    x, y, s, r = ctx.debug_candidates(0, level)
    got = sorted(zip(y.tolist(), x.tolist(), s.tolist(), r.view(np.uint32).tolist()))
    c = oracle_trace["cand"]
    m = c["level"] == level
    # the oracle computes Harris only for survivors of retainBest #1; the GPU computes it for every candidate
    return got, c[m], oracle_trace["keep1"][m]


@pytest.mark.parametrize("name", ["corners1", "noise", "ramp"])
def test_pyramid_bit_exact(gpu_ctx, oracle, frames, name):
    img = frames[name]
    gpu_ctx.extract(img)
    _, _, tr = oracle.orb_extract_trace(img)
    for l in range(8):
        got = gpu_ctx.debug_level(0, l)
        assert got.shape == tr["level"][l].shape
        assert np.array_equal(got, tr["level"][l]), "level %d differs" % l


@pytest.mark.parametrize("name", ["corners1", "noise"])
def test_fast_nms_harris_candidates(gpu_ctx, oracle, frames, name):
    img = frames[name]
    gpu_ctx.extract(img)
    _, _, tr = oracle.orb_extract_trace(img)
    for l in range(8):
        got, oc, keep1 = _cand_sets(gpu_ctx, tr, l)
        want_pos = sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["fast_score"].tolist()))
        assert [g[:3] for g in got] == want_pos, "FAST/NMS set differs at level %d" % l
        # Harris response (float bits) for the candidates the oracle scored
        gmap = {(g[0], g[1]): g[3] for g in got}
        for c in oc[keep1]:
            assert gmap[(int(c["y"]), int(c["x"]))] == int(np.float32(c["response"]).view(np.uint32)), (l, c)


def _border_dots(w, h, seed=11):
    """isolated bright pixels (each one a FAST corner) on a background whose contrast stays below the FAST threshold, placed on
    the first / last positions FAST can report: their 9 x 9 Harris windows leave the image by one pixel on every side and in
    every corner (BORDER_REFLECT_101), and the bottom-right ones end at the last byte of the frame"""
    rng = np.random.default_rng(seed)
    img = rng.integers(40, 60, size=(h, w), dtype=np.uint8)
    xs = list(range(3, w - 12, 16)) + [w - 4]   # a 16 px lattice (isolated corners) that starts and ends on the FAST border
    ys = list(range(3, h - 12, 16)) + [h - 4]
    for y in ys:
        for x in xs:
            img[y, x] = 250
    return img


@pytest.mark.parametrize("w,h", [(640, 480), (642, 481), (203, 150)])
def test_harris_windows_at_the_image_border(afv, oracle, w, h):
    """k_harris reads the windows from the level image itself: the reflected row / column, the left-edge byte shift and the
    last-bytes-of-the-image path are only reached by corners on the outermost FAST positions"""
    img = _border_dots(w, h)
    ctx = afv.Context(max_width=w, max_height=h, max_batch=1)
    ctx.extract(img)
    _, _, tr = oracle.orb_extract_trace(img)
    n_checked = 0
    for l in range(8):
        got, oc, keep1 = _cand_sets(ctx, tr, l)
        assert [g[:3] for g in got] == sorted(zip(oc["y"].tolist(), oc["x"].tolist(), oc["fast_score"].tolist())), l
        gmap = {(g[0], g[1]): g[3] for g in got}
        for c in oc[keep1]:
            assert gmap[(int(c["y"]), int(c["x"]))] == int(np.float32(c["response"]).view(np.uint32)), (l, c)
            n_checked += 1
    c0 = tr["cand"][(tr["cand"]["level"] == 0) & tr["keep1"]]
    on_edge = {(int(c["x"]), int(c["y"])) for c in c0}
    assert {(3, 3), (w - 4, h - 4), (3, h - 4), (w - 4, 3)} <= on_edge, "the corner positions must be candidates for this test to bite"
    assert n_checked >= 49
    ctx.close()


def test_harris_integer_sums(gpu_ctx, oracle, frames):
    """the response is a pure function of the integer sums (a,b,c): equal float bits <=> equal sums for the oracle's
    expression; check the expression itself against the oracle's on the sums of real candidates"""
    img = frames["corners7"]
    _, _, tr = oracle.orb_extract_trace(img)
    c = tr["cand"][tr["keep1"]][:200]
    for e in c:
        assert np.float32(oracle.harris_response(e["ha"], e["hb"], e["hc"])) == np.float32(e["response"])


@pytest.mark.parametrize("name", ["corners1", "corners7", "noise", "ramp", "constant"])
def test_quadtree_selection_per_level(gpu_ctx, oracle, frames, name):
    img = frames[name]
    kps, _ = gpu_ctx.extract(img)
    okps, _, tr = oracle.orb_extract_trace(img)
    o = 0
    for l in range(8):
        x, y, r = gpu_ctx.debug_selected(0, l)
        n = tr["t_counts"][l]
        assert len(x) == n, "level %d: %d vs %d selected" % (l, len(x), n)
        ok = okps[o:o + n]
        o += n
        ls = np.float32(tr["lscale"][l])
        assert np.array_equal(x.astype(np.float32) * ls, ok["x"]) and np.array_equal(y.astype(np.float32) * ls, ok["y"]), l
        assert np.array_equal(r.view(np.uint32), ok["response"].view(np.uint32)), l


@pytest.mark.parametrize("name", ["corners1", "noise", "ramp"])
def test_blur_bit_exact(gpu_ctx, oracle, frames, name):
    img = frames[name]
    gpu_ctx.extract(img)
    _, _, tr = oracle.orb_extract_trace(img)
    for l in (0, 3, 7):
        assert np.array_equal(gpu_ctx.debug_blur_level(0, l), tr["blurred"][l]), l


@pytest.mark.parametrize("name", ["corners1", "corners7", "noise", "ramp", "constant"])
def test_extract_end_to_end_bit_exact(gpu_ctx, oracle, frames, name):
    img = frames[name]
    kps, desc = gpu_ctx.extract(img)
    okps, odesc = oracle.orb_extract(img)
    assert len(kps) == len(okps)
    for field in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kps[field].view(np.uint32), okps[field].view(np.uint32)), field
    assert np.array_equal(kps["octave"], okps["octave"]) and np.all(kps["class_id"] == -1)
    assert np.array_equal(desc, odesc)


def _perturbed_keypoints(afv, k, seed):
    """keypoints a vocabulary builder might hand back: sub-pixel positions, arbitrary angles, octaves reassigned, order shuffled"""
    s = afv.synth
    n = len(k)
    r = s.lcg_states(seed, 4 * n).reshape(4, n)
    out = k.copy()
    out["x"] = k["x"] + ((r[0] % 2001).astype(np.float32) - np.float32(1000)) * np.float32(0.0013)
    out["y"] = k["y"] + ((r[1] % 2001).astype(np.float32) - np.float32(1000)) * np.float32(0.0013)
    out["angle"] = (r[2] % 360000).astype(np.float32) * np.float32(0.001)
    oct2 = (k["octave"] + (r[3] % 3).astype(np.int32) - 1).clip(0, 7)
    out["octave"] = oct2
    return out[np.argsort(r[3] % 9973, kind="stable")]


@pytest.mark.parametrize("name", ["corners1", "corners7", "noise"])
def test_detect_then_compute_is_extract(gpu_ctx, oracle, frames, name):
    """the split entry points (FeatureExtractor.h:123-124, Feature_orb32.cpp:26-53): afv_orb_detect gives extract's keypoints byte for byte,
    afv_orb_compute at them gives extract's descriptors"""
    img = frames[name]
    k, d = gpu_ctx.extract(img)
    kd = gpu_ctx.detect(img)
    assert kd.tobytes() == k.tobytes()
    assert np.array_equal(gpu_ctx.compute(img, kd), d)
    wk, wd = oracle.orb_extract(img)
    assert kd.tobytes() == wk.tobytes() and np.array_equal(oracle.orb_compute(img, wk), wd)


@pytest.mark.parametrize("name", ["corners1", "noise"])
def test_compute_at_caller_given_keypoints(gpu_ctx, oracle, frames, afv, name):
    """cv::ORB::compute semantics: every keypoint described in ITS octave at cvRound(pt / scale) with ITS angle, in the caller's order
    (keypoints moved by up to 1.3 px, re-angled, pushed one octave up or down, shuffled) - against the oracle's restatement"""
    img = frames[name]
    k, _ = gpu_ctx.extract(img)
    for seed in (5, 6):
        q = _perturbed_keypoints(afv, k, seed)
        assert np.array_equal(gpu_ctx.compute(img, q), oracle.orb_compute(img, q))
    # keypoints at the very edge of their level (centre on the last row / column and one beyond: the patch is all apron on one side)
    e = k[:6].copy()
    e["octave"] = [0, 0, 3, 3, 7, 7]
    e["x"] = [0.0, 639.6, 0.2, 639.0, 1.0, 636.0]
    e["y"] = [0.0, 479.6, 479.0, 0.3, 478.0, 2.0]
    assert np.array_equal(gpu_ctx.compute(img, e), oracle.orb_compute(img, e))
    assert len(gpu_ctx.compute(img, k[:0])) == 0


def test_compute_refuses_keypoints_it_cannot_describe(gpu_ctx, frames, afv):
    img = frames["corners1"]
    k, _ = gpu_ctx.extract(img)
    for field, value in (("octave", 8), ("octave", -1), ("x", 700.0), ("y", -3.0), ("x", np.nan)):
        bad = k[:4].copy()
        bad[field][2] = value
        with pytest.raises(afv._lib.AfvError):
            gpu_ctx.compute(img, bad)


def test_plugin_virtuals_one_by_one(afv, oracle):
    """detectKeypoints -> filterKeypoints -> computeDescriptors -> merge, as detectAndCompute composes them (Feature_orb32.cpp:11-18)"""
    ext = afv.FeatureExtractor_orb32(1000)
    img = afv.synth.corners_frame(3)
    kl = ext.filterKeypoints(ext.detectKeypoints(img))
    dl = ext.computeDescriptors(kl, img)
    k = np.concatenate([kl[l] for l in sorted(kl)])
    d = np.concatenate([dl[l] for l in sorted(dl)])
    wk, wd = oracle.orb_extract(img)
    assert k.tobytes() == wk.tobytes() and np.array_equal(d, wd)
    k2, d2 = ext.detectAndCompute(img)
    assert k2.tobytes() == k.tobytes() and np.array_equal(d2, d)


def test_extract_1280x720_two_roots(gpu_ctx, oracle, afv):
    """1280x720: DistributeOctTree starts from nIni = round(1280/720) = 2 root cells"""
    img = afv.synth.corners_frame(11, 1280, 720)
    kps, desc = gpu_ctx.extract(img)
    okps, odesc = oracle.orb_extract(img)
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)


def test_small_and_odd_sizes(afv, oracle):
    ctx = afv.Context(max_width=333, max_height=251, max_batch=1)
    for (w, h, seed) in [(333, 251, 5), (320, 240, 6), (199, 151, 8)]:
        img = afv.synth.corners_frame(seed, w, h)
        kps, desc = ctx.extract(img)
        okps, odesc = oracle.orb_extract(img)
        assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), (w, h)
    ctx.close()


def test_other_budgets(afv, oracle):
    """2000 features (the initialisation extractor, Tracking.h:239) and a non-default FAST threshold / level count"""
    img = afv.synth.corners_frame(21)
    for (nf, nl, th) in [(2000, 8, 20), (500, 5, 12), (1000, 1, 30)]:
        ctx = afv.Context(nfeatures=nf, nlevels=nl, fast_threshold=th)
        kps, desc = ctx.extract(img)
        okps, odesc = oracle.orb_extract(img, oracle.default_params(nf, nl, 1.2, th))
        assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), (nf, nl, th)
        ctx.close()


@pytest.mark.parametrize("path", [1, 0])
@pytest.mark.parametrize("sf,nl", [(1.1892, 8), (1.5, 4), (2.0, 4), (2.0, 3), (1.2, 2), (1.2, 4), (1.3, 6)])
def test_other_scale_factors_and_level_counts(afv, oracle, sf, nl, path):
    """FeatureExtractor.scaleFactor / numOctaves are settings (settings/*.yaml:6-7: 1.2, 1.1892 = 2^(1/4), 1.5, 2.0 across the
    reference's plugins): the resize coefficient tables at non-1.2 ratios (2.0 is the exact 2:1 case: weights 128 / 128), the wider
    source window of k_resize_level above a ratio of 1.37, the one-launch pyramid's plan, quotas and capacities with few levels.
    Both kernel paths (path 1 = small-batch kernels for the single frame, 0 = batch kernels), pyramid level by level and end to end."""
    img = afv.synth.corners_frame(33)
    ctx = afv.Context(nfeatures=1000, nlevels=nl, scale_factor=sf)
    ctx.set_small_batch_path(path)
    kps, desc = ctx.extract(img)
    prm = oracle.default_params(1000, nl, sf, 20)
    okps, odesc, tr = oracle.orb_extract_trace(img, prm)
    for l in range(nl):
        got = ctx.debug_level(0, l)
        assert got.shape == tr["level"][l].shape and np.array_equal(got, tr["level"][l]), (sf, nl, l)
    assert len(kps) == len(okps) and len(kps) > 300
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), (sf, nl)
    size, s2, inf = ctx.size_sigma(kps)
    osz, os2, oinf = oracle.size_sigma(kps, sf)
    assert np.array_equal(size, osz) and np.array_equal(s2, os2) and np.array_equal(inf, oinf)
    ctx.close()


def test_tiny_budgets_keep_the_unconditional_first_split(afv, oracle):
    """DistributeOctTree's first split round is unconditional (ORBextractor.cc:283-366): a level whose quota is 0..3 still
    returns up to 4 * nIni keypoints, so a 5-feature extractor yields ~26 keypoints.  Capacities must follow."""
    img = afv.synth.corners_frame(21)
    for nf in (5, 12, 40):
        ctx = afv.Context(nfeatures=nf)
        kps, desc = ctx.extract(img)
        okps, odesc = oracle.orb_extract(img, oracle.default_params(nf, 8, 1.2, 20))
        assert len(okps) > nf
        assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc), nf
        assert ctx.cap >= len(kps)
        ctx.close()
    wide = afv.synth.corners_frame(3, 1280, 400)          # nIni = 3 roots
    ctx = afv.Context(nfeatures=12, max_width=1280, max_height=400)
    kps, desc = ctx.extract(wide)
    okps, odesc = oracle.orb_extract(wide, oracle.default_params(12, 8, 1.2, 20))
    assert len(okps) > 40 and kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    ctx.close()


def test_device_call_is_ordered_with_torchs_current_stream(afv, oracle):
    """the *_device entry points run on torch's CURRENT stream (default stream included): producing the input and reading
    the outputs on that stream needs no device-wide synchronisation"""
    import torch
    ctx = afv.Context(max_batch=16)
    frames_h = afv.synth.corners_batch(70, 16)
    want = [oracle.orb_extract(f) for f in frames_h[:3]]
    for use_side in (False, True):
        stream = torch.cuda.Stream() if use_side else torch.cuda.current_stream()
        with torch.cuda.stream(stream):
            for rep in range(3):
                staging = torch.from_numpy(frames_h).cuda(non_blocking=True)
                frames = (staging.to(torch.int16) + 0).to(torch.uint8)        # produced by kernels on this stream
                kps, desc, n, status = ctx.extract_batch_device(frames)
                n_h = n.cpu()                                                  # stream-ordered copy, no device sync
                total = int(n.sum().item())
                assert int(status.cpu().item()) == 0
                assert total == int(n_h.sum()) and total > 16 * 900
                for i in range(3):
                    k = kps[i, :int(n_h[i])].cpu().numpy().view(afv.KP_DTYPE).reshape(-1)
                    assert k.tobytes() == want[i][0].tobytes(), (use_side, rep, i)
                    assert np.array_equal(desc[i, :int(n_h[i])].cpu().numpy(), want[i][1])
                del kps, desc, n, status, frames                               # let the caching allocator reuse them
    ctx.close()


def test_strided_input_and_batch(gpu_ctx, oracle, afv):
    base = np.zeros((480, 704), np.uint8)
    frames = [afv.synth.corners_frame(30 + i) for i in range(5)]
    base[:, :640] = frames[0]
    kps, desc = gpu_ctx.extract(base[:, :640])  # non-contiguous rows are copied by the wrapper; exercise C stride path too
    res = gpu_ctx.extract_batch(frames)
    for i, (k, d) in enumerate(res):
        ok, od = oracle.orb_extract(frames[i])
        assert k.tobytes() == ok.tobytes() and np.array_equal(d, od), i
    assert kps.tobytes() == res[0][0].tobytes() and np.array_equal(desc, res[0][1])


@pytest.mark.parametrize("pinned", [False, True])
def test_host_batch_pipeline_chunks(afv, oracle, pinned):
    """afv_orb_extract_batch pipelines batches >= 128 frames in chunks of 64 (H2D of chunk k+1 / compute of k / D2H of k-1 on
    separate streams); page-locked caller buffers are DMA'd in place, pageable ones go through the pinned arena.  Every frame
    must equal the single-frame path / the oracle, for both kinds of memory and for a ragged last chunk."""
    import torch
    nf = 150                                  # 64 + 64 + 22
    ctx = afv.Context(max_batch=nf)
    uniq = [afv.synth.corners_frame(400 + i) for i in range(6)] + [afv.synth.constant_frame(90)]
    ref = [oracle.orb_extract(u) for u in uniq]
    frames = torch.from_numpy(np.stack([uniq[i % 7] for i in range(nf)]))
    cap = ctx.cap
    kps = torch.zeros((nf, cap, 7), dtype=torch.float32)
    desc = torch.zeros((nf, cap, 32), dtype=torch.uint8)
    n = torch.zeros((nf,), dtype=torch.int32)
    if pinned:
        frames, kps, desc, n = frames.pin_memory(), kps.pin_memory(), desc.pin_memory(), n.pin_memory()
    for rep in range(2):                      # second call reuses the arena / events
        ctx.extract_batch_host(frames, kps, desc, n)
        kn, dn, nn = kps.numpy(), desc.numpy(), n.numpy()
        for i in range(nf):
            rk, rd = ref[i % 7]
            assert nn[i] == len(rk), (pinned, rep, i)
            assert kn[i, :nn[i]].reshape(-1).view(afv.KP_DTYPE).tobytes() == rk.tobytes() and np.array_equal(dn[i, :nn[i]], rd), (pinned, rep, i)
    # the list-of-arrays wrapper (pageable numpy frames with a row stride) goes through the same entry point
    base = np.zeros((480, 704), np.uint8)
    base[:, :640] = uniq[2]
    res = ctx.extract_batch([base[:, :640], uniq[3]])
    assert res[1][0].tobytes() == ref[3][0].tobytes() and np.array_equal(res[0][1], ref[2][1])
    ctx.close()


def test_device_resident_batch_matches_host_path(gpu_ctx, afv):
    import torch
    frames = np.stack([afv.synth.corners_frame(40 + i) for i in range(6)])
    t = torch.from_numpy(frames).cuda()
    kps, desc, n, status = gpu_ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    host = gpu_ctx.extract_batch(list(frames))
    kps = kps.cpu().numpy(); desc = desc.cpu().numpy(); n = n.cpu().numpy()
    for i in range(6):
        k = kps[i, :n[i]].reshape(-1).view(afv.KP_DTYPE)
        assert k.tobytes() == host[i][0].tobytes() and np.array_equal(desc[i, :n[i]], host[i][1]), i


def test_capacity_error_is_reported(gpu_ctx, afv):
    img = afv.synth.corners_frame(1)
    with pytest.raises(afv._lib.AfvError) as e:
        gpu_ctx.extract(img, cap=100)
    assert e.value.code == afv._lib.ECAPACITY


def test_size_sigma(gpu_ctx, oracle, afv):
    kps, _ = gpu_ctx.extract(afv.synth.corners_frame(2))
    size, s2, inf = gpu_ctx.size_sigma(kps)
    osz, os2, oinf = oracle.size_sigma(kps)
    assert np.array_equal(size, osz) and np.array_equal(s2, os2) and np.array_equal(inf, oinf)


def test_plugin_interface(afv, oracle):
    ext = afv.FeatureExtractor_orb32(1000, afv.FeatureExtractorSettings(
        {"FeatureExtractor.numOctaves": 8, "FeatureExtractor.scaleFactor": 1.2, "FeatureExtractor.detectionTh": 20.0}))
    img = afv.synth.corners_frame(3)
    kps, desc, sigma2, inf, size = ext(img)
    okps, odesc = oracle.orb_extract(img)
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    assert sigma2.shape == (len(kps), 2, 2) and np.allclose(sigma2[:, 0, 0] * inf[:, 0, 0], 1.0, rtol=1e-6)
    assert ext.GetLevels() == 8 and ext.mnFeaturesPerLevel == [217, 181, 151, 126, 105, 87, 73, 60]
    assert ext.settings.ON_automaticTuning is False
    k0, d0 = ext.detectAndCompute(np.zeros((0, 0), np.uint8))
    assert len(k0) == 0


# ---------------- committed golden fixtures (tests/golden/, independent of the oracle binary on this box) ----------------
import os
import zlib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["toy", "corners1", "corners2", "noise3"])
def test_hip_reproduces_golden_fixtures(gpu_ctx, afv, name):
    gold = np.load(os.path.join(GOLD, "orb32_expected.npz"))
    if name == "toy":
        img = np.load(os.path.join(GOLD, "toy_gray.npz"))["gray"]  # real 640x480 TUM frame of the reference's toy sequence
    elif name == "noise3":
        img = afv.synth.noise_frame(3)
    else:
        img = afv.synth.corners_frame(int(name[-1]))
    kps, desc = gpu_ctx.extract(img)
    assert kps.tobytes() == gold[name + "_kps"].tobytes()
    assert np.array_equal(desc, gold[name + "_desc"])
    for l in range(8):
        assert zlib.crc32(gpu_ctx.debug_level(0, l).tobytes()) == int(gold[name + "_level_crc"][l]), l
        x, y, s, r = gpu_ctx.debug_candidates(0, l)
        assert len(x) == int(gold[name + "_ncand"][l])
        a = np.stack([y, x, s], 1).astype(np.int32)
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        assert zlib.crc32(a.tobytes()) == int(gold[name + "_cand_crc"][l]), l
        assert len(gpu_ctx.debug_selected(0, l)[0]) == int(gold[name + "_tcounts"][l])
    for l in (0, 4, 7):
        assert zlib.crc32(gpu_ctx.debug_blur_level(0, l).tobytes()) == int(gold[name + "_blur_crc"][l]), l


def test_hip_matcher_reproduces_golden(gpu_ctx, afv):
    gold = np.load(os.path.join(GOLD, "orb32_expected.npz"))
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.6, True, ctx=gpu_ctx)
    got, n = m.SearchByBoW(afv.FeatureView(gold["shift4_desc"], angles=gold["shift4_kps"]["angle"]),
                           afv.FeatureView(gold["corners1_desc"], angles=gold["corners1_kps"]["angle"]))
    assert n == int(gold["shift4_nmatches"][0]) and np.array_equal(got, gold["shift4_match12"])


def test_hip_reproduces_toy_sequence(afv):
    """BASELINE.json configs[0] end to end on the device: the five toy frames as one batch, every frame matched against its
    predecessor through the device-resident pair matcher — compared with the committed fixtures, not with the oracle binary"""
    import torch
    gray = np.load(os.path.join(GOLD, "toy_seq_gray.npz"))["gray"]
    want = np.load(os.path.join(GOLD, "toy_seq_expected.npz"))
    ctx = afv.Context(max_batch=5)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.6, True, ctx=ctx)
    t = torch.from_numpy(gray).cuda()
    kps, desc, n, st = ctx.extract_batch_device(t)
    pa = torch.arange(1, 5, dtype=torch.int32, device="cuda")
    pb = pa - 1
    match, nm = m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True)
    torch.cuda.synchronize()
    n = n.cpu().numpy()
    kh = kps.cpu().numpy().view(np.uint8)
    dh = desc.cpu().numpy()
    for i in range(5):
        w = want["kps_%d" % i]
        assert n[i] == len(w)
        assert kh[i, :n[i]].tobytes() == w.tobytes() and np.array_equal(dh[i, :n[i]], want["desc_%d" % i])
    mh, nmh = match.cpu().numpy(), nm.cpu().numpy()
    for j, i in enumerate(range(1, 5)):
        assert nmh[j] == int(want["nmatch_%d" % i][0])
        assert np.array_equal(mh[j, :n[i]], want["match_%d" % i])
    ctx.close()


def test_batch_properties_at_full_size(gpu_ctx, afv):
    """size-independent properties on a device batch: per-level counts within [quota, quota+2], ascending octaves,
    idempotence (same frames -> same bytes), independence from batch position"""
    import torch
    ctx = afv.Context(max_batch=32)
    frames = np.stack([afv.synth.corners_frame(200 + (i % 8)) for i in range(32)])
    t = torch.from_numpy(frames).cuda()
    k1, d1, n1, st = ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    k1, d1, n1 = k1.cpu().numpy().copy(), d1.cpu().numpy().copy(), n1.cpu().numpy().copy()
    k2, d2, n2, st = ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    assert np.array_equal(n1, n2.cpu().numpy()) and int(st.item()) == 0
    quota = ctx.geometry()["quota"]
    for i in range(32):
        n = int(n1[i])
        ki = k1[i, :n].reshape(-1).view(afv.KP_DTYPE)
        assert ki.tobytes() == k2.cpu().numpy()[i, :n].tobytes() and np.array_equal(d1[i, :n], d2.cpu().numpy()[i, :n])
        assert ki.tobytes() == k1[i % 8, :n].tobytes() and np.array_equal(d1[i, :n], d1[i % 8, :n])  # position independent
        assert np.all(np.diff(ki["octave"]) >= 0)
        counts = np.bincount(ki["octave"], minlength=8)
        assert all(quota[l] <= counts[l] <= quota[l] + 2 for l in range(8))
    ctx.close()


def test_split_batch_two_streams(afv, oracle):
    """batches >= 64 frames are split over the context's two streams (odd sizes included): every frame must still be
    bit-identical to the single-frame path, and the frame-pair matcher split must equal the oracle"""
    import torch
    nf = 67
    ctx = afv.Context(max_batch=nf)
    uniq = [afv.synth.corners_frame(300 + i) for i in range(5)]
    frames = np.stack([uniq[i % 5] for i in range(nf)])
    ref = [oracle.orb_extract(u) for u in uniq]
    t = torch.from_numpy(frames).cuda()
    kps, desc, n, st = ctx.extract_batch_device(t)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.6, True, ctx=ctx)
    pa = torch.arange(nf, dtype=torch.int32, device="cuda")
    pb = (pa + nf - 1) % nf
    match, nm = m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0)
    torch.cuda.synchronize()
    assert int(st.item()) == 0
    kps = kps.cpu().numpy(); desc = desc.cpu().numpy(); n = n.cpu().numpy(); match = match.cpu().numpy(); nm = nm.cpu().numpy()
    for i in range(nf):
        rk, rd = ref[i % 5]
        assert n[i] == len(rk)
        assert kps[i, :n[i]].reshape(-1).view(afv.KP_DTYPE).tobytes() == rk.tobytes() and np.array_equal(desc[i, :n[i]], rd), i
    want = {}
    for i in range(nf):
        a, b = i % 5, ((i + nf - 1) % nf) % 5
        if (a, b) not in want:
            want[(a, b)] = oracle.search_by_bow_kf_kf(ref[a][1], ref[b][1], angle1=ref[a][0]["angle"], angle2=ref[b][0]["angle"],
                                                      th_low=75.0, nnratio=0.6, check_orientation=True)
        w, wn = want[(a, b)]
        assert nm[i] == wn and np.array_equal(match[i, :n[i]], w), i
    # the split can be switched off and gives the same bytes
    ctx.set_split_threshold(1 << 30)
    k2, d2, n2, _ = ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    assert np.array_equal(n2.cpu().numpy(), n) and np.array_equal(d2.cpu().numpy()[:, :900], desc[:, :900])
    ctx.close()


def test_large_frame_and_budget(afv, oracle):
    """1920x1080 frame, 3000 features (quota level 0 = 651 nodes), radix / quadtree capacities at scale"""
    img = afv.synth.corners_frame(77, 1920, 1080)
    ctx = afv.Context(nfeatures=3000, max_width=1920, max_height=1080, max_batch=1)
    kps, desc = ctx.extract(img)
    okps, odesc = oracle.orb_extract(img, oracle.default_params(3000, 8, 1.2, 20), cap=3200)
    assert len(kps) == len(okps) > 2900
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    ctx.close()


def test_noise_frame_worst_case_candidates(afv, oracle):
    """pure noise: FAST fires on a third of the pixels -> candidate lists far beyond the register / LDS fast paths of
    the select kernel (global fallback), retainBest with massive score ties"""
    img = afv.synth.noise_frame(9, 800, 600)
    ctx = afv.Context(max_width=800, max_height=600)
    kps, desc = ctx.extract(img)
    okps, odesc = oracle.orb_extract(img)
    assert kps.tobytes() == okps.tobytes() and np.array_equal(desc, odesc)
    n0 = len(ctx.debug_candidates(0, 0)[0])
    assert n0 > 10240  # beyond ST * CPT
    ctx.close()


def test_stage_profile_sampling_and_chunk_settings(afv):
    """afv_profile_enable(ctx, n) times every n-th batch call; afv_set_split_chunks takes 0 (automatic) or 2..64"""
    import torch
    ctx = afv.Context(max_batch=8)
    t = torch.from_numpy(afv.synth.corners_batch(3, 8)).cuda()
    ctx.profile_enable(True, every=2)
    for _ in range(4):
        ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    st = ctx.profile_read()
    assert st["fast_nms"]["launches"] == 2 and st["fast_nms"]["units"] == 16 and st["retain_harris"]["launches"] == 2
    assert st["fast_nms"]["total_ms"] > 0 and st["pyramid"]["launches"] == 2
    ctx.profile_enable(True)            # every call, figures reset
    ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    assert ctx.profile_read()["describe"]["launches"] == 1
    ctx.profile_enable(False)
    ctx.set_split_chunks(0)
    ctx.set_split_chunks(6)
    with pytest.raises(Exception):
        ctx.set_split_chunks(1)
    ctx.close()
