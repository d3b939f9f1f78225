"""CPU: pin the oracle.  The reference has no tests or golden vectors and cannot be built here, so the
oracle is "parity unpinned" for the OpenCV stages; what CAN be pinned is pinned here:
  * known-answer values derivable from the reference's text (quotas, level sizes, umax, thresholds, BRIEF table);
  * algebraic identities of each restated stage (what the formulas must satisfy whatever the data);
  * the committed golden fixtures (tests/golden/, produced by tests/golden/make_golden.py).
"""
import math
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "orb32_expected.npz"))


# ---------------- known answers from the reference text ----------------
def test_quotas_known_answers(oracle):
    # FeatureExtractor.cpp:97-108 with nfeatures 1000 / 2000 (Tracking.h:239,332); SURVEY.md §4
    assert oracle.quotas_extractor(1000).tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert oracle.quotas_extractor(2000).tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    # cv::ORB with nfeatures*10 (Feature_orb32.cpp:22)
    assert oracle.quotas_cvorb(10000).tolist() == [2172, 1810, 1508, 1257, 1047, 873, 727, 606]
    assert oracle.quotas_extractor(1000).sum() == 1000 and oracle.quotas_cvorb(10000).sum() == 10000


def test_level_geometry_known_answers(oracle):
    lw, lh, ls = oracle.level_geometry(640, 480)
    assert lw.tolist() == [640, 533, 444, 370, 309, 257, 214, 179]
    assert lh.tolist() == [480, 400, 333, 278, 231, 193, 161, 134]
    assert int((lw.astype(np.int64) * lh).sum()) == 950532  # algorithmic pixel count
    lw, lh, _ = oracle.level_geometry(1280, 720)
    assert int((lw.astype(np.int64) * lh).sum()) == 2853088
    assert ls[0] == 1.0 and np.float32(ls[1]) == np.float32(1.2)


def test_umax_and_gauss_taps(oracle):
    assert oracle.umax().tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # disc of 749 pixels
    u = oracle.umax()
    assert (2 * u[0] + 1) + 2 * int(sum(2 * u[v] + 1 for v in range(1, 16))) == 749
    taps = oracle.gauss7_taps()
    assert taps.tolist() == [18, 34, 49, 55, 49, 34, 18] and taps.sum() == 257
    # margins: none of the 256*g values is near a rounding boundary
    g = np.exp(-0.5 * (np.arange(7) - 3.0) ** 2 / 4.0)
    g = g / g.sum() * 256
    assert np.all(np.abs(g - np.round(g)) < 0.46)


def test_brief_pattern_table(oracle):
    pat = oracle.brief_pattern()
    assert pat.shape == (1024,) and pat.min() == -13 and pat.max() == 12
    assert zlib.crc32(pat.astype(np.int8).tobytes()) == 0xD1A39030  # tools/gen_brief_pattern.py
    assert pat[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]          # FeatureExtractor.h:221-222
    assert pat[-4:].tolist() == [-1, -6, 0, -11]                     # FeatureExtractor.h:476
    # every rotated tap stays inside the 37x37 patch the HIP kernel stages
    r = np.hypot(pat[0::2].astype(float), pat[1::2].astype(float))
    assert r.max() < 18.5


def test_hamming_swar_equals_popcount(oracle, afv):
    d = afv.synth.random_descriptors(9, 64)
    for i in range(0, 64, 2):
        want = int(np.unpackbits(d[i] ^ d[i + 1]).sum())
        assert oracle.hamming256(d[i], d[i + 1]) == want == oracle.hamming_bytes(d[i], d[i + 1])
    z = np.zeros(32, np.uint8); o = np.full(32, 255, np.uint8)
    assert oracle.hamming256(z, z) == 0 and oracle.hamming256(z, o) == 256


def test_rotation_histogram_quirks(oracle):
    # FeatureMatcher.cc:1587-1599: factor 1/30 with a 30-bin histogram => only bins 0..12 are ever hit
    assert oracle.rotation_bin(10.0, 5.0) == 0
    assert oracle.rotation_bin(5.0, 10.0) == 12      # -5 + 360 = 355 -> round(11.83) = 12
    assert oracle.rotation_bin(45.0, 0.0) == 2       # round(1.5) = 2: half away from zero
    assert oracle.rotation_bin(15.0, 0.0) == 1       # round(0.5) = 1
    assert max(oracle.rotation_bin(a, 0.0) for a in np.arange(0, 360, 0.5)) == 12
    # computeThreeMaxima (:1631-1668): strict >, 10 % rule
    h = [0] * 30
    h[3], h[7], h[9] = 100, 50, 9
    assert oracle.three_maxima(h) == (3, 7, -1)
    h[9] = 10
    assert oracle.three_maxima(h) == (3, 7, 9)
    h[7] = 9
    assert oracle.three_maxima(h) == (3, 9, -1)      # 10 is not < 0.1f*100: second kept, third (9) dropped
    h[9] = 9
    assert oracle.three_maxima(h) == (3, -1, -1)
    assert oracle.three_maxima([5, 5, 5] + [0] * 27) == (0, 1, 2)  # ties: earlier bin wins
    assert oracle.three_maxima([0] * 30) == (-1, -1, -1)


def test_thresholds_from_reference_yaml_values(afv):
    # settings/orb32_settings.yaml:6-11
    afv.FeatureMatcher.setDescriptorDistanceThresholds({"FeatureMatcher.matchingTh": 75.0})
    m = afv.FeatureMatcher
    assert m.TH_LOW == m.TH_HIGH == m.descDistTh_low_reloc == m.descDistTh_high_reloc == 75.0 and m.HISTO_LENGTH == 30


# ---------------- stage identities ----------------
def test_fast_atan2_matches_atan2_within_spec(oracle):
    rng = np.random.default_rng(1)
    for _ in range(2000):
        y, x = rng.integers(-200000, 200000, 2)
        if x == 0 and y == 0:
            continue
        want = math.degrees(math.atan2(y, x)) % 360.0
        got = oracle.fast_atan2(y, x)
        d = abs(got - want)
        assert min(d, 360 - d) < 0.3 and 0.0 <= got <= 360.0  # OpenCV documents ~0.3 degree accuracy
    assert oracle.fast_atan2(0, 0) == 0.0 and oracle.fast_atan2(0, 5) == 0.0
    assert abs(oracle.fast_atan2(5, 0) - 90.0) < 1e-3 and abs(oracle.fast_atan2(0, -5) - 180.0) < 1e-3


def test_sincos_is_correctly_rounded(oracle):
    """the explicit double algorithm must agree with libm's double cos/sin rounded to float"""
    bad = 0
    angles = np.concatenate([np.linspace(0, 360, 7201, dtype=np.float32), np.float32([33.3, 179.99, 359.9, 0.0, 90.0, 270.0])])
    for a in angles:
        c, s = oracle.sincos_deg(a)
        t = np.float64(np.float32(a) * np.float32(math.pi / np.float32(180.0)))
        bad += (np.float32(math.cos(t)) != np.float32(c)) + (np.float32(math.sin(t)) != np.float32(s))
    assert bad == 0
    assert oracle.sincos_deg(0.0) == (1.0, 0.0)


def test_resize_identities(oracle, afv):
    img = afv.synth.noise_frame(5, 64, 48)
    assert np.array_equal(oracle.resize_linear_exact(img, 64, 48), img)               # identity
    c = np.full((48, 64), 173, np.uint8)
    assert np.all(oracle.resize_linear_exact(c, 53, 40) == 173)                        # weights sum to 256
    r = oracle.resize_linear_exact(img, 53, 40)
    assert r.shape == (40, 53)
    # exact 2:1 decimation = 2x2 box average with round-half-up ((sum*64*64... +2^15)>>16 == (sum+2)>>2)
    half = oracle.resize_linear_exact(img, 32, 24).astype(int)
    s = img.astype(int)
    box = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, box)
    # upscale path exercises the clamped borders
    up = oracle.resize_linear_exact(img[:8, :8], 19, 13)
    assert up[0, 0] == img[0, 0] and up[-1, -1] == img[7, 7]


def test_fast_on_constructed_corner(oracle):
    img = np.full((32, 32), 100, np.uint8)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0),
            (-3, 1), (-2, 2), (-1, 3)]
    cx = cy = 15
    for k in range(9):  # exactly 9 contiguous brighter pixels, margin 50
        dx, dy = ring[k]
        img[cy + dy, cx + dx] = 150
    x, y, s = oracle.fast9_16(img, 20)
    assert (cx, cy) in set(zip(x.tolist(), y.tolist()))
    assert s[list(zip(x.tolist(), y.tolist())).index((cx, cy))] == 49  # largest t with ring > v+t is 49
    img[cy + ring[4][1], cx + ring[4][0]] = 100  # break the arc: 4 + 4 contiguous only
    x, y, s = oracle.fast9_16(img, 20)
    assert (cx, cy) not in set(zip(x.tolist(), y.tolist()))
    # threshold is strict: margin exactly 20 is not a corner
    img2 = np.full((32, 32), 100, np.uint8)
    for k in range(9):
        dx, dy = ring[k]
        img2[cy + dy, cx + dx] = 120
    assert len(oracle.fast9_16(img2, 20)[0]) == 0
    img2[img2 == 120] = 121
    x, y, s = oracle.fast9_16(img2, 20)
    assert (cx, cy) in set(zip(x.tolist(), y.tolist())) and s.max() == 20


def test_fast_nms_and_margins(oracle, afv):
    img = afv.synth.noise_frame(2, 96, 64)
    x, y, s = oracle.fast9_16(img, 20)
    assert len(x) > 50
    assert x.min() >= 3 and y.min() >= 3 and x.max() <= 96 - 4 and y.max() <= 64 - 4
    sm = oracle.fast_score_map(img, 20).astype(int)
    for xi, yi, si in zip(x, y, s):
        nb = sm[yi - 1:yi + 2, xi - 1:xi + 2].copy()
        assert nb[1, 1] == si
        nb[1, 1] = -1
        assert si > nb.max()
    # raster order (row-major)
    key = y.astype(np.int64) * 96 + x
    assert np.all(np.diff(key) > 0)


def test_harris_hand_computed(oracle):
    # vertical step edge: Ix = 4*step on the two columns next to the edge, Iy = 0
    img = np.zeros((40, 40), np.uint8)
    img[:, 20:] = 10
    a, b, c, r = oracle.harris(img, 20, 20)
    # block columns 17..23; Ix != 0 only at x = 19 and 20: (10-0)*2 + 10 + 10 = 40 -> 2 cols * 7 rows * 1600
    assert (a, b, c) == (2 * 7 * 1600, 0, 0)
    assert r == oracle.harris_response(a, b, c) and r < 0  # pure edge: det = 0, -k*trace^2
    f = np.float32
    scale = f(1) / f(4 * 7 * 255.0)
    want = (f(a) * f(b) - f(c) * f(c) - f(0.04) * (f(a) + f(b)) * (f(a) + f(b))) * (scale * scale * scale * scale)
    assert f(r) == want


def test_ic_angle_symmetries(oracle):
    yy, xx = np.mgrid[0:64, 0:64]
    right = np.clip(xx * 4, 0, 255).astype(np.uint8)   # brighter to the right -> 0 degrees
    down = np.clip(yy * 4, 0, 255).astype(np.uint8)    # brighter downwards  -> 90 degrees (image y axis)
    assert abs(oracle.ic_angle(right, 32, 32)) < 1e-3 or abs(oracle.ic_angle(right, 32, 32) - 360) < 1e-3
    assert abs(oracle.ic_angle(down, 32, 32) - 90.0) < 1e-3
    assert abs(oracle.ic_angle(right[:, ::-1].copy(), 31, 32) - 180.0) < 1e-3
    assert oracle.ic_angle(np.full((64, 64), 9, np.uint8), 32, 32) == 0.0


def test_blur_identities(oracle, afv):
    c = np.full((40, 50), 64, np.uint8)
    # 257*257*64/65536 = 64.50098 -> 65: the taps sum to 257, not 256 (as in OpenCV's 8U path)
    assert np.all(oracle.gaussian_blur7(c) == 65)
    assert np.all(oracle.gaussian_blur7(np.zeros((40, 50), np.uint8)) == 0)
    assert np.all(oracle.gaussian_blur7(np.full((40, 50), 255, np.uint8)) == 255)  # saturates
    img = afv.synth.noise_frame(4, 50, 40)
    b = oracle.gaussian_blur7(img)
    # direct 2-D evaluation with reflect-101 and round-half-even
    taps = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    S = np.zeros((40, 50), np.int64)
    for i in range(7):
        for j in range(7):
            S += taps[i] * taps[j] * p[i:i + 40, j:j + 50]
    q, r = S >> 16, S & 0xFFFF
    q = q + ((r > 32768) | ((r == 32768) & (q & 1 == 1)))
    assert np.array_equal(b, np.minimum(q, 255).astype(np.uint8))
    # symmetric under flips
    assert np.array_equal(oracle.gaussian_blur7(img[::-1, ::-1].copy()), b[::-1, ::-1])


def test_brief_rotation_consistency(oracle, afv):
    """rotating the image by 90 degrees and the angle by 90 gives the same descriptor (pattern taps land on the same
    pixels because cos/sin are exact at multiples of 90)"""
    img = afv.synth.noise_frame(8, 96, 96)
    bl = oracle.gaussian_blur7(img)
    d0 = oracle.brief_descriptor(img, bl, 48, 48, 0.0)
    img90 = np.rot90(img, k=-1).copy()   # clockwise: (x, y) -> (95 - y, x)
    bl90 = np.rot90(bl, k=-1).copy()
    d90 = oracle.brief_descriptor(img90, bl90, 95 - 48, 48, 90.0)
    assert np.array_equal(d0, d90)
    # bit i of byte b is test 8*b+i: first test compares taps (8,-3) and (9,5)
    t0 = int(bl[48 - 3, 48 + 8]) < int(bl[48 + 5, 48 + 9])
    assert (d0[0] & 1) == int(t0)


def test_retain_best_semantics(oracle):
    r = np.float32([5, 1, 5, 3, 9, 3, 3, 0])
    assert oracle.retain_best_mask(r, 8).all() and oracle.retain_best_mask(r, 100).all()
    assert oracle.retain_best_mask(r, 1).tolist() == [0, 0, 0, 0, 1, 0, 0, 0]
    assert oracle.retain_best_mask(r, 2).tolist() == [1, 0, 1, 0, 1, 0, 0, 0]   # ties at the boundary are all kept
    assert oracle.retain_best_mask(r, 4).sum() == 6
    assert oracle.retain_best_mask(r, 0).sum() == 0


def test_quadtree_properties(oracle, afv):
    s = afv.synth
    for (n, N) in [(0, 217), (1, 217), (2, 217), (217, 217), (3000, 217), (3000, 60), (20000, 434), (500, 1000)]:
        st = s.lcg_states(n + N, 3 * max(n, 1))
        px = (st[:n] % 6400).astype(np.float32) / 10.0
        py = (st[n:2 * n] % 4800).astype(np.float32) / 10.0
        resp = (st[2 * n:3 * n] % 1000).astype(np.float32)
        tb = np.arange(n, dtype=np.int64)
        sel = oracle.quadtree(px, py, resp, N, 640, 480, tb)
        assert len(set(sel.tolist())) == len(sel)
        if n == 0:
            assert len(sel) == 0
            continue
        distinct = len(set(zip(px.tolist(), py.tolist())))
        if distinct >= N + 3:
            assert N <= len(sel) <= N + 2, (n, N, len(sel))     # ORBextractor.cc:427: each split is net <= +3
        else:
            assert len(sel) <= distinct
        # order independence: shuffling the input (with raster tie-break ids) selects the same points in the same order
        perm = np.argsort(s.lcg_states(7, n), kind="stable")
        sel2 = oracle.quadtree(px[perm], py[perm], resp[perm], N, 640, 480, tb[perm])
        assert np.array_equal(perm[sel2], sel)


def test_quadtree_picks_max_response_per_cell(oracle):
    # four points in four different quadrants + a weaker twin next to each: N = 4 keeps the strong ones
    px = np.float32([100, 101, 500, 501, 100, 101, 500, 501])
    py = np.float32([100, 101, 100, 101, 400, 401, 400, 401])
    resp = np.float32([1, 2, 4, 3, 5, 6, 8, 7])
    sel = oracle.quadtree(px, py, resp, 4, 640, 480)
    assert sorted(sel.tolist()) == [1, 2, 5, 6]
    # list order: children are pushed front n1..n4 => n4 (bottom right) first
    assert sel.tolist() == [6, 5, 2, 1]


def test_extract_variants_agree(oracle, afv):
    img = afv.synth.corners_frame(5, 320, 240)
    k0, d0 = oracle.orb_extract(img, variant=0)
    k1, d1 = oracle.orb_extract(img, variant=1)
    assert k0.tobytes() == k1.tobytes() and np.array_equal(d0, d1) and len(k0) > 300


def test_compute_alone_reproduces_extract(oracle, afv):
    """afvo_orb_compute (cv::ORB::compute at given keypoints, Feature_orb32.cpp:42-53) on extract's own keypoints gives extract's
    descriptors; a keypoint moved into another octave is described THERE; off-level keypoints are refused"""
    img = afv.synth.corners_frame(2)
    k, d = oracle.orb_extract(img)
    assert np.array_equal(oracle.orb_compute(img, k), d)
    assert np.array_equal(oracle.orb_compute(img, k[::-1]), d[::-1])
    q = k[:1].copy()
    q["octave"] = 2
    tr = oracle.orb_extract_trace(img)[2]
    cx, cy = int(np.rint(q["x"][0] / np.float32(1.44))), int(np.rint(q["y"][0] / np.float32(1.44)))
    assert 0 <= cx < tr["lw"][2] and 0 <= cy < tr["lh"][2]
    assert not np.array_equal(oracle.orb_compute(img, q), d[:1])
    bad = k[:1].copy()
    bad["x"] = 2000.0
    with pytest.raises(RuntimeError):
        oracle.orb_compute(img, bad)


def test_extract_structure(oracle, afv):
    img = afv.synth.corners_frame(1)
    kps, desc, tr = oracle.orb_extract_trace(img)
    q = oracle.quotas_extractor(1000)
    assert all(q[l] <= tr["t_counts"][l] <= q[l] + 2 for l in range(8))
    assert np.all(np.diff(kps["octave"]) >= 0)                       # mergeKeypointLevels: ascending level
    assert np.all(kps["class_id"] == -1) and np.all((kps["angle"] >= 0) & (kps["angle"] <= 360))
    ls = np.float32(tr["lscale"])
    assert np.array_equal(kps["size"], np.float32(31) * ls[kps["octave"]])
    # edgeThreshold = 0: keypoints may sit 3 px from the level border (Feature_orb32.cpp:23)
    xl = np.rint(kps["x"] / ls[kps["octave"]]).astype(int)
    assert xl.min() >= 3
    # keep2 subset of keep1; retainBest thresholds respected
    assert not np.any(tr["keep2"] & ~tr["keep1"])
    assert len(oracle.orb_extract(afv.synth.constant_frame(50))[0]) == 0


def test_size_sigma(oracle):
    k = np.zeros(8, oracle.KP_DTYPE)
    k["octave"] = np.arange(8)
    size, s2, inf = oracle.size_sigma(k)
    assert size[0] == 1.0 and abs(size[7] - 1.2 ** 7) < 1e-5
    assert np.allclose(s2, size * size) and np.allclose(inf * s2, 1.0)


# ---------------- matcher restatement: hand-checkable cases ----------------
def _d(bits):
    d = np.zeros(32, np.uint8)
    for b in range(bits):
        d[b // 8] |= 1 << (b % 8)
    return d


def test_m2_greedy_order_dependence(oracle):
    """SearchByBoW(KF,KF) is greedy: an earlier row takes the column, a later row must settle for its next best"""
    d1 = np.stack([_d(0), _d(2)])
    d2 = np.stack([_d(1), _d(40)])
    m, n = oracle.search_by_bow_kf_kf(d1, d2, th_low=75.0, nnratio=0.6)
    # row 0: dists (1, 40) -> takes col 0; row 1: col 0 gone -> only col 1 (38), second = FLT_MAX -> accepted
    assert m.tolist() == [0, 1] and n == 2
    m, n = oracle.search_by_bow_kf_kf(d1[::-1].copy(), d2, th_low=75.0, nnratio=0.6)
    # reversed rows: row 0 (= _d(2)): dists (1, 38) -> col 0; row 1 (= _d(0)): only col 1 (40) -> accepted
    assert m.tolist() == [0, 1] and n == 2
    m, n = oracle.search_by_bow_kf_kf(d1, d2, th_low=30.0, nnratio=0.6)
    assert m.tolist() == [0, -1] and n == 1


def test_m3_indexing_and_inclusive_threshold(oracle):
    dkf = np.stack([_d(0), _d(100)])
    df = np.stack([_d(75), _d(101), _d(200)])
    out, n = oracle.search_by_bow_kf_frame(dkf, df, th_low=75.0, nnratio=0.9)
    # KF0: dists (75,101,200): best 75 <= 75 and 75 < 0.9*101 -> F0 := KF0.  KF1: F0 taken; dists to F1,F2 = (1,100) -> F1 := KF1
    assert out.tolist() == [0, 1, -1] and n == 2
    out, n = oracle.search_by_bow_kf_kf(dkf, df, th_low=75.0, nnratio=0.9)
    assert out.tolist() == [-1, 1] and n == 1   # strict < in the KF-KF flavour


def test_m2_validity_and_nodes(oracle):
    d1 = np.stack([_d(0), _d(0), _d(0)])
    d2 = np.stack([_d(1), _d(2), _d(3)])
    # node 5 holds {0,1} x {1,2}; node 9 holds {2} x {0}
    fv1 = [(5, [0, 1]), (9, [2])]
    fv2 = [(5, [1, 2]), (7, []), (9, [0])]
    m, n = oracle.search_by_bow_kf_kf(d1, d2, fv1, fv2, th_low=75.0, nnratio=1.0)
    # row0 in node5: dists (2,3) -> col1 (2 < 1.0*3).  row1: only col2 left, second FLT_MAX -> col2.  row2 node9 -> col0
    assert m.tolist() == [1, 2, 0] and n == 3
    m, n = oracle.search_by_bow_kf_kf(d1, d2, fv1, fv2, valid1=[1, 0, 1], valid2=[1, 1, 1], th_low=75.0, nnratio=1.0)
    assert m.tolist() == [1, -1, 0]
    m, n = oracle.search_by_bow_kf_kf(d1, d2, fv1, fv2, valid1=[1, 1, 1], valid2=[0, 0, 1], th_low=75.0, nnratio=1.0)
    assert m.tolist() == [2, -1, -1] and n == 1


def test_m4_last_minimum_wins_and_epipolar_gate(oracle):
    d1 = np.stack([_d(0)])
    d2 = np.stack([_d(10), _d(10), _d(50)])
    p1 = np.float32([[100, 100]])
    p2 = np.float32([[50, 100], [60, 100], [70, 300]])
    F = np.float32([[0, 0, 0], [0, 0, -1], [0, 1, 0]])     # line in image 2: y2 = y1
    s2 = np.float32([1, 1, 1])
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, (1e6, 0.0), th_low=75.0)
    assert m.tolist() == [1] and n == 1                     # equal distances: the LATER column wins (<=, :736)
    p2[1, 1] = 103.0                                        # 9 px^2 off the line > 3.84 * sigma^2
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, (1e6, 0.0), th_low=75.0)
    assert m.tolist() == [0]
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, (50.0, 100.0), th_low=75.0)
    assert m.tolist() == [-1] and n == 0                    # col 0 sits on the epipole (:741-748), col 1 off the line
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, (1e6, 0.0), has_mp1=[1], th_low=75.0)
    assert m.tolist() == [-1]


def test_m4_stereo_branches(oracle):
    """FeatureMatcher.cc:705-709, :727-731, :741: bOnlyStereo skips features without a right match on both sides; the epipole-distance
    test only applies to mono-mono candidates"""
    d1 = np.stack([_d(0), _d(0)])
    d2 = np.stack([_d(10), _d(12)])
    p1 = np.float32([[100, 100], [200, 100]])
    p2 = np.float32([[50, 100], [60, 100]])                 # both on the line y2 = y1, both close to the epipole (55, 100)
    F = np.float32([[0, 0, 0], [0, 0, -1], [0, 1, 0]])
    s2 = np.float32([1, 1])
    ep = (55.0, 100.0)                                      # 25 px^2 < 100 * sqrt(1): mono-mono candidates are skipped
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, ep, th_low=75.0)
    assert m.tolist() == [-1, -1] and n == 0
    # a right-image coordinate on EITHER side switches the epipole test off for that candidate
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, ep, th_low=75.0, u_right1=[90.0, -1.0])
    assert m.tolist() == [0, -1] and n == 1                 # row 0 is stereo: takes the closer descriptor (column 0)
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, ep, th_low=75.0, u_right2=[-1.0, 40.0])
    assert m.tolist() == [1, 1] and n == 2                  # only column 1 escapes the epipole test
    # mvuRight == 0 counts as stereo here (>= 0, :705) ...
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, ep, th_low=75.0, u_right1=[0.0, -1.0])
    assert m.tolist() == [0, -1]
    # bOnlyStereo: mono features are skipped outright, on both sides
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, (1e6, 0.0), th_low=75.0, u_right1=[90.0, -1.0], u_right2=[-1.0, 40.0],
                                           only_stereo=True)
    assert m.tolist() == [1, -1] and n == 1
    m, n = oracle.search_for_triangulation(d1, d2, p1, p2, s2, F, (1e6, 0.0), th_low=75.0, only_stereo=True)
    assert m.tolist() == [-1, -1] and n == 0                # monocular keyframes: nothing qualifies


def test_l2sqr(oracle):
    a = np.float32([1, 2, 3, 4, 5]); b = np.float32([0, 0, 0, 0, 0])
    assert oracle.l2sqr(a, b) == 55.0
    rng = np.random.default_rng(3)
    x = rng.random(128, dtype=np.float32); y = rng.random(128, dtype=np.float32)
    assert abs(oracle.l2sqr(x, y) - float(((x.astype(np.float64) - y) ** 2).sum())) < 1e-4


# ---------------- committed golden fixtures ----------------

def test_float_vocabulary_descent_by_hand(oracle, afv):
    """afvo_bow_transform_f32: squared differences in float, summed in double, first minimum wins - against an independent numpy descent
    (checked wherever the best child leads the runner-up by more than the summation order could matter), and on a tie"""
    voc = afv.Vocabulary.random_float(3, k=4, L=2, dim=64)
    d = afv.synth.lcg_bytes(5, 50 * 64).reshape(50, 64).astype(np.float32) ** 2
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    leaf, nid = oracle.bow_transform(voc, d, 1)
    checked = 0
    for i in range(50):
        node, level, clear = 0, 0, True
        while voc.child_ptr[node + 1] > voc.child_ptr[node]:
            ch = voc.child_idx[voc.child_ptr[node]:voc.child_ptr[node + 1]]
            dist = ((d[i][None, :].astype(np.float64) - voc.node_desc[ch].astype(np.float64)) ** 2).sum(axis=1)
            srt = np.sort(dist)
            clear = clear and (srt[1] - srt[0] > 1e-6)
            node = int(ch[np.argmin(dist)])
            level += 1
            if level == voc.L - 1 and clear:
                assert nid[i] == node
        if clear:
            assert leaf[i] == node
            checked += 1
    assert checked > 40
    voc.node_desc[1:5] = voc.node_desc[2]   # all level-1 children identical: the first one wins
    leaf, nid = oracle.bow_transform(voc, d, 1)
    assert np.all(nid == 1)


@pytest.mark.parametrize("name", ["toy", "corners1", "corners2", "noise3"])
def test_oracle_reproduces_golden(oracle, afv, gold, name):
    if name == "toy":
        img = np.load(os.path.join(GOLD, "toy_gray.npz"))["gray"]
    elif name == "noise3":
        img = afv.synth.noise_frame(3)
    else:
        img = afv.synth.corners_frame(int(name[-1]))
    kps, desc, tr = oracle.orb_extract_trace(img)
    assert kps.tobytes() == gold[name + "_kps"].tobytes()
    assert np.array_equal(desc, gold[name + "_desc"])
    assert [zlib.crc32(l.tobytes()) for l in tr["level"]] == gold[name + "_level_crc"].tolist()
    assert [zlib.crc32(l.tobytes()) for l in tr["blurred"]] == gold[name + "_blur_crc"].tolist()
    assert tr["t_counts"] == gold[name + "_tcounts"].tolist()


def test_golden_matcher_vector(oracle, afv, gold):
    m, n = oracle.search_by_bow_kf_kf(gold["shift4_desc"], gold["corners1_desc"], angle1=gold["shift4_kps"]["angle"],
                                      angle2=gold["corners1_kps"]["angle"], th_low=75.0, nnratio=0.6, check_orientation=True)
    assert n == int(gold["shift4_nmatches"][0]) and np.array_equal(m, gold["shift4_match12"])
    # a 4 px shift must mostly match keypoints 4 px apart
    ok = m >= 0
    dx = gold["shift4_kps"]["x"][ok] - gold["corners1_kps"]["x"][m[ok]]
    assert np.mean(np.abs(dx - 4.0 * 1.0) < 8.0) > 0.9 and n > 300


def test_oracle_reproduces_toy_sequence(oracle):
    """BASELINE.json configs[0]: all five frames of the reference's toy sequence, and every frame matched against its predecessor"""
    gray = np.load(os.path.join(GOLD, "toy_seq_gray.npz"))["gray"]
    want = np.load(os.path.join(GOLD, "toy_seq_expected.npz"))
    assert gray.shape == (5, 480, 640)
    prev = None
    for i in range(5):
        kps, desc = oracle.orb_extract(gray[i])
        assert kps.tobytes() == want["kps_%d" % i].tobytes() and np.array_equal(desc, want["desc_%d" % i])
        if prev is not None:
            m, n = oracle.search_by_bow_kf_kf(desc, prev[1], angle1=kps["angle"], angle2=prev[0]["angle"], th_low=75.0, nnratio=0.6,
                                              check_orientation=True)
            assert n == int(want["nmatch_%d" % i][0]) > 200 and np.array_equal(m, want["match_%d" % i])
        prev = (kps, desc)


# ---------------- projection-guided matching core ----------------
def test_projection_hand_checked(oracle, afv):
    """three features in one grid neighbourhood, two map points competing for the closest one"""
    desc = np.stack([_d(0), _d(8), _d(60)])
    pts = np.float32([[100, 100], [104, 100], [300, 300]])
    F = afv.FrameGridView(desc, pts, np.float32([1.0, 1.0, 1.0]))
    q = np.stack([_d(1), _d(2)])
    Q = afv.ProjectionQueries(q, [101, 101], [100, 100], [10, 10], [0.8, 0.8], [1.3, 1.3])
    # q0: dists (1, 7) -> same size band: 1 <= 0.8*7 -> takes feature 0.  q1: feature 0 occupied -> only feature 1 (6) -> takes it
    a, n = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    assert a.tolist() == [0, 1, -1] and n == 2
    # ratio test rejects when best and second are close AND in the same scale band ...
    a, n = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.1)
    assert a.tolist() == [-1, -1, -1] and n == 0  # q0: 1 > 0.1*7 rejected; q1 sees (2, 6): 2 > 0.1*6 rejected as well
    F2 = afv.FrameGridView(desc, pts, np.float32([1.0, 1.25, 1.0]))
    a2, n2 = oracle.match_projection(F2, Q, th_high=75.0, nnratio=0.1)
    assert a2.tolist()[0] == 0  # second best lies in another scale band -> no ratio test (FeatureMatcher.cc:144)
    # window and size band filters
    Q3 = afv.ProjectionQueries(q, [101, 101], [100, 100], [2.5, 10], [0.8, 1.1], [1.3, 1.3])
    a3, n3 = oracle.match_projection(F, Q3, th_high=75.0, nnratio=0.8)
    assert a3.tolist() == [0, -1, -1]  # q0 sees only feature 0 (|dx| < 2.5); q1's size band excludes every feature
    # last-frame flavour: best only, inclusive threshold
    a4, n4 = oracle.match_projection(F, Q, th_high=1.0, nnratio=0.8, last_frame=True)
    assert a4.tolist() == [0, -1, -1] and n4 == 1


def test_projection_stereo_gates_hand_checked(oracle, afv):
    """FeatureMatcher.cc:114-119 / :1367-1372 (a feature with mvuRight > 0 is skipped when the query's projected right coordinate lies
    further away than the gate) and :880-894 (Fuse: 3-dof reprojection gate 7.8 for stereo keypoints, 2-dof 5.99 otherwise)"""
    desc = np.stack([_d(0), _d(8), _d(60)])
    pts = np.float32([[100, 100], [104, 100], [300, 300]])
    q = np.stack([_d(1), _d(2)])
    # feature 0 has a right match at 80, feature 1 is monocular (-1): only feature 0 can be gated out
    F = afv.FrameGridView(desc, pts, np.float32([1.0, 1.0, 1.0]), u_right=[80.0, -1.0, -1.0])
    Q = afv.ProjectionQueries(q, [101, 101], [100, 100], [10, 10], [0.8, 0.8], [1.3, 1.3], ur=[81.0, 95.0], er_max=[2.0, 2.0])
    a, n = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    # q0: |81 - 80| = 1 <= 2: as in the mono case it takes feature 0; q1: feature 0 is occupied anyway -> feature 1
    assert a.tolist() == [0, 1, -1] and n == 2
    Q.ur = np.float32([90.0, 95.0])  # q0 now 10 px off on the right image: feature 0 is skipped, q0 falls back to feature 1 (7 bits)
    a, n = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    assert a.tolist() == [-1, 0, -1] and n == 1  # ... and q1 (15 px off feature 0, feature 1 taken) finds nothing
    F.u_right = np.float32([0.0, -1.0, -1.0])    # mvuRight == 0 is NOT stereo for these two searches (> 0, :114 / :1367)
    a, n = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8)
    assert a.tolist() == [0, 1, -1]
    a, n = oracle.match_projection(F, Q, th_high=75.0, nnratio=0.8, last_frame=True)
    assert a.tolist() == [0, 1, -1]
    # Fuse: e2 * inf with inf = 1: mono gate 5.99, stereo gate 7.8 on ex^2 + ey^2 + er^2
    F = afv.FrameGridView(desc[:1], pts[:1], np.float32([1.0]), inf=[1.0], u_right=[80.0])
    Qf = afv.ProjectionQueries(q[:1], [102.0], [101.0], [10], [0.8], [1.3], ur=[81.5])
    b, n = oracle.match_projection(F, Qf, th_high=75.0, fuse=True)
    assert b.tolist() == [0]                     # 4 + 1 + 2.25 = 7.25 <= 7.8
    Qf.ur = np.float32([82.0])
    b, n = oracle.match_projection(F, Qf, th_high=75.0, fuse=True)
    assert b.tolist() == [-1]                    # 4 + 1 + 4 = 9 > 7.8
    F.u_right = np.float32([-1.0])
    b, n = oracle.match_projection(F, Qf, th_high=75.0, fuse=True)
    assert b.tolist() == [0]                     # mono keypoint: 4 + 1 = 5 <= 5.99
    Qf.u = np.float32([102.5])
    b, n = oracle.match_projection(F, Qf, th_high=75.0, fuse=True)
    assert b.tolist() == [-1]                    # 6.25 + 1 > 5.99
    F.u_right = np.float32([0.0])                # mvuRight == 0 IS stereo in Fuse (>= 0, :880): er = 82 - 0
    b, n = oracle.match_projection(F, Qf, th_high=75.0, fuse=True)
    assert b.tolist() == [-1]


def test_bow_transform_hand_tree(afv, oracle):
    """k=2, L=2 tree with hand-picked node descriptors: descent by Hamming argmin, first minimum wins, node_at_level at
    depth L - levelsup (DBoW2 transform semantics restated; parity unpinned)."""
    parent = [0, 0, 0, 1, 1, 2, 2]
    desc = np.zeros((7, 32), np.uint8)
    desc[1] = 0x00; desc[2] = 0xFF
    desc[3] = 0x00; desc[4] = 0x0F; desc[5] = 0xFF; desc[6] = 0xF0
    leaf = [False, False, False, True, True, True, True]
    voc = afv.Vocabulary(2, 2, parent, desc, np.ones(7), leaf)
    assert voc.child_ptr.tolist() == [0, 2, 4, 6, 6, 6, 6, 6] and voc.child_idx.tolist() == [1, 2, 3, 4, 5, 6]
    assert voc.word_id.tolist() == [-1, -1, -1, 0, 1, 2, 3]
    q = np.zeros((4, 32), np.uint8)
    q[0] = 0x01        # near node 1, then node 3
    q[1] = 0x0F        # equidistant from 1 and 2 (128 each) -> first child (1); then node 4 exactly
    q[2] = 0xFE        # near 2, then 5
    q[3] = 0xF0        # tie 1/2 -> 1; children 3 (d=128) / 4 (d=256) -> 3
    lf, nid = oracle.bow_transform(voc, q, levelsup=1)
    assert lf.tolist() == [3, 4, 5, 3] and nid.tolist() == [1, 1, 2, 1]
    lf, nid = oracle.bow_transform(voc, q, levelsup=4)
    assert lf.tolist() == [3, 4, 5, 3] and nid.tolist() == [0, 0, 0, 0]
    lf, nid = oracle.bow_transform(voc, q, levelsup=0)
    assert nid.tolist() == lf.tolist()


def test_oracle_binding_mirrors_match_afvo_h(tmp_path):
    """the ctypes structures of oracle/binding.py against the layouts gcc gives oracle/afvo.h (sizes and field offsets): the checker's own
    plumbing, checked the same way as the product's (tests/test_cabi.py)"""
    import ctypes as C
    import os
    import subprocess
    import oracle.binding as ob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pairs = [("afvo_keypoint", ob.Keypoint), ("afvo_params", ob.Params), ("afvo_trace", ob.Trace), ("afvo_bow_job", ob.BowJob),
             ("afvo_tri_job", ob.TriJob), ("afvo_proj_job", ob.ProjJob), ("afvo_l2_job", ob.L2Job), ("afvo_vocab", ob.Vocab)]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "afvo.h"', 'int main(void) {']
    for cname, st in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=gnu11", "-I", os.path.join(root, "oracle"), str(src), "-o", str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.splitlines())
    for cname, st in pairs:
        assert int(got[cname]) == C.sizeof(st), (cname, got[cname], C.sizeof(st))


def test_distinctive_descriptor_by_hand(oracle):
    """MapPoint.cc:279-349: three descriptors at distances d01 = 8, d02 = 24, d12 = 16: rows sorted (0, 8, 24), (0, 8, 16), (0, 16, 24),
    medians (entry 0.5 * 2 = 1) 8, 8, 16 -> row 0 (the first of the tie); four: entry 0.5 * 3 = 1 (the LOWER middle)"""
    d = np.zeros((3, 32), np.uint8)
    d[1, 0] = 0xFF
    d[2, 0] = 0xFF; d[2, 1] = 0xFF; d[2, 2] = 0xFF
    assert oracle.distinctive_descriptor(d) == (0, 8)
    assert oracle.distinctive_descriptor(d[::-1].copy()) == (1, 8)      # rows (0,16,24), (0,8,16), (0,8,24): medians 16, 8, 8
    four = np.zeros((4, 32), np.uint8)
    four[1, 0] = 0x01; four[2, 0] = 0x03; four[3, 0] = 0xFF             # distances from row 0: 1, 2, 8
    idx, med = oracle.distinctive_descriptor(four)
    assert med == 1 and idx == 0
    assert oracle.distinctive_descriptor(np.zeros((0, 32), np.uint8))[0] == -1
    assert oracle.distinctive_descriptor(np.zeros((1, 32), np.uint8)) == (0, 0)


def test_float_descriptors_by_hand(oracle, afv):
    """DescriptorDistance for float descriptors = cv::norm(a, b, NORM_L2SQR) as a float (Feature_sift128.cpp:132-134) inside the control flow
    of the searches: numbers small enough to do in the head"""
    # SearchByBoW(KF, KF), one node: row 0 sees (3.81, 0.01, 4) -> column 1; row 1 sees column 0 at 0.01 and column 2 at 0 -> column 2
    d1 = np.float32([[0, 0, 0, 0], [1, 1, 1, 1]])
    d2 = np.float32([[1, 1, 1, 0.9], [0, 0, 0, 0.1], [1, 1, 1, 1]])
    m, n = oracle.search_by_bow_kf_kf(d1, d2, th_low=0.5, nnratio=0.8)
    assert m.tolist() == [1, 2] and n == 2
    m, n = oracle.search_by_bow_kf_kf(d1, d2, th_low=0.005, nnratio=0.8)   # best < th_low, strictly
    assert m.tolist() == [-1, 2] and n == 1
    # the first of two equal distances wins, the ratio test then fails (0 < 0.8 * 0 is false)
    m, n = oracle.search_by_bow_kf_kf(np.float32([[1, 1, 1, 1]]), np.float32([[1, 1, 1, 1], [1, 1, 1, 1]]), th_low=0.5, nnratio=0.8)
    assert m.tolist() == [-1] and n == 0
    # projection search, last-frame flavour (best only): q0 sees (0.81, 0.0100000.) -> feature 1; q1 finds feature 1 taken -> feature 0
    F = afv.FrameGridView(np.float32([[0, 0, 0, 0], [1, 0, 0, 0], [5, 5, 5, 5]]), np.float32([[100, 100], [102, 100], [300, 300]]),
                          np.float32([1, 1, 1]))
    Q = afv.ProjectionQueries(np.float32([[0.9, 0, 0, 0], [0, 0, 0, 0]]), [101, 101], [100, 100], [10, 10], [0.8, 0.8], [1.3, 1.3])
    a, n = oracle.match_projection(F, Q, th_high=0.5, nnratio=0.8, last_frame=True)
    assert a.tolist() == [1, 0, -1] and n == 2
    a, n = oracle.match_projection(F, Q, th_high=0.5, nnratio=0.8)          # local-map flavour: 0.01 <= 0.8 * 0.81, then no second candidate
    assert a.tolist() == [1, 0, -1] and n == 2
    a, n = oracle.match_projection(F, Q, th_high=0.5, nnratio=0.01)         # 0.01 > 0.01 * 0.81 in the same scale band: q0 rejected, q1 sees (0, 1): 0 <= 0.01
    assert a.tolist() == [1, -1, -1] and n == 1
    assert oracle.l2sqr(np.float32([0.9, 0, 0, 0]), np.float32([1, 0, 0, 0])) == np.float32(np.float64(np.float32(0.9) - np.float32(1)) ** 2)


def test_distinctive_float_descriptor_by_hand(oracle):
    """MapPoint::ComputeDistinctiveDescriptors with float distances: rows 0, 1, 2 on a line at 0, 1, 3 (one coordinate): the rows of squared
    distances are (0 1 9), (0 1 4), (0 4 9) with medians 1, 1, 4 - the FIRST least median wins (strict <, MapPoint.cc:333)"""
    d = np.zeros((3, 4), np.float32)
    d[:, 0] = [0, 1, 3]
    i, m = oracle.distinctive_descriptor(d)
    assert i == 0 and m == np.float32(1)
    d[:, 0] = [0, 2, 3]            # rows (0 4 9), (0 1 4), (0 1 9): medians 4, 1, 1 -> row 1
    i, m = oracle.distinctive_descriptor(d)
    assert i == 1 and m == np.float32(1)
    assert oracle.distinctive_descriptor(np.zeros((0, 4), np.float32))[0] == -1


def test_golden_float_matcher_vectors(oracle, afv, gold):
    """tests/golden/float_matchers_expected.npz (made by tests/golden/make_golden_float.py): the float-descriptor paths of the matchers on the
    frames of the ORB32 fixture - the oracle must still answer what it answered when the fixture was committed"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_float", os.path.join(GOLD, "make_golden_float.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    fg = np.load(os.path.join(GOLD, "float_matchers_expected.npz"))
    k1, ks, f1, fs, z1, zs = mk.scene(gold)
    assert [int(np.frombuffer(f.tobytes(), np.uint8).astype(np.uint64).sum()) for f in (f1, fs)] == fg["row_crc"].tolist()  # the inputs are the same
    F, Q, Qi, sets = mk.views(k1, ks, f1, fs, z1, zs)
    m, n = oracle.search_by_bow_kf_kf(fs, f1, angle1=ks["angle"], angle2=k1["angle"], th_low=mk.TH, nnratio=0.8, check_orientation=True)
    assert n == int(fg["bow_n"][0]) and np.array_equal(m, fg["bow_match12"]) and n > 300
    a, n = oracle.match_projection(F, Q, th_high=mk.TH, nnratio=0.9, check_orientation=True, last_frame=True)
    assert n == int(fg["proj_n"][0]) and np.array_equal(a, fg["proj_assign"]) and n > 300
    i12, n = oracle.match_initialization(F, Qi, th_low=mk.TH, nnratio=0.9, check_orientation=True)
    assert n == int(fg["init_n"][0]) and np.array_equal(i12, fg["init_match12"]) and n > 100
    best = [oracle.distinctive_descriptor(s) for s in sets]
    assert [b[0] for b in best] == fg["dist_best"].tolist() and np.array_equal(np.float32([b[1] for b in best]), fg["dist_median"])
