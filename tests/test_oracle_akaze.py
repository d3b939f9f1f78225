"""CPU: the AKAZE61 restatement (oracle/akaze.c) — known-answer / property tests and the committed golden fixture.
PARITY UNPINNED against the reference's libAKAZE fork (absent); see oracle/akaze.h."""
import math
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def akz():
    from oracle import akaze_binding
    return akaze_binding


def test_plan_kats(akz):
    """Feature_akaze61.cpp:9-15 with settings/akaze61_settings.yaml: omax 2, nsublevels 4 -> 8 evolution levels"""
    p = akz.make_plan(1280, 720)
    assert p.nlevels == 8
    assert [(p.lv[i].w, p.lv[i].h, p.lv[i].octave, p.lv[i].sublevel) for i in range(8)] == \
        [(1280, 720, 0, j) for j in range(4)] + [(640, 360, 1, j) for j in range(4)]
    assert [p.lv[i].sigma_size for i in range(8)] == [2, 3, 3, 4, 2, 3, 3, 4]
    es = [1.6 * 2 ** (i / 4) for i in range(8)]
    assert np.allclose([p.lv[i].esigma for i in range(8)], es, rtol=1e-6)
    assert [p.lv[i].nsteps for i in range(8)] == [0, 3, 3, 4, 4, 5, 6, 7]
    for i in range(1, 8):   # a FED cycle covers exactly the evolution time between two levels; no step exceeds the cycle bound
        tau = list(p.lv[i].tau)[:p.lv[i].nsteps]
        assert abs(sum(tau) - (p.lv[i].etime - p.lv[i - 1].etime)) < 1e-5
        assert min(tau) > 0
    assert (p.ksize_soffset, p.ksize_one) == (9, 5)
    assert abs(sum(list(p.gauss_soffset)[:9]) - 1) < 1e-6 and abs(sum(list(p.gauss_one)[:5]) - 1) < 1e-6
    assert list(p.gauss_one)[:5] == list(p.gauss_one)[:5][::-1]
    # a third octave would be 80 x 40 or larger only: 300 x 150 keeps 2 octaves with omax 2
    assert akz.make_plan(300, 150).nlevels == 8


def test_filters_on_simple_images(akz):
    p = akz.make_plan(96, 64)
    const = np.full((64, 96), 0.25, np.float32)
    taps = np.array(list(p.gauss_one)[:5], np.float32)
    assert np.allclose(akz.gauss(const, taps), 0.25, atol=1e-7)
    ramp = np.tile(np.arange(96, dtype=np.float32) / 96, (64, 1))
    g = akz.gauss(ramp, taps)
    assert np.allclose(g[:, 4:-4], ramp[:, 4:-4], atol=1e-6)            # a linear ramp is a fixed point away from the border
    assert np.array_equal(akz.halfsample(ramp)[:, 0], (ramp[0, 0] + ramp[0, 1]) * 0.5 * np.ones(32, np.float32))
    # zero gradient -> conductivity 1; the diffusion step keeps a constant image and conserves the mean of any image
    assert np.all(akz.flow_g2(const, 0.1) == 1.0)
    rng = np.random.default_rng(3)
    img = rng.random((64, 96), dtype=np.float32)
    flow = akz.flow_g2(akz.gauss(img, taps), 0.5)
    assert np.all((flow > 0) & (flow <= 1))
    out = akz.nld_step(img, flow, 0.2)
    assert abs(float(out.mean(dtype=np.float64)) - float(img.mean(dtype=np.float64))) < 1e-6
    assert out.min() >= img.min() - 1e-6 and out.max() <= img.max() + 1e-6
    assert np.array_equal(akz.nld_step(const, np.ones_like(const), 0.25), const)
    # Hessian of a quadratic bowl: Lxx = Lyy = 2a, Lxy = 0 in the interior (scale-normalised derivatives)
    yy, xx = np.mgrid[0:64, 0:96].astype(np.float32)
    bowl = 1e-3 * ((xx - 48) ** 2 + (yy - 32) ** 2)
    lx, ly, ldet = akz.hessian(bowl, 2)
    assert np.allclose(ldet[20:44, 20:76], (2e-3 * 4) ** 2, rtol=1e-3)   # (sigma_size^2 * 2a)^2
    assert np.allclose(lx[32, 20:76], 2 * 2e-3 * (xx[32, 20:76] - 48), atol=1e-4)


def test_get_angle_quadrants(akz):
    for x, y in ((1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (0.3, -2.5), (-7, 0.01)):
        want = math.atan2(y, x) % (2 * math.pi)
        assert abs(akz.get_angle(x, y) - want) < 5e-7, (x, y)
    assert akz.get_angle(1.0, -0.0) == 0.0


def test_detection_hand_cases(akz):
    """two equal-level maxima closer than the keypoint radius: the later, weaker one is dropped; a later, stronger one takes
    the earlier one's slot (Find_Scale_Space_Extrema's compare-with-same-scale rule)"""
    opts = akz.default_options()
    plan = akz.make_plan(200, 120, opts)
    def levels_with(points):
        lv = []
        for i in range(plan.nlevels):
            L = plan.lv[i]
            d = np.zeros((L.h, L.w), np.float32)
            lv.append(dict(Lt=d.copy(), Lsmooth=d.copy(), Lx=d.copy(), Ly=d.copy(), Ldet=d))
        for (lvl, x, y, v) in points:
            lv[lvl]["Ldet"][y, x] = v
        return lv
    k = akz.find_extrema(plan, levels_with([(0, 80, 60, 0.01), (0, 82, 60, 0.005)]))
    assert len(k) == 1 and (k["x"][0], k["y"][0]) == (80, 60)
    k = akz.find_extrema(plan, levels_with([(0, 80, 60, 0.005), (0, 82, 60, 0.01), (0, 120, 60, 0.02)]))
    assert len(k) == 2 and (k["x"][0], k["y"][0]) == (82, 60) and k["x"][1] == 120   # the stronger point took slot 0
    k = akz.find_extrema(plan, levels_with([(0, 80, 60, 0.005), (0, 84, 60, 0.01)]))       # 4 px apart > size 2.4: both stay
    assert len(k) == 2
    # upper-level filter: a level-0 point under a stronger level-1 point is removed; below the threshold nothing is found
    k = akz.find_extrema(plan, levels_with([(0, 80, 60, 0.005), (1, 81, 60, 0.01)]))
    assert len(k) == 1 and k["class_id"][0] == 1
    assert len(akz.find_extrema(plan, levels_with([(0, 80, 60, 0.0004)]))) == 0
    # too close to the border for the descriptor: dropped
    assert len(akz.find_extrema(plan, levels_with([(0, 10, 60, 0.01)]))) == 0
    # octave-1 points come back in level-0 coordinates
    k = akz.find_extrema(plan, levels_with([(4, 50, 30, 0.01)]))
    assert len(k) == 1 and (k["x"][0], k["y"][0], k["octave"][0], k["class_id"][0]) == (100, 60, 1, 4)


def test_mldb_layout(akz):
    plan = akz.make_plan(200, 120)
    lv = []
    for i in range(plan.nlevels):
        L = plan.lv[i]
        yy, xx = np.mgrid[0:L.h, 0:L.w].astype(np.float32)
        lv.append(dict(Lt=xx / L.w, Lsmooth=xx / L.w, Lx=np.ones((L.h, L.w), np.float32), Ly=np.zeros((L.h, L.w), np.float32),
                       Ldet=np.zeros((L.h, L.w), np.float32)))
    kp = np.zeros(1, akz.KP_DTYPE)
    kp["x"], kp["y"], kp["size"], kp["octave"], kp["class_id"] = 100, 60, 4.8, 0, 0
    k2, d = akz.compute_descriptors(plan, lv, kp)
    assert abs(k2["angle"][0]) < 1e-6                       # gradient along +x everywhere
    bits = np.unpackbits(d[0], bitorder="little")
    assert bits[486:].sum() == 0                            # 61 bytes = 486 bits + 2 clear bits
    # intensity channel of the 2 x 2 grid: cells are visited with x (i) outer: (left,top) (left,bottom) (right,top) (right,bottom);
    # intensity grows with x, so only the four right > left comparisons are 0 ... pairs (0,1) equal -> 0, (0,2) 0<..., etc.
    assert bits[:6].tolist() == [0, 0, 0, 0, 0, 0]         # value_i > value_j never holds for i < j when values grow with i
    # the gradient channels are constant (dx = 1 rotated by angle 0, dy = 0): no strict inequality
    assert bits[6:18].sum() == 0


def test_golden_fixture(akz, oracle):
    g = np.load(os.path.join(GOLD, "akaze61_expected.npz"))
    toy = np.load(os.path.join(GOLD, "toy_gray.npz"))["gray"]
    plan = akz.make_plan(640, 480)
    levels, k0 = akz.full_evolution(toy, plan)
    assert np.float32(k0) == g["toy_k0"]
    assert np.array_equal(np.array([zlib.crc32(levels[i]["Lt"].tobytes()) for i in range(8)], np.uint32), g["toy_lt_crc"])
    assert np.array_equal(np.array([zlib.crc32(levels[i]["Ldet"].tobytes()) for i in range(8)], np.uint32), g["toy_ldet_crc"])
    kp = akz.subpixel(plan, levels, akz.find_extrema(plan, levels))
    assert len(kp) == int(g["toy_ndetected"])
    q = oracle.quotas_extractor(1000, 8, 1.1892)
    chosen = []
    for lvl in range(8):
        idx = np.nonzero(kp["class_id"] == lvl)[0]
        if len(idx):
            chosen.append(idx[oracle.quadtree(kp["x"][idx], kp["y"][idx], kp["response"][idx], int(q[lvl]), 640, 480, tiebreak=np.arange(len(idx)))])
    kps, desc = akz.compute_descriptors(plan, levels, kp[np.concatenate(chosen)])
    assert np.array_equal(kps, g["toy_kps"]) and np.array_equal(desc, g["toy_desc"])
