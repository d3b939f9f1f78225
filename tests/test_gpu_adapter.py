"""-m gpu: the C++ host adapter (anyfeature-vslam_amd/adapter/afv_adapter.hpp) run as a plain C++ process — no python,
no torch in the loop — and checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "anyfeature-vslam_amd", "adapter", "adapter_selftest")


def test_cpp_adapter_end_to_end(afv, oracle, tmp_path):
    assert os.path.exists(BIN), "adapter_selftest is not built: __graft_entry__.build() compiles it (g++, no GPU needed)"
    img = afv.synth.corners_frame(1)
    raw = tmp_path / "frame.raw"
    raw.write_bytes(img.tobytes())
    out = str(tmp_path / "o")
    r = subprocess.run([BIN, str(raw), "640", "480", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    n1, n2, nm = [int(v) for v in lines[0].split()]
    k1 = np.fromfile(out + ".kps1", dtype=afv.KP_DTYPE); d1 = np.fromfile(out + ".desc1", dtype=np.uint8).reshape(-1, 32)
    k2 = np.fromfile(out + ".kps2", dtype=afv.KP_DTYPE); d2 = np.fromfile(out + ".desc2", dtype=np.uint8).reshape(-1, 32)
    m21 = np.fromfile(out + ".match21", dtype=np.int32)
    size1 = np.fromfile(out + ".size1", dtype=np.float32)
    ok1, od1 = oracle.orb_extract(img)
    ok2, od2 = oracle.orb_extract(np.roll(img, 4, axis=1))
    assert len(k1) == n1 and len(k2) == n2
    assert k1.tobytes() == ok1.tobytes() and np.array_equal(d1, od1)
    assert k2.tobytes() == ok2.tobytes() and np.array_equal(d2, od2)
    want, wn = oracle.search_by_bow_kf_kf(od2, od1, angle1=ok2["angle"], angle2=ok1["angle"], th_low=75.0, nnratio=0.6,
                                          check_orientation=True)
    assert nm == wn and np.array_equal(m21, want) and wn > 300
    assert np.array_equal(size1, oracle.size_sigma(ok1)[0])
    # FeatureExtractor::mvImagePyramid of the last extracted frame (ImagePyramid accessor: what Frame::ComputeStereoMatches reads)
    _, _, tr2 = oracle.orb_extract_trace(np.roll(img, 4, axis=1))
    pyr = np.fromfile(out + ".pyramid2", dtype=np.uint8)
    assert pyr.tobytes() == b"".join(np.ascontiguousarray(l).tobytes() for l in tr2["level"])

    # Vocabulary::transform + SearchByBoW over the feature vectors (the selftest's LCG tree, k = 6, L = 2, levelsup 1)
    from oracle import akaze_binding as akz
    k, L = 6, 2
    n = 1 + k + k * k
    parent = [0] * (1 + k) + [1 + (i - 1 - k) // k for i in range(1 + k, n)]
    leaf = [False] * (1 + k) + [True] * (k * k)
    weight = np.ones(n); weight[[i for i in range(1 + k, n) if i % 7 == 0]] = 0.0; weight[0] = 0.0
    nd = np.fromfile(out + ".vocdesc", dtype=np.uint8).reshape(n, 32)
    voc = afv.Vocabulary(k, L, parent, nd, weight, leaf)
    fvs = []
    for d in (od1, od2):
        lf, nid = oracle.bow_transform(voc, d, 1)
        fv = {}
        for i in np.nonzero(voc.weight[lf] > 0)[0].tolist():
            fv.setdefault(int(nid[i]), []).append(i)
        fvs.append(sorted(fv.items()))
    flat = np.fromfile(out + ".fv1", dtype=np.int32).reshape(-1, 2)
    assert flat.tolist() == [[nd_, f] for nd_, idx in fvs[0] for f in idx]
    nb = int(lines[1].split()[3])
    wantb, wnb = oracle.search_by_bow_kf_kf(od1, od2, fvs[0], fvs[1], None, None, ok1["angle"], ok2["angle"], 75.0, 0.6, True)
    assert nb == wnb and np.array_equal(np.fromfile(out + ".bowmatch12", dtype=np.int32), wantb)
    # ---- the projection-guided wrappers of FeatureMatcherHip (SearchByProjection x4, Fuse x2, SearchBySim3,
    # SearchForInitialization) vs the oracle on the views the selftest built ----
    size2 = np.fromfile(out + ".size2", dtype=np.float32)
    inf1 = np.fromfile(out + ".inf1", dtype=np.float32)
    assert np.array_equal(size2, oracle.size_sigma(ok2)[0]) and np.array_equal(inf1, oracle.size_sigma(ok1)[2])
    f32 = np.float32
    F1 = afv.FrameGridView(od1, np.stack([ok1["x"], ok1["y"]], 1), size1, angles=ok1["angle"], inf=inf1)
    F2 = afv.FrameGridView(od2, np.stack([ok2["x"], ok2["y"]], 1), size2, angles=ok2["angle"], inf=oracle.size_sigma(ok2)[2])
    Q2 = afv.ProjectionQueries(od2, ok2["x"] - f32(4.0), ok2["y"], f32(15.0) * size2, size2 / f32(1.2), size2 * f32(1.2), angles=ok2["angle"])
    Q1 = afv.ProjectionQueries(od1, ok1["x"] + f32(4.0), ok1["y"], f32(15.0) * size1, size1 / f32(1.2), size1 * f32(1.2), angles=ok1["angle"])
    n1 = len(ok1)
    QI = afv.ProjectionQueries(od1, ok1["x"], ok1["y"], np.full(n1, 100.0, f32), np.zeros(n1, f32), np.full(n1, 3.6, f32),
                               valid=(ok1["octave"] == 0).astype(np.uint8), angles=ok1["angle"])
    counts = [int(v) for v in lines[2].split()[1:]]
    rd = lambda name: np.fromfile(out + name, dtype=np.int32)
    cases = [
        (".p_local", oracle.match_projection(F1, Q2, th_high=75.0, nnratio=0.8)),
        (".p_last", oracle.match_projection(F1, Q2, th_high=75.0, nnratio=0.8, check_orientation=True, last_frame=True)),
        (".p_reloc", oracle.match_projection(F1, Q2, th_high=60.0, nnratio=0.8, check_orientation=True, last_frame=True)),
        (".p_sim3p", oracle.match_projection(F1, Q2, th_high=75.0, nnratio=0.8, check_orientation=False, last_frame=True)),
        (".p_fuse", oracle.match_projection(F1, Q2, th_high=75.0, fuse=True)),
    ]
    F1n = afv.FrameGridView(od1, np.stack([ok1["x"], ok1["y"]], 1), size1, angles=ok1["angle"])
    cases.append((".p_fuse3", oracle.match_projection(F1n, Q2, th_high=75.0, fuse=True)))
    cases.append((".p_sim3", oracle.match_sim3(F2, Q1, F1, Q2, th_high=75.0)))
    cases.append((".p_init", oracle.match_initialization(F2, QI, th_low=75.0, nnratio=0.9, check_orientation=True)))
    for (name, (want_v, want_n)), got_n in zip(cases, counts):
        assert got_n == want_n, (name, got_n, want_n)
        assert np.array_equal(rd(name), want_v), name
    assert counts[0] > 300 and counts[6] > 200 and counts[7] > 50
    # AKAZE61 plugin through the C++ adapter vs the oracle pipeline
    ka = np.fromfile(out + ".akz_kps", dtype=afv.KP_DTYPE); da = np.fromfile(out + ".akz_desc", dtype=np.uint8).reshape(-1, 61)
    plan = akz.make_plan(640, 480)
    levels, _ = akz.full_evolution(img, plan)
    kp = akz.subpixel(plan, levels, akz.find_extrema(plan, levels))
    q = oracle.quotas_extractor(1000, 8, 1.1892)
    chosen = []
    for lvl in range(8):
        idx = np.nonzero(kp["class_id"] == lvl)[0]
        if len(idx):
            chosen.append(idx[oracle.quadtree(kp["x"][idx], kp["y"][idx], kp["response"][idx], int(q[lvl]), 640, 480, tiebreak=np.arange(len(idx)))])
    wk, wd = akz.compute_descriptors(plan, levels, kp[np.concatenate(chosen)])
    assert ka.tobytes() == wk.tobytes() and np.array_equal(da, wd)
