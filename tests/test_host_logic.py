"""CPU: host-side logic of the python mirror (settings statics, FeatureView flattening, synthetic generators, sharding)."""
import numpy as np


def test_settings_statics_are_shared(afv):
    S = afv.FeatureExtractorSettings
    s1 = S({"FeatureExtractor.numOctaves": 6, "FeatureExtractor.scaleFactor": 1.3, "FeatureExtractor.detectionTh": 15.0})
    s2 = S("none")  # the second extractor inherits the statics (Tracking.cc:84, FeatureExtractor.cpp:26,36-38)
    assert (s2.nOctaves, s2.scaleFactor, s2.detectTh) == (6, 1.3, 15.0) == (s1.nOctaves, s1.scaleFactor, s1.detectTh)
    S({"FeatureExtractor.numOctaves": 8, "FeatureExtractor.scaleFactor": 1.2, "FeatureExtractor.detectionTh": 20.0})
    assert S().nOctaves == 8 and S().ON_automaticTuning is True


def test_settings_from_yaml_file(afv, tmp_path):
    p = tmp_path / "orb32_settings.yaml"
    p.write_text("%YAML:1.0\n\nFeatureExtractor.numOctaves: 8\nFeatureExtractor.scaleFactor: 1.2\nFeatureExtractor.detectionTh: 20.0\n"
                 "FeatureMatcher.matchingTh: 75.0\n")
    s = afv.FeatureExtractorSettings(str(p))
    assert (s.nOctaves, s.scaleFactor, s.detectTh) == (8, 1.2, 20.0)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(str(p))
    assert afv.FeatureMatcher.TH_LOW == 75.0


def test_feature_view_csr(afv):
    d = afv.synth.random_descriptors(1, 10)
    v = afv.FeatureView(d, [(2, [0, 3]), (5, []), (9, [1, 2, 4])])
    ids, ptr, flat, n = v.csr()
    assert ids.tolist() == [2, 5, 9] and ptr.tolist() == [0, 2, 2, 5] and flat.tolist() == [0, 3, 1, 2, 4] and n == 3
    assert afv.FeatureView(d).csr()[3] == 0 and afv.FeatureView(np.zeros((0, 32), np.uint8)).N == 0


def test_synth_is_reproducible_and_matches_scalar_lcg(afv):
    s = afv.synth
    x, ref = 12345, []
    for _ in range(50):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        ref.append((x >> 8) & 255)
    assert s.lcg_bytes(12345, 50).tolist() == ref
    f1, f2 = s.corners_frame(3), s.corners_frame(3)
    assert np.array_equal(f1, f2) and f1.shape == (480, 640) and f1.dtype == np.uint8
    assert not np.array_equal(f1, s.corners_frame(4))
    assert s.corners_frame(3, 1280, 720).shape == (720, 1280)
    d = s.random_descriptors(5, 100)
    p = s.perturbed_descriptors(d, 6)
    flips = np.unpackbits(d ^ p, axis=1).sum(1)
    assert 10 < np.median(flips) < 45 and (flips > 90).mean() > 0.15  # ~10 % bit flips, ~30 % rows replaced


def test_shard_range_partitions_exactly(afv):
    dist = __import__("importlib").import_module("anyfeature-vslam_amd.dist")
    for n in (0, 1, 7, 8, 1000, 10000):
        for world in (1, 2, 3, 8):
            rs = [dist.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1 and sizes == dist.shard_sizes(n, world)
    a, b = dist.lcg_pairs(3, 1000, 50)
    assert a.min() >= 0 and a.max() < 50 and np.all(a != b)


def test_library_shard_range_matches_python(afv):
    """afv_shard_range (C-ABI, what a C++ host uses) and dist.shard_range (Python mirror) are the same block partition"""
    import importlib
    tbl = importlib.import_module("anyfeature-vslam_amd.table")
    d = importlib.import_module("anyfeature-vslam_amd.dist")
    for n in (0, 1, 7, 8, 9, 10000, 10007):
        for world in (1, 2, 3, 4, 8):
            covered = 0
            for r in range(world):
                lo, hi = tbl.shard_range(n, r, world)
                assert (lo, hi) == d.shard_range(n, r, world)
                assert lo == covered
                covered = hi
            assert covered == n


def test_table_and_comm_reject_bad_arguments(afv):
    import ctypes as C
    lib = afv._lib.load()
    h = C.c_void_p()
    E = afv._lib.EINVAL
    assert lib.afv_table_create(None, 4, 4, C.byref(h)) == E
    lib.afv_table_destroy(None)
    assert lib.afv_table_set(None, 0, None, None, 0) == E
    assert lib.afv_table_set_featvec(None, 0, None, None, None, 0) == E
    assert lib.afv_table_set_geometry(None, 0, None, None, None) == E
    assert lib.afv_table_set_valid(None, 0, None) == E
    assert lib.afv_table_match_pairs(None, None, None, 0, 75.0, 0.75, 1, None, None) == E
    assert lib.afv_table_match_pairs_device(None, None, None, 0, 75.0, 0.75, 1, None, None, None) == E
    assert lib.afv_table_match_bow(None, None, None, 0, 75.0, 0.75, 1, None, None) == E
    assert lib.afv_table_match_triangulation(None, None, None, None, 0, None, None) == E
    assert lib.afv_table_broadcast(None, None, 0, None) == E
    assert lib.afv_table_sync_counts(None) == E
    assert lib.afv_comm_unique_id(None) == E
    assert lib.afv_comm_create(None, None, 1, 0, C.byref(h)) == E
    lib.afv_comm_destroy(None)
    assert lib.afv_comm_rank(None) == E and lib.afv_comm_size(None) == E
    assert lib.afv_comm_broadcast(None, None, 0, 0, None) == E
    assert lib.afv_comm_allgather(None, None, None, 0, None) == E
    assert lib.afv_set_split_chunks(None, 4) == E


def test_keyframe_table_generator_is_reproducible(afv):
    t1, a1, n1 = afv.synth.keyframe_table(6, 64, seed=3)
    t2, a2, n2 = afv.synth.keyframe_table(6, 64, seed=3)
    assert np.array_equal(t1, t2) and np.array_equal(a1, a2) and np.array_equal(n1, n2)
    assert t1.shape == (6, 64, 32) and a1.dtype == np.float32 and (a1 >= 0).all() and (a1 < 360).all()
    # neighbours share most rows up to ~10 % bit flips, distant keyframes do not
    near = np.unpackbits(t1[0] ^ t1[1], axis=1).sum(axis=1)
    far = np.unpackbits(t1[0] ^ t1[5], axis=1).sum(axis=1)
    assert (near < 75).sum() > 30 and (far < 75).sum() < (near < 75).sum()


def test_bench_self_launches_ranks_without_a_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run (one rank per
    GPU, rendezvous on 127.0.0.1) instead of asserting — before anything touches the GPU or the process's stdout"""
    import importlib
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    bench = importlib.import_module("bench")
    seen = {}

    def fake_execv(exe, argv):
        seen["exe"], seen["argv"] = exe, list(argv)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--workload", "pairs10k"])
    try:
        bench.main()
    except SystemExit:
        pass
    a = seen["argv"]
    assert seen["exe"] == sys.executable and a[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in a and a[a.index("--nproc-per-node") + 1] == "4"
    assert a[a.index("--master-addr") + 1] == "127.0.0.1" and a[-6:] == ["--gpus", "4", "--steps", "3", "--workload", "pairs10k"]
    assert bench._REAL_STDOUT is None          # stdout untouched on the launching side


# ---------------- the one-launch pyramid's work plan (host logic of k_pyramid_fused), replayed on the CPU ----------------
def _pyramid_plan(afv, w, h, nlevels, sf, tw, th):
    import ctypes as C
    lib = afv._lib.load()
    p = afv._lib.OrbParams(1000, nlevels, sf, 20, w, h, 1)
    regions = np.zeros(8 * 4096 * 4, np.int16)
    info = np.zeros(4 + 6 * 8, np.int32)
    tables = np.zeros(2 * 8 * 2 * 4096, np.int16)
    rc = lib.afv_debug_pyramid_plan(C.byref(p), w, h, tw, th, afv._lib.ptr(regions), regions.size, afv._lib.ptr(info), afv._lib.ptr(tables),
                                    tables.size)
    assert rc == 0, rc
    nl, ntx, nty, lds = (int(v) for v in info[:4])
    f = info[4:].reshape(6, 8)
    rx = regions[:nl * ntx * 4].reshape(nl, ntx, 4).astype(int)
    ry = regions[nl * ntx * 4:nl * (ntx + nty) * 4].reshape(nl, nty, 4).astype(int)
    tab = tables.reshape(-1, 2).astype(int)
    return dict(nl=nl, ntx=ntx, nty=nty, lds=lds, w=f[0], h=f[1], pitch=f[2], lg_q=f[3], tabx=f[4], taby=f[5], rx=rx, ry=ry, tab=tab)


def _replay_fused_pyramid(P, img):
    """the data flow of k_pyramid_fused: per top-level tile, regions of levels 1.. computed from the level-0 window through
    region-relative tables; the owned rectangle (x widened to dwords) is stored.  Returns the levels and a write-count map."""
    nl = P["nl"]
    levels = [img] + [np.zeros((P["h"][l], (P["w"][l] + 3) // 4 * 4 + 4), np.int64) - 1 for l in range(1, nl)]
    for ty in range(P["nty"]):
        for tx in range(P["ntx"]):
            rx0, ry0 = P["rx"][0, tx], P["ry"][0, ty]
            S = np.zeros((ry0[1] - ry0[0] + 1, (rx0[1] - rx0[0] + 4) // 4 * 4), np.int64)
            src = img[ry0[0]:ry0[1] + 1, rx0[0]:rx0[0] + S.shape[1]]
            S[:, :src.shape[1]] = src
            prev_x, prev_y = rx0, ry0
            for l in range(1, nl):
                rx, ry = P["rx"][l, tx], P["ry"][l, ty]
                dwp, dh = rx[1] - rx[0] + 1, ry[1] - ry[0] + 1
                assert dwp % 4 == 0 and rx[0] % 4 == 0 and dwp <= P["pitch"][l] and dwp // 4 <= (1 << P["lg_q"][l])
                sw, sh = prev_x[1] - prev_x[0] + 1, S.shape[0]
                X = rx[0] + np.arange(dwp)
                inside = X < P["w"][l]
                ex = P["tab"][P["tabx"][l] + np.minimum(X, P["w"][l] - 1)]
                ox = np.where(inside, ex[:, 0] - prev_x[0], 0)
                wx = np.where(inside, ex[:, 1], 0)
                assert ox.min() >= 0 and ox.max() < sw, (l, tx)
                ey = P["tab"][P["taby"][l] + ry[0] + np.arange(dh)]
                oy, wy = ey[:, 0] - prev_y[0], ey[:, 1]
                assert oy.min() >= 0 and oy.max() < sh, (l, ty)
                # a tap with a non-zero weight must be a pixel the region really holds
                assert np.all((ox + 1 < sw) | (wx == 0)) and np.all((oy + 1 < sh) | (wy == 0))
                hrow = (256 - wx) * S[:, ox] + wx * S[:, np.minimum(ox + 1, sw - 1)]
                D = (hrow[oy] * (256 - wy)[:, None] + hrow[np.minimum(oy + 1, sh - 1)] * wy[:, None] + 32768) >> 16
                gx = rx[0] + np.arange(dwp)
                own_x = (gx // 4 * 4 >= (rx[2] & ~3)) & (gx // 4 * 4 < rx[3])
                gy = ry[0] + np.arange(dh)
                own_y = (gy >= ry[2]) & (gy < ry[3])
                tgt = levels[l]
                for yi in np.nonzero(own_y)[0]:
                    row = tgt[gy[yi]]
                    cols = gx[own_x]
                    old = row[cols]
                    new = D[yi][own_x]
                    assert np.all((old == -1) | (old == new)), "two workgroups disagree on a pixel"
                    row[cols] = new
                S, prev_x, prev_y = D, rx, ry
    return levels


def test_one_launch_pyramid_plan_reproduces_the_level_chain(afv, oracle):
    """k_pyramid_fused computes every level from the level-0 window of its tile: the plan must make every pixel of every level
    owned by some tile, computed from taps the tile holds, and equal to the level-by-level INTER_LINEAR_EXACT chain"""
    cases = [(640, 480, 8, 1.2, 16, 8), (1280, 720, 8, 1.2, 16, 8), (640, 480, 8, 1.2, 32, 16), (642, 481, 8, 1.2, 32, 16), (333, 251, 8, 1.2, 16, 16), (1280, 720, 8, 1.2, 32, 16),
             (640, 480, 8, 1.1892, 32, 16), (640, 480, 4, 1.5, 32, 16), (640, 480, 3, 2.0, 32, 32), (640, 480, 2, 1.2, 64, 32),
             (752, 480, 8, 1.2, 48, 24)]
    for (w, h, nl, sf, tw, th) in cases:
        P = _pyramid_plan(afv, w, h, nl, sf, tw, th)
        img = afv.synth.corners_frame(5, w, h).astype(np.int64)
        levels = _replay_fused_pyramid(P, img)
        ref = img.astype(np.uint8)
        for l in range(1, nl):
            ref = oracle.resize_linear_exact(ref, int(P["w"][l]), int(P["h"][l]))
            got = levels[l][:, :P["w"][l]]
            assert got.min() >= 0, "level %d of %s: a pixel nobody owns" % (l, (w, h, nl, sf))
            assert np.array_equal(got, ref), "level %d of %s differs" % (l, (w, h, nl, sf))
        assert P["lds"] <= 160 * 1024


def test_bench_table_broadcast_bytes_and_job_partition(afv):
    """config #4 bookkeeping bench.py reports for N > 1: one replication of the K = 1000 x 1000 x 32 B table (+ angles, counts) is
    36 004 000 bytes, and the 10 000 jobs are cut into contiguous per-rank ranges that cover them exactly"""
    import importlib
    import bench
    tbl = importlib.import_module("anyfeature-vslam_amd.table")
    assert bench.table_broadcast_bytes(1000, 1000) == 36004000
    for world in (1, 2, 4, 8):
        rs = [tbl.shard_range(10000, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == 10000 and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in rs) - min(b - a for a, b in rs) <= 1
