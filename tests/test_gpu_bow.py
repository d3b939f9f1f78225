"""-m gpu: SURVEY 8f rank 2 — BoW quantisation (tree descent on the GPU) vs the oracle, and the extract -> transform ->
SearchByBoW chain it feeds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,L,levelsup", [(6, 3, 1), (10, 3, 2), (4, 5, 4), (3, 2, 4)])
def test_transform_nodes_match_oracle(afv, oracle, gpu_ctx, k, L, levelsup):
    voc = afv.Vocabulary.random(7 + k, k=k, L=L, ctx=gpu_ctx)
    desc = afv.synth.random_descriptors(3, 1500)
    leaf, nid = voc.transform_nodes(desc, levelsup)
    oleaf, onid = oracle.bow_transform(voc, desc, levelsup)
    assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid)
    assert np.all(voc.is_leaf[leaf])
    if L - levelsup <= 0:
        assert np.all(nid == 0)
    voc.close()


def test_ties_pick_first_child(afv, oracle, gpu_ctx):
    voc = afv.Vocabulary.random(5, k=5, L=2, ctx=gpu_ctx)
    voc.node_desc[1:6] = voc.node_desc[1]          # all level-1 children identical: first one must win
    desc = afv.synth.random_descriptors(9, 64)
    leaf, nid = voc.transform_nodes(desc, 1)
    oleaf, onid = oracle.bow_transform(voc, desc, 1)
    assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid) and np.all(nid == 1)
    voc.close()


def test_text_round_trip_and_vectors(afv, oracle, gpu_ctx, tmp_path):
    voc = afv.Vocabulary.random(11, k=6, L=3, ctx=gpu_ctx)
    p = tmp_path / "voc.txt"
    voc.saveToTextFile(str(p))
    voc2 = afv.Vocabulary.loadFromTextFile(str(p), ctx=gpu_ctx)
    assert voc2.k == 6 and voc2.L == 3 and voc2.size() == 6 ** 3
    assert np.array_equal(voc2.child_ptr, voc.child_ptr) and np.array_equal(voc2.child_idx, voc.child_idx)
    assert np.array_equal(voc2.node_desc, voc.node_desc) and np.allclose(voc2.weight, voc.weight)
    desc = afv.synth.random_descriptors(21, 900)
    bow, fv = voc2.transform(desc, levelsup=2)
    assert abs(sum(bow.values()) - 1.0) < 1e-9 and all(b >= 0 for b in bow.values())
    ids = [n for n, _ in fv]
    assert ids == sorted(ids) and all(idx == sorted(idx) for _, idx in fv)
    oleaf, onid = oracle.bow_transform(voc2, desc, 2)
    kept = voc2.weight[oleaf] > 0
    assert sorted(i for _, idx in fv for i in idx) == np.nonzero(kept)[0].tolist()
    for n, idx in fv:
        assert np.all(onid[idx] == n)
    voc.close(); voc2.close()


def test_extract_transform_match_chain(afv, oracle, gpu_ctx):
    """the reference's keyframe pipeline: extract -> ComputeBoW -> SearchByBoW(KF, KF), all three stages on the GPU"""
    img = afv.synth.corners_frame(1)
    k1, d1 = gpu_ctx.extract(img)
    k2, d2 = gpu_ctx.extract(np.roll(img, 3, axis=1))
    voc = afv.Vocabulary.random(3, k=8, L=3, ctx=gpu_ctx)
    _, fv1 = voc.transform(d1, levelsup=2)
    _, fv2 = voc.transform(d2, levelsup=2)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.75, True, ctx=gpu_ctx)
    got, n = m.SearchByBoW(afv.FeatureView(d1, fv1, angles=k1["angle"]), afv.FeatureView(d2, fv2, angles=k2["angle"]))
    want, wn = oracle.search_by_bow_kf_kf(d1, d2, fv1, fv2, None, None, k1["angle"], k2["angle"], 75.0, 0.75, True)
    assert n == wn and np.array_equal(got, want) and wn > 100
    voc.close()


def test_transform_at_the_shipped_vocabulary_shape(afv, oracle, gpu_ctx):
    """k = 10, L = 6 (createVocabulary.py:39-42 defaults; ORBvoc.txt): 1 111 111 nodes, a 53 MB device image - the lower levels of the
    descent miss L2, which the toy trees above never do"""
    voc = afv.Vocabulary.random(101, k=10, L=6, ctx=gpu_ctx)
    assert len(voc.weight) == 1111111 and voc.size() == 10 ** 6
    desc = afv.synth.random_descriptors(5, 2000)
    leaf, nid = voc.transform_nodes(desc, 4)
    oleaf, onid = oracle.bow_transform(voc, desc, 4)
    assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid)
    assert np.all(voc.is_leaf[leaf]) and len(np.unique(nid)) > 50     # level-2 nodes: at most 100
    # descriptors that ARE node descriptors must find their own leaf
    own = np.nonzero(voc.is_leaf)[0][afv.synth.lcg_states(3, 64) % 10 ** 6]
    d2 = voc.node_desc[own]
    leaf2, _ = voc.transform_nodes(d2, 4)
    oleaf2, _ = oracle.bow_transform(voc, d2, 4)
    assert np.array_equal(leaf2, oleaf2)
    voc.close()


def test_ragged_trees_and_wide_nodes(afv, oracle, gpu_ctx):
    """nodes with 1 .. 37 children (more than a 16-lane row: several chunks), leaves at different depths, ids not in breadth-first order"""
    s = afv.synth
    n = 4000
    st = s.lcg_states(17, n)
    parent = np.zeros(n, np.int32)
    for i in range(1, n):
        parent[i] = int(st[i] % min(i, 120)) if i < 3000 else 120 + int(st[i] % 2500)   # early nodes collect many children, later ones 0 - 3
    has_child = np.zeros(n, bool)
    has_child[parent[1:]] = True
    desc = s.lcg_bytes(18, n * 32).reshape(n, 32)
    voc = afv.Vocabulary(37, 60, parent, desc, np.ones(n), ~has_child, ctx=gpu_ctx)
    counts = np.diff(voc.child_ptr)
    assert counts.max() > 16 and counts[counts > 0].min() == 1
    q = s.random_descriptors(19, 1200)
    for levelsup in (1, 3, 57, 58, 70):
        leaf, nid = voc.transform_nodes(q, levelsup)
        oleaf, onid = oracle.bow_transform(voc, q, levelsup)
        assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid), levelsup
    voc.close()


@pytest.mark.parametrize("k,L,levelsup", [(8, 3, 1), (10, 4, 2), (6, 3, 2), (3, 5, 1)])
@pytest.mark.parametrize("ori", [False, True])
def test_bow_guided_matchers_over_small_and_large_nodes(afv, oracle, gpu_ctx, k, L, levelsup, ori):
    """SearchByBoW(KF, KF) and SearchByBoW(KF, F) over FeatureVectors whose nodes hold a handful of features (the register-resident node
    walk: both sides <= 64) as well as hundreds (the general walk), with validity masks"""
    s = afv.synth
    img = s.corners_frame(2)
    k1, d1 = gpu_ctx.extract(img)
    k2, d2 = gpu_ctx.extract(np.roll(img, 4, axis=1))
    voc = afv.Vocabulary.random(13 + k, k=k, L=L, ctx=gpu_ctx)
    _, fv1 = voc.transform(d1, levelsup=levelsup)
    _, fv2 = voc.transform(d2, levelsup=levelsup)
    sizes = [len(idx) for _, idx in fv1]
    v1 = (s.lcg_bytes(5, len(d1)) > 40).astype(np.uint8); v2 = (s.lcg_bytes(6, len(d2)) > 40).astype(np.uint8)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    m = afv.FeatureMatcher(0.8, ori, ctx=gpu_ctx)
    A = afv.FeatureView(d1, fv1, valid=v1, angles=k1["angle"]); B = afv.FeatureView(d2, fv2, valid=v2, angles=k2["angle"])
    got, n = m.SearchByBoW(A, B)
    want, wn = oracle.search_by_bow_kf_kf(d1, d2, fv1, fv2, v1, v2, k1["angle"], k2["angle"], 75.0, 0.8, ori)
    assert n == wn and np.array_equal(got, want) and wn > 30, (sizes[:8], n, wn)
    got, n = m.SearchByBoW(A, B, frame=True)
    want, wn = oracle.search_by_bow_kf_frame(d1, d2, fv1, fv2, v1, k1["angle"], k2["angle"], 75.0, 0.8, ori)
    assert n == wn and np.array_equal(got, want) and wn > 30
    voc.close()


# ---- descriptor-generic: Vocabulary::transform has a case per descriptor type (Vocabulary.cpp:156-206); binary ones of other sizes ----
@pytest.mark.parametrize("nbytes", [61, 48, 20])
@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 2), (4, 5, 4)])
def test_transform_of_other_descriptor_sizes(afv, oracle, gpu_ctx, nbytes, k, L, levelsup):
    voc = afv.Vocabulary.random(17 + k, k=k, L=L, ctx=gpu_ctx, desc_bytes=nbytes)
    desc = afv.synth.lcg_bytes(3, 1500 * nbytes).reshape(1500, nbytes)
    leaf, nid = voc.transform_nodes(desc, levelsup)
    oleaf, onid = oracle.bow_transform(voc, desc, levelsup)
    assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid) and np.all(voc.is_leaf[leaf])
    # descriptors that ARE node descriptors find their own leaf; descriptors of another width are refused
    own = np.nonzero(voc.is_leaf)[0][:64]
    assert np.array_equal(voc.transform_nodes(voc.node_desc[own], levelsup)[0], oracle.bow_transform(voc, voc.node_desc[own], levelsup)[0])
    voc.close()


@pytest.mark.parametrize("nbytes", [61, 48])
@pytest.mark.parametrize("ori", [False, True])
def test_bow_guided_matchers_on_other_descriptor_sizes(afv, oracle, gpu_ctx, nbytes, ori):
    """SearchByBoW(KF, KF) / (KF, F) over 61- and 48-byte descriptors with FeatureVectors from a vocabulary of that size"""
    s = afv.synth
    img = s.corners_frame(2)
    k1, d1 = gpu_ctx.extract(img)
    k2, d2 = gpu_ctx.extract(np.roll(img, 4, axis=1))
    wide = lambda d: np.ascontiguousarray(np.concatenate([d, np.roll(d, 5, axis=1) ^ np.uint8(0x5A)], 1)[:, :nbytes])
    d1, d2 = wide(d1), wide(d2)
    th = float(round(75.0 * nbytes / 32.0))
    voc = afv.Vocabulary.random(23, k=8, L=3, ctx=gpu_ctx, desc_bytes=nbytes)
    _, fv1 = voc.transform(d1, levelsup=2)
    _, fv2 = voc.transform(d2, levelsup=2)
    v1 = (s.lcg_bytes(5, len(d1)) > 40).astype(np.uint8); v2 = (s.lcg_bytes(6, len(d2)) > 40).astype(np.uint8)
    afv.FeatureMatcher.setDescriptorDistanceThresholds(th)
    try:
        m = afv.FeatureMatcher(0.8, ori, ctx=gpu_ctx)
        A = afv.FeatureView(d1, fv1, valid=v1, angles=k1["angle"]); B = afv.FeatureView(d2, fv2, valid=v2, angles=k2["angle"])
        got, n = m.SearchByBoW(A, B)
        want, wn = oracle.search_by_bow_kf_kf(d1, d2, fv1, fv2, v1, v2, k1["angle"], k2["angle"], th, 0.8, ori)
        assert n == wn and np.array_equal(got, want) and wn > 30
        got, n = m.SearchByBoW(A, B, frame=True)
        want, wn = oracle.search_by_bow_kf_frame(d1, d2, fv1, fv2, v1, k1["angle"], k2["angle"], th, 0.8, ori)
        assert n == wn and np.array_equal(got, want) and wn > 30
        # brute force (no FeatureVectors) of wide descriptors: the popcount walk (the MFMA / top-k pair path is for 32-byte sets)
        got, n = m.SearchByBoW(afv.FeatureView(d1, angles=k1["angle"]), afv.FeatureView(d2, angles=k2["angle"]))
        want, wn = oracle.search_by_bow_kf_kf(d1, d2, None, None, None, None, k1["angle"], k2["angle"], th, 0.8, ori)
        assert n == wn and np.array_equal(got, want) and wn > 100
    finally:
        afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
    voc.close()


# ---- float descriptors: the non-binary cases of Vocabulary::transform (Vocabulary.cpp:158-187: SIFT128, SURF64, KAZE64, R2D2) ----
def _float_descriptors(afv, seed, n, dim):
    d = afv.synth.lcg_bytes(seed, n * dim).reshape(n, dim).astype(np.float32) ** 2
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


@pytest.mark.parametrize("dim", [128, 64, 256])
@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 2), (4, 5, 4), (20, 2, 1)])
def test_float_vocabulary_descent(afv, oracle, gpu_ctx, dim, k, L, levelsup):
    """L2^2 descent (float differences and squares, double accumulation in index order: DBoW2's float descriptor classes), first minimum
    wins; 20 children = two chunks of the 16-lane row"""
    voc = afv.Vocabulary.random_float(31 + k, k=k, L=L, ctx=gpu_ctx, dim=dim)
    desc = _float_descriptors(afv, 7, 1200, dim)
    leaf, nid = voc.transform_nodes(desc, levelsup)
    oleaf, onid = oracle.bow_transform(voc, desc, levelsup)
    assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid) and np.all(voc.is_leaf[leaf])
    assert len(np.unique(leaf)) > 100
    # descriptors that ARE node descriptors (distance exactly 0 somewhere on the way); the vectors come out as for binary vocabularies
    own = np.nonzero(voc.is_leaf)[0][:96]
    assert np.array_equal(voc.transform_nodes(voc.node_desc[own], levelsup)[0], oracle.bow_transform(voc, voc.node_desc[own], levelsup)[0])
    bow, fv = voc.transform(desc, levelsup=levelsup)
    assert abs(sum(bow.values()) - 1.0) < 1e-9 and [n for n, _ in fv] == sorted(n for n, _ in fv)
    voc.close()


def test_float_vocabulary_ties_and_kind_checks(afv, oracle, gpu_ctx):
    voc = afv.Vocabulary.random_float(5, k=5, L=2, ctx=gpu_ctx, dim=128)
    voc.node_desc[1:6] = voc.node_desc[3]           # all level-1 children identical: the first one wins
    desc = _float_descriptors(afv, 9, 64, 128)
    leaf, nid = voc.transform_nodes(desc, 1)
    oleaf, onid = oracle.bow_transform(voc, desc, 1)
    assert np.array_equal(leaf, oleaf) and np.array_equal(nid, onid) and np.all(nid == 1)
    # a binary vocabulary refuses floats and the other way round (no silent reinterpretation of the bytes)
    import ctypes as C
    b = afv.Vocabulary.random(5, k=5, L=2, ctx=gpu_ctx)
    out = np.zeros(64, np.int32)
    assert gpu_ctx.lib.afv_bow_transform_f32(gpu_ctx.handle, b._device(), desc.ctypes.data, 64, 1, out.ctypes.data, out.ctypes.data) == afv._lib.EINVAL
    assert gpu_ctx.lib.afv_bow_transform(gpu_ctx.handle, voc._device(), desc.ctypes.data, 64, 1, out.ctypes.data, out.ctypes.data) == afv._lib.EINVAL
    h = C.c_void_p()
    assert gpu_ctx.lib.afv_vocab_create_f32(gpu_ctx.handle, 5, 2, len(voc.weight), voc.child_ptr.ctypes.data, voc.child_idx.ctypes.data,
                                            voc.node_desc.ctypes.data, 100, C.byref(h)) == afv._lib.EINVAL
    fr = afv.Frame(gpu_ctx)
    fr.extract(afv.synth.corners_frame(1))
    with pytest.raises(afv._lib.AfvError):
        fr.ComputeBoW(voc)
    fr.close(); voc.close(); b.close()
