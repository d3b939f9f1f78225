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
