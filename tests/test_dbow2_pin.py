"""Pinned parity of the BoW quantisation against a real DBoW2: active when tests/golden/dbow2_pin.npz exists
(tools/pin_against_dbow2.py, run where the reference's DBoW2 fork and ORBvoc.txt are available).  Without the file the pinned tests
skip; the self-test below proves the kit end to end with a stand-in answer produced by the oracle (it pins nothing)."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIN = os.path.join(ROOT, "tests", "golden", "dbow2_pin.npz")
NEED = "no tests/golden/dbow2_pin.npz: run tools/pin_against_dbow2.py where DBoW2 + ORBvoc.txt are available (parity unpinned until then)"


def _subset_vocabulary(afv, d, ctx=None):
    """the visited part of the real vocabulary as a Vocabulary: dense ids in the order of the original ids (children keep their DBoW2
    order), leaves of the ORIGINAL tree stay leaves, unvisited inner nodes become childless (never reached by the pinned descents)"""
    ids = d["node_id"]
    dense = {int(o): i for i, o in enumerate(ids)}
    parent = np.array([dense[int(p)] if i else 0 for i, p in enumerate(d["parent"])], np.int32)
    voc = afv.Vocabulary(int(d["k"]), int(d["L"]), parent, d["desc"], d["weight"], d["is_leaf"], ctx=ctx)
    return voc, ids


def _check(afv, d, leaf, nid):
    voc_ids = d["node_id"]
    got_word = d["word_id"][leaf]
    assert np.array_equal(got_word, d["answer_word"]), "word ids differ from DBoW2 for %d of %d descriptors" % (
        int((got_word != d["answer_word"]).sum()), len(got_word))
    assert np.array_equal(voc_ids[nid], d["answer_node"]), "node ids at levelsup differ from DBoW2"
    assert np.array_equal(d["weight"][leaf], d["answer_weight"]), "word weights differ"


@pytest.mark.parametrize("path", [PIN if os.path.exists(PIN) else None])
def test_oracle_descent_against_real_dbow2(afv, oracle, path):
    if path is None:
        pytest.skip(NEED)
    d = np.load(path)
    voc, _ = _subset_vocabulary(afv, d)
    leaf, nid = oracle.bow_transform(voc, d["request"], int(d["levelsup"]))
    _check(afv, d, leaf, nid)


@pytest.mark.gpu
@pytest.mark.parametrize("path", [PIN if os.path.exists(PIN) else None])
def test_hip_descent_against_real_dbow2(afv, gpu_ctx, path):
    if path is None:
        pytest.skip(NEED)
    d = np.load(path)
    voc, _ = _subset_vocabulary(afv, d, ctx=gpu_ctx)
    leaf, nid = voc.transform_nodes(d["request"], int(d["levelsup"]))
    _check(afv, d, leaf, nid)
    voc.close()


def test_dbow2_pinning_kit_end_to_end_with_a_stand_in_answer(afv, oracle, tmp_path):
    """request -> (stand-in for DBoW2: the oracle's descent on a random vocabulary written in DBoW2's text format) -> pin -> consumer"""
    tool = os.path.join(ROOT, "tools", "pin_against_dbow2.py")
    text = open(tool).read()
    assert "anyfeature" not in text.replace("AnyFeature", "") and "import oracle" not in text     # runs where only numpy exists
    env = dict(os.environ, AFV_PIN_OUT=str(tmp_path))
    subprocess.run([sys.executable, tool, "request"], check=True, env=env, capture_output=True)
    req = np.loadtxt(os.path.join(str(tmp_path), "dbow2_request.txt"), dtype=np.int64).astype(np.uint8)
    voc = afv.Vocabulary.random(41, k=7, L=5)
    vpath = os.path.join(str(tmp_path), "voc.txt")
    voc.saveToTextFile(vpath)
    leaf, nid = oracle.bow_transform(voc, req, 4)
    with open(os.path.join(str(tmp_path), "answer.txt"), "w") as fh:
        for lf, nd in zip(leaf, nid):
            fh.write("%d %d %.17g\n" % (voc.word_id[lf], nd, voc.weight[lf]))
    subprocess.run([sys.executable, tool, "pin", vpath, os.path.join(str(tmp_path), "answer.txt")], check=True, env=env, capture_output=True)
    d = np.load(os.path.join(str(tmp_path), "dbow2_pin.npz"))
    assert int(d["vocabulary_nodes"]) == len(voc.weight) and len(d["node_id"]) < len(voc.weight)     # a subset travels, not the tree
    sub, ids = _subset_vocabulary(afv, d)
    l2, n2 = oracle.bow_transform(sub, d["request"], 4)
    _check(afv, d, l2, n2)
    # and a corrupted answer is caught
    bad = dict(d)
    bad["answer_word"] = d["answer_word"].copy()
    bad["answer_word"][3] += 1
    with pytest.raises(AssertionError):
        _check(afv, bad, l2, n2)
