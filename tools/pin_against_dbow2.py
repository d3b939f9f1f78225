#!/usr/bin/env python3
"""Pinning kit for the BoW quantisation: DBoW2 is an EMPTY submodule in the reference (.gitmodules:1-3), so
vocabulary.py + k_bow_transform follow upstream DBoW2's published algorithm (TemplatedVocabulary::loadFromTextFile / transform) and
nothing in this image can confirm it.  This script needs a real ORBvoc.txt and the output of a real DBoW2 `transform` for a set of
descriptors it hands out; it runs anywhere (numpy only) and may import nothing from this repository.

  1. python tools/pin_against_dbow2.py request                         -> tests/golden/dbow2_request.txt (N lines of 32 byte values)
  2. with the reference's DBoW2 (any build of the fork: the call is the one Vocabulary::transform makes, src/Vocabulary.cpp:156-206):
         ORBVocabulary voc; voc.loadFromTextFile("ORBvoc.txt");
         for each line i of dbow2_request.txt:  cv::Mat d(1, 32, CV_8U) <- the 32 values
             DBoW2::WordId wid; DBoW2::NodeId nid; DBoW2::WordValue w;
             voc.transform(d, wid, w, &nid, 4);            // TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
             printf("%u %u %.17g\n", wid, nid, w);          -> dbow2_answer.txt
  3. python tools/pin_against_dbow2.py pin ORBvoc.txt dbow2_answer.txt  -> tests/golden/dbow2_pin.npz  (commit it)

The pin file carries DBoW2's answers AND the part of the vocabulary the descents visit (every visited node with all its siblings:
original ids, parents, descriptors, weights, leaf flags - a few hundred KB instead of the 145 MB text file), so tests/test_dbow2_pin.py
can rebuild a tree on which the descent of every request descriptor is the one DBoW2 performed, and compare node by node: the CPU test
checks vocabulary.py's loader conventions + the oracle, the -m gpu test k_bow_transform.  Until somebody runs steps 2-3 the BoW path stays
"parity unpinned" (DESIGN.md section 2)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("AFV_PIN_OUT") or os.path.join(ROOT, "tests", "golden")
NREQ = 400


def lcg_bytes(seed, n):
    a = np.full(n, 1664525, dtype=np.uint32)
    with np.errstate(over="ignore"):
        A = np.multiply.accumulate(a, dtype=np.uint32)
        geo = np.concatenate([np.ones(1, np.uint32), A[:-1]])
        Cs = (np.add.accumulate(geo, dtype=np.uint32) * np.uint32(1013904223)).astype(np.uint32)
        st = A * np.uint32(seed & 0xFFFFFFFF) + Cs
    return ((st >> np.uint32(8)) & np.uint32(255)).astype(np.uint8)


def request_descriptors():
    return lcg_bytes(20240901, NREQ * 32).reshape(NREQ, 32)


def load_vocabulary_text(path):
    """DBoW2 text format (TemplatedVocabulary::loadFromTextFile): 'k L scoring weighting', then per node 'parent isLeaf d0..d31 weight';
    node ids are assigned in file order starting at 1 (0 = root)"""
    with open(path) as fh:
        k, L = [int(v) for v in fh.readline().split()[:2]]
        parent, leaf, desc, weight = [0], [0], [np.zeros(32, np.uint8)], [0.0]
        for line in fh:
            t = line.split()
            if len(t) < 35:
                continue
            parent.append(int(t[0])); leaf.append(int(t[1]) > 0)
            desc.append(np.array(t[2:34], dtype=np.int64).astype(np.uint8)); weight.append(float(t[34]))
    return k, L, np.array(parent, np.int32), np.array(leaf, bool), np.stack(desc), np.array(weight, np.float64)


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "request":
        os.makedirs(OUT, exist_ok=True)
        p = os.path.join(OUT, "dbow2_request.txt")
        np.savetxt(p, request_descriptors(), fmt="%d")
        print("wrote", p)
        return
    if len(sys.argv) == 4 and sys.argv[1] == "pin":
        k, L, parent, leaf, desc, weight = load_vocabulary_text(sys.argv[2])
        ans = np.loadtxt(sys.argv[3], ndmin=2)
        req = request_descriptors()
        if len(ans) != len(req):
            sys.exit("expected %d answer lines, got %d" % (len(req), len(ans)))
        word_id = np.full(len(parent), -1, np.int64)
        word_id[leaf] = np.arange(int(leaf.sum()))                      # words are numbered in node order (loadFromTextFile)
        leaf_of_word = np.nonzero(leaf)[0]
        keep = {0}
        children = {}
        for i in range(1, len(parent)):
            children.setdefault(int(parent[i]), []).append(i)
        for wid in ans[:, 0].astype(np.int64):                          # the path DBoW2 took: the word's ancestors, with all their siblings
            n = int(leaf_of_word[wid])
            while n != 0:
                keep.update(children[int(parent[n])])
                n = int(parent[n])
        ids = np.array(sorted(keep), np.int64)
        np.savez_compressed(os.path.join(OUT, "dbow2_pin.npz"), k=np.int32(k), L=np.int32(L), node_id=ids, parent=parent[ids], is_leaf=leaf[ids],
                            desc=desc[ids], weight=weight[ids], word_id=word_id[ids], request=req, answer_word=ans[:, 0].astype(np.int64),
                            answer_node=ans[:, 1].astype(np.int64), answer_weight=ans[:, 2].astype(np.float64), levelsup=np.int32(4),
                            vocabulary_nodes=np.int64(len(parent)))
        print("wrote dbow2_pin.npz: %d of %d nodes kept" % (len(ids), len(parent)))
        return
    sys.exit(__doc__)


if __name__ == "__main__":
    main()
