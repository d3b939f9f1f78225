"""Host-side mirror of the reference's Vocabulary wrapper (include/Vocabulary.h, src/Vocabulary.cpp) for binary
descriptors: DBoW2 tree storage, the text loader and transform(); the per-descriptor tree descent runs on the GPU
(afv_bow_transform).  DBoW2 itself is an empty submodule in the reference: format and semantics follow upstream DBoW2
(TemplatedVocabulary::loadFromTextFile / transform) — parity unpinned."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ptr


class Vocabulary:
    def __init__(self, k, L, parent, node_desc, weight, is_leaf, ctx=None):
        """nodes in DBoW2 id order (node 0 = root); children lists keep insertion order (= id order per parent)"""
        self.k, self.L = int(k), int(L)
        parent = np.asarray(parent, np.int32)
        n = len(parent)
        order = np.argsort(parent[1:], kind="stable") + 1          # children grouped by parent, ascending id inside
        counts = np.bincount(parent[1:], minlength=n)
        self.child_ptr = np.zeros(n + 1, np.int32)
        self.child_ptr[1:] = np.cumsum(counts)
        self.child_idx = np.ascontiguousarray(order, np.int32)
        node_desc = np.asarray(node_desc)
        # binary descriptors (uint8 rows: ORB32, AKAZE61, BRISK48 ...) or float ones (float32 rows of 64 / 128 / 256: SIFT128, SURF64, KAZE64,
        # R2D2 - the non-binary cases of Vocabulary::transform, Vocabulary.cpp:158-187)
        self.is_float = node_desc.dtype.kind == "f"
        self.node_desc = np.ascontiguousarray(node_desc, np.float32 if self.is_float else np.uint8)
        self.weight = np.asarray(weight, np.float64)
        self.is_leaf = np.asarray(is_leaf, bool)
        # word ids are assigned to leaves in node order (loadFromTextFile)
        self.word_id = np.full(n, -1, np.int32)
        self.word_id[self.is_leaf] = np.arange(int(self.is_leaf.sum()), dtype=np.int32)
        self.ctx = ctx
        self._handle = None

    # ---- DBoW2 text format (ORBvoc.txt): "k L scoring weighting" then one line per node: parent isLeaf d0..d31 weight ----
    @classmethod
    def loadFromTextFile(cls, path, ctx=None):
        with open(path) as fh:
            k, L, _scoring, _weighting = [int(v) for v in fh.readline().split()[:4]]
            parent, leaf, desc, weight = [0], [False], [None], [0.0]
            nbytes = None
            for line in fh:
                t = line.split()
                if len(t) < 4:
                    continue
                if nbytes is None:
                    nbytes = len(t) - 3   # parent, isLeaf, F::L descriptor bytes (32 ORB, 61 AKAZE, 48 BRISK ...), weight
                if len(t) != nbytes + 3:
                    continue
                parent.append(int(t[0])); leaf.append(int(t[1]) > 0)
                desc.append(np.array(t[2:2 + nbytes], dtype=np.int64).astype(np.uint8)); weight.append(float(t[2 + nbytes]))
        desc[0] = np.zeros(nbytes or 32, np.uint8)
        return cls(k, L, parent, np.stack(desc), weight, leaf, ctx)

    def saveToTextFile(self, path):
        with open(path, "w") as fh:
            fh.write("%d %d 0 0\n" % (self.k, self.L))
            parent = np.zeros(len(self.weight), np.int32)
            for p in range(len(self.weight)):
                parent[self.child_idx[self.child_ptr[p]:self.child_ptr[p + 1]]] = p
            for i in range(1, len(self.weight)):
                fh.write("%d %d %s %r\n" % (parent[i], int(self.is_leaf[i]), " ".join(str(int(b)) for b in self.node_desc[i]),
                                            float(self.weight[i])))

    def size(self):
        return int(self.is_leaf.sum())

    def _device(self):
        if self._handle is None:
            from .extractor import Context
            self.ctx = self.ctx or Context()
            h = C.c_void_p()
            create = self.ctx.lib.afv_vocab_create_f32 if self.is_float else self.ctx.lib.afv_vocab_create
            rc = create(self.ctx.handle, self.k, self.L, len(self.weight), ptr(self.child_ptr), ptr(self.child_idx),
                        ptr(self.node_desc), self.node_desc.shape[1], C.byref(h))
            self.ctx.check(rc, "afv_vocab_create")
            self._handle = h
            stopped = np.ascontiguousarray(~(self.weight > 0), np.uint8)   # DBoW2 transform: a word enters the vectors only if(w > 0)
            stopped[~self.is_leaf] = 0
            if stopped.any():
                self.ctx.check(self.ctx.lib.afv_vocab_set_stopped(self.ctx.handle, h, ptr(stopped)), "afv_vocab_set_stopped")
        return self._handle

    def close(self):
        if self._handle is not None:
            self.ctx.lib.afv_vocab_destroy(self.ctx.handle, self._handle)
            self._handle = None

    def transform_nodes(self, descriptors, levelsup=4):
        descriptors = np.ascontiguousarray(descriptors, np.float32 if self.is_float else np.uint8)
        n = len(descriptors)
        leaf = np.zeros(max(n, 1), np.int32); nid = np.zeros(max(n, 1), np.int32)
        fn = self.ctx.lib.afv_bow_transform_f32 if self.is_float else self.ctx.lib.afv_bow_transform
        rc = fn(self.ctx.handle, self._device(), ptr(descriptors), n, int(levelsup), ptr(leaf), ptr(nid)) if n else 0
        if n == 0:
            self._device()
        self.ctx.check(rc, "afv_bow_transform")
        return leaf[:n], nid[:n]

    def transform(self, descriptors, levelsup=4):
        """Vocabulary::transform(mDescriptors, mBowVec, mFeatVec) (Vocabulary.cpp:156-206): returns
        (BowVector {word_id: weight}, L1-normalised (TF-IDF weighting, L1 scoring: the ORB-SLAM vocabulary settings),
         FeatureVector [(node_id, [feature indices ascending])] ascending by node id)."""
        leaf, nid = self.transform_nodes(descriptors, levelsup)
        return self.vectors_from_nodes(leaf, nid)

    def vectors_from_nodes(self, leaf, nid):
        """BowVector / FeatureVector from the per-descriptor (leaf node, node at level L - levelsup) the GPU descent returned"""
        w = self.weight[leaf]
        keep = w > 0                                   # stopped words are skipped (DBoW2 transform: if(w > 0))
        bow = {}
        for wid, ww in zip(self.word_id[leaf[keep]].tolist(), w[keep].tolist()):
            bow[wid] = bow.get(wid, 0.0) + ww          # BowVector::addWeight
        s = sum(abs(v) for v in bow.values())
        if s > 0:
            bow = {kk: vv / s for kk, vv in bow.items()}   # BowVector::normalize(L1)
        fv = {}
        for i in np.nonzero(keep)[0].tolist():
            fv.setdefault(int(nid[i]), []).append(i)   # FeatureVector::addFeature, features arrive in ascending order
        return dict(sorted(bow.items())), sorted(fv.items())

    # ---- synthetic vocabulary for tests (no ORBvoc.txt offline) ----
    @classmethod
    def random_float(cls, seed, k=6, L=3, ctx=None, dim=128):
        """a synthetic float vocabulary: node descriptors = non-negative unit vectors (SIFT-like)"""
        v = cls.random(seed, k, L, None, desc_bytes=dim)
        d = v.node_desc.astype(np.float32) ** 2
        d[0] = 1.0
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        d[0] = 0.0
        parent = np.zeros(len(v.weight), np.int32)
        for p in range(len(v.weight)):
            parent[v.child_idx[v.child_ptr[p]:v.child_ptr[p + 1]]] = p
        return cls(k, L, parent, d.astype(np.float32), v.weight, v.is_leaf, ctx)

    @classmethod
    def random(cls, seed, k=6, L=3, ctx=None, desc_bytes=32):
        from .synth import lcg_bytes, lcg_states
        parent, leaf = [0], [False]
        frontier = [0]
        for level in range(1, L + 1):
            nxt = []
            for p in frontier:
                for _ in range(k):
                    parent.append(p); leaf.append(level == L); nxt.append(len(parent) - 1)
            frontier = nxt
        n = len(parent)
        desc = lcg_bytes(seed, n * desc_bytes).reshape(n, desc_bytes)
        weight = 0.5 + (lcg_states(seed + 1, n) % 1000).astype(np.float64) / 500.0
        weight[(lcg_states(seed + 2, n) % 37) == 0] = 0.0   # a few stopped words
        desc[0] = 0; weight[0] = 0.0                        # the root carries no descriptor / weight
        return cls(k, L, parent, desc, weight, leaf, ctx)
