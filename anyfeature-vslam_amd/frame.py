"""Host-side mirror of the reference's Frame as far as the front end goes (include/Frame.h, src/Frame.cc:171-240,333-433): the
device-resident frame of the C-ABI (afv_frame_*).

In the reference a Frame is built once - ExtractFeatures (Frame.cc:242-259), UndistortKeyPoints (:403-433), AssignFeaturesToGrid
(:225-240) - and then read by every matcher of the tracking step.  `Frame` keeps that object on the GPU: `extract` returns the host
vectors the reference's members hold (mvKeys, mDescriptors, keyPtsSigma2 / Inf / Size) AND leaves keypoints, descriptors, the
per-feature scale data and the 64 x 48 grid in HBM; SearchByProjection / Fuse / SearchForInitialization / ComputeBoW /
SearchByBoW(KF, F) / the promotion to a keyframe then run against them without uploading the frame again.
Plumbing only: every method is one C-ABI call.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, FrameParams, ProjQueries, ptr

FRAME_GRID_COLS, FRAME_GRID_ROWS = 64, 48  # Frame.h:40-41


class Frame:
    def __init__(self, ctx, min_x=0.0, min_y=0.0, max_x=640.0, max_y=480.0, grid_cols=FRAME_GRID_COLS, grid_rows=FRAME_GRID_ROWS,
                 distorted=False, cap=0, desc_bytes=32, float_dim=0):
        """desc_bytes: size of one binary descriptor (32 ORB, 61 AKAZE, 48 BRISK ...; <= 64); float_dim > 0: float descriptors of that many
        floats instead (SIFT128, SURF64, KAZE64 ...: L2^2 distances) - the reference's matchers dispatch on the descriptor type
        (FeatureMatcher.cc:1508-1531); frames that are not 32-byte binary are filled with set_features"""
        self.ctx, self.lib = ctx, ctx.lib
        self.float_dim = int(float_dim)
        self.desc_bytes = 4 * self.float_dim if self.float_dim else int(desc_bytes)
        p = _lib.sized(FrameParams)
        p.min_x, p.min_y, p.max_x, p.max_y = float(min_x), float(min_y), float(max_x), float(max_y)
        p.grid_cols, p.grid_rows, p.distorted, p.cap = int(grid_cols), int(grid_rows), int(bool(distorted)), int(cap)
        p.desc_bytes = int(desc_bytes)
        p.float_dim = self.float_dim
        self.params = p
        h = C.c_void_p()
        ctx.check(self.lib.afv_frame_create(ctx.handle, C.byref(p), C.byref(h)), "afv_frame_create")
        self.handle = h
        self.sizeTolerance = np.float32(ctx.params.scale_factor)           # Frame.cc:73
        self.invSizeTolerance = np.float32(1.0) / self.sizeTolerance        # Frame.cc:74
        self.grid_inv_w = np.float32(grid_cols) / (np.float32(max_x) - np.float32(min_x))   # Frame.cc:201
        self.grid_inv_h = np.float32(grid_rows) / (np.float32(max_y) - np.float32(min_y))   # Frame.cc:202

    def close(self):
        if getattr(self, "handle", None) and getattr(self.ctx, "handle", None):
            self.lib.afv_frame_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def N(self):
        return int(self.lib.afv_frame_count(self.handle))

    # ---- Frame::Frame: extraction into the resident frame ----
    def extract(self, gray, host_outputs=True):
        """FeatureExtractor::operator() (FeatureExtractor.cpp:111-121) into the frame.  Returns (mvKeys, mDescriptors) like
        Context.extract, or N when host_outputs is False (the device copy only)."""
        gray = np.ascontiguousarray(gray, np.uint8)
        h, w = gray.shape
        if not host_outputs:
            self.ctx.check(self.lib.afv_frame_extract(self.handle, ptr(gray), w, h, gray.strides[0], None, None, 0, None), "afv_frame_extract")
            return self.N
        cap = self.ctx.cap
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self.ctx.check(self.lib.afv_frame_extract(self.handle, ptr(gray), w, h, gray.strides[0], ptr(kps), ptr(desc), cap, C.byref(n)),
                       "afv_frame_extract")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def set_features(self, kps, desc, sizes=None, u_right=None):
        """a frame whose features come from elsewhere (stereo rigs, another extractor, tests): kps (KP_DTYPE), desc [n, desc_bytes]"""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        if self.float_dim:
            desc = np.ascontiguousarray(desc, np.float32).reshape(-1, self.float_dim)
        else:
            desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, self.desc_bytes)
        sz = None if sizes is None else np.ascontiguousarray(sizes, np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        self.ctx.check(self.lib.afv_frame_set_features(self.handle, ptr(kps), ptr(desc), len(kps), ptr(sz), ptr(ur)), "afv_frame_set_features")

    def set_undistorted(self, x, y):
        """mvKeysUn of a `distorted` frame (cv::undistortPoints is the caller's, Frame.cc:403-433); builds the grid"""
        x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
        self.ctx.check(self.lib.afv_frame_set_undistorted(self.handle, ptr(x), ptr(y)), "afv_frame_set_undistorted")

    def grid(self):
        """(cell_ptr[cols * rows + 1], cell_idx[...]) of the device-built grid, cell = ix * rows + iy"""
        nc = self.params.grid_cols * self.params.grid_rows
        cp = np.zeros(nc + 1, np.int32)
        ci = np.zeros(max(self.N, 1), np.int32)
        self.ctx.check(self.lib.afv_frame_get_grid(self.handle, ptr(cp), ptr(ci)), "afv_frame_get_grid")
        return cp, ci[:cp[nc]]

    def device_views(self):
        """raw device pointers (ints): kps, desc, x, y, size, angle, n"""
        out = [C.c_void_p() for _ in range(7)]
        self.ctx.check(self.lib.afv_frame_device_ptrs(self.handle, *[C.byref(o) for o in out]), "afv_frame_device_ptrs")
        return dict(zip(("kps", "desc", "x", "y", "size", "angle", "n"), (o.value for o in out)))

    # ---- Frame::ComputeBoW ----
    def bow_transform_nodes(self, vocabulary, levelsup=4):
        """afv_frame_bow_transform alone: (leaf node, node at depth L - levelsup) per feature; the FeatureVector is built on the device and
        stays with the frame.  ComputeBoW = this + the host-side BowVector / FeatureVector containers"""
        n = self.N
        leaf = np.zeros(max(n, 1), np.int32); nid = np.zeros(max(n, 1), np.int32)
        nn = C.c_int32(0)
        self.ctx.check(self.lib.afv_frame_bow_transform(self.handle, vocabulary._device(), int(levelsup), ptr(leaf), ptr(nid), C.byref(nn)),
                       "afv_frame_bow_transform")
        self._nnodes = int(nn.value)
        return leaf[:n], nid[:n]

    def ComputeBoW(self, vocabulary, levelsup=4):
        """Frame::ComputeBoW (Frame.cc:397-401): returns (BowVector, FeatureVector) like Vocabulary.transform; the FeatureVector also
        stays on the device with the frame"""
        return vocabulary.vectors_from_nodes(*self.bow_transform_nodes(vocabulary, levelsup))

    def featvec(self):
        """the resident FeatureVector as [(node_id, [feature indices])]"""
        n, nn = self.N, getattr(self, "_nnodes", 0)
        ids = np.zeros(max(nn, 1), np.int32); sp = np.zeros(max(nn, 1) + 1, np.int32); idx = np.zeros(max(n, 1), np.int32)
        self.ctx.check(self.lib.afv_frame_get_featvec(self.handle, ptr(ids), ptr(sp), ptr(idx)), "afv_frame_get_featvec")
        return [(int(ids[k]), idx[sp[k]:sp[k + 1]].tolist()) for k in range(nn)]

    # ---- the projection searches ----
    def _queries(self, q, th, nnratio, mode, check_orientation, occupied=None, qref=None):
        """q: matcher.ProjectionQueries; qref = (DescriptorTable, slots, idx): the queries' descriptors as rows of a keyframe table"""
        s = _lib.sized(ProjQueries)
        keep = []
        s.nq = q.n
        if qref is None:
            if q.n and (q.descriptors.dtype.kind == "f") != bool(self.float_dim):
                raise ValueError("the queries must carry the frame's kind of descriptor")
            s.qdesc = ptr(q.descriptors); s.desc_bytes = q.descriptors.shape[1] * q.descriptors.itemsize if q.n else self.desc_bytes
        else:
            table, slots, idx = qref
            sl = np.ascontiguousarray(slots, np.int32); ix = np.ascontiguousarray(idx, np.int32)
            keep += [sl, ix]
            s.qref_table = table.handle; s.qref_slot = ptr(sl); s.qref_idx = ptr(ix); s.desc_bytes = 32
        s.qvalid = ptr(q.valid); s.qu = ptr(q.u); s.qv = ptr(q.v); s.qr = ptr(q.r)
        s.qmin_size = ptr(q.min_size); s.qmax_size = ptr(q.max_size); s.qangle = ptr(q.angles); s.qoccupies = ptr(q.occupies)
        s.q_ur = ptr(q.ur); s.q_er_max = ptr(q.er_max)
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        keep.append(occ)
        s.occupied = ptr(occ)
        s.th_high = float(th); s.nnratio = float(nnratio); s.check_orientation = int(bool(check_orientation)); s.mode = int(mode)
        return s, keep

    def SearchByProjection(self, matcher, queries, last_frame=False, occupied=None, qref=None):
        """matching core of FeatureMatcher::SearchByProjection(F, vpMapPoints, th) (FeatureMatcher.cc:73-154) / (CurrentFrame, LastFrame)
        (:1291-1402) against the resident frame; matcher supplies TH_HIGH / mfNNratio / mbCheckOrientation"""
        s, keep = self._queries(queries, matcher.TH_HIGH, matcher.mfNNratio, _lib.PROJ_LASTFRAME if last_frame else _lib.PROJ_LOCALMAP,
                                matcher.mbCheckOrientation, occupied, qref)
        n = self.N
        out = np.full(max(n, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        self.ctx.check(self.lib.afv_frame_match_projection(self.handle, C.byref(s), ptr(out), ptr(nm)), "afv_frame_match_projection")
        return out[:n].copy(), int(nm[0])

    def Fuse(self, matcher, queries, use_inf_gate=True):
        s, keep = self._queries(queries, matcher.TH_LOW, matcher.mfNNratio, 0, False)
        out = np.full(max(queries.n, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        self.ctx.check(self.lib.afv_frame_match_fuse(self.handle, C.byref(s), int(bool(use_inf_gate)), ptr(out), ptr(nm)), "afv_frame_match_fuse")
        return out[:queries.n].copy(), int(nm[0])

    def SearchForInitialization(self, matcher, F2, vbPrevMatched, windowSize=100.0):
        """SearchForInitialization(F1 = self, F2, vbPrevMatched, vnMatches12, windowSize) (FeatureMatcher.cc:399-557) between two resident
        frames; vbPrevMatched [N1, 2] float32 (not refreshed here: the caller owns it, :551-553)"""
        pm = np.ascontiguousarray(vbPrevMatched, np.float32).reshape(-1, 2)
        px, py = np.ascontiguousarray(pm[:, 0]), np.ascontiguousarray(pm[:, 1])
        n1 = self.N
        out = np.full(max(n1, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        self.ctx.check(self.lib.afv_frame_match_initialization(self.handle, F2.handle, ptr(px), ptr(py), float(windowSize), float(matcher.TH_LOW),
                                                               float(matcher.mfNNratio), int(matcher.mbCheckOrientation), ptr(out), ptr(nm)),
                       "afv_frame_match_initialization")
        return out[:n1].copy(), int(nm[0])
