"""Host-side mirror of the reference's FeatureMatcher (include/FeatureMatcher.h:36-118, src/FeatureMatcher.cc).

The reference walks KeyFrame / Frame / MapPoint objects; a drop-in flattens them (under their mutexes) into the
arrays below — that is what `FeatureView` holds — and calls the C-ABI.  Method names, thresholds and return values
follow the reference: SearchByBoW(KF,KF) -> (vpMatches12, nmatches), SearchByBoW(KF,F) -> (vpMapPointMatches, n),
SearchForTriangulation -> (vMatchedPairs, n).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MatchJob, ProjJob, TriJob, ptr


def DescriptorDistance_orb32(a, b):
    """Feature_orb32.cpp:67-84 (host utility; the kernels use xor + v_bcnt on the GPU)."""
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return float(_lib.load().afv_hamming256(ptr(a), ptr(b)))


def DescriptorDistance_sift128(a, b):
    """Feature_sift128.cpp:132-134 (and the surf64 / kaze64 / r2d2 twins): cv::norm(a, b, NORM_L2SQR) narrowed to Descriptor_Distance_Type =
    float, in cv::norm's order - float differences, squares in double, 4-way partial sums added to one accumulator (host utility; the
    kernels evaluate the same expression per candidate)"""
    a = np.ascontiguousarray(a, np.float32).ravel(); b = np.ascontiguousarray(b, np.float32).ravel()
    v = (a - b).astype(np.float64)
    n4 = len(v) // 4 * 4
    q = v[:n4].reshape(-1, 4) ** 2
    groups = ((q[:, 0] + q[:, 1]) + q[:, 2]) + q[:, 3]
    s = np.float64(0.0)
    for g in groups.tolist() + (v[n4:] ** 2).tolist():   # sequential: the summation order is part of the result
        s = s + np.float64(g)
    return float(np.float32(s))


def ComputeDistinctiveDescriptors(ctx, descriptor_sets):
    """MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:279-349) for a batch of map points: descriptor_sets = one [N_s, bytes] uint8 array per
    map point (the descriptors of its observations, in observation order); returns (best index per set | -1 for an empty one, its median)"""
    if any(np.asarray(d).dtype.kind == "f" for d in descriptor_sets):  # float descriptors: L2^2 distances, the median comes back as a float
        sets = [np.ascontiguousarray(d, np.float32) for d in descriptor_sets]
        dim = next((d.reshape(len(d), -1).shape[1] for d in sets if len(d)), 128)
        sets = [d.reshape(len(d), dim) if len(d) else np.zeros((0, dim), np.float32) for d in sets]
        ptrs = np.zeros(len(sets) + 1, np.int32)
        ptrs[1:] = np.cumsum([len(d) for d in sets])
        flat = np.ascontiguousarray(np.concatenate(sets) if sets and ptrs[-1] else np.zeros((0, dim), np.float32))
        best = np.zeros(max(len(sets), 1), np.int32); med = np.zeros(max(len(sets), 1), np.float32)
        ctx.check(ctx.lib.afv_distinctive_descriptors_f32(ctx.handle, ptr(flat), dim, ptr(ptrs), len(sets), ptr(best), ptr(med)),
                  "afv_distinctive_descriptors_f32")
        return best[:len(sets)].copy(), med[:len(sets)].copy()
    sets = [np.ascontiguousarray(d, np.uint8) for d in descriptor_sets]
    nbytes = next((d.reshape(len(d), -1).shape[1] for d in sets if len(d)), 32)
    sets = [d.reshape(len(d), nbytes) if len(d) else np.zeros((0, nbytes), np.uint8) for d in sets]
    ptrs = np.zeros(len(sets) + 1, np.int32)
    ptrs[1:] = np.cumsum([len(d) for d in sets])
    flat = np.ascontiguousarray(np.concatenate([d.reshape(-1, nbytes) for d in sets]) if sets and ptrs[-1] else np.zeros((0, nbytes), np.uint8))
    best = np.zeros(max(len(sets), 1), np.int32); med = np.zeros(max(len(sets), 1), np.int32)
    ctx.check(ctx.lib.afv_distinctive_descriptors(ctx.handle, ptr(flat), nbytes, ptr(ptrs), len(sets), ptr(best), ptr(med)), "afv_distinctive_descriptors")
    return best[:len(sets)].copy(), med[:len(sets)].copy()


def _descriptor_rows(descriptors):
    """mDescriptors as the matchers take it: uint8 rows (binary descriptors, Hamming) or float32 rows (SIFT128 / SURF64 / KAZE64 / R2D2 ...:
    cv::norm(a, b, NORM_L2SQR), Feature_sift128.cpp:132-134) - FeatureMatcher::DescriptorDistance dispatches on the type
    (FeatureMatcher.cc:1508-1531)"""
    d = np.asarray(descriptors)
    return np.ascontiguousarray(d, np.float32 if d.dtype.kind == "f" else np.uint8)


class FeatureView:
    """Flattened view of one KeyFrame / Frame for matching.

    descriptors  N x desc_bytes uint8           (KeyFrame::mDescriptors, const after construction KeyFrame.h:190)
    featvec      None (brute force) or list of (node_id, [feature indices]) ascending by node id
                 (DBoW2::FeatureVector, KeyFrame.h:194)
    valid        N uint8: map point exists and !isBad()  (for SearchForTriangulation: "has a map point")
    angles       N float32 degrees (mvKeysUn[i].angle)
    pts          N x 2 float32 (mvKeysUn[i].pt), sigma2 N float32 (GetKeyPt1DSigma2) — triangulation only
    u_right      None (monocular keyframe) or N float32 (KeyFrame::mvuRight, < 0: no right-image match) — triangulation only
    """

    def __init__(self, descriptors, featvec=None, valid=None, angles=None, pts=None, sigma2=None, u_right=None):
        self.descriptors = _descriptor_rows(descriptors)   # uint8 rows (Hamming) or float32 rows (L2^2)
        if self.descriptors.ndim != 2:
            self.descriptors = self.descriptors.reshape(0, 32)
        self.N = self.descriptors.shape[0]
        self.featvec = featvec
        self.valid = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        self.angles = None if angles is None else np.ascontiguousarray(angles, np.float32)
        self.pts = None if pts is None else np.ascontiguousarray(pts, np.float32)
        self.sigma2 = None if sigma2 is None else np.ascontiguousarray(sigma2, np.float32)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        self._csr = None

    def csr(self):
        if self._csr is None:
            if self.featvec is None:
                self._csr = (None, None, None, 0)
            else:
                ids = np.array([n for n, _ in self.featvec], np.int32)
                ptrs = np.zeros(len(self.featvec) + 1, np.int32)
                for i, (_, idx) in enumerate(self.featvec):
                    ptrs[i + 1] = ptrs[i] + len(idx)
                flat = (np.concatenate([np.asarray(idx, np.int32) for _, idx in self.featvec])
                        if self.featvec else np.zeros(0, np.int32))
                self._csr = (ids, ptrs, np.ascontiguousarray(flat, np.int32), len(self.featvec))
        return self._csr


class FrameGridView:
    """What the projection-guided matchers read from a Frame: mvKeysUn (pts, angles), keyPtsSize, mDescriptors, the
    occupancy of F.pts (point present with NumberOfObservations() > 0) and the feature grid parameters
    (Frame.cc:100-101: mnMinX, mnMinY, mfGridElementWidthInv/HeightInv; Frame.h:40-41: 64 x 48 cells)."""

    def __init__(self, descriptors, pts, sizes, angles=None, occupied=None, min_x=0.0, min_y=0.0, max_x=640.0, max_y=480.0,
                 grid_cols=64, grid_rows=48, size_tolerance=1.2, inf=None, u_right=None):
        self.descriptors = _descriptor_rows(descriptors)
        self.N = self.descriptors.shape[0]
        pts = np.asarray(pts, np.float32).reshape(-1, 2)
        self.x = np.ascontiguousarray(pts[:, 0]); self.y = np.ascontiguousarray(pts[:, 1])
        self.sizes = np.ascontiguousarray(sizes, np.float32)
        self.angles = None if angles is None else np.ascontiguousarray(angles, np.float32)
        self.occupied = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        self.inf = None if inf is None else np.ascontiguousarray(inf, np.float32)  # GetKeyPt1DInf (Fuse)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)  # mvuRight (stereo frames; None: mono)
        self.min_x, self.min_y = np.float32(min_x), np.float32(min_y)
        self.grid_inv_w = np.float32(grid_cols) / (np.float32(max_x) - np.float32(min_x))
        self.grid_inv_h = np.float32(grid_rows) / (np.float32(max_y) - np.float32(min_y))
        self.grid_cols, self.grid_rows = grid_cols, grid_rows
        self.sizeTolerance = np.float32(size_tolerance)               # Frame.cc:73 extractor->GetScaleFactor()
        self.invSizeTolerance = np.float32(1.0) / self.sizeTolerance   # Frame.cc:74


class ProjectionQueries:
    """Projected map points / last-frame keypoints in the reference's iteration order: descriptor, projected (u, v),
    window radius r, admissible keyPtsSize band, validity, angle (last-frame mode), occupies (observations > 0)."""

    def __init__(self, descriptors, u, v, r, min_size, max_size, valid=None, angles=None, occupies=None, ur=None, er_max=None):
        self.descriptors = _descriptor_rows(descriptors)
        self.n = self.descriptors.shape[0]
        f = lambda a: np.ascontiguousarray(a, np.float32)
        self.u, self.v, self.r, self.min_size, self.max_size = f(u), f(v), f(r), f(min_size), f(max_size)
        self.valid = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        self.angles = None if angles is None else f(angles)
        self.occupies = None if occupies is None else np.ascontiguousarray(occupies, np.uint8)
        # stereo frames: the projected right-image coordinate (mTrackProjXR / u - mbf * invzc / Fuse's ur) and, for the projection
        # searches, the gate on |ur - mvuRight| (r * trackSigma, FeatureMatcher.cc:117; the window radius, :1371)
        self.ur = None if ur is None else f(ur)
        self.er_max = None if er_max is None else f(er_max)


class FeatureMatcher:
    """FeatureMatcher(nnratio=0.6, checkOri=true) (FeatureMatcher.h:41)."""
    TH_LOW = 0.0   # FeatureMatcher.cc:56-59 (all four are set from matchingTh, :1533-1545)
    TH_HIGH = 0.0
    descDistTh_high_reloc = 0.0
    descDistTh_low_reloc = 0.0
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True, ctx=None):
        from .extractor import Context
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        self.ctx = ctx or Context()
        self.lib = self.ctx.lib

    @classmethod
    def setDescriptorDistanceThresholds(cls, settings):
        """settings: matchingTh float, dict with 'FeatureMatcher.matchingTh', or YAML path (FeatureMatcher.cc:1533-1545)."""
        if isinstance(settings, str):
            import yaml
            with open(settings) as fh:
                text = fh.read()
            if text.startswith("%YAML"):
                text = text.split("\n", 1)[1]
            settings = yaml.safe_load(text)
        th = float(settings["FeatureMatcher.matchingTh"]) if isinstance(settings, dict) else float(settings)
        cls.TH_LOW = th
        cls.TH_HIGH = th
        cls.descDistTh_low_reloc = th
        cls.descDistTh_high_reloc = th

    @staticmethod
    def DescriptorDistance(a, b):
        """FeatureMatcher::DescriptorDistance (FeatureMatcher.cc:1508-1531): dispatched on the descriptor type - float rows: L2^2; 32-byte rows:
        the ORB popcount; other binary rows: Hamming over their bytes"""
        a = np.asarray(a)
        if a.dtype.kind == "f":
            return DescriptorDistance_sift128(a, b)
        if a.size == 32:
            return DescriptorDistance_orb32(a, b)
        return float(np.unpackbits(np.bitwise_xor(np.ascontiguousarray(a, np.uint8).ravel(), np.ascontiguousarray(b, np.uint8).ravel())).sum())

    def _job(self, v1, v2, mode, keep):
        j = MatchJob()
        j.desc1 = ptr(v1.descriptors); j.n1 = v1.N
        j.desc2 = ptr(v2.descriptors); j.n2 = v2.N
        j.desc_bytes = v1.descriptors.shape[1] if v1.N else (v2.descriptors.shape[1] if v2.N else 32)
        is_float = (v1.N and v1.descriptors.dtype.kind == "f") or (v2.N and v2.descriptors.dtype.kind == "f")
        if is_float:  # float descriptors: rows of desc_bytes / 4 floats, L2^2 distances (AFV_MATCH_FLOAT32)
            if v1.N and v2.N and (v1.descriptors.dtype != v2.descriptors.dtype or v1.descriptors.shape[1] != v2.descriptors.shape[1]):
                raise ValueError("both sides must carry the same kind of descriptor")
            j.desc_bytes *= 4
            mode |= _lib.MATCH_FLOAT32
        i1, p1, f1, n1 = v1.csr(); i2, p2, f2, n2 = v2.csr()
        j.node_id1 = ptr(i1); j.seg_ptr1 = ptr(p1); j.seg_idx1 = ptr(f1); j.nnodes1 = n1
        j.node_id2 = ptr(i2); j.seg_ptr2 = ptr(p2); j.seg_idx2 = ptr(f2); j.nnodes2 = n2
        j.valid1 = ptr(v1.valid); j.valid2 = ptr(v2.valid)
        j.angle1 = ptr(v1.angles); j.angle2 = ptr(v2.angles)
        j.th_low = self.TH_LOW; j.nnratio = self.mfNNratio
        j.check_orientation = int(self.mbCheckOrientation); j.mode = mode
        keep.extend([i1, p1, f1, i2, p2, f2])
        return j

    def _run_bow(self, pairs, mode):
        keep = []
        jobs = (MatchJob * len(pairs))(*[self._job(a, b, mode, keep) for a, b in pairs])
        nouts = [(b.N if mode == _lib.MATCH_KF_FRAME else a.N) for a, b in pairs]
        out = np.full(max(sum(nouts), 1), -1, np.int32)
        nm = np.zeros(len(pairs), np.int32)
        self.ctx.check(self.lib.afv_match_bow(self.ctx.handle, jobs, len(pairs), ptr(out), ptr(nm)), "afv_match_bow")
        res, o = [], 0
        for k, n in enumerate(nouts):
            res.append((out[o:o + n].copy(), int(nm[k])))
            o += n
        return res

    def SearchByBoW(self, pKF, other, frame=False):
        """SearchByBoW(KF1,KF2) (FeatureMatcher.cc:561-660) -> (vpMatches12[N1] = idx in KF2 | -1, nMatches);
        with frame=True: SearchByBoW(KF,Frame) (:186-283) -> (vpMapPointMatches[F.N] = idx in KF | -1, nMatches)."""
        mode = _lib.MATCH_KF_FRAME if frame else _lib.MATCH_KF_KF
        return self._run_bow([(pKF, other)], mode)[0]

    def SearchByBoW_batch(self, pairs, frame=False):
        mode = _lib.MATCH_KF_FRAME if frame else _lib.MATCH_KF_KF
        return self._run_bow(list(pairs), mode)

    def SearchForTriangulation(self, pKF1, pKF2, F12, epipole, bOnlyStereo=False):
        """FeatureMatcher.cc:662-790 -> (vMatchedPairs [(idx1, idx2)...] ascending idx1, nMatches).
        pKF*.valid = has-map-point masks; epipole = projection of camera 1's centre in image 2 (:669-675); pKF*.u_right =
        mvuRight of stereo keyframes (:705, :727)."""
        keep = []
        t = _lib.sized(TriJob)
        t.bow = self._job(pKF1, pKF2, _lib.MATCH_KF_KF, keep)
        x1 = np.ascontiguousarray(pKF1.pts[:, 0]); y1 = np.ascontiguousarray(pKF1.pts[:, 1])
        x2 = np.ascontiguousarray(pKF2.pts[:, 0]); y2 = np.ascontiguousarray(pKF2.pts[:, 1])
        t.x1 = ptr(x1); t.y1 = ptr(y1); t.x2 = ptr(x2); t.y2 = ptr(y2); t.sigma2_2 = ptr(pKF2.sigma2)
        F = np.asarray(F12, np.float32).reshape(9)
        for i in range(9):
            t.F12[i] = float(F[i])
        t.ex, t.ey = float(epipole[0]), float(epipole[1])
        t.u_right1 = ptr(pKF1.u_right); t.u_right2 = ptr(pKF2.u_right); t.only_stereo = int(bool(bOnlyStereo))
        out = np.full(max(pKF1.N, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        jobs = (TriJob * 1)(t)
        self.ctx.check(self.lib.afv_match_triangulation(self.ctx.handle, jobs, 1, ptr(out), ptr(nm)), "afv_match_triangulation")
        out = out[:pKF1.N]
        pairs = [(int(i), int(out[i])) for i in np.nonzero(out >= 0)[0]]
        return pairs, int(nm[0])

    def _proj_job(self, F, queries):
        j = _lib.sized(ProjJob)
        j.desc = ptr(F.descriptors); j.n = F.N; j.desc_bytes = F.descriptors.shape[1] if F.N else 32
        if F.descriptors.dtype.kind == "f" or queries.descriptors.dtype.kind == "f":  # float descriptors: L2^2 distances
            if F.descriptors.dtype != queries.descriptors.dtype or (F.N and queries.n and F.descriptors.shape[1] != queries.descriptors.shape[1]):
                raise ValueError("frame and queries must carry the same kind of descriptor")
            j.float_dim = F.descriptors.shape[1] if F.N else queries.descriptors.shape[1]
            j.desc_bytes = 4 * j.float_dim
        j.x = ptr(F.x); j.y = ptr(F.y); j.size = ptr(F.sizes); j.angle = ptr(F.angles); j.occupied = ptr(F.occupied)
        j.inf = ptr(F.inf)
        j.min_x = float(F.min_x); j.min_y = float(F.min_y); j.grid_inv_w = float(F.grid_inv_w); j.grid_inv_h = float(F.grid_inv_h)
        j.grid_cols = F.grid_cols; j.grid_rows = F.grid_rows
        j.nq = queries.n; j.qdesc = ptr(queries.descriptors); j.qvalid = ptr(queries.valid)
        j.qu = ptr(queries.u); j.qv = ptr(queries.v); j.qr = ptr(queries.r)
        j.qmin_size = ptr(queries.min_size); j.qmax_size = ptr(queries.max_size)
        j.qangle = ptr(queries.angles); j.qoccupies = ptr(queries.occupies)
        j.nnratio = self.mfNNratio
        j.size_tol = float(F.sizeTolerance); j.inv_size_tol = float(F.invSizeTolerance)
        j.u_right = ptr(getattr(F, "u_right", None)); j.q_ur = ptr(queries.ur); j.q_er_max = ptr(queries.er_max)
        return j

    def Fuse(self, pKF, queries):
        """matching core of Fuse(pKF, vpMapPoints, th) (FeatureMatcher.cc:794-940; stereo keyframes: pKF.u_right + queries.ur): returns (bestIdx per map point
        | -1, nFused candidates); the Replace / AddObservation surgery (:918-936) stays with the caller."""
        j = self._proj_job(pKF, queries)
        j.th_high = self.TH_LOW
        out = np.full(max(queries.n, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        jobs = (ProjJob * 1)(j)
        self.ctx.check(self.lib.afv_match_fuse(self.ctx.handle, jobs, 1, ptr(out), ptr(nm)), "afv_match_fuse")
        return out[:queries.n].copy(), int(nm[0])

    def _run_projection(self, F, queries, th, mode, check_orientation, stereo=True):
        j = self._proj_job(F, queries)
        if not stereo:  # the relocalisation / Sim3 flavours have no mvuRight branch (FeatureMatcher.cc:287-397, :1404-1506)
            j.u_right = None
        j.th_high = float(th)
        j.check_orientation = int(bool(check_orientation)); j.mode = mode
        out = np.full(max(F.N, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        jobs = (ProjJob * 1)(j)
        self.ctx.check(self.lib.afv_match_projection(self.ctx.handle, jobs, 1, ptr(out), ptr(nm)), "afv_match_projection")
        return out[:F.N].copy(), int(nm[0])

    def SearchByProjection(self, F, queries, last_frame=False):
        """matching core of SearchByProjection(F, vpMapPoints, radiusTh) (FeatureMatcher.cc:73-154) or, with
        last_frame=True, of SearchByProjection(CurrentFrame, LastFrame, ...) (:1291-1402).  Stereo frames: F.u_right + queries.ur /
        queries.er_max (:114-119, :1367-1372).  Returns
        (assign[F.N] = query index now stored in F.pts[i] | -1, nmatches)."""
        return self._run_projection(F, queries, self.TH_HIGH, _lib.PROJ_LASTFRAME if last_frame else _lib.PROJ_LOCALMAP,
                                    self.mbCheckOrientation)

    def SearchByProjection_reloc(self, CurrentFrame, queries, useHighMatchingThreshold=False):
        """matching core of SearchByProjection(CurrentFrame, pKF, sAlreadyFound, radiusTh, useHigh) (FeatureMatcher.cc:1404-1506):
        queries = pKF's map points in keyframe feature order (valid = good, not in sAlreadyFound, projects inside the image
        and the distance band; angles = pKF->mvKeysUn[i].angle); CurrentFrame.occupied = pts[i] != NULL."""
        th = self.descDistTh_high_reloc if useHighMatchingThreshold else self.descDistTh_low_reloc
        return self._run_projection(CurrentFrame, queries, th, _lib.PROJ_LASTFRAME, self.mbCheckOrientation, stereo=False)

    def SearchByProjection_sim3(self, pKF, queries):
        """matching core of SearchByProjection(pKF, Scw, vpPoints, vpMatched, radiusTh) (FeatureMatcher.cc:287-397):
        pKF.occupied = vpMatched[i] != NULL; size band = predictedSize / , * sizeTolerance (:365-367); no orientation check;
        accept bestDist <= TH_LOW (:380).  assign[i] = index into vpPoints now stored in vpMatched[i]."""
        return self._run_projection(pKF, queries, self.TH_LOW, _lib.PROJ_LASTFRAME, False, stereo=False)

    def Fuse_sim3(self, pKF, queries):
        """matching core of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (FeatureMatcher.cc:942-1064): as Fuse without the
        reprojection gate."""
        j = self._proj_job(pKF, queries)
        j.inf = None
        j.u_right = None
        j.th_high = self.TH_LOW
        out = np.full(max(queries.n, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        jobs = (ProjJob * 1)(j)
        self.ctx.check(self.lib.afv_match_fuse(self.ctx.handle, jobs, 1, ptr(out), ptr(nm)), "afv_match_fuse")
        return out[:queries.n].copy(), int(nm[0])

    def SearchBySim3(self, pKF1, queries1, pKF2, queries2):
        """SearchBySim3 (FeatureMatcher.cc:1066-1287): queries1 = KF1's map points (one per KF1 feature) projected into KF2,
        queries2 the reverse.  Returns (match12[N1] = KF2 feature index | -1, nFound)."""
        j12 = self._proj_job(pKF2, queries1); j21 = self._proj_job(pKF1, queries2)
        j12.th_high = j21.th_high = self.TH_HIGH
        out = np.full(max(queries1.n, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        self.ctx.check(self.lib.afv_match_sim3(self.ctx.handle, C.byref(j12), C.byref(j21), ptr(out), ptr(nm)), "afv_match_sim3")
        return out[:queries1.n].copy(), int(nm[0])

    def SearchForInitialization(self, F1_queries, F2, vbPrevMatched=None):
        """SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (FeatureMatcher.cc:399-557).
        F1_queries: ProjectionQueries over F1's features (valid = octave 0, u/v = vbPrevMatched, r = windowSize,
        min_size = 0, max_size = F1.maxKeyPtSize, angles).  Returns (vnMatches12, nMatches) and, when vbPrevMatched (N1 x 2)
        is given, refreshes it in place like :551-553."""
        j = self._proj_job(F2, F1_queries)
        j.th_high = self.TH_LOW
        j.check_orientation = int(self.mbCheckOrientation)
        out = np.full(max(F1_queries.n, 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        jobs = (ProjJob * 1)(j)
        self.ctx.check(self.lib.afv_match_initialization(self.ctx.handle, jobs, 1, ptr(out), ptr(nm)), "afv_match_initialization")
        out = out[:F1_queries.n].copy()
        if vbPrevMatched is not None:
            m = out >= 0
            vbPrevMatched[m, 0] = F2.x[out[m]]; vbPrevMatched[m, 1] = F2.y[out[m]]
        return out, int(nm[0])

    def match_l2(self, desc1, desc2, th_low, nnratio=None, valid1=None, valid2=None):
        """float descriptors (SIFT128 ...): brute force with SearchByBoW(KF,KF) control flow, distance =
        cv::norm(a,b,NORM_L2SQR) (Feature_sift128.cpp:132-134)."""
        desc1 = np.ascontiguousarray(desc1, np.float32); desc2 = np.ascontiguousarray(desc2, np.float32)
        v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        out = np.full(max(len(desc1), 1), -1, np.int32)
        nm = np.zeros(1, np.int32)
        rc = self.lib.afv_match_l2(self.ctx.handle, ptr(desc1), len(desc1), ptr(desc2), len(desc2), desc1.shape[1], ptr(v1), ptr(v2),
                                   float(th_low), float(self.mfNNratio if nnratio is None else nnratio), ptr(out), ptr(nm))
        self.ctx.check(rc, "afv_match_l2")
        return out[:len(desc1)].copy(), int(nm[0])

    def match_l2_pairs_device(self, desc, n, pair_a, pair_b, th_low, nnratio=None, match=None, nmatches=None, stream=None):
        """match_l2 over a device-resident table: desc (nsets, cap, dim) float32 CUDA tensor (dim 64 / 128), n (nsets,) int32, pair
        lists int32; returns (match (npairs, cap) int32, nmatches (npairs,) int32) on the device (asynchronous)."""
        import torch
        cap, dim = desc.shape[1], desc.shape[2]
        npairs = pair_a.numel()
        if match is None:
            match = torch.empty((npairs, cap), dtype=torch.int32, device=desc.device)
        if nmatches is None:
            nmatches = torch.empty((npairs,), dtype=torch.int32, device=desc.device)
        # raw pointers cross the C boundary: a wrong dtype / a strided view / a tensor on another device would be read as garbage jobs
        for name, t_, dt in (("desc", desc, torch.float32), ("n", n, torch.int32), ("pair_a", pair_a, torch.int32), ("pair_b", pair_b, torch.int32),
                             ("match", match, torch.int32), ("nmatches", nmatches, torch.int32)):
            if t_.dtype != dt or not t_.is_cuda or t_.device != desc.device or not t_.is_contiguous():
                raise TypeError("match_l2_pairs_device: %s must be a contiguous %s tensor on %s" % (name, dt, desc.device))
        if pair_b.numel() != npairs or match.numel() < npairs * cap or nmatches.numel() < npairs:
            raise ValueError("match_l2_pairs_device: pair lists / outputs of different lengths")
        rc = _lib.launch_ordered(self.ctx, desc.device, stream, lambda s: self.lib.afv_match_l2_pairs_device(
            self.ctx.handle, desc.data_ptr(), n.data_ptr(), cap, dim, pair_a.data_ptr(), pair_b.data_ptr(), npairs, float(th_low),
            float(self.mfNNratio if nnratio is None else nnratio), match.data_ptr(), nmatches.data_ptr(), s),
            (desc, n, pair_a, pair_b, match, nmatches))
        self.ctx.check(rc, "afv_match_l2_pairs_device")
        return match, nmatches

    def match_pairs_device(self, desc, kps, n, pair_a, pair_b, th_low=None, check_orientation=None, match=None, nmatches=None,
                           stream=None):
        """device-resident brute-force SearchByBoW(KF,KF) over a descriptor table (torch CUDA tensors)."""
        import torch
        nsets, cap = desc.shape[0], desc.shape[1]
        npairs = pair_a.numel()
        if match is None:
            match = torch.empty((npairs, cap), dtype=torch.int32, device=desc.device)
        if nmatches is None:
            nmatches = torch.empty((npairs,), dtype=torch.int32, device=desc.device)
        co = self.mbCheckOrientation if check_orientation is None else check_orientation
        rc = _lib.launch_ordered(self.ctx, desc.device, stream, lambda s: self.lib.afv_match_bruteforce_pairs_device(
            self.ctx.handle, desc.data_ptr(), kps.data_ptr() if kps is not None else None, n.data_ptr(), nsets, cap,
            pair_a.data_ptr(), pair_b.data_ptr(), npairs, float(self.TH_LOW if th_low is None else th_low), self.mfNNratio,
            int(bool(co)), match.data_ptr(), nmatches.data_ptr(), s), (desc, kps, n, pair_a, pair_b, match, nmatches))
        self.ctx.check(rc, "afv_match_bruteforce_pairs_device")
        return match, nmatches
