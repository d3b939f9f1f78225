"""Keyframe descriptor table resident in HBM + RCCL communicator (mirror of the afv_table_* / afv_comm_* C-ABI).

Reference call shape: LoopClosing::ComputeSim3 (LoopClosing.cc:255-281) / Tracking::Relocalization (Tracking.cc:1162-1182)
run SearchByBoW once per candidate keyframe, LocalMapping::CreateNewMapPoints (LocalMapping.cc:238-297)
SearchForTriangulation per neighbour; the descriptors they re-read are const after keyframe construction
(KeyFrame.h:190).  `DescriptorTable` keeps them on the device; batches of (slot a, slot b) jobs run against it.
Plumbing only: every method is one C-ABI call.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import TableTriJob, ptr


def _i32(a):
    return np.ascontiguousarray(a, np.int32)


class Communicator:
    """afv_comm: RCCL communicator, one process per GPU.  `exchange_id(id_bytes_or_None) -> id_bytes` moves rank 0's
    128-byte id to every rank through any host channel (torch.distributed object broadcast, a file, a socket)."""

    def __init__(self, ctx, rank, world, exchange_id):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank, self.world = int(rank), int(world)
        ident = None
        if self.rank == 0:
            buf = (C.c_uint8 * 128)()
            ctx.check(self.lib.afv_comm_unique_id(buf), "afv_comm_unique_id")
            ident = bytes(buf)
        ident = exchange_id(ident)
        assert isinstance(ident, (bytes, bytearray)) and len(ident) == 128
        h = C.c_void_p()
        ctx.check(self.lib.afv_comm_create(ctx.handle, (C.c_uint8 * 128).from_buffer_copy(ident), self.world, self.rank, C.byref(h)),
                  "afv_comm_create")
        self.handle = h

    def broadcast(self, tensor, root=0, stream=None):
        """in-place broadcast of a contiguous CUDA tensor (ncclBroadcast over xGMI), asynchronous on the context's stream"""
        assert tensor.is_cuda and tensor.is_contiguous()
        self.ctx.check(self.lib.afv_comm_broadcast(self.handle, tensor.data_ptr(), tensor.numel() * tensor.element_size(), int(root), stream),
                       "afv_comm_broadcast")

    def allgather(self, send, recv, stream=None):
        assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()
        nbytes = send.numel() * send.element_size()
        assert recv.numel() * recv.element_size() == nbytes * self.world
        self.ctx.check(self.lib.afv_comm_allgather(self.handle, send.data_ptr(), recv.data_ptr(), nbytes, stream), "afv_comm_allgather")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.afv_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(n_units, rank, world):
    """afv_shard_range: contiguous block partition (same rule as dist.shard_range, evaluated by the library)"""
    lo, hi = C.c_long(), C.c_long()
    _lib.load().afv_shard_range(int(n_units), int(rank), int(world), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class DescriptorTable:
    def __init__(self, ctx, nsets, cap):
        self.ctx, self.lib = ctx, ctx.lib
        self.nsets, self.cap = int(nsets), int(cap)
        h = C.c_void_p()
        ctx.check(self.lib.afv_table_create(ctx.handle, self.nsets, self.cap, C.byref(h)), "afv_table_create")
        self.handle = h

    def close(self):
        if getattr(self, "handle", None):
            self.lib.afv_table_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- filling ----
    def set(self, slot, desc, angles=None):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        ang = None if angles is None else np.ascontiguousarray(angles, np.float32)
        self.ctx.check(self.lib.afv_table_set(self.handle, int(slot), ptr(desc), ptr(ang), len(desc)), "afv_table_set")

    def set_featvec(self, slot, node_id, seg_ptr, seg_idx):
        node_id, seg_ptr, seg_idx = _i32(node_id), _i32(seg_ptr), _i32(seg_idx)
        self.ctx.check(self.lib.afv_table_set_featvec(self.handle, int(slot), ptr(node_id), ptr(seg_ptr), ptr(seg_idx), len(node_id)),
                       "afv_table_set_featvec")

    def set_geometry(self, slot, x, y, sigma2, u_right=None):
        """mvKeysUn positions and GetKeyPt1DSigma2 of a slot; u_right = KeyFrame::mvuRight of a stereo keyframe (None: monocular)"""
        x, y, s = (np.ascontiguousarray(v, np.float32) for v in (x, y, sigma2))
        self.ctx.check(self.lib.afv_table_set_geometry(self.handle, int(slot), ptr(x), ptr(y), ptr(s)), "afv_table_set_geometry")
        if u_right is not None:
            u = np.ascontiguousarray(u_right, np.float32)
            self.ctx.check(self.lib.afv_table_set_u_right(self.handle, int(slot), ptr(u)), "afv_table_set_u_right")

    def set_valid(self, slot, valid):
        """valid[i] = feature i has a good map point (None = all valid); honoured by match_bow"""
        v = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        self.ctx.check(self.lib.afv_table_set_valid(self.handle, int(slot), ptr(v)), "afv_table_set_valid")

    def device_views(self):
        """zero-copy torch views of the table: desc uint8 [nsets, cap, 32], angle float32 [nsets, cap], n int32 [nsets]"""
        import torch
        d, a, n = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx.check(self.lib.afv_table_device_ptrs(self.handle, C.byref(d), C.byref(a), C.byref(n)))
        dev = torch.device("cuda", self.ctx.device)

        def view(p, nbytes, dtype, shape):
            class _Holder:  # __cuda_array_interface__ carrier; the table owns the memory
                pass
            h = _Holder()
            typestr = {torch.uint8: "|u1", torch.float32: "<f4", torch.int32: "<i4"}[dtype]
            h.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (p.value, False), "version": 2}
            return torch.as_tensor(h, device=dev)
        return (view(d, self.nsets * self.cap * 32, torch.uint8, (self.nsets, self.cap, 32)),
                view(a, self.nsets * self.cap * 4, torch.float32, (self.nsets, self.cap)),
                view(n, self.nsets * 4, torch.int32, (self.nsets,)))

    def upload(self, table, angles, counts):
        """fill every slot from host arrays [nsets, cap, 32] / [nsets, cap] / [nsets]: three bulk copies through the
        zero-copy views when torch can wrap the pointers, else one afv_table_set per slot"""
        try:
            import torch
            d, a, n = self.device_views()
            d.copy_(torch.from_numpy(np.ascontiguousarray(table, np.uint8)))
            a.copy_(torch.from_numpy(np.ascontiguousarray(angles, np.float32)))
            n.copy_(torch.from_numpy(np.ascontiguousarray(counts, np.int32)))
            torch.cuda.synchronize(d.device)
            self.sync_counts()
        except (ImportError, TypeError, RuntimeError):
            for k in range(self.nsets):
                self.set(k, table[k, :counts[k]], angles[k, :counts[k]])

    def sync_counts(self):
        self.ctx.check(self.lib.afv_table_sync_counts(self.handle), "afv_table_sync_counts")

    # ---- replication ----
    def broadcast(self, comm, root=0):
        """one RCCL broadcast per array + one for the replica image (FeatureVector structure, per-slot flags); returns the device time
        of the broadcasts in ms"""
        ms = C.c_float(0.0)
        self.ctx.check(self.lib.afv_table_broadcast(comm.handle, self.handle, int(root), C.byref(ms)), "afv_table_broadcast")
        return float(ms.value)

    def clone_into(self, other):
        """replica without a communicator: everything this table holds, rebuilt in `other` the way a broadcast receiver does"""
        other.ctx.check(self.lib.afv_table_clone(self.handle, other.handle), "afv_table_clone")
        return other

    # ---- matching ----
    def match_pairs(self, pair_a, pair_b, th_low, nnratio, check_orientation=True, want_matches=True):
        """brute-force SearchByBoW(KF,KF) per pair; host arrays in, host arrays out"""
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = len(pa)
        m = np.empty((n, self.cap), np.int32) if want_matches else None
        nm = np.zeros(n, np.int32)
        self.ctx.check(self.lib.afv_table_match_pairs(self.handle, ptr(pa), ptr(pb), n, float(th_low), float(nnratio), int(bool(check_orientation)),
                                                      ptr(m), ptr(nm)), "afv_table_match_pairs")
        return m, nm

    def match_pairs_device(self, pair_a, pair_b, th_low, nnratio, check_orientation=True, match=None, nmatches=None, stream=None):
        """device tensors in (int32 pair lists), device tensors out; asynchronous on torch's current stream"""
        import torch
        n = pair_a.numel()
        if match is None:
            match = torch.empty((n, self.cap), dtype=torch.int32, device=pair_a.device)
        if nmatches is None:
            nmatches = torch.empty((n,), dtype=torch.int32, device=pair_a.device)
        rc = _lib.launch_ordered(self.ctx, pair_a.device, stream, lambda s: self.lib.afv_table_match_pairs_device(
            self.handle, pair_a.data_ptr(), pair_b.data_ptr(), n, float(th_low), float(nnratio), int(bool(check_orientation)),
            match.data_ptr(), nmatches.data_ptr(), s), (pair_a, pair_b, match, nmatches))
        self.ctx.check(rc,
                       "afv_table_match_pairs_device")
        return match, nmatches

    def match_bow(self, pair_a, pair_b, th_low, nnratio, check_orientation=True, want_matches=True):
        """BoW-guided SearchByBoW(KF,KF) per pair over the stored FeatureVectors"""
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = len(pa)
        m = np.empty((n, self.cap), np.int32) if want_matches else None
        nm = np.zeros(n, np.int32)
        self.ctx.check(self.lib.afv_table_match_bow(self.handle, ptr(pa), ptr(pb), n, float(th_low), float(nnratio), int(bool(check_orientation)),
                                                    ptr(m), ptr(nm)), "afv_table_match_bow")
        return m, nm

    def match_bow_frame(self, slots, frame, th_low, nnratio, check_orientation=True, want_matches=True):
        """relocalisation batch: SearchByBoW(KF, Frame) of ONE frame (a matcher.FeatureView: descriptors, FeatureVector, angles) against
        the keyframes in `slots`.  Returns match_f[nslots, frame.N] (keyframe feature per frame feature, -1 = none) and nmatches[nslots]"""
        from ._lib import FrameView
        sl = _i32(slots)
        desc = np.ascontiguousarray(frame.descriptors, np.uint8).reshape(-1, 32)
        ids, sp, flat, nn = frame.csr()
        ids, sp, flat = _i32(ids), _i32(sp), _i32(flat)
        ang = None if frame.angles is None else np.ascontiguousarray(frame.angles, np.float32)
        fv = FrameView(desc.ctypes.data if len(desc) else None, len(desc), None if ang is None else ang.ctypes.data,
                       ids.ctypes.data if nn else None, sp.ctypes.data if nn else None, flat.ctypes.data if nn else None, int(nn))
        m = np.empty((len(sl), len(desc)), np.int32) if want_matches else None
        nm = np.zeros(len(sl), np.int32)
        self.ctx.check(self.lib.afv_table_match_bow_frame(self.handle, ptr(sl), len(sl), C.byref(fv), float(th_low), float(nnratio),
                                                          int(bool(check_orientation)), ptr(m), ptr(nm)), "afv_table_match_bow_frame")
        return m, nm

    def set_from_frame(self, slot, frame):
        """KeyFrame::KeyFrame(Frame&) (KeyFrame.cc:36-60): a resident frame (frame.Frame) becomes keyframe `slot`, device to device"""
        self.ctx.check(self.lib.afv_table_set_from_frame(self.handle, int(slot), frame.handle), "afv_table_set_from_frame")

    def match_bow_frame_resident(self, slots, frame, th_low, nnratio, check_orientation=True, want_matches=True):
        """match_bow_frame with the frame side taken from a resident frame (frame.Frame after ComputeBoW): nothing of the frame travels"""
        sl = _i32(slots)
        m = np.empty((len(sl), frame.N), np.int32) if want_matches else None
        nm = np.zeros(len(sl), np.int32)
        self.ctx.check(self.lib.afv_table_match_bow_frame_h(self.handle, ptr(sl), len(sl), frame.handle, float(th_low), float(nnratio),
                                                            int(bool(check_orientation)), ptr(m), ptr(nm)), "afv_table_match_bow_frame_h")
        return m, nm

    def match_triangulation(self, pair_a, pair_b, F12, epipoles, th_low, has_mp1=None, has_mp2=None, only_stereo=False):
        """SearchForTriangulation per pair.  F12: [npairs, 9] (row-major), epipoles: [npairs, 2]; has_mp1/2: per pair uint8
        arrays or None"""
        pa, pb = _i32(pair_a), _i32(pair_b)
        n = len(pa)
        F12 = np.ascontiguousarray(F12, np.float32).reshape(n, 9)
        ep = np.ascontiguousarray(epipoles, np.float32).reshape(n, 2)
        jobs = (TableTriJob * n)()
        keep = []
        for p in range(n):
            j = jobs[p]
            j.struct_size = C.sizeof(TableTriJob)
            for k in range(9):
                j.F12[k] = float(F12[p, k])
            j.ex, j.ey, j.th_low = float(ep[p, 0]), float(ep[p, 1]), float(th_low)
            j.only_stereo = int(bool(only_stereo))
            for name, src in (("has_mp1", has_mp1), ("has_mp2", has_mp2)):
                if src is not None and src[p] is not None:
                    a = np.ascontiguousarray(src[p], np.uint8)
                    keep.append(a)
                    setattr(j, name, a.ctypes.data)
        m = np.empty((n, self.cap), np.int32)
        nm = np.zeros(n, np.int32)
        self.ctx.check(self.lib.afv_table_match_triangulation(self.handle, ptr(pa), ptr(pb), jobs, n, ptr(m), ptr(nm)),
                       "afv_table_match_triangulation")
        return m, nm
