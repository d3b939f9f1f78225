"""ctypes binding of libafv_hip.so (the C-ABI declared in include/afv_hip.h).

Plumbing only: no arithmetic happens here.  The library is the hand-written HIP path; if it is missing the import
fails loudly — there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libafv_hip.so")


def use_library(path):
    """measurement tooling only (bench.py --lib, tools/experiments.py): bind a variant build of the library instead of the in-tree
    one.  Must be called before the first load(); nothing in the environment can redirect the loader."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("the library is already loaded")
    LIB_PATH = os.path.abspath(path)


ABI_VERSION = 6  # include/afv_hip.h AFV_ABI_VERSION this mirror was written against
MAX_LEVELS = 8
DESC_BYTES = 32
OK, EINVAL, ENODEV, ENOMEM, EHIP, ECAPACITY, EUNSUPPORTED, ETIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7
MATCH_KF_KF, MATCH_KF_FRAME = 0, 1
MATCH_FLOAT32 = 0x100  # OR-ed into MatchJob.mode: float rows, L2^2 distances (include/afv_hip.h AFV_MATCH_FLOAT32)
PROJ_LOCALMAP, PROJ_LASTFRAME = 0, 1
STAGES = ("pyramid", "fast_nms", "select_quadtree", "describe", "match_topk", "match_resolve", "retain_harris")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("nlevels", C.c_int32), ("scale_factor", C.c_float),
                ("fast_threshold", C.c_int32), ("max_width", C.c_int32), ("max_height", C.c_int32),
                ("max_batch", C.c_int32)]


class Geometry(C.Structure):
    _fields_ = [("nlevels", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("lw", C.c_int32 * MAX_LEVELS), ("lh", C.c_int32 * MAX_LEVELS), ("lscale", C.c_float * MAX_LEVELS),
                ("quota", C.c_int32 * MAX_LEVELS), ("cv_quota", C.c_int32 * MAX_LEVELS),
                ("cand_cap", C.c_int32 * MAX_LEVELS)]


class MatchJob(C.Structure):
    _fields_ = [("desc1", C.c_void_p), ("n1", C.c_int32), ("desc2", C.c_void_p), ("n2", C.c_int32),
                ("desc_bytes", C.c_int32),
                ("node_id1", C.c_void_p), ("seg_ptr1", C.c_void_p), ("seg_idx1", C.c_void_p), ("nnodes1", C.c_int32),
                ("node_id2", C.c_void_p), ("seg_ptr2", C.c_void_p), ("seg_idx2", C.c_void_p), ("nnodes2", C.c_int32),
                ("valid1", C.c_void_p), ("valid2", C.c_void_p), ("angle1", C.c_void_p), ("angle2", C.c_void_p),
                ("th_low", C.c_float), ("nnratio", C.c_float), ("check_orientation", C.c_int32), ("mode", C.c_int32)]


class TriJob(C.Structure):
    """struct_size first: the runtime strides the array by it and reads only what it covers (include/afv_hip.h)"""
    _fields_ = [("struct_size", C.c_uint32), ("bow", MatchJob), ("x1", C.c_void_p), ("y1", C.c_void_p), ("x2", C.c_void_p), ("y2", C.c_void_p),
                ("sigma2_2", C.c_void_p), ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float),
                ("u_right1", C.c_void_p), ("u_right2", C.c_void_p), ("only_stereo", C.c_int32)]


class TableTriJob(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float), ("has_mp1", C.c_void_p), ("has_mp2", C.c_void_p),
                ("th_low", C.c_float), ("only_stereo", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("desc32", C.c_void_p), ("n", C.c_int32), ("angle", C.c_void_p), ("node_id", C.c_void_p), ("seg_ptr", C.c_void_p),
                ("seg_idx", C.c_void_p), ("nnodes", C.c_int32)]


class ProjJob(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("desc", C.c_void_p), ("n", C.c_int32), ("desc_bytes", C.c_int32),
                ("x", C.c_void_p), ("y", C.c_void_p), ("size", C.c_void_p), ("angle", C.c_void_p), ("occupied", C.c_void_p),
                ("inf", C.c_void_p), ("min_x", C.c_float), ("min_y", C.c_float), ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float),
                ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("nq", C.c_int32),
                ("qdesc", C.c_void_p), ("qvalid", C.c_void_p), ("qu", C.c_void_p), ("qv", C.c_void_p), ("qr", C.c_void_p),
                ("qmin_size", C.c_void_p), ("qmax_size", C.c_void_p), ("qangle", C.c_void_p), ("qoccupies", C.c_void_p),
                ("th_high", C.c_float), ("nnratio", C.c_float), ("size_tol", C.c_float), ("inv_size_tol", C.c_float),
                ("check_orientation", C.c_int32), ("mode", C.c_int32),
                ("u_right", C.c_void_p), ("q_ur", C.c_void_p), ("q_er_max", C.c_void_p), ("float_dim", C.c_int32)]


class FrameParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("distorted", C.c_int32), ("cap", C.c_int32), ("desc_bytes", C.c_int32),
                ("float_dim", C.c_int32)]


class ProjQueries(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("nq", C.c_int32), ("qdesc", C.c_void_p), ("desc_bytes", C.c_int32),
                ("qvalid", C.c_void_p), ("qu", C.c_void_p), ("qv", C.c_void_p), ("qr", C.c_void_p), ("qmin_size", C.c_void_p),
                ("qmax_size", C.c_void_p), ("qangle", C.c_void_p), ("qoccupies", C.c_void_p), ("q_ur", C.c_void_p), ("q_er_max", C.c_void_p),
                ("occupied", C.c_void_p), ("th_high", C.c_float), ("nnratio", C.c_float), ("check_orientation", C.c_int32), ("mode", C.c_int32),
                ("qref_table", C.c_void_p), ("qref_slot", C.c_void_p), ("qref_idx", C.c_void_p)]


def sized(struct):
    """a job record with its struct_size filled in"""
    obj = struct()
    obj.struct_size = C.sizeof(struct)
    return obj


# every symbol include/afv_hip.h declares: (name, restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
SYMBOLS = {
    "afv_abi_version": (_i, []),
    "afv_default_orb_params": (None, [C.POINTER(OrbParams)]),
    "afv_create": (_i, [_i, C.POINTER(OrbParams), C.POINTER(_vp)]),
    "afv_destroy": (None, [_vp]),
    "afv_strerror": (C.c_char_p, [_i]),
    "afv_last_error": (C.c_char_p, [_vp]),
    "afv_max_keypoints_per_frame": (_i, [_vp]),
    "afv_stream": (_vp, [_vp]),
    "afv_orb_extract": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, C.POINTER(_i)]),
    "afv_orb_detect": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, C.POINTER(_i)]),
    "afv_orb_compute": (_i, [_vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "afv_orb_extract_batch": (_i, [_vp, C.POINTER(_vp), _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "afv_orb_extract_batch_device": (_i, [_vp, _vp, _i, _i, _i, _i, _sz, _vp, _vp, _i, _vp, _vp, _vp]),
    "afv_orb_size_sigma": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "afv_match_bow": (_i, [_vp, C.POINTER(MatchJob), _i, _vp, _vp]),
    "afv_match_triangulation": (_i, [_vp, C.POINTER(TriJob), _i, _vp, _vp]),
    "afv_match_bruteforce_pairs_device": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _f, _f, _i, _vp, _vp, _vp]),
    "afv_table_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "afv_table_destroy": (None, [_vp]),
    "afv_table_set": (_i, [_vp, _i, _vp, _vp, _i]),
    "afv_table_set_featvec": (_i, [_vp, _i, _vp, _vp, _vp, _i]),
    "afv_table_set_geometry": (_i, [_vp, _i, _vp, _vp, _vp]),
    "afv_table_set_u_right": (_i, [_vp, _i, _vp]),
    "afv_table_set_valid": (_i, [_vp, _i, _vp]),
    "afv_table_device_ptrs": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "afv_table_sync_counts": (_i, [_vp]),
    "afv_table_match_pairs": (_i, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp]),
    "afv_table_match_pairs_device": (_i, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp, _vp]),
    "afv_table_match_bow": (_i, [_vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp]),
    "afv_table_match_triangulation": (_i, [_vp, _vp, _vp, C.POINTER(TableTriJob), _i, _vp, _vp]),
    "afv_table_match_bow_frame": (_i, [_vp, _vp, _i, C.POINTER(FrameView), _f, _f, _i, _vp, _vp]),
    "afv_table_broadcast": (_i, [_vp, _vp, _i, C.POINTER(_f)]),
    "afv_table_clone": (_i, [_vp, _vp]),
    "afv_comm_unique_id": (_i, [_vp]),
    "afv_comm_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "afv_comm_destroy": (None, [_vp]),
    "afv_comm_rank": (_i, [_vp]),
    "afv_comm_size": (_i, [_vp]),
    "afv_comm_broadcast": (_i, [_vp, _vp, _sz, _i, _vp]),
    "afv_comm_allgather": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "afv_shard_range": (None, [C.c_long, _i, _i, C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "afv_match_l2": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _f, _f, _vp, _vp]),
    "afv_match_l2_pairs_device": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "afv_match_projection": (_i, [_vp, C.POINTER(ProjJob), _i, _vp, _vp]),
    "afv_match_fuse": (_i, [_vp, C.POINTER(ProjJob), _i, _vp, _vp]),
    "afv_match_sim3": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "afv_match_initialization": (_i, [_vp, _vp, _i, _vp, _vp]),
    "afv_vocab_create": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, C.POINTER(_vp)]),
    "afv_distinctive_descriptors": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "afv_distinctive_descriptors_f32": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp]),
    "afv_vocab_create_f32": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, C.POINTER(_vp)]),
    "afv_bow_transform_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "afv_vocab_destroy": (None, [_vp, _vp]),
    "afv_bow_transform": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "afv_vocab_set_stopped": (_i, [_vp, _vp, _vp]),
    "afv_frame_create": (_i, [_vp, C.POINTER(FrameParams), C.POINTER(_vp)]),
    "afv_frame_destroy": (None, [_vp]),
    "afv_frame_extract": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, C.POINTER(_i)]),
    "afv_frame_set_features": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "afv_frame_set_undistorted": (_i, [_vp, _vp, _vp]),
    "afv_frame_count": (_i, [_vp]),
    "afv_frame_device_ptrs": (_i, [_vp] + [C.POINTER(_vp)] * 7),
    "afv_frame_get_grid": (_i, [_vp, _vp, _vp]),
    "afv_frame_bow_transform": (_i, [_vp, _vp, _i, _vp, _vp, C.POINTER(C.c_int32)]),
    "afv_frame_get_featvec": (_i, [_vp, _vp, _vp, _vp]),
    "afv_frame_match_projection": (_i, [_vp, C.POINTER(ProjQueries), _vp, _vp]),
    "afv_frame_match_fuse": (_i, [_vp, C.POINTER(ProjQueries), _i, _vp, _vp]),
    "afv_frame_match_initialization": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _i, _vp, _vp]),
    "afv_table_set_from_frame": (_i, [_vp, _i, _vp]),
    "afv_table_match_bow_frame_h": (_i, [_vp, _vp, _i, _vp, _f, _f, _i, _vp, _vp]),
    "afv_set_projection_resolve": (_i, [_vp, _i]),
    "afv_hamming256": (_i, [_vp, _vp]),
    # include/afv_akaze.h
    "afv_akaze_default_params": (None, [_vp]),
    "afv_akaze_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "afv_akaze_destroy": (None, [_vp]),
    "afv_akaze_last_error": (C.c_char_p, [_vp]),
    "afv_akaze_plan_for": (_i, [_vp, _i, _i, _vp]),
    "afv_akaze_scale_space": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_size_t]),
    "afv_akaze_scale_space_device": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_size_t]),
    "afv_akaze_synchronize": (_i, [_vp]),
    "afv_akaze_get_plane": (_i, [_vp, _i, _i, _i, _vp]),
    "afv_akaze_get_kcontrast": (_i, [_vp, _i, _vp]),
    "afv_akaze_detect": (_i, [_vp]),
    "afv_akaze_get_keypoints": (_i, [_vp, _i, _vp, _i, _vp]),
    "afv_akaze_get_candidates": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "afv_akaze_describe": (_i, [_vp]),
    "afv_akaze_get_features": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "afv_akaze_extract": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_size_t, _vp, _vp, _i, _vp]),
    "afv_akaze_extract_device": (_i, [_vp, _vp, _i, _i, _i, _i, C.c_size_t]),
    "afv_akaze_get_quotas": (_i, [_vp, _vp]),
    "afv_akaze_set_step_by_step": (_i, [_vp, _i]),
    "afv_akaze_set_suppress_engine": (_i, [_vp, _i, _i]),
    "afv_akaze_debug_neighbour_cap": (_i, [_vp, _i]),
    "afv_akaze_profile_enable": (_i, [_vp, _i]),
    "afv_akaze_profile_read": (_i, [_vp, _vp, _vp, _vp]),
    "afv_profile_enable": (_i, [_vp, _i]),
    "afv_profile_read": (_i, [_vp, _vp, _vp, _vp]),
    "afv_set_split_threshold": (_i, [_vp, _i]),
    "afv_set_split_chunks": (_i, [_vp, _i]),
    "afv_set_match_engine": (_i, [_vp, _i]),
    "afv_set_small_batch_path": (_i, [_vp, _i, _i]),
    "afv_set_match_resolve": (_i, [_vp, _i]),
    "afv_set_l2_chunk_pairs": (_i, [_vp, _i]),
    "afv_debug_pyramid_plan": (_i, [C.POINTER(OrbParams), _i, _i, _i, _i, _vp, _i, _vp, _vp, _i]),
    "afv_set_pipeline_chunk": (_i, [_vp, _i, _i]),
    "afv_get_geometry": (_i, [_vp, C.POINTER(Geometry)]),
    "afv_debug_get_level": (_i, [_vp, _i, _i, _vp]),
    "afv_debug_get_candidates": (_i, [_vp, _i, _i, _vp, _vp, _i, C.POINTER(_i)]),
    "afv_debug_get_selected": (_i, [_vp, _i, _i, _vp, _vp, _vp, _i, C.POINTER(_i)]),
    "afv_debug_blur_level": (_i, [_vp, _i, _i, _vp]),
    "afv_num_stages": (_i, []),
}

_lib = None


def load():
    """dlopen libafv_hip.so and bind every declared symbol.  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # Load order matters: PyTorch ships its own libamdhip64; if libafv_hip.so pulled in /opt/rocm's copy first, a
    # later `import torch` would bring a SECOND HIP runtime into the process and one of the two fails to see the
    # GPU.  Importing torch first makes libafv_hip.so bind to the runtime that is already loaded.
    try:
        import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build the HIP extension first (python __graft_entry__.py or "
                          "anyfeature-vslam_amd/build.py); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    # the profile getters write afv_num_stages() entries: a (variant) build with more stages than this mirror knows would overrun the
    # arrays the wrapper hands it
    if lib.afv_abi_version() != ABI_VERSION:
        raise ImportError("%s speaks ABI revision %d, this mirror %d" % (LIB_PATH, lib.afv_abi_version(), ABI_VERSION))
    if lib.afv_num_stages() > len(STAGES):
        raise ImportError("%s reports %d profile stages, this mirror knows %d" % (LIB_PATH, lib.afv_num_stages(), len(STAGES)))
    _lib = lib
    return lib


class AfvError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = load().afv_strerror(code).decode()
        super().__init__("afv error %d (%s)%s" % (code, msg, (": " + detail) if detail else ""))


def strerror(code):
    return load().afv_strerror(int(code)).decode()


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def launch_ordered(ctx, device, stream, fn, tensors=()):
    """Call fn(hipStream_t) so that the work is ordered with torch's CURRENT stream.
    An explicit `stream` or a non-default torch stream is handed to the library as is.  torch reports its default stream
    as handle 0, which the C-ABI reads as "the context's own non-blocking stream" (not ordered with torch): in that case
    the context's stream is bridged with events through torch.cuda.ExternalStream: wait for torch's stream before, make
    torch's stream wait after."""
    if stream is not None:
        return fn(stream)
    import torch
    cur = torch.cuda.current_stream(device)
    if cur.cuda_stream:
        return fn(cur.cuda_stream)
    ext = getattr(ctx, "_ext_stream", None)
    if ext is None:
        ext = torch.cuda.ExternalStream(ctx.stream, device=device)
        ctx._ext_stream = ext
    ext.wait_stream(cur)
    rc = fn(None)
    cur.wait_stream(ext)
    # no tensor.record_stream(ext): everything torch does with these tensors later (including freeing and reusing their
    # memory) is enqueued on `cur` AFTER the wait above, and the allocator must never touch the context's stream — it is
    # destroyed with the context while cached blocks live on
    return rc
