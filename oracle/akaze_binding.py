"""ctypes binding of oracle/libakz.so (AKAZE61 restatement).  TEST INFRASTRUCTURE ONLY — see oracle/akaze.h."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_LEVELS, MAX_FED = 16, 32


class Options(C.Structure):
    _fields_ = [("omax", C.c_int32), ("nsublevels", C.c_int32), ("soffset", C.c_float), ("derivative_factor", C.c_float),
                ("dthreshold", C.c_float), ("min_dthreshold", C.c_float), ("kcontrast_percentile", C.c_float),
                ("kcontrast_nbins", C.c_int32)]


class LevelInfo(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("octave", C.c_int32), ("sublevel", C.c_int32), ("sigma_size", C.c_int32),
                ("esigma", C.c_float), ("etime", C.c_float), ("nsteps", C.c_int32), ("tau", C.c_float * MAX_FED)]


class Plan(C.Structure):
    _fields_ = [("nlevels", C.c_int32), ("w", C.c_int32), ("h", C.c_int32), ("lv", LevelInfo * MAX_LEVELS),
                ("gauss_soffset", C.c_float * 32), ("ksize_soffset", C.c_int32), ("gauss_one", C.c_float * 8), ("ksize_one", C.c_int32)]


class Planes(C.Structure):
    _fields_ = [("Lt", C.c_void_p), ("Lsmooth", C.c_void_p), ("Lx", C.c_void_p), ("Ly", C.c_void_p), ("Ldet", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libakz.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, "libakz.so"])
        _lib = C.CDLL(path)
        _lib.akz_kcontrast.restype = C.c_float
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_options():
    o = Options()
    lib().akz_default_options(C.byref(o))
    return o


def make_plan(w, h, opts=None):
    opts = opts or default_options()
    p = Plan()
    rc = lib().akz_make_plan(C.byref(opts), int(w), int(h), C.byref(p))
    if rc:
        raise ValueError("akz_make_plan failed")
    return p


def convert(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.float32)
    lib().akz_convert(_p(gray), w, w, h, _p(out))
    return out


def gauss(src, taps):
    src = np.ascontiguousarray(src, np.float32)
    taps = np.ascontiguousarray(taps, np.float32)
    out = np.empty_like(src)
    lib().akz_gauss(_p(src), src.shape[1], src.shape[0], _p(taps), len(taps), _p(out))
    return out


def kcontrast(img, plan, opts=None):
    opts = opts or default_options()
    img = np.ascontiguousarray(img, np.float32)
    return float(lib().akz_kcontrast(_p(img), img.shape[1], img.shape[0], C.byref(plan), C.byref(opts)))


def halfsample(src):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().akz_halfsample(_p(src), w, h, _p(out), w // 2, h // 2)
    return out


def flow_g2(ls, k):
    ls = np.ascontiguousarray(ls, np.float32)
    out = np.empty_like(ls)
    lib().akz_flow_g2(_p(ls), ls.shape[1], ls.shape[0], C.c_float(k), _p(out))
    return out


def nld_step(lt, flow, tau):
    lt = np.ascontiguousarray(lt, np.float32); flow = np.ascontiguousarray(flow, np.float32)
    out = np.empty_like(lt)
    lib().akz_nld_step(_p(lt), _p(flow), lt.shape[1], lt.shape[0], C.c_float(tau), _p(out))
    return out


def hessian(ls, sigma_size):
    ls = np.ascontiguousarray(ls, np.float32)
    lx = np.empty_like(ls); ly = np.empty_like(ls); ldet = np.empty_like(ls)
    lib().akz_hessian(_p(ls), ls.shape[1], ls.shape[0], int(sigma_size), _p(lx), _p(ly), _p(ldet))
    return lx, ly, ldet


def scale_space(gray, plan=None, opts=None):
    """returns (list of dict(Lt, Lsmooth) per level, kcontrast of level 0)"""
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    opts = opts or default_options()
    plan = plan or make_plan(w, h, opts)
    arr = (Planes * plan.nlevels)()
    keep = []
    for i in range(plan.nlevels):
        L = plan.lv[i]
        d = {"Lt": np.zeros((L.h, L.w), np.float32), "Lsmooth": np.zeros((L.h, L.w), np.float32)}
        keep.append(d)
        arr[i].Lt = _p(d["Lt"]); arr[i].Lsmooth = _p(d["Lsmooth"])
    k0 = C.c_float()
    lib().akz_scale_space(C.byref(plan), C.byref(opts), _p(gray), w, arr, C.byref(k0))
    return keep, float(k0.value)


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def _planes_array(plan, levels, keys=("Lt", "Lsmooth", "Lx", "Ly", "Ldet")):
    arr = (Planes * plan.nlevels)()
    for i in range(plan.nlevels):
        for k in keys:
            if k in levels[i]:
                setattr(arr[i], k, _p(levels[i][k]))
    return arr


def full_evolution(gray, plan=None, opts=None):
    """scale space + multiscale derivatives: list of dict(Lt, Lsmooth, Lx, Ly, Ldet)"""
    opts = opts or default_options()
    gray = np.ascontiguousarray(gray, np.uint8)
    plan = plan or make_plan(gray.shape[1], gray.shape[0], opts)
    levels, k0 = scale_space(gray, plan, opts)
    for i in range(plan.nlevels):
        lx, ly, ldet = hessian(levels[i]["Lsmooth"], plan.lv[i].sigma_size)
        levels[i].update(Lx=lx, Ly=ly, Ldet=ldet)
    return levels, k0


def level_candidates(plan, level, ldet, opts=None):
    opts = opts or default_options()
    ldet = np.ascontiguousarray(ldet, np.float32)
    cap = ldet.size // 4 + 64
    out = np.zeros(cap, np.int32)
    n = lib().akz_level_candidates(C.byref(plan), C.byref(opts), int(level), _p(ldet), _p(out), cap)
    return out[:n].copy()


def find_extrema(plan, levels, opts=None, cap=200000):
    opts = opts or default_options()
    arr = _planes_array(plan, levels)
    out = np.zeros(cap, KP_DTYPE)
    n = lib().akz_find_extrema(C.byref(plan), C.byref(opts), arr, _p(out), cap)
    return out[:n].copy()


def subpixel(plan, levels, kpts):
    arr = _planes_array(plan, levels)
    kpts = np.ascontiguousarray(kpts, KP_DTYPE).copy()
    n = lib().akz_subpixel(C.byref(plan), arr, _p(kpts), len(kpts))
    return kpts[:n].copy()


def compute_descriptors(plan, levels, kpts):
    arr = _planes_array(plan, levels)
    kpts = np.ascontiguousarray(kpts, KP_DTYPE).copy()
    desc = np.zeros((max(len(kpts), 1), 61), np.uint8)
    lib().akz_compute_descriptors(C.byref(plan), arr, _p(kpts), len(kpts), _p(desc))
    return kpts, desc[:len(kpts)].copy()


def get_angle(x, y):
    lib().akz_get_angle.restype = C.c_float
    return float(lib().akz_get_angle(C.c_float(x), C.c_float(y)))
