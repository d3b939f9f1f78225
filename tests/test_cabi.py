"""CPU: the C-ABI library loads, exports every symbol include/afv_hip.h and include/afv_akaze.h declare, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "afv_hip.h")).read() + open(os.path.join(ROOT, "include", "afv_akaze.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(afv_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_symbols_are_exported(afv):
    lib = afv._lib.load()
    declared = _declared_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libafv_hip.so does not export %s" % name
    # and the binding table knows each of them
    assert sorted(afv._lib.SYMBOLS) == declared


def test_keypoint_layout_matches_cv_keypoint(afv):
    # cv::KeyPoint = {Point2f pt; float size, angle, response; int octave, class_id} = 28 bytes
    dt = afv.KP_DTYPE
    assert dt.itemsize == 28
    assert [dt.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == [0, 4, 8, 12, 16, 20, 24]


def test_defaults_and_error_strings(afv):
    lib = afv._lib.load()
    p = afv._lib.OrbParams()
    lib.afv_default_orb_params(C.byref(p))
    assert (p.nfeatures, p.nlevels, p.fast_threshold, p.max_width, p.max_height, p.max_batch) == (1000, 8, 20, 640, 480, 1)
    assert abs(p.scale_factor - 1.2) < 1e-6
    assert lib.afv_strerror(0) == b"ok" and lib.afv_strerror(-2) == b"no usable HIP device"
    assert lib.afv_strerror(-99) == b"unknown error"


def test_no_cpu_fallback_without_gpu(afv):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(afv._lib.AfvError) as e:
        afv.Context()
    assert e.value.code == afv._lib.ENODEV
    with pytest.raises(afv._lib.AfvError):
        afv.FeatureExtractor_orb32(1000)


def test_invalid_arguments_do_not_crash(afv):
    lib = afv._lib.load()
    h = C.c_void_p()
    assert lib.afv_create(0, None, C.byref(h)) == afv._lib.EINVAL
    lib.afv_destroy(None)  # no-op
    assert lib.afv_max_keypoints_per_frame(None) == afv._lib.EINVAL
    assert lib.afv_last_error(None) == b""
    assert lib.afv_orb_extract(None, None, 0, 0, 0, None, None, 0, None) == afv._lib.EINVAL
    assert lib.afv_match_bow(None, None, 0, None, None) == afv._lib.EINVAL
    assert lib.afv_profile_enable(None, 1) == afv._lib.EINVAL
    # the device-resident Frame and its consumers (round 5)
    fh = C.c_void_p()
    assert lib.afv_frame_create(None, None, C.byref(fh)) == afv._lib.EINVAL
    lib.afv_frame_destroy(None)  # no-op
    assert lib.afv_frame_extract(None, None, 0, 0, 0, None, None, 0, None) == afv._lib.EINVAL
    assert lib.afv_frame_set_features(None, None, None, 0, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_set_undistorted(None, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_count(None) == afv._lib.EINVAL
    assert lib.afv_frame_get_grid(None, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_bow_transform(None, None, 4, None, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_get_featvec(None, None, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_match_projection(None, None, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_match_fuse(None, None, 1, None, None) == afv._lib.EINVAL
    assert lib.afv_frame_match_initialization(None, None, None, None, 100.0, 75.0, 0.9, 1, None, None) == afv._lib.EINVAL
    assert lib.afv_table_set_from_frame(None, 0, None) == afv._lib.EINVAL
    assert lib.afv_table_match_bow_frame_h(None, None, 0, None, 75.0, 0.7, 1, None, None) == afv._lib.EINVAL
    assert lib.afv_set_projection_resolve(None, 1) == afv._lib.EINVAL
    assert lib.afv_vocab_set_stopped(None, None, None) == afv._lib.EINVAL


def test_versioned_job_records_carry_their_size(afv):
    """afv_tri_job / afv_proj_job / afv_table_tri_job grew in place; now their first field is struct_size - the runtime
    strides by it, reads only what it covers, and rejects records that do not carry a plausible one.  The mirrors fill it (sized())."""
    L = afv._lib
    for st in (L.TriJob, L.ProjJob, L.TableTriJob, L.FrameParams, L.ProjQueries):
        assert st._fields_[0][0] == "struct_size" and L.sized(st).struct_size == C.sizeof(st)
    assert C.sizeof(L.ProjJob) % 8 == 0 and L.ProjJob.u_right.offset > L.ProjJob.mode.offset   # the stereo tail stays at the end


def test_akaze_abi_without_gpu(afv):
    """the AKAZE entry points: plan (host only) works anywhere; creation fails loudly without a device; NULLs are EINVAL"""
    import torch
    lib = afv._lib.load()
    prm = afv.akaze.default_params()
    assert (prm.omax, prm.nsublevels, prm.kcontrast_nbins, prm.nfeatures) == (2, 4, 300, 1000)
    assert abs(prm.dthreshold - 0.0005) < 1e-9 and abs(prm.scale_factor - 1.1892) < 1e-6
    plan = afv.akaze.plan_for(prm, 1280, 720)
    assert plan.nlevels == 8 and [plan.lv[i].nsteps for i in range(8)] == [0, 3, 3, 4, 4, 5, 6, 7]
    with pytest.raises(ValueError):
        afv.akaze.plan_for(prm, 8, 8)
    h = C.c_void_p()
    assert lib.afv_akaze_create(0, None, C.byref(h)) == afv._lib.EINVAL
    lib.afv_akaze_destroy(None)
    assert lib.afv_akaze_detect(None) == afv._lib.EINVAL
    assert lib.afv_akaze_extract(None, None, 0, 0, 0, 0, 0, None, None, 0, None) == afv._lib.EINVAL
    assert lib.afv_akaze_get_features(None, 0, None, None, 0, None) == afv._lib.EINVAL
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            afv.AkazeContext()


def test_tools_do_not_import_the_oracle():
    """tools/ are not test infrastructure either"""
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            text = open(os.path.join(ROOT, "tools", f)).read()
            assert "from oracle" not in text and "import oracle" not in text, f


def test_host_hamming_utility(afv, oracle):
    d = afv.synth.random_descriptors(4, 16)
    for i in range(0, 16, 2):
        assert afv.DescriptorDistance_orb32(d[i], d[i + 1]) == float(oracle.hamming256(d[i], d[i + 1]))


def test_host_descriptor_distance_dispatch(afv, oracle):
    """FeatureMatcher::DescriptorDistance (FeatureMatcher.cc:1508-1531) as the host utility: float rows -> L2^2 in cv::norm's order, 61-byte
    rows -> Hamming over the bytes"""
    s = afv.synth
    for dim in (128, 64, 6):
        a = (s.lcg_bytes(5, dim).astype(np.float32) / np.float32(255.0)) ** 2
        b = (s.lcg_bytes(6, dim).astype(np.float32) / np.float32(251.0))
        assert afv.FeatureMatcher.DescriptorDistance(a, b) == float(oracle.l2sqr(a, b))
    d = s.lcg_bytes(7, 2 * 61).reshape(2, 61)
    assert afv.FeatureMatcher.DescriptorDistance(d[0], d[1]) == float(oracle.hamming_bytes(d[0], d[1]))


def test_product_does_not_import_the_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "anyfeature-vslam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "afvo" not in text and "libakz" not in text and \
                    "akaze_binding" not in text, os.path.join(dirpath, f)


def test_adapter_compiles_in_its_opencv_configuration():
    """afv_adapter.hpp has cv::Mat / cv::KeyPoint branches (-DAFV_WITH_OPENCV) that the POD selftest never compiles; OpenCV is
    absent here, so they are syntax-checked against a mock of the few OpenCV names they use (tests/opencv_mock)."""
    import subprocess
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DAFV_WITH_OPENCV", "-I", os.path.join(ROOT, "tests", "opencv_mock"),
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "anyfeature-vslam_amd", "adapter"),
           os.path.join(ROOT, "tests", "opencv_mock", "compile_check.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr


def test_python_struct_mirrors_match_the_headers(tmp_path):
    """the ctypes structures of _lib.py / akaze.py against the layouts gcc gives include/afv_hip.h and include/afv_akaze.h: size of every
    structure and offset of every field (a field added to a header and forgotten in a mirror shifts everything behind it)"""
    import ctypes as C
    import importlib
    import subprocess
    pkg = importlib.import_module("anyfeature-vslam_amd")
    lib, akz = pkg._lib, importlib.import_module("anyfeature-vslam_amd.akaze")
    pairs = [("afv_orb_params", lib.OrbParams), ("afv_geometry", lib.Geometry), ("afv_match_job", lib.MatchJob), ("afv_tri_job", lib.TriJob),
             ("afv_table_tri_job", lib.TableTriJob), ("afv_frame_view", lib.FrameView), ("afv_proj_job", lib.ProjJob),
             ("afv_frame_params", lib.FrameParams), ("afv_proj_queries", lib.ProjQueries),
             ("afv_akaze_params", akz.AkazeParams), ("afv_akaze_level", akz.AkazeLevel), ("afv_akaze_plan", akz.AkazePlan)]
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "afv_hip.h"', '#include "afv_akaze.h"', 'int main(void) {']
    for cname, st in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in st._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr  # a field name the header does not have fails here
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True, timeout=60).stdout.splitlines())
    for cname, st in pairs:
        assert int(got[cname]) == C.sizeof(st), (cname, got[cname], C.sizeof(st))
        for fname, _ in st._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(st, fname).offset, (cname, fname)
