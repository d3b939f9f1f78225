"""-m gpu: the C++ host adapter (anyfeature-vslam_amd/adapter/afv_adapter.hpp) run as a plain C++ process — no python,
no torch in the loop — and checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "anyfeature-vslam_amd", "adapter", "adapter_selftest")


def test_cpp_adapter_end_to_end(afv, oracle, tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("adapter_selftest not built (run __graft_entry__.build())")
    img = afv.synth.corners_frame(1)
    raw = tmp_path / "frame.raw"
    raw.write_bytes(img.tobytes())
    out = str(tmp_path / "o")
    r = subprocess.run([BIN, str(raw), "640", "480", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    n1, n2, nm = [int(v) for v in r.stdout.split()]
    k1 = np.fromfile(out + ".kps1", dtype=afv.KP_DTYPE); d1 = np.fromfile(out + ".desc1", dtype=np.uint8).reshape(-1, 32)
    k2 = np.fromfile(out + ".kps2", dtype=afv.KP_DTYPE); d2 = np.fromfile(out + ".desc2", dtype=np.uint8).reshape(-1, 32)
    m21 = np.fromfile(out + ".match21", dtype=np.int32)
    size1 = np.fromfile(out + ".size1", dtype=np.float32)
    ok1, od1 = oracle.orb_extract(img)
    ok2, od2 = oracle.orb_extract(np.roll(img, 4, axis=1))
    assert len(k1) == n1 and len(k2) == n2
    assert k1.tobytes() == ok1.tobytes() and np.array_equal(d1, od1)
    assert k2.tobytes() == ok2.tobytes() and np.array_equal(d2, od2)
    want, wn = oracle.search_by_bow_kf_kf(od2, od1, angle1=ok2["angle"], angle2=ok1["angle"], th_low=75.0, nnratio=0.6,
                                          check_orientation=True)
    assert nm == wn and np.array_equal(m21, want) and wn > 300
    assert np.array_equal(size1, oracle.size_sigma(ok1)[0])
