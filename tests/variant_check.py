"""measurement helper (not a pytest module): a VARIANT build of the library (tools/experiments.py) against the oracle on a few frames:
   python tests/variant_check.py anyfeature-vslam_amd/build_exp/libafv_<name>.so"""
import importlib, sys, json
sys.path.insert(0, ".")
lib = sys.argv[1] if len(sys.argv) > 1 else None
if lib:
    importlib.import_module("anyfeature-vslam_amd._lib").use_library(lib)
afv = importlib.import_module("anyfeature-vslam_amd")
import oracle, numpy as np
ctx = afv.Context()
ok = True
for seed in (1, 2, 3):
    img = afv.synth.corners_frame(seed) if seed < 3 else afv.synth.noise_frame(3)
    k, d = ctx.extract(img)
    ok_, od = oracle.orb_extract(img)
    ok = ok and k.tobytes() == ok_.tobytes() and np.array_equal(d, od)
ctx2 = afv.Context(max_width=1280, max_height=720, max_batch=4)
fr = [afv.synth.corners_frame(5, 1280, 720), afv.synth.corners_frame(6, 1280, 720)]
res = ctx2.extract_batch(fr)
for f, (k, d) in zip(fr, res):
    ok_, od = oracle.orb_extract(f)
    ok = ok and k.tobytes() == ok_.tobytes() and np.array_equal(d, od)
print("bit-exact vs oracle:", ok)
