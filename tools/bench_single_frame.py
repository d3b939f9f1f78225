"""stage breakdown of the single-frame path"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
afv = importlib.import_module("anyfeature-vslam_amd")
ctx = afv.Context()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1   # afv_set_small_batch_path: 0 = the batch kernels, 1 = library default
ctx.set_small_batch_path(mode)
print("small-batch path mode", mode)
for name, img in (("corners", afv.synth.corners_frame(1)), ("toy", np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "toy_gray.npz"))["gray"])):
    t = torch.from_numpy(img[None]).cuda()
    for _ in range(5): ctx.extract_batch_device(t)
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    N = 50
    t0 = time.perf_counter()
    for _ in range(N):
        ctx.extract_batch_device(t); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    st = ctx.profile_read(); ctx.profile_enable(False)
    print(name, "device-resident 1 frame: %.1f us wall;" % (dt * 1e6), {k: round(v["total_ms"] / N * 1e3, 1) for k, v in st.items() if v["launches"]}, "(us)")
    for _ in range(5): ctx.extract(img)
    t0 = time.perf_counter()
    for _ in range(N): ctx.extract(img)
    print(name, "host-buffer path: %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
