"""would the AKAZE61 batch gain from running its halves / quarters on separate streams? (scale space = HBM-bound, suppression = latency-bound)
one context of B frames against N contexts of B / N frames enqueued back to back, each on its own stream"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
afv = importlib.import_module("anyfeature-vslam_amd")
B, steps = 64, 5
frames_h = afv.synth.corners_batch(1, B, 1280, 720)
frames = torch.from_numpy(frames_h).cuda()
for n in (1, 2, 4, 8):
    ctxs = [afv.AkazeContext(afv.akaze.default_params(max_batch=B // n)) for _ in range(n)]
    parts = [frames[i * (B // n):(i + 1) * (B // n)] for i in range(n)]
    def step():
        for c, p in zip(ctxs, parts):
            c.extract_device(p)
    for _ in range(2):
        step()
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for c in ctxs: c.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print("contexts %d x %d frames: %.3f ms per %d frames = %.0f frames/s" % (n, B // n, dt * 1e3, B, B / dt), flush=True)
    for c in ctxs: c.close()
