import importlib, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
afv = importlib.import_module("anyfeature-vslam_amd")
B = 512
ctx = afv.Context(max_batch=B)
fr_h = afv.synth.corners_batch(1, 64)
fr = torch.from_numpy(np.concatenate([fr_h] * 8)).pin_memory()
cap = ctx.cap
kps = torch.zeros((B, cap, 7), dtype=torch.float32).pin_memory(); desc = torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory(); n = torch.zeros((B,), dtype=torch.int32).pin_memory()
ctx.extract_batch_host(fr, kps, desc, n)
ctx.extract_batch_host(fr, kps, desc, n)
t0 = time.perf_counter()
ctx.extract_batch_host(fr, kps, desc, n)
print("call ms", (time.perf_counter() - t0) * 1e3)
