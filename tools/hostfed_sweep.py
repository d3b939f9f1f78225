"""GPU-box measurement: host-fed pipeline (afv_orb_extract_batch on page-locked buffers) against its parts."""
import importlib, json, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
afv = importlib.import_module("anyfeature-vslam_amd")
B = 512
ctx = afv.Context(max_batch=B)
fr_h = afv.synth.corners_batch(1, 64)
fr = torch.from_numpy(np.concatenate([fr_h] * 8)).pin_memory()
cap = ctx.cap
kps = torch.zeros((B, cap, 7), dtype=torch.float32).pin_memory(); desc = torch.zeros((B, cap, 32), dtype=torch.uint8).pin_memory(); n = torch.zeros((B,), dtype=torch.int32).pin_memory()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
d_fr = fr.cuda(); dk = torch.empty((B, cap, 7), dtype=torch.float32, device="cuda"); dd = torch.empty((B, cap, 32), dtype=torch.uint8, device="cuda"); dn = torch.empty((B,), dtype=torch.int32, device="cuda"); st = torch.zeros((1,), dtype=torch.int32, device="cuda")
side = torch.cuda.Stream()
def dev():
    with torch.cuda.stream(side): ctx.extract_batch_device(d_fr, dk, dd, dn, st, cap)
for sc in (0, 4, 8):
    ctx.set_split_chunks(sc)
    print("device-resident extract, %d chunks (0 = automatic): %.3f ms" % (sc, t(dev)), flush=True)
ctx.set_split_chunks(0)
print("H2D only: %.3f ms" % t(lambda: d_fr.copy_(fr, non_blocking=True)))
print("D2H only: %.3f ms" % t(lambda: (kps.copy_(dk, non_blocking=True), desc.copy_(dd, non_blocking=True))))
for ch in (16, 32, 48, 64, 96):
    for ahead in (4, 8, 16):
        ctx.set_pipeline_chunk(ch, ahead)
        print("host-fed chunk %d ahead %d: %.3f ms" % (ch, ahead, t(lambda: ctx.extract_batch_host(fr, kps, desc, n))), flush=True)
