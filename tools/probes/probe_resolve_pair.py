"""phase 2 on ONE overlapping pair (frame vs the same frame rolled by 3 px) and on batches of such pairs: kernel time by hipEvents, both engines"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
afv = importlib.import_module("anyfeature-vslam_amd")
B, W, H = 128, 640, 480
ctx = afv.Context(max_batch=B, device=0)
m = afv.FeatureMatcher(0.6, True, ctx=ctx)
base = afv.synth.corners_batch(9001, B // 2, W, H)
fr = np.empty((B, H, W), np.uint8); fr[0::2] = base; fr[1::2] = np.roll(base, 3, axis=2)
frames = torch.from_numpy(fr).cuda(0)
kps, desc, n, st = ctx.extract_batch_device(frames)
torch.cuda.synchronize()
for npairs in (1, 8, 32, 64):
    pa = torch.arange(1, 2 * npairs, 2, dtype=torch.int32, device="cuda")   # overlapping pairs only: (1,0), (3,2), ...
    pb = pa - 1
    for eng in (0, 1):
        ctx.set_match_resolve(eng)
        ctx.set_small_batch_path(0)
        for _ in range(3):
            mt, nm = m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True)
        torch.cuda.synchronize()
        ctx.profile_enable(True)
        for _ in range(10):
            m.match_pairs_device(desc, kps, n, pa, pb, th_low=75.0, check_orientation=True)
        torch.cuda.synchronize()
        s = ctx.profile_read(); ctx.profile_enable(False)
        print("pairs %3d engine %d: topk %.1f us  resolve %.1f us  (matches/pair %.0f)" % (npairs, eng, s["match_topk"]["total_ms"] / s["match_topk"]["launches"] * 1e3,
              s["match_resolve"]["total_ms"] / s["match_resolve"]["launches"] * 1e3, float(nm.float().mean())))
