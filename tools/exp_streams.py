"""experiment: split the batch over N contexts (N streams) so latency-bound kernels of one half overlap VALU-bound ones of the other"""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
afv = importlib.import_module("anyfeature-vslam_amd")
B = 256
frames = torch.from_numpy(afv.synth.corners_batch(1, 64)).cuda().repeat(B // 64, 1, 1).contiguous()
afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
for nctx in (1, 2, 4):
    sub = B // nctx
    ctxs = [afv.Context(max_batch=sub) for _ in range(nctx)]
    ms = [afv.FeatureMatcher(0.6, True, ctx=c) for c in ctxs]
    streams = [c.lib.afv_stream(c.handle) for c in ctxs]
    cap = ctxs[0].cap
    bufs = []
    for i in range(nctx):
        bufs.append(dict(kps=torch.empty((sub, cap, 7), device="cuda"), desc=torch.empty((sub, cap, 32), dtype=torch.uint8, device="cuda"),
                         n=torch.empty((sub,), dtype=torch.int32, device="cuda"), st=torch.zeros((1,), dtype=torch.int32, device="cuda"),
                         match=torch.empty((sub, cap), dtype=torch.int32, device="cuda"), nm=torch.empty((sub,), dtype=torch.int32, device="cuda"),
                         pa=torch.arange(sub, dtype=torch.int32, device="cuda"), pb=(torch.arange(sub, dtype=torch.int32, device="cuda") + sub - 1) % sub))
    def step():
        for i, c in enumerate(ctxs):
            b = bufs[i]
            c.extract_batch_device(frames[i * sub:(i + 1) * sub], b["kps"], b["desc"], b["n"], b["st"], cap, stream=streams[i])
            ms[i].match_pairs_device(b["desc"], b["kps"], b["n"], b["pa"], b["pb"], th_low=75.0, check_orientation=True, match=b["match"],
                                     nmatches=b["nm"], stream=streams[i])
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    nk = sum(int(b["n"].sum().item()) for b in bufs)
    print("contexts=%d  ms/step=%.3f  Mkp/s=%.1f" % (nctx, dt * 1e3, nk / dt / 1e6))
    del ctxs, ms
