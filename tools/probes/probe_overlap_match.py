"""What would hiding the match behind the next batch's extraction buy?  Two contexts (own streams), double-buffered outputs: while
context E extracts batch k+1, context M matches batch k.  Upper bound for a pipelined extract + match entry point."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.getcwd())
afv = importlib.import_module("anyfeature-vslam_amd")
B, W, H = 512, 640, 480
dev = torch.device("cuda", 0)
ctxE = afv.Context(max_batch=B, device=0)
ctxM = afv.Context(max_batch=B, device=0)
afv.FeatureMatcher.setDescriptorDistanceThresholds(75.0)
mE = afv.FeatureMatcher(0.6, True, ctx=ctxE)
mM = afv.FeatureMatcher(0.6, True, ctx=ctxM)
frames = torch.from_numpy(afv.synth.corners_batch(1, B, W, H)).to(dev)
cap = ctxE.cap
bufs = []
for _ in range(2):
    bufs.append(dict(kps=torch.empty((B, cap, 7), dtype=torch.float32, device=dev), desc=torch.empty((B, cap, 32), dtype=torch.uint8, device=dev),
                     n=torch.empty((B,), dtype=torch.int32, device=dev), st=torch.zeros((1,), dtype=torch.int32, device=dev),
                     match=torch.empty((B, cap), dtype=torch.int32, device=dev), nm=torch.empty((B,), dtype=torch.int32, device=dev),
                     ev=torch.cuda.Event(), evm=torch.cuda.Event()))
pa = torch.arange(B, dtype=torch.int32, device=dev); pb = (pa + (B - 1)) % B
sE, sM = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

def serial(steps):
    b = bufs[0]
    for _ in range(steps):
        with torch.cuda.stream(sE):
            ctxE.extract_batch_device(frames, b["kps"], b["desc"], b["n"], b["st"], cap)
            mE.match_pairs_device(b["desc"], b["kps"], b["n"], pa, pb, th_low=75.0, check_orientation=True, match=b["match"], nmatches=b["nm"])

def piped(steps):
    for k in range(steps):
        b = bufs[k & 1]
        with torch.cuda.stream(sE):
            sE.wait_event(b["evm"])            # the match that read this buffer two steps ago is done
            ctxE.extract_batch_device(frames, b["kps"], b["desc"], b["n"], b["st"], cap)
            b["ev"].record(sE)
        with torch.cuda.stream(sM):
            sM.wait_event(b["ev"])
            mM.match_pairs_device(b["desc"], b["kps"], b["n"], pa, pb, th_low=75.0, check_orientation=True, match=b["match"], nmatches=b["nm"])
            b["evm"].record(sM)

for fn in (serial, piped):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(20); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    nk = float(bufs[0]["n"].sum().item())
    print(fn.__name__, round(dt * 1e3, 4), "ms/step", round(nk / dt / 1e6, 2), "M kp/s")
