"""GPU-box probe (run under `timeout`, in its own process): does this HIP runtime accept hipStreamLegacy ((hipStream_t)1) in the
*_device entry points?  Prints 'legacy ok' or dies; the product never passes it (see _lib.launch_ordered)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

afv = importlib.import_module("anyfeature-vslam_amd")
ctx = afv.Context(max_batch=4)
frames = torch.from_numpy(afv.synth.corners_batch(1, 4)).cuda()
torch.cuda.synchronize()
print("calling with stream handle 1", flush=True)
kps, desc, n, st = ctx.extract_batch_device(frames, stream=1)
torch.cuda.synchronize()
print("legacy ok", n.cpu().tolist(), flush=True)
