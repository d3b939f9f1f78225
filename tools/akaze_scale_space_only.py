"""runs only scale space + Hessian (for PMC passes): python tools/akaze_scale_space_only.py [batch] [steps]"""
import importlib
import sys

import torch

sys.path.insert(0, ".")
afv = importlib.import_module("anyfeature-vslam_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = afv.AkazeContext(afv.akaze.default_params(max_batch=B))
frames = torch.from_numpy(afv.synth.corners_batch(1, B, 1280, 720)).cuda()
for _ in range(steps):
    ctx.scale_space_device(frames)
ctx.synchronize()
