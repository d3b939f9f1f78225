"""GPU-box probe: raw H2D / D2H bandwidth between page-locked host memory and the device (the bound of the host-fed pipeline)."""
import json
import time

import torch

res = {}
for mb in (20, 157):
    n = mb * 1000 * 1000
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        res["%s_%dMB_GBps" % (name, mb)] = n * 10 / (time.perf_counter() - t0) / 1e9
# both directions at once on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
res["duplex_157MB_each_way_GBps"] = n * 10 / (time.perf_counter() - t0) / 1e9
print(json.dumps(res))
