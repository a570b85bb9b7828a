"""Development aid: per-frame host time of forward / backward and caching-allocator activity (device mallocs per frame)."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

fdgs = importlib.import_module("4dgaussians_amd")
syn = fdgs.synthetic
dev = torch.device("cuda:0")
L = fdgs._lib.lib()
N, W, H, dcfg = bench.WORKLOADS["cfg4_dynerf_300k_1352x1014"]
pc = syn.SynthModel(N, dcfg, seed=6666, device=dev)
pipe = syn.PipelineParams()
bg = torch.zeros(3, device=dev)
cams = [c.to(dev) for c in syn.orbit_cameras(W, H, n=160)]
dimg = torch.rand(3, H, W, device=dev)
params = [p for p in pc.parameters() if p.requires_grad]
for i in range(12):
    for p_ in params:
        p_.grad = None
    s0 = torch.cuda.memory_stats()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = fdgs.render(cams[i], pc, pipe, bg, stage="fine")
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    res["render"].backward(dimg)
    t3 = time.perf_counter(); torch.cuda.synchronize(); t4 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print(f"frame {i}: fwd host {1e3*(t1-t0):.2f} ms (+drain {1e3*(t2-t1):.2f}), bwd host {1e3*(t3-t2):.2f} ms (+drain {1e3*(t4-t3):.2f}); "
          f"device mallocs +{s1['num_device_alloc']-s0['num_device_alloc']}, frees +{s1['num_device_free']-s0['num_device_free']}, "
          f"retries +{s1['num_alloc_retries']-s0['num_alloc_retries']}, reserved {s1['reserved_bytes.all.current']/2**30:.2f} GiB, "
          f"allocated peak {s1['allocated_bytes.all.peak']/2**30:.2f} GiB")
