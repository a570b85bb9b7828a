"""Development aid: cProfile of the Python/ctypes host side of one bench frame (where does the host time go?)."""
import cProfile
import importlib
import io
import os
import pstats
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
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(6666)).to(dev)
params = [p for p in pc.parameters() if p.requires_grad]
acc = torch.zeros(3, device=dev)
dimg = torch.empty(3, H, W, device=dev)
st, ptr = fdgs._lib.stream_ptr, fdgs._lib.ptr


def step(i):
    cam = cams[i % len(cams)]
    for p_ in params:
        p_.grad = None
    res = fdgs.render(cam, pc, pipe, bg, stage="fine")
    img = res["render"]
    acc.zero_()
    fdgs._lib.check(L.fdgs_l1_stats(st(), img.numel(), ptr(img), ptr(target), 1.0 / img.numel(), ptr(dimg), ptr(acc)))
    img.backward(dimg)


for i in range(5):
    step(i)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for i in range(K):
    step(i)
torch.cuda.synchronize()
print(f"wall {(time.perf_counter() - t0) / K * 1e3:.3f} ms/frame; cpu_count={os.cpu_count()} torch threads={torch.get_num_threads()}")
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
