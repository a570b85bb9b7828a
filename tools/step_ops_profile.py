"""Which torch-side ops (copies, fills, elementwise kernels) one bench step launches besides the libfdgs kernels: torch.profiler over a few
steps, grouped by op with input shapes.  usage (GPU box): python tools/step_ops_profile.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdgs = importlib.import_module("4dgaussians_amd")
syn = fdgs.synthetic
dev = torch.device("cuda:0")
N, W, H = 300_000, 1352, 1014
pc = syn.SynthModel(N, "dynerf_default", seed=6666, device=dev)
fdgs.densify.spatial_reorder(pc)
cams = [c.to(dev) for c in syn.orbit_cameras(W, H, n=16)]
target = torch.rand(3, H, W, device=dev)
params = [p for p in pc.parameters() if p.requires_grad]
acc = torch.zeros(3, device=dev)
import ctypes
_nb = ctypes.c_size_t()
fdgs._lib.check(fdgs._lib.lib().fdgs_l1_stats_scratch_bytes(_nb))
l1_scratch = torch.zeros(_nb.value, dtype=torch.uint8, device=dev)
dimg = torch.empty(3, H, W, device=dev)
L = fdgs._lib.lib()
pipe, bg = syn.PipelineParams(), torch.zeros(3, device=dev)


def step(i):
    for p_ in params:
        p_.grad = None
    res = fdgs.render(cams[i % len(cams)], pc, pipe, bg, stage="fine")
    img = res["render"]
    fdgs._lib.check(L.fdgs_l1_stats_assign(fdgs._lib.stream_ptr(), img.numel(), fdgs._lib.ptr(img), fdgs._lib.ptr(target), 1.0 / img.numel(),
                                           fdgs._lib.ptr(dimg), fdgs._lib.ptr(acc), fdgs._lib.ptr(l1_scratch)))
    img.backward(dimg)


for i in range(4):
    step(i)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=50))
