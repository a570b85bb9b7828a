#!/usr/bin/env python
"""tools/d1_ab.py -- the deformation forward alone on the bench scene (config 4 shape), with and without saved activations; per-kernel time from
the library's own HIP-event timing.  A/B through the environment: FDGS_LIB (a variant build), FDGS_D1_FORM (16 / 32), FDGS_D1_WGS.  (development, GPU)"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdgs = importlib.import_module("4dgaussians_amd")
syn = fdgs.synthetic
dev = torch.device("cuda:0")
N = int(os.environ.get("N", "300000"))
pc = syn.SynthModel(N, os.environ.get("CFG", "dynerf_default"), seed=6666, device=dev)
fdgs.densify.spatial_reorder(pc)
L = fdgs._lib.lib()


def run(grad, n=20):
    def once():
        outs = fdgs.deformation.deform(pc._deformation, pc._xyz, pc._scaling, pc._rotation, pc._opacity, shs_dc=pc._features_dc,
                                       shs_rest=pc._features_rest, time=0.37, activate=True)
        return outs
    ctx = torch.enable_grad() if grad else torch.no_grad()
    with ctx:
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        L.fdgs_timing_enable(1)
        for _ in range(n):
            once()
        buf = ctypes.create_string_buffer(1 << 14)
        L.fdgs_timing_report(buf, len(buf), 1)
        L.fdgs_timing_enable(0)
    for line in buf.value.decode().strip().splitlines():
        name, cnt, tot = line.split()
        if name == "deform_fwd":
            return float(tot) / int(cnt)


tag = f"lib={os.path.basename(os.environ.get('FDGS_LIB', 'libfdgs.so'))} form={os.environ.get('FDGS_D1_FORM', '16')} wgs={os.environ.get('FDGS_D1_WGS', '-')}"
print(f"[d1_ab] {tag}: deform_fwd saving {run(True):.4f} ms, forward-only {run(False):.4f} ms", flush=True)
