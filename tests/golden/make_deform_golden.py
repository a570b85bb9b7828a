"""Generates tests/golden/deform_<cfg>.npz by running the REFERENCE's own deform_network (imported from
/root/reference per SURVEY.md Appendix E) on seeded inputs.  Run here (the GPU box has no /root/reference):
    python tests/golden/make_deform_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import deform_oracle as DO  # noqa: E402

synthetic = importlib.import_module("4dgaussians_amd.synthetic")


def main():
    deform_network = DO.import_reference_deform_network()
    for cfg in ("dnerf_bouncingballs", "hypernerf_default", "dynerf_default"):
        torch.manual_seed(6666)
        args = synthetic.deform_args(cfg)
        # keep fixtures small: shrink plane resolution (the arithmetic is resolution-agnostic)
        args.kplanes_config["resolution"] = [8, 8, 8, 6]
        net = deform_network(args)
        with torch.no_grad():
            for name, p in net.named_parameters():
                if "grids" in name:
                    p.add_(0.1 * torch.randn_like(p))
        n = 96
        g = synthetic.make_gaussians(n, seed=7)
        xyz = g["xyz"] * 1.1  # some points outside the aabb -> border clamp
        net.deformation_net.set_aabb([1.3, 1.3, 1.3], [-1.3, -1.3, -1.3])
        shs = torch.cat([g["features_dc"], g["features_rest"]], 1)
        t = torch.rand(n, 1)
        t[:4] = torch.tensor([[0.0], [1.0], [0.5], [1.2]])
        # forward + backward of the REFERENCE modules: outputs, and the gradients of sum(out * w) for seeded weights w
        # with respect to every input and every parameter (the GPU tests compare the HIP backward with these directly)
        ins = [x.clone().requires_grad_(True) for x in (xyz, g["scaling"], g["rotation"], g["opacity"], shs)]
        out = net(*ins, t)
        ws = [torch.randn(o.shape, generator=torch.Generator().manual_seed(99 + i)) for i, o in enumerate(out)]
        pnames = [k for k, p in net.named_parameters() if p.requires_grad]
        grads = torch.autograd.grad(sum((o * w).sum() for o, w in zip(out, ws)), ins + [dict(net.named_parameters())[k] for k in pnames],
                                    allow_unused=True)
        out = [o.detach() for o in out]
        d = {"sd." + k: v.detach().numpy() for k, v in net.state_dict().items()}
        for k, w in zip(("xyz", "scales", "rot", "opacity", "shs"), ws):
            d["w." + k] = w.numpy()
        for k, gr in zip(["in.xyz", "in.scales", "in.rot", "in.opacity", "in.shs"] + ["sd." + k for k in pnames], grads):
            if gr is not None:
                d["grad." + k] = gr.numpy()
        for k, v in zip(("xyz", "scales", "rot", "opacity", "shs", "t"), (xyz, g["scaling"], g["rotation"], g["opacity"], shs, t)):
            d["in." + k] = v.numpy()
        for k, v in zip(("xyz", "scales", "rot", "opacity", "shs"), out):
            d["out." + k] = v.numpy()
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"deform_{cfg}.npz")
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
