#!/usr/bin/env python
"""Turn "parity unpinned" into a pin the day a CUDA box exists.

Run this ANYWHERE the reference's rasterizer wheel is installed (the un-vendored CUDA submodule named at .gitmodules:5-7 of
the reference, `pip install submodules/depth-diff-gaussian-rasterization`, NVIDIA GPU):

    python tools/dump_cuda_golden.py [--out tests/golden]

It renders the seeded scenes below with the upstream `diff_gaussian_rasterization` (forward + backward, with and without a
depth gradient, SH and precomputed-colour / precomputed-covariance variants) and writes tests/golden/raster_cuda_<case>.npz:
every input, image, depth, radii and every gradient.  Commit those files: tests/test_cuda_golden.py consumes them when
present -- the CPU leg pins oracle/raster_oracle.c to them, the GPU leg compares the HIP rasterizer with them directly.
The seeded scenes come from tests/scenes.py (torch-CPU / numpy; libfdgs.so is NOT needed), so a bare checkout of this repo
next to the upstream wheel is enough.

Nothing in this repository can execute it: there is no CUDA device and no upstream source here.
"""
import argparse
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CASES = {
    "cfg1_20k_400": dict(n=20000, width=400, height=400, seed=0, theta=30.0),
    "ragged_3k_201x77": dict(n=3000, width=201, height=77, seed=1, theta=-100.0, scale_boost=4.0),
    "deg1_500_64x48": dict(n=500, width=64, height=48, seed=2, theta=170.0, sh_degree=1, scale_boost=10.0),
    "deg0_culled_5k": dict(n=5000, width=320, height=240, seed=3, theta=0.0, sh_degree=0, extent=3.0),
    "cfg2_100k_800": dict(n=100000, width=800, height=800, seed=11, theta=100.0, scale_boost=2.0),
}


def upstream_module():
    # this repo ships an import-name shim `diff_gaussian_rasterization/` at its root: make sure the UPSTREAM wheel is loaded
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    import diff_gaussian_rasterization as dgr
    if not hasattr(dgr, "_C"):
        raise SystemExit("imported a diff_gaussian_rasterization without the CUDA extension `_C`: not the upstream wheel")
    return dgr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    args = ap.parse_args()
    dgr = upstream_module()
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    from scenes import raster_scene       # numpy/torch-CPU scene builder shared with the tests (seeded)
    dev = torch.device("cuda")
    for name, case in CASES.items():
        sc = raster_scene(**case)
        t = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
        rs = dgr.GaussianRasterizationSettings(
            image_height=sc["image_height"], image_width=sc["image_width"], tanfovx=sc["tanfovx"], tanfovy=sc["tanfovy"],
            bg=torch.tensor(sc["bg"], device=dev), scale_modifier=1.0, viewmatrix=torch.tensor(sc["viewmatrix"], device=dev),
            projmatrix=torch.tensor(sc["projmatrix"], device=dev), sh_degree=sc["sh_degree"], campos=torch.tensor(sc["campos"], device=dev),
            prefiltered=False, debug=False)
        out = {"in." + k: v for k, v in sc.items() if isinstance(v, np.ndarray)}
        out["in.meta"] = np.array([sc["image_height"], sc["image_width"], sc["tanfovx"], sc["tanfovy"], sc["sh_degree"]], np.float64)
        rng = np.random.default_rng(7)
        for variant in ("color", "color_depth"):
            for v in t.values():
                v.grad = None
            m2d = torch.zeros_like(t["means3D"], requires_grad=True)
            color, radii, depth = dgr.GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None,
                                                             opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                                             cov3D_precomp=None)
            dc = rng.standard_normal(tuple(color.shape)).astype(np.float32) / color.numel()
            dd = rng.standard_normal(tuple(depth.shape)).astype(np.float32) / depth.numel()
            loss = (color * torch.tensor(dc, device=dev)).sum()
            if variant == "color_depth":
                loss = loss + (depth * torch.tensor(dd, device=dev)).sum()
            loss.backward()
            out.update({"out.color": color.detach().cpu().numpy(), "out.depth": depth.detach().cpu().numpy(),
                        "out.radii": radii.cpu().numpy(), f"{variant}.dL_dcolor": dc, f"{variant}.dL_ddepth": dd,
                        f"{variant}.grad.means2D": m2d.grad.cpu().numpy()})
            for k, v in t.items():
                out[f"{variant}.grad.{k}"] = v.grad.cpu().numpy()
        path = os.path.join(args.out, f"raster_cuda_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
