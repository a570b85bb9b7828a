#!/usr/bin/env python
"""Turn "parity unpinned" into a pin the day a CUDA box exists.

Run this ANYWHERE the reference's rasterizer wheel is installed (the un-vendored CUDA submodule named at .gitmodules:5-7 of
the reference, `pip install submodules/depth-diff-gaussian-rasterization`, NVIDIA GPU):

    python tools/dump_cuda_golden.py [--out tests/golden]

It renders the seeded scenes below with the upstream `diff_gaussian_rasterization` (forward + backward, with and without a
depth gradient, SH and precomputed-colour / precomputed-covariance variants) and writes tests/golden/raster_cuda_<case>.npz:
every input, image, depth, radii and every gradient, plus the STAGE tensors of the forward (per-Gaussian depth / radius / 2-D mean /
cov3D / conic+opacity / rgb / tiles_touched, per-pixel final T and n_contrib, num_rendered -- decoded from the extension's scratch
buffers) for the plain call and for the `[1,4,4]`-matrices + debug=True call of scene/dataset_readers.py:485-508.  Commit those files: tests/test_cuda_golden.py consumes them when
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


def _carve(buf, fields, n_of):
    """Upstream scratch buffers are carved field by field with 128-byte alignment (`obtain(chunk, ptr, count, 128)` in
    rasterizer_impl's GeometryState / ImageState ::fromChunk).  fields: (name, dtype, count-expression).  Returns {name: array} and the
    number of bytes consumed.  The layout below is restated from the published upstream source [from memory -- the raw buffers are
    dumped too, so a differing fork revision can be re-decoded later without a CUDA box]."""
    out, off = {}, 0
    base = buf.ctypes.data
    for name, dt, cnt in fields:
        addr = (base + off + 127) // 128 * 128
        off = addr - base
        n = n_of(cnt)
        nbytes = n * np.dtype(dt).itemsize
        if off + nbytes > buf.size:
            return out, off       # (layout assumption broke: keep what decoded so far)
        out[name] = np.frombuffer(buf, dtype=dt, count=n, offset=off).copy()
        off += nbytes
    return out, off


def decode_stage_tensors(geom, img, P, H, W):
    """Stage tensors the HIP tests compare (tests/test_gpu_raster.py): depths, clamped, radii, means2D, cov3D, conic_opacity, rgb,
    tiles_touched from the geometry buffer; accum_alpha (= final T), n_contrib from the image buffer."""
    g, _ = _carve(geom, [("depths", np.float32, "P"), ("clamped", np.bool_, "3P"), ("internal_radii", np.int32, "P"),
                         ("means2D", np.float32, "2P"), ("cov3D", np.float32, "6P"), ("conic_opacity", np.float32, "4P"),
                         ("rgb", np.float32, "3P"), ("tiles_touched", np.uint32, "P")],
                  lambda c: {"P": P, "2P": 2 * P, "3P": 3 * P, "4P": 4 * P, "6P": 6 * P}[c])
    i, _ = _carve(img, [("accum_alpha", np.float32, "N"), ("n_contrib", np.uint32, "N")], lambda c: H * W)
    g.update(i)
    return g


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
        # ---- stage tensors: the upstream extension called directly (the call `_RasterizeGaussians.forward` makes,
        # gaussian_renderer/__init__.py:120-128), once more with the [1,4,4] matrices and debug=True of the second construction
        # site (scene/dataset_readers.py:485-508) -- results must not depend on either
        with torch.no_grad():
            for tag, vm, pm, dbg in (("stage", torch.tensor(sc["viewmatrix"], device=dev), torch.tensor(sc["projmatrix"], device=dev), False),
                                     ("stage144", torch.tensor(sc["viewmatrix"], device=dev)[None], torch.tensor(sc["projmatrix"], device=dev)[None], True)):
                res = dgr._C.rasterize_gaussians(torch.tensor(sc["bg"], device=dev), t["means3D"].detach(), torch.Tensor([]).to(dev), t["opacities"].detach(),
                                                 t["scales"].detach(), t["rotations"].detach(), 1.0, torch.Tensor([]).to(dev), vm, pm, sc["tanfovx"], sc["tanfovy"],
                                                 sc["image_height"], sc["image_width"], t["shs"].detach(), sc["sh_degree"], torch.tensor(sc["campos"], device=dev),
                                                 False, dbg)
                res = list(res)
                num_rendered = int(res[0])
                bufs = [r for r in res[1:] if r.dtype == torch.uint8]            # geomBuffer, binningBuffer, imgBuffer (in this order)
                imgs = [r for r in res[1:] if r.dtype in (torch.float32, torch.int32)]
                out[f"{tag}.num_rendered"] = np.array([num_rendered], np.int64)
                out[f"{tag}.color"] = imgs[0].cpu().numpy()
                geom, imgb = bufs[0].cpu().numpy(), bufs[-1].cpu().numpy()
                for k, v in decode_stage_tensors(geom, imgb, sc["means3D"].shape[0], sc["image_height"], sc["image_width"]).items():
                    out[f"{tag}.{k}"] = v
                if tag == "stage" and sc["means3D"].shape[0] <= 20000:      # raw bytes of the small cases: re-decodable without a CUDA box
                    out["stage.raw_geomBuffer"], out["stage.raw_imgBuffer"] = geom, imgb
        path = os.path.join(args.out, f"raster_cuda_{name}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
