"""tools/parity_windows.py [N] -- what justifies oracle/parity.py's attribution windows: how far a FLOAT32 evaluation of the reference's own
deformation arithmetic (the pinned oracle, torch CPU) lands from its float64 evaluation, in the units the windows are stated in:
ReLU pre-activations in u32 * (sum |w x| + |b|), plane coordinates in texels and in float32 ulps of the axis extent.  (CPU, ~1 minute.)"""
import importlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deform_oracle as DO  # noqa: E402

U32 = 2.0 ** -24


def measure(cfg, n, scene="shell", seed=6666, t=0.37):
    syn = importlib.import_module("4dgaussians_amd.synthetic")
    pc = syn.SynthModel(n, cfg, seed=seed, scene=scene)
    sd = {k: v.detach().clone() for k, v in pc._deformation.state_dict().items()}
    sd64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd.items()}
    L = {k: getattr(pc, k).detach() for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")}

    def run(sdx, dt):
        dec = DO.KinkDecisions(n)
        with torch.no_grad():
            DO.deform_forward(sdx, pc._deformation.args, L["_xyz"].to(dt), L["_scaling"].to(dt), L["_rotation"].to(dt), L["_opacity"].to(dt),
                              torch.cat([L["_features_dc"], L["_features_rest"]], 1).to(dt), torch.full((n, 1), t, dtype=dt), decisions=dec)
        return dec.captured
    c32, c64 = run(sd, torch.float32), run(sd64, torch.float64)
    out = {"relu_u32_units": {}, "coordinate": {}}
    for layer in c64:
        err = (c32[layer][0].double() - c64[layer][0]).abs() / (c64[layer][1] * U32)
        out["relu_u32_units"][layer] = {"max": float(err.max()), "median": float(err.median())}
    aabb, x = sd["deformation_net.grid.aabb"], L["_xyz"]
    for lvl in range(DO.count_levels(sd)):
        size = sd[f"deformation_net.grid.grids.{lvl}.0"].shape[3]
        p32 = (((x - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1.0) + 1.0) / 2.0 * (size - 1)
        a64, x64 = aabb.double(), x.double()
        p64 = (((x64 - a64[0]) * (2.0 / (a64[1] - a64[0])) - 1.0) + 1.0) / 2.0 * (size - 1)
        e = float((p32.double() - p64).abs().max())
        out["coordinate"][f"level{lvl}_size{size}"] = {"max_texels": e, "max_ulp32_of_extent": e / 2.0 ** (math.floor(math.log2(size - 1)) - 23)}
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    for cfg in ("dynerf_default", "hypernerf_default", "dnerf_bouncingballs"):
        m = measure(cfg, n)
        print(cfg, n, "Gaussians")
        for layer, v in m["relu_u32_units"].items():
            print(f"   {layer:18s} |pre32 - pre64| / (u32 * (sum|wx| + |b|)): max {v['max']:7.1f}  median {v['median']:.2f}")
        for k, v in m["coordinate"].items():
            print(f"   {k:18s} |p32 - p64|: max {v['max_texels']:.2e} texels = {v['max_ulp32_of_extent']:.1f} ulp32 of the axis extent")
