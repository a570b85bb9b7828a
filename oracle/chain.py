"""render() restated with the CPU oracles -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The chain the full-size GPU parity tests, bench.py's parity block and __graft_entry__.smoke() check the HIP path against: deformation oracle
(pinned to the reference modules, oracle/deform_oracle.py) -> C rasterizer oracle forward -> L1-loss image gradient against a seeded random
target -> C analytic backward -> torch-CPU autograd (or the float64 evaluation) through the deformation.  Itself pinned to the reference's OWN
render() source: tests/test_reference_render_over_shim.py (CPU leg).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / parity leg may import this module.
"""
import math

import numpy as np
import torch


def oracle_render_chain(pc, cam, stage="fine", target_seed=0, with_depth_grad=False, grad_dtype=torch.float32, both=False, deformed=None):
    """render() restated with the CPU oracles: deformation oracle (pinned to the reference modules) -> C rasterizer oracle
    forward -> L1-loss image gradient against a seeded random target -> C analytic backward -> torch-CPU autograd through the
    deformation.  `pc` is a CPU SynthModel.  Returns (oracle object, dL/dimage, dL/ddepth, {parameter name: gradient or None}).
    `grad_dtype=torch.float64`: the autograd pass through the deformation is a float64 re-evaluation of the same oracle function fed
    with the same (float32-chain) upstream gradients (oracle.deform_oracle.backward_float64) -- THE gradient reference of the full-size
    checks; the returned dict then also carries "__ctx" = (sd, flags, leaves, time, upstream gradients), what oracle.parity.attribute needs
    to name the rows on which an implementation took a ReLU / texel-cell decision the other way.  `both=True` also returns the float32
    autograd gradients under "__float32" (tools only).
    `deformed` (fine stage): (means3D, scales, rotations, opacity, shs) as CPU tensors -- the IMPLEMENTATION's deformed Gaussians.  The C
    rasterizer oracle then blends THOSE (forward and backward) instead of the deformation oracle's own outputs, and the upstream gradients
    it returns are chained through the oracle's deformation as before.  Why: the rasterizer takes discrete decisions on its inputs -- above
    all the front-to-back ORDER of two Gaussians whose view depths differ by less than the deformation's float32 rounding (at 2 M Gaussians
    there are ~1e5 such pairs; when one of them is large, opaque and overlapping, eight tiles change colour by 0.1 although every input
    agrees to 1e-6, and which pair flips depends on the summation order of the host's GEMM threads).  Parity is therefore factored:
    deformation outputs against the deformation oracle (returned under "__deformed" for that comparison), rasterizer against the rasterizer
    oracle ON THE SAME INPUTS, gradients chained."""
    from oracle import deform_oracle as DO
    from oracle.raster_oracle import RasterOracle
    n = pc._xyz.shape[0]
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "poc" not in k and "aabb" not in k)
          for k, v in pc._deformation.state_dict().items()}
    leaves = {k: getattr(pc, k).detach().clone().requires_grad_(True)
              for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")}
    shs = torch.cat([leaves["_features_dc"], leaves["_features_rest"]], 1)
    if stage == "fine":
        m3, sc, rot, op, sh = DO.deform_forward(sd, pc._deformation.args, leaves["_xyz"], leaves["_scaling"], leaves["_rotation"],
                                                leaves["_opacity"], shs, torch.full((n, 1), cam.time), activate=True)
    else:
        m3, sh = leaves["_xyz"], shs
        sc, op = torch.exp(leaves["_scaling"]), torch.sigmoid(leaves["_opacity"])
        rot = torch.nn.functional.normalize(leaves["_rotation"])
    f = lambda x: np.ascontiguousarray(x.detach().cpu().float().numpy())
    H, W = cam.image_height, cam.image_width
    r3, rsc, rrot, rop, rsh = (m3, sc, rot, op, sh) if deformed is None else [d.reshape(t.shape) for d, t in zip(deformed, (m3, sc, rot, op, sh))]
    o = RasterOracle(means3D=f(r3), scales=f(rsc), rotations=f(rrot), opacities=f(rop), shs=f(rsh), viewmatrix=f(cam.world_view_transform),
                     projmatrix=f(cam.full_proj_transform), campos=f(cam.camera_center), bg=np.zeros(3, np.float32), image_height=H,
                     image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), sh_degree=3)
    rng = np.random.default_rng(target_seed)
    target = rng.random(o.color.shape).astype(np.float32)
    dc = (np.sign(o.color - target) / o.color.size).astype(np.float32)
    dd = (rng.standard_normal(o.depth.shape).astype(np.float32) / o.depth.size) if with_depth_grad else None
    go = o.backward(dc, dd)
    outs = [m3, sc, rot, op, sh]
    gouts = [torch.tensor(go["means3D"]), torch.tensor(go["scales"]), torch.tensor(go["rotations"]),
             torch.tensor(go["opacities"]).reshape(op.shape), torch.tensor(go["shs"]).reshape(sh.shape)]
    if grad_dtype != torch.float32 and stage == "fine":
        grads = DO.backward_float64(sd, pc._deformation.args, leaves, cam.time, gouts)
        grads["__ctx"] = (sd, pc._deformation.args, {k: v.detach() for k, v in leaves.items()}, cam.time, gouts)
        if both:
            wanted = list(leaves.values()) + [v for v in sd.values() if v.requires_grad]
            wnames = list(leaves.keys()) + ["_deformation." + k for k, v in sd.items() if v.requires_grad]
            g32 = torch.autograd.grad(outs, wanted, grad_outputs=gouts, allow_unused=True)
            grads["__float32"] = {k: (None if g is None else g.numpy()) for k, g in zip(wnames, g32)}
    else:
        wanted = list(leaves.values()) + ([v for v in sd.values() if v.requires_grad] if stage == "fine" else [])
        wnames = list(leaves.keys()) + (["_deformation." + k for k, v in sd.items() if v.requires_grad] if stage == "fine" else [])
        g_ref = torch.autograd.grad(outs, wanted, grad_outputs=gouts, allow_unused=True)
        grads = {k: (None if g is None else g.numpy()) for k, g in zip(wnames, g_ref)}
    grads["__means2D"] = go["means2D"]
    grads["__deformed"] = [t.detach() for t in (m3, sc, rot, op, sh)]      # the deformation oracle's own outputs
    return o, dc, dd, grads
