"""The reference's OWN float32 modules as a comparator on the device (round 6) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
__graft_entry__.smoke() and bench.py's rank-0 parity leg may import this module.

`reference_f32_frame` runs the reference's `render()` (gaussian_renderer/__init__.py:18) with the reference's `deform_network`
(scene/deformation.py:161; byte-compiled into oracle/_ref, imported sourceless by oracle/ref_modules.py) as torch float32 ops on the SAME
device, state copied from the model under test, over this repository's rasterizer shim, forward + backward with the given upstream image
gradient.  Both legs of the comparison then share the HIP rasterizer and differ in the deformation only: torch's GEMMs / grid_sample /
autograd against the fused HIP kernels.  This is the apples-to-apples figure for north_star's 1e-3 gradient tolerance -- an f32 path against
the reference's own f32 path, RAW, no kink attribution -- reported next to the float64-oracle figure (oracle/parity.py), which attributes the
rows where a float32 evaluation takes a ReLU / texel-cell decision the other way.
"""
import copy
import importlib

import numpy as np
import torch

from . import ref_modules

GAUSS = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


def available():
    return ref_modules.available()


def _rel(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a))


def reference_twin(pc, hyper):
    """A copy of the model whose `_deformation` is the REFERENCE's deform_network built from `hyper` (the args namespace the model's own
    network was built from), state_dict copied."""
    ns = ref_modules.load()
    dev = pc._xyz.device
    net = ns.deform_network(hyper)
    net.load_state_dict(pc._deformation.state_dict(), strict=True)
    own = pc._deformation
    pc._deformation = None                      # (do not deep-copy the network under test)
    try:
        twin = copy.deepcopy(pc)
    finally:
        pc._deformation = own
    twin._deformation = net.to(dev)
    return ns, twin


def reference_f32_frame(pc, hyper, cam, pipe, bg, dcolor):
    """(image, radii, {parameter name: gradient}, viewspace gradient) of the reference's render() + deform_network in float32 on the device."""
    ns, twin = reference_twin(pc, hyper)
    res = ns.render(cam, twin, pipe, bg, stage="fine")
    res["render"].backward(dcolor)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in twin.named_parameters() if p.grad is not None}
    out = (res["render"].detach().cpu().numpy(), res["radii"].cpu().numpy(), grads, res["viewspace_points"].grad.detach().cpu().numpy())
    del twin, res
    torch.cuda.empty_cache()
    return out


def compare_raw(pc, ref_grads, ref_img=None, img=None):
    """Group-wise RAW relative L2 of the model's .grad against the reference-f32 gradients (same groups as bench.py's parity block)."""
    named = dict(pc.named_parameters())
    groups = {"xyz": ["_xyz"], "scaling": ["_scaling"], "rotation": ["_rotation"], "opacity": ["_opacity"], "f_dc": ["_features_dc"],
              "f_rest": ["_features_rest"], "planes": [k for k in ref_grads if "grids" in k],
              "mlp": [k for k in ref_grads if k.startswith("_deformation.") and "grids" not in k]}
    out = {}
    for gname, keys in groups.items():
        keys = [k for k in keys if k in ref_grads and named[k].grad is not None]
        if not keys:
            continue
        a = np.concatenate([named[k].grad.detach().cpu().numpy().ravel() for k in keys])
        b = np.concatenate([ref_grads[k].ravel() for k in keys])
        out[gname] = float(f"{_rel(a, b):.3e}")
    worst = max(((k, _rel(named[k].grad.detach().cpu().numpy(), v)) for k, v in ref_grads.items()
                 if named[k].grad is not None and float(np.abs(v).max()) > 0), key=lambda kv: kv[1])
    rep = {"groups": out, "worst_single_tensor": {"name": worst[0], "rel_l2": float(f"{worst[1]:.3e}")}}
    if ref_img is not None and img is not None:
        d = np.abs(np.asarray(img, np.float64) - np.asarray(ref_img, np.float64))
        rep["image_mean_abs"], rep["image_max_abs"] = float(f"{d.mean():.3e}"), float(f"{d.max():.3e}")
    return rep
