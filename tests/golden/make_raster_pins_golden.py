"""Generates tests/golden/raster_pins.npz: everything of the rasterizer's arithmetic that the reference DOES hold in-tree
as python, evaluated by the reference's own functions imported from /root/reference (run here; the GPU box has no reference):
  * cov3D = strip_symmetric(L L^T), L = build_scaling_rotation(scaling_modifier * s, q)   utils/general_utils.py:63-116,
                                                                                           scene/gaussian_model.py:29-34
  * getWorld2View2 / getProjectionMatrix / Camera.{world_view_transform, full_proj_transform, camera_center}
                                                                 utils/graphics_utils.py:38-71, scene/cameras.py:59-64
  * eval_sh colours  clamp_min(eval_sh(deg, shs_view, dir) + 0.5, 0)        utils/sh_utils.py:57-112,
                                                                              gaussian_renderer/__init__.py:106-111
    python tests/golden/make_raster_pins_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import densify_oracle as DN  # noqa: E402  (reference import recipe + device="cuda" redirect)

synthetic = importlib.import_module("4dgaussians_amd.synthetic")


def main():
    DN.import_reference_gaussian_model()          # puts /root/reference on sys.path with the stubs the imports need
    import utils.general_utils as gu
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2
    from utils.sh_utils import eval_sh
    from scene.cameras import Camera
    d = {}
    g = synthetic.make_gaussians(512, seed=31)
    s, q = torch.exp(g["scaling"]), g["rotation"]                  # raw quaternions: build_rotation normalises
    for mod in (1.0, 0.7):
        with DN.reference_on_cpu():
            L = gu.build_scaling_rotation(mod * s, q)
            cov = gu.strip_symmetric(L @ L.transpose(1, 2))
        d[f"cov3D.mod{mod}"] = cov.numpy()
    d["cov.scales"], d["cov.rotations_raw"] = s.numpy(), q.numpy()
    # cameras: the poses synthetic.make_camera builds, pushed through the reference's Camera class
    poses = [(400, 400, 30.0, -30.0, 4.0), (1352, 1014, -60.0, -30.0, 4.0), (536, 960, 170.0, 12.0, 2.5), (201, 77, 0.0, -80.0, 6.0)]
    for i, (W, H, th, ph, rad) in enumerate(poses):
        c2w = synthetic._pose_spherical(th, ph, rad)
        m = np.linalg.inv(c2w)
        R = -np.transpose(m[:3, :3]).copy()
        R[:, 0] = -R[:, 0]
        T = -m[:3, 3]
        cam = synthetic.make_camera(W, H, theta_deg=th, phi_deg=ph, radius=rad)
        ref = Camera(colmap_id=0, R=R, T=T, FoVx=cam.FoVx, FoVy=cam.FoVy, image=torch.zeros(3, H, W), gt_alpha_mask=None,
                     image_name="x", uid=0, data_device="cpu", time=0.0)
        d[f"cam{i}.pose"] = np.array([W, H, th, ph, rad], np.float64)
        d[f"cam{i}.R"], d[f"cam{i}.T"] = R.astype(np.float64), T.astype(np.float64)
        d[f"cam{i}.w2v"] = getWorld2View2(R, T)
        d[f"cam{i}.proj"] = getProjectionMatrix(0.01, 100.0, cam.FoVx, cam.FoVy).numpy()
        d[f"cam{i}.world_view_transform"] = ref.world_view_transform.numpy()
        d[f"cam{i}.full_proj_transform"] = ref.full_proj_transform.numpy()
        d[f"cam{i}.camera_center"] = ref.camera_center.numpy()
    # SH colours as render() forms them on the python path
    shs = torch.cat([g["features_dc"], g["features_rest"]], 1)
    cam = synthetic.make_camera(400, 400, theta_deg=30.0)
    dirs = g["xyz"] - cam.camera_center.repeat(shs.shape[0], 1)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    for deg in range(4):
        d[f"sh.colors.deg{deg}"] = torch.clamp_min(eval_sh(deg, shs.transpose(1, 2).view(-1, 3, 16), dirs) + 0.5, 0.0).numpy()
    d["sh.shs"], d["sh.xyz"], d["sh.campos"] = shs.numpy(), g["xyz"].numpy(), cam.camera_center.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raster_pins.npz")
    np.savez_compressed(path, **d)
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
