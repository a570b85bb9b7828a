"""Synthetic scenes and cameras for tests and bench.

Everything here is host-side numpy/torch-CPU plumbing: it generates inputs of the shapes named by
BASELINE.json's configs (there are no datasets in the container).  Camera conventions follow the reference:
`scene/cameras.py:59-64` (transposed world->view and full projection, row-vector convention),
`utils/graphics_utils.py:38-71` (getWorld2View2 / getProjectionMatrix), and the D-NeRF orbit of
`scene/dataset_readers.py:218-257` (pose_spherical(theta, -30, 4.0)).
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

SH_C0 = 0.28209479177387814


@dataclass
class SynthCamera:
    """Duck-type of the reference `Camera` fields that `render()` reads (scene/cameras.py:29-64)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor
    time: float
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device):
        return SynthCamera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                           self.world_view_transform.to(device), self.full_proj_transform.to(device),
                           self.camera_center.to(device), self.time, self.znear, self.zfar)


def _pose_spherical(theta_deg, phi_deg, radius):
    th, ph = math.radians(theta_deg), math.radians(phi_deg)
    trans = np.eye(4); trans[2, 3] = radius
    rphi = np.array([[1, 0, 0, 0], [0, math.cos(ph), -math.sin(ph), 0], [0, math.sin(ph), math.cos(ph), 0], [0, 0, 0, 1.0]])
    rth = np.array([[math.cos(th), 0, -math.sin(th), 0], [0, 1, 0, 0], [math.sin(th), 0, math.cos(th), 0], [0, 0, 0, 1.0]])
    flip = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    return (flip @ rth @ rphi @ trans).astype(np.float32)


def world_to_view(R, T):
    """World->view 4x4 for camera rotation R (stored transposed, as in the reference) and translation T."""
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


def projection_matrix(znear, zfar, fovx, fovy):
    tx, ty = math.tan(fovx / 2), math.tan(fovy / 2)
    top, right = ty * znear, tx * znear
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(width, height, theta_deg=0.0, time=0.0, phi_deg=-30.0, radius=4.0, camera_angle_x=0.6911112070083618):
    c2w = _pose_spherical(theta_deg, phi_deg, radius)
    m = np.linalg.inv(c2w)
    R = -np.transpose(m[:3, :3]).copy()
    R[:, 0] = -R[:, 0]
    T = -m[:3, 3]
    focal = 0.5 * width / math.tan(camera_angle_x / 2)  # same focal on both axes
    fovx = 2 * math.atan(width / (2 * focal))
    fovy = 2 * math.atan(height / (2 * focal))
    wvt = torch.tensor(world_to_view(R, T)).transpose(0, 1).contiguous()
    proj = torch.tensor(projection_matrix(0.01, 100.0, fovx, fovy)).transpose(0, 1).contiguous()
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return SynthCamera(int(width), int(height), fovx, fovy, wvt, full, center, float(time))


def orbit_cameras(width, height, n=160, n_times=None):
    """The reference's video orbit: theta in linspace(-180,180,n+1)[:-1], t = linspace(0,1,n)."""
    thetas = np.linspace(-180, 180, n + 1)[:-1]
    times = np.linspace(0, 1, n) if n_times is None else (np.arange(n) % n_times) / max(n_times - 1, 1)
    return [make_camera(width, height, float(th), float(t)) for th, t in zip(thetas, times)]


SHELL_RADII = (0.75, 1.0, 1.25)       # scene="shell": three concentric spheres (x the extent), thin (sigma 0.01)
SHELL_OPACITY = (0.02, 0.2)          # ... of translucent splats: U(0.02, 0.2)
SHELL_SCALE = 0.5                     # ... half the cube scene's footprint


def make_gaussians(n, seed=6666, sh_degree=3, extent=1.3, device="cpu", scene="cube"):
    """Canonical Gaussians as the reference stores them (scene/gaussian_model.py:137-164 layout):
    xyz [N,3], log-scales [N,3], raw quaternions [N,4] (w first), opacity logits [N,1], features_dc [N,1,3],
    features_rest [N,15,3].
    scene="cube" is SURVEY 8d's generator (uniform in a cube, opacity U(0.05, 0.95)): a volume that occludes itself -- most of its visible
    Gaussians sit behind saturated pixels and receive no gradient.  scene="shell" is the other regime a trained model can be in:
    surface-like (three thin concentric spheres), translucent (opacity U(0.02, 0.2)), smaller splats -- nearly every visible Gaussian is
    blended by some pixel and receives a gradient, so the deformation backward has no dead tiles to skip."""
    g = torch.Generator().manual_seed(seed)
    if scene == "cube":
        xyz = (torch.rand(n, 3, generator=g) * 2 - 1) * extent
    elif scene == "shell":
        d = torch.randn(n, 3, generator=g)
        d = d / d.norm(dim=1, keepdim=True)
        r = torch.tensor(SHELL_RADII)[torch.randint(0, len(SHELL_RADII), (n,), generator=g)] * extent
        xyz = d * (r + 0.01 * torch.randn(n, generator=g))[:, None]
    else:
        raise ValueError(f"unknown synthetic scene {scene!r}")
    base = math.log(0.5 * 2 * extent * n ** (-1.0 / 3.0) * (SHELL_SCALE if scene == "shell" else 1.0))
    scaling = base + 0.3 * torch.randn(n, 3, generator=g)
    rotation = torch.randn(n, 4, generator=g)
    if scene == "shell":
        u = torch.rand(n, 1, generator=g) * (SHELL_OPACITY[1] - SHELL_OPACITY[0]) + SHELL_OPACITY[0]
    else:
        u = torch.rand(n, 1, generator=g) * 0.9 + 0.05
    opacity = torch.log(u / (1 - u))
    dc = ((torch.rand(n, 1, 3, generator=g) - 0.5) / SH_C0)
    rest = 0.05 * torch.randn(n, (sh_degree + 1) ** 2 - 1, 3, generator=g)
    out = dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity, features_dc=dc, features_rest=rest)
    return {k: v.to(device).contiguous() for k, v in out.items()}


# deformation hyper-parameters of the BASELINE.json configs (values from arguments/<dataset>/*.py of the reference)
DEFORM_CONFIGS = {
    # arguments/dnerf/bouncingballs.py:3-9 + dnerf_default.py:22-32
    "dnerf_bouncingballs": dict(net_width=64, defor_depth=0, bounds=1.6, multires=[1, 2],
                                kplanes_config=dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=32,
                                                    resolution=[64, 64, 64, 75]),
                                no_dx=False, no_ds=False, no_dr=False, no_do=True, no_dshs=True),
    # arguments/hypernerf/default.py:1-15 (+ broom2.py T-res 100 is a variant)
    "hypernerf_default": dict(net_width=128, defor_depth=1, bounds=1.6, multires=[1, 2, 4],
                              kplanes_config=dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=16,
                                                  resolution=[64, 64, 64, 150]),
                              no_dx=False, no_ds=False, no_dr=False, no_do=True, no_dshs=True),
    # arguments/dynerf/default.py:1-21
    "dynerf_default": dict(net_width=128, defor_depth=0, bounds=1.6, multires=[1, 2],
                           kplanes_config=dict(grid_dimensions=2, input_coordinate_dim=4, output_coordinate_dim=16,
                                               resolution=[64, 64, 64, 150]),
                           no_dx=False, no_ds=False, no_dr=False, no_do=False, no_dshs=False),
}


def deform_args(name, **overrides):
    """argparse-like namespace with every field `deform_network.__init__` reads
    (scene/deformation.py:25-32,164-177; scene/hexplane.py:110-146)."""
    from argparse import Namespace
    base = dict(timebase_pe=4, posebase_pe=10, scale_rotation_pe=2, opacity_pe=2, timenet_width=64, timenet_output=32,
                no_grid=False, empty_voxel=False, grid_pe=0, static_mlp=False, apply_rotation=False)
    cfg = dict(DEFORM_CONFIGS[name])
    cfg["kplanes_config"] = dict(cfg["kplanes_config"])
    base.update(cfg)
    base.update(overrides)
    return Namespace(**base)


class SynthModel(torch.nn.Module):
    """Minimal stand-in for the reference's GaussianModel: exactly the attributes `render()` reads
    (scene/gaussian_model.py:36-44,108-131): parameters, activations, get_xyz / get_features, the deformation module."""

    def __init__(self, n, deform_cfg="dynerf_default", seed=6666, sh_degree=3, device="cpu", perturb_time_planes=0.1,
                 deformation=None, scene="cube"):
        """`perturb_time_planes` (name kept from SURVEY 8d, which perturbs the time planes only): 0.1 * N(0, 1) is added to EVERY plane of the
        HexPlane field -- spatial planes too -- so that no factor of the six-plane product is constant in any coordinate (the reference's
        initialisation leaves the time planes at exactly one, scene/hexplane.py:64-67, and the deformation would not depend on t at all).
        Every measured number of this repository is on this generator; changing it would change the benchmark scene."""
        super().__init__()
        from . import deformation as D
        g = make_gaussians(n, seed=seed, sh_degree=sh_degree, scene=scene)
        self._xyz = torch.nn.Parameter(g["xyz"])
        self._scaling = torch.nn.Parameter(g["scaling"])
        self._rotation = torch.nn.Parameter(g["rotation"])
        self._opacity = torch.nn.Parameter(g["opacity"])
        self._features_dc = torch.nn.Parameter(g["features_dc"])
        self._features_rest = torch.nn.Parameter(g["features_rest"])
        self.max_sh_degree = sh_degree
        self.active_sh_degree = sh_degree
        torch.manual_seed(seed)
        if deformation is None:
            self._deformation = D.deform_network(deform_args(deform_cfg))
            gen = torch.Generator().manual_seed(seed + 1)
            with torch.no_grad():
                for name, p in self._deformation.named_parameters():
                    if "grids" in name and perturb_time_planes:
                        p.add_(perturb_time_planes * torch.randn(p.shape, generator=gen))
            self._deformation.deformation_net.set_aabb(g["xyz"].max(0).values.tolist(), g["xyz"].min(0).values.tolist())
        else:
            self._deformation = deformation
        self._deformation_table = torch.ones(n, dtype=torch.bool)
        self.scaling_activation = torch.exp
        self.rotation_activation = torch.nn.functional.normalize
        self.opacity_activation = torch.sigmoid
        self.to(device)

    def optimizer_groups(self, lr=0.0):
        """The eight parameter groups of GaussianModel.training_setup (scene/gaussian_model.py:165-183), same names/order."""
        d = self._deformation
        return [{"params": [self._xyz], "lr": lr, "name": "xyz"},
                {"params": list(d.get_mlp_parameters()), "lr": lr, "name": "deformation"},
                {"params": list(d.get_grid_parameters()), "lr": lr, "name": "grid"},
                {"params": [self._features_dc], "lr": lr, "name": "f_dc"},
                {"params": [self._features_rest], "lr": lr, "name": "f_rest"},
                {"params": [self._opacity], "lr": lr, "name": "opacity"},
                {"params": [self._scaling], "lr": lr, "name": "scaling"},
                {"params": [self._rotation], "lr": lr, "name": "rotation"}]

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier=1):
        s = scaling_modifier * torch.exp(self._scaling)
        q = self._rotation / self._rotation.norm(dim=1, keepdim=True)
        r, x, y, z = q.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z),
                         1 - 2 * (x * x + z * z), 2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x),
                         1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
        Lm = R * s[:, None, :]
        S = Lm @ Lm.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)


class PipelineParams:
    """arguments/__init__.py:66-73 defaults."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
