"""Host plumbing dry run (CPU, no kernels): fdgs.render / render_views forward + backward and the two-node path executed on CPU tensors
against a FAKE library object whose entry points only answer the size queries and the one read-back.  Nothing is computed -- the test
exists so that a Python-level slip (a missing slot, a wrong argument count, a None where a tensor is expected) is caught here and not on
the GPU box.  The product has no CPU path: the fake is installed by monkeypatch in this test only."""
import ctypes
import importlib

import pytest
import torch

fdgs = importlib.import_module("4dgaussians_amd")
syn = fdgs.synthetic


class _FakeLib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        if not name.startswith("fdgs_"):
            raise AttributeError(name)

        def fn(*a):
            self.calls.append(name)
            if name in ("fdgs_geom_bytes", "fdgs_img_bytes", "fdgs_binning_bytes", "fdgs_deform_saved_bytes", "fdgs_deform_bwd_scratch_bytes"):
                a[-1].value = 4096
            elif name == "fdgs_bin_prepare":
                ctypes.cast(a[3], ctypes.POINTER(ctypes.c_uint32))[0] = 7
            elif name == "fdgs_deform_bwd_live_tiles":
                for i in range(4):
                    a[3][i] = 1
            elif name == "fdgs_last_error":
                return b""
            elif name == "fdgs_abi_version":
                return 3
            return 0
        return fn


@pytest.fixture
def fake(monkeypatch):
    f = _FakeLib()
    monkeypatch.setattr(fdgs._lib, "lib", lambda: f)
    monkeypatch.setattr(fdgs._lib, "stream_ptr", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(fdgs.rasterizer, "stream_ptr", lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(fdgs.deformation, "stream_ptr", lambda: ctypes.c_void_p(0), raising=False)
    monkeypatch.setattr(fdgs.rasterizer, "_pinned_u32", lambda dev: torch.zeros(1, dtype=torch.int32))
    # rasterize_forward / forward_impl refuse non-HIP tensors: the dry run claims to be one
    monkeypatch.setattr(fdgs.rasterizer, "_is_hip_device", lambda dev: True)
    monkeypatch.setattr(fdgs.deformation, "_is_hip_device", lambda dev: True)
    return f


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


@pytest.mark.parametrize("fused", [True, False])
def test_render_forward_backward_plumbing(fake, fused, monkeypatch):
    monkeypatch.setattr(fdgs.renderer, "FUSED_BACKWARD", fused)
    pc = syn.SynthModel(300, "dynerf_default", seed=1)
    cam = syn.make_camera(64, 48, theta_deg=10.0, time=0.3)
    res = fdgs.render(cam, pc, _Pipe(), torch.zeros(3), stage="fine")
    assert set(res) == {"render", "viewspace_points", "visibility_filter", "radii", "depth"}
    assert res["visibility_filter"].dtype == torch.bool and res["visibility_filter"].shape == (300,)
    res["render"].sum().backward()
    assert pc._xyz.grad is not None and res["viewspace_points"].grad is not None
    assert "fdgs_raster_bwd" in fake.calls and "fdgs_deform_bwd" in fake.calls
    # coarse stage and a no-grad forward
    with torch.no_grad():
        fdgs.render(cam, pc, _Pipe(), torch.zeros(3), stage="coarse")


def test_render_views_plumbing(fake):
    pc = syn.SynthModel(200, "dnerf_bouncingballs", seed=2)
    cams = [syn.make_camera(64, 48, theta_deg=10.0 * i, time=0.1 * i) for i in range(3)]
    res = fdgs.render_views(cams, pc, _Pipe(), torch.zeros(3), stage="fine")
    assert len(res) == 3
    sum(r["render"].sum() for r in res).backward()
    assert all(r["viewspace_points"].grad is not None for r in res) and pc._xyz.grad is not None
