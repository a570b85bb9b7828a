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
            elif name == "fdgs_raster_fwd_capacity":      # (the device would deliver the count later; the fake delivers it at once)
                ctypes.cast(a[6], ctypes.POINTER(ctypes.c_uint32))[0] = 7
            elif name == "fdgs_pair_count_wait":
                a[2].value = ctypes.cast(a[1], ctypes.POINTER(ctypes.c_uint32))[0]
            elif name == "fdgs_deform_bwd_live_tiles":
                for i in range(4):
                    a[3][i] = 1
            elif name == "fdgs_raster_bwd":
                g = a[6]
                self.acc_seen = getattr(self, "acc_seen", []) + [(int(g.scratch_acc or 0), int(g.scratch_acc_zeroed))]
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
    monkeypatch.setattr(fdgs.rasterizer, "_pinned_words", lambda n: torch.zeros(n, dtype=torch.int32))
    monkeypatch.setattr(fdgs.rasterizer, "_current_stream", lambda dev: type("S", (), {"synchronize": lambda self: None})())
    monkeypatch.setattr(fdgs.rasterizer, "_tls", __import__("threading").local())
    monkeypatch.setattr(fdgs.rasterizer, "_seen", {})
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


def test_backward_accumulator_is_the_one_the_forward_zero_filled_exactly_once(fake):
    """A training frame allocates the blending backward's [P,16] accumulator at forward time and has fdgs_render_fwd zero-fill it on the way
    (fdgs_raster_params::acc_zero): the first backward of that state hands it over with scratch_acc_zeroed = 1 (no fill launch), a second
    backward of the same graph gets a fresh buffer that fdgs_raster_bwd must fill itself; a no-grad frame allocates none."""
    R = fdgs.rasterizer
    pc = syn.SynthModel(300, "dynerf_default", seed=1)
    cam = syn.make_camera(64, 48, theta_deg=10.0, time=0.3)
    res = fdgs.render(cam, pc, _Pipe(), torch.zeros(3), stage="fine")
    res["render"].sum().backward(retain_graph=True)
    res["render"].sum().backward()
    (a0, z0), (a1, z1) = fake.acc_seen
    assert a0 != 0 and a1 != 0 and z0 == 1 and z1 == 0
    st = R.RasterState()
    st.acc, st.params, st.geom = None, type("P", (), {"P": 5})(), torch.zeros(1)
    buf, zeroed = st.take_accumulator()
    assert buf.shape == (5, 16) and zeroed == 0
    with torch.no_grad():
        fdgs.render(cam, pc, _Pipe(), torch.zeros(3), stage="fine")
    assert len(fake.acc_seen) == 2


def test_render_views_plumbing(fake):
    pc = syn.SynthModel(200, "dnerf_bouncingballs", seed=2)
    cams = [syn.make_camera(64, 48, theta_deg=10.0 * i, time=0.1 * i) for i in range(3)]
    res = fdgs.render_views(cams, pc, _Pipe(), torch.zeros(3), stage="fine")
    assert len(res) == 3
    sum(r["render"].sum() for r in res).backward()
    assert all(r["viewspace_points"].grad is not None for r in res) and pc._xyz.grad is not None


def test_gradient_arena_layout_tiles_the_arena_without_overlap():
    """deformation._arena_layout: every view starts 256-byte aligned, views do not overlap, the planes' strides are channels-last and the
    per-Gaussian head ends where the zero-filled part begins; _collect caches per module and notices a replaced parameter."""
    d = fdgs.deformation
    planes = ((1, 16, 64, 64), (1, 16, 25, 64), (1, 16, 64, 25))
    mlps = ((128, 32), (128,), (3, 128), (3,))
    for combined in (True, False):
        total, head, n_fixed, specs = d._arena_layout(1000, combined, planes, mlps)
        assert n_fixed == (5 if combined else 6) and len(specs) == n_fixed + len(planes) + len(mlps)
        arena = torch.zeros(total)
        spans = []
        for i, (shape, strides, off) in enumerate(specs):
            assert off % 64 == 0
            v = arena.as_strided(shape, strides, off)
            n = v.numel()
            spans.append((off, off + n))
            v.fill_(float(i + 1))
            if n_fixed <= i < n_fixed + len(planes):
                assert v.is_contiguous(memory_format=torch.channels_last) and tuple(v.shape) == planes[i - n_fixed]
            else:
                assert v.is_contiguous()
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= total
        assert head == specs[n_fixed][2]
        for i, (shape, strides, off) in enumerate(specs):      # nobody wrote into anybody else's slice
            assert bool((arena.as_strided(shape, strides, off) == float(i + 1)).all())
    net = syn.SynthModel(50, "dynerf_default", seed=3)._deformation
    a = d._collect(net)
    b = d._collect(net)
    assert a[0] is b[0] and a[1] is b[1]
    seq = getattr(net.deformation_net, d.HEAD_NAMES[-1])
    seq[3].bias = torch.nn.Parameter(seq[3].bias.detach().clone())
    c = d._collect(net)
    assert c[1] is not a[1] and c[1][-1] is seq[3].bias
    # an INTERIOR parameter, a whole head and a grid level replaced: each is noticed (every cached slot is re-validated on every call)
    dn = net.deformation_net
    lin = getattr(dn, d.HEAD_NAMES[1])[1]
    lin.weight = torch.nn.Parameter(lin.weight.detach().clone())
    e = d._collect(net)
    assert e[1] is not c[1] and any(t is lin.weight for t in e[1])
    import copy
    setattr(dn, d.HEAD_NAMES[2], copy.deepcopy(getattr(dn, d.HEAD_NAMES[2])))
    f = d._collect(net)
    assert f[1] is not e[1] and any(t is getattr(dn, d.HEAD_NAMES[2])[3].weight for t in f[1])
    dn.grid.grids[0] = copy.deepcopy(dn.grid.grids[0])
    g = d._collect(net)
    assert g[0] is not f[0] and g[0][0] is dn.grid.grids[0][0]
    assert d._collect(net)[0] is g[0]                      # ... and an untouched module hits the cache
    # the cache lives on the module: nothing global keeps a deleted model's Parameters alive
    import gc
    import weakref
    w = weakref.ref(net.deformation_net.grid.grids[0][0])
    del net, a, b, c, e, f, g, dn, lin, seq
    gc.collect()
    assert w() is None


def test_host_time_per_frame_stays_within_budget(fake):
    """Python / ctypes / autograd time of one render() forward + backward against the fake library (no kernels, no device): what the host must
    spend per frame before any launch cost.  Measured 0.78 ms on the build container (0.15 ms forward only, where render() calls the two stages
    directly instead of through autograd.Function.apply; ~0.3 ms of the 0.78 is the autograd engine
    handing 46 gradients to their AccumulateGrad nodes): a frame whose GPU work is shorter than this is host-paced (BASELINE config 2 sits at
    0.8 ms of GPU work).  The budget is 2x the measurement -- loose enough for a loaded CI host, tight enough to catch a per-frame module walk,
    a per-tensor conversion pass or a re-introduced host synchronisation."""
    import time
    fake.calls = type("Sink", (), {"append": lambda self, x: None})()
    pc = syn.SynthModel(300, "dynerf_default", seed=1)
    cam = syn.make_camera(64, 48, theta_deg=10.0, time=0.3)
    bg, dimg = torch.zeros(3), torch.ones(3, 48, 64)
    prm = [p for p in pc.parameters() if p.requires_grad]

    def step():
        for p in prm:
            p.grad = None
        fdgs.render(cam, pc, _Pipe(), bg, stage="fine")["render"].backward(dimg)

    for _ in range(30):
        step()
    best = float("inf")
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        best = min(best, (time.perf_counter() - t0) / 100 * 1e3)
    print(f"host time per render() forward + backward: {best:.3f} ms")
    assert best < 1.7, best
