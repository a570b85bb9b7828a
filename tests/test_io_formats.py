"""Host-side format tests (CPU): the PLY point-cloud file and the deformation checkpoint are byte / key compatible with what
the reference writes (scene/gaussian_model.py:214-312).  `plyfile` is not installed here, so the byte layout is pinned to
the PLY specification as plyfile emits it for an all-float32 vertex element (header text checked literally) and to files
in the other encodings a foreign writer may produce (ascii, big-endian, doubles, extra properties and elements)."""
import importlib
import os
import types

import numpy as np
import pytest
import torch

from oracle import deform_oracle as DO

io = importlib.import_module("4dgaussians_amd.io")
fdgs = importlib.import_module("4dgaussians_amd")
syn = importlib.import_module("4dgaussians_amd.synthetic")


def model(n=37, deg=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    k = (deg + 1) ** 2
    m = types.SimpleNamespace(max_sh_degree=deg, active_sh_degree=0)
    m._xyz, m._features_dc, m._features_rest = torch.randn(n, 3, generator=g), torch.randn(n, 1, 3, generator=g), torch.randn(n, k - 1, 3, generator=g)
    m._opacity, m._scaling, m._rotation = torch.randn(n, 1, generator=g), torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g)
    return m


@pytest.mark.parametrize("n,deg", [(37, 3), (1, 0), (0, 3), (5, 1)])
def test_ply_bytes_and_round_trip(tmp_path, n, deg):
    m = model(n, deg)
    p = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    io.save_ply(m, p)
    raw = open(p, "rb").read()
    k = (deg + 1) ** 2
    names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * (k - 1))] + \
        ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {s}\n" for s in names) + "end_header\n"
    assert raw.startswith(header.encode()) and len(raw) == len(header) + n * len(names) * 4
    body = np.frombuffer(raw[len(header):], dtype="<f4").reshape(n, len(names))
    assert np.array_equal(body[:, 0:3], m._xyz.numpy()) and not body[:, 3:6].any()
    # SH coefficients are stored channel-major ([N,3,K] flattened): f_rest_j = channel j // (K-1), coefficient j % (K-1)
    if k > 1 and n:
        assert np.array_equal(body[:, 9:9 + 3 * (k - 1)].reshape(n, 3, k - 1), m._features_rest.permute(0, 2, 1).numpy())
    m2 = model(3, deg, seed=5)
    io.load_ply(m2, p, device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        t = getattr(m2, a)
        assert isinstance(t, torch.nn.Parameter) and t.requires_grad and t.is_contiguous()
        assert t.shape == getattr(m, a).shape and torch.equal(t.detach(), getattr(m, a))
    assert m2.active_sh_degree == deg


def test_reader_accepts_foreign_encodings_and_rejects_bad_files(tmp_path):
    m = model(4, 0)
    cols = {"x": [1, 2, 3, 4], "y": [5, 6, 7, 8], "z": [0, 0, 1, 1], "f_dc_0": [.1, .2, .3, .4], "f_dc_1": [0, 0, 0, 0], "f_dc_2": [1, 1, 1, 1],
            "opacity": [-1, 0, 1, 2], "scale_0": [0, 0, 0, 0], "scale_1": [1, 1, 1, 1], "scale_2": [2, 2, 2, 2],
            "rot_0": [1, 1, 1, 1], "rot_1": [0, 0, 0, 0], "rot_2": [0, 0, 0, 0], "rot_3": [0, 0, 0, 0], "red": [9, 9, 9, 9]}
    # ascii, shuffled property order, one extra uchar property, a face element afterwards
    order = ["red", "rot_3", "rot_2", "rot_1", "rot_0", "scale_2", "scale_1", "scale_0", "opacity", "f_dc_2", "f_dc_1", "f_dc_0", "z", "y", "x"]
    txt = "ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 4\n" + "".join(
        f"property {'uchar' if n == 'red' else 'float'} {n}\n" for n in order) + "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
    txt += "".join(" ".join(str(cols[n][i]) for n in order) + "\n" for i in range(4)) + "3 0 1 2\n"
    p = str(tmp_path / "a.ply")
    open(p, "w").write(txt)
    io.load_ply(m, p, device="cpu")
    assert m._xyz.tolist() == [[1, 5, 0], [2, 6, 0], [3, 7, 1], [4, 8, 1]] and m._scaling[0].tolist() == [0, 1, 2]
    assert torch.allclose(m._features_dc[:, 0, 0], torch.tensor([.1, .2, .3, .4])) and m._opacity[:, 0].tolist() == [-1, 0, 1, 2]
    # big-endian doubles
    names = [n for n in order if n != "red"]
    hdr = "ply\nformat binary_big_endian 1.0\nelement vertex 4\n" + "".join(f"property double {n}\n" for n in names) + "end_header\n"
    rec = np.zeros(4, dtype=[(n, ">f8") for n in names])
    for n in names:
        rec[n] = cols[n]
    p2 = str(tmp_path / "b.ply")
    open(p2, "wb").write(hdr.encode() + rec.tobytes())
    m3 = model(1, 0)
    io.load_ply(m3, p2, device="cpu")
    assert torch.equal(m3._xyz, m._xyz.detach()) and torch.equal(m3._rotation, m._rotation.detach())
    # errors: wrong SH degree for the file, truncated payload, not a PLY
    with pytest.raises(ValueError):
        io.load_ply(model(1, 3), p, device="cpu")
    open(p2, "wb").write(hdr.encode() + rec.tobytes()[:-8])
    with pytest.raises(ValueError):
        io.read_ply_vertices(p2)
    open(p2, "wb").write(b"solid not a ply\n")
    with pytest.raises(ValueError):
        io.read_ply_vertices(p2)


def test_deformation_checkpoint_round_trip_and_key_compatibility(tmp_path):
    args = syn.deform_args("dynerf_default")
    torch.manual_seed(0)
    net = fdgs.deform_network(args)
    pc = types.SimpleNamespace(_deformation=net, _xyz=torch.zeros(9, 3), _deformation_table=torch.rand(9) > 0.5,
                               _deformation_accum=torch.randn(9, 3))
    io.save_deformation(pc, str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["deformation.pth", "deformation_accum.pth", "deformation_table.pth"]
    torch.manual_seed(1)
    pc2 = types.SimpleNamespace(_deformation=fdgs.deform_network(args), _xyz=torch.zeros(9, 3))
    io.load_model(pc2, str(tmp_path), device="cpu")
    for (k1, v1), (k2, v2) in zip(net.state_dict().items(), pc2._deformation.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert torch.equal(pc2._deformation_table, pc._deformation_table) and torch.equal(pc2._deformation_accum, pc._deformation_accum)
    assert pc2.max_radii2D.shape == (9,)
    # without the two optional files the reference falls back to all-true / zeros
    os.remove(tmp_path / "deformation_table.pth"); os.remove(tmp_path / "deformation_accum.pth")
    io.load_model(pc2, str(tmp_path), device="cpu")
    assert pc2._deformation_table.all() and not pc2._deformation_accum.any()


@pytest.mark.skipif(not os.path.exists("/root/reference/scene/deformation.py"), reason="reference tree not present")
@pytest.mark.parametrize("cfg", ["dnerf_bouncingballs", "hypernerf_default", "dynerf_default"])
def test_deformation_pth_interchanges_with_the_reference_module(tmp_path, cfg):
    """deformation.pth written from our module loads into the reference's deform_network (strict) and vice versa."""
    args = syn.deform_args(cfg)
    ours, ref = fdgs.deform_network(args), DO.import_reference_deform_network()(args)
    torch.save(ours.state_dict(), tmp_path / "ours.pth")
    torch.save(ref.state_dict(), tmp_path / "ref.pth")
    ref.load_state_dict(torch.load(tmp_path / "ours.pth"), strict=True)
    ours.load_state_dict(torch.load(tmp_path / "ref.pth"), strict=True)
    assert list(ours.state_dict().keys()) == list(ref.state_dict().keys())
