"""On-disk formats either side of the path, byte-compatible with the reference so that point clouds and deformation
weights move between the two implementations:

    GaussianModel.save_ply / load_ply            scene/gaussian_model.py:214-227, 254-312 (binary little-endian PLY, one
                                                 `vertex` element of float properties x y z nx ny nz f_dc_* f_rest_* opacity
                                                 scale_* rot_*; SH coefficients stored channel-major: [N,3,K] flattened)
    GaussianModel.save_deformation / load_model  :232-253 (`deformation.pth` = deform_network.state_dict(),
                                                 `deformation_table.pth`, `deformation_accum.pth` via torch.save)

The reference goes through the `plyfile` package; this module reads and writes the same bytes with numpy only (plyfile's
header for an all-'f4' structured array is exactly the one written here).  The reader accepts what plyfile-written and
other common PLY files contain: ascii or binary (either endianness) scalar properties of any PLY scalar type, in any
order, extra elements after `vertex` ignored.  Host-side code: no GPU work happens here.
"""
import os

import numpy as np
import torch
from torch import nn

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def construct_list_of_attributes(pc):
    """scene/gaussian_model.py:214-227."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(pc._features_dc.shape[1] * pc._features_dc.shape[2])]
    names += [f"f_rest_{i}" for i in range(pc._features_rest.shape[1] * pc._features_rest.shape[2])]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(pc._scaling.shape[1])]
    names += [f"rot_{i}" for i in range(pc._rotation.shape[1])]
    return names


def write_ply_vertices(path, names, table):
    """table: float32 [N, len(names)].  Header and payload as plyfile writes them for PlyElement.describe(..., 'vertex')."""
    table = np.ascontiguousarray(table, dtype="<f4")
    if table.ndim != 2 or table.shape[1] != len(names):
        raise ValueError("table must be [N, len(names)]")
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {n}" for n in names]
    header.append("end_header")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(table.tobytes())


def read_ply_vertices(path):
    """-> dict name -> numpy array [N] for the `vertex` element (must be the first element, as plyfile writes it)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex, seen_vertex = None, None, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", errors="replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if in_vertex:
                    in_vertex = False                       # a later element: its data is never reached
                elif not seen_vertex:
                    if tok[1] != "vertex":
                        raise ValueError(f"{path}: first element is '{tok[1]}', expected 'vertex'")
                    count, in_vertex, seen_vertex = int(tok[2]), True, True
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in the vertex element are not supported")
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown property type '{tok[1]}'")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt is None or count is None:
            raise ValueError(f"{path}: incomplete PLY header")
        if fmt == "ascii":
            rows = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            if rows.shape != (count, len(props)):
                raise ValueError(f"{path}: expected {count} x {len(props)} values")
            return {n: rows[:, i].astype(t) for i, (n, t) in enumerate(props)}
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unknown PLY format '{fmt}'")
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        raw = f.read(dt.itemsize * count)
        if len(raw) != dt.itemsize * count:
            raise ValueError(f"{path}: truncated vertex data")
        rec = np.frombuffer(raw, dtype=dt, count=count)
        return {n: rec[n].astype(rec[n].dtype.newbyteorder("=")) for n, _ in props}      # native byte order, contiguous


def save_ply(pc, path):
    """GaussianModel.save_ply (scene/gaussian_model.py:254-270)."""
    xyz = pc._xyz.detach().cpu().numpy()
    f_dc = pc._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = pc._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    table = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, pc._opacity.detach().cpu().numpy(),
                            pc._scaling.detach().cpu().numpy(), pc._rotation.detach().cpu().numpy()), axis=1)
    write_ply_vertices(path, construct_list_of_attributes(pc), table)


def _numbered(v, prefix):
    names = sorted((n for n in v if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    return np.stack([np.asarray(v[n], dtype=np.float64) for n in names], axis=1) if names else np.zeros((len(v["x"]), 0))


def load_ply(pc, path, device=None):
    """GaussianModel.load_ply (scene/gaussian_model.py:274-312): replaces the six Parameters of `pc`, sets
    active_sh_degree = max_sh_degree.  `device` defaults to the GPU, as in the reference."""
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    v = read_ply_vertices(path)
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float64)
    n = xyz.shape[0]
    f_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1).astype(np.float64)[..., None]               # [N,3,1]
    rest = _numbered(v, "f_rest_")
    k = (pc.max_sh_degree + 1) ** 2
    if rest.shape[1] != 3 * k - 3:
        raise ValueError(f"{path}: {rest.shape[1]} f_rest_* properties, expected {3 * k - 3} for SH degree {pc.max_sh_degree}")
    rest = rest.reshape(n, 3, k - 1)
    par = lambda a: nn.Parameter(torch.tensor(a, dtype=torch.float, device=device).requires_grad_(True))
    pc._xyz = par(xyz)
    pc._features_dc = nn.Parameter(torch.tensor(f_dc, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
    pc._features_rest = nn.Parameter(torch.tensor(rest, dtype=torch.float, device=device).transpose(1, 2).contiguous().requires_grad_(True))
    pc._opacity = par(np.asarray(v["opacity"], dtype=np.float64)[..., None])
    pc._scaling = par(_numbered(v, "scale_"))
    pc._rotation = par(_numbered(v, "rot"))
    pc.active_sh_degree = pc.max_sh_degree


def save_deformation(pc, path):
    """GaussianModel.save_deformation (scene/gaussian_model.py:250-253)."""
    os.makedirs(path, exist_ok=True)
    torch.save(pc._deformation.state_dict(), os.path.join(path, "deformation.pth"))
    torch.save(pc._deformation_table, os.path.join(path, "deformation_table.pth"))
    torch.save(pc._deformation_accum, os.path.join(path, "deformation_accum.pth"))


def load_model(pc, path, device=None):
    """GaussianModel.load_model (scene/gaussian_model.py:232-249)."""
    device = device or ("cuda" if torch.cuda.is_available() else "cpu")
    pc._deformation.load_state_dict(torch.load(os.path.join(path, "deformation.pth"), map_location=device))
    pc._deformation = pc._deformation.to(device)
    n = pc._xyz.shape[0]
    pc._deformation_table = torch.ones(n, dtype=torch.bool, device=device)
    pc._deformation_accum = torch.zeros(n, 3, device=device)
    for name, attr in (("deformation_table.pth", "_deformation_table"), ("deformation_accum.pth", "_deformation_accum")):
        f = os.path.join(path, name)
        if os.path.exists(f):
            setattr(pc, attr, torch.load(f, map_location=device))
    pc.max_radii2D = torch.zeros(n, device=device)
