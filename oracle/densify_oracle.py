"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch) of the reference's densification bookkeeping.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

A Gaussian set is a plain dict `st`:
    st["param"][name], st["exp_avg"][name], st["exp_avg_sq"][name]   name in GROUPS (rows per Gaussian)
    st["xyz_gradient_accum"] [N,1], st["denom"] [N,1], st["max_radii2D"] [N], st["deformation_accum"] [N,3],
    st["deformation_table"] [N] bool
and every function below follows ONE method of the reference, step by step, in the reference's own order (clone pass,
then split pass on the enlarged set with zero-padded gradients, then removal of the split originals), so that it checks
the fused single-pass kernel's row order independently:
    add_densification_stats   scene/gaussian_model.py:516-518 (+ the max_radii2D update of train.py:261)
    _cat / _prune             cat_tensors_to_optimizer :367-389, _prune_optimizer :331-348, densification_postfix :391-407,
                              prune_points :350-365
    densify_and_clone         :440-456         densify_and_split  :409-438 (N = 2, children = R(q/|q|)(n*exp(s)) + xyz,
                              scale' = log(exp(s)/(0.8 N)))        densify :495-500 (NaN gradients -> 0)
    prune                     :481-494         reset_opacity      :269-272
Pinned by tests/test_oracle_densify.py against the reference's own GaussianModel methods run on CPU (imported from
/root/reference where present, with `device="cuda"` redirected and torch.normal fed the recorded samples) and against the
committed golden vector tests/golden/densify_small.npz produced from those methods.
"""
import torch

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


def clone_state(st):
    out = {}
    for k, v in st.items():
        out[k] = {n: t.clone() for n, t in v.items()} if isinstance(v, dict) else v.clone()
    return out


def add_densification_stats(st, viewspace_grad, update_filter, radii=None):
    f = update_filter
    if radii is not None:
        st["max_radii2D"][f] = torch.max(st["max_radii2D"][f], radii[f].to(st["max_radii2D"].dtype))
    st["xyz_gradient_accum"][f] += torch.norm(viewspace_grad[f, :2], dim=-1, keepdim=True)
    st["denom"][f] += 1


def _cat(st, new, new_table):
    for n in GROUPS:
        ext = new[n]
        st["param"][n] = torch.cat((st["param"][n], ext), 0)
        st["exp_avg"][n] = torch.cat((st["exp_avg"][n], torch.zeros_like(ext)), 0)
        st["exp_avg_sq"][n] = torch.cat((st["exp_avg_sq"][n], torch.zeros_like(ext)), 0)
    n_all = st["param"]["xyz"].shape[0]
    st["deformation_table"] = torch.cat((st["deformation_table"], new_table), -1)
    st["xyz_gradient_accum"] = torch.zeros(n_all, 1)
    st["deformation_accum"] = torch.zeros(n_all, 3)
    st["denom"] = torch.zeros(n_all, 1)
    st["max_radii2D"] = torch.zeros(n_all)


def prune_points(st, mask):
    keep = ~mask
    for n in GROUPS:
        st["param"][n] = st["param"][n][keep]
        st["exp_avg"][n] = st["exp_avg"][n][keep]
        st["exp_avg_sq"][n] = st["exp_avg_sq"][n][keep]
    for k in ("deformation_accum", "xyz_gradient_accum", "deformation_table", "denom", "max_radii2D"):
        st[k] = st[k][keep]


def rotation_matrix(q):
    q = q / torch.sqrt((q * q).sum(1))[:, None]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


def densify_and_clone(st, grads, grad_threshold, scene_extent, percent_dense):
    P = st["param"]
    sel = (torch.norm(grads, dim=-1) >= grad_threshold) & (torch.exp(P["scaling"]).max(1).values <= percent_dense * scene_extent)
    _cat(st, {n: P[n][sel] for n in GROUPS}, st["deformation_table"][sel])
    return int(sel.sum())


def densify_and_split(st, grads, grad_threshold, scene_extent, percent_dense, normals, N=2):
    P = st["param"]
    n0 = P["xyz"].shape[0]
    padded = torch.zeros(n0)
    padded[:grads.shape[0]] = grads.squeeze(-1)
    sel = (padded >= grad_threshold) & (torch.exp(P["scaling"]).max(1).values > percent_dense * scene_extent)
    ns = int(sel.sum())
    if ns == 0:
        return 0
    stds = torch.exp(P["scaling"][sel]).repeat(N, 1)
    samples = normals[:N * ns] * stds                                     # torch.normal(mean=0, std=stds)
    rots = rotation_matrix(P["rotation"][sel]).repeat(N, 1, 1)
    new = {
        "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + P["xyz"][sel].repeat(N, 1),
        "scaling": torch.log(torch.exp(P["scaling"][sel]).repeat(N, 1) / (0.8 * N)),
        "rotation": P["rotation"][sel].repeat(N, 1),
        "f_dc": P["f_dc"][sel].repeat(N, 1, 1),
        "f_rest": P["f_rest"][sel].repeat(N, 1, 1),
        "opacity": P["opacity"][sel].repeat(N, 1),
    }
    _cat(st, new, st["deformation_table"][sel].repeat(N))
    prune_points(st, torch.cat((sel, torch.zeros(N * ns, dtype=torch.bool))))
    return ns


def densify(st, max_grad, extent, percent_dense, normals):
    """normals: [>= 2*splits, 3] standard-normal samples, consumed in torch.normal's row order."""
    grads = st["xyz_gradient_accum"] / st["denom"]
    grads[grads.isnan()] = 0.0
    nc = densify_and_clone(st, grads, max_grad, extent, percent_dense)
    ns = densify_and_split(st, grads, max_grad, extent, percent_dense, normals)
    return nc, ns


def prune(st, min_opacity, extent, max_screen_size):
    P = st["param"]
    mask = (torch.sigmoid(P["opacity"]) < min_opacity).squeeze(-1)
    if max_screen_size:
        mask = mask | (st["max_radii2D"] > max_screen_size) | (torch.exp(P["scaling"]).max(1).values > 0.1 * extent)
    prune_points(st, mask)
    return int(mask.sum())


def reset_opacity(st):
    op = torch.sigmoid(st["param"]["opacity"])
    x = torch.min(op, torch.ones_like(op) * 0.01)
    st["param"]["opacity"] = torch.log(x / (1 - x))
    st["exp_avg"]["opacity"] = torch.zeros_like(st["param"]["opacity"])
    st["exp_avg_sq"]["opacity"] = torch.zeros_like(st["param"]["opacity"])


def random_state(n, seed, sh_rest=15, extent=3.0, percent_dense=0.01):
    """A seeded Gaussian set whose statistics put a good share of the Gaussians in each class (kept / clone / split)."""
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    P = {"xyz": rn(n, 3), "f_dc": rn(n, 1, 3), "f_rest": rn(n, sh_rest, 3), "opacity": rn(n, 1) * 3,
         "scaling": torch.log(torch.rand(n, 3, generator=g) * 1.3 * percent_dense * extent + 1e-3), "rotation": rn(n, 4)}
    st = {"param": P, "exp_avg": {k: rn(*v.shape) * 0.1 for k, v in P.items()},
          "exp_avg_sq": {k: torch.rand(*v.shape, generator=g) * 0.01 for k, v in P.items()}}
    denom = torch.randint(0, 4, (n, 1), generator=g).float()
    st["denom"] = denom
    st["xyz_gradient_accum"] = torch.rand(n, 1, generator=g) * 0.0006 * denom
    st["max_radii2D"] = torch.rand(n, generator=g) * 40
    st["deformation_accum"] = rn(n, 3)
    st["deformation_table"] = torch.rand(n, generator=g) > 0.3
    return st


# ---- the reference's own GaussianModel, run on CPU (only where /root/reference exists) ----
def import_reference_gaussian_model():
    import sys
    import types
    from . import deform_oracle
    deform_oracle.import_reference_deform_network()          # sys.path + the `scene` package shell + tkinter stub
    for name, attrs in (("open3d", {}), ("plyfile", {"PlyData": object, "PlyElement": object}), ("simple_knn", {}),
                        ("simple_knn._C", {"distCUDA2": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    from scene.gaussian_model import GaussianModel
    return GaussianModel


class reference_on_cpu:
    """Context manager: the reference's methods allocate with device="cuda" and draw torch.normal samples; redirect the
    former to the CPU and feed the latter from `normals` (std * n, the definition of normal(0, std)), then restore."""

    def __init__(self, normals=None):
        self.normals = normals

    def __enter__(self):
        import utils.general_utils as gu
        self._zeros, self._normal, self._gu, self._gu_zeros = torch.zeros, torch.normal, gu, None
        zeros = self._zeros

        def cpu_zeros(*a, **k):
            if k.get("device") == "cuda":
                k["device"] = "cpu"
            return zeros(*a, **k)

        def fed_normal(mean=None, std=None, **k):
            return mean + self.normals[:std.shape[0]] * std

        torch.zeros, torch.normal = cpu_zeros, fed_normal
        return self

    def __exit__(self, *exc):
        torch.zeros, torch.normal = self._zeros, self._normal
        return False


def reference_model_from_state(st, percent_dense):
    """A reference GaussianModel (constructed without __init__: no deformation network needed here) holding `st`, with a
    torch.optim.Adam whose moments are the state's -- exactly what training_setup + some steps would leave behind."""
    GM = import_reference_gaussian_model()
    m = GM.__new__(GM)
    m.setup_functions()
    m.percent_dense = percent_dense
    names = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling",
             "rotation": "_rotation"}
    groups = []
    for n in GROUPS:
        p = torch.nn.Parameter(st["param"][n].clone().requires_grad_(True))
        setattr(m, names[n], p)
        groups.append({"params": [p], "lr": 0.0, "name": n})
    m.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    for gr in m.optimizer.param_groups:
        p = gr["params"][0]
        m.optimizer.state[p] = {"step": torch.tensor(7.0), "exp_avg": st["exp_avg"][gr["name"]].clone(),
                                "exp_avg_sq": st["exp_avg_sq"][gr["name"]].clone()}
    m.xyz_gradient_accum = st["xyz_gradient_accum"].clone()
    m.denom = st["denom"].clone()
    m.max_radii2D = st["max_radii2D"].clone()
    m._deformation_accum = st["deformation_accum"].clone()
    m._deformation_table = st["deformation_table"].clone()
    return m


def state_from_reference_model(m):
    st = {"param": {}, "exp_avg": {}, "exp_avg_sq": {}}
    for gr in m.optimizer.param_groups:
        p = gr["params"][0]
        st["param"][gr["name"]] = p.detach().clone()
        s = m.optimizer.state[p]
        st["exp_avg"][gr["name"]] = s["exp_avg"].clone()
        st["exp_avg_sq"][gr["name"]] = s["exp_avg_sq"].clone()
    st["xyz_gradient_accum"] = m.xyz_gradient_accum.clone()
    st["denom"] = m.denom.clone()
    st["max_radii2D"] = m.max_radii2D.clone()
    st["deformation_accum"] = m._deformation_accum.clone()
    st["deformation_table"] = m._deformation_table.clone()
    return st
