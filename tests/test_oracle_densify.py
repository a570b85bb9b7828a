"""CPU: the densification oracle (oracle/densify_oracle.py) against the golden vector produced by the reference's own
GaussianModel methods (tests/golden/make_densify_golden.py) and, where /root/reference is present, against those methods
run live on CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import densify_oracle as DO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify_small.npz")


def load_state(g, prefix):
    st = {"param": {}, "exp_avg": {}, "exp_avg_sq": {}}
    for k in g.files:
        if not k.startswith(prefix + "."):
            continue
        parts = k.split(".")
        t = torch.from_numpy(g[k])
        if len(parts) == 3:
            st[parts[1]][parts[2]] = t
        else:
            st[parts[1]] = t
    return st


def assert_states_equal(a, b, tol=1e-6):
    for k in ("param", "exp_avg", "exp_avg_sq"):
        for n in DO.GROUPS:
            assert a[k][n].shape == b[k][n].shape, (k, n, a[k][n].shape, b[k][n].shape)
            if a[k][n].numel():
                assert (a[k][n] - b[k][n]).abs().max().item() <= tol, (k, n)
    for k in ("xyz_gradient_accum", "denom", "max_radii2D", "deformation_accum"):
        assert a[k].shape == b[k].shape and (a[k].numel() == 0 or (a[k] - b[k]).abs().max().item() <= tol), k
    assert torch.equal(a["deformation_table"].bool(), b["deformation_table"].bool())


def test_oracle_reproduces_the_reference_sequence_of_the_golden_vector():
    g = np.load(GOLD)
    _, _, pd, extent, max_grad = [float(v) for v in g["meta"]]
    st = load_state(g, "in")
    DO.add_densification_stats(st, torch.from_numpy(g["vgrad"]), torch.from_numpy(g["vis"]), torch.from_numpy(g["radii"]))
    assert_states_equal(st, load_state(g, "stats"))
    nc, ns = DO.densify(st, max_grad, extent, pd, torch.from_numpy(g["normals"]))
    assert nc > 20 and ns > 20                                   # the vector exercises both passes
    assert_states_equal(st, load_state(g, "densified"))
    st["max_radii2D"] = torch.from_numpy(g["radii_after"]).clone()
    assert DO.prune(st, 0.05, extent, 20) > 20
    assert_states_equal(st, load_state(g, "pruned"))
    DO.reset_opacity(st)
    assert_states_equal(st, load_state(g, "reset"))


@pytest.mark.skipif(not os.path.exists("/root/reference/scene/gaussian_model.py"), reason="reference tree not present")
@pytest.mark.parametrize("n,seed,size", [(300, 3, 20), (64, 4, None), (1, 5, 20)])
def test_oracle_matches_reference_methods_live(n, seed, size):
    st = DO.random_state(n, seed, sh_rest=3)
    normals = torch.randn(2 * n, 3, generator=torch.Generator().manual_seed(seed))
    m = DO.reference_model_from_state(st, 0.01)
    o = DO.clone_state(st)
    with DO.reference_on_cpu(normals):
        m.densify(0.0002, 0.005, 3.0, size, 5, 5)
        DO.densify(o, 0.0002, 3.0, 0.01, normals)
        assert_states_equal(o, DO.state_from_reference_model(m))
        r = torch.rand(m.get_xyz.shape[0], generator=torch.Generator().manual_seed(seed + 1)) * 40
        m.max_radii2D = r.clone()
        o["max_radii2D"] = r.clone()
        m.prune(0.0002, 0.05, 3.0, size)
        DO.prune(o, 0.05, 3.0, size)
        assert_states_equal(o, DO.state_from_reference_model(m))
