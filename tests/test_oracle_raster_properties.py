"""Property tests of the rasterizer oracle with hypothesis: the size-independent invariants any
correct implementation of the reference's rasterizer must satisfy, on randomly drawn small scenes.  The same properties
are asserted of the HIP path in tests/test_gpu_raster.py."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle.raster_oracle import RasterOracle
from scenes import raster_scene

SCENE = st.fixed_dictionaries(dict(n=st.integers(1, 60), width=st.integers(17, 70), height=st.integers(17, 50), seed=st.integers(0, 10_000),
                                   theta=st.floats(-180, 180), sh_degree=st.integers(0, 3), scale_boost=st.floats(0.5, 12.0)))
FEW = settings(max_examples=20, deadline=None)


def _o(sc):
    return RasterOracle(**sc, dtype=np.float64)


@given(SCENE, st.integers(0, 1000))
@FEW
def test_permutation_invariance_up_to_depth_ties(case, pseed):
    sc = raster_scene(**case, dtype=np.float64)
    a = _o(sc)
    perm = np.random.default_rng(pseed).permutation(case["n"])
    sp = dict(sc)
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        sp[k] = np.ascontiguousarray(sc[k][perm])
    b = _o(sp)
    np.testing.assert_array_equal(a.radii[perm], b.radii)
    np.testing.assert_allclose(a.color, b.color, rtol=0, atol=1e-12)     # distinct depths almost surely: same blend order
    np.testing.assert_allclose(a.depth, b.depth, rtol=0, atol=1e-12)


@given(SCENE)
@FEW
def test_zero_opacity_gaussians_contribute_nothing(case):
    sc = raster_scene(**case, dtype=np.float64)
    base = _o(sc)
    k = max(1, case["n"] // 3)
    s2 = {key: (np.concatenate([sc[key], sc[key][:k]]) if key in ("means3D", "shs", "scales", "rotations") else sc[key]) for key in sc}
    s2["means3D"] = s2["means3D"].copy()
    s2["means3D"][-k:] *= 0.93
    s2["opacities"] = np.concatenate([sc["opacities"], np.zeros((k, 1))])
    o2 = _o(s2)
    np.testing.assert_allclose(o2.color, base.color, rtol=0, atol=1e-12)
    g = o2.backward(np.ones_like(o2.color))
    assert np.all(g["means3D"][-k:] == 0) and np.all(g["scales"][-k:] == 0)   # and receive no geometric gradient


@given(SCENE)
@FEW
def test_radii_tiles_rect_consistency_and_pair_list(case):
    sc = raster_scene(**case, dtype=np.float64)
    o = _o(sc)
    tiles, rect = o.field("tiles_touched"), o.field("rect").astype(np.int64)
    assert np.all((o.radii == 0) <= (tiles == 0))                     # culled => no tiles
    area = (rect[:, 2] - rect[:, 0]) * (rect[:, 3] - rect[:, 1])
    assert np.array_equal(area[o.radii > 0], tiles[o.radii > 0])
    assert o.num_rendered == int(tiles.sum())
    keys_hi, gid = o.pairs()            # (tile ids of the sorted list, Gaussian ids)
    if o.num_rendered:
        assert np.all(np.diff(keys_hi.astype(np.int64)) >= 0)        # sorted by tile
        dep = o.field("depth")[gid]
        same = np.diff(keys_hi.astype(np.int64)) == 0
        assert np.all(np.diff(dep)[same] >= 0)                       # then by depth
        tie = same & (np.diff(dep) == 0)
        assert np.all(np.diff(gid.astype(np.int64))[tie] > 0)        # stable on equal keys
        assert np.array_equal(np.bincount(gid, minlength=case["n"]), tiles)


@given(SCENE)
@FEW
def test_n_contrib_bounded_by_tile_list_and_final_T_in_range(case):
    sc = raster_scene(**case, dtype=np.float64)
    o = _o(sc)
    fT, nc = o.image_state()
    assert np.all((fT > 0) & (fT <= 1.0))
    W, H = case["width"], case["height"]
    gx = (W + 15) // 16
    keys_hi, _ = o.pairs()
    per_tile = np.bincount(keys_hi, minlength=gx * ((H + 15) // 16)) if o.num_rendered else np.zeros(gx * ((H + 15) // 16), int)
    ty, tx = np.meshgrid(np.arange(H) // 16, np.arange(W) // 16, indexing="ij")
    assert np.all(nc <= per_tile[ty * gx + tx])
    # colour = blended + T * bg, with bg = 1: every channel >= final_T (blended part is >= 0 after the SH clamp)
    assert np.all(o.color >= fT[None] - 1e-12)


@given(SCENE)
@FEW
def test_all_culled_gives_background_and_zero_gradients(case):
    sc = raster_scene(**case, dtype=np.float64)
    sc["means3D"] = sc["means3D"] * 0 + np.array([60.0, 60.0, 60.0])
    o = _o(sc)
    assert np.all(o.radii == 0) and np.all(o.color == 1.0) and np.all(o.depth == 0.0)
    g = o.backward(np.ones_like(o.color))
    assert all(np.all(v == 0) for k, v in g.items() if v is not None)
