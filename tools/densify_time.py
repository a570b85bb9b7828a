"""Times the densification kernels at 300 k Gaussians next to the boolean-indexing formulation of the per-iteration
statistics (train.py:261-262 / scene/gaussian_model.py:516-518):  python tools/densify_time.py"""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdgs = importlib.import_module("4dgaussians_amd")
dn = importlib.import_module("4dgaussians_amd.densify")
syn = importlib.import_module("4dgaussians_amd.synthetic")

N = 300000
g = torch.Generator().manual_seed(0)


class M:
    pass


def model():
    m = M()
    m.percent_dense = 0.01
    shapes = {"_xyz": (3,), "_features_dc": (1, 3), "_features_rest": (15, 3), "_opacity": (1,), "_scaling": (3,), "_rotation": (4,)}
    names = {"_xyz": "xyz", "_features_dc": "f_dc", "_features_rest": "f_rest", "_opacity": "opacity", "_scaling": "scaling", "_rotation": "rotation"}
    groups = []
    for a, s in shapes.items():
        t = torch.randn(N, *s, generator=g)
        if a == "_scaling":
            t = torch.log(torch.rand(N, 3, generator=g) * 0.04 + 1e-3)
        p = torch.nn.Parameter(t.cuda())
        setattr(m, a, p)
        groups.append({"params": [p], "lr": 0.0, "name": names[a]})
    m.optimizer = fdgs.FusedAdam(groups, lr=0.0, eps=1e-15)
    for gr in groups:
        p = gr["params"][0]
        m.optimizer.state[p] = {"step": torch.tensor(1.0), "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)}
    m.denom = torch.randint(0, 4, (N, 1), generator=g).float().cuda()
    m.xyz_gradient_accum = (torch.rand(N, 1, generator=g) * 0.0006).cuda() * m.denom
    m.max_radii2D = (torch.rand(N, generator=g) * 40).cuda()
    m._deformation_accum = torch.zeros(N, 3, device="cuda")
    m._deformation_table = torch.ones(N, dtype=torch.bool, device="cuda")
    return m


m = model()
vg = (torch.randn(N, 3, generator=g) * 1e-3).cuda()
radii = torch.randint(0, 60, (N,), generator=g, dtype=torch.int32).cuda()
vis = radii > 0


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def stats_indexing():
    m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis].float())
    m.xyz_gradient_accum[vis] += torch.norm(vg[vis, :2], dim=-1, keepdim=True)
    m.denom[vis] += 1


print("per-iteration statistics: boolean indexing %.3f ms, fdgs_densification_stats %.3f ms"
      % (timed(stats_indexing), timed(lambda: dn.add_densification_stats(m, vg, vis, radii))))
torch.cuda.synchronize()
t0 = time.perf_counter()
res = dn.densify(m, 0.0002, 0.005, 3.0, 20)
torch.cuda.synchronize()
t1 = time.perf_counter()
res2 = dn.prune(m, 0.0002, 0.05, 3.0, 20)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("densify %d -> kept %d clones %d splits %d: %.3f ms wall;  prune -> %d rows: %.3f ms wall" % (N, *res, (t1 - t0) * 1e3, res2[0], (t2 - t1) * 1e3))
m = model()
torch.cuda.synchronize()
t0 = time.perf_counter()
res = dn.densify(m, 0.0002, 0.005, 3.0, 20)
torch.cuda.synchronize()
print("densify (second model, warm): %.3f ms wall" % ((time.perf_counter() - t0) * 1e3))
