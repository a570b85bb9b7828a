"""Development aid: where does the HOST time of one fine-stage train iteration go (render fwd+bwd + densification statistics + regulariser
+ FusedAdam)?  Prints the wall time per iteration, the time the host needs to ENQUEUE an iteration (no synchronisation inside the loop: if
this is close to the wall time the loop is host-bound), and a cProfile of the iteration.  usage (GPU box): python tools/host_profile_train.py"""
import cProfile, importlib, io, os, pstats, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
fdgs = importlib.import_module("4dgaussians_amd")
syn = fdgs.synthetic
dev = torch.device("cuda:0")
L = fdgs._lib.lib()
N, W, H, dcfg = bench.WORKLOADS["cfg4_dynerf_300k_1352x1014"]
pc = syn.SynthModel(N, dcfg, seed=6666, device=dev)
fdgs.densify.spatial_reorder(pc)
pipe, bg = syn.PipelineParams(), torch.zeros(3, device=dev)
cams = [c.to(dev) for c in syn.orbit_cameras(W, H, n=160)]
target = torch.rand(3, H, W, generator=torch.Generator().manual_seed(6666)).to(dev)
acc, dimg = torch.zeros(3, device=dev), torch.empty(3, H, W, device=dev)
st, ptr = fdgs._lib.stream_ptr, fdgs._lib.ptr
opt = fdgs.FusedAdam(pc.optimizer_groups(lr=0.0), lr=0.0, eps=1e-15)
params = [p for p in pc.parameters() if p.requires_grad]
dstat = types.SimpleNamespace(xyz_gradient_accum=torch.zeros(N, 1, device=dev), denom=torch.zeros(N, 1, device=dev), max_radii2D=torch.zeros(N, device=dev))


def render_step(i):
    for p_ in params:
        p_.grad = None
    res = fdgs.render(cams[i % len(cams)], pc, pipe, bg, stage="fine")
    img = res["render"]
    acc.zero_()
    fdgs._lib.check(L.fdgs_l1_stats(st(), img.numel(), ptr(img), ptr(target), 1.0 / img.numel(), ptr(dimg), ptr(acc)))
    img.backward(dimg)
    return res


def train_iter(i):
    res = render_step(i)
    fdgs.densify.add_densification_stats(dstat, res["viewspace_points"].grad, res["visibility_filter"], res["radii"])
    reg = fdgs.compute_regulation(pc, 0.01, 0.0001, 0.0001)
    reg.backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for name, fn in (("render fwd+bwd", render_step), ("train iteration", train_iter)):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    K = 40
    t0 = time.perf_counter()
    for i in range(K):
        fn(i)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name}: wall {t_all / K * 1e3:.3f} ms, host enqueue {t_enq / K * 1e3:.3f} ms per iteration (includes the one blocking num_rendered read-back per frame)")
pr = cProfile.Profile()
pr.enable()
for i in range(40):
    train_iter(i)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:9000])
