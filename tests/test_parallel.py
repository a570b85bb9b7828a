"""World-size-2 gloo test (CPU) of the frame-parallel driver: frame assignment, the single loss-statistics all-reduce
(numerators/denominators, so global L1/PSNR equal the single-process values on the concatenated batch), max-over-ranks
timing and barrier."""
import importlib
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = importlib.import_module("4dgaussians_amd.parallel")
    r, w, dev = par.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dev.type == "cpu"
    g = torch.Generator().manual_seed(123)
    imgs = torch.rand(4, 3, 8, 8, generator=g)
    tgts = torch.rand(4, 3, 8, 8, generator=g)
    frames = []
    total = torch.zeros(3)
    for step in range(2):
        i = par.frames_for_rank(4, step, rank, world)
        frames.append(i)
        d = imgs[i] - tgts[i]
        acc = torch.tensor([d.abs().sum(), (d * d).sum(), float(d.numel())])
        if step == 0:
            par.allreduce_loss_stats(acc)
        else:                                   # the overlapped form bench.py uses: a handle, waited for before the buffer is read
            h = par.allreduce_loss_stats(acc, async_op=True)
            assert h is not None
            h.wait()
        total += acc
    par.barrier()
    t = par.max_over_ranks(0.5 + rank, dev)
    l1, psnr = par.loss_from_stats(total)
    q.put((rank, frames, float(l1), float(psnr), t))
    torch.distributed.destroy_process_group()


def test_two_rank_frame_parallel_loss_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]          # every frame rendered exactly once
    g = torch.Generator().manual_seed(123)
    imgs = torch.rand(4, 3, 8, 8, generator=g); tgts = torch.rand(4, 3, 8, 8, generator=g)
    d = imgs - tgts
    l1 = float(d.abs().mean()); mse = float((d * d).mean())
    psnr = 20 * torch.log10(1.0 / torch.sqrt(torch.tensor(mse))).item()
    for r in res:
        assert r[2] == pytest.approx(l1, rel=1e-5) and r[3] == pytest.approx(psnr, rel=1e-5)
        assert r[4] == 1.5                                          # max over ranks


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = importlib.import_module("4dgaussians_amd.parallel")
    par.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(7)
    shapes = [(50, 3), (50, 16, 3), (7,), (4, 4), (1,)]
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    views = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(world)]     # same on every rank
    for p, gr in zip(params, views[rank]):
        p.grad = gr.clone()
    params.append(torch.nn.Parameter(torch.zeros(3)))                                  # no grad anywhere: skipped
    calls_small = par.allreduce_gradients(params, bucket_bytes=1024)                  # forces several buckets
    small = [p.grad.clone() for p in params[:-1]]
    for p, gr in zip(params, views[rank]):
        p.grad = gr.clone()
    calls_big = par.allreduce_gradients(params, average=True)
    radii = torch.tensor([1, 5, 0, 2]) if rank == 0 else torch.tensor([3, 0, 0, 9])
    vis = radii > 0
    r, v, vg = par.allreduce_densification_stats(radii, vis, torch.full((4, 3), float(rank + 1)))
    q.put((rank, calls_small, calls_big, [t.tolist() for t in small], [p.grad.tolist() for p in params[:-1]], r.tolist(), v.tolist(), vg.tolist()))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_and_densification_allreduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(7)
    shapes = [(50, 3), (50, 16, 3), (7,), (4, 4), (1,)]
    views = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(world)]
    want = [views[0][i] + views[1][i] for i in range(len(shapes))]
    for r in res:
        assert r[1] > 1 and r[2] == 1                                # bucketing: several small buckets vs one large
        for got, w in zip(r[3], want):
            assert torch.allclose(torch.tensor(got), w, atol=1e-6)
        for got, w in zip(r[4], want):
            assert torch.allclose(torch.tensor(got), w / 2, atol=1e-6)
        assert r[5] == [3, 5, 0, 9] and r[6] == [True, True, False, True]
        assert torch.allclose(torch.tensor(r[7]), torch.full((4, 3), 3.0))


def _arena_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = importlib.import_module("4dgaussians_amd.parallel")
    par.init_from_env(backend="gloo")
    shapes = [(40, 3), (40, 4), (1, 8, 6, 5), (7,), (16, 16)]

    def make(seed):
        # gradients as the fused render path leaves them: views of ONE arena at 64-element aligned offsets, planes channels-last
        g = torch.Generator().manual_seed(seed)
        sizes = [(int(torch.Size(s_).numel()) + 63) // 64 * 64 for s_ in shapes]
        arena = torch.full((sum(sizes),), float("nan"))            # (gaps stay garbage: nobody may read them)
        params, off = [], 0
        for s_, n_ in zip(shapes, sizes):
            numel = int(torch.Size(s_).numel())
            v = arena[off:off + numel]
            if len(s_) == 4:
                v = v.view(s_[0], s_[2], s_[3], s_[1]).permute(0, 3, 1, 2)     # logical [1,C,H,W], channels-last memory
            else:
                v = v.view(s_)
            v.copy_(torch.randn(*s_, generator=g))
            p_ = torch.nn.Parameter(torch.zeros(*s_))
            if len(s_) == 4:
                p_.data = p_.data.contiguous(memory_format=torch.channels_last)
            p_.grad = v
            params.append(p_)
            off += n_
        return arena, params

    arena, params = make(100 + rank)
    params.append(torch.nn.Parameter(torch.zeros(5)))          # no gradient on ANY rank (an unused sub-network): does not stop the in-place path
    ptr0 = [None if p_.grad is None else p_.grad.data_ptr() for p_ in params]
    calls = par.allreduce_gradients(params)
    assert params[-1].grad is None
    params = params[:-1]; ptr0 = ptr0[:-1]
    same_storage = [p_.grad.data_ptr() for p_ in params] == ptr0
    got = [p_.grad.contiguous().clone() for p_ in params]
    # same data through the packing path (zero-copy off) must agree
    arena2, params2 = make(100 + rank)
    calls2 = par.allreduce_gradients(params2, arena_zero_copy=False)
    agree = all(torch.allclose(a.contiguous(), b.grad.contiguous()) for a, b in zip(got, params2))
    # one rank lost a gradient: the signature differs, every rank falls back to packing together (no hang, right sums)
    arena3, params3 = make(100 + rank)
    if rank == 1:
        params3[3].grad = None
    calls3 = par.allreduce_gradients(params3)
    # a SUBSET of the arena's gradients (the caller reduces the others separately, or not at all): the tensor in the middle must not be
    # touched by this call -- the zero-copy span would have summed (and, with average=True, divided) it along with its neighbours
    arena4, params4 = make(100 + rank)
    middle_before = params4[1].grad.clone()
    par.allreduce_gradients([params4[0]] + params4[2:], average=True)
    middle_untouched = bool(torch.equal(params4[1].grad, middle_before))
    par.allreduce_gradients([params4[1]], average=True)
    q.put((rank, calls, same_storage, calls2, agree, calls3, [g_.tolist() for g_ in got], params3[3].grad.tolist(), params3[0].grad.tolist(),
           middle_untouched, params4[1].grad.tolist(), params4[2].grad.contiguous().tolist()))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_allreduce_runs_in_place_on_the_gradient_arena():
    """When every rank holds its gradients as views of one arena with the same layout (what the fused render backward returns), the
    arena is the all-reduce buffer: one collective, no packing copy, gradients stay where they are; a rank with a different layout
    sends everybody down the packing path."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_arena_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shapes = [(40, 3), (40, 4), (1, 8, 6, 5), (7,), (16, 16)]
    want = None
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        vals = [torch.randn(*s_, generator=g) for s_ in shapes]
        want = vals if want is None else [a + b for a, b in zip(want, vals)]
    g1 = torch.Generator().manual_seed(100)
    first = [torch.randn(*s_, generator=g1) for s_ in shapes]
    for r in res:
        assert r[1] == 1 and r[2] is True and r[3] == 1 and r[4] is True and r[5] == 1
        for got, w in zip(r[6], want):
            assert torch.allclose(torch.tensor(got), w, atol=1e-6)
        assert torch.allclose(torch.tensor(r[7]), first[3], atol=1e-6)       # only rank 0 had this one: sum = rank 0's values
        assert torch.allclose(torch.tensor(r[8]), want[0], atol=1e-6)
        assert r[9] is True                                                   # subset call left the tensor in the middle alone ...
        assert torch.allclose(torch.tensor(r[10]), want[1] / 2, atol=1e-6)    # ... which was then averaged exactly once
        assert torch.allclose(torch.tensor(r[11]), want[2] / 2, atol=1e-6)


def test_single_process_is_a_noop():
    par = importlib.import_module("4dgaussians_amd.parallel")
    acc = torch.tensor([1.0, 2.0, 3.0])
    assert torch.equal(par.allreduce_loss_stats(acc.clone()), acc)
    assert par.allreduce_loss_stats(acc.clone(), async_op=True) is None
    assert par.frames_for_rank(160, 5, 0, 1) == 5
    p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.ones(3)
    assert par.allreduce_gradients([p]) == 0 and torch.equal(p.grad, torch.ones(3))


# ---- launcher-free multi-rank start (bench.py --gpus N without torch.distributed.run) ----
def _spawned_job(path, backend):
    """Runs in every rank started by parallel.spawn_local: the same entry sequence bench.py's run() uses."""
    par = importlib.import_module("4dgaussians_amd.parallel")
    rank, world, dev = par.init_from_env(backend=backend)
    seen = par.ranks_seen(dev)
    times = par.gather_floats(10.0 + rank, dev)
    # one rank lacks a gradient the other has: the has-grad mask makes the buckets identical on both ranks
    a, b = torch.nn.Parameter(torch.zeros(5, device=dev)), torch.nn.Parameter(torch.zeros(3, device=dev))
    a.grad = torch.full((5,), float(rank + 1), device=dev)
    if rank == 1:
        b.grad = torch.full((3,), 7.0, device=dev)
    calls = par.allreduce_gradients([a, b])
    par.barrier()
    with open(f"{path}.{rank}", "w") as f:
        f.write(repr((rank, world, seen, times, calls, a.grad.tolist(), b.grad.tolist(), str(dev))))


def test_spawn_local_starts_ranks_and_rccl_style_handshake_on_gloo(tmp_path):
    par = importlib.import_module("4dgaussians_amd.parallel")
    path = str(tmp_path / "out")
    par.spawn_local(2, _spawned_job, (path, "gloo"))
    for r in range(2):
        rank, world, seen, times, calls, ga, gb, dev = eval(open(f"{path}.{r}").read())
        assert (rank, world, seen) == (r, 2, 2) and times == [10.0, 11.0] and calls == 1
        assert ga == [3.0] * 5 and gb == [7.0] * 3 and dev == "cpu"


def _failing_job():
    raise SystemExit(3)


def test_spawn_local_reports_failed_ranks():
    par = importlib.import_module("4dgaussians_amd.parallel")
    with pytest.raises(RuntimeError, match="exited non-zero"):
        par.spawn_local(2, _failing_job)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` with WORLD_SIZE unset and fewer than N visible GPUs must stop with a clear message before
    touching a device (on this CPU container 0 are visible; on a 1-GPU box the -m gpu twin below asks for 2)."""
    import subprocess
    import sys
    n_vis = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n_vis + 2)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and f"only {n_vis} GPU(s) visible" in r.stderr


@pytest.mark.gpu
def test_bench_refuses_two_gpus_on_a_one_gpu_box():
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    test_bench_refuses_more_gpus_than_visible()


@pytest.mark.gpu
def test_two_gpu_ranks_meet_over_rccl(tmp_path):
    """Real RCCL: two ranks, one GPU each, through the same spawn path bench.py --gpus 2 takes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    par = importlib.import_module("4dgaussians_amd.parallel")
    path = str(tmp_path / "out")
    par.spawn_local(2, _spawned_job, (path, "nccl"))
    for r in range(2):
        rank, world, seen, times, calls, ga, gb, dev = eval(open(f"{path}.{r}").read())
        assert (rank, world, seen) == (r, 2, 2) and ga == [3.0] * 5 and gb == [7.0] * 3 and dev == f"cuda:{r}"


# ---- bench.py's rank logic end to end with a stub step (no GPU): camera assignment over steps x world, timed regions reduced with
# MAX over ranks, ranks_seen, per-rank figures, rank-0-only JSON line ----
def _bench_stub_rank(path, steps, warmup, repeats):
    import importlib.util as ilu
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = ilu.spec_from_file_location("bench_stub_mod", os.path.join(root, "bench.py"))
    bench = ilu.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import argparse
    import contextlib
    import io
    args = argparse.Namespace(gpus=2, steps=steps, warmup=warmup, repeats=repeats)

    def make_step(ctx):
        def step(i, cam):
            time.sleep(0.002 * (1 + 2 * ctx["rank"]))       # rank 1 is three times slower: MAX over ranks must report it
        step.cpu_leg = lambda: ({"value": 0.3, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "stub"}, {"image_psnr_dB": 100.0, "grad_ok": True})
        return step
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = bench.run(args, make_step=make_step)
    with open(f"{path}.{out['cameras_rank_local'][0] % 2}", "w") as f:
        f.write(repr((out, buf.getvalue())))


def test_bench_rank_logic_with_a_stub_step(tmp_path):
    par = importlib.import_module("4dgaussians_amd.parallel")
    steps, warmup, repeats = 6, 2, 3
    path = str(tmp_path / "b")
    par.spawn_local(2, _bench_stub_rank, (path, steps, warmup, repeats))
    res = [eval(open(f"{path}.{r}").read()) for r in range(2)]
    import importlib.util as ilu
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = ilu.spec_from_file_location("bench_plan_mod", os.path.join(root, "bench.py"))
    bench = ilu.module_from_spec(spec)
    spec.loader.exec_module(bench)
    plan = bench.rank_plan(160, steps, warmup, repeats, 2)
    for r, (out, printed) in enumerate(res):
        assert out["cameras_rank_local"] == plan[r]                       # rank r renders camera (step * world + r) mod n
        assert out["ranks_seen"] == 2 and out["n_gpus"] == 2 and out["timed_regions"] == repeats
        assert len(out["per_rank_ms_per_step"]) == 2 and len(out["startup_s_per_rank"]) == 2
        # both ranks report the same (MAX over ranks) time, and it is the slow rank's: >= 6 ms per step
        assert out["ms_per_step"] >= 5.5 and out["per_rank_ms_per_step"][1] > 2.0 * out["per_rank_ms_per_step"][0]
        assert abs(out["value"] - 2 * steps / (out["ms_per_step"] * steps / 1e3)) < 1e-6     # whole-job frames/s over all ranks
        if r == 0:
            line = json.loads(printed.strip())                            # exactly one JSON line, on rank 0 only
            assert line["n_gpus"] == 2 and line["steps"] == steps and line["warmup"] == warmup
            # an N > 1 line verifies itself: CPU baseline + oracle parity (rank 0's leg, any world size), who ran where, what each rank did
            assert line["cpu_baseline"]["value"] == 0.3 and line["parity"]["grad_ok"] is True
            rk = line["ranks"]
            assert [x["rank"] for x in rk["per_rank"]] == [0, 1] and rk["distinct_devices"] == 2 and rk["allreduce_us_blocking"] > 0
            assert [x["num_rendered"] for x in rk["per_rank"]] == [1000, 1001]
            assert all(x["p90_ms_per_step"] >= x["median_ms_per_step"] > 0 and "device" in x for x in rk["per_rank"])
            assert rk["per_rank"][1]["median_ms_per_step"] > 2.0 * rk["per_rank"][0]["median_ms_per_step"]
        else:
            assert printed.strip() == ""
    assert res[0][0]["ms_per_step"] == res[1][0]["ms_per_step"]
    every = sorted(res[0][0]["cameras_rank_local"] + res[1][0]["cameras_rank_local"])
    assert every == list(range(2 * warmup, 2 * warmup + 2 * steps * repeats))     # each frame of the orbit rendered exactly once


def _one_rank_dies_the_other_waits():
    par = importlib.import_module("4dgaussians_amd.parallel")
    rank, world, dev = par.init_from_env(backend="gloo")
    if rank == 1:
        raise SystemExit(5)
    par.barrier()           # rank 0 would sit here until the process-group timeout


def test_spawn_local_terminates_siblings_of_a_dead_rank():
    import time
    par = importlib.import_module("4dgaussians_amd.parallel")
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="exited non-zero"):
        par.spawn_local(2, _one_rank_dies_the_other_waits)
    assert time.monotonic() - t0 < 60.0        # (not the process-group timeout of minutes)


# ---- eight ranks (the node the driver's --gpus 8 run uses), on gloo: every host-side piece of the N > 1 path at the world size it will meet ----
def _dp_worker_n(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    par = importlib.import_module("4dgaussians_amd.parallel")
    r_, w_, dev = par.init_from_env(backend="gloo")
    assert (r_, w_) == (rank, world) and par.ranks_seen(dev) == world
    g = torch.Generator().manual_seed(7)
    shapes = [(50, 3), (50, 16, 3), (7,), (4, 4), (1,)]
    params = [torch.nn.Parameter(torch.zeros(*s)) for s in shapes]
    views = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(world)]     # same on every rank
    for p, gr in zip(params, views[rank]):
        p.grad = gr.clone()
    if rank == 5:
        params[2].grad = None                                                          # one rank's views never touched this tensor
    params.append(torch.nn.Parameter(torch.zeros(3)))                                  # no grad anywhere: skipped everywhere
    calls = par.allreduce_gradients(params, bucket_bytes=4096)
    radii = torch.tensor([(rank * 3 + i * 5) % 11 for i in range(6)])
    r, v, vg = par.allreduce_densification_stats(radii, radii > 4, torch.full((6, 3), float(rank + 1)))
    acc = torch.tensor([float(rank), float(rank * rank), 10.0])
    h = par.allreduce_loss_stats(acc, async_op=True)
    h.wait()
    # the densification plan check: agreeing ranks pass, one rank with another split count makes EVERY rank raise (nobody is left in a collective)
    par.check_same_plan([100, 7, 3, 1], dev)
    try:
        par.check_same_plan([100, 7, 3 + (1 if rank == 6 else 0), 1], dev)
        raised = False
    except RuntimeError as e:
        raised = "disagree" in str(e)
    q.put((rank, calls, [p.grad.tolist() for p in params[:-1]], params[-1].grad is None, r.tolist(), v.tolist(), vg.tolist(), acc.tolist(), raised,
           par.max_over_ranks(float(rank), dev), par.gather_floats(rank * 0.5, dev)))
    torch.distributed.destroy_process_group()


def test_eight_rank_gradient_statistics_and_plan_check_on_gloo():
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker_n, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(7)
    shapes = [(50, 3), (50, 16, 3), (7,), (4, 4), (1,)]
    views = [[torch.randn(*s, generator=g) for s in shapes] for _ in range(world)]
    want = [sum(views[r][i] for r in range(world) if not (r == 5 and i == 2)) for i in range(len(shapes))]
    radii = [[(r * 3 + i * 5) % 11 for i in range(6)] for r in range(world)]
    for rank, calls, grads, skipped, rr, vv, vg, acc, raised, mx, gf in res:
        assert calls > 1 and skipped and raised
        for got, w in zip(grads, want):
            assert torch.allclose(torch.tensor(got), w, atol=1e-5)
        assert rr == [max(radii[r][i] for r in range(world)) for i in range(6)]
        assert vv == [any(radii[r][i] > 4 for r in range(world)) for i in range(6)]
        assert torch.allclose(torch.tensor(vg), torch.full((6, 3), float(sum(range(1, world + 1)))))
        assert acc == [float(sum(range(world))), float(sum(r * r for r in range(world))), 10.0 * world]
        assert mx == float(world - 1) and gf == [0.5 * r for r in range(world)]


def _bench_stub_rank_n(path, steps, warmup, repeats, world):
    import importlib.util as ilu
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = ilu.spec_from_file_location("bench_stub_mod", os.path.join(root, "bench.py"))
    bench = ilu.module_from_spec(spec)
    spec.loader.exec_module(bench)
    import argparse
    import contextlib
    import io
    args = argparse.Namespace(gpus=world, steps=steps, warmup=warmup, repeats=repeats)

    def make_step(ctx):
        def step(i, cam):
            time.sleep(0.001 * (1 + (ctx["rank"] == world - 1)))       # the last rank is twice as slow
        step.cpu_leg = lambda: ({"value": 0.3, "unit": "frames/s", "cores": 1, "kind": "port", "sample": "stub"}, {"image_psnr_dB": 100.0, "grad_ok": True})
        return step
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = bench.run(args, make_step=make_step)
    with open(f"{path}.{os.environ['RANK']}", "w") as f:
        f.write(repr((out, buf.getvalue())))


def test_bench_rank_logic_at_world_size_eight(tmp_path):
    """bench.run's skeleton (camera plan, barrier-bracketed regions reduced with MAX over ranks, rank-0 leg, per-rank report, one JSON line)
    with EIGHT ranks on gloo -- the world size the driver's scaling run launches, which no test had reached."""
    par = importlib.import_module("4dgaussians_amd.parallel")
    world, steps, warmup, repeats = 8, 4, 1, 2
    path = str(tmp_path / "b8")
    par.spawn_local(world, _bench_stub_rank_n, (path, steps, warmup, repeats, world))
    res = [eval(open(f"{path}.{r}").read()) for r in range(world)]
    import importlib.util as ilu
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = ilu.spec_from_file_location("bench_plan_mod8", os.path.join(root, "bench.py"))
    bench = ilu.module_from_spec(spec)
    spec.loader.exec_module(bench)
    plan = bench.rank_plan(160, steps, warmup, repeats, world)
    for r, (out, printed) in enumerate(res):
        assert out["cameras_rank_local"] == plan[r] and out["ranks_seen"] == world and out["n_gpus"] == world
        assert len(out["per_rank_ms_per_step"]) == world
        assert abs(out["value"] - world * steps / (out["ms_per_step"] * steps / 1e3)) < 1e-6      # whole-job frames/s
        if r == 0:
            line = json.loads(printed.strip())
            rk = line["ranks"]
            assert [x["rank"] for x in rk["per_rank"]] == list(range(world)) and rk["distinct_devices"] == world
            assert line["cpu_baseline"]["value"] == 0.3 and line["parity"]["grad_ok"] is True
        else:
            assert printed.strip() == ""
    assert len({res[r][0]["ms_per_step"] for r in range(world)}) == 1                           # everybody reports the MAX over ranks
    every = sorted(c for r in range(world) for c in res[r][0]["cameras_rank_local"])
    first = world * warmup
    assert every == [c % 160 for c in range(first, first + world * steps * repeats)]              # every frame of the orbit exactly once


def test_host_core_assignment_of_ranks():
    par = importlib.import_module("4dgaussians_amd.parallel")
    assert par.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = range(192)
    # no NUMA information: even slices, disjoint, covering
    shares = [par.cpus_for_rank(r, 8, allowed) for r in range(8)]
    assert all(len(s) == 24 for s in shares) and sorted(c for s in shares for c in s) == list(range(192))
    # two NUMA nodes, four GPUs each: every rank inside its node, the four ranks of a node disjoint
    numa = [list(range(0, 96))] * 4 + [list(range(96, 192))] * 4
    shares = [par.cpus_for_rank(r, 8, allowed, numa) for r in range(8)]
    assert all(set(s) <= set(numa[r]) and len(s) == 24 for r, s in enumerate(shares))
    assert sorted(c for s in shares for c in s) == list(range(192))
    # cores outside the process's allowed set are never chosen; a rank never ends up with nothing
    assert par.cpus_for_rank(0, 8, [5], numa) == [5]
    assert par.cpus_for_rank(7, 8, range(4)) != []
