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
        par.allreduce_loss_stats(acc)
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


def test_single_process_is_a_noop():
    par = importlib.import_module("4dgaussians_amd.parallel")
    acc = torch.tensor([1.0, 2.0, 3.0])
    assert torch.equal(par.allreduce_loss_stats(acc.clone()), acc)
    assert par.frames_for_rank(160, 5, 0, 1) == 5
    p = torch.nn.Parameter(torch.zeros(3)); p.grad = torch.ones(3)
    assert par.allreduce_gradients([p]) == 0 and torch.equal(p.grad, torch.ones(3))
