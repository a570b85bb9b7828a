"""Frame-parallel multi-GPU driver (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The render path shards over independent frames/views (the reference already loops over `batch_size` independent
views per step and sums their losses, train.py:180-201): every rank holds the full (replicated) Gaussian and
deformation parameters, renders its own cameras, and the only data-path exchange is ONE tiny all-reduce per step of
the loss statistics [sum|err|, sum err^2, count] (numerators/denominators, not per-rank means, so the global L1 / PSNR
equal what the reference computes on the concatenated batch, train.py:197-203).  No tile- or Gaussian-sharding: a
frame is ~ms of work and 300k Gaussians are ~100 MB against 288 GB of HBM.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, world, device).
    `timeout_s` (default FDGS_PG_TIMEOUT_S or 600) bounds every collective: a rank that died leaves its peers with an error after
    minutes, not after the process-group default of tens of minutes."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available() and backend != "gloo"
    device = torch.device(f"cuda:{local}") if use_cuda else torch.device("cpu")
    if use_cuda:
        n_vis = torch.cuda.device_count()
        if local >= n_vis:
            raise RuntimeError(f"rank {rank} (LOCAL_RANK {local}) has no GPU of its own: {n_vis} device(s) visible; "
                               "the frame-parallel path runs one process per GPU and never shares a device")
        torch.cuda.set_device(device)
        if world > 1:
            # ranks on THIS node (torchrun exports LOCAL_WORLD_SIZE; otherwise the GPUs visible here): the global world size would give a
            # multi-node job slices for ranks that live elsewhere and pack the local ones into the first 1 / nnodes of the cores
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "0")) or min(world, n_vis)
            pin_host_thread(device, local, max(local_world, local + 1))
    if world > 1 and not dist.is_initialized():
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        t = float(timeout_s if timeout_s is not None else os.environ.get("FDGS_PG_TIMEOUT_S", "600"))
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=t))
    return rank, world, device


# ---- host placement: one rank = one host thread that enqueues ~40 launches per frame; on an 8-GPU node the ranks' threads are kept on the
# cores of THEIR GPU's NUMA node (the enqueue path is ~1 ms of a 1.8 ms step: a thread migrating across sockets, or eight of them sharing a
# few cores, shows up directly in the p90).  FDGS_NO_PIN=1 opts out.
pinned_to = None          # what pin_host_thread did in this process (reported per rank by bench.py)


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/.../local_cpulist)."""
    cpus = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        else:
            cpus.append(int(part))
    return sorted(set(cpus))


def cpus_for_rank(local_rank, world, allowed, numa_cpus=None):
    """Host cores for rank `local_rank` of `world` on this node: the allowed cores of its GPU's NUMA node when known -- shared fairly when
    several ranks' GPUs sit on the same node (the k-th of m ranks on a node gets the k-th m-th of its cores; `numa_cpus` = one core list per
    rank) -- else an even slice of the allowed cores.  Never empty."""
    allowed = sorted(allowed)
    if numa_cpus is not None and numa_cpus[local_rank]:
        mine = [c for c in numa_cpus[local_rank] if c in set(allowed)]
        peers = [r for r in range(world) if numa_cpus[r] == numa_cpus[local_rank]]
        if mine:
            k, m = peers.index(local_rank), len(peers)
            share = mine[k * len(mine) // m:(k + 1) * len(mine) // m]
            return share or mine
    n = len(allowed)
    share = allowed[local_rank * n // world:(local_rank + 1) * n // world]
    return share or allowed


def _numa_cpus_of_gpu(index):
    """Cores local to GPU `index` (sysfs local_cpulist of its PCI function), or None when the platform does not say."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f"{int(getattr(pr, 'pci_domain_id', 0)):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            cpus = parse_cpulist(f.read())
        return cpus or None
    except Exception:
        return None


def pin_host_thread(device, local_rank, world):
    """sched_setaffinity of this process to its share of the host cores (see above).  Returns the core list, or None when pinning is off
    or unsupported."""
    global pinned_to
    if os.environ.get("FDGS_NO_PIN", "0") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        allowed = os.sched_getaffinity(0)
        numa = [_numa_cpus_of_gpu(r) for r in range(world)] if device.type == "cuda" else None
        if numa is not None and any(x is None for x in numa):
            numa = None
        cpus = cpus_for_rank(local_rank, world, allowed, numa)
        os.sched_setaffinity(0, cpus)
        pinned_to = {"cpus": len(cpus), "first": cpus[0], "last": cpus[-1], "by": "numa_node_of_gpu" if numa is not None else "even_slice"}
        return cpus
    except Exception:
        return None


def check_same_plan(counts, device, what="densify"):
    """Data-parallel replicas must restructure identically: all-reduce MIN and MAX of `counts` (ints) and raise on EVERY rank when they differ
    -- before anything depends on the plan (a rank that went on alone would leave the others blocked in the next collective until the
    process-group timeout).  No-op in a single-process job."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    mine = torch.tensor([int(c) for c in counts], device=device, dtype=torch.int64)
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError(f"{what}: ranks disagree on the plan (kept, clones, splits, samples-drawn-here): min {lo.tolist()} max {hi.tolist()}, "
                           f"this rank {mine.tolist()} -- reduce the densification statistics over the ranks first "
                           "(parallel.allreduce_densification_stats)")


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


PORT_CLASH_EXIT = 97


def _spawn_entry(rank, world, port, fn, args):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL needs it)
    try:
        fn(*args)
    except BaseException as e:
        # a rendezvous port taken between the parent's probe and rank 0's bind is the ONE failure worth a relaunch: it gets its own exit code
        # (anything else -- an assertion in fn, a bad argument -- must be reported once, not re-executed with its side effects)
        if "address already in use" in repr(e).lower() or "eaddrinuse" in repr(e).lower():
            import traceback
            traceback.print_exc()
            os._exit(PORT_CLASH_EXIT)
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def spawn_local(world, fn, args=(), poll_s=0.2, attempts=3):
    """Launcher-free form of `torch.distributed.run --nnodes=1 --nproc-per-node world`: starts `world` processes on this
    node, rank r with RANK = LOCAL_RANK = r (-> cuda:r in init_from_env), rendezvous on 127.0.0.1 at a free port, and
    calls fn(*args) in each.  All ranks are polled together: the first non-zero exit terminates the siblings (they would otherwise
    sit in a collective until the process-group timeout) and raises.  The rendezvous port is probed and released before the children
    bind it; if another process takes it in between, the rank that hits "address already in use" exits with PORT_CLASH_EXIT and the launch
    is retried on a new port (`attempts` times).  Any other failure is reported once.  `fn` must be importable (module-level)."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    last = None
    for attempt in range(attempts):
        port = _free_port()
        procs = [ctx.Process(target=_spawn_entry, args=(r, world, port, fn, args)) for r in range(world)]
        for p in procs:
            p.start()
        bad = []
        while True:
            codes = [p.exitcode for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad or all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
        if not bad:
            return
        for p in procs:                       # first failure: do not leave the others blocked in a collective
            if p.exitcode is None:
                p.terminate()
        for p in procs:
            p.join(10)
            if p.exitcode is None:
                p.kill()
                p.join(5)
        last = bad
        # only a lost rendezvous port (its own exit code, _spawn_entry) is retried; anything else is the job's own failure
        if not any(c == PORT_CLASH_EXIT for _, c in bad) or attempt == attempts - 1:
            break
    raise RuntimeError(f"spawn_local: ranks exited non-zero: {last}")


def ranks_seen(device):
    """Number of ranks that took part in one all-reduce(SUM) of ones: proof that every rank joined the RCCL job."""
    t = torch.ones(1, device=device, dtype=torch.float32)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(round(float(t.item())))


def gather_floats(value, device):
    """[value of rank 0, value of rank 1, ...] on every rank."""
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t)
        return [float(x.item()) for x in out]
    return [float(value)]


def gather_objects(obj):
    """[obj of rank 0, obj of rank 1, ...] on every rank (small picklable objects: device identities, per-rank statistics)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]


def device_identity(device):
    """What proves which physical GPU a rank ran on: UUID and PCI bus id where the runtime reports them, plus name and LOCAL_RANK."""
    ident = {"device": str(device), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid(), "host_cpus_pinned": pinned_to}
    if device.type == "cuda":
        pr = torch.cuda.get_device_properties(device)
        ident["name"] = pr.name
        for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
            v = getattr(pr, k, None)
            if v is not None:
                ident[k] = str(v)
    return ident


def allreduce_latency_us(device, iters=50):
    """Blocking latency of the path's only collective (the 3-float loss-statistics all-reduce), mean of `iters` after 5 warm-ups, MAX over
    ranks.  In the timed steps it runs asynchronously under the next frame; this is what it would cost on the critical path."""
    import time
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return None
    t = torch.zeros(3, device=device)
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    for _ in range(5):
        dist.all_reduce(t)
    sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        dist.all_reduce(t)
    sync()
    return max_over_ranks((time.perf_counter() - t0) / iters * 1e6, device)


def frames_for_rank(n_frames_total, step, rank, world):
    """Index of the frame rank `rank` renders at step `step` (round-robin over a camera list)."""
    return (step * world + rank) % n_frames_total


def allreduce_loss_stats(acc, async_op=False):
    """acc: float tensor [3] = [sum|err|, sum err^2, n] of this rank's frame -> global sums (in place).
    `async_op=True` returns a handle (None in a single-process job) instead of making the compute stream wait: the collective then runs
    on RCCL's own stream UNDER the next frame's kernels, and the ranks are no longer forced into lock-step once per frame (a
    12-byte all-reduce is pure latency, ~20 us on the critical path of a 2 ms step, and a per-step rendezvous makes every step as slow as
    the slowest rank's).  Call `.wait()` on the handle before reading `acc` or handing the buffer to the next reduction."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if async_op:
            return dist.all_reduce(acc, op=dist.ReduceOp.SUM, async_op=True)
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return None if async_op else acc


def loss_from_stats(acc):
    """(L1, PSNR) of the global batch from reduced statistics (utils/loss_utils.py:20-21, utils/image_utils.py:17-38)."""
    n = acc[2].clamp_min(1)
    l1 = acc[0] / n
    mse = acc[1] / n
    psnr = 20 * torch.log10(1.0 / torch.sqrt(mse.clamp_min(1e-20)))
    return l1, psnr


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- data-parallel train step: every rank renders its own views, gradients are summed ----
def allreduce_gradients(params, bucket_bytes=512 << 20, average=False, arena_zero_copy=True):
    """Sum (or average) `.grad` of `params` over the ranks, in place.

    The reference sums the losses of the `batch_size` views of one step before one backward (train.py:180-219), so the
    gradient of a step is the SUM over its views: with the views spread over ranks that is one all-reduce(sum).  Gradients
    are packed into few large flat buckets first (one for the whole model at 300 k Gaussians: ~71 MB of Gaussian rows +
    <= 30 MB of planes and MLP weights): xGMI is point-to-point, a ring all-reduce is bound per link, and a handful of
    100-MB messages use the links far better than hundreds of small tensors would.  The parameter LIST (same order on
    every rank) is authoritative: one small all-reduce(MAX) of a has-grad mask runs first; a parameter whose grad is None
    on every rank is skipped everywhere, one whose grad is None on this rank only gets `zeros_like(p)` here, so the flat
    buckets have the same size on all ranks whatever each rank's views happened to touch.
    Returns the number of gradient collectives issued (the mask exchange is not counted)."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return 0
    world = dist.get_world_size()
    params = list(params)
    if not params:
        return 0
    dev0 = next((p.grad.device for p in params if p.grad is not None), params[0].device)
    # Layout signature: the fused render path hands every gradient back as a VIEW OF ONE ARENA (deformation.backward_prepare), so the
    # arena itself can be the all-reduce buffer -- no torch.cat into a bucket, no copy back.  That is only legal when every rank has
    # the same view structure (same parameter -> (storage ordinal, element offset, numel)); a rank whose gradient is None, or that came
    # through another code path, has a different signature, MIN != MAX below, and every rank takes the packing path together.
    stor, layout = {}, []
    for p_ in params:
        g = p_.grad
        # (dense = numel consecutive elements from storage_offset: row-major, or channels-last like the HexPlane gradients)
        if g is None:
            layout.append(None)          # no gradient here: fine when that is so on every rank (the signature covers the pattern)
            continue
        if not (g.is_contiguous() or (g.dim() == 4 and g.is_contiguous(memory_format=torch.channels_last))):
            layout.append("strided")     # cannot be reduced through a flat view of its storage: packing path
            continue
        sid = stor.setdefault(g.untyped_storage().data_ptr(), len(stor))
        layout.append((sid, g.storage_offset(), g.numel(), str(g.dtype)))
    # (a digest, not hash(): Python salts str hashes per process, and the signature has to agree ACROSS processes)
    import hashlib
    sig = int.from_bytes(hashlib.blake2b(repr(layout).encode(), digest_size=7).digest(), "little")
    s_lo, s_hi = int(sig & 0x3FFFFFFF), int((sig >> 30) & 0x3FFFFFF)
    has = torch.tensor([0 if p_.grad is None else 1 for p_ in params] + [s_lo, s_hi, -s_lo, -s_hi], dtype=torch.int32, device=dev0)
    dist.all_reduce(has, op=dist.ReduceOp.MAX)
    has = has.tolist()
    same_layout = all(l != "strided" for l in layout) and any(l is not None for l in layout) and has[-4] == -has[-2] and has[-3] == -has[-1]
    # ... and it only pays when it does not multiply the collectives: every dtype's gradients in ONE storage (separately allocated
    # gradients are better packed into one bucket than reduced one by one)
    per_dtype = {}
    for l in layout:
        if isinstance(l, tuple):
            per_dtype.setdefault(l[3], set()).add(l[0])
    same_layout = same_layout and all(len(v) == 1 for v in per_dtype.values())
    # ... and only when the listed gradients COVER the span that would be reduced: the arena also holds gradients of parameters that are
    # not in `params` (a caller reducing [xyz, rotation] and then [scaling], or skipping a frozen tensor in the middle), and those must not
    # be summed -- or divided -- a second time.  Gaps below the arena's 64-element alignment are padding nobody reads.  (A function of the
    # layout alone, which the digest above has just shown to be the same on every rank: every rank takes the same branch.)
    if same_layout:
        by_storage = {}
        for l in layout:
            if isinstance(l, tuple):
                by_storage.setdefault(l[0], []).append((l[1], l[1] + l[2]))
        for iv in by_storage.values():
            iv.sort()
            if any(b0 - a1 >= 64 or b0 < a1 for (_, a1), (b0, _) in zip(iv, iv[1:])):
                same_layout = False
    has = has[:-4]
    if same_layout and arena_zero_copy:
        # one flat view per storage, spanning its first to its last gradient element (alignment gaps inside the span are reduced
        # along with the rest: nobody reads them), cut into bucket-sized pieces without copying
        calls = 0
        spans = {}
        for p_, l in zip(params, layout):
            if l is None:
                continue                 # (None on every rank, or the signatures would differ)
            lo, hi = spans.get(l[0], (l[1], l[1] + l[2]))
            spans[l[0]] = (min(lo, l[1]), max(hi, l[1] + l[2]))
            spans.setdefault(("t", l[0]), p_.grad)
        for sid in sorted(k for k in spans if not isinstance(k, tuple)):
            lo, hi = spans[sid]
            g0 = spans[("t", sid)]
            flat = torch.empty(0, dtype=g0.dtype, device=g0.device).set_(g0.untyped_storage(), lo, (hi - lo,))
            step = max(1, bucket_bytes // flat.element_size())
            for a0 in range(0, hi - lo, step):
                piece = flat[a0:a0 + step]
                dist.all_reduce(piece, op=dist.ReduceOp.SUM)
                if average:
                    piece.div_(world)
                calls += 1
        return calls
    groups = {}
    for p, h in zip(params, has):
        if not h:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        groups.setdefault((p.grad.dtype, p.grad.device), []).append(p.grad)
    calls = 0
    for (_, _), grads in sorted(groups.items(), key=lambda kv: str(kv[0])):
        bucket, size = [], 0
        for g in grads + [None]:
            if g is not None and (not bucket or size + g.numel() * g.element_size() <= bucket_bytes):
                bucket.append(g)
                size += g.numel() * g.element_size()
                continue
            flat = torch.cat([b.reshape(-1) for b in bucket])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            if average:
                flat.div_(world)
            off = 0
            for b in bucket:
                b.copy_(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            calls += 1
            bucket, size = ([g], g.numel() * g.element_size()) if g is not None else ([], 0)
    return calls


def allreduce_densification_stats(radii, visibility_filter, viewspace_grad):
    """Cross-rank form of what train.py:195-196,219-225 does over the views of one step: element-wise MAX of the radii,
    OR of the visibility masks, SUM of the view-space position gradients.  Returns the three reduced tensors."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return radii, visibility_filter, viewspace_grad
    r = radii.clone()
    dist.all_reduce(r, op=dist.ReduceOp.MAX)
    v = visibility_filter.to(torch.int32)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    g = viewspace_grad.clone()
    dist.all_reduce(g, op=dist.ReduceOp.SUM)
    return r, v.bool(), g
