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


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    device = torch.device(f"cuda:{local}") if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def frames_for_rank(n_frames_total, step, rank, world):
    """Index of the frame rank `rank` renders at step `step` (round-robin over a camera list)."""
    return (step * world + rank) % n_frames_total


def allreduce_loss_stats(acc):
    """acc: float tensor [3] = [sum|err|, sum err^2, n] of this rank's frame -> global sums (in place)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    return acc


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
