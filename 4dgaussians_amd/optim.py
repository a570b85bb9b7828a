"""Drop-in for the reference's optimizer (`torch.optim.Adam(l, lr=0.0, eps=1e-15)`, scene/gaussian_model.py:184;
stepped at train.py:291): the same `torch.optim.Optimizer` surface -- `param_groups` with per-group "lr"/"name" that
`update_learning_rate` (scene/gaussian_model.py:198-212) rewrites every iteration, and a per-parameter `state` dict with
the keys "step", "exp_avg", "exp_avg_sq" that the densification code edits in place (`replace_tensor_to_optimizer`,
`_prune_optimizer`, `cat_tensors_to_optimizer`, :316-400) and that `state_dict()` / `load_state_dict()` persist in
checkpoints -- but `step()` is ONE multi-tensor HIP launch (csrc/adam.hip) instead of ~5 elementwise kernels per tensor.

    self.optimizer = fdgs.FusedAdam(l, lr=0.0, eps=1e-15)        # scene/gaussian_model.py:184
"""
import torch

from . import _lib
from ._lib import AdamTensor, check, stream_ptr


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam implements the reference's configuration: no weight decay, no amsgrad")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _lib.lib()
        buckets = {}   # (beta1, beta2, eps) -> list of descriptors; one launch per distinct hyper-parameter set
        keep = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda" or p.dtype != torch.float32:
                    raise _lib.FdgsError("FusedAdam steps float32 parameters on the GPU only")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                g, m, v = p.grad, st["exp_avg"], st["exp_avg_sq"]
                # one flat walk over memory: all four tensors must be dense with the same strides
                if not _dense(p):
                    raise _lib.FdgsError("FusedAdam needs dense (non-overlapping) parameters")
                if g.stride() != p.stride():
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                if m.stride() != p.stride():
                    m = st["exp_avg"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(m)
                if v.stride() != p.stride():
                    v = st["exp_avg_sq"] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(v)
                st["step"] += 1
                d = AdamTensor()
                d.param, d.grad, d.exp_avg, d.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                d.n, d.lr, d.step = p.numel(), float(group["lr"]), int(st["step"])
                buckets.setdefault((float(b1), float(b2), float(group["eps"])), []).append(d)
                keep.append((g, m, v))
        for (b1, b2, eps), ds in buckets.items():
            arr = (AdamTensor * len(ds))(*ds)
            check(L.fdgs_adam_step(stream_ptr(), len(ds), arr, b1, b2, eps))
        return loss


def _dense(t):
    """True when the tensor's elements tile one contiguous memory range in some dimension order (contiguous, channels_last,
    any permutation of a contiguous tensor): the kernel walks memory linearly, the logical order is irrelevant to Adam."""
    sizes_strides = sorted((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1)
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True
