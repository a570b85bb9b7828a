"""Drop-ins for the image losses the reference evaluates right after render() (train.py:201-214):

    from utils.loss_utils import l1_loss, ssim, l2_loss        ->  fdgs.losses.l1_loss / ssim / l2_loss
    from utils.image_utils import psnr                         ->  fdgs.losses.psnr

same names, argument meaning and return shapes (utils/loss_utils.py:20-66, utils/image_utils.py:14-38), computed by the
HIP kernels of csrc/loss.hip and csrc/api.hip, plus `image_loss`, the fused form of train.py's
`Ll1 + lambda_dssim * (1 - ssim)` (one forward launch for all statistics, one backward launch for dL/dimage).
Gradients flow to the first (rendered) image only; the ground-truth image must not require grad.  No CPU fallback.
"""
from collections import namedtuple

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _as4(t):
    if t.dim() == 3:
        return t[None]
    if t.dim() != 4:
        raise ValueError("images must be [C,H,W] or [B,C,H,W]")
    return t


def _pair(a, b):
    if a.device.type != "cuda":
        raise _lib.FdgsError("the image-loss kernels run on the GPU only")
    if a.shape != b.shape:
        raise ValueError(f"image shapes differ: {tuple(a.shape)} vs {tuple(b.shape)}")
    if b.requires_grad:
        raise NotImplementedError("gradients flow to the rendered image only; the ground-truth image must not require grad")
    return a.detach().float().contiguous(), b.detach().float().contiguous()


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, grad_on):
        x, y = _pair(a, b)
        n = x.numel()
        acc = torch.zeros(3, device=x.device, dtype=torch.float32)
        # grad_on = torch.is_grad_enabled() at the call site: needs_input_grad stays True under torch.no_grad(), and
        # evaluation (training_report, render.py) must not pay for a gradient image that no backward will read
        grad = torch.empty_like(x) if (grad_on and ctx.needs_input_grad[0]) else None
        ctx.saved_grad_image = grad is not None
        check(_lib.lib().fdgs_l1_stats(stream_ptr(), n, ptr(x), ptr(y), 1.0 / max(n, 1), ptr(grad), ptr(acc)))
        if grad is not None:
            ctx.save_for_backward(grad)
        return acc[0] / max(n, 1)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None


def l1_loss(network_output, gt):
    """mean |network_output - gt| (utils/loss_utils.py:20-21)."""
    return _L1.apply(network_output, gt, torch.is_grad_enabled())


def l2_loss(network_output, gt):
    """mean (network_output - gt)^2 (utils/loss_utils.py:23-24; imported but never called by train.py) -- plain torch ops."""
    return ((network_output - gt) ** 2).mean()


def mse(img1, img2):
    """per batch item mean squared error, [B,1] (utils/image_utils.py:14-15)."""
    x, y = _pair(_as4(img1), _as4(img2))
    B = x.shape[0]
    acc = torch.zeros(B, 3, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    n = x.numel() // max(B, 1)
    for i in range(B):
        check(L.fdgs_l1_stats(stream_ptr(), n, ptr(x[i]), ptr(y[i]), 0.0, None, ptr(acc[i])))
    return (acc[:, 1] / max(n, 1)).view(B, 1)


@torch.no_grad()
def psnr(img1, img2, mask=None):
    """20 log10(1 / sqrt(mse)) per batch item, [B,1] (utils/image_utils.py:17-38).  With a mask the reference selects the
    masked pixels of all three channels before the mean; that (evaluation-only) variant runs as torch ops on the device."""
    if mask is not None:
        a, b = img1.flatten(1), img2.flatten(1)
        m = mask.flatten(1).repeat(3, 1) != 0
        a, b = a[m], b[m]
        e = ((a - b) ** 2).view(a.shape[0], -1).mean(1, keepdim=True)
        p = 20 * torch.log10(1.0 / torch.sqrt(e.float()))
        return p[~torch.isinf(p)] if torch.isinf(p).any() else p
    return 20 * torch.log10(1.0 / torch.sqrt(mse(img1, img2)))


def _fwd(x, y, want_maps):
    B, C, H, W = x.shape
    acc = torch.zeros(B, 4, device=x.device, dtype=torch.float32)
    maps = torch.empty(3, B * C, H, W, device=x.device, dtype=torch.float32) if want_maps else None
    check(_lib.lib().fdgs_image_loss_fwd(stream_ptr(), B, C, H, W, ptr(x), ptr(y), ptr(maps), ptr(acc)))
    return acc, maps


def _bwd(x, y, maps, w_l1, w_ssim, gs):
    B, C, H, W = x.shape
    dimg = torch.empty_like(x)
    check(_lib.lib().fdgs_image_loss_bwd(stream_ptr(), B, C, H, W, ptr(x), ptr(y), ptr(maps), w_l1, w_ssim, ptr(gs), ptr(dimg)))
    return dimg


class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, size_average, grad_on):
        x, y = _pair(_as4(a), _as4(b))
        need = bool(grad_on and ctx.needs_input_grad[0])
        ctx.saved_maps = need
        acc, maps = _fwd(x, y, need)
        ctx.size_average, ctx.shape = size_average, a.shape
        if need:
            ctx.save_for_backward(x, y, maps)
        n_item = max(x.numel() // max(x.shape[0], 1), 1)
        return acc[:, 3].sum() / (n_item * x.shape[0]) if size_average else acc[:, 3] / n_item

    @staticmethod
    def backward(ctx, g):
        x, y, maps = ctx.saved_tensors
        n_item = max(x.numel() // max(x.shape[0], 1), 1)
        if ctx.size_average:
            gs = g.detach().float().reshape(1).contiguous()
            d = _bwd(x, y, maps, 0.0, 1.0 / (n_item * x.shape[0]), gs)
        else:
            d = _bwd(x, y, maps, 0.0, 1.0 / n_item, None) * g.detach().float().view(-1, 1, 1, 1)
        return d.view(ctx.shape), None, None, None


def ssim(img1, img2, window_size=11, size_average=True):
    """SSIM with the reference's 11x11 Gaussian window (utils/loss_utils.py:40-66): the mean of the map, or one mean per
    batch item when size_average is False."""
    if window_size != 11:
        raise NotImplementedError("the kernel implements the reference's fixed window_size = 11")
    return _SSIM.apply(img1, img2, bool(size_average), torch.is_grad_enabled())


ImageLoss = namedtuple("ImageLoss", ["loss", "l1", "mse", "ssim"])


class _ImageLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, lam, grad_on):
        x, y = _pair(_as4(a), _as4(b))
        n = max(x.numel(), 1)
        need = bool(grad_on and ctx.needs_input_grad[0])
        ctx.saved_maps = need
        ctx.shape = a.shape
        if lam == 0.0:
            acc = torch.zeros(3, device=x.device, dtype=torch.float32)
            grad = torch.empty_like(x) if need else None
            check(_lib.lib().fdgs_l1_stats(stream_ptr(), x.numel(), ptr(x), ptr(y), 1.0 / n, ptr(grad), ptr(acc)))
            ctx.fused = False
            if need:
                ctx.save_for_backward(grad)
            l1, sq = acc[0] / n, acc[1] / n
            s = torch.full((), float("nan"), device=x.device)
            ctx.mark_non_differentiable(l1, sq, s)
            return l1.clone(), l1, sq, s
        acc, maps = _fwd(x, y, need)
        tot = acc.sum(0)
        l1, sq, s = tot[0] / n, tot[1] / n, tot[3] / n
        ctx.fused, ctx.lam = True, lam
        if need:
            ctx.save_for_backward(x, y, maps)
        ctx.mark_non_differentiable(l1, sq, s)
        return l1 + lam * (1.0 - s), l1, sq, s

    @staticmethod
    def backward(ctx, g, *_unused):
        if not ctx.fused:
            (grad,) = ctx.saved_tensors
            return (g * grad).view(ctx.shape), None, None, None
        x, y, maps = ctx.saved_tensors
        n = max(x.numel(), 1)
        gs = g.detach().float().reshape(1).contiguous()
        return _bwd(x, y, maps, 1.0 / n, -ctx.lam / n, gs).view(ctx.shape), None, None, None


def image_loss(image, gt, lambda_dssim=0.0):
    """Fused `Ll1 + lambda_dssim * (1 - ssim(image, gt))` of train.py:201-214.  Returns ImageLoss(loss, l1, mse, ssim):
    `loss` carries the gradient to `image`; l1 / mse / ssim are detached 0-dim device tensors (psnr = -10 log10(mse);
    ssim is NaN when lambda_dssim == 0 and the SSIM pass is skipped, as the reference skips it)."""
    return ImageLoss(*_ImageLoss.apply(image, gt, float(lambda_dssim), torch.is_grad_enabled()))
