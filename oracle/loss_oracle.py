"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, float32 or float64) of the reference's image losses.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows:
  utils/loss_utils.py:20-21   l1_loss  = mean |a - b|
  utils/loss_utils.py:26-28   gaussian(11, 1.5): exp(-(x-5)^2 / (2 sigma^2)) as a float32 tensor, divided by its sum
  utils/loss_utils.py:30-34   create_window: outer product of that vector, one copy per channel (depthwise filter)
  utils/loss_utils.py:46-66   _ssim: mu = w*img, sigma = w*(img img) - mu^2, map = (2 mu1 mu2 + C1)(2 s12 + C2) /
                              ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)), C1 = 0.01^2, C2 = 0.03^2, zero padding 5, mean of the map
  utils/image_utils.py:14-38  mse / psnr per batch item: 20 log10(1 / sqrt(mean (a-b)^2))
The filter is applied here as a row pass followed by a column pass (the window is an outer product), which is also how
the HIP kernel does it; tests/test_oracle_loss.py pins this against the reference's own ssim() (imported from
/root/reference where present) and against the committed golden vector tests/golden/loss_ssim.npz.
"""
import math

import torch
import torch.nn.functional as F

C1, C2 = 0.01 ** 2, 0.03 ** 2


def window_1d(dtype=torch.float32):
    v = torch.tensor([math.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=torch.float32)
    return (v / v.sum()).to(dtype)


def blur(img):
    """Depthwise 11x11 Gaussian blur with zero padding; img [B,C,H,W]."""
    c = img.shape[1]
    g = window_1d(img.dtype)
    rows = F.conv2d(img, g.view(1, 1, 1, 11).expand(c, 1, 1, 11), padding=(0, 5), groups=c)
    return F.conv2d(rows, g.view(1, 1, 11, 1).expand(c, 1, 11, 1), padding=(5, 0), groups=c)


def ssim_map(a, b):
    mu1, mu2 = blur(a), blur(b)
    s1 = blur(a * a) - mu1 * mu1
    s2 = blur(b * b) - mu2 * mu2
    s12 = blur(a * b) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def ssim(a, b, size_average=True):
    if a.dim() == 3:
        a, b = a[None], b[None]
    m = ssim_map(a, b)
    return m.mean() if size_average else m.flatten(1).mean(1)


def l1_loss(a, b):
    return (a - b).abs().mean()


def psnr(a, b):
    if a.dim() == 3:
        a, b = a[None], b[None]
    mse = ((a - b) ** 2).flatten(1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def import_reference_loss_utils():
    """The reference's own utils/loss_utils.py (needs /root/reference); its `import lpips` is satisfied by an empty stub."""
    import importlib.util
    import sys
    import types
    sys.modules.setdefault("lpips", types.ModuleType("lpips"))
    spec = importlib.util.spec_from_file_location("_ref_loss_utils", "/root/reference/utils/loss_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
