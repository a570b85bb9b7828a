"""Generates tests/golden/loss_ssim.npz from the REFERENCE's own l1_loss / ssim (utils/loss_utils.py:20-66), value and
autograd gradient wrt the first image.  Run in the build container (needs /root/reference):
    python tests/golden/make_loss_golden.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import loss_oracle as LO  # noqa: E402

ref = LO.import_reference_loss_utils()
gen = torch.Generator().manual_seed(777)
B, C, H, W = 2, 3, 37, 53                     # not multiples of the 32-pixel tile; smaller than one tile in H on purpose
gt = torch.rand(B, C, H, W, generator=gen)
gt = torch.nn.functional.avg_pool2d(gt, 5, 1, 2)            # some spatial structure
img = (gt + 0.15 * torch.randn(B, C, H, W, generator=gen)).clamp(0, 1).requires_grad_(True)
lam = 0.2
l1 = ref.l1_loss(img, gt)
s = ref.ssim(img, gt)
loss = l1 + lam * (1.0 - s)
(grad,) = torch.autograd.grad(loss, img, retain_graph=True)
(grad_s,) = torch.autograd.grad(s, img)
s_items = ref.ssim(img, gt, size_average=False)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loss_ssim.npz"),
                    img=img.detach().numpy(), gt=gt.numpy(), lambda_dssim=np.float64(lam), l1=np.float64(l1.item()),
                    ssim=np.float64(s.item()), ssim_items=s_items.detach().numpy(), loss=np.float64(loss.item()),
                    grad=grad.numpy(), grad_ssim=grad_s.numpy())
print("wrote loss_ssim.npz: l1 %.6f ssim %.6f loss %.6f" % (l1.item(), s.item(), loss.item()))
