"""Times the fused image loss (csrc/loss.hip) at the BASELINE config-4 image size: python tools/loss_time.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
losses = importlib.import_module("4dgaussians_amd.losses")
torch.manual_seed(0)
gt = torch.rand(1, 3, 1014, 1352, device="cuda"); img = (gt + 0.1 * torch.randn_like(gt)).clamp(0, 1).requires_grad_(True)
for lam in (0.0, 0.2):
    for _ in range(5):
        o = losses.image_loss(img, gt, lam); o.loss.backward(); img.grad = None
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    for _ in range(50):
        o = losses.image_loss(img, gt, lam)
    e[1].record()
    for _ in range(50):
        o = losses.image_loss(img, gt, lam); o.loss.backward(); img.grad = None
    e[2].record(); torch.cuda.synchronize()
    print("lambda", lam, "fwd %.3f ms  fwd+bwd %.3f ms" % (e[0].elapsed_time(e[1]) / 50, e[1].elapsed_time(e[2]) / 50))
