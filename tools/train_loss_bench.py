"""Times the fused training-loss epilogue (gsr_training_loss, forward + gradient) at 640x480 against the reference's
torch formulation (loss_utils-style ssim via F.conv2d + autograd) on the same GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from gs_localization_amd import train_epilogue as TE
dev = torch.device("cuda:0")
H, W = 480, 640
torch.manual_seed(0)
gt = torch.rand(3, H, W, device=dev); img = (gt + 0.05 * torch.randn_like(gt)).clamp(0, 1)
depth = 2 + torch.rand(H, W, device=dev); pseudo = 100 / depth + torch.randn_like(depth)

def torch_style(x, d):
    g1 = torch.tensor([np.exp(-(i - 5) ** 2 / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float32, device=dev); g1 = g1 / g1.sum()
    w = (g1[:, None] @ g1[None, :])[None, None].expand(3, 1, 11, 11).contiguous()
    c = lambda t: F.conv2d(t, w, padding=5, groups=3)
    m1, m2 = c(x), c(gt)
    s = (((2 * m1 * m2 + 1e-4) * (2 * (c(x * gt) - m1 * m2) + 9e-4)) / ((m1 * m1 + m2 * m2 + 1e-4) * (c(x * x) - m1 * m1 + c(gt * gt) - m2 * m2 + 9e-4))).mean()
    def pc(a, b):
        a = a - a.mean(); b = b - b.mean()
        return (a * b).sum() / ((a * a).sum() * (b * b).sum()).sqrt()
    dd = d.reshape(-1)
    pd = torch.minimum(1 - pc(-pseudo.reshape(-1), dd), 1 - pc(1 / (pseudo.reshape(-1) + 200.), dd))
    return 0.8 * (x - gt).abs().mean() + 0.2 * (1 - s) + 0.1 * pd

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def fused():
    x = img.clone().requires_grad_(True); d = depth.clone().requires_grad_(True)
    TE.training_loss(x, gt, 0.2, d, pseudo, 0.1).backward()
def ref():
    x = img.clone().requires_grad_(True); d = depth.clone().requires_grad_(True)
    torch_style(x, d).backward()
tf, tr = timeit(fused), timeit(ref)
N = H * W
bytes_moved = 3 * N * (8 + 12) + 3 * N * (12 + 8 + 4) + N * (8 + 8 + 4)
print(f"training loss fwd+bwd 640x480: fused {tf:.1f} us, torch ops {tr:.1f} us ({tr / tf:.1f}x); {bytes_moved / 1e6:.1f} MB algorithmic "
      f"=> {bytes_moved / tf / 1e3:.0f} GB/s ({bytes_moved / tf / 1e3 / 8000:.3f} of HBM peak; launch-bound at this size)")
