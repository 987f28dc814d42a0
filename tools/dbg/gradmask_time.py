"""Device time of gsr_grad_mask's launches at 640x480 / 1024x576 (HIP events around 200 back-to-back calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import pipelines as PL
for H, W in ((480, 640), (576, 1024)):
    img = torch.rand(3, H, W, device="cuda:0")
    kp = np.random.default_rng(0).uniform(0, min(H, W) - 1, (500, 2)).astype(np.float32)
    kpt = torch.tensor(kp, device="cuda:0")
    for k in (None, kpt):
        PL.grad_mask(img, 1.1, k)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            PL.grad_mask(img, 1.1, k)
        b.record(); torch.cuda.synchronize()
        print(f"{W}x{H} keypoints {0 if k is None else 500}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (host-bound if launches dominate)")
