"""Generates tests/golden/train_loss_vectors.npz by running the REFERENCE's own loss code
(/root/reference/gaussian_splatting/utils/loss_utils.py, pure torch) in the build container:
l1_loss, ssim and the autograd gradient of (1 - lambda) * L1 + lambda * (1 - ssim) w.r.t. the image.
Only inputs and outputs are stored.  Run:  python tests/golden/make_train_loss_golden.py"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/gaussian_splatting/utils/loss_utils.py"
spec = importlib.util.spec_from_file_location("ref_loss_utils", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for name, (H, W, seed) in {"a": (40, 56, 0), "b": (17, 33, 1), "c": (64, 48, 2)}.items():
    rng = np.random.default_rng(seed)
    # smooth-ish images in [0, 1] plus noise, so that both flat and textured windows occur
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base = 0.5 + 0.4 * np.sin(xx / 7.0 + seed) * np.cos(yy / 5.0)
    gt = np.clip(np.stack([base, base[::-1], base.T[:H, :W] if base.T.shape == (H, W) else base * 0.8]) +
                 0.05 * rng.normal(size=(3, H, W)), 0, 1).astype(np.float32)
    img = np.clip(gt + 0.1 * rng.normal(size=(3, H, W)), 0, 1).astype(np.float32)
    for lam in (0.2,):
        x = torch.tensor(img, requires_grad=True)
        g = torch.tensor(gt)
        Ll1 = ref.l1_loss(x, g)
        s = ref.ssim(x, g)
        loss = (1.0 - lam) * Ll1 + lam * (1.0 - s)
        loss.backward()
        out[f"{name}_img"], out[f"{name}_gt"] = img, gt
        out[f"{name}_lambda"] = np.float32(lam)
        out[f"{name}_Ll1"], out[f"{name}_ssim"], out[f"{name}_loss"] = np.float32(Ll1.item()), np.float32(s.item()), np.float32(loss.item())
        out[f"{name}_grad"] = x.grad.numpy().astype(np.float32)
out["window_1d"] = ref.gaussian(11, 1.5).numpy()
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_loss_vectors.npz")
np.savez_compressed(path, **out)
print(path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if not k.endswith(("img", "gt", "grad"))})
