"""CPU restatement (torch, float64 by default) of the training-step loss -- TEST INFRASTRUCTURE ONLY.

  l1_loss, window_1d, ssim                       gaussian_splatting/utils/loss_utils.py:17-63
  training_loss                                  gaussian_splatting/train.py:92-108
  pearson_corrcoef                               torchmetrics.functional.regression (dependency of train.py:25, not in
                                                 this image): cov / sqrt(var_x * var_y), clamped to [-1, 1]
  densification_stats                            train.py:142-145, scene/gaussian_model.py:405-407

Pinned: l1_loss and ssim (values and autograd gradients) against the reference's own loss_utils.py, imported in the
build container by tests/golden/make_train_loss_golden.py -> tests/golden/train_loss_vectors.npz.
Parity unpinned: the Pearson term (torchmetrics is absent) -- restated from its published definition."""
from math import exp

import numpy as np
import torch
import torch.nn.functional as F


def l1_loss(x, y):
    """loss_utils.py:17-18"""
    return (x - y).abs().mean()


def window_1d(size=11, sigma=1.5):
    """loss_utils.py:23-25: float32 tensor of Python-float exponentials, normalised in float32."""
    g = torch.tensor([exp(-((i - size // 2) ** 2) / (2.0 * sigma * sigma)) for i in range(size)], dtype=torch.float32)
    return g / g.sum()


def ssim(x, y, size=11):
    """loss_utils.py:27-63 (size_average=True): depthwise filtering with the outer product of window_1d, zero padding;
    the window is built in float32 like the reference and then cast to the compute dtype."""
    ch = x.size(-3)
    w1 = window_1d(size).unsqueeze(1)
    w2 = (w1 @ w1.t()).float()[None, None].expand(ch, 1, size, size).contiguous().to(x.dtype)
    filt = lambda t: F.conv2d(t, w2, padding=size // 2, groups=ch)
    m1, m2 = filt(x), filt(y)
    v1 = filt(x * x) - m1 * m1
    v2 = filt(y * y) - m2 * m2
    v12 = filt(x * y) - m1 * m2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * m1 * m2 + c1) * (2 * v12 + c2)) / ((m1 * m1 + m2 * m2 + c1) * (v1 + v2 + c2))
    return smap.mean()


def pearson_corrcoef(preds, target):
    x, y = preds.reshape(-1), target.reshape(-1)
    mx, my = x.mean(), y.mean()
    n = x.numel()
    var_x = ((x - mx) ** 2).sum() / (n - 1)
    var_y = ((y - my) ** 2).sum() / (n - 1)
    corr_xy = ((x - mx) * (y - my)).sum() / (n - 1)
    return torch.clamp(corr_xy / (var_x * var_y).sqrt(), -1.0, 1.0)


def training_loss(image, gt_image, lambda_dssim, depth=None, pseudo_depth=None, depth_weight=0.1, dtype=torch.float64):
    """Returns dict(loss, Ll1, ssim, pseudo, dL_dimage, dL_ddepth) as numpy float64."""
    img = torch.tensor(np.asarray(image), dtype=dtype, requires_grad=True)
    gt = torch.tensor(np.asarray(gt_image), dtype=dtype)
    Ll1 = l1_loss(img, gt)
    s = ssim(img, gt)
    loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - s)
    d, pd = None, torch.zeros((), dtype=dtype)
    if depth is not None:
        d = torch.tensor(np.asarray(depth), dtype=dtype, requires_grad=True)
        m32 = torch.tensor(np.asarray(pseudo_depth), dtype=torch.float32)
        m = m32.to(dtype).reshape(-1, 1)
        inv = (1 / (m32 + 200.)).to(dtype).reshape(-1, 1)            # evaluated in float32 by the reference
        dd = d.reshape(-1, 1)
        a, b = 1 - pearson_corrcoef(-m, dd), 1 - pearson_corrcoef(inv, dd)
        pd = a if not (b < a) else b                                   # Python min(): first argument wins ties
        loss = loss + depth_weight * pd
    loss.backward()
    n = lambda t: t.detach().numpy().astype(np.float64)
    return dict(loss=float(loss.detach()), Ll1=float(Ll1.detach()), ssim=float(s.detach()), pseudo=float(pd.detach()), dL_dimage=n(img.grad),
                dL_ddepth=n(d.grad) if d is not None else None)


def densification_stats(radii, dL_dmean2D, max_radii2D, xyz_gradient_accum, denom):
    vis = radii > 0
    out_r, out_a, out_d = max_radii2D.copy(), xyz_gradient_accum.copy(), denom.copy()
    out_r[vis] = np.maximum(out_r[vis], radii[vis].astype(np.float32))
    out_a[vis] += np.sqrt((dL_dmean2D[vis, :2].astype(np.float32) ** 2).sum(axis=1, dtype=np.float32))
    out_d[vis] += 1
    return out_r, out_a, out_d
