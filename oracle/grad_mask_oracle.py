"""TEST INFRASTRUCTURE (CPU oracle): the per-frame gradient mask every reference localiser refines under.

Restates, in numpy,
  gs_localization/pipelines/tools/camera_utils.py:164-193     Camera.compute_grad_mask (both branches)
  gs_localization/pipelines/tools/descent_utils.py:33-50      image_gradient   (reflect pad, Scharr 3x3, 1/32)
  gs_localization/pipelines/tools/descent_utils.py:53-67      image_gradient_mask (3x3 validity, |gray| > 0.01)
  gs_localization/pipelines/7scenes_localize_full_dslam.py:126-149,355-360   create_mask (k x k keypoint boxes) OR-ed in
Pinned by tests/golden/grad_mask_vectors.npz (tests/golden/make_grad_mask_golden.py runs the IMPORTED reference functions
on the CPU) -- bit for bit, the mask being boolean.  Only tests/, smoke() and bench.py's cpu_baseline leg may import this.

Arithmetic that the pin fixes (torch 2.10 CPU, the reference's image sizes):
  gray      = ((r + g) + b) / 3        in float32 (torch's CPU mean: sum, then a division; a CUDA run multiplies by fl(1/3))
  gradients = a chain of fused multiply-adds over the taps in row-major order, starting from 0, then x 1/32 (exact)
  intensity = sqrt(gv*gv + gh*gh)      every operation rounded to float32
  median    = torch.median: the LOWER median, sorted element (n - 1) // 2
  mask      = intensity > fl32(median * fl32(edge_threshold))
"""
import numpy as np

SCHARR_V = ((3, 10, 3), (0, 0, 0), (-3, -10, -3))      # conv_x of image_gradient: the "v" output (descent_utils.py:39-41,45-47)
SCHARR_H = ((3, 0, -3), (10, 0, -10), (3, 0, -3))      # conv_y: the "h" output (descent_utils.py:36-38,48-50)


def gray_of(image):
    """original_image.mean(dim=0, keepdim=True) (camera_utils.py:167), [3,H,W] float32 -> [H,W] float32"""
    im = np.asarray(image, np.float32)
    return ((im[0] + im[1]) + im[2]) / np.float32(3)


def _fma_conv(p, k, H, W):
    acc = np.zeros((H, W), np.float32)
    for i in range(3):
        for j in range(3):
            if k[i][j] == 0:
                continue
            # fl32(x * k + acc) with one rounding: exact in float64 for these operands (24-bit x small integer + 24-bit)
            acc = (p[i:i + H, j:j + W].astype(np.float64) * np.float64(k[i][j]) + acc.astype(np.float64)).astype(np.float32)
    return acc * np.float32(1.0 / 32.0)


def image_gradient(gray):
    """descent_utils.py:33-50 on a [H,W] image -> (grad_v, grad_h)"""
    H, W = gray.shape
    p = np.pad(gray, 1, mode="reflect")
    return _fma_conv(p, SCHARR_V, H, W), _fma_conv(p, SCHARR_H, H, W)


def image_gradient_mask(gray, eps=0.01):
    """descent_utils.py:53-67: True where all nine pixels of the reflect-padded 3x3 neighbourhood have |gray| > eps
    (the two outputs of the reference function are the same array)"""
    H, W = gray.shape
    ok = np.abs(np.pad(gray, 1, mode="reflect")) > np.float32(eps)
    out = np.ones((H, W), bool)
    for i in range(3):
        for j in range(3):
            out &= ok[i:i + H, j:j + W]
    return out


def grad_intensity(image):
    """camera_utils.py:167-172 -> [H,W] float32"""
    gray = gray_of(image)
    gv, gh = image_gradient(gray)
    ok = image_gradient_mask(gray).astype(np.float32)
    gv = gv * ok
    gh = gh * ok
    return np.sqrt((gv * gv + gh * gh).astype(np.float32)).astype(np.float32)


def lower_median(x):
    x = np.sort(np.asarray(x, np.float32).reshape(-1))
    return x[(x.size - 1) // 2]


def create_mask(keypoints, width, height, k):
    """7scenes_localize_full_dslam.py:126-149: a box of 2 (k // 2) + 1 pixels around int(x), int(y) of every keypoint"""
    m = np.zeros((height, width), bool)
    h = k // 2
    for pt in np.asarray(keypoints, np.float32).reshape(-1, 2):
        x, y = int(pt[0]), int(pt[1])
        m[max(0, y - h):min(height, y + h + 1), max(0, x - h):min(width, x + h + 1)] = True
    return m


def compute_grad_mask(image, edge_threshold, keypoints=None, box_k=10):
    """camera_utils.py:189-193 (every dataset type but "replica") [+ the keypoint boxes] -> [H,W] bool"""
    inten = grad_intensity(image)
    thr = np.float32(lower_median(inten) * np.float32(edge_threshold))
    m = inten > thr
    if keypoints is not None and len(keypoints):
        m = m | create_mask(keypoints, image.shape[2], image.shape[1], box_k)
    return m


def compute_grad_mask_replica(image, edge_threshold, rows=32, cols=32):
    """camera_utils.py:174-188 (config["Dataset"]["type"] == "replica") -> [H,W] float32.  Quirks kept: the two in-place writes
    happen in sequence, so the ones written first are cleared again whenever 1 <= median x multiplier; pixels outside the
    rows x cols grid of int(h / rows) x int(w / cols) blocks keep their raw intensity."""
    inten = grad_intensity(image).copy()
    H, W = inten.shape
    bh, bw = int(H / rows), int(W / cols)
    for r in range(rows):
        for c in range(cols):
            blk = inten[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw]
            if blk.size == 0:
                continue
            t = np.float32(lower_median(blk) * np.float32(edge_threshold))
            blk[blk > t] = 1
            blk[blk <= t] = 0
    return inten
