"""Generates tests/golden/grad_mask_vectors.npz by EXECUTING the reference's own mask code on the CPU
(development container only: needs /root/reference).  The .npz holds inputs and expected outputs -- data, no source.

Executed reference code:
  gs_localization/pipelines/tools/descent_utils.py:33-67      image_gradient, image_gradient_mask   (imported)
  gs_localization/pipelines/tools/camera_utils.py:164-193     Camera.compute_grad_mask, both branches (imported)
  gs_localization/pipelines/7scenes_localize_full_dslam.py:126-149   create_mask -- that script cannot be imported (its
      module body opens D:/ paths and needs h5py / cv2), so the function's definition is taken out of the parsed file with
      `ast` and compiled as it stands: the reference's code runs, none of it is written down here
  ... :355-360   grad_mask | create_mask(keypoints, k=10)    (the call sequence, restated below)

Images are synthetic uint8 pictures turned into tensors the way the reference's datasets do (image / 255.0 -> float32,
7scenes_localize_full_dslam.py:228-232): smooth shading, hard-edged shapes, a textured patch, sensor noise, an unobserved
(black) border region that exercises the validity mask, and one picture that is more than half black (median 0).
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
PL = "gs_localization/pipelines/tools/"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "grad_mask_vectors.npz")


def load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_modules():
    # descent_utils builds its filters with device="cuda": run them on the CPU
    _tensor, _ones = torch.tensor, torch.ones
    torch.tensor = lambda *a, **k: _tensor(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    torch.ones = lambda *a, **k: _ones(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    sys.modules.setdefault("tools", types.ModuleType("tools"))
    load(PL + "graphics_utils.py", "tools.graphics_utils")
    desc = load(PL + "descent_utils.py", "tools.descent_utils")
    cam = load(PL + "camera_utils.py", "tools.camera_utils")
    return desc, cam


def reference_function(script, name):
    """the FunctionDef `name` of a reference script, compiled from the parsed file (the rest of the script is not executed)"""
    path = os.path.join(REF, script)
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(fn) == 1
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    return ns[name]


def picture(rng, H, W, noise=1.5, black_border=True, mostly_black=False):
    """uint8 [H, W, 3]"""
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    img = np.zeros((H, W, 3))
    for c in range(3):          # smooth shading
        img[..., c] = 120 + 60 * np.sin(x / W * rng.uniform(2, 7) + rng.uniform(0, 6)) * np.cos(y / H * rng.uniform(2, 7) + rng.uniform(0, 6))
    for _ in range(14):         # hard-edged rectangles and discs
        col = rng.uniform(10, 245, 3)
        if rng.uniform() < 0.5:
            x0, y0 = rng.uniform(0, W), rng.uniform(0, H)
            m = (np.abs(x - x0) < rng.uniform(10, W / 5)) & (np.abs(y - y0) < rng.uniform(10, H / 5))
        else:
            m = (x - rng.uniform(0, W)) ** 2 + (y - rng.uniform(0, H)) ** 2 < rng.uniform(8, H / 4) ** 2
        img[m] = col
    ty, tx = int(H * 0.55), int(W * 0.1)          # a textured patch
    img[ty:ty + H // 4, tx:tx + W // 4] += rng.normal(0, 25, (H // 4, W // 4, 1))
    img[int(H * 0.1):int(H * 0.2), int(W * 0.6):int(W * 0.9)] = 255.0          # a saturated, perfectly flat region
    img += rng.normal(0, noise, img.shape)
    if black_border:            # unobserved pixels (the rectified 7-Scenes / undistorted Cambridge frames have them)
        img[:, :int(W * 0.04)] = 0
        img[int(H * 0.93):] = 0
        img[(x - W * 0.8) ** 2 + (y - H * 0.7) ** 2 < (H * 0.08) ** 2] = 1.0          # |gray| = 1/255 < 0.01: invalid as well
    if mostly_black:
        img[(x > W * 0.3) | (y > H * 0.8)] = 0
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def keypoints(rng, n, W, H):
    k = np.stack([rng.uniform(0, W - 1, n), rng.uniform(0, H - 1, n)], 1).astype(np.float32)
    k[:6] = [[0.2, 0.9], [W - 1, H - 1], [W - 1.5, 0.0], [3.99, H - 2.01], [W / 2, H / 2], [5.0, 5.0]]      # corners, borders, truncation
    return k


def main():
    desc, camu = reference_modules()
    create_mask = reference_function("gs_localization/pipelines/7scenes_localize_full_dslam.py", "create_mask")
    rng = np.random.default_rng(20261004)
    cases = [  # name, H, W, edge_threshold, #keypoints, picture kwargs  (sizes: 7-Scenes, the Cambridge scripts' 1024x576, LLFF fern)
        ("scenes640", 480, 640, 1.1, 700, {}),
        ("cambridge1024", 576, 1024, 1.0, 1500, dict(noise=0.0)),
        ("fern504", 378, 504, 1.1, 300, dict(black_border=False)),
        ("dark640", 480, 640, 1.1, 40, dict(mostly_black=True, noise=0.8)),
    ]
    out = {"cases": np.array([c[0] for c in cases])}
    for name, H, W, thr, nk, kw in cases:
        u8 = picture(rng, H, W, **kw)
        image = torch.from_numpy(u8 / 255.0).clamp(0.0, 1.0).permute(2, 0, 1).to(dtype=torch.float32)
        cam = camu.Camera(0, image, None, torch.eye(4), torch.eye(4), 1.0, 1.0, 0.0, 0.0, 1.0, 1.0, H, W, device="cpu")
        cfg = {"Training": {"edge_threshold": thr}, "Dataset": {"type": "tum"}}
        cam.compute_grad_mask(cfg)
        gm = cam.grad_mask
        assert gm.dtype == torch.bool and tuple(gm.shape) == (1, H, W)
        kp = keypoints(rng, nk, W, H)
        boxes = create_mask(mkpts_lst=kp, width=W, height=H, k=10)
        both = gm | torch.tensor(boxes)
        # intermediate values, for diagnosis when a mask bit differs: the intensity (subsampled) and its median
        gray = image.mean(dim=0, keepdim=True)
        gv, gh = desc.image_gradient(gray)
        mv, mh = desc.image_gradient_mask(gray)
        inten = torch.sqrt((gv * mv) ** 2 + (gh * mh) ** 2)
        cam.compute_grad_mask({"Training": {"edge_threshold": thr}, "Dataset": {"type": "replica"}})
        rep = cam.grad_mask
        assert rep.dtype == torch.float32
        out[name + "_image_u8"] = u8
        out[name + "_edge_threshold"] = np.float64(thr)
        out[name + "_keypoints"] = kp
        out[name + "_mask_bits"] = np.packbits(gm.numpy()[0])
        out[name + "_mask_with_boxes_bits"] = np.packbits(both.numpy()[0])
        out[name + "_median"] = np.float32(inten.median().item())
        out[name + "_intensity_sub"] = inten.numpy()[0, ::5, ::5].copy()
        # replica branch: zeros and ones inside the 32 x 32 grid of blocks, raw intensities outside it
        r = rep.numpy()[0]
        bh, bw = int(H / 32), int(W / 32)
        inside = r[:32 * bh, :32 * bw]
        assert np.isin(inside, (0.0, 1.0)).all()
        out[name + "_replica_inside_bits"] = np.packbits(inside > 0.5)
        out[name + "_replica_right"] = r[:, 32 * bw:].copy()
        out[name + "_replica_bottom"] = r[32 * bh:, :].copy()
        print(name, "mask share", float(gm.float().mean()), "with boxes", float(both.float().mean()), "median", float(inten.median()),
              "replica ones", float((inside > 0.5).mean()))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
