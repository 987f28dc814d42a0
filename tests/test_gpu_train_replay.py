"""-m gpu : BASELINE.json config 4 as a loop (S-train-garden): train.py's step sequence through package
(A) and the fused loss epilogue while the number of Gaussians grows 0.2 M -> 1.5 M on the densification cadence
(tests/train_replay.py).  At three checkpoints of P -- the first step, the middle of the schedule, and 1.5 M -- the step's
forward and backward are compared with the CPU oracle on exactly the tensors and pixel gradients the step used, and the
workspaces / speculation state of the drop-in package are checked to have survived the change of P."""
import os

import numpy as np
import pytest
import torch

from tests import util as U

pytestmark = pytest.mark.gpu

# fixed per-tensor bounds (rel-L1) of the training step's gradients against the oracle (see _check_step_against_oracle)
TRAIN_GRAD_TOL = {"means3D": 1e-4, "means2D": 1e-4, "opacities": 1e-4, "sh": 1e-4, "scales": 1e-4, "rotations": 1e-4}


def _check_step_against_oracle(tr, it):
    from oracle import oracle as O
    tr.step(it, keep=True)
    L = tr.last
    sc = tr.as_scene(L["act"])
    vw = L["view"]
    cam = U.scene_inputs(sc, vw["w2c"])
    gc, gd, ga = (g.cpu().numpy() for g in L["pix_grads"])
    assert np.abs(gd).sum() > 0 and np.abs(gc).sum() > 0           # grad_depth != 0: the Pearson term is live
    # Tolerance.  The reference's backward rebuilds every pixel's transmittance from the saved opacity image,
    # T_final = 1 - alpha (backward.cu:445), and 95 % of these pixels are saturated (alpha > 0.999): one ulp of alpha is 1e-4 of
    # T_final and scales every weight of that pixel.  With the SSIM gradient changing sign from pixel to pixel the per-Gaussian
    # sums cancel: the algorithm's own answer moves by 4e-4 ... 2e-3 when its alpha image is nudged by +-1 ulp (printed below as
    # `conditioning`, oracle against oracle).  Two correct fp32 forwards (v_exp_f32 here, expf there) differ by such ulps, so
    # the backwards cannot agree to the 2e-5 of the localisation configs; measured: 1e-5 ... 4e-5 at the three sizes.  The bound
    # is FIXED, 1e-4 per tensor -- a quarter of the smallest conditioning figure; a wrong list order or a dropped instance shows
    # as 1e-2.  (Handing the oracle's backward the HIP forward's opacity image instead was tried: its own n_contrib then no
    # longer matches that image on the pixels where the T < 1e-4 test flips, and the error grows to 4e-4.)  Sums in double in
    # the oracle, as in test_large_images: its fp32 atomics' order noise, 3e-6 here, stays out of the comparison.
    O.set_accumulate_double(True)
    try:
        f, go = U.oracle_run(sc, cam, (gc, gd, ga), pose=False)
        a_own = f.alpha
        up = np.random.default_rng(0).uniform(size=a_own.shape) < 0.5
        f.alpha = np.where(up, np.nextafter(a_own, np.float32(2)), np.nextafter(a_own, np.float32(0))).astype(np.float32)
        g_nudged = O.backward(f, gc, gd, ga, pose_mode=False)
        f.alpha = a_own
    finally:
        O.set_accumulate_double(False)
    assert np.array_equal(L["radii"].cpu().numpy(), f.radii)
    for k, ref in (("image", f.color), ("depth", f.depth), ("alpha", f.alpha)):
        e = U.rel_l1(L[k].detach().cpu().numpy(), ref)
        assert e <= 1e-4, (it, tr.P, k, e)
    errs, cond = {}, {}
    for k in ("means3D", "means2D", "opacities", "sh", "scales", "rotations"):
        got = L["grads"][k].cpu().numpy()
        errs[k] = U.rel_l1(got.reshape(go[k].shape), go[k])
        cond[k] = U.rel_l1(g_nudged[k], go[k])
    print("train replay at P =", tr.P, "gradient errors", {k: float("%.1e" % v) for k, v in errs.items()},
          "conditioning (oracle, alpha +-1 ulp)", {k: float("%.1e" % v) for k, v in cond.items()})
    for k, e in errs.items():
        assert e <= TRAIN_GRAD_TOL[k], (it, tr.P, k, e)
    return sc.P


def test_train_loop_with_growing_model_matches_the_oracle_at_three_sizes():
    from oracle import oracle as O
    from gs_localization_amd import rasterizer as RZ
    from tests.train_replay import TrainReplay
    O.set_threads(min(64, os.cpu_count() or 1))
    RZ._spec_cache.clear()
    # the cadence of train.py:147-149 compressed: P changes at iterations 2, 4, 6, 8, 10 (x1.5 each time, 0.2 M -> 1.5 M)
    tr = TrainReplay(P0=200_000, P1=1_500_000, densify_from=1, densification_interval=2, densify_until=12)
    sizes, checked = [], []
    losses = []
    for it in range(1, 12):
        if it in (1, 7, 11):
            checked.append(_check_step_against_oracle(tr, it))
            losses.append(tr.last["loss"])
        else:
            losses.append(float(tr.step(it)))
        sizes.append(tr.P)
        assert tr.max_radii2D.shape[0] == tr.P and tr.xyz_gradient_accum.shape[0] == tr.P
    assert checked[0] == 200_000 and 600_000 < checked[1] < 800_000 and checked[2] == 1_500_000
    assert sizes[0] == 200_000 and sizes[-1] == 1_500_000 and sorted(sizes) == sizes and len(set(sizes)) == 6
    assert all(np.isfinite(losses))
    # consumers of the densification statistics saw this step's visible Gaussians (train.py:142-145)
    assert float(tr.denom.sum()) > 0 and float(tr.max_radii2D.max()) > 0
    # package (A) does not speculate by default (train.py: Adam moves every opacity between two visits of a view, most guesses would
    # miss -- measured, rasterizer.speculation_enabled): plain forwards, reproducible bit for bit.  Switched on explicitly it keeps
    # one state per CAMERA (the storage address of the camera's view matrix tensor); a view's second render is then speculative and
    # bit-identical to its first; a state built before P changed is still exact afterwards.
    assert RZ.speculation_counters() == (0, 0)
    with torch.no_grad():
        a = tr.render(tr.views[3])
        b = tr.render(tr.views[3])
        assert RZ.speculation_counters() == (0, 0)
        os.environ["GSR_SPECULATION"] = "1"
        try:
            c = tr.render(tr.views[3])
            assert RZ.speculation_counters() == (0, 0)          # first render of this camera with a state: complete lists, records bounds
            d = tr.render(tr.views[3])
            assert RZ.speculation_counters() == (1, 0)
            e = tr.render(tr.views[5])                   # another camera: its own state, seeded with view 3's bounds (a guess like any other)
            e2 = tr.render(tr.views[5])
            v1, m1 = RZ.speculation_counters()
        finally:
            os.environ.pop("GSR_SPECULATION", None)
    assert torch.equal(e["image"], e2["image"]) and torch.equal(e["radii"], e2["radii"])
    for k in ("image", "depth", "alpha", "radii"):
        for other in (b, c, d):
            assert torch.equal(a[k], other[k]), k
    assert v1 >= 2 and v1 + m1 >= 3


def test_train_loop_through_many_changes_of_P_keeps_its_memory_and_its_parity():
    """BASELINE.json config 4 as written is 7 000 iterations with P changing 65 times and an opacity reset (train.py:142-152); the full
    run is tools/train_7k.py (HISTORY.md has its numbers).  Here the same loop compressed to 330 steps at 648x420: 24 changes of P
    (0.1 M -> 0.5 M), the opacity reset in the middle, then 40 steps at constant P during which NOTHING may grow -- workspaces are
    regrown when P changes and a leak would show as a rising peak --, finite losses throughout, and the oracle's forward / backward on
    the step after the last change of P."""
    from oracle import oracle as O
    from gs_localization_amd import rasterizer as RZ
    from tests.train_replay import TrainReplay
    O.set_threads(min(64, os.cpu_count() or 1))
    RZ._spec_cache.clear()
    tr = TrainReplay(P0=100_000, P1=500_000, W=648, H=420, densify_from=20, densification_interval=11, densify_until=290, opacity_reset_interval=150)
    losses, changes, last_change = [], 0, 0
    for it in range(1, 291):
        before = tr.P
        losses.append(float(tr.step(it)))
        if tr.P != before:
            changes += 1
            last_change = it
    assert changes >= 20 and tr.P == 500_000 and last_change >= 270, (changes, tr.P, last_change)
    assert all(np.isfinite(losses))
    assert float(torch.sigmoid(tr.par["opacity"].detach()).max()) > 0.011          # (the reset at 150 capped every opacity at 0.01; Adam has moved them since)
    _check_step_against_oracle(tr, 291)
    for it in range(292, 300):
        tr.step(it)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base_alloc = torch.cuda.memory_allocated()
    for it in range(300, 331):
        losses.append(float(tr.step(it)))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses))
    # (a step allocates its temporaries -- images, workspaces, gradients -- and frees them: the peak over thirty steps stays within one
    # step's worth above the resting level, and the resting level itself does not move)
    assert torch.cuda.memory_allocated() <= base_alloc + (8 << 20), (torch.cuda.memory_allocated(), base_alloc)
    peak1 = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    for it in range(331, 341):
        tr.step(it)
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() <= peak1 + (8 << 20), (torch.cuda.max_memory_allocated(), peak1)
