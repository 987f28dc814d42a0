"""-m gpu : heavy tiles split across workgroups (gsr_kernels.h, SegCtl; round 5).

On a structured scene a few tiles need lists ten times the mean; the native loop's speculative iterations cut such a tile's list into
depth ranges that as many workgroups walk in parallel, in both compositing kernels.  Against the unsplit walk the results differ by
rounding only (transmittance carried as a product of per-range products), so:
  * the split loop against the same loop with GSR_REFINE_NO_SPLIT: same poses (2e-6), same images / n_touched / radii, same gradient
    tensors, on scenes where tiles really are split (gsr_debug_seg_stats says how many);
  * the split path against the CPU oracle directly, at full size: tests/test_gpu_lean.py::
    test_headline_path_against_the_oracle_at_the_pose_of_its_last_forward on S-room-640 / S-1M-640-object (asserts there that tiles
    were split in the iteration whose images and gradients it compares);
  * the deterministic option never splits (its promise is the same bits whatever the lists looked like)."""
import numpy as np
import pytest
import torch

from gs_localization_amd import _lib, scenes as S
from tests import util as U
from tests.test_gpu_lean import _setup, _run, _same_path

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _room(P):
    return S.s_room_640(P=P)


def _object(P):
    return S.s_1m_640_object(P=P)


@pytest.mark.parametrize("make,P", [(_room, 300_000), (_object, 400_000)], ids=["room-300k", "object-400k"])
def test_split_loop_equals_the_unsplit_loop_up_to_rounding(make, P):
    from tests import replay as PL
    sc = make(P)
    model, bg, view, init = _setup(sc, seed=7)
    K = 10
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    split = _run(fr, view(), init, bg, K, flags=0, lean_min_P=1)
    blocks, ntiles_split, kmax, budget = fr.seg_stats()
    assert budget == 2 * 1200 and ntiles_split >= 5 and kmax >= 3 and blocks >= 1200 + 2 * ntiles_split, (blocks, ntiles_split, kmax, budget)
    g_split = {k: getattr(fr, "g_" + k).detach().clone() for k in ("m3d", "sh", "opac", "scale", "rot", "tau")}
    fr2 = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    plain = _run(fr2, view(), init, bg, K, flags=_lib.REFINE_NO_SPLIT, lean_min_P=1)
    assert split["info"]["fallbacks"] <= plain["info"]["fallbacks"] + 2, (split["info"], plain["info"])
    assert torch.allclose(split["R"], plain["R"], atol=2e-6) and torch.allclose(split["T"], plain["T"], atol=2e-6)
    # images: as a whole to 2e-5, 99.9 % of the pixels within 5e-4 -- and a handful may differ by up to a per cent: a pixel terminates where
    # T (1 - alpha) < 1e-4, the split walk carries T as a product of per-range products, and where the two roundings fall on different
    # sides of the threshold one walk blends a last splat of weight up to alpha T ~ 1e-2 that the other does not (the same flip
    # separates any two fp32 evaluation orders; the oracle comparison below holds the split walk to the same per-pixel bars as the unsplit)
    for k, scale in (("color", 1.0), ("alpha", 1.0), ("depth", 10.0)):
        d = (split[k] - plain[k]).abs()
        assert float(d.sum() / plain[k].abs().sum().clamp_min(1e-30)) <= 2e-5, k
        assert float(torch.quantile(d.flatten().float(), 0.999)) <= scale * 5e-4, k
        assert int((d > scale * 5e-3).sum().item()) <= 8 and float(d.max()) <= scale * 5e-2, (k, int((d > scale * 5e-3).sum().item()), float(d.max()))
    nt = int(plain["n_touched"].sum().item())
    assert int((split["n_touched"] - plain["n_touched"]).abs().sum().item()) <= max(2, int(1e-4 * nt))
    assert int((split["radii"] != plain["radii"]).sum().item()) <= max(2, int(5e-5 * split["radii"].numel()))
    for k in g_split:
        a, b = g_split[k].cpu().numpy(), getattr(fr2, "g_" + k).detach().cpu().numpy()
        # (two runs of the loop end ~1e-7 apart in pose, which moves the gradients of a loss made of sign functions by a few 1e-5:
        # the bar of tests/test_gpu_lean.py's loop-against-loop comparisons; the oracle comparison at 2e-5 is the direct test's)
        assert U.rel_l1(a, b) <= 2e-4, (k, U.rel_l1(a, b))


def test_split_forward_and_backward_against_the_oracle_on_a_mid_size_room():
    """One speculative iteration with split tiles, checked against the CPU oracle at the pose it ran with: the second forward of a
    two-iteration call (the first bins completely and measures the tiles' work; the second is speculative, split, and the call's last,
    so its images, n_touched and gradient tensors are what the call returns)."""
    import os
    from oracle import oracle as O
    from tests import replay as PL
    from tests.test_gpu_lean import _camera_of_the_pose_state
    O.set_threads(min(64, os.cpu_count() or 1))
    sc = _room(300_000)
    model, bg, view, init = _setup(sc, seed=9)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    vp = view()
    gt_image, gt_depth = vp.original_image.clone(), vp.depth.clone()
    run = _run(fr, vp, init, bg, 2, flags=0, lean_min_P=1)
    blocks, ntiles_split, kmax, budget = fr.seg_stats()
    assert ntiles_split >= 5 and kmax >= 3, (blocks, ntiles_split, kmax)
    info = run["info"]
    assert info["iters"] == 2 and info["fallbacks"] <= 1, {k: info[k] for k in ("iters", "fallbacks")}      # (a failed speculation's retry runs the split list too)
    vm, pm, cp = _camera_of_the_pose_state(info["R_last_forward_host"], info["T_last_forward_host"], S.camera_matrices(sc)[2])
    f = O.forward(sc.means3D, sc.opacities, vm, pm, cp, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=sc.sh_degree, shs=sc.shs,
                  scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
    assert np.array_equal(run["radii"].cpu().numpy(), f.radii)
    for k, b in (("color", f.color), ("depth", f.depth), ("alpha", f.alpha)):
        assert U.rel_l1(run[k].cpu().numpy(), b) <= 1e-4, (k, U.rel_l1(run[k].cpu().numpy(), b))
    nt = run["n_touched"].cpu().numpy()
    assert np.abs(nt - f.n_touched).sum() <= max(2, 1e-4 * f.n_touched.sum()), int(np.abs(nt - f.n_touched).sum())
    ex = info["exposure_last_forward_host"]

    class _V:
        pass
    v = _V()
    v.exposure_a, v.exposure_b = torch.tensor([float(ex[0])], device=DEV), torch.tensor([float(ex[1])], device=DEV)
    v.original_image, v.depth, v.grad_mask = gt_image, gt_depth, torch.ones((1, sc.H, sc.W), dtype=torch.bool, device=DEV)
    ti, td = run["color"].clone().requires_grad_(True), run["depth"].clone().requires_grad_(True)
    PL.tracking_loss(PL.TRACKING_CONFIG, ti, td, run["alpha"], v).backward()
    go = O.backward(f, ti.grad.cpu().numpy(), td.grad.cpu().numpy(), np.zeros((1, sc.H, sc.W), np.float32), pose_mode=True)
    assert U.rel_l1(fr.g_tau.cpu().numpy(), go["tau"]) <= 1e-5, U.rel_l1(fr.g_tau.cpu().numpy(), go["tau"])
    for k, ok in (("m3d", "means3D"), ("sh", "sh"), ("opac", "opacities"), ("scale", "scales"), ("rot", "rotations")):
        a, b = getattr(fr, "g_" + k).cpu().numpy(), go[ok]
        assert U.rel_l1(a.reshape(b.shape), b) <= 2e-5, (k, U.rel_l1(a.reshape(b.shape), b))
        worst, share, at = U.row_errors(a.reshape(b.shape), b)
        assert worst <= 0.5 and share <= 1e-3, (k, worst, share, at)      # (bars of the full-size S-room-640 comparison, tests/test_gpu_lean.py)


def test_the_deterministic_option_never_splits():
    from tests import replay as PL
    sc = _room(300_000)
    model, bg, view, init = _setup(sc, seed=7)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    a = _run(fr, view(), init, bg, 6, flags=_lib.REFINE_DETERMINISTIC, lean_min_P=1)
    b = _run(fr, view(), init, bg, 6, flags=_lib.REFINE_DETERMINISTIC, lean_min_P=1, speculative=False)
    assert torch.equal(a["R"], b["R"]) and torch.equal(a["T"], b["T"])
    assert torch.equal(a["color"], b["color"]) and torch.equal(a["depth"], b["depth"]) and torch.equal(a["n_touched"], b["n_touched"])
