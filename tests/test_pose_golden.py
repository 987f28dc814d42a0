"""CPU: everything that stands in for the reference's pose-refinement Python is pinned to the reference itself.

Fixtures: tests/golden/pose_loop_vectors.npz and ref_python_vectors.npz -- outputs of the reference's own pose_utils.py,
descent_utils.py, graphics_utils.py, camera_utils.py and torch.optim.Adam, recorded by tests/golden/make_pose_golden.py and
make_golden.py in the development container.  Checked against them here:
  * tests/replay.py (the driver the -m gpu tests and bench.py's python-loop leg use instead of the un-runnable scripts),
  * gs_localization_amd/scenes.py (numpy camera maths every test scene is built with),
  * the CPU oracle's pose gradient dL/dtau (float64 autograd through the reference's SE3_exp),
  * the CPU oracle inside the reference's loop body (the recorded 8-iteration refinement is reproduced).
The HIP kernels are compared with the same fixtures in tests/test_gpu_refine.py."""
import os
import types

import numpy as np
import pytest
import torch

from gs_localization_amd import scenes as S
from tests import replay as RP
from tests.util import rel_l1

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def pg():
    return np.load(os.path.join(HERE, "golden", "pose_loop_vectors.npz"))


def test_scene_camera_maths_against_reference(golden):
    for c, Pref in zip(golden["proj_intr"], golden["proj_P"]):
        fx, fy, cx, cy, W, H = c
        assert np.allclose(S.projection_matrix(0.01, 100.0, fx, fy, cx, cy, int(W), int(H)), Pref, rtol=1e-6, atol=1e-7)
    for tau, Tref in zip(golden["se3_tau"], golden["se3_T"]):
        assert np.allclose(S.se3_exp(tau), Tref, atol=1e-12)
        assert np.allclose(RP.se3_exp(torch.tensor(tau)).numpy(), Tref, atol=1e-12)


def test_replay_frame_matrices_against_reference_camera(pg):
    sc = S.small(P=8, W=640, H=480, fx=525.0)
    fr = RP.QueryFrame(0, torch.tensor(pg["cam_proj_raw_T"]), sc, "cpu")
    assert np.allclose(RP.intrinsics_projection(sc, "cpu").numpy(), pg["cam_proj_raw_T"], rtol=1e-6, atol=1e-7)
    fr.update_RT(torch.tensor(pg["cam_R"]), torch.tensor(pg["cam_T"]))
    assert torch.allclose(fr.world_view_transform, torch.tensor(pg["cam_view"]), atol=1e-7)
    assert torch.allclose(fr.full_proj_transform, torch.tensor(pg["cam_fullproj"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(fr.camera_center, torch.tensor(pg["cam_center"]), atol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mono", [0, 1])
def test_replay_tracking_loss_and_gradients_against_reference(pg, tag, mono):
    k = f"track_{tag}_"
    expo = pg[k + "exposure"]
    fr = types.SimpleNamespace(exposure_a=torch.tensor(expo[:1], requires_grad=True), exposure_b=torch.tensor(expo[1:], requires_grad=True),
                               original_image=torch.tensor(pg[k + "gt"]), depth=pg[k + "gt_depth"], grad_mask=torch.tensor(pg[k + "mask"]))
    im, dp = torch.tensor(pg[k + "image"], requires_grad=True), torch.tensor(pg[k + "depth"], requires_grad=True)
    cfg = {"Training": {"monocular": bool(mono), "alpha": 0.99, "opacity_threshold": 0.99}}
    loss = RP.tracking_loss(cfg, im, dp, torch.tensor(pg[k + "opacity"]), fr)
    loss.backward()
    r = f"track_{tag}_mono{mono}_"
    assert abs(loss.item() - float(pg[r + "loss"])) < 1e-7
    assert np.allclose(im.grad.numpy(), pg[r + "dimage"], rtol=1e-6, atol=1e-10)
    dd = dp.grad.numpy() if dp.grad is not None else np.zeros_like(pg[r + "ddepth"])
    assert np.allclose(dd, pg[r + "ddepth"], rtol=1e-6, atol=1e-10)
    assert np.allclose([fr.exposure_a.grad.item(), fr.exposure_b.grad.item()], pg[r + "dexposure"], rtol=1e-5, atol=1e-9)


def test_replay_adam_and_pose_update_against_reference_trajectory(pg):
    sc = S.small(P=8, W=640, H=480, fx=525.0)
    fr = RP.QueryFrame(0, torch.tensor(pg["cam_proj_raw_T"]), sc, "cpu")
    fr.update_RT(torch.tensor(pg["traj_R0"]), torch.tensor(pg["traj_T0"]))
    opt = RP.pose_adam(fr)
    for it, g in enumerate(pg["traj_grads"]):
        fr.cam_rot_delta.grad, fr.cam_trans_delta.grad = torch.tensor(g[0:3]), torch.tensor(g[3:6])
        fr.exposure_a.grad, fr.exposure_b.grad = torch.tensor(g[6:7]), torch.tensor(g[7:8])
        with torch.no_grad():
            opt.step()
            conv = bool(RP.apply_pose_delta(fr, float(pg["traj_threshold"])))
        assert conv == bool(pg["traj_converged"][it]), it
        assert torch.allclose(fr.R, torch.tensor(pg["traj_R"][it]), atol=1e-6), it
        assert torch.allclose(fr.T, torch.tensor(pg["traj_T"][it]), atol=1e-6), it
        assert np.allclose([fr.exposure_a.item(), fr.exposure_b.item()], pg["traj_exposure"][it], atol=1e-7)
        assert float(fr.cam_rot_delta.detach().abs().sum()) == 0 and float(fr.cam_trans_delta.detach().abs().sum()) == 0
    te, re = RP.pose_errors(np.eye(3), np.zeros(3), pg["traj_R0"], pg["traj_T0"])
    assert te > 0.5 and 20.0 < re < 25.0


@pytest.mark.parametrize("name", ["sh3", "offcentre_white", "partial_tiles"])
def test_oracle_pose_gradient_against_float64_reference_se3(pg, name):
    """SURVEY.md 8(c) fixture 8: the oracle's closed-form dL/dtau (what the HIP kernels are compared with at full size) vs
    float64 autograd through the reference's SE3_exp"""
    from oracle import oracle as O
    P, W, H, deg, seed = (int(x) for x in pg[f"tau_{name}_scene"])
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=seed)
    cx, cy, bg = pg[f"tau_{name}_cxcy_bg"]
    sc.cx, sc.cy = float(cx), float(cy)
    sc.bg[:] = bg
    f = O.forward_scene(sc, pg["tau_w2c"], want_n_touched=True)
    g = O.backward(f, pg[f"tau_{name}_gc"], pg[f"tau_{name}_gd"], np.zeros((1, H, W), np.float32), pose_mode=True)
    assert rel_l1(g["tau"], pg[f"tau_{name}_expected"]) <= 1e-5


def loop_mask(g, H, W):
    """the mask a recorded loop ran under: all pixels (pose_loop_vectors.npz, round 1) or the reference's own per-frame mask
    (masked_loop_vectors.npz, round 6: compute_grad_mask | create_mask(keypoints), recorded from the reference's functions)"""
    if "loop_mask_bits" not in g:
        return torch.ones((1, H, W), dtype=torch.bool)
    return torch.tensor(np.unpackbits(g["loop_mask_bits"])[:H * W].reshape(1, H, W).astype(bool))


@pytest.fixture(scope="module")
def pgm():
    return np.load(os.path.join(HERE, "golden", "masked_loop_vectors.npz"))


def test_recorded_masked_reference_loop_is_reproduced_by_the_replay_around_the_oracle(pgm):
    """the same under the mask every localiser of the reference refines under (tests/golden/make_masked_loop_golden.py)"""
    assert 0.3 < float(loop_mask(pgm, int(pgm["loop_scene"][2]), int(pgm["loop_scene"][1])).float().mean()) < 0.8
    test_recorded_reference_loop_is_reproduced_by_the_replay_around_the_oracle(pgm)


def test_recorded_mask_is_what_the_oracle_computes(pgm):
    """the recorded mask itself (Camera.compute_grad_mask | create_mask, run by the reference's code) against oracle/grad_mask_oracle.py"""
    from oracle import grad_mask_oracle as G
    P, W, H, deg, seed = (int(x) for x in pgm["loop_scene"])
    want = np.unpackbits(pgm["loop_mask_bits"])[:H * W].reshape(H, W).astype(bool)
    want0 = np.unpackbits(pgm["loop_mask_without_boxes_bits"])[:H * W].reshape(H, W).astype(bool)
    assert np.array_equal(G.compute_grad_mask(pgm["loop_gt_image"], 1.1), want0)
    assert np.array_equal(G.compute_grad_mask(pgm["loop_gt_image"], 1.1, pgm["loop_keypoints"], 10), want)


def test_recorded_reference_loop_is_reproduced_by_the_replay_around_the_oracle(pg):
    """SURVEY.md 8(c) fixture 9 on the CPU: replay.py's loss / Adam / pose update around the oracle's render and backward walk
    the same 8 poses the reference's own functions walked when the fixture was recorded."""
    from oracle import oracle as O
    P, W, H, deg, seed = (int(x) for x in pg["loop_scene"])
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=seed, scale_med=float(pg["loop_scale_med"]))
    fr = RP.QueryFrame(0, RP.intrinsics_projection(sc, "cpu"), sc, "cpu")
    fr.original_image, fr.depth = torch.tensor(pg["loop_gt_image"]), torch.tensor(pg["loop_gt_depth"])
    fr.grad_mask = loop_mask(pg, H, W)
    init = torch.tensor(pg["loop_init"])
    fr.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
    opt = RP.pose_adam(fr)
    for it in range(len(pg["loop_R"])):
        f = O.forward(sc.means3D, sc.opacities, fr.world_view_transform.numpy(), fr.full_proj_transform.numpy(), fr.camera_center.numpy(),
                      W, H, sc.tanfovx, sc.tanfovy, sc.bg, sh_degree=deg, shs=sc.shs, scales=sc.scales, rotations=sc.rotations,
                      want_n_touched=True)
        im, dp = torch.tensor(f.color, requires_grad=True), torch.tensor(f.depth, requires_grad=True)
        opt.zero_grad()
        loss = RP.tracking_loss(RP.TRACKING_CONFIG, im, dp, torch.tensor(f.alpha), fr)
        loss.backward()
        g = O.backward(f, im.grad.numpy(), dp.grad.numpy(), np.zeros((1, H, W), np.float32), pose_mode=True)
        assert rel_l1(g["tau"], pg["loop_tau"][it]) < 1e-4, it
        fr.cam_trans_delta.grad, fr.cam_rot_delta.grad = torch.tensor(g["tau"][:3].copy()), torch.tensor(g["tau"][3:].copy())
        with torch.no_grad():
            opt.step()
            RP.apply_pose_delta(fr)
        assert abs(loss.item() - float(pg["loop_loss"][it])) <= 1e-5 * float(pg["loop_loss"][it])
        assert torch.allclose(fr.R, torch.tensor(pg["loop_R"][it]), atol=2e-6), it
        assert torch.allclose(fr.T, torch.tensor(pg["loop_T"][it]), atol=2e-6), it
    # known answer: the refinement works its way down the tracking loss (8 of the 50 iterations: the pose is not there yet)
    assert float(pg["loop_loss"][-1]) < 0.8 * float(pg["loop_loss"][0])
