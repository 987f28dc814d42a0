"""-m gpu : the reference's pipeline chained once, shortened -- train (300 steps of train.py's loop with densification) -> point_cloud.ply
-> GaussianMap.from_ply -> per frame gsr_grad_mask + FusedRefiner.refine with the early exit, query frames rendered from the WORLD
(tests/trained_map.py; the full-length run is tools/trained_map.py, numbers in DESIGN.md).  What no other test has: a map whose
anisotropy, opacities and order are what training left, a residual that is real (the map is not the world), and the direct-oracle
parity check on THAT map."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_save_load_localise_and_check_against_the_oracle(tmp_path):
    from oracle import oracle as O
    from tests import trained_map as TM
    from tests.test_gpu_lean import oracle_check_at_the_last_forward, _run
    path = str(tmp_path / "point_cloud" / "iteration_300" / "point_cloud.ply")
    world, train = TM.train_room_map(path, steps=300, world_P=120_000, P0=30_000, P1=60_000, sh_degree=3, n_views=12)
    assert train["P_last"] > 1.5 * train["P_first"] and train["loss_first_last"][1] < 0.7 * train["loss_first_last"][0], train
    assert train["psnr_view0_db"] > 15.0, train
    rep, (gmap, fr, frames, inits, bg) = TM.localise_against(path, world, n_frames=8, in_flight=4, start=(0.03, 2.0))
    assert rep["map_gaussians"] == train["P_last"]
    # started 3 cm / 2 deg off; a 300-step map is a rough one -- the refinement must still pull every frame in
    assert rep["pose_err_cm_median"] < 0.6 * 3.0 and rep["pose_err_deg_median"] < 0.6 * 2.0, rep
    assert 0.35 < rep["mask_share"] < 0.8, rep
    print("trained map:", {k: rep[k] for k in ("pose_err_cm_median", "pose_err_deg_median", "iterations_used_median", "single_frame_iters_per_s", "in_flight_iters_per_s", "mask_share")}, train)
    # direct-oracle parity on the trained map, at the pose of the call's last forward, under the frame's own mask
    O.set_threads(min(64, os.cpu_count() or 1))
    sc = TM.scene_of_map(gmap, world)
    run = _run(fr, frames[0], inits[0], bg, 8, flags=0, lean_min_P=1)
    print(*oracle_check_at_the_last_forward(sc, fr, run, frames[0], frames[0].original_image, frames[0].depth))
    # ... and the unchanged-scripts route on the same map: the reference-style Python loop on the drop-in pose package (autograd, torch's Adam,
    # update_pose; tests/replay.py) walks the same poses as the native loop
    def fresh(f):
        for t_ in (frames[f].exposure_a, frames[f].exposure_b, frames[f].cam_rot_delta, frames[f].cam_trans_delta):
            t_.data = torch.zeros_like(t_.data)
        return frames[f]
    from tests import replay as RP
    from tests import util as U
    # one iteration: the same dL/dtau (1e-5) and the same pose step; six iterations: the same trajectory up to what Adam makes of the
    # last digits of a gradient that a tenth of the pixels carries (a rough map is rarely opaque enough for `opacity > 0.99`; the map
    # itself differs from run to run -- training adds with fp32 atomics): seen 1e-6 ... 4e-5 of poses that move 1e-3 per step
    R1, T1, _ = fr.refine(fresh(1), RP.TRACKING_CONFIG, inits[1][:3, :3].clone(), inits[1][:3, 3].clone(), bg, iters=1, stop_on_converged=False, warm_start=False)
    R1, T1, tau_native = R1.clone(), T1.clone(), fr.g_tau.clone()
    f1 = fresh(1)
    f1.update_RT(inits[1][:3, :3].clone(), inits[1][:3, 3].clone())
    RP.loop_iteration(f1, RP.TRACKING_CONFIG, gmap, bg, RP.pose_adam(f1))
    tau_python = torch.cat([f1.cam_trans_delta.grad, f1.cam_rot_delta.grad])
    assert U.rel_l1(tau_native.cpu().numpy(), tau_python.cpu().numpy()) <= 1e-5, U.rel_l1(tau_native.cpu().numpy(), tau_python.cpu().numpy())
    assert torch.allclose(R1, f1.R, atol=2e-6) and torch.allclose(T1, f1.T, atol=2e-6)
    Rn, Tn, _ = fr.refine(fresh(1), RP.TRACKING_CONFIG, inits[1][:3, :3].clone(), inits[1][:3, 3].clone(), bg, iters=6, stop_on_converged=False, warm_start=False)
    Rn, Tn = Rn.clone(), Tn.clone()
    Rp, Tp, _ = RP.python_loop(fresh(1), RP.TRACKING_CONFIG, inits[1][:3, :3].clone(), inits[1][:3, 3].clone(), gmap, bg, iters=6)
    assert torch.allclose(Rn, Rp, atol=2e-4) and torch.allclose(Tn, Tp, atol=2e-4), (float((Rn - Rp).abs().max()), float((Tn - Tp).abs().max()))
