"""-m gpu : the fused pose-refinement epilogue.

Pinned to the reference's own Python through committed fixtures (tests/golden/pose_loop_vectors.npz, produced by
tests/golden/make_pose_golden.py which imports pose_utils.py / descent_utils.py / camera_utils.py and torch.optim.Adam):
tracking loss + gradients, Adam + update_pose trajectories, camera matrices, float64 dL/dtau, and an 8-iteration refinement.
The remaining tests compare the native loop with the reference-style Python loop of tests/replay.py on the same rasterizer."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from tests.test_pose_golden import loop_mask

from gs_localization_amd import scenes as S
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return None if t is None else t.data_ptr()


@pytest.fixture(scope="module")
def pose_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pose_loop_vectors.npz"))


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("mono", [0, 1])
def test_tracking_loss_kernel_matches_reference_fixture(pose_golden, tag, mono):
    """k_tracking_loss against get_loss_tracking + torch autograd of the reference (descent_utils.py:85-123)"""
    from gs_localization_amd import _lib
    lib, g = _lib.load(), pose_golden
    k = f"track_{tag}_"
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=DEV)
    image, depth, opacity, gt, gt_depth = (t(g[k + n]) for n in ("image", "depth", "opacity", "gt", "gt_depth"))
    H, W = gt_depth.shape
    m8 = t(g[k + "mask"].reshape(H, W).astype(np.uint8))
    expo = t(g[k + "exposure"])
    gi, gd, ga = torch.empty_like(image), torch.empty(1, H, W, device=DEV), torch.empty(1, H, W, device=DEV)
    out = torch.empty(4, device=DEV)
    _lib.check(lib.gsr_tracking_loss(W, H, _p(image), _p(depth), _p(opacity), _p(gt), _p(gt_depth), _p(m8), _p(expo), 0.99, 0.01,
                                     int(mono), _p(gi), _p(gd), _p(ga), _p(out), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    r = f"track_{tag}_mono{mono}_"
    loss = float(g[r + "loss"])
    assert abs(out[0].item() - loss) <= 1e-6 * max(1.0, abs(loss))
    assert torch.allclose(gi.cpu(), torch.tensor(g[r + "dimage"]), rtol=1e-5, atol=1e-9)
    assert torch.allclose(gd.cpu(), torch.tensor(g[r + "ddepth"]), rtol=1e-5, atol=1e-9)
    assert float(ga.abs().sum()) == 0.0
    da, db = g[r + "dexposure"]
    assert abs(out[1].item() - da) <= 2e-5 * abs(da) + 1e-8
    assert abs(out[2].item() - db) <= 2e-5 * abs(db) + 1e-8


def test_pose_step_matches_reference_adam_and_update_pose_fixture(pose_golden):
    """k_pose_init / k_pose_step against the reference's Camera, torch.optim.Adam (four groups) and update_pose
    (camera_utils.py:144-158, 7scenes_localize_full_dslam.py:33-64, pose_utils.py:105-122): 16 recorded steps"""
    from gs_localization_amd import _lib
    lib, g = _lib.load(), pose_golden
    st = torch.zeros(_lib.POSE_STATE_FLOATS)
    st[0:9] = torch.tensor(g["traj_R0"]).reshape(-1)
    st[9:12] = torch.tensor(g["traj_T0"])
    state = st.to(DEV)
    proj_d = torch.tensor(g["cam_proj_raw_T"]).contiguous().to(DEV)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.gsr_pose_init(_p(state), _p(proj_d), stream))
    torch.cuda.synchronize()
    assert torch.allclose(state[48:64].cpu().reshape(4, 4), torch.tensor(g["cam_view"]), atol=1e-7)
    assert torch.allclose(state[64:80].cpu().reshape(4, 4), torch.tensor(g["cam_fullproj"]), rtol=1e-5, atol=1e-6)
    assert torch.allclose(state[80:83].cpu(), torch.tensor(g["cam_center"]), atol=1e-6)
    assert g["traj_converged"].any() and not g["traj_converged"].all()
    for it, gr in enumerate(g["traj_grads"]):
        dtau = torch.tensor(np.concatenate([gr[3:6], gr[0:3]]), device=DEV)          # [rho, theta]
        lo = torch.tensor([0.5, gr[6], gr[7], 0.0], device=DEV)
        _lib.check(lib.gsr_pose_step(_p(state), _p(dtau), _p(lo), _p(proj_d), 0.001, float(g["traj_threshold"]), stream))
        torch.cuda.synchronize()
        s = state.cpu()
        assert torch.allclose(s[0:9].reshape(3, 3), torch.tensor(g["traj_R"][it]), atol=2e-6), it
        assert torch.allclose(s[9:12], torch.tensor(g["traj_T"][it]), atol=2e-6), it
        ea, eb = g["traj_exposure"][it]
        assert abs(s[18].item() - ea) < 1e-6 and abs(s[19].item() - eb) < 1e-6
        assert bool(s[37].item()) == bool(g["traj_converged"][it]), it
        assert float(s[12:18].abs().sum()) == 0.0
        assert torch.allclose(s[48:64].reshape(4, 4), torch.tensor(g["traj_view"][it]), atol=2e-6)


@pytest.mark.parametrize("name", ["sh3", "offcentre_white", "partial_tiles"])
def test_pose_gradient_matches_float64_fixture(pose_golden, name):
    """dL/dtau of the HIP path against float64 autograd through the REFERENCE's SE3_exp (SURVEY.md 8(c) fixture 8)"""
    from tests import util as U
    g = pose_golden
    P, W, H, deg, seed = (int(x) for x in g[f"tau_{name}_scene"])
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=seed)
    cx, cy, bg = g[f"tau_{name}_cxcy_bg"]
    sc.cx, sc.cy = float(cx), float(cy)
    sc.bg[:] = bg
    cam = U.scene_inputs(sc, g["tau_w2c"])
    grads = (g[f"tau_{name}_gc"], g[f"tau_{name}_gd"], np.zeros((1, H, W), np.float32))
    _, got = U.hip_run(sc, cam, grads, pose=True)
    assert U.rel_l1(got["tau"], g[f"tau_{name}_expected"]) <= 1e-5


def test_native_loop_follows_the_recorded_reference_loop_under_the_reference_mask():
    """... and under the mask every localiser of the reference refines under: the loop recorded by tests/golden/make_masked_loop_golden.py
    (Camera.compute_grad_mask | create_mask(keypoints) -> get_loss_tracking -> Adam -> update_pose, all the reference's own code around
    the CPU oracle).  The product's gsr_grad_mask reproduces the recorded mask bit for bit from the recorded observation."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "masked_loop_vectors.npz"))
    test_native_loop_follows_the_recorded_reference_loop(g)
    from gs_localization_amd import pipelines as PLN
    P, W, H, deg, seed = (int(x) for x in g["loop_scene"])
    got = PLN.grad_mask(torch.tensor(g["loop_gt_image"], device=DEV), 1.1, g["loop_keypoints"], 10)
    assert np.array_equal(got.cpu().numpy()[0], loop_mask(g, H, W).numpy()[0])


def test_native_loop_follows_the_recorded_reference_loop(pose_golden):
    """gsr_refine, k iterations, against the pose after k bodies of the reference's loop (its get_loss_tracking, Adam and
    update_pose around the CPU oracle's render / backward; SURVEY.md 8(c) fixture 9).  Each iteration moves every pose
    component by ~lr = 1e-3; the poses agree to 2e-6 and the last iteration's dL/dtau to 1e-5."""
    from tests import replay as PL, util as U
    g = pose_golden
    P, W, H, deg, seed = (int(x) for x in g["loop_scene"])
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=seed, scale_med=float(g["loop_scale_med"]))
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = torch.tensor(g["loop_init"], device=DEV)
    for k in (1, 4, 8):
        for spec in (False, True):
            vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, DEV), sc, DEV)
            vp.original_image = torch.tensor(g["loop_gt_image"], device=DEV)
            vp.depth = torch.tensor(g["loop_gt_depth"], device=DEV)
            vp.grad_mask = loop_mask(g, H, W).to(DEV)
            fr = PL.FusedRefiner(model, H, W, device=DEV)
            R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=k, speculative=spec)
            assert info["iters"] == k
            assert torch.allclose(R.cpu(), torch.tensor(g["loop_R"][k - 1]), atol=2e-6), (k, spec)
            assert torch.allclose(T.cpu(), torch.tensor(g["loop_T"][k - 1]), atol=2e-6), (k, spec)
            assert U.rel_l1(fr.g_tau.cpu().numpy(), g["loop_tau"][k - 1]) <= 1e-5, (k, spec)
        # (the loss is a mean of |render - observation| of a few 1e-2 per pixel, masked by opacity > 0.99: the rasterizers'
        # 1e-4 image differences and a handful of pixels on the other side of the mask move it by a fraction of a percent)
        assert abs(info["loss"] - float(g["loop_loss"][k - 1])) <= 2e-2 * float(g["loop_loss"][k - 1])


def _setup(sc, seed=0):
    from tests import replay as PL
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    return model, bg, (lambda: PL.make_frame(sc, model, DEV, bg)), PL.perturbed_start(seed, device=DEV)


def test_native_loop_matches_python_loop_and_converges():
    from tests import replay as PL
    sc = S.small(P=20000, W=160, H=120, sh_degree=3, seed=5, scale_med=0.03)
    model, bg, view, init = _setup(sc)
    cfg = PL.TRACKING_CONFIG
    for iters in (1, 8):
        vp1, vp2 = view(), view()
        R1, T1, _ = PL.python_loop(vp1, cfg, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=iters)
        fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
        R2, T2, info = fr.refine(vp2, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters)
        assert info["iters"] == iters
        # every iteration moves each pose component by ~lr = 1e-3 (Adam); both loops must take the same path
        assert torch.allclose(R1, R2, atol=2e-5), (iters, (R1 - R2).abs().max())
        assert torch.allclose(T1, T2, atol=2e-5), (iters, (T1 - T2).abs().max())
        assert abs(vp1.exposure_a.item() - vp2.exposure_a.item()) < 2e-5
    # known answer (SURVEY 8(c) fixture 9): 2 cm / 1 deg off, 50 iterations bring the pose back
    vp = view()
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV, gaussian_grads=False)
    R, T, info = fr.refine(vp, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=50)
    te0, re0 = PL.pose_errors(np.eye(3), np.zeros(3), init[:3, :3].cpu().numpy(), init[:3, 3].cpu().numpy())
    te, re = PL.pose_errors(np.eye(3), np.zeros(3), R.cpu().numpy(), T.cpu().numpy())
    assert te < 0.5 * te0 and re < 0.5 * re0, (te0, re0, te, re)


def test_native_loop_stops_on_convergence_like_reference():
    from tests import replay as PL
    sc = S.small(P=5000, W=96, H=64, sh_degree=1, seed=6, scale_med=0.05)
    model, bg, view, init = _setup(sc)
    vp = view()
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    eye = torch.eye(4, device=DEV)
    # start AT the ground truth with a huge threshold: the first update_pose already reports convergence
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, eye[:3, :3].clone(), eye[:3, 3].clone(), bg, iters=20, converged_threshold=1.0)
    assert info["converged"] and info["iters"] == 1
    vp2 = view()
    R2, T2, _ = PL.python_loop(vp2, PL.TRACKING_CONFIG, eye[:3, :3].clone(), eye[:3, 3].clone(), model, bg, iters=20)
    R3, T3, info3 = fr.refine(view(), PL.TRACKING_CONFIG, eye[:3, :3].clone(), eye[:3, 3].clone(), bg, iters=20, stop_on_converged=False)
    assert info3["iters"] == 20


def test_speculative_binning_is_exact_and_falls_back():
    """The native loop drops tile instances behind the depth each tile needed one iteration earlier.  The
    result must not depend on it: same pose as with complete lists, also when the bounds are made
    absurdly tight so that the device-side check fails and forwards are redone."""
    from tests import replay as PL
    sc = S.small(P=60000, W=160, H=128, sh_degree=2, seed=9, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=2)
    cfg = PL.TRACKING_CONFIG
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    runs = {}
    for name, kw in (("full", dict(speculative=False)), ("spec", dict(speculative=True)),
                     ("tight", dict(speculative=True, bound_margin=(0.6, 0.0)))):
        vp = view()
        R, T, info = fr.refine(vp, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=12, stop_on_converged=False, count_instances=True, **kw)
        runs[name] = (R.clone(), T.clone(), info, fr.color.clone(), fr.depth.clone())
    Rf, Tf, inf_f, cf, df = runs["full"]
    assert inf_f["fallbacks"] == 0
    for name in ("spec", "tight"):
        R, T, info, c, d = runs[name]
        assert torch.allclose(R, Rf, atol=2e-6) and torch.allclose(T, Tf, atol=2e-6), name
        # poses agree to ~1e-6 (fp32 atomics reorder the gradient sums); a 1e-6 rad pose change moves the image by ~1e-4
        assert torch.allclose(c, cf, atol=5e-4) and torch.allclose(d, df, atol=5e-3), name
    assert runs["spec"][2]["fallbacks"] == 0
    assert runs["spec"][2]["num_rendered"] < 0.6 * inf_f["num_rendered"]      # lists really got shorter
    assert runs["tight"][2]["fallbacks"] > 0                                   # and the safety net really fires
    # ... on the DEVICE: a failed group's successor retries with the bounds the failed forward recorded (LoopGuard tags); the host
    # steps in (drain + complete lists) only when the retry fails too or a bin overflowed
    assert runs["tight"][2]["host_redos"] < runs["tight"][2]["fallbacks"], runs["tight"][2]


def test_concurrent_frames_on_one_gpu_match_sequential():
    """bench.py keeps several frames in flight per GPU (one host thread + one stream each): the library must be
    re-entrant -- same poses as refining the frames one after the other."""
    import threading
    from tests import replay as PL
    sc = S.small(P=30000, W=160, H=128, sh_degree=3, seed=11, scale_med=0.04)
    model, bg, view, _ = _setup(sc)
    cfg = PL.TRACKING_CONFIG
    F = 3
    inits = []
    for f in range(F):
        rng = np.random.default_rng(50 + f)
        tau = np.concatenate([rng.normal(size=3) * 0.01, rng.normal(size=3) * 0.01])
        inits.append(torch.tensor(S.se3_exp(tau), dtype=torch.float32, device=DEV))
    frs = [PL.FusedRefiner(model, sc.H, sc.W, device=DEV) for _ in range(F)]
    seq = []
    for f in range(F):
        R, T, _ = frs[f].refine(view(), cfg, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=10, stop_on_converged=False)
        seq.append((R.clone(), T.clone()))
    vps = [view() for _ in range(F)]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(F)]
    out = [None] * F

    def work(f):
        with torch.cuda.stream(streams[f]):
            out[f] = frs[f].refine(vps[f], cfg, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=10, stop_on_converged=False)
    ts = [threading.Thread(target=work, args=(f,)) for f in range(F)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    for f in range(F):
        assert torch.allclose(out[f][0], seq[f][0], atol=2e-6) and torch.allclose(out[f][1], seq[f][1], atol=2e-6), f


def test_native_loop_gradient_tensors_stay_consistent():
    """The native loop zero-fills the Gaussian-gradient tensors once and then only touches rows that change
    (dirty bits).  After several iterations they must equal what a fresh autograd backward gives at the
    same pose -- including exact zeros in rows that dropped out of view."""
    from tests import replay as PL
    from tests.util import rel_l1
    sc = S.small(P=40000, W=160, H=128, sh_degree=3, seed=21, scale_med=0.04)
    # make visibility change between iterations: a thick shell of splats right at the near plane
    sc.means3D[:4000, 2] = 0.2 + np.random.default_rng(0).uniform(-0.01, 0.01, 4000).astype(np.float32)
    model, bg, view, init = _setup(sc, seed=3)
    cfg = PL.TRACKING_CONFIG
    frA = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    frA.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=6, stop_on_converged=False)
    frB = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    vpB = view()
    frB.refine(vpB, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False)
    for t in (model.get_xyz, model.get_features, model.get_opacity, model.get_scaling, model.get_rotation):
        t.grad = None
    pkg = PL.render(vpB, model, bg)
    PL.tracking_loss(cfg, pkg["render"], pkg["depth"], pkg["opacity"], vpB).backward()
    ref = dict(m3d=model.get_xyz.grad, sh=model.get_features.grad, opac=model.get_opacity.grad,
               scale=model.get_scaling.grad, rot=model.get_rotation.grad)
    got = dict(m3d=frA.g_m3d, sh=frA.g_sh, opac=frA.g_opac, scale=frA.g_scale, rot=frA.g_rot)
    for k in ref:
        a, b = got[k].detach().cpu().numpy(), ref[k].detach().cpu().numpy().reshape(got[k].shape)
        assert rel_l1(a, b) < 2e-4, (k, rel_l1(a, b))
        # rows that are exactly zero in the fresh backward must be exactly zero here too (nothing stale left)
        zero_rows = np.all(b.reshape(b.shape[0], -1) == 0, axis=1)
        assert zero_rows.sum() > 1000
        assert np.all(a.reshape(a.shape[0], -1)[zero_rows] == 0), k


def test_gradient_tensors_and_covariances_carried_from_one_frame_to_the_next():
    """gsr_refine_args.carry_state: a refiner's second refine() neither zero-fills its gradient tensors nor rebuilds the 3D
    covariances.  A second frame at a very different pose (other Gaussians in view) must come out exactly as from a fresh
    refiner -- rows the first frame wrote and the second does not touch must be zero again -- and writing into one of the
    tensors from outside must withdraw the promise."""
    from tests import replay as PL
    sc = S.small(P=40000, W=160, H=128, sh_degree=3, seed=22, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=4)
    cfg = PL.TRACKING_CONFIG
    far = torch.tensor(S.se3_exp([0.6, -0.4, 0.5, 0.25, -0.5, 0.2]), dtype=torch.float32, device=DEV)
    names = ("g_m2d", "g_conic", "g_opac", "g_col", "g_m3d", "g_cov", "g_sh", "g_scale", "g_rot")

    def second_frame(fr):
        vp = view()
        R, T, info = fr.refine(vp, cfg, far[:3, :3].clone(), far[:3, 3].clone(), bg, iters=4, stop_on_converged=False)
        torch.cuda.synchronize()
        return R.clone(), T.clone(), {n: getattr(fr, n).detach().cpu().numpy().copy() for n in names}, fr.color.cpu().numpy().copy()

    fresh = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    Rf, Tf, gf, cf = second_frame(fresh)
    used = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    used.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False)
    assert used._carry.value == 3
    first_rows = used.g_sh.detach().abs().sum(dim=(1, 2)).cpu().numpy() != 0
    Ru, Tu, gu, cu = second_frame(used)
    assert used._carry.value == 3
    assert torch.allclose(Rf, Ru, atol=1e-6) and torch.allclose(Tf, Tu, atol=1e-6)
    assert U.rel_l1(cu, cf) <= 1e-6
    second_rows = np.abs(gf["g_sh"]).sum(axis=(1, 2)) != 0
    assert (first_rows & ~second_rows).sum() > 30, "the two frames must see different Gaussians for this test to mean anything"
    for n in names:
        zero = np.all(gf[n].reshape(gf[n].shape[0], -1) == 0, axis=1)
        assert np.all(gu[n].reshape(gu[n].shape[0], -1)[zero] == 0), n          # nothing stale from the first frame
        assert U.rel_l1(gu[n], gf[n]) <= 2e-5, n
    # an outside write into a gradient tensor: the next call must start from a zero fill again
    used.g_sh.add_(1.0)
    Rw, Tw, gw, _ = second_frame(used)
    assert U.rel_l1(gw["g_sh"], gf["g_sh"]) <= 2e-5
    assert torch.allclose(Rf, Rw, atol=1e-6)


@pytest.mark.parametrize("W,H,mono,conv_thr", [(150, 100, False, 1e-4), (96, 70, True, 3e-3)])
def test_final_state_equals_fresh_render_at_returned_pose(W, H, mono, conv_thr):
    """Whatever the loop did internally (speculation, frozen iterations after convergence, in-kernel clears), what it
    leaves behind -- image, depth, opacity, n_touched, radii -- must be the render of the pose it reports as the last
    forward's, like the reference's last render_pkg.  Odd image sizes, monocular config, early exit."""
    from tests import replay as PL
    sc = S.small(P=12000, W=W, H=H, sh_degree=2, seed=21, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=4)
    cfg = {"Training": dict(PL.TRACKING_CONFIG["Training"], monocular=mono)}
    fr = PL.FusedRefiner(model, H, W, device=DEV)
    for stop in (True, False):
        vp = view()
        R, T, info = fr.refine(vp, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=40, converged_threshold=conv_thr,
                               stop_on_converged=stop)
        torch.cuda.synchronize()
        if stop and conv_thr > 1e-3:
            assert info["converged"] and info["iters"] < 40        # this configuration must take the early exit
        if stop and info["converged"]:
            Rl, Tl = R, T                                  # converged: the last forward was rendered at the final pose
        else:
            continue                                       # (not converged: the last forward precedes the last update)
        chk = view()
        chk.update_RT(Rl.clone(), Tl.clone())
        with torch.no_grad():
            pkg = PL.render(chk, model, bg)
        # the native state keeps R, T in fp32 on the device; the camera rebuilt on the host agrees to ~1e-7
        assert torch.allclose(fr.color, pkg["render"], atol=2e-4), float((fr.color - pkg["render"]).abs().max())
        assert torch.allclose(fr.depth, pkg["depth"], atol=2e-3)
        assert torch.allclose(fr.alpha, pkg["opacity"], atol=2e-4)
        assert int((fr.radii != pkg["radii"]).sum()) <= max(2, int(5e-5 * fr.radii.numel()))
        nt = pkg["n_touched"]
        assert int((fr.n_touched - nt).abs().sum()) <= max(4, int(2e-3 * int(nt.sum())))
    assert info["iters"] == 40


def test_warm_start_of_the_speculation_is_exact():
    """Frame sequences: refine() can start speculating from the bounds the previous call left in the workspace.
    Same result as a cold start -- from the same frame, from a neighbouring pose, and from an unrelated one (where
    the verification has to catch the stale bounds)."""
    from tests import replay as PL
    sc = S.small(P=40000, W=160, H=128, sh_degree=2, seed=13, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=3)
    cfg = PL.TRACKING_CONFIG
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)

    def run(start, warm):
        vp = view()
        R, T, info = fr.refine(vp, cfg, start[:3, :3].clone(), start[:3, 3].clone(), bg, iters=10, stop_on_converged=False, warm_start=warm)
        return R.clone(), T.clone(), dict(info), fr.color.clone()
    cold = run(init, False)
    assert (fr._warm.value & 0xFF) in (1, 2)
    warm_same = run(init, True)                        # bounds of the same frame
    near = torch.tensor(S.se3_exp([0.004, -0.003, 0.002, 0.002, -0.001, 0.001]), dtype=torch.float32, device=DEV) @ init
    cold_near = run(near, False)
    run(init, False)
    warm_near = run(near, True)                        # bounds of a neighbouring frame
    far = torch.tensor(S.se3_exp([0.3, 0.2, -0.4, 0.25, -0.2, 0.1]), dtype=torch.float32, device=DEV)
    cold_far = run(far, False)
    run(init, False)
    warm_far = run(far, True)                          # unrelated bounds: must still be exact
    for a, b, name in ((cold, warm_same, "same"), (cold_near, warm_near, "near"), (cold_far, warm_far, "far")):
        assert torch.allclose(a[0], b[0], atol=2e-6) and torch.allclose(a[1], b[1], atol=2e-6), name
        # (far from the map's sweet spot a 3e-7 pose difference -- fp32 atomics reorder the gradient sums -- already
        # moves single pixels by 1e-3)
        assert torch.allclose(a[3], b[3], atol=5e-3 if name == "far" else 5e-4), name
    assert warm_same[2]["fallbacks"] == 0
    # a cold start bins its first iteration completely, a warm one does not: fewer instances in a 1-iteration call
    fr.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, stop_on_converged=False, warm_start=False, count_instances=True)
    n_cold = fr.last_info["num_rendered"]
    fr.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, stop_on_converged=False, warm_start=True, count_instances=True)
    assert fr.last_info["num_rendered"] < 0.6 * n_cold


@pytest.mark.parametrize("W,H", [(1920, 1080),       # 8 160 tiles
                                 (2576, 1616)])      # 16 261 tiles (more than a workgroup's LDS could hold bounds for)
def test_native_loop_on_large_images_matches_python_loop(W, H):
    from tests import replay as PL
    sc = S.small(P=30000, W=W, H=H, sh_degree=2, seed=8, scale_med=0.03)
    model, bg, view, init = _setup(sc)
    cfg = PL.TRACKING_CONFIG
    vp1, vp2 = view(), view()
    R1, T1, _ = PL.python_loop(vp1, cfg, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=6)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    R2, T2, info = fr.refine(vp2, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=6)
    assert info["iters"] == 6
    assert torch.allclose(R1, R2, atol=2e-5), (R1 - R2).abs().max()
    assert torch.allclose(T1, T2, atol=2e-5), (T1 - T2).abs().max()


def test_speculation_on_a_half_empty_scene():
    """Tiles that never saturate (sky, holes in the map) have no finite depth bound: nothing is ever dropped from them,
    so an unsaturated pixel there is not a failed speculation.  Half of this image is dense, the other half sparse and
    faint; the loop must speculate without a single redone forward and agree with complete lists."""
    from tests import replay as PL
    sc = S.small(P=60000, W=160, H=128, sh_degree=1, seed=14, scale_med=0.04)
    right = sc.means3D[:, 0] > 0
    keep = ~right | (np.arange(sc.P) % 40 == 0)            # thin the right half out
    sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs = (np.ascontiguousarray(x[keep]) for x in
                                                                   (sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs))
    sc.opacities[sc.means3D[:, 0] > 0] *= 0.2               # and make it faint
    model, bg, view, init = _setup(sc, seed=3)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    out = {}
    for spec in (False, True):
        vp = view()
        R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=12, stop_on_converged=False,
                               speculative=spec, count_instances=True)
        out[spec] = (R.clone(), T.clone(), info, fr.alpha.clone())
    a = out[False][3]
    assert float((a[..., sc.W // 2 + 16:] < 0.9).float().mean()) > 0.5       # the right half really is unsaturated
    assert float((a[..., :sc.W // 2 - 16] > 0.999).float().mean()) > 0.9     # and the left half saturated
    assert out[True][2]["fallbacks"] == 0
    assert out[True][2]["num_rendered"] < 0.8 * out[False][2]["num_rendered"]
    assert torch.allclose(out[True][0], out[False][0], atol=2e-6) and torch.allclose(out[True][1], out[False][1], atol=2e-6)


def test_converged_exit_with_an_overflowing_speculative_bin_returns_a_verified_render():
    """ADVICE (round 1): when the loop stops on convergence, the forward it hands back was enqueued speculatively; if that
    speculation failed -- here: a tile whose bin overflows its capacity (8192 entries at this size), so the compositing kernel
    gives up on it -- the returned images must still be those of a complete render at the final pose.  Dense faint splats
    make every tile's list longer than that even after the depth bounds (nothing saturates, so nothing can be dropped); a huge
    threshold makes the first update 'converge'."""
    from tests import replay as PL
    sc = S.small(P=250000, W=64, H=48, sh_degree=1, seed=43, scale_med=0.12)
    sc.opacities[:] = np.clip(sc.opacities * 0.01, 0.005, 0.01)
    model, bg, view, init = _setup(sc, seed=5)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    vp = view()
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=10, converged_threshold=1.0)
    torch.cuda.synchronize()
    assert info["converged"] and info["iters"] == 1
    assert info["fallbacks"] >= 1          # the speculative forward at the final pose was caught and redone
    chk = view()
    chk.update_RT(R.clone(), T.clone())
    with torch.no_grad():
        pkg = PL.render(chk, model, bg)
    assert torch.allclose(fr.color, pkg["render"], atol=2e-4), float((fr.color - pkg["render"]).abs().max())
    assert torch.allclose(fr.depth, pkg["depth"], atol=2e-3)
    assert torch.allclose(fr.alpha, pkg["opacity"], atol=2e-4)
    assert int((fr.radii != pkg["radii"]).sum()) <= max(2, int(5e-5 * fr.radii.numel()))
    nt = pkg["n_touched"]
    assert int((fr.n_touched - nt).abs().sum()) <= max(4, int(2e-3 * int(nt.sum())))


def test_work_balanced_tile_order_changes_nothing(monkeypatch):
    """The native loop launches the compositing kernels in a per-iteration tile order sorted by the work the previous
    iteration measured (tile_order_from_work).  Any permutation of the tiles is correct: same poses and images with the
    ordering switched off (GSR_NO_BALANCE, read by gsr_refine on every call), up to the order of the fp32 atomics."""
    from tests import replay as PL
    sc = S.small(P=60000, W=208, H=160, sh_degree=2, seed=12, scale_med=0.04)      # 13 x 10 tiles, a ragged last row of blocks
    model, bg, view, init = _setup(sc, seed=4)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    out = {}
    for name in ("balanced", "plain"):
        if name == "plain":
            monkeypatch.setenv("GSR_NO_BALANCE", "1")
        R, T, info = fr.refine(view(), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=10, stop_on_converged=False)
        out[name] = (R.clone(), T.clone(), fr.color.clone(), fr.depth.clone(), info)
    monkeypatch.delenv("GSR_NO_BALANCE")
    (R1, T1, c1, d1, i1), (R2, T2, c2, d2, i2) = out["balanced"], out["plain"]
    assert i1["fallbacks"] == 0 and i2["fallbacks"] == 0
    assert torch.allclose(R1, R2, atol=2e-6) and torch.allclose(T1, T2, atol=2e-6)
    assert torch.allclose(c1, c2, atol=5e-4) and torch.allclose(d1, d2, atol=5e-3)


# (k_preprocess_lean: tests/test_gpu_lean.py)


def test_long_bins_are_ordered_lazily():
    """A tile that does not saturate has no depth bound and gets its complete list in the native loop's bin; up to the bin
    capacity (8192 entries here) such a bin is ordered slice by slice instead of failing the speculation: same poses and images
    as the loop without speculation, no forward redone, although most tiles carry lists of a few thousand entries."""
    from tests import replay as PL
    sc = S.small(P=45000, W=64, H=48, sh_degree=1, seed=47, scale_med=0.1)
    sc.opacities[:] = np.clip(sc.opacities * 0.01, 0.002, 0.01)
    model, bg, view, init = _setup(sc, seed=7)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    out = {}
    for spec in (False, True):
        R, T, info = fr.refine(view(), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=6, stop_on_converged=False,
                               speculative=spec, count_instances=True)
        out[spec] = (R.clone(), T.clone(), info, fr.color.clone(), fr.alpha.clone())
    ntiles = 4 * 3
    assert out[True][2]["num_rendered"] > 2048 * ntiles and out[True][2]["num_rendered"] <= 8192 * ntiles      # long bins, within capacity
    assert float(out[True][4].max()) < 0.999                                                                     # nothing saturates
    assert out[True][2]["fallbacks"] == 0
    assert torch.allclose(out[True][0], out[False][0], atol=2e-6) and torch.allclose(out[True][1], out[False][1], atol=2e-6)
    assert torch.allclose(out[True][3], out[False][3], atol=5e-4)


def test_host_redo_of_a_warm_started_call_survives_an_overflowing_complete_list_bin():
    """ADVICE (round 3): a warm-started call never runs a complete-list forward of its own, so when the host has to redo a
    forward (here: the frozen render behind a converged update, at a pose a huge learning rate threw far from the one the depth
    bounds were recorded at) the complete lists go into the fixed-capacity bins of k_preprocess_bin for the first time -- and one
    tile of this scene holds 90 % of the map behind an opaque wall (its speculative list is short, its complete list is twenty
    times its bin).  The redo must fall back to count -> scan -> emit instead of reporting an error, and what it returns must be
    the render at the final pose."""
    from tests import replay as PL
    base = S.small(P=20000, W=320, H=240, sh_degree=1, seed=51, scale_med=0.05)
    rng = np.random.default_rng(52)
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    # an opaque wall at z = 1 m around the optical axis ...
    gx, gy = np.meshgrid(np.linspace(-0.45, 0.45, 13), np.linspace(-0.45, 0.45, 13))
    wall = np.stack([gx.ravel() + 0.03, gy.ravel(), np.full(gx.size, 1.0)], 1)
    nw = wall.shape[0]
    # ... and 180 000 small splats 3 m away, all inside a few pixels behind it
    nc = 180000
    clus = np.stack([0.094 + rng.normal(0, 0.004, nc), rng.normal(0, 0.004, nc), rng.uniform(3.0, 3.2, nc)], 1)
    ident = np.tile(np.array([[1.0, 0, 0, 0]]), (nw + nc, 1))
    base.means3D = f32(np.concatenate([base.means3D, wall, clus]))
    base.scales = f32(np.concatenate([base.scales, np.full((nw, 3), 0.15), np.full((nc, 3), 0.01)]))
    base.rotations = f32(np.concatenate([base.rotations, ident]))
    base.opacities = f32(np.concatenate([base.opacities, np.full((nw, 1), 0.99), np.full((nc, 1), 0.5)]))
    base.shs = f32(np.concatenate([base.shs, rng.normal(0, 1, (nw + nc,) + base.shs.shape[1:]) * 0.3]))
    sc = base
    model, bg, view, init = _setup(sc, seed=6)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    # call 1 (cold): leaves depth bounds behind; its own complete-list forward overflows and goes through the exact path
    _, _, info1 = fr.refine(view(), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=3, stop_on_converged=False)
    assert info1["fallbacks"] >= 1 and (fr._warm.value & 0xFF) in (1, 2)
    # call 2 (warm): the first update "converges" (huge threshold) after a 0.05 rad / 5 cm step; the frozen forward at that pose
    # fails its verification somewhere, and the host's redo meets the overflowing bin
    vp = view()
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=10, lr=0.05, converged_threshold=1.0,
                           warm_start=True)
    torch.cuda.synchronize()
    assert info["converged"] and info["iters"] == 1
    assert info["host_redos"] >= 1, info          # (otherwise this scene no longer exercises the redo: make the step larger)
    chk = view()
    chk.update_RT(R.clone(), T.clone())
    with torch.no_grad():
        pkg = PL.render(chk, model, bg)
    assert torch.allclose(fr.color, pkg["render"], atol=2e-4), float((fr.color - pkg["render"]).abs().max())
    assert torch.allclose(fr.depth, pkg["depth"], atol=2e-3)
    assert torch.allclose(fr.alpha, pkg["opacity"], atol=2e-4)
    assert int((fr.radii != pkg["radii"]).sum()) <= max(2, int(5e-5 * fr.radii.numel()))


def test_native_loop_with_precomputed_inputs_matches_the_python_loop():
    """render() (B) can be configured to hand the rasterizer precomputed colours and / or covariances instead of SH coefficients
    and scale / rotation (tools/__init__.py:85-112: pipe.convert_SHs_python, pipe.compute_cov3D_python).  gsr_refine_args expresses
    both since ABI 3 (missing item 4): the native loop in each mode against the reference-style Python loop on the pose
    package in the same mode -- same poses, and the gradients of the precomputed tensors it maintains against a fresh backward."""
    import math
    from tests import replay as PL
    from diff_gaussian_rasterization_pose import GaussianRasterizationSettings, GaussianRasterizer
    sc = S.small(P=15000, W=128, H=96, sh_degree=2, seed=61, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=6)
    rng = np.random.default_rng(62)
    q = sc.rotations.astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                   2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = Rm * sc.scales.astype(np.float64)[:, None, :]
    Sig = M @ M.transpose(0, 2, 1)
    cov6 = torch.tensor(np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], 1).astype(np.float32), device=DEV)
    cols = torch.tensor(rng.uniform(0.0, 1.0, (sc.P, 3)).astype(np.float32), device=DEV)

    def render(frame, colors, cov):
        means2D = torch.zeros_like(model.get_xyz, requires_grad=True)
        rs = GaussianRasterizationSettings(
            image_height=sc.H, image_width=sc.W, tanfovx=math.tan(0.5 * frame.FoVx), tanfovy=math.tan(0.5 * frame.FoVy), bg=bg, scale_modifier=1.0,
            viewmatrix=frame.world_view_transform, projmatrix=frame.full_proj_transform, projmatrix_raw=frame.projection_matrix,
            sh_degree=model.active_sh_degree, campos=frame.camera_center, prefiltered=False, debug=False)
        return GaussianRasterizer(rs)(means3D=model.get_xyz, means2D=means2D, opacities=model.get_opacity,
                                      shs=None if colors is not None else model.get_features, colors_precomp=colors,
                                      scales=None if cov is not None else model.get_scaling, rotations=None if cov is not None else model.get_rotation,
                                      cov3D_precomp=cov, theta=frame.cam_rot_delta, rho=frame.cam_trans_delta)

    for colors, cov in ((cols, None), (None, cov6), (cols, cov6)):
        c_ = None if colors is None else colors.clone().requires_grad_(True)
        v_ = None if cov is None else cov.clone().requires_grad_(True)
        vp = view()
        vp.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
        opt = PL.pose_adam(vp)
        for _ in range(6):
            img, radii, depth, opac, nt = render(vp, c_, v_)
            opt.zero_grad()
            if c_ is not None: c_.grad = None
            if v_ is not None: v_.grad = None
            PL.tracking_loss(PL.TRACKING_CONFIG, img, depth, opac, vp).backward()
            last_pose = (vp.R.clone(), vp.T.clone())
            with torch.no_grad():
                opt.step()
                PL.apply_pose_delta(vp)
        fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV, colors_precomp=colors, cov3D_precomp=cov)
        R2, T2, info = fr.refine(view(), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=6, stop_on_converged=False)
        assert info["iters"] == 6
        assert torch.allclose(vp.R, R2, atol=2e-5) and torch.allclose(vp.T, T2, atol=2e-5), (colors is not None, cov is not None)
        # the gradient tensors the loop maintains are those of its LAST backward (at the pose before the last update)
        if c_ is not None:
            assert U.rel_l1(fr.g_col.cpu().numpy(), c_.grad.cpu().numpy()) <= 2e-4
            assert fr.g_sh is None
        if v_ is not None:
            assert U.rel_l1(fr.g_cov.cpu().numpy(), v_.grad.cpu().numpy()) <= 2e-4
            assert fr.g_scale is None and fr.g_rot is None
