"""-m gpu : the fused pose-refinement epilogue against plain PyTorch fp32 references of
the same ops: tracking loss + gradient (torch autograd of the mirror of descent_utils.py, itself pinned to
the reference by tests/golden), Adam + update_pose (torch.optim.Adam + the mirror of pose_utils.py), and
the whole native loop against the reference-style Python loop on the same rasterizer."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from gs_localization_amd import scenes as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _p(t):
    return None if t is None else t.data_ptr()


@pytest.mark.parametrize("mono", [False, True])
def test_tracking_loss_kernel_matches_torch_autograd(mono):
    from gs_localization_amd import _lib, pipelines as PL
    lib = _lib.load()
    torch.manual_seed(0)
    H, W = 37, 53
    image = torch.rand(3, H, W, device=DEV, requires_grad=True)
    depth = (torch.rand(1, H, W, device=DEV) * 4 + 0.5).requires_grad_(True)
    opacity = torch.rand(1, H, W, device=DEV) * 0.1 + 0.93
    gt = torch.rand(3, H, W, device=DEV)
    gt_depth = torch.rand(H, W, device=DEV) * 4
    gt_depth[torch.rand(H, W, device=DEV) < 0.2] = 0
    mask = torch.rand(1, H, W, device=DEV) < 0.7

    class VP:
        pass
    vp = VP()
    vp.exposure_a = torch.tensor([0.07], device=DEV, requires_grad=True)
    vp.exposure_b = torch.tensor([-0.03], device=DEV, requires_grad=True)
    vp.original_image, vp.depth, vp.grad_mask = gt, gt_depth, mask
    cfg = {"Training": {"monocular": mono, "alpha": 0.99, "opacity_threshold": 0.99}}
    loss = PL.get_loss_tracking(cfg, image, depth, opacity, vp)
    loss.backward()
    gi, gd, ga = torch.empty_like(image), torch.empty(1, H, W, device=DEV), torch.empty(1, H, W, device=DEV)
    out = torch.empty(4, device=DEV)
    expo = torch.tensor([0.07, -0.03], device=DEV)
    m8 = mask.reshape(H, W).to(torch.uint8).contiguous()
    _lib.check(lib.gsr_tracking_loss(W, H, _p(image.detach()), _p(depth.detach()), _p(opacity), _p(gt), _p(gt_depth), _p(m8),
                                     _p(expo), 0.99, 0.01, int(mono), _p(gi), _p(gd), _p(ga), _p(out),
                                     torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert abs(out[0].item() - loss.item()) <= 1e-6 * max(1.0, abs(loss.item()))
    assert torch.allclose(gi, image.grad, rtol=1e-5, atol=1e-9)
    dref = depth.grad if depth.grad is not None else torch.zeros_like(gd)
    assert torch.allclose(gd, dref, rtol=1e-5, atol=1e-9)
    assert float(ga.abs().sum()) == 0.0
    assert abs(out[1].item() - vp.exposure_a.grad.item()) <= 2e-5 * abs(vp.exposure_a.grad.item()) + 1e-8
    assert abs(out[2].item() - vp.exposure_b.grad.item()) <= 2e-5 * abs(vp.exposure_b.grad.item()) + 1e-8


def test_pose_step_matches_torch_adam_and_update_pose():
    from gs_localization_amd import _lib, pipelines as PL
    lib = _lib.load()
    rng = np.random.default_rng(3)
    proj = PL.getProjectionMatrix2(0.01, 100.0, 320, 240, 525, 525, 640, 480).transpose(0, 1).contiguous()
    T0 = torch.tensor(S.se3_exp([0.3, -0.2, 0.5, 0.1, -0.3, 0.2]), dtype=torch.float32)
    cam = PL.Camera(0, None, None, torch.eye(4), proj, 525, 525, 320, 240, 1.0, 1.0, 480, 640, device="cpu")
    cam.update_RT(T0[:3, :3].clone(), T0[:3, 3].clone())
    opt = PL.make_pose_optimizer(cam)
    st = torch.zeros(_lib.POSE_STATE_FLOATS)
    st[0:9] = T0[:3, :3].reshape(-1)
    st[9:12] = T0[:3, 3]
    state = st.to(DEV)
    proj_d = proj.to(DEV)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.gsr_pose_init(_p(state), _p(proj_d), stream))
    torch.cuda.synchronize()
    assert torch.allclose(state[48:64].cpu().reshape(4, 4), cam.world_view_transform, atol=1e-7)
    assert torch.allclose(state[64:80].cpu().reshape(4, 4), cam.full_proj_transform, rtol=1e-5, atol=1e-6)
    assert torch.allclose(state[80:83].cpu(), cam.camera_center, atol=1e-6)
    for it in range(12):
        scale = 10.0 ** rng.uniform(-6, 1)           # also drives |tau| below the 1e-4 threshold sometimes
        g = (rng.normal(size=8) * scale).astype(np.float32)
        cam.cam_rot_delta.grad = torch.tensor(g[0:3])
        cam.cam_trans_delta.grad = torch.tensor(g[3:6])
        cam.exposure_a.grad = torch.tensor(g[6:7])
        cam.exposure_b.grad = torch.tensor(g[7:8])
        with torch.no_grad():
            opt.step()
            conv = bool(PL.update_pose(cam, 1e-4))
        dtau = torch.tensor(np.concatenate([g[3:6], g[0:3]]), device=DEV)          # [rho, theta]
        lo = torch.tensor([0.5, g[6], g[7], 0.0], device=DEV)
        _lib.check(lib.gsr_pose_step(_p(state), _p(dtau), _p(lo), _p(proj_d), 0.001, 1e-4, stream))
        torch.cuda.synchronize()
        s = state.cpu()
        assert torch.allclose(s[0:9].reshape(3, 3), cam.R, atol=2e-6), it
        assert torch.allclose(s[9:12], cam.T, atol=2e-6), it
        assert abs(s[18].item() - cam.exposure_a.item()) < 1e-6 and abs(s[19].item() - cam.exposure_b.item()) < 1e-6
        assert bool(s[37].item()) == conv, it
        assert float(s[12:18].abs().sum()) == 0.0
        assert torch.allclose(s[48:64].reshape(4, 4), cam.world_view_transform, atol=2e-6)


def _setup(sc, seed=0):
    from gs_localization_amd import pipelines as PL
    W, H = sc.W, sc.H
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    proj = PL.getProjectionMatrix2(0.01, 100.0, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, W=W, H=H).transpose(0, 1).to(DEV)
    fovx, fovy = PL.focal2fov(sc.fx, W), PL.focal2fov(sc.fy, H)
    gt = torch.eye(4, device=DEV)

    def view():
        vp = PL.Camera(0, None, None, gt, proj, sc.fx, sc.fy, sc.cx, sc.cy, fovx, fovy, H, W, device=DEV)
        with torch.no_grad():
            pkg = PL.render(vp, model, PL.PipelineParams(), bg)
        vp.original_image = pkg["render"].detach().clone()
        vp.depth = pkg["depth"].detach()[0].clone()
        vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=DEV)
        return vp
    rng = np.random.default_rng(seed)
    dt = rng.normal(size=3); dt *= 0.02 / np.linalg.norm(dt)
    dr = rng.normal(size=3); dr *= math.radians(1.0) / np.linalg.norm(dr)
    init = torch.tensor(S.se3_exp(np.concatenate([dt, dr])), dtype=torch.float32, device=DEV)
    return model, bg, view, init


def test_native_loop_matches_python_loop_and_converges():
    from gs_localization_amd import pipelines as PL
    sc = S.small(P=20000, W=160, H=120, sh_degree=3, seed=5, scale_med=0.03)
    model, bg, view, init = _setup(sc)
    cfg = PL.TRACKING_CONFIG
    for iters in (1, 8):
        vp1, vp2 = view(), view()
        R1, T1, _ = PL.gradient_decent(vp1, cfg, init[:3, :3].clone(), init[:3, 3].clone(), model, PL.PipelineParams(), bg, iters=iters)
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
    from gs_localization_amd import pipelines as PL
    sc = S.small(P=5000, W=96, H=64, sh_degree=1, seed=6, scale_med=0.05)
    model, bg, view, init = _setup(sc)
    vp = view()
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    eye = torch.eye(4, device=DEV)
    # start AT the ground truth with a huge threshold: the first update_pose already reports convergence
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, eye[:3, :3].clone(), eye[:3, 3].clone(), bg, iters=20, converged_threshold=1.0)
    assert info["converged"] and info["iters"] == 1
    vp2 = view()
    R2, T2, _ = PL.gradient_decent(vp2, PL.TRACKING_CONFIG, eye[:3, :3].clone(), eye[:3, 3].clone(), model, PL.PipelineParams(), bg, iters=20)
    R3, T3, info3 = fr.refine(view(), PL.TRACKING_CONFIG, eye[:3, :3].clone(), eye[:3, 3].clone(), bg, iters=20, stop_on_converged=False)
    assert info3["iters"] == 20


def test_speculative_binning_is_exact_and_falls_back():
    """The native loop drops tile instances behind the depth each tile needed one iteration earlier.  The
    result must not depend on it: same pose as with complete lists, also when the bounds are made
    absurdly tight so that the device-side check fails and forwards are redone."""
    from gs_localization_amd import pipelines as PL
    sc = S.small(P=60000, W=160, H=128, sh_degree=2, seed=9, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=2)
    cfg = PL.TRACKING_CONFIG
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    runs = {}
    for name, kw in (("full", dict(speculative=False)), ("spec", dict(speculative=True)),
                     ("tight", dict(speculative=True, bound_margin=(0.6, 0.0)))):
        vp = view()
        R, T, info = fr.refine(vp, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=12, stop_on_converged=False, **kw)
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


def test_concurrent_frames_on_one_gpu_match_sequential():
    """bench.py keeps several frames in flight per GPU (one host thread + one stream each): the library must be
    re-entrant -- same poses as refining the frames one after the other."""
    import threading
    from gs_localization_amd import pipelines as PL
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
    from gs_localization_amd import pipelines as PL
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
    pkg = PL.render(vpB, model, PL.PipelineParams(), bg)
    PL.get_loss_tracking(cfg, pkg["render"], pkg["depth"], pkg["opacity"], vpB).backward()
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


@pytest.mark.parametrize("W,H,mono,conv_thr", [(150, 100, False, 1e-4), (96, 70, True, 3e-3)])
def test_final_state_equals_fresh_render_at_returned_pose(W, H, mono, conv_thr):
    """Whatever the loop did internally (speculation, frozen iterations after convergence, in-kernel clears), what it
    leaves behind -- image, depth, opacity, n_touched, radii -- must be the render of the pose it reports as the last
    forward's, like the reference's last render_pkg.  Odd image sizes, monocular config, early exit."""
    from gs_localization_amd import pipelines as PL
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
            pkg = PL.render(chk, model, PL.PipelineParams(), bg)
        # the native state keeps R, T in fp32 on the device; the camera rebuilt on the host agrees to ~1e-7
        assert torch.allclose(fr.color, pkg["render"], atol=2e-4), float((fr.color - pkg["render"]).abs().max())
        assert torch.allclose(fr.depth, pkg["depth"], atol=2e-3)
        assert torch.allclose(fr.alpha, pkg["opacity"], atol=2e-4)
        assert int((fr.radii != pkg["radii"]).sum()) <= 2
        nt = pkg["n_touched"]
        assert int((fr.n_touched - nt).abs().sum()) <= max(4, int(2e-3 * int(nt.sum())))
    assert info["iters"] == 40


def test_warm_start_of_the_speculation_is_exact():
    """Frame sequences: refine() can start speculating from the bounds the previous call left in the workspace.
    Same result as a cold start -- from the same frame, from a neighbouring pose, and from an unrelated one (where
    the verification has to catch the stale bounds)."""
    from gs_localization_amd import pipelines as PL
    sc = S.small(P=40000, W=160, H=128, sh_degree=2, seed=13, scale_med=0.04)
    model, bg, view, init = _setup(sc, seed=3)
    cfg = PL.TRACKING_CONFIG
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)

    def run(start, warm):
        vp = view()
        R, T, info = fr.refine(vp, cfg, start[:3, :3].clone(), start[:3, 3].clone(), bg, iters=10, stop_on_converged=False, warm_start=warm)
        return R.clone(), T.clone(), dict(info), fr.color.clone()
    cold = run(init, False)
    assert fr._warm.value in (1, 2)
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
    fr.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, stop_on_converged=False)
    n_cold = fr.last_info["num_rendered"]
    fr.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, stop_on_converged=False, warm_start=True)
    assert fr.last_info["num_rendered"] < 0.6 * n_cold


@pytest.mark.parametrize("W,H", [(1920, 1080),       # 8 160 tiles
                                 (2576, 1616)])      # 16 261 tiles (more than a workgroup's LDS could hold bounds for)
def test_native_loop_on_large_images_matches_python_loop(W, H):
    from gs_localization_amd import pipelines as PL
    sc = S.small(P=30000, W=W, H=H, sh_degree=2, seed=8, scale_med=0.03)
    model, bg, view, init = _setup(sc)
    cfg = PL.TRACKING_CONFIG
    vp1, vp2 = view(), view()
    R1, T1, _ = PL.gradient_decent(vp1, cfg, init[:3, :3].clone(), init[:3, 3].clone(), model, PL.PipelineParams(), bg, iters=6)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    R2, T2, info = fr.refine(vp2, cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=6)
    assert info["iters"] == 6
    assert torch.allclose(R1, R2, atol=2e-5), (R1 - R2).abs().max()
    assert torch.allclose(T1, T2, atol=2e-5), (T1 - T2).abs().max()


def test_speculation_on_a_half_empty_scene():
    """Tiles that never saturate (sky, holes in the map) have no finite depth bound: nothing is ever dropped from them,
    so an unsaturated pixel there is not a failed speculation.  Half of this image is dense, the other half sparse and
    faint; the loop must speculate without a single redone forward and agree with complete lists."""
    from gs_localization_amd import pipelines as PL
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
                               speculative=spec)
        out[spec] = (R.clone(), T.clone(), info, fr.alpha.clone())
    a = out[False][3]
    assert float((a[..., sc.W // 2 + 16:] < 0.9).float().mean()) > 0.5       # the right half really is unsaturated
    assert float((a[..., :sc.W // 2 - 16] > 0.999).float().mean()) > 0.9     # and the left half saturated
    assert out[True][2]["fallbacks"] == 0
    assert out[True][2]["num_rendered"] < 0.8 * out[False][2]["num_rendered"]
    assert torch.allclose(out[True][0], out[False][0], atol=2e-6) and torch.allclose(out[True][1], out[False][1], atol=2e-6)
