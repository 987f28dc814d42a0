"""-m gpu : the deterministic option (GSR_REFINE_DETERMINISTIC / debug bit 2 of gsr_backward; SURVEY.md section 5 asks for a determinism
option "for tests").  Every sum that crosses workgroups is added in 64-bit fixed point, so

  * two runs on the same inputs give the same BITS -- poses, loss, images, every gradient tensor;
  * the loop's variants give the same bits AMONG EACH OTHER: speculative lists (k_preprocess_lean, depth-bounded bins, device-side
    retries) against GSR_REFINE_NO_LEAN against complete lists in every iteration.  A list entry the speculation drops is one that no
    pixel of its tile blends, so its contribution to every sum is exactly zero: with order-independent sums nothing may differ.  The
    tolerance tests of test_gpu_lean.py / test_gpu_refine.py cannot see a discrepancy below the atomics' noise (1e-7 ... 1e-6 in
    the pose); this one can;
  * the results agree with the default mode's to rounding, and with the CPU oracle to the parity suite's tolerances.
"""
import os

import numpy as np
import pytest
import torch

from gs_localization_amd import _lib, scenes as S
from tests import util as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DET = _lib.REFINE_DETERMINISTIC
GRADS = ("m2d", "conic", "opac", "col", "m3d", "cov", "sh", "scale", "rot", "tau")


def _run(fr, vp, init, bg, iters, **kw):
    from tests import replay as PL
    kw.setdefault("warm_start", False)
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, stop_on_converged=False, **kw)
    torch.cuda.synchronize()
    out = dict(R=R.clone(), T=T.clone(), info=info, color=fr.color.clone(), depth=fr.depth.clone(), alpha=fr.alpha.clone(),
               n_touched=fr.n_touched.clone(), radii=fr.radii.clone(), loss=fr.loss_out.clone())
    for k in GRADS:
        out["g_" + k] = getattr(fr, "g_" + k).clone()
    return out


def _bit_equal(a, b, name, skip=()):
    bad = [k for k in a if k != "info" and k not in skip and not torch.equal(a[k], b[k])]
    detail = {k: float((a[k].double() - b[k].double()).abs().max()) for k in bad}
    assert not bad, (name, detail)


def _loop_variants(sc, K, seed, lean_min_P=0):
    from tests import replay as PL
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(seed, 0.02, 1.0, device=DEV)
    view = lambda: PL.make_frame(sc, model, DEV, bg)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    kw = dict(lean_min_P=lean_min_P) if lean_min_P else {}
    spec1 = _run(fr, view(), init, bg, K, flags=DET, **kw)
    spec2 = _run(PL.FusedRefiner(model, sc.H, sc.W, device=DEV), view(), init, bg, K, flags=DET, **kw)      # fresh workspaces
    nolean = _run(fr, view(), init, bg, K, flags=DET | _lib.REFINE_NO_LEAN, **kw)
    plain = _run(fr, view(), init, bg, K, flags=DET, speculative=False, **kw)
    default = _run(fr, view(), init, bg, K, flags=0, **kw)
    return spec1, spec2, nolean, plain, default


def _check_variants(spec1, spec2, nolean, plain, default, K, grad_bar=2e-4):
    assert spec1["info"]["lean_iters"] >= 1 and nolean["info"]["lean_iters"] == 0 and plain["info"]["lean_iters"] == 0, spec1["info"]
    _bit_equal(spec1, spec2, "two deterministic runs")
    _bit_equal(spec1, nolean, "speculative vs GSR_REFINE_NO_LEAN")
    # (n_touched of the last forward counts on the lists it has: identical, the last forward of every variant keeps complete radii)
    _bit_equal(spec1, plain, "speculative vs complete lists")
    # against the default mode: same numbers up to the rounding of the sums
    assert torch.allclose(spec1["R"], default["R"], atol=2e-6) and torch.allclose(spec1["T"], default["T"], atol=2e-6)
    for k in ("m3d", "sh", "opac", "scale", "rot", "tau"):
        a, b = spec1["g_" + k].cpu().numpy(), default["g_" + k].cpu().numpy()
        assert U.rel_l1(a, b) <= grad_bar, (k, U.rel_l1(a, b))


def test_loop_variants_agree_bit_for_bit_small_map():
    sc = S.small(P=90000, W=176, H=144, sh_degree=3, seed=14, scale_med=0.035)
    _check_variants(*_loop_variants(sc, 9, seed=6, lean_min_P=1), 9)


def test_loop_variants_agree_bit_for_bit_at_the_headline_size():
    """S-1M-640 (BASELINE.json configs[1]): 12 iterations, the product's own thresholds (the lean kernel runs: P >= 200 000)."""
    _check_variants(*_loop_variants(S.s_1m_640(), 12, seed=3), 12)


@pytest.mark.parametrize("name", ["object", "walls", "room"])
def test_loop_variants_agree_bit_for_bit_on_the_structured_scenes(name):
    """the structured variants of the headline scene (gs_localization_amd/scenes.py) in the bit-for-bit set, at full
    size -- speculative lists (with the depth bounds widened at discontinuities, verification failures and device-side retries these
    scenes provoke) against GSR_REFINE_NO_LEAN against complete lists.  The deterministic option never splits a tile; the default-mode
    run next to it does (S-room-640, object), hence the wider bar on its gradients (tests/test_gpu_split.py: what two loops whose pixels
    fall on different sides of the 1e-4 threshold differ by)."""
    _check_variants(*_loop_variants(S.VARIANTS[name](), 8, seed=3), 8, grad_bar=1e-3)


def test_diagnostic_switches_and_pose_only_mode_change_no_bit():
    """GSR_REFINE_NO_BALANCE (tile launch order), GSR_REFINE_SH_SEPARATE (k_sh_color behind the lean kernel instead of the fused
    colour) and a refiner without Gaussian gradients (pose only) against the default configuration."""
    from tests import replay as PL
    sc = S.small(P=90000, W=176, H=144, sh_degree=3, seed=14, scale_med=0.035)
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(6, 0.02, 1.0, device=DEV)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    base = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 9, flags=DET, lean_min_P=1)
    for name, fl in (("no balance", _lib.REFINE_NO_BALANCE), ("sh separate", _lib.REFINE_SH_SEPARATE)):
        _bit_equal(_run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 9, flags=DET | fl, lean_min_P=1), base, name)
    fr2 = PL.FusedRefiner(model, sc.H, sc.W, device=DEV, gaussian_grads=False)
    R, T, info = fr2.refine(PL.make_frame(sc, model, DEV, bg), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=9,
                            stop_on_converged=False, warm_start=False, lean_min_P=1, flags=DET)
    torch.cuda.synchronize()
    assert torch.equal(R, base["R"]) and torch.equal(T, base["T"]) and torch.equal(fr2.g_tau, base["g_tau"]) and torch.equal(fr2.color, base["color"])


def test_image_with_more_tiles_than_the_fused_binning_kernel_takes():
    """1296 x 840 (the training configuration's size): 4 293 tiles, complete lists through count -> scan -> emit instead of
    k_preprocess_bin; speculative against complete lists."""
    from tests import replay as PL
    sc = S.small(P=200000, W=1296, H=840, sh_degree=1, seed=5, scale_med=0.02)
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(4, 0.02, 1.0, device=DEV)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    spec = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 8, flags=DET)
    plain = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 8, flags=DET, speculative=False)
    assert spec["info"]["lean_iters"] >= 1
    _bit_equal(spec, plain, "speculative vs complete lists at 1296 x 840")


def test_half_empty_scene_with_a_large_start_offset():
    """Half of the image dense, half sparse and faint (tiles without a depth bound), and a start 5 cm / 3 degrees off so that the view
    moves under the speculation (bounds that go stale, retried forwards).  The deterministic loop with and without speculation ends in
    the same bits."""
    from tests import replay as PL
    sc = S.small(P=60000, W=160, H=128, sh_degree=1, seed=14, scale_med=0.04)
    right = sc.means3D[:, 0] > 0
    keep = ~right | (np.arange(sc.P) % 40 == 0)
    sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs = (np.ascontiguousarray(x[keep]) for x in
                                                                   (sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs))
    sc.opacities[sc.means3D[:, 0] > 0] *= 0.2
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(9, 0.05, 3.0, device=DEV)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    spec = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 20, flags=DET, lean_min_P=1)
    plain = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 20, flags=DET, speculative=False)
    print("speculative:", {k: spec["info"][k] for k in ("fallbacks", "lean_iters", "host_redos") if k in spec["info"]})
    _bit_equal(spec, plain, "speculative vs complete lists, half-empty scene")


def test_overflowing_first_forward_does_not_hand_stale_bounds_to_its_retry():
    """Found by tools/fuzz_speculation.py under the deterministic option (round 3).  Splats 0.2 m wide: every tile's complete list
    overflows the fixed-capacity bins of the first forward, whose tiles then left the compositing kernel WITHOUT recording a depth
    bound; the speculative group enqueued behind it (the device-side retry) binned with whatever the buffers held -- lists that
    were not depth-prefixes -- passed its verification and moved the pose 1.4e-4 away from the plain loop's and the Python loop's.
    The overflowing tile now records "no bound".  A previous call on the same workspace provides the stale bounds."""
    from tests import replay as PL
    sc = S.small(P=60000, W=150, H=185, sh_degree=3, seed=1026614153, scale_med=0.2)
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = torch.tensor(S.se3_exp(np.array([0.020133432009333933, -0.011872109397651393, 0.006790385897898772, 0.01091599154672567,
                                            -0.019092896950391746, -0.002401738727523628])), dtype=torch.float32, device=DEV)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    for K in (1, 2, 4):
        plain = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, K, flags=DET, speculative=False)
        spec = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, K, flags=DET, lean_min_P=1)
        assert plain["info"]["fallbacks"] >= 1, plain["info"]          # the scene really overflows the bins
        _bit_equal(spec, plain, f"speculative vs complete lists, K = {K}")
    vp = PL.make_frame(sc, model, DEV, bg)
    Rp, Tp, _ = PL.python_loop(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=4)
    assert torch.allclose(spec["R"], Rp, atol=2e-6) and torch.allclose(spec["T"], Tp, atol=2e-6), \
        (float((spec["R"] - Rp).abs().max()), float((spec["T"] - Tp).abs().max()))


def test_concurrent_frames_and_warm_starts_bit_for_bit():
    """What bench.py and the split driver do -- several frames in flight on one GPU (one host thread and one stream each), every call
    warm-started from the bounds the previous frame left in its workspace -- against the same frames refined one after the other from
    cold starts: the same bits (the tolerance versions: test_gpu_refine.py)."""
    import threading
    from tests import replay as PL
    sc = S.small(P=40000, W=160, H=128, sh_degree=2, seed=13, scale_med=0.04)
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    F, rounds = 3, 3
    starts = [[torch.tensor(S.se3_exp(np.random.default_rng(90 + 10 * r + f).normal(size=6) * (0.01 if r < 2 else 0.15)), dtype=torch.float32, device=DEV)
               for f in range(F)] for r in range(rounds)]                     # (the last round starts far off: stale bounds)
    cold = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    want = [[_run(cold, PL.make_frame(sc, model, DEV, bg), starts[r][f], bg, 10, flags=DET, lean_min_P=1) for f in range(F)] for r in range(rounds)]
    frs = [PL.FusedRefiner(model, sc.H, sc.W, device=DEV) for _ in range(F)]
    streams = [torch.cuda.Stream(device=DEV) for _ in range(F)]
    vps = [[PL.make_frame(sc, model, DEV, bg) for f in range(F)] for r in range(rounds)]
    got = [[None] * F for _ in range(rounds)]
    torch.cuda.synchronize()

    def work(f):
        with torch.cuda.stream(streams[f]):
            for r in range(rounds):
                got[r][f] = _run(frs[f], vps[r][f], starts[r][f], bg, 10, flags=DET, lean_min_P=1, warm_start=(r > 0))
    ts = [threading.Thread(target=work, args=(f,)) for f in range(F)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    for r in range(rounds):
        for f in range(F):
            _bit_equal(got[r][f], want[r][f], f"round {r} frame {f}")


def test_dropin_backward_is_reproducible(monkeypatch):
    """The stateless packages: GSR_DETERMINISTIC=1 -> debug bit 2 of gsr_backward.  Two backward passes over the same forward give
    the same bits; against the default mode: rounding (the default mode is the one the parity suite checks against the CPU oracle)."""
    from tests import replay as PL
    sc = S.small(P=20000, W=128, H=96, sh_degree=3, seed=3, scale_med=0.05)
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(2, 0.02, 1.0, device=DEV)

    def grads():
        vp = PL.make_frame(sc, model, DEV, bg)
        vp.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
        for t in (model.get_xyz, model.get_features, model.get_opacity, model.get_scaling, model.get_rotation):
            t.grad = None
        pkg = PL.render(vp, model, bg)
        PL.tracking_loss(PL.TRACKING_CONFIG, pkg["render"], pkg["depth"], pkg["opacity"], vp).backward()
        torch.cuda.synchronize()
        return dict(m3d=model.get_xyz.grad.clone(), sh=model.get_features.grad.clone(), opac=model.get_opacity.grad.clone(),
                    scale=model.get_scaling.grad.clone(), rot=model.get_rotation.grad.clone(),
                    tau=torch.cat([vp.cam_trans_delta.grad, vp.cam_rot_delta.grad]).clone())

    default = grads()
    monkeypatch.setenv("GSR_DETERMINISTIC", "1")
    a, b = grads(), grads()
    for k in a:
        assert torch.equal(a[k], b[k]), k
        assert U.rel_l1(a[k].cpu().numpy(), default[k].cpu().numpy()) <= 2e-5, (k, U.rel_l1(a[k].cpu().numpy(), default[k].cpu().numpy()))


def test_backward_refuses_the_option_on_a_workspace_its_forward_did_not_size_for_it(monkeypatch):
    """The deterministic option's accumulator records are four times the default ones and the FORWARD sizes the geometry workspace:
    a backward with debug bit 2 behind a forward without it must be refused (GSR_E_INVALID), not write 144 P bytes past the buffer."""
    from gs_localization_amd import rasterizer as RZ
    sc = S.small(P=4000, W=96, H=64, sh_degree=1, seed=5, scale_med=0.05)
    cam = U.scene_inputs(sc, np.eye(4))
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=DEV)
    rs = RZ.GaussianRasterizationSettingsPose(image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=t(sc.bg), scale_modifier=1.0,
                                              viewmatrix=t(cam["view"]), projmatrix=t(cam["proj"]), projmatrix_raw=t(cam["proj_raw"]), sh_degree=1,
                                              campos=t(cam["campos"]), prefiltered=False, debug=False)
    e = torch.Tensor([]).to(DEV)
    monkeypatch.delenv("GSR_DETERMINISTIC", raising=False)
    R, color, radii, depth, alpha, nt, saved, consts = RZ._forward_impl(t(sc.means3D), t(sc.shs), e, t(sc.opacities), t(sc.scales), t(sc.rotations), e, rs, True)
    assert consts[4] is False
    need = dict(sh=True, scales=True, rotations=True)
    gi, gd, ga = torch.ones_like(color), torch.ones_like(depth), torch.zeros_like(alpha)
    with pytest.raises(RuntimeError, match="deterministic"):
        RZ._backward_impl(rs, R, saved, consts[:4] + (True,), gi, gd, ga, True, need)
    # ... and the pair that belongs together still works, in either mode
    out = RZ._backward_impl(rs, R, saved, consts, gi, gd, ga, True, need)
    assert torch.isfinite(out[8]).all()
    monkeypatch.setenv("GSR_DETERMINISTIC", "1")
    R2, color2, _, depth2, alpha2, _, saved2, consts2 = RZ._forward_impl(t(sc.means3D), t(sc.shs), e, t(sc.opacities), t(sc.scales), t(sc.rotations), e, rs, True)
    assert consts2[4] is True
    out2 = RZ._backward_impl(rs, R2, saved2, consts2, gi, gd, ga, True, need)
    assert U.rel_l1(out2[8].cpu().numpy(), out[8].cpu().numpy()) <= 1e-5


def test_deterministic_sums_have_the_range_for_large_splats_under_unit_pixel_gradients(monkeypatch):
    """Splats hundreds of pixels wide under white-noise pixel gradients of unit variance: the conic's per-tile moment sums reach 1e8 ...
    1e10 and cancel over the tiles (test_gpu_parity.py::test_large_images).  The first version of the option kept one 2^-40 word per
    sum and wrapped here; the (coarse, remainder) pair does not: same gradients as the default mode to the order noise of ITS float
    atomics, and bit-reproducible."""
    sc = S.small(P=3000, W=1280, H=720, sh_degree=1, seed=31, scale_med=0.03)
    cam = U.scene_inputs(sc, np.eye(4))
    grads = U.random_grads(sc, seed=1)
    _, g0 = U.hip_run(sc, cam, grads, pose=True)
    monkeypatch.setenv("GSR_DETERMINISTIC", "1")
    _, g1 = U.hip_run(sc, cam, grads, pose=True)
    _, g2 = U.hip_run(sc, cam, grads, pose=True)
    assert float(np.abs(g1["means2D"]).max()) > 50.0          # (the regime: sums far beyond what a mean loss produces)
    for k in ("means3D", "means2D", "opacities", "sh", "scales", "rotations", "tau"):
        assert np.array_equal(g1[k], g2[k]), k
        assert U.rel_l1(g1[k], g0[k]) <= 1e-4, (k, U.rel_l1(g1[k], g0[k]))


def test_non_finite_gradients_are_not_swallowed(monkeypatch):
    """An integer sum cannot hold a NaN: the deterministic backward counts non-finite addends per Gaussian and hands such a Gaussian
    NaN gradients -- the same Gaussians the float atomics of the default mode poison."""
    sc = S.small(P=5000, W=96, H=64, sh_degree=1, seed=7, scale_med=0.05)
    cam = U.scene_inputs(sc, np.eye(4))
    gc, gd, ga = U.random_grads(sc, seed=2)
    gc[1, 30, 40] = np.nan
    gd[0, 10, 70] = np.inf
    _, g0 = U.hip_run(sc, cam, (gc, gd, ga), pose=True)
    monkeypatch.setenv("GSR_DETERMINISTIC", "1")
    _, g1 = U.hip_run(sc, cam, (gc, gd, ga), pose=True)
    bad0, bad1 = ~np.isfinite(g0["means3D"]).all(axis=1), ~np.isfinite(g1["means3D"]).all(axis=1)
    assert bad0.sum() > 0 and np.array_equal(bad0, bad1), (int(bad0.sum()), int(bad1.sum()))
    ok = ~bad0
    assert U.rel_l1(g1["means3D"][ok], g0["means3D"][ok]) <= 1e-4
    assert not np.isfinite(g0["tau"]).all() and not np.isfinite(g1["tau"]).all()          # (the pose gradient sums over all of them)


def test_survivor_sublists_hold_a_fully_visible_map():
    """140 000 Gaussians, all of them in view: the survivor work lists at their fullest (every sub-list takes whole 1 024-Gaussian
    stretches, whichever kernel files the survivors: surv_cap).  Complete lists (k_preprocess_bin) against the speculative loop (other
    kernels fill the lists) and against the Python loop."""
    from tests import replay as PL
    sc = S.small(P=140000, W=320, H=240, sh_degree=1, seed=17, scale_med=0.02)
    sc.means3D[:, :2] *= np.float32(0.45)          # everything well inside the frustum
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(4, 0.01, 0.5, device=DEV)
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    plain = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 4, flags=DET, speculative=False, count_instances=True)
    spec = _run(fr, PL.make_frame(sc, model, DEV, bg), init, bg, 4, flags=DET, lean_min_P=1)
    assert int((plain["radii"] > 0).sum()) > 0.9 * sc.P          # (the premise: nearly every Gaussian is visible)
    _bit_equal(spec, plain, "speculative vs complete lists, fully visible map")
    Rp, Tp, _ = PL.python_loop(PL.make_frame(sc, model, DEV, bg), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=4)
    assert torch.allclose(plain["R"], Rp, atol=2e-6) and torch.allclose(plain["T"], Tp, atol=2e-6)


@pytest.mark.parametrize("spec", [True, False])
def test_gradient_rows_after_an_early_exit_are_those_of_the_last_stepped_iteration(spec):
    """The loop writes the gradients of the Gaussians' own parameters ONCE per call, from the records of the last iteration whose
    pose step ran (gsr::PreBwdArgs::role).  Three ways to get there must give the same bits -- the bits of a loop that writes the rows in
    every iteration (GSR_REFINE_GRADS_EVERY_ITERATION) --: (a) the iterations are used up (the
    host launches the final pass), (b) the loop converges early, a frozen forward at the final pose runs behind the last
    iteration and ITS chain-rule launch does the final pass -- from the other set of lists / records / splat records and the camera
    the pose step saved, (c) it converges in its very last iteration (no frozen forward; host).  A second call on the same
    workspaces must find them clean."""
    from tests import replay as PL
    sc = S.small(P=20000, W=160, H=120, sh_degree=3, seed=5, scale_med=0.03)
    model = PL.GaussianMap.from_scene(sc, device=DEV)
    bg = torch.zeros(3, device=DEV)
    init = PL.perturbed_start(0, device=DEV)
    view = lambda: PL.make_frame(sc, model, DEV, bg)
    cfg = PL.TRACKING_CONFIG

    def call(fr, iters, flags=DET, **kw):
        R, T, info = fr.refine(view(), cfg, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, flags=flags, speculative=spec,
                               warm_start=False, **kw)
        torch.cuda.synchronize()
        out = dict(R=R.clone(), T=T.clone(), info=info)
        for k in GRADS:
            out["g_" + k] = getattr(fr, "g_" + k).clone()
        return out

    # (b): a threshold the update norm falls below after a few iterations
    k, early, frB = None, None, None
    for thr in (2.0e-3, 1.8e-3, 1.5e-3, 1.2e-3, 1.0e-3):
        frB = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
        early = call(frB, 30, converged_threshold=thr)
        if early["info"]["converged"] and 2 <= early["info"]["iters"] <= 25:
            k = early["info"]["iters"]
            break
    assert k is not None, early["info"]
    # (a): exactly k iterations, no early exit
    frA = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    full = call(frA, k, stop_on_converged=False)
    assert float(full["g_m3d"].abs().sum()) > 0 and float(full["g_sh"].abs().sum()) > 0
    _bit_equal(full, early, "iterations used up / early exit with a frozen forward behind it")
    # ... and the same bits as a loop that writes the rows in EVERY iteration, the way a sequence of gsr_backward calls would
    # (GSR_REFINE_GRADS_EVERY_ITERATION): writing them once is not a different result, only less work nobody could have observed
    EVERY = DET | _lib.REFINE_GRADS_EVERY_ITERATION
    _bit_equal(full, call(PL.FusedRefiner(model, sc.H, sc.W, device=DEV), k, flags=EVERY, stop_on_converged=False), "rows once / rows in every iteration")
    _bit_equal(early, call(PL.FusedRefiner(model, sc.H, sc.W, device=DEV), 30, flags=EVERY, converged_threshold=thr), "rows once / every iteration, early exit")
    # (c): converges in the last iteration it was given
    frC = PL.FusedRefiner(model, sc.H, sc.W, device=DEV)
    last = call(frC, k, converged_threshold=thr)
    assert last["info"]["converged"] and last["info"]["iters"] == k
    _bit_equal(full, last, "iterations used up / convergence in the last one")
    # the workspaces of (b) and (c) serve another call like fresh ones
    init2 = PL.perturbed_start(7, device=DEV)
    def again(fr):
        R, T, info = fr.refine(view(), cfg, init2[:3, :3].clone(), init2[:3, 3].clone(), bg, iters=3, flags=DET, speculative=spec,
                               stop_on_converged=False, warm_start=False)
        torch.cuda.synchronize()
        out = dict(R=R.clone(), T=T.clone(), info=info)
        for kk in GRADS:
            out["g_" + kk] = getattr(fr, "g_" + kk).clone()
        return out
    fresh = again(PL.FusedRefiner(model, sc.H, sc.W, device=DEV))
    _bit_equal(fresh, again(frB), "second call on the workspaces of an early exit")
    _bit_equal(fresh, again(frC), "second call on the workspaces of a convergence in the last iteration")
