"""-m gpu : the drop-in packages call gsr_forward_speculative (include/gsr.h): a render that follows a nearby render
reuses that one's per-tile depth bounds and skips the global sorts.  The speculation is verified on the device, so the
outputs must be those of the plain forward -- bit for bit for the images, radii and n_touched (same lists up to each
pixel's termination, same summation order), to summation-order noise for the gradients -- whether the guess holds
(nearby pose), misses (far pose, other scene) or is switched off."""
import math
import os

import numpy as np
import pytest
import torch

from gs_localization_amd import scenes as S, rasterizer as RZ
from tests import util as U

pytestmark = pytest.mark.gpu


def _w2c(tau):
    return S.se3_exp(np.asarray(tau, np.float64))


def _run(sc, w2c, pose, spec, grads):
    old = os.environ.get("GSR_SPECULATION")
    os.environ["GSR_SPECULATION"] = "1" if spec else "0"
    try:
        return U.hip_run(sc, U.scene_inputs(sc, w2c), grads, pose=pose)
    finally:
        if old is None:
            del os.environ["GSR_SPECULATION"]
        else:
            os.environ["GSR_SPECULATION"] = old


def _same(a, b, pose):
    for k in ("color", "depth", "alpha", "radii"):
        assert np.array_equal(a[k], b[k]), k
    if pose:
        assert np.array_equal(a["n_touched"], b["n_touched"])


def _close_grads(ga, gb, pose):
    for k in ("means3D", "means2D", "opacities", "sh", "scales", "rotations"):
        assert U.rel_l1(ga[k], gb[k]) <= 2e-6, k
    if pose:
        assert U.rel_l1(ga["tau"], gb["tau"]) <= 2e-6


@pytest.mark.parametrize("pose", [False, True])
def test_nearby_render_is_speculative_and_identical(pose):
    sc = S.small(P=30000, W=160, H=128, sh_degree=3, seed=21, scale_med=0.04)
    grads = U.random_grads(sc, seed=3)
    RZ._spec_cache.clear()
    first = _w2c([0, 0, 0, 0, 0, 0])
    near = _w2c([0.004, -0.003, 0.002, 0.002, -0.001, 0.0015])
    _run(sc, first, pose, True, grads)                       # no bounds yet: complete lists, records the bounds
    assert RZ.speculation_counters() == (0, 0)
    o_s, g_s = _run(sc, near, pose, True, grads)             # guess from the first render
    assert RZ.speculation_counters() == (1, 0)
    o_p, g_p = _run(sc, near, pose, False, grads)            # plain gsr_forward
    _same(o_s, o_p, pose)
    _close_grads(g_s, g_p, pose)
    o_s2, _ = _run(sc, near, pose, True, grads)              # and again, now from the bounds a speculative render recorded
    assert RZ.speculation_counters() == (2, 0)
    _same(o_s2, o_p, pose)


def test_missed_guess_is_redone_and_backs_off():
    sc = S.small(P=30000, W=160, H=128, sh_degree=2, seed=22, scale_med=0.04)
    other = S.small(P=30000, W=160, H=128, sh_degree=2, seed=23, scale_med=0.04)       # same image size, other map,
    other.means3D *= np.float32(1.5); other.scales *= np.float32(1.5)                   # 1.5x further away: the bounds are useless
    grads = U.random_grads(sc, seed=4)
    RZ._spec_cache.clear()
    _run(sc, _w2c([0] * 6), True, True, grads)
    o_s, g_s = _run(other, _w2c([0] * 6), True, True, grads)
    v, m = RZ.speculation_counters()
    assert (v, m) == (0, 1)
    o_p, g_p = _run(other, _w2c([0] * 6), True, False, grads)
    _same(o_s, o_p, True)
    _close_grads(g_s, g_p, True)
    # ever deeper maps (scaled about the camera): every guess misses; after two misses in a row the guess is tried less and less often
    import copy
    for i in range(12):
        deeper = copy.deepcopy(other)
        deeper.means3D *= np.float32(1.2 ** (i + 1))          # the same picture, everything 1.2x further away each time
        deeper.scales *= np.float32(1.2 ** (i + 1))
        _run(deeper, _w2c([0] * 6), True, True, None)
    v, m = RZ.speculation_counters()
    assert v == 0 and 3 <= m <= 5, (v, m)
    other = deeper
    # a nearby render still succeeds as soon as it is tried again
    for _ in range(10):
        o, _ = _run(other, _w2c([0.002, 0, 0, 0, 0.001, 0]), True, True, None)
    assert RZ.speculation_counters()[0] >= 1
    o_p, _ = _run(other, _w2c([0.002, 0, 0, 0, 0.001, 0]), True, False, None)
    _same(o, o_p, True)


def test_python_loop_uses_the_speculation_and_matches_the_plain_loop():
    """The reference-style loop on the drop-in package: same poses with and without the speculation."""
    from tests import replay as PL
    sc = S.small(P=30000, W=160, H=128, sh_degree=3, seed=11, scale_med=0.04)
    dev = torch.device("cuda:0")
    model = PL.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    init = torch.tensor(S.se3_exp(np.array([0.01, -0.008, 0.006, 0.004, -0.003, 0.005])), dtype=torch.float32, device=dev)
    res = {}
    for spec in (False, True):
        os.environ["GSR_SPECULATION"] = "1" if spec else "0"
        RZ._spec_cache.clear()
        vp = PL.make_frame(sc, model, dev, bg)
        R, T, _ = PL.python_loop(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=12)
        res[spec] = (R.detach().clone(), T.detach().clone(), RZ.speculation_counters())
    os.environ.pop("GSR_SPECULATION", None)
    assert res[False][2] == (0, 0)
    assert res[True][2][0] >= 10 and res[True][2][1] == 0
    assert torch.allclose(res[True][0], res[False][0], atol=2e-6) and torch.allclose(res[True][1], res[False][1], atol=2e-6)


def test_randomised_scenes_and_pose_walks():
    """tools/fuzz_speculation.py on 40 random cases: odd image sizes, sparse / faint / half-empty scenes, pose walks with
    jumps, both packages, and the native loop with and without speculation."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CASES="40", SEED="11")
    env.pop("GSR_SPECULATION", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_speculation.py")], env=env, cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "40 cases ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_spec_state_contract_through_the_c_abi():
    """gsr_forward_speculative with state == NULL is gsr_forward; a state sized for one image size is refused for another
    (GSR_E_INVALID + message) instead of being read out of bounds."""
    import ctypes as C
    from gs_localization_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    sc = S.small(P=2000, W=64, H=48, sh_degree=1, seed=31)
    cam = U.scene_inputs(sc)
    t = lambda a: torch.tensor(np.ascontiguousarray(a, np.float32), device=dev)
    means, opac, shs, scales, rots = t(sc.means3D), t(sc.opacities), t(sc.shs), t(sc.scales), t(sc.rotations)
    bg, view, proj, campos = t(sc.bg), t(cam["view"]), t(cam["proj"]), t(cam["campos"])

    def call(state, W, H):
        bufs = [RZ._Workspace(dev) for _ in range(3)]
        color = torch.empty((3, H, W), device=dev); depth = torch.empty((1, H, W), device=dev); alpha = torch.empty((1, H, W), device=dev)
        radii = torch.empty(sc.P, dtype=torch.int32, device=dev)
        rc = lib.gsr_forward_speculative(state, bufs[0].fn, bufs[0].ctx, bufs[1].fn, bufs[1].ctx, bufs[2].fn, bufs[2].ctx, sc.P, 1, 4, bg.data_ptr(), W, H,
                                         means.data_ptr(), shs.data_ptr(), None, opac.data_ptr(), scales.data_ptr(), 1.0, rots.data_ptr(),
                                         None, view.data_ptr(), proj.data_ptr(), campos.data_ptr(), sc.tanfovx, sc.tanfovy, 0,
                                         color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), radii.data_ptr(), 0, None,
                                         torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return rc, color
    rc0, c0 = call(None, sc.W, sc.H)
    assert rc0 > 0
    st = _lib.SpecState()
    buf = torch.empty(int(lib.gsr_spec_state_bytes(sc.W, sc.H)), dtype=torch.uint8, device=dev)
    st.device_buffer = buf.data_ptr()
    rc1, c1 = call(C.byref(st), sc.W, sc.H)
    rc2, c2 = call(C.byref(st), sc.W, sc.H)
    assert rc1 == rc0 and st.valid == 1 and st.last_speculative == 1 and st.n_speculative == 1
    assert torch.equal(c0, c1) and torch.equal(c0, c2)
    rc3, _ = call(C.byref(st), sc.W + 16, sc.H)
    assert rc3 < 0 and b"another image size" in lib.gsr_last_error()


def test_per_camera_states_and_their_memory_budget(monkeypatch):
    """Package (A) with the speculation switched on keeps one state per camera (the storage address of the view matrix tensor), a new
    camera's state starts from the bounds of the view rendered last, and states beyond GSR_SPEC_CACHE_MB are dropped oldest first.
    Whatever the cache does, every render equals the plain one."""
    import torch
    import diff_gaussian_rasterization as pkg
    monkeypatch.setenv("GSR_SPECULATION", "1")
    sc = S.small(P=20000, W=160, H=128, sh_degree=2, seed=41, scale_med=0.04)
    dev = "cuda:0"
    t = lambda a: torch.tensor(np.asarray(a, np.float32), device=dev)
    means3D, opac, shs, scales, rots, bg = t(sc.means3D), t(sc.opacities), t(sc.shs), t(sc.scales), t(sc.rotations), t(sc.bg)
    cams = []
    for k in range(4):                                       # four persistent cameras, a few millimetres apart
        c = U.scene_inputs(sc, _w2c([0.003 * k, -0.002 * k, 0.001 * k, 0.001 * k, 0.0, 0.0005 * k]))
        cams.append(dict(view=t(c["view"]), proj=t(c["proj"]), campos=t(c["campos"])))

    def render(c):
        rs = pkg.GaussianRasterizationSettings(image_height=sc.H, image_width=sc.W, tanfovx=sc.tanfovx, tanfovy=sc.tanfovy, bg=bg, scale_modifier=1.0,
                                               viewmatrix=c["view"], projmatrix=c["proj"], sh_degree=sc.sh_degree, campos=c["campos"],
                                               prefiltered=False, debug=False)
        with torch.no_grad():
            return pkg.GaussianRasterizer(rs)(means3D=means3D, means2D=torch.zeros_like(means3D), opacities=opac, shs=shs, colors_precomp=None,
                                              scales=scales, rotations=rots, cov3D_precomp=None)
    monkeypatch.setenv("GSR_SPECULATION", "0")
    plain = [render(c) for c in cams]
    monkeypatch.setenv("GSR_SPECULATION", "1")
    RZ._spec_cache.clear()
    for rnd in range(3):
        for c, ref in zip(cams, plain):
            out = render(c)
            for a, b in zip(out, ref):
                assert torch.equal(a, b)
    assert len(RZ._spec_cache.states) == 4                   # one state per camera
    v, m = RZ.speculation_counters()
    assert v >= 8 + 3 and m == 0                             # every revisit speculated; so did the first visits of cameras 2-4 (seeded)
    # a budget of two states: the cache never holds more, and the renders stay right
    one = next(iter(RZ._spec_cache.states.values()))[1].numel()
    monkeypatch.setenv("GSR_SPEC_CACHE_MB", str(max(1, (2 * one + (1 << 20) - 1) >> 20)))
    RZ._spec_cache.clear()
    for rnd in range(2):
        for c, ref in zip(cams, plain):
            out = render(c)
            assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            assert RZ._spec_cache.bytes <= (int(os.environ["GSR_SPEC_CACHE_MB"]) << 20)
    assert len(RZ._spec_cache.states) <= 2
