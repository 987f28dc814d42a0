"""Randomised check of the depth speculation: many small random scenes (sizes, densities, opacities, SH degrees, image
shapes with partial tiles), each rendered (a) through the drop-in package at a sequence of nearby / far poses with and
without gsr_forward_speculative -- images, radii and n_touched must be bit-identical -- and (b) through the native loop
with and without speculation -- same poses.  Prints the number of cases, verified / missed guesses and redone forwards."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S, rasterizer as RZ
from tests import replay as PL
from tests import util as U
dev = torch.device("cuda:0")
N = int(os.environ.get("CASES", 120))
rng = np.random.default_rng(int(os.environ.get("SEED", 1)))
tot_v = tot_m = tot_fb = noisy = 0
for case in range(N):
    W = int(rng.integers(40, 260)); H = int(rng.integers(40, 200))
    P = int(rng.choice([200, 2000, 20000, 60000]))
    deg = int(rng.integers(0, 4))
    sc = S.small(P=P, W=W, H=H, sh_degree=deg, seed=int(rng.integers(1 << 30)), scale_med=float(rng.choice([0.01, 0.03, 0.08, 0.2])))
    if rng.random() < 0.4:          # faint scene: many tiles never saturate
        sc.opacities *= np.float32(rng.choice([0.05, 0.3]))
    if rng.random() < 0.3:          # hole: drop the Gaussians of one image half
        keep = sc.means3D[:, 0] * rng.choice([-1, 1]) < 0.2
        sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs = (np.ascontiguousarray(x[keep]) for x in
                                                                       (sc.means3D, sc.scales, sc.rotations, sc.opacities, sc.shs))
    if sc.P == 0:
        continue
    pose = bool(rng.integers(2))
    # (a) drop-in packages: a walk of poses, small steps with an occasional jump
    RZ._spec_cache.states.clear()
    tau = np.zeros(6)
    for step in range(6):
        tau = tau + rng.normal(size=6) * (0.003 if rng.random() < 0.8 else 0.2)
        w2c = S.se3_exp(tau)
        outs = []
        for spec in ("1", "0"):
            os.environ["GSR_SPECULATION"] = spec
            o, _ = U.hip_run(sc, U.scene_inputs(sc, w2c), None, pose=pose)
            outs.append(o)
        for k in ("color", "depth", "alpha", "radii") + (("n_touched",) if pose else ()):
            if not np.array_equal(outs[0][k], outs[1][k]):
                print("MISMATCH", case, step, k, W, H, P, deg); sys.exit(1)
    v, m = RZ.speculation_counters(); tot_v += v; tot_m += m
    os.environ["GSR_SPECULATION"] = "1"
    # (b) native loop
    if case % 3 == 0:
        model = PL.GaussianMap.from_scene(sc, device=dev)
        bg = torch.zeros(3, device=dev)
        def view():
            vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
            with torch.no_grad():
                pkg = PL.render(vp, model, bg)
            vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
            return vp
        init = torch.tensor(S.se3_exp(rng.normal(size=6) * 0.01), dtype=torch.float32, device=dev)
        fr = PL.FusedRefiner(model, H, W, device=dev)
        res = []
        for spec in (False, True):
            R, T, info = fr.refine(view(), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=10, stop_on_converged=bool(case % 2), speculative=spec)
            res.append((R.clone(), T.clone(), info))
        tot_fb += res[1][2]["fallbacks"]
        if res[0][2]["iters"] != res[1][2]["iters"] or not (torch.allclose(res[0][0], res[1][0], atol=5e-6) and torch.allclose(res[0][1], res[1][1], atol=5e-6)):
            # ill-conditioned case or a real difference?  The plain loop against ITSELF shows how far the order of the fp32 atomics
            # alone moves this scene's pose in ten iterations
            R2, T2, info2 = fr.refine(view(), PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=10, stop_on_converged=bool(case % 2), speculative=False)
            d_spec = max((res[0][0] - res[1][0]).abs().max().item(), (res[0][1] - res[1][1]).abs().max().item())
            d_self = max((res[0][0] - R2).abs().max().item(), (res[0][1] - T2).abs().max().item())
            brief = lambda i: {k: v for k, v in i.items() if k in ("iters", "converged", "fallbacks", "host_redos", "lean_iters")}
            print("LOOP DIFFERENCE case", case, W, H, P, deg, brief(res[0][2]), brief(res[1][2]), "speculative vs plain %.2e, plain vs plain %.2e" % (d_spec, d_self), flush=True)
            if res[0][2]["iters"] != res[1][2]["iters"] or d_spec > max(5e-6, 4.0 * d_self):
                print("LOOP MISMATCH"); sys.exit(1)
            noisy += 1
print(f"{N} cases ok: drop-in guesses verified {tot_v}, missed {tot_m}; native loop forwards redone {tot_fb}; ill-conditioned loop cases (plain loop differs from itself as much) {noisy}")
