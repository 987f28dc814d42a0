"""Diagnostic: what a whole 50-iteration refinement call costs for query poses scattered around the map (the frames of
tools/localize_split.py), next to the steady-state iteration rate: per-call time, forwards redone after a failed speculation,
instances binned by the last forward.  Poses at the edge of the synthetic map's coverage leave most tiles unsaturated: those
tiles have no depth bound, carry their complete lists, and the loop runs at the exact path's rate or below."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S
from tests import replay as RP
dev = torch.device("cuda:0")
sc = S._draw("S-chess-split", 800_000, 640, 480, 525.0, 525.0, 0.5, 6.0, 0.01, 0.6, 3, 0)
gmap = RP.GaussianMap.from_scene(sc, device=dev); bg = torch.zeros(3, device=dev); proj = RP.intrinsics_projection(sc, dev)
fr_ = RP.FusedRefiner(gmap, sc.H, sc.W, device=dev)
full_mask = torch.ones((1, sc.H, sc.W), dtype=torch.bool, device=dev)
def setup(f):
    rng = np.random.default_rng(7000 + f)
    gt = S.se3_exp(np.concatenate([rng.uniform(-0.3, 0.3, 3), np.radians(rng.uniform(-10, 10, 3))]))
    dt = rng.normal(size=3); dt *= 0.03 / np.linalg.norm(dt); dr = rng.normal(size=3); dr *= math.radians(2.0) / np.linalg.norm(dr)
    return gt, S.se3_exp(np.concatenate([dt, dr])) @ gt
def observe(f, gt):
    fr = RP.QueryFrame(f, proj, sc, dev, gt_w2c=torch.tensor(gt, dtype=torch.float32, device=dev))
    g = torch.tensor(gt, dtype=torch.float32, device=dev); fr.update_RT(g[:3, :3].clone(), g[:3, 3].clone())
    with torch.no_grad(): obs = RP.render(fr, gmap, bg)
    fr.original_image, fr.depth = obs["render"].detach().clone(), obs["depth"].detach()[0].clone(); fr.grad_mask = full_mask
    return fr
frames = [(observe(f, setup(f)[0]), setup(f)) for f in range(12)]
torch.cuda.synchronize()
for it in (50, 50, 200):
    T = dict(setup=0.0, refine=0.0, errs=0.0)
    for fr, (gt, init) in frames:
        t0 = time.perf_counter()
        i0 = torch.tensor(init, dtype=torch.float32, device=dev); R0, T0 = i0[:3, :3].clone(), i0[:3, 3].clone()
        t1 = time.perf_counter()
        R, Tt, info = fr_.refine(fr, RP.TRACKING_CONFIG, R0, T0, bg, iters=it)
        t2 = time.perf_counter()
        te, re = RP.pose_errors(gt[:3, :3], gt[:3, 3], R.detach().cpu().numpy(), Tt.detach().cpu().numpy())
        t3 = time.perf_counter()
        T["setup"] += t1 - t0; T["refine"] += t2 - t1; T["errs"] += t3 - t2
    print(it, "iterations:", {k: round(1e3 * v / len(frames), 3) for k, v in T.items()}, "ms per frame; refine it/s", round(it * len(frames) / T["refine"]))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for fr, (gt, init) in frames:
    i0 = torch.tensor(init, dtype=torch.float32, device=dev)
    fr_.refine(fr, RP.TRACKING_CONFIG, i0[:3, :3].clone(), i0[:3, 3].clone(), bg, iters=50)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
for fr, (gt, init) in frames[:6]:
    i0 = torch.tensor(init, dtype=torch.float32, device=dev)
    t0 = time.perf_counter()
    R, Tt, info = fr_.refine(fr, RP.TRACKING_CONFIG, i0[:3, :3].clone(), i0[:3, 3].clone(), bg, iters=50, count_instances=True)
    print(round(1e3 * (time.perf_counter() - t0), 2), "ms", {k: v for k, v in info.items() if not torch.is_tensor(v) and not isinstance(v, dict)})
os.environ["GSR_DEBUG_TILES"] = "1"
fr, (gt, init) = frames[0]
i0 = torch.tensor(init, dtype=torch.float32, device=dev)
fr_.refine(fr, RP.TRACKING_CONFIG, i0[:3, :3].clone(), i0[:3, 3].clone(), bg, iters=50)
