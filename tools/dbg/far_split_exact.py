"""Bit-for-bit check at scale: query frames of the synthetic split (800 k Gaussians, poses up to 0.3 m / 10 deg from the map's reference
view: tiles that never saturate, failed and retried speculations, held tiles), each refined for up to 50 iterations by the
deterministic loop with and without speculation (and warm-started from the previous frame's bounds, as the split driver does)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as RP
dev = torch.device("cuda:0")
P = int(os.environ.get("GAUSSIANS", 800000)); NF = int(os.environ.get("FRAMES", 8))
spread = (float(os.environ.get("SPREAD_M", 0.3)), float(os.environ.get("SPREAD_DEG", 10.0)))
sc = S._draw("S-chess-split", P, 640, 480, 525.0, 525.0, 0.5, 6.0, 0.01, 0.6, 3, 0)
gmap = RP.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
proj = RP.intrinsics_projection(sc, dev)
full_mask = torch.ones((1, sc.H, sc.W), dtype=torch.bool, device=dev)
def frame_setup(f):
    rng = np.random.default_rng(7000 + f)
    gt = S.se3_exp(np.concatenate([rng.uniform(-spread[0], spread[0], 3), np.radians(rng.uniform(-spread[1], spread[1], 3))]))
    off = rng.uniform(0.1, 1.0)
    dt = rng.normal(size=3); dt *= 0.05 * off / np.linalg.norm(dt)
    dr = rng.normal(size=3); dr *= math.radians(3.0 * off) / np.linalg.norm(dr)
    return gt, S.se3_exp(np.concatenate([dt, dr])) @ gt
def observe(f, gt):
    fr = RP.QueryFrame(f, proj, sc, dev, gt_w2c=torch.tensor(gt, dtype=torch.float32, device=dev))
    g = torch.tensor(gt, dtype=torch.float32, device=dev)
    fr.update_RT(g[:3, :3].clone(), g[:3, 3].clone())
    with torch.no_grad():
        obs = RP.render(fr, gmap, bg)
    fr.original_image, fr.depth = obs["render"].detach().clone(), obs["depth"].detach()[0].clone()
    fr.grad_mask = full_mask
    return fr
ref_s = RP.FusedRefiner(gmap, sc.H, sc.W, device=dev)      # speculative, warm-started from frame to frame (the driver's use)
ref_p = RP.FusedRefiner(gmap, sc.H, sc.W, device=dev)      # complete lists
keys = ("m2d", "conic", "opac", "col", "m3d", "cov", "sh", "scale", "rot", "tau")
nbad = 0
for f in range(NF):
    gt, init = frame_setup(f)
    i0 = torch.tensor(init, dtype=torch.float32, device=dev)
    outs = []
    for name, r, kw in (("spec", ref_s, dict()), ("plain", ref_p, dict(speculative=False))):
        R, T, info = r.refine(observe(f, gt), RP.TRACKING_CONFIG, i0[:3, :3].clone(), i0[:3, 3].clone(), bg, iters=50, flags=_lib.REFINE_DETERMINISTIC, **kw)
        torch.cuda.synchronize()
        o = {"R": R.clone(), "T": T.clone(), "color": r.color.clone(), "alpha": r.alpha.clone(), "radii": r.radii.clone(), "n_touched": r.n_touched.clone()}
        for k in keys: o["g_" + k] = getattr(r, "g_" + k).clone()
        outs.append((o, info))
    bad = {k: float((outs[0][0][k].double() - outs[1][0][k].double()).abs().max()) for k in outs[0][0] if not torch.equal(outs[0][0][k], outs[1][0][k])}
    b = lambda i: {k: i[k] for k in ("iters", "converged", "fallbacks", "host_redos", "lean_iters") if k in i}
    print("frame", f, "MISMATCH" if (bad or outs[0][1]["iters"] != outs[1][1]["iters"]) else "ok", b(outs[0][1]), b(outs[1][1]), bad, flush=True)
    nbad += bool(bad)
print("frames differing:", nbad)
