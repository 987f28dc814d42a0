"""Repro hunt: scenes of large splats (bins that overflow their fixed capacity) -- deterministic speculative loop against deterministic
plain loop, bit for bit; prints the first tensors that differ and both infos."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
brief = lambda i: {k: v for k, v in i.items() if k in ("iters", "converged", "fallbacks", "host_redos", "lean_iters", "num_rendered")}
nbad = 0
for seed in range(int(os.environ.get("SEEDS", 12))):
    rng = np.random.default_rng(100 + seed)
    W, H = int(rng.integers(150, 260)), int(rng.integers(100, 200))
    sc = S.small(P=60000, W=W, H=H, sh_degree=3, seed=int(rng.integers(1 << 30)), scale_med=float(os.environ.get("SCALE", 0.2)))
    model = PL.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    init = torch.tensor(S.se3_exp(rng.normal(size=6) * 0.01), dtype=torch.float32, device=dev)
    fr = PL.FusedRefiner(model, H, W, device=dev)
    K = int(os.environ.get("K", 6))
    res = []
    for name, kw in (("plain", dict(speculative=False)), ("spec", dict(speculative=True)), ("plain2", dict(speculative=False))):
        vp = PL.make_frame(sc, model, dev, bg)
        R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=K, stop_on_converged=False,
                               warm_start=False, lean_min_P=1, flags=_lib.REFINE_DETERMINISTIC | int(os.environ.get("FLAGS", 0)), **kw)
        torch.cuda.synchronize()
        out = {"R": R.clone(), "T": T.clone(), "color": fr.color.clone(), "radii": fr.radii.clone(), "n_touched": fr.n_touched.clone(), "loss": fr.loss_out.clone(),
               "g_tau": fr.g_tau.clone(), "g_m3d": fr.g_m3d.clone()}
        res.append((name, out, info))
    for a, b in ((0, 1), (0, 2)):
        bad = {k: float((res[a][1][k].double() - res[b][1][k].double()).abs().max()) for k in res[a][1] if not torch.equal(res[a][1][k], res[b][1][k])}
        if bad:
            nbad += 1
            print("seed", seed, W, H, res[a][0], "vs", res[b][0], bad, brief(res[a][2]), brief(res[b][2]), flush=True)
    print("seed", seed, W, H, [brief(r[2]) for r in res], flush=True)
print("mismatching pairs:", nbad)
