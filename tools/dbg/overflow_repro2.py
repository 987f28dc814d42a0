import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
brief = lambda i: {k: v for k, v in i.items() if k in ("fallbacks", "host_redos", "lean_iters")}
for seed in (8, 4):
    rng = np.random.default_rng(100 + seed)
    W, H = int(rng.integers(150, 260)), int(rng.integers(100, 200))
    sc = S.small(P=60000, W=W, H=H, sh_degree=3, seed=int(rng.integers(1 << 30)), scale_med=0.2)
    model = PL.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    init = torch.tensor(S.se3_exp(rng.normal(size=6) * 0.01), dtype=torch.float32, device=dev)
    fr = PL.FusedRefiner(model, H, W, device=dev)
    for K in range(1, 7):
        vp = PL.make_frame(sc, model, dev, bg)
        Rp, Tp = PL.python_loop(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=K)[:2]
        row = []
        for name, kw in (("plain", dict(speculative=False)), ("spec", dict(speculative=True)), ("spec_nolean", dict(speculative=True, flags=_lib.REFINE_DETERMINISTIC | _lib.REFINE_NO_LEAN))):
            vp = PL.make_frame(sc, model, dev, bg)
            kw.setdefault("flags", _lib.REFINE_DETERMINISTIC)
            R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=K, stop_on_converged=False,
                                   warm_start=False, lean_min_P=1, **kw)
            torch.cuda.synchronize()
            row.append((name, float((R - Rp).abs().max()), float((T - Tp).abs().max()), brief(info), R.clone(), T.clone()))
        print("seed", seed, "K", K, [(n, "%.1e" % a, "%.1e" % b, i) for n, a, b, i, _, _ in row],
              "plain-spec %.1e" % float((row[0][4] - row[1][4]).abs().max()), "spec-nolean %.1e" % float((row[1][4] - row[2][4]).abs().max()), flush=True)
