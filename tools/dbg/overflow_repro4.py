import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
seed = 8
rng = np.random.default_rng(100 + seed)
W, H = int(rng.integers(150, 260)), int(rng.integers(100, 200))
sc = S.small(P=60000, W=W, H=H, sh_degree=3, seed=int(rng.integers(1 << 30)), scale_med=0.2)
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
init = torch.tensor(S.se3_exp(rng.normal(size=6) * 0.01), dtype=torch.float32, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
for K in (1, 2, 3):
    for name, kw in (("plain", dict(speculative=False)), ("spec", dict(speculative=True))):
        print("----", name, "K", K, flush=True); sys.stderr.flush()
        vp = PL.make_frame(sc, model, dev, bg)
        R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=K, stop_on_converged=False,
                               warm_start=False, lean_min_P=1, count_instances=True, flags=_lib.REFINE_DETERMINISTIC | _lib.REFINE_LOG_REDO, **kw)
        torch.cuda.synchronize()
        print({k: v for k, v in info.items() if k in ("fallbacks", "host_redos", "lean_iters", "num_rendered")}, flush=True)
