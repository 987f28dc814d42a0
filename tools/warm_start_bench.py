"""Frame-sequence mode: per-frame time of a 30-iteration refinement with a cold start (first iteration binned with the
global sorts) and with the speculation warm-started from the previous frame's depth bounds (FusedRefiner warm_start)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
with torch.no_grad():
    pkg = PL.render(vp, model, bg)
vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
rng = np.random.default_rng(0)
inits = []
for k in range(20):          # a "sequence": start poses scattered 1-2 cm / 0.5-1 deg around the map pose
    tau = np.concatenate([rng.normal(size=3) * 0.008, rng.normal(size=3) * 0.008])
    inits.append(torch.tensor(S.se3_exp(tau), dtype=torch.float32, device=dev))
for warm in (False, True, False, True):
    fr.refine(vp, PL.TRACKING_CONFIG, inits[0][:3, :3].clone(), inits[0][:3, 3].clone(), bg, iters=30, stop_on_converged=False)
    torch.cuda.synchronize(); t0 = time.perf_counter(); fb = 0
    for i0 in inits:
        fr.refine(vp, PL.TRACKING_CONFIG, i0[:3, :3].clone(), i0[:3, 3].clone(), bg, iters=30, stop_on_converged=False, warm_start=warm)
        fb += fr.last_info["fallbacks"]
    torch.cuda.synchronize(); el = (time.perf_counter() - t0) / len(inits)
    print(f"warm_start={warm}: {el*1e3:.3f} ms per 30-iteration frame ({30/el:.0f} it/s), redone forwards {fb}", flush=True)
