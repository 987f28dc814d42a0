import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev); bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg); init = PL.perturbed_start(1000, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=200, stop_on_converged=False, speculative=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("plain loop", round(200 / dt), "it/s", info["fallbacks"])
