import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.VARIANTS["object"]()
bg = torch.zeros(3, device=dev)
model = PL.GaussianMap.from_scene(sc, device=dev)
frames = [PL.make_frame(sc, model, dev, bg, uid=u) for u in (0, 1)]
inits = [PL.perturbed_start(1000 + u, device=dev) for u in (0, 1)]
fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
def call(g, iters):
    R, T, info = fr.refine(frames[g], PL.TRACKING_CONFIG, inits[g][:3, :3].clone(), inits[g][:3, 3].clone(), bg, iters=iters, stop_on_converged=False, speculative=True,
                           flags=int(os.environ.get("FLAGS", "0")) | _lib.REFINE_LOG_REDO)
    return {k: info[k] for k in ("iters", "fallbacks", "host_redos", "lean_iters")}
for rep in range(3):
    print("frame1 x5 ", call(1, 5), flush=True)
    print("frame0 x20", call(0, 20), flush=True)
