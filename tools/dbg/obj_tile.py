import sys, os
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
    for p_ in (frames[g].exposure_a, frames[g].exposure_b):
        p_.data = torch.zeros_like(p_.data)
    R, T, info = fr.refine(frames[g], PL.TRACKING_CONFIG, inits[g][:3, :3].clone(), inits[g][:3, 3].clone(), bg, iters=iters, stop_on_converged=False, speculative=True,
                           flags=int(os.environ.get("FLAGS", "0")))
    torch.cuda.synchronize()
    w = fr.state[44:48].view(torch.int32).cpu().numpy()
    return {k: info[k] for k in ("iters", "fallbacks", "host_redos")}, int(w[0]), float(np.array([w[1]], np.int32).view(np.float32)[0]), "hold_after", int(w[2]), "hold word", hex(int(w[3])), [round(float(x), 7) for x in info["T_host"]]
for n in (1, 2, 3, 4, 8):
    call(1, 5)
    print(n, call(0, n), flush=True)
