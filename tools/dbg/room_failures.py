"""EXPERIMENT: which speculations fail on S-room-640 (GSR_REFINE_LOG_REDO lines) and how the failures spread over a call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
sc = S.VARIANTS[kind]()
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg)
init = PL.perturbed_start(1000, device=dev)
fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
for margin in (None, (1.01, 0.01), (1.05, 0.05)):
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=50, stop_on_converged=False,
                           flags=_lib.REFINE_LOG_REDO, warm_start=False, bound_margin=margin)
    print("margin", margin, {k: info[k] for k in ("iters", "fallbacks", "host_redos", "lean_iters")}, flush=True)
