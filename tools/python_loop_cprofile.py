"""cProfile of the reference-style Python loop on the drop-in pose package (host-bound: which functions hold the interpreter)."""
import sys, os, cProfile, pstats, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev); bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg); init = PL.perturbed_start(1000, device=dev)
PL.python_loop(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=20)
pr = cProfile.Profile(); pr.enable()
PL.python_loop(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), model, bg, iters=200)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
