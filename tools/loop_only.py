"""The steady state of the native loop and nothing else (one frame, S-1M-640, speculative binning): the command the rocprofv3
counter passes of tools/profile_round.sh run, so that per-launch averages are not mixed with the other legs of bench.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg)
init = PL.perturbed_start(1000, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
# (LOOP_PLAIN=1: without depth speculation -- complete lists, the exact-bin path, in every iteration)
fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, stop_on_converged=False, count_instances=True,
          speculative=not os.environ.get("LOOP_PLAIN"))
torch.cuda.synchronize()
print("iterations", iters, fr.last_info)
