"""The steady state of the native loop and nothing else (one frame, speculative binning unless LOOP_PLAIN=1): the command the
rocprofv3 counter passes of tools/profile_round.sh / tools/profile_r04.sh run, so that per-launch averages are not mixed with
the other legs of bench.py.  SCENE = s_1m_640 (default) | s_800k_chess | s_3m_cam | s_3m_cam_1024 | s_50k_fern."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = getattr(S, os.environ.get("SCENE", "s_1m_640"))(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg)
init = PL.perturbed_start(1000, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
# (LOOP_PLAIN=1: without depth speculation -- complete lists in every iteration)
spec = not os.environ.get("LOOP_PLAIN")
fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=3, stop_on_converged=False, speculative=spec)
torch.cuda.synchronize(); t0 = time.perf_counter()
fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, stop_on_converged=False, count_instances=True,
          speculative=spec)
torch.cuda.synchronize()
print(sc.name, "iterations", iters, "it/s %.0f" % (iters / (time.perf_counter() - t0)), fr.last_info)
