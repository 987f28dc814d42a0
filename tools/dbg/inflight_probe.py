"""Round 6: F frames in flight on a structured scene (SCENE=s_room_640 | s_1m_640_walls | ...), with and without split tiles
(GSR_NO_SPLIT=1): it/s and failed forwards per frame.  The bench's scene_variants leg runs ONE frame; `value` runs the uniform cloud."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from gs_localization_amd import scenes as S
from tests import replay as RP
dev = torch.device("cuda:0"); bg = torch.zeros(3, device=dev)
sc = getattr(S, os.environ.get("SCENE", "s_room_640"))()
gmap = RP.GaussianMap.from_scene(sc, device=dev)
F = 16
frames = [RP.make_frame(sc, gmap, dev, bg, uid=f) for f in range(F)]
inits = [RP.perturbed_start(1000 + f, device=dev) for f in range(F)]
refs = [RP.FusedRefiner(gmap, sc.H, sc.W, device=dev) for _ in range(F)]
K = int(os.environ.get("K", "50"))
call = lambda s, f, k=K: refs[s].refine(frames[f], RP.TRACKING_CONFIG, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=k, stop_on_converged=False)
streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
for nf in (1, 2, 4, 8, 16):
    stats = [None] * nf
    def worker(s):
        with torch.cuda.stream(streams[s]):
            _, _, inf = call(s, s)
            stats[s] = inf["fallbacks"]
            streams[s].synchronize()
    for s in range(nf):
        with torch.cuda.stream(streams[s]):
            call(s, (s + 1) % F, 5)
    torch.cuda.synchronize()
    th = [threading.Thread(target=worker, args=(s,)) for s in range(nf)]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%s %s %2d in flight: %.0f it/s; failed forwards per frame: %s" % (sc.name, "NO_SPLIT" if os.environ.get("GSR_NO_SPLIT") else "split", nf, nf * K / el, stats), flush=True)
