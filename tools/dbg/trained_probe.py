"""Round 6: how does the native loop behave on a TRAINED map (tests/trained_map.py)?  Trains (or loads PLY=path), then: one frame --
per-kernel times, failed forwards, host redos; F frames in flight for F in 1, 2, 4, 8, 16.
usage: python tools/dbg/trained_probe.py [steps] [world] [p0] [p1]      (PLY=/tmp/x.ply to reuse a map written by an earlier run)"""
import os, sys, time, threading, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
import bench
from gs_localization_amd import _lib, scenes as S
from tests import replay as RP, trained_map as TM
a = [int(x) for x in sys.argv[1:]]
steps, world_P, P0, P1 = (a + [7000, 1_000_000, 200_000, 800_000][len(a):])[:4]
path = os.environ.get("PLY") or "/tmp/gsr_probe_map/point_cloud.ply"
if os.path.exists(path):
    world = S.s_room_640(P=world_P, seed=0)
else:
    world, rep = TM.train_room_map(path, steps=steps, world_P=world_P, P0=P0, P1=P1)
    print("trained:", rep)
lib = _lib.load(); dev = torch.device("cuda:0"); bg = torch.zeros(3, device=dev)
gmap = RP.GaussianMap.from_ply(path, device=dev)
wmap = RP.GaussianMap.from_scene(world, device=dev, requires_grad=False)
n = lambda t: t.detach().cpu().numpy()
sc = n(gmap.get_scaling); op = n(gmap.get_opacity).reshape(-1)
print("map: P = %d; scale median %.4f, max-axis 99th percentile %.3f; anisotropy (max/min) median %.1f; opacity < 0.05: %.2f, > 0.9: %.2f" % (
    sc.shape[0], float(np.median(sc)), float(np.quantile(sc.max(1), 0.99)), float(np.median(sc.max(1) / sc.min(1))), float((op < 0.05).mean()), float((op > 0.9).mean())))
rng = np.random.default_rng(5)
F = 16
frames, inits = [], []
for f in range(F):
    gt = S.se3_exp(np.concatenate([rng.uniform(-0.25, 0.25, 3), np.radians(rng.uniform(-10, 10, 3))]))
    dt = rng.normal(size=3); dt *= 0.02 / np.linalg.norm(dt); dr = rng.normal(size=3); dr *= np.radians(1.0) / np.linalg.norm(dr)
    inits.append(torch.tensor(S.se3_exp(np.concatenate([dt, dr])) @ gt, dtype=torch.float32, device=dev))
    frames.append(TM.world_frame(world, wmap, gt, f, dev, bg))
refs = [RP.FusedRefiner(gmap, world.H, world.W, device=dev) for _ in range(F)]
call = lambda s, f, K=50, **kw: refs[s].refine(frames[f], RP.TRACKING_CONFIG, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=K, stop_on_converged=False, **kw)
call(0, 1, 5); call(0, 0)
torch.cuda.synchronize(); t0 = time.perf_counter()
_, _, info = call(0, 0, count_instances=True)
torch.cuda.synchronize(); el = time.perf_counter() - t0
kms, cnt = bench._profile_ms(lib, lambda: call(0, 0), 2)
print("one frame, K = 50: %.0f it/s;" % (50 / el), {k: info[k] for k in ("fallbacks", "host_redos", "lean_iters", "num_rendered")}, "kernels us / iteration:", {k: round(1e3 * v / 50, 1) for k, v in kms.items() if v > 0},
      "launches per call:", {k: v // 2 for k, v in cnt.items() if v})
print("seg stats", refs[0].seg_stats())
streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
for nf in (1, 2, 4, 8, 16):
    stats = [None] * nf
    def worker(s):
        with torch.cuda.stream(streams[s]):
            _, _, inf = call(s, s)
            stats[s] = (inf["fallbacks"], inf["host_redos"])
            streams[s].synchronize()
    for s in range(nf):
        with torch.cuda.stream(streams[s]):
            call(s, (s + 1) % F, 5)
    torch.cuda.synchronize()
    th = [threading.Thread(target=worker, args=(s,)) for s in range(nf)]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("%2d in flight: %.0f it/s; (failed forwards, host redos) per frame: %s" % (nf, nf * 50 / el, stats))
