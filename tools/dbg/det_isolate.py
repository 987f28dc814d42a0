"""Isolate why sixteen deterministic calls in flight differ from the same calls one at a time (tools/dbg/inflight_early_exit.py)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as RP
dev = torch.device("cuda:0"); bg = torch.zeros(3, device=dev)
sc = getattr(S, os.environ.get("SCENE", "s_room_640"))(P=int(os.environ.get("P", "400000")))
gmap = RP.GaussianMap.from_scene(sc, device=dev)
F = int(os.environ.get("F", "16"))
STOP = os.environ.get("STOP", "1") == "1"
frames = [RP.make_frame(sc, gmap, dev, bg, uid=f, mask=os.environ.get("MASK", "reference")) for f in range(F)]
inits = [RP.perturbed_start(1000 + f, float(os.environ.get("TR", "0.004")), float(os.environ.get("ROT", "0.2")), device=dev) for f in range(F)]
DET = _lib.REFINE_DETERMINISTIC
def call(fr, f):
    for t_ in (frames[f].exposure_a, frames[f].exposure_b):          # (refine() hands the refined exposure back in the frame: start every call from zero)
        t_.data = torch.zeros(1, device=dev)
    R, T, inf = fr.refine(frames[f], RP.TRACKING_CONFIG, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=(30 if STOP else 6), converged_threshold=2.2e-3,
                          stop_on_converged=STOP, warm_start=False, flags=DET, lean_min_P=int(os.environ.get("LEAN_MIN_P", "0")), speculative=os.environ.get("SPEC", "1") == "1")
    torch.cuda.synchronize()
    return dict(R=R.clone(), T=T.clone(), color=fr.color.clone(), nt=fr.n_touched.clone(), radii=fr.radii.clone(), tau=fr.g_tau.clone(), iters=inf["iters"], fb=inf["fallbacks"])
def same(a, b):
    return {k: (torch.equal(a[k], b[k]) if torch.is_tensor(a[k]) else a[k] == b[k]) for k in a}
solo1 = [call(RP.FusedRefiner(gmap, sc.H, sc.W, device=dev), f) for f in range(F)]
solo2 = [call(RP.FusedRefiner(gmap, sc.H, sc.W, device=dev), f) for f in range(F)]
print(dict((k, os.environ.get(k)) for k in ("SCENE", "P", "MASK", "TR", "ROT", "LEAN_MIN_P", "SPEC", "STOP")), "fallbacks", [g["fb"] for g in solo1][:4]); print("solo vs solo (fresh refiners):", sum(all(same(a, b).values()) for a, b in zip(solo1, solo2)), "of", F, "identical;", same(solo1[0], solo2[0]))
if os.environ.get("QUICK"): sys.exit(0)
one = RP.FusedRefiner(gmap, sc.H, sc.W, device=dev)
solo3 = [call(one, f) for f in range(F)]
print("solo vs solo (ONE refiner reused, cold starts):", sum(all(same(a, b).values()) for a, b in zip(solo1, solo3)), "of", F, "identical;", same(solo1[1], solo3[1]))
refs = [RP.FusedRefiner(gmap, sc.H, sc.W, device=dev) for _ in range(F)]
streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
got = [None] * F
def worker(s):
    with torch.cuda.stream(streams[s]):
        got[s] = call(refs[s], s)
th = [threading.Thread(target=worker, args=(s,)) for s in range(F)]
[x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
print("in flight (fresh refiners) vs solo:", sum(all(same(a, b).values()) for a, b in zip(solo1, got)), "of", F, "identical;", same(solo1[0], got[0]), "iters", [g["iters"] for g in got], "fallbacks", [g["fb"] for g in got], "solo fb", [g["fb"] for g in solo1])
