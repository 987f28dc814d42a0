"""Round 6 stress: sixteen frames in flight on S-room-640, every call leaving by the EARLY EXIT (frozen forward + the closing n_touched pass),
five rounds; then every frame's images / radii / n_touched against a single-frame call of the same frame (deterministic flag: bit for bit)."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as RP
dev = torch.device("cuda:0"); bg = torch.zeros(3, device=dev)
sc = S.s_room_640(P=400_000)
gmap = RP.GaussianMap.from_scene(sc, device=dev)
F = 16
frames = [RP.make_frame(sc, gmap, dev, bg, uid=f) for f in range(F)]
inits = [RP.perturbed_start(1000 + f, 0.004, 0.2, device=dev) for f in range(F)]
refs = [RP.FusedRefiner(gmap, sc.H, sc.W, device=dev) for _ in range(F)]
streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
def call(s, f, flags=0):
    for t_ in (frames[f].exposure_a, frames[f].exposure_b):          # (refine() hands the refined exposure back in the frame: start every call from zero)
        t_.data = torch.zeros(1, device=dev)
    return refs[s].refine(frames[f], RP.TRACKING_CONFIG, inits[f][:3, :3].clone(), inits[f][:3, 3].clone(), bg, iters=30, converged_threshold=2.2e-3,
                          stop_on_converged=True, warm_start=False, flags=flags)
out = [None] * F
def worker(s, flags):
    with torch.cuda.stream(streams[s]):
        R, T, inf = call(s, s, flags)
        out[s] = (inf["iters"], inf["converged"], inf["fallbacks"], refs[s].color.clone(), refs[s].n_touched.clone(), refs[s].radii.clone(), R.clone(), T.clone())
        streams[s].synchronize()
for rnd in range(5):
    flags = _lib.REFINE_DETERMINISTIC if (rnd == 4 or os.environ.get("ALWAYS_DET")) else 0
    th = [threading.Thread(target=worker, args=(s, flags)) for s in range(F)]
    t0 = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; torch.cuda.synchronize()
    print("round %d: %.1f ms; iterations used %s; converged %d of %d; failed forwards %s" % (rnd, 1e3 * (time.perf_counter() - t0), [o[0] for o in out], sum(o[1] for o in out), F, [o[2] for o in out]), flush=True)
# last round ran under the deterministic option: a single-frame call of each frame must give the same bits
bad = 0
for s in range(F):
    fr = RP.FusedRefiner(gmap, sc.H, sc.W, device=dev)
    for t_ in (frames[s].exposure_a, frames[s].exposure_b):
        t_.data = torch.zeros(1, device=dev)
    R, T, inf = fr.refine(frames[s], RP.TRACKING_CONFIG, inits[s][:3, :3].clone(), inits[s][:3, 3].clone(), bg, iters=30, converged_threshold=2.2e-3, stop_on_converged=True,
                          warm_start=False, flags=_lib.REFINE_DETERMINISTIC)
    torch.cuda.synchronize()
    parts = dict(color=torch.equal(fr.color, out[s][3]), n_touched=torch.equal(fr.n_touched, out[s][4]), radii=torch.equal(fr.radii, out[s][5]), R=torch.equal(R, out[s][6]),
                 T=torch.equal(T, out[s][7]), iters=inf["iters"] == out[s][0])
    same = all(parts.values())
    if not same and bad < 3:
        print("  frame", s, parts, "iters", inf["iters"], out[s][0], "fallbacks", inf["fallbacks"], out[s][2], "max |dcolor| %.3g" % float((fr.color - out[s][3]).abs().max()),
              "dR %.3g" % float((R - out[s][6]).abs().max()), "n_touched diff", int((fr.n_touched - out[s][4]).abs().sum()))
    bad += 0 if same else 1
    del fr
print("deterministic: sixteen in flight against one at a time:", "bit for bit" if bad == 0 else "%d frames DIFFER" % bad)
