"""Sanity at a map size beyond the BASELINE configs (8 M Gaussians: 2.4 GB of geometry workspace, 1.5 GB of SH rows -- offsets past
2^31 bytes): the deterministic loop with and without speculation, bit for bit, and the rates."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
P = int(os.environ.get("GAUSSIANS", 8_000_000))
sc = S._draw("S-big", P, 640, 480, 525.0, 525.0, 0.5, 6.0, 0.006, 0.6, 3, 0)
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
init = PL.perturbed_start(3, 0.02, 1.0, device=dev)
fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
outs = []
for name, kw in (("spec", {}), ("plain", dict(speculative=False))):
    vp = PL.make_frame(sc, model, dev, bg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=12, stop_on_converged=False, warm_start=False,
                           flags=_lib.REFINE_DETERMINISTIC, count_instances=True, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    outs.append({"R": R.clone(), "T": T.clone(), "color": fr.color.clone(), "g_m3d": fr.g_m3d.clone(), "g_sh": fr.g_sh.clone(), "g_tau": fr.g_tau.clone(), "radii": fr.radii.clone()})
    print(name, "%.1f it/s" % (12 / dt), {k: info[k] for k in ("fallbacks", "lean_iters", "num_rendered", "host_redos")}, "geometry MB", _lib.load().gsr_geometry_bytes(P) >> 20, flush=True)
bad = [k for k in outs[0] if not torch.equal(outs[0][k], outs[1][k])]
print("bit-identical" if not bad else ("MISMATCH " + str(bad)), "| finite:", bool(torch.isfinite(outs[0]["g_sh"]).all()), "| alpha mean %.3f" % float(fr.alpha.mean()))
