"""EXPERIMENT: what does a spatially sorted map (3D Morton order of the means, as a loader of a static map could produce) buy the
native loop?  SCENE as tools/loop_only.py; SORT=0|1; LOOP_PLAIN=1 for complete lists.  Prints it/s and the per-kernel HIP-event times."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
sc = getattr(S, os.environ.get("SCENE", "s_1m_640"))(); H, W = sc.H, sc.W
def morton3(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.minimum(((p - lo) / np.maximum(hi - lo, 1e-9) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    code = np.zeros(len(p), np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code
if os.environ.get("SORT", "0") == "1":
    perm = np.argsort(morton3(sc.means3D.astype(np.float64)), kind="stable")
    for k in ("means3D", "scales", "rotations", "opacities", "shs"):
        setattr(sc, k, np.ascontiguousarray(getattr(sc, k)[perm]))
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg)
init = PL.perturbed_start(1000, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
lib = _lib.load()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
for spec in (True, False):
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False, speculative=spec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 100
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=N, stop_on_converged=False, speculative=spec)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.gsr_profile_enable((1 << nk) - 1)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=40, stop_on_converged=False, speculative=spec)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    print(sc.name, "SORT", os.environ.get("SORT", "0"), "spec" if spec else "plain", "it/s %.0f" % (N / dt),
          {names[i]: round(1e3 * ms[i] / 40, 1) for i in range(nk) if ms[i] > 0}, flush=True)
