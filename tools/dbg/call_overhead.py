"""Per-call cost of FusedRefiner.refine on S-1M-640: wall time for several iteration counts (intercept = what a call costs
beyond its iterations), and the host time of the Python part alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gs_localization_amd import scenes as S
from tests import replay as PL
dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
with torch.no_grad():
    pkg = PL.render(vp, model, bg)
vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
init = torch.tensor(S.se3_exp([0.01, 0.01, 0.01, 0.01, 0.0, 0.0]), dtype=torch.float32, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
kw = dict(stop_on_converged=False, speculative=True)
for warm in (False, True):
    for iters in (1, 2, 5, 20, 50):
        for _ in range(3):
            fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, warm_start=warm, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        N = 20
        for _ in range(N):
            fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, warm_start=warm, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
        print(f"warm_start={warm} iters {iters:3d}: {1e3 * dt:.3f} ms per call", flush=True)

import cProfile, pstats, io
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(50):
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, **kw)
torch.cuda.synchronize()
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(18); print(st.getvalue()[:4000])
import ctypes as C
from gs_localization_amd import _lib
lib = _lib.load()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
lib.gsr_profile_enable((1 << nk) - 1)
for _ in range(10):
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, **kw)
torch.cuda.synchronize()
ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
print("GPU kernel time per 1-iteration call (us):", {names[i]: round(1e3 * ms[i] / 10, 1) for i in range(nk)}, "sum %.1f" % (1e3 * sum(ms) / 10))
