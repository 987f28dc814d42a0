"""Native loop on S-1M-640, single frame, plain vs speculative: wall time per iteration and the per-kernel HIP-event
times (ms per iteration) from the library's own profiler."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
lib = _lib.load(); dev = torch.device("cuda:0")
sc = S.s_1m_640(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
with torch.no_grad():
    pkg = PL.render(vp, model, bg)
vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
init = torch.tensor(S.se3_exp([0.01, 0.01, 0.01, 0.01, 0.0, 0.0]), dtype=torch.float32, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
import time
MARGIN = tuple(float(x) for x in os.environ["MARGIN"].split(",")) if "MARGIN" in os.environ else None
for spec in [False, True, False, True]:
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=3, stop_on_converged=False, speculative=spec, bound_margin=MARGIN, count_instances=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=100, stop_on_converged=False, speculative=spec, bound_margin=MARGIN, count_instances=True)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 100
    info = dict(fr.last_info)
    lib.gsr_profile_enable((1 << nk) - 1)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=40, stop_on_converged=False, speculative=spec, bound_margin=MARGIN, count_instances=True)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    d = {names[i]: round(ms[i] / 40, 4) for i in range(nk)}        # ms per iteration (all launches of that kernel)
    print("spec", spec, "wall ms/iter %.4f" % (wall * 1e3), info, d, "sum %.4f" % sum(d.values()), flush=True)
