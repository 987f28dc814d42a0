"""What does k_preprocess_lean do in the SECOND iteration of a cold call (the first speculative one, bounds from a complete-list forward)?
Timing build: candidates / instances per wave, cycles per phase.  GSR_LIB_PATH=build_ab/libgsr_timing.so SCENE=s_1m_640_object python tools/dbg/lean_cold.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
lib = _lib.load(); dev = torch.device("cuda:0"); bg = torch.zeros(3, device=dev)
sc = getattr(S, os.environ.get("SCENE", "s_1m_640_object"))()
model = PL.GaussianMap.from_scene(sc, device=dev)
vp = PL.make_frame(sc, model, dev, bg); init = PL.perturbed_start(1000, device=dev)
fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
out = (C.c_ulonglong * 64)()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
for iters in (2, 3, 6):
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=1, stop_on_converged=False, warm_start=False)
    torch.cuda.synchronize(); lib.gsr_debug_timing(out)
    lib.gsr_profile_enable((1 << nk) - 1)
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=iters, stop_on_converged=False, warm_start=False, count_instances=True)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    lib.gsr_debug_timing(out); v = [int(x) for x in out]
    nw = (sc.P + 255) // 256
    print("iters", iters, {k: info[k] for k in ("fallbacks", "lean_iters", "num_rendered")}, "preprocess launches %d, %.1f us in all" % (cnt[names.index("preprocess_fwd")], 1e3 * ms[names.index("preprocess_fwd")]))
    lab = ["bounds", "conservative", "exact", " geometry", " walk+appends", " survivors", " SH", "appended", "appended to unbounded tiles", "lifetime", "candidates", "instances"]
    print("   lean totals per wave over its launches:", {l: round(v[32 + i] / nw) for i, l in enumerate(lab) if l})
