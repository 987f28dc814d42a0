import sys, os, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
lib = _lib.load()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
kind = sys.argv[1]
sc = S.VARIANTS[kind]() if kind in S.VARIANTS else S.s_1m_640()
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg)
init = PL.perturbed_start(1000, device=dev)
for flags, tag in ((0, "lean"), (_lib.REFINE_NO_LEAN, "nolean"), (_lib.REFINE_NO_LEAN | _lib.REFINE_SH_SEPARATE, "nolean+shsep")):
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=5, stop_on_converged=False, flags=flags)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=50, stop_on_converged=False, flags=flags, count_instances=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.gsr_profile_enable((1 << nk) - 1)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=20, stop_on_converged=False, flags=flags)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    print(kind, tag, "it/s %.0f" % (50 / dt), {k: info[k] for k in ("fallbacks", "lean_iters", "num_rendered")}, {names[i]: (round(1e3 * ms[i] / max(cnt[i], 1), 1), cnt[i]) for i in range(nk) if ms[i] > 0}, flush=True)
