"""EXPERIMENT: the native loop on a structured scene with and without split tiles -- it/s, kernel times, split statistics.
usage: python tools/dbg/split_probe.py [room|object|walls|base] [P]"""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gs_localization_amd import scenes as S, _lib
from tests import replay as PL
dev = torch.device("cuda:0")
lib = _lib.load()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
kind = sys.argv[1] if len(sys.argv) > 1 else "room"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
sc = S.VARIANTS[kind](P=P) if kind in S.VARIANTS else S.s_1m_640(P=P)
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.make_frame(sc, model, dev, bg)
init = PL.perturbed_start(1000, device=dev)
for flags, tag in ((_lib.REFINE_NO_SPLIT, "warm-up"), (0, "split"), (_lib.REFINE_NO_DILATE, "nodilate"), (_lib.REFINE_NO_SPLIT, "nosplit"), (0, "split")):
    fr = PL.FusedRefiner(model, sc.H, sc.W, device=dev)
    for p_ in (vp.exposure_a, vp.exposure_b):      # (refine() leaves the refined exposure in the camera: every variant starts from zero)
        p_.data = torch.zeros_like(p_.data)
    kw = dict(iters=5, stop_on_converged=False, flags=flags)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 50
    R, T, info = fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=N, stop_on_converged=False, flags=flags)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = fr.seg_stats() if flags != _lib.REFINE_NO_SPLIT else None
    lib.gsr_profile_enable((1 << nk) - 1)
    fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=20, stop_on_converged=False, flags=flags)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    print("%-8s %-8s it/s %6.0f" % (kind, tag, N / dt), {k: info[k] for k in ("fallbacks", "host_redos", "lean_iters")}, "seg", st,
          {names[i]: round(1e3 * ms[i] / 20, 1) for i in range(nk) if ms[i] > 0},
          "T", [round(float(x), 6) for x in info["T_host"]], flush=True)
