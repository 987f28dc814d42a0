"""per-kernel HIP-event times of package (A)'s forward + backward at the train config (1296x840, SH1), P = 0.2 / 1.5 M;
argv: optional list of WxH:P cases instead"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gs_localization_amd import _lib
from tests.train_replay import TrainReplay
lib = _lib.load()
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
cases = [(1296, 840, 200_000), (1296, 840, 1_500_000)]
if len(sys.argv) > 1:
    cases = [(int(c.split(":")[0].split("x")[0]), int(c.split(":")[0].split("x")[1]), int(c.split(":")[1])) for c in sys.argv[1:]]
for W, H, P in cases:
    tr = TrainReplay(P0=P, P1=P, W=W, H=H, densify_from=10**9)
    for it in range(1, 6): tr.step(it)
    torch.cuda.synchronize()
    lib.gsr_profile_enable((1 << nk) - 1)
    N = 20
    for it in range(6, 6 + N): tr.step(it)
    torch.cuda.synchronize()
    ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
    print("%dx%d" % (W, H), P, "R", tr.last_R if hasattr(tr, "last_R") else "", {names[i]: round(1e3 * ms[i] / N, 1) for i in range(nk)}, "us per step; sum %.1f" % (1e3 * sum(ms) / N), flush=True)
    del tr; torch.cuda.empty_cache()
