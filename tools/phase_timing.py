"""Diagnostic: per-phase shader-clock totals inside k_render_fwd / k_render_bwd_mfma on S-1M-640 (native loop).
Needs the timing build:  GSR_TIMING=1 python gs_localization_amd/build.py && python tools/phase_timing.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gs_localization_amd import _lib, scenes as S
from tests import replay as PL
lib = _lib.load(); dev = torch.device("cuda:0")
sc = getattr(S, os.environ.get("SCENE", "s_1m_640"))(); H, W = sc.H, sc.W
model = PL.GaussianMap.from_scene(sc, device=dev)
bg = torch.zeros(3, device=dev)
vp = PL.QueryFrame(0, PL.intrinsics_projection(sc, dev), sc, dev)
with torch.no_grad():
    pkg = PL.render(vp, model, bg)
vp.original_image = pkg["render"].clone(); vp.depth = pkg["depth"][0].clone(); vp.grad_mask = PL.reference_mask(vp.original_image)
init = torch.tensor(S.se3_exp([0.01, 0.01, 0.01, 0.01, 0.0, 0.0]), dtype=torch.float32, device=dev)
fr = PL.FusedRefiner(model, H, W, device=dev)
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
ITERS = 40
nk = lib.gsr_profile_kernel_count(); names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=3, stop_on_converged=False, speculative=not os.environ.get("LOOP_PLAIN"))
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
assert lib.gsr_debug_timing(out) == 0, "not a GSR_TIMING build"
lib.gsr_profile_enable((1 << nk) - 1)
fr.refine(vp, PL.TRACKING_CONFIG, init[:3, :3].clone(), init[:3, 3].clone(), bg, iters=ITERS, stop_on_converged=False, speculative=not os.environ.get("LOOP_PLAIN"))
torch.cuda.synchronize()
ms = (C.c_double * nk)(); cnt = (C.c_longlong * nk)(); lib.gsr_profile_collect(ms, cnt); lib.gsr_profile_enable(0)
print({names[i]: (round(ms[i] / max(cnt[i], 1), 4), cnt[i]) for i in range(nk)})
lib.gsr_debug_timing(out)
v = [int(x) for x in out]
NT = ((W + 15) // 16) * ((H + 15) // 16)
nw = NT * 4 * ITERS        # waves (K7); the packed forward kernel has 2 per tile
def show(name, base, labels):
    print(name, "(cycles per wave, mean)")
    tot = 0
    for i, l in enumerate(labels):
        if l: print("  %-28s %10.0f" % (l, v[base + i] / nw)); tot += v[base + i] / nw if i < 9 else 0
    print("  %-28s %10.0f" % ("sum of phases", tot))
nw = NT * 4 * ITERS
show("k_render_fwd", 0, ["bins load", "sort + writeback", "batch top (barrier_and)", "staging gathers + barrier", "compaction", "compositing loop",
                         "after loop", "epilogue (incl. the wait below)", "  of which: waiting for the tile's other waves", "(wave lifetime)", "batches", "loop iterations"])
nw = NT * 4 * ITERS
show("k_render_bwd_mfma", 16, ["prologue", "batch top barrier", "staging + barrier", "compaction", "weights (8 splats)", "mfma + lds atomics",
                               "(loop exit)", "barrier after groups", "recombine + global atomics", "(wave lifetime)", "batches", "list entries"])
if os.environ.get("LOOP_PLAIN"):
    nw = (sc.P + 2047) // 2048 * 8 * (ITERS + 0)      # k_preprocess_bin: 8 waves per 2 048 Gaussians
    show("k_preprocess_bin", 32, ["geometry (preprocess_one x 4)", "barrier", "count walk", "barrier", "reserve (global atomics)", "barrier before a band", "emit walk (all bands)",
                                  "", "", "(wave lifetime)", "", ""])
nw = (sc.P + 255) // 256 * ITERS      # one wave per 256 Gaussians
show("k_preprocess_lean", 32, ["bounds -> LDS + barrier", "conservative pass (4 x 64 Gaussians)", "exact pass on the compacted candidates", "  geometry incl. hoisted loads",
                               "  footprint walk + appends", "  survivors' list + dirty rows", "  SH colour", "", "", "(wave lifetime)", "candidates", "instances walked"])
# (working waves of the chain-rule kernel: the ones that ran at least one round)
rounds = max(v[48 + 10], 1)
print("k_preprocess_bwd (cycles per ROUND of <= 64 Gaussians, mean over %d rounds = %.1f per launch; %d Gaussians per launch)" % (rounds, rounds / ITERS, v[48 + 11] / ITERS))
for i, l in enumerate(["fill: list -> records -> queue", "barrier, records again, SH rows in", "loads + covariance chain + mean2D", "SH backward",
                       "stores, scale / rotation, pose sums", "SH rows out, queue shift", "fp64 reduction + dL/dtau atomics (per wave)", "ticket", "pose step (last wave)"]):
    print("  %-44s %10.0f" % (l, v[48 + i] / (rounds if i < 6 else ITERS * (2048 if i < 8 else 1))))
# Is a kernel's duration its mean wave or its slowest one?  (rows = wave positions; all tiles of K6 / K7 are resident at once)
for name, base, launches in (("k_render_fwd", 0, ITERS), ("k_render_bwd_mfma", 16, ITERS), ("k_preprocess_lean", 32, ITERS), ("k_preprocess_bwd", 48, ITERS)):
    print("%-20s wave lifetime per launch (cycles): median %8.0f   99th percentile %8.0f   max %8.0f   (%d rows)" %
          (name, v[base + 14] / launches, v[base + 15] / launches, v[base + 12] / launches, v[base + 13]))
