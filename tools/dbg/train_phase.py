"""Diagnostic (timing build: GSR_TIMING=1 python gs_localization_amd/build.py): per-phase shader clocks of k_render_fwd /
k_render_bwd_mfma in the train step (tests/train_replay.py, 1296x840, SH1).  argv: P (default 1 500 000)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gs_localization_amd import _lib
from tests.train_replay import TrainReplay
lib = _lib.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
tr = TrainReplay(P0=P, P1=P, densify_from=10**9)
for it in range(1, 4): tr.step(it)
torch.cuda.synchronize()
out = (C.c_ulonglong * 64)()
assert lib.gsr_debug_timing(out) == 0, "not a GSR_TIMING build"
N = 10
for it in range(4, 4 + N): tr.step(it)
torch.cuda.synchronize()
lib.gsr_debug_timing(out)
v = [int(x) for x in out]
nw = ((tr.W + 15) // 16) * ((tr.H + 15) // 16) * 4 * N
for name, base, labels in (("k_render_fwd", 0, ["bins load", "sort + writeback", "batch top (barrier_and)", "staging gathers + barrier", "compaction", "compositing loop",
                                                 "after loop", "epilogue (incl. the wait below)", "  of which: waiting for the tile's other waves", "(wave lifetime)", "batches", "loop iterations"]),
                           ("k_render_bwd_mfma", 16, ["prologue", "batch top barrier", "staging + barrier", "compaction", "weights (8 splats)", "mfma + lds atomics",
                                                      "(loop exit)", "barrier after groups", "recombine + global atomics", "(wave lifetime)", "batches", "list entries"])):
    print(name, "(cycles per wave, mean)")
    for i, l in enumerate(labels):
        print("  %-46s %10.0f" % (l, v[base + i] / nw))
