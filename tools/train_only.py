"""BASELINE.json config 4 and nothing else: N train.py-style steps (tests/train_replay.py: package (A) forward at 1296x840, SH1,
white background, random camera per step; fused loss epilogue; backward; densification statistics; torch Adam) at a fixed number
of Gaussians -- the command the rocprofv3 passes of tools/profile_r04.sh run.  argv: steps [P]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.train_replay import TrainReplay
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1_500_000
tr = TrainReplay(P0=P, P1=P, densify_from=10**9)
for it in range(1, 4):
    tr.step(it)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(4, 4 + steps):
    tr.step(it)
torch.cuda.synchronize()
print("train steps", steps, "P", P, "ms/step %.3f" % (1e3 * (time.perf_counter() - t0) / steps))
