"""per-step wall time and allocator state of train.py-style steps (a leak shows as `reserved` growing every step)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gs_localization_amd import rasterizer as RZ
from tests.train_replay import TrainReplay
P = int(os.environ.get("P", 1_500_000))
tr = TrainReplay(P0=P, P1=P, W=1296, H=840, densify_from=10**9)
for it in range(1, 41):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(it)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(it, "ms %.2f" % (1e3 * dt), "reserved GB %.1f" % (torch.cuda.memory_reserved() / 2**30), "mallocs", st["num_device_alloc"], "frees", st["num_device_free"], flush=True)
