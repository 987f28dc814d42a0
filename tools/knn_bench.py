"""Times distCUDA2 (gsr_dist2_knn3) on synthetic clouds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from simple_knn._C import distCUDA2
rng = np.random.default_rng(0)
for P, kind in ((100_000, "uniform"), (1_000_000, "uniform"), (1_000_000, "clustered")):
    if kind == "uniform":
        pts = rng.uniform(-5, 5, size=(P, 3)).astype(np.float32)
    else:
        c = rng.uniform(-10, 10, size=(64, 3)); pts = (c[rng.integers(0, 64, P)] + rng.normal(size=(P, 3)) * rng.uniform(0.05, 1.0, size=(P, 1))).astype(np.float32)
    t = torch.tensor(pts, device="cuda:0")
    distCUDA2(t); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): distCUDA2(t)
    torch.cuda.synchronize()
    print(f"distCUDA2 P={P} {kind}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
