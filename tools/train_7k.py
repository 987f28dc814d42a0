"""BASELINE.json config 4 as written: train.py's step for 7 000 iterations on the synthetic garden (tests/train_replay.py) with the
real cadence of gaussian_splatting/train.py:142-152 -- densification every 100 iterations from 500 (here until 7 000), opacity reset at
3 000 -- through package (A) and the fused loss epilogue.  Reports wall time, mean ms per step (overall and per thousand), the number of
times P changed, peak device memory over the run and after the last change of P (a workspace leak would show as growth), and that
every loss was finite.  usage: python tools/train_7k.py [steps] [P0] [P1]   (one JSON line on stdout)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.train_replay import TrainReplay


def run(steps=7000, P0=200_000, P1=1_500_000, densify_from=500, interval=100, reset=3000, W=1296, H=840, log_every=1000):
    tr = TrainReplay(P0=P0, P1=P1, W=W, H=H, densify_from=densify_from, densification_interval=interval, densify_until=steps, opacity_reset_interval=reset)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    losses = torch.zeros(steps, device=tr.dev)
    sizes, marks, changes = [], [], 0
    peak_after_last_change, last_change_it = 0, 0
    t0 = time.perf_counter()
    t_mark = t0
    for it in range(1, steps + 1):
        P_before = tr.P
        losses[it - 1] = tr.step(it)
        if tr.P != P_before:
            changes += 1
            last_change_it = it
        if it == last_change_it + 20 and changes:          # (the allocator has settled: from here on nothing may grow)
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
        if it % log_every == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            marks.append({"iteration": it, "P": tr.P, "ms_per_step": 1e3 * (now - t_mark) / log_every})
            t_mark = now
        sizes.append(tr.P)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    tail_peak = torch.cuda.max_memory_allocated()
    lo = losses.cpu().numpy()
    return {"steps": steps, "wall_s": wall, "mean_ms_per_step": 1e3 * wall / steps, "P_first": sizes[0], "P_last": sizes[-1], "P_changes": changes,
            "last_P_change_at": last_change_it, "per_thousand": marks, "losses_finite": bool(np.isfinite(lo).all()),
            "loss_first_last": [float(lo[0]), float(lo[-1])], "peak_memory_after_last_P_change_MiB": tail_peak / 2**20,
            "memory_allocated_at_end_MiB": torch.cuda.memory_allocated() / 2**20,
            "workload": f"train.py step x {steps}, {W}x{H}, SH1, white background, densify every {interval} from {densify_from}, opacity reset at {reset}"}


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)), flush=True)
