"""Times train.py-style steps (gaussian_splatting/train.py:71-161) on the drop-in package (A) through tests/train_replay.py:
fixed P = 0.2 / 0.8 / 1.5 M (SH degree 1, white background, 1296x840, random camera per step, grad_depth != 0), then the
whole S-train-garden schedule compressed (P growing 0.2 -> 1.5 M, densification every `interval` steps).
Prints ms per step and where it goes (rasterizer forward + activations / loss epilogue / backward / statistics + Adam)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gs_localization_amd import rasterizer as RZ
from tests.train_replay import TrainReplay, time_steps

W, H = int(os.environ.get("TW", 1296)), int(os.environ.get("TH", 840))


def main():
    for P in (200_000, 800_000, 1_500_000):
        RZ._spec_cache.clear()
        tr = TrainReplay(P0=P, P1=P, W=W, H=H, densify_from=10**9)
        r, _ = time_steps(tr, 1, 50, warm=5)
        print(f"train step {W}x{H} P={P:8d} SH1: {r['ms_per_step']:6.2f} ms/step; render() {r['render_fwd_ms']:.2f} ms of which the rasterizer forward {r['rasterizer_fwd_ms']:.2f} ms, "
              f"loss epilogue {r['loss_epilogue_ms']:.2f} ms, backward {r['backward_ms']:.2f} ms, densification stats + Adam {r['stats_and_adam_ms']:.2f} ms; "
              f"speculative forwards verified/missed {RZ.speculation_counters()}", flush=True)
        del tr
        torch.cuda.empty_cache()
    # the schedule: 700 steps, densification every 10 from step 50 on (65 changes of P)
    RZ._spec_cache.clear()
    tr = TrainReplay(P0=200_000, P1=1_500_000, W=W, H=H, densify_from=50, densification_interval=10, densify_until=700)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(1, 701):
        tr.step(it)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"S-train-garden schedule compressed to 700 steps (P 200000 -> {tr.P}, {tr.events} densifications): {el:.2f} s, {1e3 * el / 700:.2f} ms/step "
          f"including the densifications", flush=True)


if __name__ == "__main__":
    main()
