"""Rewrites the measured-numbers blocks of DESIGN.md and README.md (between the `numbers:begin` / `numbers:end` markers) from
gpurun_out/r05/bench_default.json (python bench.py) and bench_driver.json (the driver's command: --gpus 1 --steps 20 --warmup 5)."""
import json, os, re
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ld = lambda n: json.loads([l for l in open(os.path.join(R, "gpurun_out", "r05", n)) if l.startswith("{")][-1])
dd, dr = ld("bench_default.json"), ld("bench_driver.json")
cam, tr = dd["cam_step"]["per_size"], dd["train_step"]
h = dd["dropin_host_us_per_call"]
kp = [c["kernels_ms_per_iter_plain"] for c in cam]
k15 = tr["kernels_ms_per_step_1500000"]
us = lambda v: f"{1e3 * v:.0f}"
sv = dd["scene_variants"]["per_scene"]
svrow = lambda r: f'| {r["scene"]} | {r["R_eff_own_binning"] / 1e3:.0f} k | {r["speculative_iters_per_s"]:.0f} | {r["plain_iters_per_s"]:.0f} | {1e-9 * r["R_eff_own_binning"] * r["speculative_iters_per_s"]:.2f} | ' \
    + " / ".join(us(r["kernels_ms_per_iter_speculative"].get(k, 0.0)) for k in ("preprocess_fwd", "render_fwd", "render_bwd", "preprocess_bwd")) \
    + f' | {r["speculative_call_stats"]["fallbacks"]} | {r["pose_err_cm_deg_after_refinement"][0]:.2f} cm / {r["pose_err_cm_deg_after_refinement"][1]:.2f}° |'
base_row = f'| S-1M-640 (uniform cloud, `value`\'s scene) | {dd["config"]["R_eff_own_binning"] / 1e3:.0f} k | {dd["single_frame_iters_per_s"]:.0f} | {dd["plain_loop_iters_per_s"]:.0f} | {1e-9 * dd["config"]["R_eff_own_binning"] * dd["single_frame_iters_per_s"]:.2f} | ' \
    + " / ".join(us(dd["kernels_ms_per_iter_native_single_frame"].get(k, 0.0)) for k in ("preprocess_fwd", "render_fwd", "render_bwd", "preprocess_bwd")) + ' | 0 | |'
design = f'''Round 5, one MI355X through `gpurun` (boxes differ by ±10 %; `gpurun_out/r05/bench_default.json`, `bench_driver.json`; the judge's
numbers are the driver's `BENCH_r05.json`; round 4's block: git history), {dd["config"]["frames_in_flight_per_gpu"]} frames in flight, one HIP hardware queue each:

| | `value` | stream of frames | single frame (cold start) | plain loop | Python loop | steady state | per-call overhead |
|---|---|---|---|---|---|---|---|
| K = 50 (default) | **{dd["value"]:.0f}** (repeats {dd["value_repeats"][1]:.0f}, {dd["value_repeats"][2]:.0f}) | {dd["stream_of_frames_iters_per_s"]:.0f} | {dd["single_frame_iters_per_s"]:.0f} ({dd["single_frame_cold_start_iters_per_s"]:.0f}) | {dd["plain_loop_iters_per_s"]:.0f} | {dd["python_loop_iters_per_s"]:.0f} | {1e3*dd["steady_state_ms_per_iter"]:.0f} µs | {dd["per_call_overhead_ms"]:.2f} ms |
| K = 20 (the driver's command) | **{dr["value"]:.0f}** ({dr["value_repeats"][1]:.0f}, {dr["value_repeats"][2]:.0f}) | {dr["stream_of_frames_iters_per_s"]:.0f} | {dr["single_frame_iters_per_s"]:.0f} ({dr["single_frame_cold_start_iters_per_s"]:.0f}) | {dr["plain_loop_iters_per_s"]:.0f} | {dr["python_loop_iters_per_s"]:.0f} | {1e3*dr["steady_state_ms_per_iter"]:.0f} µs | {dr["per_call_overhead_ms"]:.2f} ms |

(`value`: one K-iteration call per frame in flight, so its timed region ends with the slowest of the sixteen streams alone on the GPU --
they do not get equal shares; "stream of frames": 64 frames through the same sixteen workers, next frame to whoever is free.)

(round 4's driver run, K = 20: 9 278 / 6 444 / 3 940 / 496.)

**Structured variants of the headline scene** (`scene_variants` leg; one frame, K = 50, warm start from another frame; "need" = list
entries a forward must composite, summed over tiles, after the own exact tile culling -- under the reference's bounding rule the
oracle counts 315 k / 472 k / 986 k / 912 k; round 4, `tools/dbg/robust_probe.py`: object 3 010, walls 2 320 it/s; round 5
before split tiles: room 648):

| scene | need | speculative it/s | complete lists it/s | G entries composited / s | preprocess / K6 / K7 / chain rule (µs) | failed forwards per call | pose error after refinement |
|---|---|---|---|---|---|---|---|
{base_row}
{chr(10).join(svrow(r) for r in sv)}

VERDICT r4 asked for object ≥ 5 000 and walls ≥ 4 000: NOT met.  The walls are not a failing speculation -- they bin 1.07 M instances for
a need of 0.99 M (reference rule), three to four times the uniform cloud's work, and the loop composites MORE entries per second there
than on the cloud; the
object and the room are heavy tiles (split across workgroups: §3.2; what a split still costs: §7).

Median pose error after 50 iterations from 2 cm / 1°: {dd["pose_err_cm_median"]:.2f} cm /
{dd["pose_err_deg_median"]:.2f}° (Adam moves every component by ≈lr per step, as in the reference).  Drop-in host time per call on a scene with negligible GPU
work: `render()` {h["forward"]:.0f} µs forward / {h["backward_incl_two_torch_sums"]:.0f} µs backward, of which the rasterizer module alone {h["rasterizer_module_forward"]:.0f} / {h["rasterizer_module_backward"]:.0f} µs
(the K = 20 run of the same session: {dr["dropin_host_us_per_call"]["forward"]:.0f} / {dr["dropin_host_us_per_call"]["backward_incl_two_torch_sums"]:.0f} and {dr["dropin_host_us_per_call"]["rasterizer_module_forward"]:.0f} / {dr["dropin_host_us_per_call"]["rasterizer_module_backward"]:.0f} -- host times move by ±20 % between runs on one box; round 4: 341 / 241 and 152 / 166).  CPU oracle on {dd["cpu_baseline"]["cores"]} host threads: {dd["cpu_baseline"]["value"]:.2f} it/s.

| config | speculative it/s | complete lists it/s (round 3) | kernels per iteration, complete lists (µs) |
|---|---|---|---|
| S-3M-cam 852×480 (config 3) | {cam[0]["speculative_iters_per_s"]:.0f} | {cam[0]["plain_iters_per_s"]:.0f} (1 703) | bin {us(kp[0]["preprocess_fwd"])}, K6 {us(kp[0]["render_fwd"])} (164), K7 {us(kp[0]["render_bwd"])}, K8 {us(kp[0]["preprocess_bwd"])} (64) |
| S-3M-cam 1024×576 (the script's size) | {cam[1]["speculative_iters_per_s"]:.0f} | {cam[1]["plain_iters_per_s"]:.0f} (1 510) | preprocess {us(kp[1]["preprocess_fwd"])}, count {us(kp[1]["tile_count"])}, scan {us(kp[1]["tile_scan"])}, emit {us(kp[1]["tile_emit"])}, K6 {us(kp[1]["render_fwd"])} (181), K7 {us(kp[1]["render_bwd"])}, K8 {us(kp[1]["preprocess_bwd"])} (58) |

`train.py` step (config 4, 1296×840, SH1): {tr["per_P"][0]["ms_per_step"]:.2f} / {tr["per_P"][1]["ms_per_step"]:.2f} / {tr["per_P"][2]["ms_per_step"]:.2f} ms at 0.2 / 0.8 / 1.5 M Gaussians; at 1.5 M the rasterizer forward is
{tr["per_P"][2]["rasterizer_fwd_ms"]:.2f} ms (preprocess {us(k15["preprocess_fwd"])} + count {us(k15["tile_count"])} + scan {us(k15["tile_scan"])} + emit {us(k15["tile_emit"])} + K6 {us(k15["render_fwd"])} µs (round 3: 240) and the blocking count read), the backward
K7 {us(k15["render_bwd"])} (round 3: 283; launched heaviest tile first now) + K8 {us(k15["preprocess_bwd"])} µs (43), loss epilogue {tr["per_P"][2]["loss_epilogue_ms"]:.2f} ms, torch's Adam + statistics {tr["per_P"][2]["stats_and_adam_ms"]:.2f} ms.
BASELINE config 4 as written (`tools/train_7k.py`, 7 000 steps, 64 changes of P from 0.2 to 1.5 M, opacity reset at 3 000): 17.5 s, 2.50 ms per step, 1 441 MiB peak (final kernels; 17.9 s / 2.55 ms before the record touches of K7).
VERDICT r4's targets for the training forward (≤ 0.44 ms, no blocking read above 2 048 tiles) and `python_loop_iters_per_s` ≥ 650 are NOT met; HISTORY.md (round 5) has the host-time breakdown that says why the latter cannot be met from this side of the boundary.
'''
readme = f'''Round 5 on one MI355X (`bench.py` defaults: 50 iterations per refinement call as in the reference, {dd["config"]["frames_in_flight_per_gpu"]} frames in flight; gpurun boxes --
the driver's own run is `BENCH_r05.json`; round 4's driver run, 20 iterations per call, measured 9 278 / 6 444 / 3 940 / 496):
≈{dd["value"]:.0f} it/s whole-GPU ({dr["value"]:.0f} with 20 iterations per call; {dd["stream_of_frames_iters_per_s"]:.0f} / {dr["stream_of_frames_iters_per_s"]:.0f} on a stream of frames), {dd["single_frame_iters_per_s"]:.0f} it/s for a single frame ({dr["single_frame_iters_per_s"]:.0f}), {dd["plain_loop_iters_per_s"]:.0f} it/s without depth
speculation (complete lists every iteration), {min(dr["python_loop_iters_per_s"], dd["python_loop_iters_per_s"]):.0f}–{max(dr["python_loop_iters_per_s"], dd["python_loop_iters_per_s"]):.0f} it/s for the reference-style Python loop on the drop-in packages (host-bound),
{dd["cpu_baseline"]["value"]:.2f} it/s for the CPU oracle on {dd["cpu_baseline"]["cores"]} host threads.  S-3M-cam (3 M Gaussians): {cam[0]["speculative_iters_per_s"]:.0f} it/s at 852×480, {cam[1]["speculative_iters_per_s"]:.0f} at 1024×576.
A `train.py` step at 1.5 M Gaussians / 1296×840: {tr["per_P"][2]["ms_per_step"]:.1f} ms.
Structured variants of the headline scene (one frame): {" / ".join(f'{r["variant"]} {r["speculative_iters_per_s"]:.0f}' for r in sv)} it/s -- they need {" / ".join(f'{r["R_eff_own_binning"] / dd["config"]["R_eff_own_binning"]:.1f}' for r in sv)} × the uniform cloud's composited entries.
`GSR_DETERMINISTIC=1` (`GSR_REFINE_DETERMINISTIC`): bit-reproducible gradients and poses; under it the speculative, the plain and the
no-lean loop produce identical bits (`tests/test_gpu_deterministic.py`, `tools/fuzz_speculation.py`).
'''
for name, block in (("DESIGN.md", design), ("README.md", readme)):
    p = os.path.join(R, name)
    s = open(p).read()
    s = re.sub(r"<!-- numbers:begin -->\n.*?<!-- numbers:end -->\n", lambda m: "<!-- numbers:begin -->\n" + block + "<!-- numbers:end -->\n", s, flags=re.S)
    open(p, "w").write(s)
    print(name, "updated")
