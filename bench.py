#!/usr/bin/env python3
"""bench.py -- render+backward pose-refinement iterations/s on the headline scene S-1M-640
(640x480, 1 M Gaussians, SH3), one process per GPU.

One refinement iteration = one body of the reference's loop
(gs_localization/pipelines/7scenes_localize_full_dslam.py:66-91): render() through the pose rasterizer
-> tracking loss -> backward (dL/dtau every iteration; the gradient tensors of the Gaussians' own parameters, which
nobody can read before the refinement call returns, are written once per call from the last stepped iteration's records --
what one loss.backward() of that iteration adds; `config.loop`) -> Adam step -> update_pose ->
convergence flag.  Query frames are independent, so every rank (GPU) refines its own frames against its
own replica of the map (weak scaling, no data-path collective; one gather of the results at the end),
and keeps F frames in flight (one host thread + one HIP stream each): the compositing kernels are chains of
dependent per-wave work, so a second frame's kernels fill the issue slots the first one leaves empty.

A "step" is one iteration of every frame in flight on a rank; `value` = world * F * K / time.
Reported next to it, first-class: `single_frame_iters_per_s` (one frame alone on the native loop: the number comparable with
the reference's one-frame-at-a-time loop), `plain_loop_iters_per_s` (native loop without depth speculation: complete lists
every iteration) and `python_loop_iters_per_s` (the reference-style Python loop on the drop-in packages: what the unchanged
scripts would run).

Prints ONE JSON line on rank 0 with extra objects:
  roofline     -- dominant kernel's algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak; whole-iteration
                  traffic and issue-slot utilisation from the committed rocprofv3 counter passes (profiles/)
  cpu_baseline -- the CPU oracle (a port, oracle/gs_oracle.c, OpenMP) on the same scene, rank 0, N=1 (+ other scenes / 1 thread)
  train_step   -- BASELINE.json config 4: ms per train.py-style step at P = 0.2 / 0.8 / 1.5 M (rank 0, N=1)
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md, "HBM3E peak BW 8.0 TB/s spec"


def algorithmic_bytes(P, V, R, R_eff, N, M, ntiles):
    """SURVEY.md section 8(d) 'Algorithmic bytes per fwd+bwd iteration' -- the REFERENCE algorithm's bytes, attributed to the
    kernels here that do that work (the three binning kernels stand for the reference's emit + 64-bit key sort + ranges)."""
    passes = math.ceil((32 + math.ceil(math.log2(ntiles))) / 8)
    per = {
        "preprocess_fwd": P * 44 + V * 48,
        "sh_color": P * 12 * M,
        "tile_count": 0,
        "tile_scan": R * 8,
        "tile_emit": R * 12 + passes * R * 24,
        "render_fwd": R_eff * 44 + N * 24,
        "bwd_zero": 0,
        "render_bwd": N * 24 + R_eff * 44 + R_eff * 36,
        "preprocess_bwd": V * (48 + 36) + P * (44 + 12 * M) + P * (40 + 12 * M),
        "pose_step": 0,
    }
    return per, passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50,
                    help="iterations per refinement call (the reference refines a frame for at most 50: 7scenes_localize_full_dslam.py:66)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--frames-in-flight", type=int, default=16,
                    help="query frames refined concurrently per GPU, one HIP hardware queue each (measured on one MI355X, round 4, K = 20 / 50: "
                         "8 frames 7 930-9 150 / 9 750-9 940 it/s, 12: 7 700-8 840 / 9 220-10 140, 16: 9 700-9 820 / 10 240-10 310, 20: 9 780-9 890 / 10 410-10 540)")
    ap.add_argument("--repeats", type=int, default=5, help="the timed region is run this many times; `value` is the MEDIAN (a region is ~35 ms at K = 20: one of them is +-3 % noise), all of them are reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="also time the CPU port on one thread on S-800k-chess (minutes)")
    ap.add_argument("--no-train-leg", action="store_true")
    ap.add_argument("--no-cam-leg", action="store_true", help="skip BASELINE.json config 3 (S-3M-cam, 852x480 and 1024x576)")
    ap.add_argument("--no-variants-leg", action="store_true", help="skip the structured variants of the headline scene (object / walls / S-room-640)")
    ap.add_argument("--only-variants", default=None, help="diagnostics: run ONLY the scene_variants leg (comma-separated names, or 'all') and print it")
    ap.add_argument("--trained-map", type=int, default=0, metavar="STEPS",
                    help="also run the reference's pipeline chained once (tests/trained_map.py: train STEPS steps -> point_cloud.ply -> from_ply -> masks + "
                         "refinement with the early exit) and report it as `trained_map`; 7000 = BASELINE config 4's cadence (adds ~40 s); 0 = off")
    ap.add_argument("--pose-only", action="store_true", help="skip the Gaussian-parameter gradients (not the headline)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for N > 1; gloo (collectives on host tensors) lets the N > 1 code path be rehearsed on a box with one GPU")
    ap.add_argument("--device-index", type=int, default=None, help="GPU of this rank (default: LOCAL_RANK); rehearsals put every rank on GPU 0")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, torch.distributed.run over
        # RCCL), BEFORE anything touches the GPU in this process, hand their output through and exit with their status.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    # F frames in flight = F HIP streams.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues
    # (default 4): with twelve streams on four queues, kernels of three frames queue up behind each other in order.  One queue per
    # frame in flight (at most 16), set before the first HIP call; an explicit setting in the environment wins.
    # (measured on one MI355X, K = 20 / 50: 4 queues 8 710-8 810 / 9 930-9 970 it/s, 8 queues 9 290-9 550 / 10 160-10 380, 16 queues
    # 9 170-9 850 / 10 300-10 480; single frame unchanged)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(16, args.frames_in_flight))))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    dev_index = local_rank if args.device_index is None else args.device_index
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or without a launcher)")
    # A process group exists whenever a launcher started this process -- also for ONE rank (`torch.distributed.run
    # --nproc-per-node 1`): barrier, all_reduce and the result gather then run through RCCL on device tensors exactly as on
    # eight GPUs.  Without a launcher (the driver's N = 1 run) there is no group and no collective.
    grouped = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
    from gs_localization_amd import shard as _shard
    if grouped:
        _shard.init_process_group(args.backend, rank, world, device=torch.device("cuda", dev_index))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")      # where the tensors of the (tiny) collectives live

    from gs_localization_amd import _lib, scenes as S, shard
    from tests import replay as PL      # the reference-style Python loop on the drop-in packages (test infrastructure)
    lib = _lib.load()
    assert lib.gsr_device_ok() == 1, "no gfx950 device"
    if args.only_variants:
        only = None if args.only_variants == "all" else set(args.only_variants.split(","))
        print(json.dumps({"scene_variants": scene_variants_leg(lib, dev, K=args.steps, only=only)}), flush=True)
        return

    K, Wm, F = args.steps, max(args.warmup, 1), max(args.frames_in_flight, 1)
    sc = S.s_1m_640(P=args.gaussians)
    W, H, M = sc.W, sc.H, sc.shs.shape[1]
    N, ntiles = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    model = PL.GaussianMap.from_scene(sc, device=dev, requires_grad=not args.pose_only)
    background = torch.zeros(3, dtype=torch.float32, device=dev)
    config = PL.TRACKING_CONFIG
    w2c_gt = np.eye(4)

    # query frames of this rank (global ids rank*F .. rank*F+F-1): GT pose = identity, start pose off by
    # (2 cm, 1 deg) in a per-frame random direction (SURVEY 8(c) fixture 9)
    frame_ids = [rank * F + f for f in range(F)]
    inits = [PL.perturbed_start(1000 + fid, device=dev) for fid in frame_ids]
    vps = [PL.make_frame(sc, model, dev, background, uid=fid) for fid in frame_ids]
    vp, init = vps[0], inits[0]
    # Round 6: every frame is refined under the mask the reference's localisers build per frame -- compute_grad_mask (Scharr gradient
    # above 1.1 x its median, camera_utils.py:164-193) OR-ed with 10 x 10-ish boxes around ~500 keypoints
    # (7scenes_localize_full_dslam.py:355-360) -- computed by gsr_grad_mask INSIDE every timed call, as the scripts compute it in
    # front of every gradient_decent().  `mask_mode` "ones" (all pixels: what rounds 1-5 measured) is the secondary leg.
    keypoints = [PL.frame_keypoints(W, H, fid) for fid in frame_ids]
    ones_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
    mask_share = float(np.mean([float(v.grad_mask.float().mean()) for v in vps]))
    mask_share_no_boxes = float(PL.reference_mask(vp.original_image, keypoints=False).float().mean())
    mask_mode = {"m": "reference"}

    def frame_mask(g):
        from gs_localization_amd import pipelines as _P
        if mask_mode["m"] == "ones":
            return ones_mask
        return _P.grad_mask(vps[g].original_image, config["Training"]["edge_threshold"], keypoints[g], 10)
    w2c_init = init.cpu().numpy().astype(np.float64)

    def reset(v=vp, i0=init):
        v.update_RT(i0[:3, :3].clone(), i0[:3, 3].clone())
        for p_ in (v.cam_rot_delta, v.cam_trans_delta, v.exposure_a, v.exposure_b):
            p_.data.zero_()
        return PL.pose_adam(v)

    def barrier():
        if grouped:
            dist.barrier()

    nk = lib.gsr_profile_kernel_count()
    names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]

    def collect():
        ms = (C.c_double * nk)()
        cnt = (C.c_longlong * nk)()
        _lib.check(lib.gsr_profile_collect(ms, cnt))
        return {names[i]: (ms[i], cnt[i]) for i in range(nk)}

    # ---- (0) warm-up of the Python loop with every kernel bracketed by HIP events -> per-kernel breakdown
    opt = reset()
    lib.gsr_profile_enable((1 << nk) - 1)
    for _ in range(Wm):
        conv, _pkg = PL.loop_iteration(vp, config, model, background, opt)
        bool(conv)
    torch.cuda.synchronize()
    kernels_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in collect().items()}
    lib.gsr_profile_enable(0)
    del _pkg

    # scene statistics at the start pose (V, R under the reference rule, list entries ordered, R_eff of own binning)
    stats = (C.c_longlong * 4)()
    reset()
    pkg = PL.render(vp, model, background)          # fresh graph: saved tensors still alive
    sv = pkg["render"].grad_fn.saved_tensors
    _lib.check(lib.gsr_forward_stats(sc.P, W, H, sv[5].data_ptr(), sv[7].data_ptr(), sv[9].data_ptr(), stats,
                                     torch.cuda.current_stream().cuda_stream))
    V, R, R_ordered, R_eff_culled = (int(stats[i]) for i in range(4))
    del pkg, sv
    # the byte model is defined on the REFERENCE's binning (bounding-square rule), whatever the
    # implementation emits.  V and R under that rule come from the GPU stats; R_eff under that rule needs the
    # reference lists, so it is taken from the CPU oracle's forward at this pose (cpu_baseline leg).
    cpu, cpu_more = None, None
    R_eff, R_eff_src = R_eff_culled, "own culled binning (no CPU oracle run)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, ref_counts = cpu_baseline(sc, w2c_init)
        # (the oracle builds its fp32 camera matrices from the fp64 pose, the GPU path from fp32 R, T: a handful
        # of Gaussians on the cull boundaries may differ)
        if abs(ref_counts["V"] - V) <= 1e-3 * V and abs(ref_counts["R"] - R) <= 1e-3 * R:
            R_eff, R_eff_src = ref_counts["R_eff"], "reference bounding rule (CPU oracle at the same pose)"
        cpu_more = cpu_baseline_other_scenes(args.cpu_baseline_full)
    per_kernel_bytes, passes = algorithmic_bytes(sc.P, V, R, R_eff, N, M, ntiles)
    total_bytes = sum(per_kernel_bytes.values())

    # ---- (a) the reference's own Python loop on the drop-in packages (torch autograd, Adam, update_pose)
    opt = reset()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        conv, _pkg = PL.loop_iteration(vp, config, model, background, opt)
        bool(conv)          # the reference's `if converged: break` forces this host sync every iteration
    torch.cuda.synchronize(); barrier()
    elapsed_py = time.perf_counter() - t0
    del _pkg

    # ---- (a') what one forward / backward through the drop-in pose package costs the HOST: the same calls on a scene so small
    # (2 000 Gaussians, 64x48) that the GPU work is a few launch latencies -- wall time per call, the forward's one blocking read
    # included.  (Of the Python loop's ~1.8 ms per iteration the library's kernels are ~0.3 ms; the rest is this and torch.)
    host_us = None
    if rank == 0:
        tiny = S.small(P=2000, W=64, H=48, sh_degree=3, seed=3, scale_med=0.06)
        tmodel = PL.GaussianMap.from_scene(tiny, device=dev)
        tvp = PL.make_frame(tiny, tmodel, dev, background)
        fw = bw = 0.0
        for it_ in range(60):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tpkg = PL.render(tvp, tmodel, background)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            (tpkg["render"].sum() + tpkg["depth"].sum()).backward()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            if it_ >= 10:
                fw += t1 - t0; bw += t2 - t1
        host_us = {"forward": 1e6 * fw / 50, "backward_incl_two_torch_sums": 1e6 * bw / 50, "scene": "2 000 Gaussians, 64x48: GPU work negligible"}
        # ... and the library's own share of it: the rasterizer module called directly with settings built once (render() spends the
        # rest on the camera matrices, the settings tuple and a zeros_like -- the reference's Python), backward with ready-made
        # gradient images through torch.autograd.backward (no loss kernels)
        from diff_gaussian_rasterization_pose import GaussianRasterizationSettings as _RS, GaussianRasterizer as _RZ
        rs = _RS(image_height=int(tvp.image_height), image_width=int(tvp.image_width), tanfovx=math.tan(0.5 * tvp.FoVx), tanfovy=math.tan(0.5 * tvp.FoVy),
                 bg=background, scale_modifier=1.0, viewmatrix=tvp.world_view_transform, projmatrix=tvp.full_proj_transform,
                 projmatrix_raw=tvp.projection_matrix, sh_degree=tmodel.active_sh_degree, campos=tvp.camera_center, prefiltered=False, debug=False)
        rz = _RZ(raster_settings=rs)
        m2d = torch.zeros_like(tmodel.get_xyz, requires_grad=True)
        gi, gd = torch.ones((3, tiny.H, tiny.W), device=dev), torch.ones((1, tiny.H, tiny.W), device=dev)
        fw = bw = 0.0
        for it_ in range(60):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o_ = rz(means3D=tmodel.get_xyz, means2D=m2d, opacities=tmodel.get_opacity, shs=tmodel.get_features, colors_precomp=None,
                    scales=tmodel.get_scaling, rotations=tmodel.get_rotation, cov3D_precomp=None, theta=tvp.cam_rot_delta, rho=tvp.cam_trans_delta)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            torch.autograd.backward([o_[0], o_[2]], [gi, gd])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            if it_ >= 10:
                fw += t1 - t0; bw += t2 - t1
        host_us["rasterizer_module_forward"] = 1e6 * fw / 50
        host_us["rasterizer_module_backward"] = 1e6 * bw / 50
        del tiny, tmodel, tvp, tpkg, o_

    # ---- (b) the native loop (gsr_refine), one frame at a time; per-kernel breakdown from a separate short run
    frs = [PL.FusedRefiner(model, H, W, device=dev, gaussian_grads=not args.pose_only) for _ in range(F)]

    def native(f, iters, stop=False, speculative=True, warm=None, frame=None, flags=None):
        # warm=None: the refiner's default -- a frame starts from the depth bounds its PREDECESSOR on this refiner left behind
        # (consecutive frames of a sequence), verified on the device like every speculation.  The predecessor is always another
        # query frame here (another start pose): refiner slot f takes frame `frame` (default f).
        g = f if frame is None else frame % F
        vps[g].grad_mask = frame_mask(g)          # (per call, inside every timed region: the reference computes it per frame too)
        return frs[f].refine(vps[g], config, inits[g][:3, :3].clone(), inits[g][:3, 3].clone(), background, iters=iters,
                             stop_on_converged=stop, speculative=speculative, warm_start=warm, flags=flags)

    def timed_single(spec, iters=K, warm=None, flags=None):
        native(0, Wm, speculative=spec, frame=1, flags=flags)           # the predecessor: frame 1 (its bounds are what a warm start gets)
        barrier(); torch.cuda.synchronize()
        t = time.perf_counter()
        native(0, iters, speculative=spec, warm=warm, frame=0, flags=flags)
        torch.cuda.synchronize(); barrier()
        return time.perf_counter() - t
    # (every leg: the best of three)
    elapsed_single = min(timed_single(True) for _ in range(3))
    elapsed_plain = min(timed_single(False) for _ in range(3))
    elapsed_cold = min(timed_single(True, warm=False) for _ in range(3))          # first iteration bins completely (no bounds from a previous frame)
    # (diagnostics, next to the headline: the same single-frame call with the Gaussian-parameter gradient rows written by EVERY iteration
    # instead of once per call -- GSR_REFINE_GRADS_EVERY_ITERATION; same results, tests/test_gpu_deterministic.py)
    from gs_localization_amd import _lib as _L
    elapsed_rows_every = min(timed_single(True, flags=_L.REFINE_GRADS_EVERY_ITERATION) for _ in range(3))
    # per-call fixed cost: one K-iteration call against the marginal cost of an iteration inside a long call
    elapsed_long = min(timed_single(True, iters=4 * K) for _ in range(2))
    steady_ms = 1e3 * (elapsed_long - elapsed_single) / (3 * K)
    per_call_overhead_ms = 1e3 * elapsed_single - K * steady_ms
    # the mask on its own (device time + its launches, one frame): part of per_call_overhead_ms, since every timed call computes it
    torch.cuda.synchronize()
    t_m = time.perf_counter()
    for _ in range(50):
        frame_mask(0)
    torch.cuda.synchronize()
    grad_mask_ms = 1e3 * (time.perf_counter() - t_m) / 50
    # secondary leg: the same single-frame call with every pixel in the mask (rounds 1-5's workload)
    mask_mode["m"] = "ones"
    elapsed_single_ones = min(timed_single(True) for _ in range(3))
    mask_mode["m"] = "reference"
    PROF_ITERS = 40
    native_ms = {}
    for spec in (True, False):
        lib.gsr_profile_enable((1 << nk) - 1)
        native(0, PROF_ITERS, speculative=spec)
        torch.cuda.synchronize()
        prof = collect()
        lib.gsr_profile_enable(0)
        # ms per ITERATION (all launches of that kernel; the first iteration of a speculative frame bins completely, the others by tile)
        native_ms[spec] = {k: v[0] / PROF_ITERS for k, v in prof.items()}
    dominant = max(native_ms[True], key=native_ms[True].get)

    # ---- (c) TIMED REGION of `value`: F frames in flight per rank, K iterations each, native loop.
    # Only the dominant kernel is bracketed by HIP events (on the stream it is launched on).
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    results = [None] * F
    spans = [None] * F          # (diagnostics: when each worker's call started and ended)

    # F persistent host threads (one per frame slot, each with its own stream), released together: starting a thread costs ~0.1 ms,
    # which inside a timed region of K = 20 iterations (25 ms) would be a few per cent of it
    job = {"iters": 0, "stop": False, "shift": 0, "quit": False}
    gate_in, gate_out = threading.Barrier(F + 1), threading.Barrier(F + 1)

    def worker(f):
        while True:
            gate_in.wait()
            if job["quit"]:
                return
            shift = job["shift"]
            try:
                with torch.cuda.stream(streams[f]):
                    t_in = time.perf_counter()
                    if job.get("queue") is not None:
                        # a stream of frames: every worker takes the next frame when it is done with its own (what
                        # tools/localize_split.py does with shard.FrameQueue) -- no worker waits for the slowest one's single frame
                        while True:
                            with job["lock"]:
                                nxt = next(job["queue"], None)
                            if nxt is None:
                                break
                            results[f] = native(f, job["iters"], job["stop"], frame=nxt, flags=job.get("flags"))
                    else:
                        results[(f + shift) % F] = native(f, job["iters"], job["stop"], frame=f + shift, flags=job.get("flags"))
                    spans[f] = (t_in, time.perf_counter())
            except Exception as ex:      # re-raised in the main thread
                results[(f + shift) % F] = ex
            gate_out.wait()
    pool = [threading.Thread(target=worker, args=(f,), daemon=True) for f in range(F)]
    [t.start() for t in pool]

    def run_all(iters, stop=False, shift=0):
        # (shift: refiner slot f takes frame f + shift -- every repeat hands each refiner another frame than the one whose depth
        # bounds it still holds, as consecutive frames of a sequence would)
        job.update(iters=iters, stop=stop, shift=shift)
        gate_in.wait()
        gate_out.wait()
        for e in results:
            if isinstance(e, Exception):
                raise e
    run_all(Wm)
    torch.cuda.synchronize()
    # one launch in 16 is bracketed: an event pair around every launch keeps the kernel from overlapping its neighbours
    # of the other frames in flight (measured: -5 % on `value`)
    lib.gsr_profile_sampling(16)
    lib.gsr_profile_enable(1 << names.index(dominant))
    elapsed_runs, run_stats = [], []
    dom_ms, dom_n = 0.0, 0
    for rep in range(max(1, args.repeats)):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_all(K, shift=rep + 1)
        torch.cuda.synchronize(); barrier()
        elapsed_runs.append(time.perf_counter() - t0)
        run_stats.append({k: int(sum(r[2][k] for r in results)) for k in ("fallbacks", "host_redos", "lean_iters")})
        if os.environ.get("GSR_BENCH_SPANS"):      # (diagnostics: the frames in flight do not get equal shares of the GPU -- see HISTORY.md, round 4)
            print("rep", rep, "total %.2f ms; call starts (ms after t0) %s; ends %s" % (1e3 * elapsed_runs[-1],
                  " ".join("%.2f" % (1e3 * (a - t0)) for a, _ in spans), " ".join("%.2f" % (1e3 * (b - t0)) for _, b in spans)), file=sys.stderr)
        if rep == 0:
            dom_ms, dom_n = collect()[dominant]
            lib.gsr_profile_enable(0)
            lib.gsr_profile_sampling(1)
    elapsed_first = elapsed_runs[0]
    elapsed = sorted(elapsed_runs)[len(elapsed_runs) // 2]          # the median repeat is `value` (a 34 ms headline is +-3 % noise)
    if grouped:
        t = torch.tensor([elapsed, elapsed_py, elapsed_single, elapsed_plain, elapsed_cold] + elapsed_runs, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in t.tolist()]
        elapsed_py, elapsed_single, elapsed_plain, elapsed_cold, elapsed_runs = vals[1], vals[2], vals[3], vals[4], vals[5:]
        elapsed_first = elapsed_runs[0]
        elapsed = sorted(elapsed_runs)[len(elapsed_runs) // 2]          # (max over ranks per repeat, then the median repeat)
    # how many ranks really took part (a launcher that started fewer than --gpus would otherwise go unnoticed)
    ranks_seen = 1
    if grouped:
        ones = torch.ones(1, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(ones.item())))
    if ranks_seen != args.gpus:
        sys.exit(f"bench.py: {ranks_seen} ranks took part, --gpus {args.gpus} expected")

    # ---- the same K iterations per frame on a STREAM of frames (4 F of them through the F workers, next frame to whoever is free):
    # reported next to `value`, whose timed region is one frame per worker and therefore ends with its slowest worker alone on the GPU
    # (the frames in flight do not get equal shares of it: a third between the first and the last to finish, GSR_BENCH_SPANS=1)
    job.update(queue=iter(range(1, 4 * F + 1)), lock=threading.Lock())
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_all(K)
    torch.cuda.synchronize(); barrier()
    elapsed_stream = time.perf_counter() - t0
    job.update(queue=None)

    # ---- secondary legs at F frames in flight (median of three regions each): every pixel in the mask (rounds 1-5's workload), and the
    # Gaussian-parameter gradient rows written by EVERY iteration (diagnostic flag) under the reference's mask
    def timed_all(nrep=3):
        ts = []
        run_all(Wm)
        for rep in range(nrep):
            barrier(); torch.cuda.synchronize()
            t0_ = time.perf_counter()
            run_all(K, shift=rep + 1)
            torch.cuda.synchronize(); barrier()
            ts.append(time.perf_counter() - t0_)
        t_ = sorted(ts)[len(ts) // 2]
        if grouped:
            tt = torch.tensor([t_], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_ = float(tt.item())
        return t_
    mask_mode["m"] = "ones"
    elapsed_ones = timed_all()
    mask_mode["m"] = "reference"
    job.update(flags=_L.REFINE_GRADS_EVERY_ITERATION)
    elapsed_rows_every_all = timed_all()
    job.update(flags=None)

    # ---- pose error of full 50-iteration refinements with the reference's early exit (untimed), all frames gathered
    run_all(50, stop=True)
    torch.cuda.synchronize()
    rows = []
    for f in range(F):
        Rr, Tt, info = results[f]
        te, re = PL.pose_errors(w2c_gt[:3, :3], w2c_gt[:3, 3], Rr.detach().cpu().numpy(), Tt.detach().cpu().numpy())
        rows.append([float(frame_ids[f]), te, re, float(info["iters"])])
    te0, re0 = PL.pose_errors(w2c_gt[:3, :3], w2c_gt[:3, 3], w2c_init[:3, :3], w2c_init[:3, 3])
    res = shard.gather_results(torch.tensor(rows, dtype=torch.float64, device=coll_dev), world * F, rank, world)

    job["quit"] = True
    gate_in.wait()
    [t.join() for t in pool]
    train, cam, variants = None, None, None
    if rank == 0 and world == 1 and not (args.no_train_leg and args.no_cam_leg and args.no_variants_leg):
        del frs, vps, model
        torch.cuda.empty_cache()
        if not args.no_variants_leg:
            variants = scene_variants_leg(lib, dev, K=K)
            torch.cuda.empty_cache()
        if not args.no_cam_leg:
            cam = cam_step_leg(lib, dev)
            torch.cuda.empty_cache()
        if not args.no_train_leg:
            train = train_step_leg(lib)
    trained = None
    if rank == 0 and world == 1 and args.trained_map > 0:
        import tempfile
        from tests import trained_map as TM
        ply = os.path.join(tempfile.mkdtemp(prefix="gsr_map_"), "point_cloud", "iteration_%d" % args.trained_map, "point_cloud.ply")
        tworld, treport = TM.train_room_map(ply, steps=args.trained_map)
        lrep, _ = TM.localise_against(ply, tworld, n_frames=32, in_flight=F, start=(0.02, 1.0))
        trained = {"workload": "train.py's loop on a synthetic room world -> point_cloud.ply -> GaussianMap.from_ply -> gsr_grad_mask + FusedRefiner.refine, early exit; "
                               "query frames rendered from the WORLD; starts 2 cm / 1 deg off (tests/trained_map.py)", "train": treport, "localise": lrep}

    if rank == 0:
        res = res.cpu().numpy()
        dom_avg_ms = dom_ms / max(dom_n, 1)
        achieved = per_kernel_bytes[dominant] / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        traffic, iter_traffic, issue = None, None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            traffic = tj.get(dominant)
            # (a steady-state iteration of the native loop: k_preprocess_lean with the SH colour inside -- maps below 200 k Gaussians:
            # k_preprocess + k_sh_color --, the two compositing kernels, the chain-rule kernel)
            loop_kernels = (("preprocess_lean",) if "preprocess_lean" in tj else ("preprocess_fwd", "sh_color")) + ("render_fwd", "render_bwd", "preprocess_bwd")
            if all(k in tj for k in loop_kernels):
                iter_traffic = sum(tj[k] for k in loop_kernels)
        except Exception:
            pass
        loop_valu_frac = None
        try:
            ij = json.load(open(os.path.join(ROOT, "profiles", "issue.json")))
            d_ = ij.get(dominant)
            if d_:      # two different pipes, reported side by side (never summed: MFMA and VALU instructions of different waves overlap)
                issue = {"valu_issue_frac": d_["valu_issue_frac"], "mfma_pipe_frac": d_["mfma_pipe_frac"],
                         "binding_pipe": "valu" if d_["valu_issue_frac"] >= d_["mfma_pipe_frac"] else "mfma"}
            # whole steady-state iteration: wave-level vector instructions x 4 cycles each over the chip's 1 024 SIMDs at 2.4 GHz,
            # against the measured time per iteration (single frame)
            lk = (("preprocess_lean",) if "preprocess_lean" in ij else ("preprocess_fwd", "sh_color")) + ("render_fwd", "render_bwd", "preprocess_bwd")
            if all(k in ij and "valu_insts_per_launch" in ij[k] for k in lk):
                insts = sum(ij[k]["valu_insts_per_launch"] for k in lk)
                loop_valu_frac = {"valu_wave_insts_per_iter": insts,
                                  "single_frame": insts * 4.0 / (1024 * 2.4e9) / (steady_ms * 1e-3),
                                  "at_value": insts * 4.0 / (1024 * 2.4e9) / (elapsed / (F * K))}
        except Exception:
            pass
        iters_total = world * F * K
        single = world * K / elapsed_single
        out = {
            "metric": "render+backward iters/sec @640x480, 1M Gaussians; median pose err (cm/deg)",
            "value": iters_total / elapsed,
            "unit": "iters/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "collectives": (f"{args.backend} ({'RCCL' if args.backend == 'nccl' else 'host tensors'}): barrier, all_reduce(max), all_gather of the result rows"
                            if grouped else "none (no launcher: one rank without a process group)"),
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "S-1M-640 pose refinement (BASELINE.json configs[1] shape at 1M Gaussians)",
                       "width": W, "height": H, "gaussians": sc.P, "sh_degree": sc.sh_degree,
                       "V": V, "R": R, "R_eff": R_eff, "R_eff_source": R_eff_src,
                       "list_entries_ordered_own_binning": R_ordered, "R_eff_own_binning": R_eff_culled, "sort_passes_reference": passes,
                       "algorithmic_bytes_per_iter": total_bytes, "frames_in_flight_per_gpu": F,
                       "iterations_per_step": F, "gaussian_grads": not args.pose_only,
                       "parallelism": f"frames: {world} GPU x {F} in flight",
                       "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "loop": "native gsr_refine (render, tracking loss, backward, Adam, update_pose per iteration; dL/dtau every iteration, the gradient tensors "
                               "of the Gaussians' own parameters -- which nobody can read before the call returns -- written once per call, from the last "
                               "stepped iteration's records: what ONE loss.backward() of that iteration adds -- the reference never zeroes the map tensors' .grad inside its loop, so there it is the sum over the iterations; nobody reads either)",
                       "iterations_per_call": K,
                       "grad_mask": "the reference's per-frame mask, computed by gsr_grad_mask inside every timed call: compute_grad_mask (camera_utils.py:164-193, "
                                    "edge_threshold 1.1) | create_mask over %d seeded keypoints, k = 10 (7scenes_localize_full_dslam.py:355-360)" % PL.N_KEYPOINTS,
                       "grad_mask_pixel_share": mask_share, "grad_mask_pixel_share_without_keypoint_boxes": mask_share_no_boxes,
                       "grad_mask_ms_per_frame": grad_mask_ms,
                       "warm_policy": "every refinement call starts from the depth bounds ANOTHER query frame (another start pose) left in its "
                                      "refiner's workspace, verified on the device; single_frame_cold_start_iters_per_s has no bounds to start from",
                       "timing": "value = MEDIAN of `repeats` timed regions (value_first_repeat: the first); single-frame / plain / cold legs = best of three calls"},
            "value_first_repeat": iters_total / elapsed_first,
            "value_repeats": [iters_total / e for e in elapsed_runs],
            "value_repeats_stats": run_stats,
            "stream_of_frames_iters_per_s": 4 * F * K / elapsed_stream,
            "single_frame_iters_per_s": single,
            "single_frame_iters_per_s_gradient_rows_every_iteration": world * K / elapsed_rows_every,
            "single_frame_cold_start_iters_per_s": world * K / elapsed_cold,
            # one refinement call of K iterations on one frame: its time, the marginal cost of an iteration inside a long call,
            # and what the call costs on top of K of those (host set-up, first iteration, the n_touched pass, final read-back)
            "single_frame_call_ms": 1e3 * elapsed_single,
            "steady_state_ms_per_iter": steady_ms,
            "per_call_overhead_ms": per_call_overhead_ms,
            "plain_loop_iters_per_s": world * K / elapsed_plain,
            "python_loop_iters_per_s": world * K / elapsed_py,
            "dropin_host_us_per_call": host_us,
            "pose_err_cm_median": 100.0 * float(np.median(res[:, 1])),
            "pose_err_deg_median": float(np.median(res[:, 2])),
            "pose_err_init_cm_deg": [100.0 * te0, re0],
            "refine_iters_median": float(np.median(res[:, 3])),
            "kernels_ms_python_loop": {k: round(v, 4) for k, v in kernels_ms.items()},
            "kernels_ms_per_iter_native_single_frame": {k: round(v, 4) for k, v in native_ms[True].items()},
            "kernels_ms_per_iter_native_plain_loop": {k: round(v, 4) for k, v in native_ms[False].items()},
            # roofline.achieved / frac: the dominant kernel's algorithmic bytes over its duration with ONE frame on the GPU (HIP events around
            # every launch of the kernel, on its stream, in this process) -- the figure that describes the kernel.  In the timed region of
            # `value` the same launch is stretched by the fifteen other frames' kernels sharing the chip: achieved_at_value / frac_at_value.
            "roofline": {"bound": "hbm", "kernel": dominant,
                         "achieved": (per_kernel_bytes[dominant] / (native_ms[True][dominant] * 1e-3) / 1e9) if native_ms[True][dominant] > 0 else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (per_kernel_bytes[dominant] / (native_ms[True][dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS) if native_ms[True][dominant] > 0 else None,
                         "traffic": traffic,
                         "traffic_source": "profiles/traffic.json (builder-run rocprofv3 FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied; not measured in this run)",
                         "algorithmic_bytes_per_launch": per_kernel_bytes[dominant],
                         "avg_launch_ms": native_ms[True][dominant],
                         "achieved_at_value": achieved, "frac_at_value": achieved / HBM_PEAK_GBS, "avg_launch_ms_at_value": dom_avg_ms,
                         "launches_timed_at_value": int(dom_n),
                         # what the kernel is actually bound by (rocprofv3 SQ pass, profiles/issue.json): its vector-issue and
                         # matrix-pipe utilisations, two pipes side by side; and the whole loop's vector-issue fraction
                         "bound_measured": "valu_issue",
                         "issue_frac": issue,
                         "loop_valu_issue_frac": loop_valu_frac,
                         "issue_source": "profiles/issue.json (builder-run rocprofv3 SQ counter passes; not measured in this run)",
                         # whole iteration: HBM bytes per steady-state iteration (sum of the loop's kernels, profiles/traffic.json)
                         # and that traffic at the measured single-frame / in-flight rates against the HBM peak
                         "iter_traffic_bytes": iter_traffic,
                         "iter_traffic_source": "profiles/traffic.json (builder-run rocprofv3; not measured in this run)",
                         "iter_traffic_frac_single_frame": (iter_traffic * single / 1e9 / HBM_PEAK_GBS) if iter_traffic else None,
                         "iter_traffic_frac_at_value": (iter_traffic * iters_total / elapsed / 1e9 / HBM_PEAK_GBS / world) if iter_traffic else None,
                         # the reference algorithm's bytes per iteration (all kernels) x measured iterations/s against the
                         # HBM peak: > 1 means the loop runs faster than the reference's traffic could even be streamed
                         "reference_bytes_rate_frac": total_bytes * iters_total / elapsed / 1e9 / HBM_PEAK_GBS / world},
        }
        out["value_all_ones_mask"] = iters_total / elapsed_ones
        out["value_gradient_rows_every_iteration"] = iters_total / elapsed_rows_every_all
        out["single_frame_iters_per_s_all_ones_mask"] = world * K / elapsed_single_ones
        if cpu is not None:
            out["cpu_baseline"] = cpu
            out["cpu_baseline_other_scenes"] = cpu_more
        if variants is not None:
            out["scene_variants"] = variants
        # The driver's record keeps `config` and `roofline` whole and only the NAMES of the other keys: the secondary
        # rates are therefore repeated here, rounded.
        sec = {"value_all_ones_mask": out["value_all_ones_mask"], "value_gradient_rows_every_iteration": out["value_gradient_rows_every_iteration"],
               "value_first_repeat": out["value_first_repeat"], "stream_of_frames_iters_per_s": out["stream_of_frames_iters_per_s"],
               "single_frame_iters_per_s": single, "single_frame_iters_per_s_all_ones_mask": out["single_frame_iters_per_s_all_ones_mask"],
               "single_frame_iters_per_s_gradient_rows_every_iteration": out["single_frame_iters_per_s_gradient_rows_every_iteration"],
               "single_frame_cold_start_iters_per_s": out["single_frame_cold_start_iters_per_s"], "plain_loop_iters_per_s": out["plain_loop_iters_per_s"],
               "python_loop_iters_per_s": out["python_loop_iters_per_s"], "steady_state_ms_per_iter": steady_ms, "per_call_overhead_ms": per_call_overhead_ms,
               "pose_err_cm_deg_median": [out["pose_err_cm_median"], out["pose_err_deg_median"]], "refine_iters_median": out["refine_iters_median"],
               "kernels_us_per_iter_single_frame": {k: round(1e3 * v, 1) for k, v in native_ms[True].items() if v > 0}}
        def _rows(leg):
            return (leg.get("per_scene") or leg.get("per_size") or []) if isinstance(leg, dict) else []
        if variants is not None:
            sec["scene_variants_iters_per_s"] = {r_["variant"]: {"speculative": round(r_["speculative_iters_per_s"], 1), "complete_lists": round(r_["plain_iters_per_s"], 1),
                                                                 "speculative_gradient_rows_every_iteration": round(r_["speculative_iters_per_s_gradient_rows_every_iteration"], 1),
                                                                 "speculative_16_in_flight": round(r_["speculative_16_frames_in_flight"]["iters_per_s"], 1),
                                                                 "kernels_us": {k: round(1e3 * v, 1) for k, v in r_["kernels_ms_per_iter_speculative"].items()}}
                                                 for r_ in _rows(variants)}
        if cam is not None:
            sec["cam_step_iters_per_s"] = {"%s %dx%d" % (r_["scene"], r_["width"], r_["height"]): {"speculative": round(r_["speculative_iters_per_s"], 1),
                                                                                                   "complete_lists": round(r_["plain_iters_per_s"], 1)} for r_ in _rows(cam)}
        if train is not None:
            sec["train_step_per_P"] = train.get("per_P")
        out["config"]["secondary"] = {k: (round(v, 2) if isinstance(v, float) else v) for k, v in sec.items()}
        if cam is not None:
            out["cam_step"] = cam
        if trained is not None:
            out["trained_map"] = trained
            out["config"]["trained_map"] = {"train_ms_per_step": round(trained["train"]["ms_per_step"], 3), "map_gaussians": trained["localise"]["map_gaussians"],
                                           "pose_err_cm_deg_median": [round(trained["localise"]["pose_err_cm_median"], 3), round(trained["localise"]["pose_err_deg_median"], 4)],
                                           "single_frame_iters_per_s": round(trained["localise"]["single_frame_iters_per_s"], 1),
                                           "in_flight_iters_per_s": round(trained["localise"]["in_flight_iters_per_s"], 1)}
        if train is not None:
            out["train_step"] = train
        print(json.dumps(out), flush=True)
    if grouped:
        _shard.destroy_process_group()


def _cpu_time(sc, w2c, threads, backward, budget_s, max_n):
    from oracle import oracle as O
    from gs_localization_amd import scenes as S
    O.set_threads(threads)
    view, proj, _, campos = S.camera_matrices(sc, w2c)
    rng = np.random.default_rng(0)
    gc = rng.normal(size=(3, sc.H, sc.W)).astype(np.float32)
    gd = rng.normal(size=(1, sc.H, sc.W)).astype(np.float32)
    ga = np.zeros((1, sc.H, sc.W), np.float32)
    n, t0 = 0, time.perf_counter()
    while True:
        f = O.forward(sc.means3D, sc.opacities, view, proj, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg,
                      sh_degree=sc.sh_degree, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
        if backward:
            O.backward(f, gc, gd, ga, pose_mode=True)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_n:
            break
    return n, el, f


def cpu_baseline(sc, w2c):
    """The oracle (a port of the reference algorithm, not the reference itself) on the host cores:
    fwd+bwd of the rasterizer with pose gradients on the same scene and pose; bounded to ~10-30 s in all.
    The port does not scale with threads everywhere (its duplicate / radix-sort steps are serial, as restated from
    rasterizer_impl.cu:70-138; page faults of per-call state): the rate is measured at several thread counts and the BEST one is
    `value`, all of them are listed.  Also returns the reference-rule counts (V, R, R_eff) of that forward."""
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    tried, f = [], None
    for thr in sorted({1, 8, 16, 32, 64, cores}):
        if thr > cores:
            continue
        n, el, f = _cpu_time(sc, w2c, thr, True, 3.5, 6)
        tried.append({"threads": thr, "iters_per_s": n / el, "iterations": n})
    best = max(tried, key=lambda r: r["iters_per_s"])
    counts = {"V": int((f.radii > 0).sum()), "R": int(f.num_rendered), "R_eff": O.r_eff(f)}
    return ({"value": best["iters_per_s"], "unit": "iters/s", "cores": best["threads"], "kind": "port", "host_cores": cores,
             "per_thread_count": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()} for r in tried],
             "sample": f"best of {len(tried)} thread counts, {best['iterations']} rasterizer fwd+bwd iterations (pose gradients) of the same S-1M-640 scene and pose at "
                       f"{best['threads']} threads (OpenMP over Gaussians / tiles); excludes loss/Adam/update_pose (negligible on the CPU)"}, counts)


def cpu_baseline_other_scenes(full):
    """the CPU port at 1 thread and on all cores, on S-50k-fern (forward only: BASELINE.json config 0, the
    reference's CPU-runnable case) and S-800k-chess (fwd+bwd, config 1).  Bounded: a few seconds each; the one-thread run
    of S-800k-chess takes minutes and only runs with --cpu-baseline-full."""
    from gs_localization_amd import scenes as S
    cores = os.cpu_count() or 1
    out = []
    fern, chess = S.s_50k_fern(), S.s_800k_chess()
    plan = [(fern, "S-50k-fern", False, 1, 4.0, 3), (fern, "S-50k-fern", False, min(cores, 16), 3.0, 20), (chess, "S-800k-chess", True, min(cores, 32), 6.0, 10)]
    if full:
        plan.append((chess, "S-800k-chess", True, 1, 1.0, 1))
    for sc, name, bwd, thr, budget, max_n in plan:
        n, el, _ = _cpu_time(sc, np.eye(4), thr, bwd, budget, max_n)
        out.append({"scene": name, "pass": "fwd+bwd" if bwd else "fwd", "cores": thr, "value": n / el, "unit": "iters/s", "kind": "port",
                    "sample": f"{n} iterations in {el:.1f} s"})
    return out


def _profile_ms(lib, fn, n):
    """per-kernel HIP-event milliseconds per call of fn() (all launches of each kernel id), n calls"""
    nk = lib.gsr_profile_kernel_count()
    names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]
    lib.gsr_profile_enable((1 << nk) - 1)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = (C.c_double * nk)()
    cnt = (C.c_longlong * nk)()
    lib.gsr_profile_collect(ms, cnt)
    lib.gsr_profile_enable(0)
    return {names[i]: ms[i] / n for i in range(nk)}, {names[i]: int(cnt[i]) for i in range(nk)}


def _stats_of(lib, grad_fn, P, W, H):
    """(V, R under the reference's bounding rule, list entries ordered, R_eff of the own binning) of the forward behind grad_fn"""
    from gs_localization_amd import _lib
    sv = grad_fn.saved_tensors
    st = (C.c_longlong * 4)()
    _lib.check(lib.gsr_forward_stats(P, W, H, sv[5].data_ptr(), sv[7].data_ptr(), sv[9].data_ptr(), st, torch.cuda.current_stream().cuda_stream))
    return tuple(int(st[i]) for i in range(4))


def _roofline(kernels_ms, bytes_per_kernel, profile_json, note):
    """`roofline` sub-object of a secondary workload: the kernel the step spends most time in, its SURVEY 8(d) bytes over its HIP-event
    duration against the HBM peak, and the counter traffic of the committed rocprofv3 pass (profiles/<profile_json>) if it is there."""
    dom = max(bytes_per_kernel, key=lambda k: kernels_ms.get(k, 0.0))
    ms = kernels_ms.get(dom, 0.0)
    ach = bytes_per_kernel[dom] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", profile_json)))
        # (the profiler names the preprocess kernels apart; the library's timing id "preprocess_fwd" covers all three)
        alias = {"preprocess_fwd": ("preprocess_bin", "preprocess_lean", "preprocess_fwd")}.get(dom, (dom,))
        traffic = next((tj[k]["hbm_bytes_corrected"] for k in alias if k in tj), None)
    except Exception:
        pass
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": bytes_per_kernel[dom], "avg_launch_ms": ms, "bound_measured": "valu_issue / wave latency (profiles/)", "note": note}


def cam_step_leg(lib, dev):
    """BASELINE.json config 3 (S-3M-cam: 3 M Gaussians, SH3, z in [2, 60] m) at the size BASELINE quotes (852x480)
    and at the size the reference's Cambridge script really renders (1024x576, cambridge_localize_full.py:366): refinement
    iterations/s of the native loop, speculative and with complete lists, 20 iterations per call like that script (:65), per-kernel
    times, and a roofline line for the kernel each spends most time in."""
    from gs_localization_amd import scenes as S
    from tests import replay as PL
    rows = []
    bg = torch.zeros(3, dtype=torch.float32, device=dev)
    for make, prof in ((S.s_3m_cam, "cam"), (S.s_3m_cam_1024, "cam1024")):
        sc = make()
        W, H, M = sc.W, sc.H, sc.shs.shape[1]
        N, ntiles = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        model = PL.GaussianMap.from_scene(sc, device=dev)
        frames = [PL.make_frame(sc, model, dev, bg, uid=u) for u in (0, 1)]
        inits = [PL.perturbed_start(2000 + u, device=dev) for u in (0, 1)]
        pkg = PL.render(frames[0], model, bg)
        V, R, R_ord, R_eff = _stats_of(lib, pkg["render"].grad_fn, sc.P, W, H)
        del pkg
        fr = PL.FusedRefiner(model, H, W, device=dev)
        K = 20

        def call(g, iters, spec, warm=None):
            return fr.refine(frames[g], PL.TRACKING_CONFIG, inits[g][:3, :3].clone(), inits[g][:3, 3].clone(), bg, iters=iters,
                             stop_on_converged=False, speculative=spec, warm_start=warm)
        res = {}
        for spec in (True, False):
            best = 1e9
            for _ in range(3):
                call(1, 3, spec)                      # the predecessor frame: its bounds are what the warm start gets
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                call(0, K, spec)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            kms, _ = _profile_ms(lib, lambda: call(0, K, spec), 2)
            res[spec] = (K / best, {k: round(v / K, 4) for k, v in kms.items() if v > 0})
        per, _ = algorithmic_bytes(sc.P, V, R, R_eff, N, M, ntiles)
        loop_kernels = {k: per[k] for k in ("render_fwd", "render_bwd", "preprocess_bwd", "preprocess_fwd")}
        rows.append({"scene": sc.name, "width": W, "height": H, "gaussians": sc.P, "tiles": ntiles, "V": V, "R": R, "R_eff_own_binning": R_eff,
                     "list_entries_ordered": R_ord, "iterations_per_call": K,
                     "speculative_iters_per_s": res[True][0], "plain_iters_per_s": res[False][0],
                     "kernels_ms_per_iter_speculative": res[True][1], "kernels_ms_per_iter_plain": res[False][1],
                     "roofline": _roofline({k: v for k, v in res[True][1].items()}, loop_kernels, f"r04_{prof}_spec_traffic.json",
                                           "speculative loop; R_eff of the own (culled) binning: no CPU oracle run at 3 M Gaussians"),
                     "roofline_plain": _roofline({k: v for k, v in res[False][1].items()}, loop_kernels, f"r04_{prof}_plain_traffic.json",
                                                 "complete lists in every iteration")})
        del fr, frames, model
        torch.cuda.empty_cache()
    return {"workload": "S-3M-cam pose refinement (BASELINE.json configs[3]), native loop, one frame at a time, 20 iterations per call", "per_size": rows}


def scene_variants_leg(lib, dev, K=50, only=None):
    """Structured variants of the headline scene (gs_localization_amd/scenes.py: "object" = half of the map inside a cone a tenth
    of the image wide, "walls" = two thin depth layers, "room" = S-room-640, a box room of flattened splats with bimodal
    opacities -- the nearest stand-in for a trained indoor map): refinement iterations/s of the native loop on ONE frame,
    speculative and with complete lists, K iterations per call, per-kernel times, and a roofline line for the kernel each
    spends most time in.  A uniform random cloud (S-1M-640, `value`) is the friendliest scene this path can get."""
    from gs_localization_amd import scenes as S
    from tests import replay as PL
    rows = []
    bg = torch.zeros(3, dtype=torch.float32, device=dev)
    for vname, make in S.VARIANTS.items():
        if only is not None and vname not in only:
            continue
        sc = make()
        W, H, M = sc.W, sc.H, sc.shs.shape[1]
        N, ntiles = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        model = PL.GaussianMap.from_scene(sc, device=dev)
        frames = [PL.make_frame(sc, model, dev, bg, uid=u) for u in (0, 1)]
        inits = [PL.perturbed_start(1000 + u, device=dev) for u in (0, 1)]
        pkg = PL.render(frames[0], model, bg)
        V, R, R_ord, R_eff = _stats_of(lib, pkg["render"].grad_fn, sc.P, W, H)
        del pkg
        fr = PL.FusedRefiner(model, H, W, device=dev)

        def call(g, iters, spec, warm=None, flags=None):
            return fr.refine(frames[g], PL.TRACKING_CONFIG, inits[g][:3, :3].clone(), inits[g][:3, 3].clone(), bg, iters=iters,
                             stop_on_converged=False, speculative=spec, warm_start=warm, flags=flags)
        res = {}
        for spec in (True, False):
            best, info = 1e9, None
            for _ in range(3):
                call(1, 5, spec)                      # the predecessor frame: its bounds are what the warm start gets
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, _, inf = call(0, K, spec)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
                if el < best:
                    best, info = el, {k: inf[k] for k in ("fallbacks", "host_redos", "lean_iters")}
            kms, _ = _profile_ms(lib, lambda: call(0, K, spec), 2)
            res[spec] = (K / best, {k: round(v / K, 4) for k, v in kms.items() if v > 0}, info)
        # (diagnostics: the speculative loop with the Gaussian-parameter gradient rows written by every iteration, GSR_REFINE_GRADS_EVERY_ITERATION)
        from gs_localization_amd import _lib as _L
        best_every = 1e9
        for _ in range(3):
            call(1, 5, True, flags=_L.REFINE_GRADS_EVERY_ITERATION)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            call(0, K, True, flags=_L.REFINE_GRADS_EVERY_ITERATION)
            torch.cuda.synchronize()
            best_every = min(best_every, time.perf_counter() - t0)
        # pose error of a full refinement with the reference's early exit
        Rr, Tt, inf = fr.refine(frames[0], PL.TRACKING_CONFIG, inits[0][:3, :3].clone(), inits[0][:3, 3].clone(), bg, iters=50, stop_on_converged=True)
        te, re = PL.pose_errors(np.eye(3), np.zeros(3), inf["R_host"], inf["T_host"])
        # Round 6: the same scene with SIXTEEN frames in flight (one refiner, stream and host thread each, as `value` runs the uniform
        # cloud).  Until round 6 no bench line measured this, and split tiles made it collapse (S-room-640: 81 it/s) -- gsr_api.hip,
        # GSR_SPLIT_MAX_CALLS.
        FF = 16
        ffr = [fr] + [PL.FusedRefiner(model, H, W, device=dev) for _ in range(FF - 1)]
        fframes = frames + [PL.make_frame(sc, model, dev, bg, uid=u) for u in range(2, FF)]
        finits = inits + [PL.perturbed_start(1000 + u, device=dev) for u in range(2, FF)]
        fstreams = [torch.cuda.Stream(device=dev) for _ in range(FF)]
        ffail = [0] * FF

        def fworker(s_, iters_):
            with torch.cuda.stream(fstreams[s_]):
                _, _, inf_ = ffr[s_].refine(fframes[s_], PL.TRACKING_CONFIG, finits[s_][:3, :3].clone(), finits[s_][:3, 3].clone(), bg, iters=iters_, stop_on_converged=False)
                ffail[s_] = inf_["fallbacks"]
                fstreams[s_].synchronize()
        best_ff = 1e9
        for rep_ in range(3):
            for s_ in range(FF):      # (predecessor: another frame, so that every timed call warm-starts from foreign bounds)
                with torch.cuda.stream(fstreams[s_]):
                    ffr[s_].refine(fframes[(s_ + 1) % FF], PL.TRACKING_CONFIG, finits[(s_ + 1) % FF][:3, :3].clone(), finits[(s_ + 1) % FF][:3, 3].clone(), bg, iters=5, stop_on_converged=False)
            torch.cuda.synchronize()
            th_ = [threading.Thread(target=fworker, args=(s_, K)) for s_ in range(FF)]
            t0 = time.perf_counter()
            [x.start() for x in th_]; [x.join() for x in th_]
            torch.cuda.synchronize()
            best_ff = min(best_ff, time.perf_counter() - t0)
        in_flight = {"frames": FF, "iters_per_s": FF * K / best_ff, "failed_forwards_per_frame_mean": float(np.mean(ffail))}
        del ffr, fframes
        per, _ = algorithmic_bytes(sc.P, V, R, R_eff, N, M, ntiles)
        loop_kernels = {k: per[k] for k in ("render_fwd", "render_bwd", "preprocess_bwd", "preprocess_fwd")}
        rows.append({"scene": sc.name, "variant": vname, "width": W, "height": H, "gaussians": sc.P, "tiles": ntiles, "V": V, "R": R,
                     "R_eff_own_binning": R_eff, "list_entries_ordered": R_ord, "iterations_per_call": K,
                     "speculative_iters_per_s": res[True][0], "plain_iters_per_s": res[False][0],
                     "speculative_iters_per_s_gradient_rows_every_iteration": K / best_every,
                     "speculative_16_frames_in_flight": in_flight,
                     "speculative_call_stats": res[True][2], "plain_call_stats": res[False][2],
                     "pose_err_cm_deg_after_refinement": [100.0 * te, re], "refine_iters": inf["iters"],
                     "kernels_ms_per_iter_speculative": res[True][1], "kernels_ms_per_iter_plain": res[False][1],
                     "roofline": _roofline({k: v for k, v in res[True][1].items()}, loop_kernels, f"r05_{vname}_spec_traffic.json",
                                           "speculative loop, one frame; R_eff of the own (culled) binning"),
                     "roofline_plain": _roofline({k: v for k, v in res[False][1].items()}, loop_kernels, f"r05_{vname}_plain_traffic.json",
                                                 "complete lists in every iteration")})
        del fr, frames, model
        torch.cuda.empty_cache()
    return {"workload": f"structured variants of S-1M-640 (1 M Gaussians, 640x480, SH3), native loop, one frame at a time, {K} iterations per call",
            "per_scene": rows}


def train_step_leg(lib):
    """BASELINE.json config 4 (train.py, S-train-garden): ms per step through package (A) + the fused loss epilogue + torch Adam,
    at three sizes of the model (1296x840, SH degree 1, white background, random camera per step); per-kernel times and a roofline
    line of the rasterizer at 1.5 M."""
    from tests.train_replay import TrainReplay, time_steps
    rows = []
    roof, kernels = None, None
    for P in (200_000, 800_000, 1_500_000):
        tr = TrainReplay(P0=P, P1=P, densify_from=10**9)
        r, it = time_steps(tr, 1, 30, warm=5)
        rows.append({k: (round(float(v), 4) if not isinstance(v, int) else v) for k, v in r.items()})
        if P == 1_500_000:
            state = {"it": it}

            def one():
                tr.step(state["it"]); state["it"] += 1
            kms, _ = _profile_ms(lib, one, 10)
            kernels = {k: round(v, 4) for k, v in kms.items() if v > 0}
            out = tr.render(tr.views[0])
            V, R, R_ord, R_eff = _stats_of(lib, out["image"].grad_fn, tr.P, tr.W, tr.H)
            N, ntiles = tr.W * tr.H, ((tr.W + 15) // 16) * ((tr.H + 15) // 16)
            per, _ = algorithmic_bytes(tr.P, V, R, R_eff, N, 4, ntiles)
            roof = _roofline(kms, {k: per[k] for k in ("render_fwd", "render_bwd", "preprocess_bwd", "preprocess_fwd", "tile_emit")},
                             "r04_train_traffic.json", f"V={V} R={R} R_eff(own binning)={R_eff}, {ntiles} tiles")
            del out
        del tr
        torch.cuda.empty_cache()
    return {"workload": "train.py step, 1296x840, SH1, white background, random camera per step, grad_depth != 0 (tests/train_replay.py)",
            "per_P": rows, "kernels_ms_per_step_1500000": kernels, "roofline": roof}


if __name__ == "__main__":
    main()
