#!/usr/bin/env python3
"""BASELINE.json config 2: a whole test split of query frames refined on the GPUs of one node.

The reference's driver (gs_localization/pipelines/7scenes_localize_full_dslam.py:352-389) walks the test images one by one
on one GPU: initial pose from the feature-matching stage, `gradient_decent` (up to 50 iterations, early exit on convergence),
pose errors, medians and the recall table.  Here: one process per GPU (torchrun), the map replicated on every rank, frames
handed out on demand from a shared counter (gs_localization_amd/shard.py -- a frame costs 1 ... 50 iterations), F frames in
flight per GPU on the native loop (`FusedRefiner`), ONE gather of the result rows at the end (RCCL over xGMI).

No dataset exists here: the split is synthetic -- a map of --gaussians Gaussians (S-800k-chess by default), --frames query
poses scattered --spread (0.1 m / 4 deg) around the map's reference view, each observed as the map's own render at that pose, each
started from an initial pose up to 5 cm / 3 deg off (a different amount per frame, so that iteration counts differ).

  python tools/localize_split.py --frames 64                                   # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
      tools/localize_split.py --frames 2000                                     # one node
Prints one JSON line on rank 0: frames/s, medians, recall, per-rank balance.  Scaling numbers need a multi-GPU node."""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--gaussians", type=int, default=800_000)
    ap.add_argument("--in-flight", type=int, default=8, help="frames refined concurrently per GPU")
    ap.add_argument("--preload", type=int, default=512, help="observations of the first PRELOAD frames are rendered before the clock starts "
                    "(the reference reads its query images from disk before refining them); later frames render theirs inside the loop")
    ap.add_argument("--assign", choices=("queue", "static"), default="queue")
    ap.add_argument("--chunk", type=int, default=1, help="frames claimed per trip to the shared counter")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--spread", type=float, nargs=2, default=(0.1, 4.0), metavar=("METRES", "DEGREES"),
                    help="query poses are scattered this far around the map's reference view.  The synthetic map is the frustum-shaped "
                         "cloud seen from that view: 0.3 m / 10 deg already looks past its edge (most tiles never saturate, no depth "
                         "bounds, complete lists: the stress case)")
    ap.add_argument("--mask", choices=("reference", "ones"), default="reference",
                    help="reference: every frame is refined under compute_grad_mask | create_mask(keypoints) as the reference's scripts build it "
                         "(7scenes_localize_full_dslam.py:355-360), computed by gsr_grad_mask inside the timed loop; ones: every pixel (rounds 1-5)")
    ap.add_argument("--gpus", type=int, default=None, help="without a launcher: start this many ranks (torch.distributed.run) and exit with their status")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo: the N > 1 path rehearsed on a box with one GPU (collectives on host tensors)")
    ap.add_argument("--device-index", type=int, default=None, help="GPU of this rank (default: LOCAL_RANK)")
    args = ap.parse_args()
    if args.gpus and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
                                  "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    # one hardware queue per frame in flight (the HIP runtime's default is four for all of a process's streams; see bench.py)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(4, min(16, args.in_flight))))
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    if args.gpus and args.gpus != world:
        sys.exit(f"localize_split.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank if args.device_index is None else args.device_index
    # (a launcher started this process -- also `--nproc-per-node 1`: the queue then counts in the store and the gather runs through
    # the backend, RCCL on device tensors, exactly as on eight GPUs; no launcher: no group, no collective)
    grouped = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ
    from gs_localization_amd import shard as _shard
    if grouped:
        _shard.init_process_group(args.backend, rank, world, device=torch.device("cuda", dev_index))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")
    from gs_localization_amd import scenes as S, shard
    from tests import replay as RP

    sc = S._draw("S-chess-split", args.gaussians, 640, 480, 525.0, 525.0, 0.5, 6.0, 0.01, 0.6, 3, 0)
    gmap = RP.GaussianMap.from_scene(sc, device=dev)
    bg = torch.zeros(3, device=dev)
    proj = RP.intrinsics_projection(sc, dev)
    F = max(1, args.in_flight)
    refiners = [RP.FusedRefiner(gmap, sc.H, sc.W, device=dev) for _ in range(F)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]

    def frame_setup(f):
        """ground-truth pose, observation and initial pose of query frame f (the same on whichever rank draws it)"""
        rng = np.random.default_rng(7000 + f)
        gt = S.se3_exp(np.concatenate([rng.uniform(-args.spread[0], args.spread[0], 3), np.radians(rng.uniform(-args.spread[1], args.spread[1], 3))]))
        off = rng.uniform(0.1, 1.0)                 # 0.5 ... 5 cm and 0.3 ... 3 deg: some frames converge early, some never
        dt = rng.normal(size=3); dt *= 0.05 * off / np.linalg.norm(dt)
        dr = rng.normal(size=3); dr *= math.radians(3.0 * off) / np.linalg.norm(dr)
        init = S.se3_exp(np.concatenate([dt, dr])) @ gt
        return gt, init

    full_mask = torch.ones((1, sc.H, sc.W), dtype=torch.bool, device=dev)
    loaded = {}

    def observe(f, gt):
        """the query frame's image and depth: the map's own render at the ground-truth pose (this tool's stand-in for a dataset)"""
        fr = RP.QueryFrame(f, proj, sc, dev, gt_w2c=torch.tensor(gt, dtype=torch.float32, device=dev))
        g = torch.tensor(gt, dtype=torch.float32, device=dev)
        fr.update_RT(g[:3, :3].clone(), g[:3, 3].clone())
        with torch.no_grad():
            obs = RP.render(fr, gmap, bg)
        fr.original_image, fr.depth = obs["render"].detach().clone(), obs["depth"].detach()[0].clone()
        fr.grad_mask = full_mask
        return fr

    def refine(slot, f):
        gt, init = frame_setup(f)
        with torch.cuda.stream(streams[slot]):
            fr = loaded.pop(f, None) or observe(f, gt)
            if args.mask == "reference":          # (per frame, in front of its refinement, like viewpoint.compute_grad_mask(config) + the keypoint boxes)
                fr.grad_mask = RP.reference_mask(fr.original_image, f)
            i0 = torch.tensor(init, dtype=torch.float32, device=dev)
            R, T, info = refiners[slot].refine(fr, RP.TRACKING_CONFIG, i0[:3, :3].clone(), i0[:3, 3].clone(), bg, iters=args.iters)
            te, re = RP.pose_errors(gt[:3, :3], gt[:3, 3], info["R_host"], info["T_host"])
        return te, re, float(info["iters"])

    refine(0, 0)                                    # warm-up (allocations, first-touch), untimed
    for f in range(min(args.frames, args.preload)):
        loaded[f] = observe(f, frame_setup(f)[0])
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    t0 = time.perf_counter()
    local, busy = shard.run_split(args.frames, refine, rank, world, slots=F, assign=args.assign, chunk=args.chunk)
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0
    if grouped:
        dist.barrier()
    wall = time.perf_counter() - t0
    res = shard.gather_results(local.to(coll_dev), args.frames, rank, world)
    stats = torch.tensor([t_rank, sum(busy) / F, float(local.shape[0]), float(local[:, 3].sum()) if local.numel() else 0.0],
                         dtype=torch.float64, device=coll_dev)
    per_rank = [torch.zeros_like(stats) for _ in range(world)]
    if grouped:
        dist.all_gather(per_rank, stats)
    else:
        per_rank = [stats]
    if rank == 0:
        res = res.cpu()
        m = shard.median_errors(res)
        pr = torch.stack(per_rank).cpu().numpy()
        out = {"workload": f"synthetic test split: {args.frames} query frames within {args.spread[0]} m / {args.spread[1]} deg, {args.gaussians} Gaussians, 640x480, up to {args.iters} iterations each",
               "n_gpus": world, "collectives": (args.backend if grouped else "none"), "frames_in_flight_per_gpu": F, "assign": args.assign, "grad_mask": args.mask, "frames_per_s": args.frames / wall,
               "iterations_per_s": float(res[:, 3].sum()) / wall, "wall_s": wall,
               "median_trans_err_cm": 100.0 * m["median_t_m"], "median_rot_err_deg": m["median_R_deg"], "recall": m["recall"],
               "iterations_per_frame": {"min": float(res[:, 3].min()), "median": float(res[:, 3].median()), "max": float(res[:, 3].max())},
               "per_rank": {"seconds_until_idle": [round(float(x), 3) for x in pr[:, 0]], "frames": [int(x) for x in pr[:, 2]],
                            "iterations": [int(x) for x in pr[:, 3]]},
               "balance_max_over_mean_idle_time": float(pr[:, 0].max() / pr[:, 0].mean())}
        print(json.dumps(out), flush=True)
    if grouped:
        _shard.destroy_process_group()


if __name__ == "__main__":
    main()
