#!/usr/bin/env python3
"""bench.py -- render+backward pose-refinement iterations/s on the headline scene S-1M-640
(640x480, 1 M Gaussians, SH3), one process per GPU.

One "step" = one body of the reference's refinement loop
(gs_localization/pipelines/7scenes_localize_full_dslam.py:66-91): render() through
`diff_gaussian_rasterization_pose` -> tracking loss -> backward (all Gaussian gradients + dL/dtau)
-> Adam step -> update_pose.  Frames are independent, so with N GPUs every rank refines its own
query frame against its own replica of the map (weak scaling, no data-path collective); the only
collective is the final gather of poses / timings.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel's algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (a port, oracle/gs_oracle.c, OpenMP) on the same scene, rank 0, N=1
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md, "HBM3E peak BW 8.0 TB/s spec"


def algorithmic_bytes(P, V, R, R_eff, N, M, ntiles):
    """SURVEY.md section 8(d) 'Algorithmic bytes per fwd+bwd iteration', split per kernel."""
    passes = math.ceil((32 + math.ceil(math.log2(ntiles))) / 8)
    per = {
        "preprocess_fwd": P * (44 + 12 * M) + V * 48,
        "scan": 0,
        "emit": R * 12,
        "sort": passes * R * 24,
        "ranges": R * 8,
        "render_fwd": R_eff * 44 + N * 24,
        "bwd_zero": 0,
        "render_bwd": N * 24 + R_eff * 44 + R_eff * 36,
        "preprocess_bwd": V * (48 + 36) + P * (44 + 12 * M) + P * (40 + 12 * M),
        "sh_color": 0,      # bytes are counted in preprocess_fwd (the colour half of the reference's K1)
        "depth_sort": 0,    # our own extra pass (sorting P Gaussians by depth); not part of the reference's byte model
    }
    return per, passes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pose-only", action="store_true", help="map tensors do not require grad (not the headline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from gs_localization_amd import _lib, scenes as S, pipelines as PL, shard
    lib = _lib.load()
    assert lib.gsr_device_ok() == 1, "no gfx950 device"

    sc = S.s_1m_640(P=args.gaussians)
    W, H, M = sc.W, sc.H, sc.shs.shape[1]
    N, ntiles = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    model = PL.GaussianMap.from_scene(sc, device=dev, requires_grad=not args.pose_only)
    pipe = PL.PipelineParams()
    background = torch.zeros(3, dtype=torch.float32, device=dev)
    proj = PL.getProjectionMatrix2(znear=0.01, zfar=100.0, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, W=W, H=H).transpose(0, 1).to(dev)
    fovx, fovy = PL.focal2fov(sc.fx, W), PL.focal2fov(sc.fy, H)
    config = PL.TRACKING_CONFIG

    def make_view(uid, w2c_gt):
        gt = torch.tensor(w2c_gt, dtype=torch.float32, device=dev)
        vp = PL.Camera(uid, None, None, gt, proj, sc.fx, sc.fy, sc.cx, sc.cy, fovx, fovy, H, W, device=dev)
        vp.update_RT(gt[:3, :3].clone(), gt[:3, 3].clone())
        with torch.no_grad():
            pkg = PL.render(vp, model, pipe, background)
        vp.original_image = pkg["render"].detach().clone()
        vp.depth = pkg["depth"].detach()[0].clone()
        vp.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
        return vp

    # query frame of this rank: GT pose = identity, start pose perturbed by (2 cm, 1 deg) (SURVEY 8(c) fixture 9)
    frame_id = shard.shard_frames(world, rank, world)[0]
    rng = np.random.default_rng(1000 + frame_id)
    d_t = rng.normal(size=3); d_t *= 0.02 / np.linalg.norm(d_t)
    d_r = rng.normal(size=3); d_r *= math.radians(1.0) / np.linalg.norm(d_r)
    w2c_gt = np.eye(4)
    w2c_init = S.se3_exp(np.concatenate([d_t, d_r])) @ w2c_gt
    vp = make_view(frame_id, w2c_gt)
    init = torch.tensor(w2c_init, dtype=torch.float32, device=dev)

    def reset():
        vp.update_RT(init[:3, :3].clone(), init[:3, 3].clone())
        for p_ in (vp.cam_rot_delta, vp.cam_trans_delta, vp.exposure_a, vp.exposure_b):
            p_.data.zero_()
        return PL.make_pose_optimizer(vp)

    def barrier():
        if world > 1:
            dist.barrier()

    nk = lib.gsr_profile_kernel_count()
    names = [lib.gsr_profile_kernel_name(i).decode() for i in range(nk)]

    def collect():
        ms = (C.c_double * nk)()
        cnt = (C.c_longlong * nk)()
        _lib.check(lib.gsr_profile_collect(ms, cnt))
        return {names[i]: (ms[i], cnt[i]) for i in range(nk)}

    # ---- warmup (W untimed steps) with every kernel bracketed by HIP events -> per-kernel breakdown
    opt = reset()
    lib.gsr_profile_enable((1 << nk) - 1)
    last_pkg = None
    for _ in range(max(args.warmup, 1)):
        conv, last_pkg = PL.refine_iteration(vp, config, model, pipe, background, opt)
        bool(conv)
    torch.cuda.synchronize()
    warm = collect()
    kernels_ms = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in warm.items()}
    dominant = max(kernels_ms, key=kernels_ms.get)

    # scene statistics of the last forward (V, R, R_eff)
    stats = (C.c_longlong * 4)()
    del last_pkg
    reset()                                               # statistics are quoted at the start pose
    last_pkg = PL.render(vp, model, pipe, background)     # fresh graph: saved tensors still alive
    rs_saved = last_pkg["render"].grad_fn
    geom_t, img_t, radii_t = rs_saved.saved_tensors[7], rs_saved.saved_tensors[9], rs_saved.saved_tensors[5]
    _lib.check(lib.gsr_forward_stats(sc.P, W, H, radii_t.data_ptr(), geom_t.data_ptr(), img_t.data_ptr(), stats,
                                     torch.cuda.current_stream().cuda_stream))
    V, R, R_emit, R_eff_culled = (int(stats[i]) for i in range(4))
    del last_pkg, rs_saved
    # the byte model is defined on the REFERENCE's binning (bounding-square rule), whatever the
    # implementation emits.  V and R under that rule come from the GPU stats; R_eff under that rule needs the
    # reference lists, so it is taken from the CPU oracle's forward at this pose (cpu_baseline leg).
    cpu = None
    R_eff, R_eff_src = R_eff_culled, "own culled binning (no CPU oracle run)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, ref_counts = cpu_baseline(sc, w2c_init)
        if ref_counts["V"] == V and ref_counts["R"] == R:
            R_eff, R_eff_src = ref_counts["R_eff"], "reference bounding rule (CPU oracle at the same pose)"
    per_kernel_bytes, passes = algorithmic_bytes(sc.P, V, R, R_eff, N, M, ntiles)
    total_bytes = sum(per_kernel_bytes.values())

    # ---- (a) the reference's own Python loop on top of the drop-in packages (torch autograd, Adam, update_pose)
    opt = reset()
    lib.gsr_profile_enable(0)
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        conv, _pkg = PL.refine_iteration(vp, config, model, pipe, background, opt)
        bool(conv)          # the reference's `if converged: break` forces this host sync every iteration
    torch.cuda.synchronize(); barrier()
    elapsed_py = time.perf_counter() - t0
    del _pkg

    # ---- (b) timed region of the headline value: the same K iterations through the native loop
    # (gsr_refine: render -> tracking loss -> backward with ALL Gaussian gradients + dL/dtau -> Adam ->
    # update_pose -> convergence flag); only the dominant kernel is bracketed by HIP events.
    fr = PL.FusedRefiner(model, H, W, device=dev, gaussian_grads=not args.pose_only)
    reset()
    fr.refine(vp, config, init[:3, :3].clone(), init[:3, 3].clone(), background, iters=max(args.warmup, 1), stop_on_converged=False)
    reset()
    lib.gsr_profile_enable(1 << names.index(dominant))
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fr.refine(vp, config, init[:3, :3].clone(), init[:3, 3].clone(), background, iters=args.steps, stop_on_converged=False)
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    dom_ms, dom_n = collect()[dominant]
    lib.gsr_profile_enable(0)
    if world > 1:
        t = torch.tensor([elapsed, elapsed_py], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_py = float(t[0].item()), float(t[1].item())

    # ---- pose error of a full 50-iteration refinement (untimed), gathered over ranks
    timed_info = fr.last_info
    # per-kernel breakdown of the native loop (separate short run, every kernel bracketed)
    reset()
    lib.gsr_profile_enable((1 << nk) - 1)
    fr.refine(vp, config, init[:3, :3].clone(), init[:3, 3].clone(), background, iters=20, stop_on_converged=False)
    torch.cuda.synchronize()
    native_ms = {k: round(v[0] / v[1], 4) if v[1] else 0.0 for k, v in collect().items()}
    lib.gsr_profile_enable(0)
    reset()
    Rr, Tt, _ = fr.refine(vp, config, init[:3, :3].clone(), init[:3, 3].clone(), background, iters=50)
    te, re = PL.pose_errors(w2c_gt[:3, :3], w2c_gt[:3, 3], Rr.detach().cpu().numpy(), Tt.detach().cpu().numpy())
    te0, re0 = PL.pose_errors(w2c_gt[:3, :3], w2c_gt[:3, 3], w2c_init[:3, :3], w2c_init[:3, 3])
    res = shard.gather_results(torch.tensor([[float(frame_id), te, re]], dtype=torch.float64, device=dev), world, rank, world)

    if rank == 0:
        res = res.cpu().numpy()
        dom_avg_ms = dom_ms / max(dom_n, 1)
        achieved = per_kernel_bytes[dominant] / (dom_avg_ms * 1e-3) / 1e9 if dom_avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dominant)
            except Exception:
                traffic = None
        out = {
            "metric": "render+backward iters/sec @640x480, 1M Gaussians; median pose err (cm/deg)",
            "value": world * args.steps / elapsed,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "S-1M-640 pose refinement (BASELINE.json configs[1] shape at 1M Gaussians)",
                       "width": W, "height": H, "gaussians": sc.P, "sh_degree": sc.sh_degree,
                       "V": V, "R": R, "R_emitted": R_emit, "R_eff": R_eff, "R_eff_source": R_eff_src,
                       "R_eff_own_binning": R_eff_culled, "sort_passes": passes,
                       "algorithmic_bytes_per_iter": total_bytes, "frames_per_rank": 1,
                       "gaussian_grads": not args.pose_only, "parallelism": f"frames x{world}",
                       "loop": "native gsr_refine (render, tracking loss, backward, Adam, update_pose per iteration)",
                       "speculative_binning": {"redone_forwards": timed_info["fallbacks"], "num_rendered_last": timed_info["num_rendered"]}},
            "python_loop_iters_per_s": world * args.steps / elapsed_py,
            "pose_err_cm_median": 100.0 * float(np.median(res[:, 1])),
            "pose_err_deg_median": float(np.median(res[:, 2])),
            "pose_err_init_cm_deg": [100.0 * te0, re0],
            "kernels_ms": {k: round(v, 4) for k, v in kernels_ms.items()},
            "native_loop_kernels_ms": native_ms,
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": per_kernel_bytes[dominant], "avg_launch_ms": dom_avg_ms,
                         "whole_iter_frac": total_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(sc, w2c):
    """The oracle (a port of the reference algorithm, not the reference itself) on the host cores:
    fwd+bwd of the rasterizer with pose gradients on the same scene and pose; bounded to ~10-30 s.
    Also returns the reference-rule counts (V, R, R_eff) of that forward."""
    from oracle import oracle as O
    from gs_localization_amd import scenes as S
    cores = os.cpu_count() or 1
    O.set_threads(cores)
    view, proj, _, campos = S.camera_matrices(sc, w2c)
    rng = np.random.default_rng(0)
    gc = rng.normal(size=(3, sc.H, sc.W)).astype(np.float32)
    gd = rng.normal(size=(1, sc.H, sc.W)).astype(np.float32)
    ga = np.zeros((1, sc.H, sc.W), np.float32)
    n, t0 = 0, time.perf_counter()
    while True:
        f = O.forward(sc.means3D, sc.opacities, view, proj, campos, sc.W, sc.H, sc.tanfovx, sc.tanfovy, sc.bg,
                      sh_degree=sc.sh_degree, shs=sc.shs, scales=sc.scales, rotations=sc.rotations, want_n_touched=True)
        O.backward(f, gc, gd, ga, pose_mode=True)
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 20:
            break
    counts = {"V": int((f.radii > 0).sum()), "R": int(f.num_rendered), "R_eff": O.r_eff(f)}
    return ({"value": n / el, "unit": "iters/s", "cores": cores, "kind": "port",
             "sample": f"{n} rasterizer fwd+bwd iterations (pose gradients) of the same S-1M-640 scene and pose, "
                       "OpenMP over tiles/Gaussians; excludes loss/Adam/update_pose (negligible on the CPU)"}, counts)


if __name__ == "__main__":
    main()
