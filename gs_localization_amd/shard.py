"""Multi-GPU sharding of the localisation workload: query frames are independent
(gs_localization/pipelines/7scenes_localize_full_dslam.py:352-377), so each rank (one process per
GPU) refines frames rank, rank+world, ... against its own replica of the map; the only exchange is
one gather of the per-frame results at the end (RCCL over xGMI on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_frames(n_frames, rank, world):
    """round-robin frame ids of this rank"""
    return list(range(rank, n_frames, world))


def gather_results(local, n_frames, rank, world, group=None):
    """local: [n_local, K] rows (frame_id, ...) of this rank.  Returns [n_frames, K] sorted by
    frame id on rank 0 (None elsewhere).  Ragged shards are padded to the largest shard."""
    if world == 1:
        return local[torch.argsort(local[:, 0])]
    K = local.shape[1]
    n_max = (n_frames + world - 1) // world
    pad = torch.full((n_max, K), -1.0, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    if rank != 0:
        return None
    allr = torch.cat(bufs, 0)
    allr = allr[allr[:, 0] >= 0]
    return allr[torch.argsort(allr[:, 0])]


def median_errors(results):
    """results [n,K] with columns (frame_id, trans_err_m, rot_err_deg, ...) -> medians + the recall
    table of 7scenes_localize_full_dslam.py:381-389"""
    te, re = results[:, 1], results[:, 2]
    out = {"median_t_m": float(te.median()), "median_R_deg": float(re.median()), "recall": {}}
    for th_t, th_R in zip([0.01, 0.02, 0.03, 0.05, 0.25, 0.5, 5.0], [1.0, 2.0, 3.0, 5.0, 2.0, 5.0, 10.0]):
        out["recall"][f"{th_t * 100:.0f}cm,{th_R:.0f}deg"] = float(((te < th_t) & (re < th_R)).double().mean())
    return out
