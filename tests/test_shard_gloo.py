"""CPU: the N>1 path on gloo, world_size 2 and 3 -- frames handed out from the shared counter (or round-robin), refined by a
stub whose cost varies per frame like the early-exit refinement's does, results gathered once and ordered by frame id."""
import os
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _cost_units(frame, world):
    return 8 if frame % world == 0 else 1          # adversarial for round-robin: rank 0 would own every expensive frame


def _worker(rank, world, port, n_frames, assign, slots, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from gs_localization_amd import shard
    shard.init_process_group("gloo", rank, world)

    def refine(slot, f):            # stands in for FusedRefiner.refine: (trans err, rot err, iterations)
        time.sleep(0.004 * _cost_units(f, world))
        return 0.001 * f, 0.1 * f, float(_cost_units(f, world))
    # a warm-up pass on the same process group first (the shared counter is never reset, so every run_split must count
    # under a key of its own -- the default derives a fresh one per call, identically on every rank)
    warm, _ = shard.run_split(3, lambda slot, f: (0.0, 0.0, 0.0), rank, world, slots=1, assign=assign)
    assert shard.gather_results(warm, 3, rank, world) is not None or rank != 0
    dist.barrier()
    t0 = time.perf_counter()
    local, busy = shard.run_split(n_frames, refine, rank, world, slots=slots, assign=assign, chunk=1)
    wall = time.perf_counter() - t0
    t = torch.tensor([wall, sum(busy)], dtype=torch.float64)
    walls = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(walls, t)
    res = shard.gather_results(local, n_frames, rank, world)
    if rank == 0:
        torch.save({"res": res, "walls": torch.stack(walls)}, out)
    else:
        assert res is None
    shard.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_queue_and_static_split_process_every_frame_once_in_order(tmp_path, world):
    n_frames = 36
    got = {}
    for k, assign in enumerate(("queue", "static")):
        out = str(tmp_path / f"res_{assign}.pt")
        mp.spawn(_worker, args=(world, 29610 + 10 * world + k, n_frames, assign, 2, out), nprocs=world, join=True)
        d = torch.load(out)
        res = d["res"]
        assert res.shape == (n_frames, 5)
        assert torch.equal(res[:, 0], torch.arange(n_frames, dtype=torch.float64))          # every frame once, ordered by id
        assert torch.allclose(res[:, 1], 0.001 * torch.arange(n_frames, dtype=torch.float64))
        got[assign] = (res, d["walls"])
    # static = round-robin owners; with this cost pattern rank 0 does all the expensive frames
    assert torch.equal(got["static"][0][:, 4], torch.arange(n_frames, dtype=torch.float64) % world)
    busy_static = got["static"][1][:, 1]
    assert busy_static[0] > 2.0 * busy_static[1:].max()
    # the shared counter evens it out: every rank is busy for about the same time and the job finishes sooner
    busy_q, wall_q = got["queue"][1][:, 1], got["queue"][1][:, 0]
    assert busy_q.max() <= 1.4 * busy_q.mean(), busy_q
    assert wall_q.max() < 0.8 * got["static"][1][:, 0].max()
    owners = got["queue"][0][:, 4]
    assert len(set(owners.tolist())) == world            # everybody took part


def test_ragged_and_tiny_splits(tmp_path):
    for n_frames in (7, 1):
        out = str(tmp_path / f"r{n_frames}.pt")
        mp.spawn(_worker, args=(2, 29700 + n_frames, n_frames, "queue", 1, out), nprocs=2, join=True)
        res = torch.load(out)["res"]
        assert res.shape == (n_frames, 5) and torch.equal(res[:, 0], torch.arange(n_frames, dtype=torch.float64))


def test_a_group_of_one_rank_still_runs_its_collectives(tmp_path):
    """`torch.distributed.run --nproc-per-node 1` builds a process group of ONE rank: the frame queue then counts in the rendezvous
    store and the result gather goes through the backend's all_gather (that is how the RCCL path is executed on a one-GPU box,
    tests/test_gpu_multirank.py); same rows as without a group."""
    out = str(tmp_path / "one.pt")
    mp.spawn(_worker, args=(1, 29790, 9, "queue", 2, out), nprocs=1, join=True)
    res = torch.load(out)["res"]
    assert res.shape == (9, 5) and torch.equal(res[:, 0], torch.arange(9, dtype=torch.float64))
    from gs_localization_amd import shard
    assert not shard.group_active()
    local, _ = shard.run_split(9, lambda slot, f: (0.001 * f, 0.1 * f, 1.0), 0, 1, slots=2)
    plain = shard.gather_results(local, 9, 0, 1)
    assert torch.allclose(plain[:, :3], res[:, :3])


def test_shard_is_a_partition():
    from gs_localization_amd import shard
    for n in (0, 1, 5, 16, 17):
        for w in (1, 2, 3, 8):
            allf = sorted(f for r in range(w) for f in shard.shard_frames(n, r, w))
            assert allf == list(range(n))
            for assign in ("static", "queue"):
                if assign == "queue" and w > 1:
                    continue              # (the shared counter needs a process group: covered above)
                seen = []
                for r in range(w):
                    q = shard.FrameQueue(n, r, w, assign, chunk=3)
                    while (f := q.claim()) is not None:
                        seen.append(f)
                assert sorted(seen) == list(range(n)), (n, w, assign)


def test_median_errors_table():
    from gs_localization_amd import shard
    res = torch.tensor([[0, 0.004, 0.3], [1, 0.02, 0.8], [2, 0.3, 4.0]], dtype=torch.float64)
    m = shard.median_errors(res)
    assert m["median_t_m"] == 0.02 and m["median_R_deg"] == 0.8
    assert abs(m["recall"]["1cm,1deg"] - 1 / 3) < 1e-12 and m["recall"]["500cm,10deg"] == 1.0
    even = torch.tensor([[0, 0.01, 1.0], [1, 0.03, 3.0]], dtype=torch.float64)          # numpy's median: mean of the two middle values
    assert abs(shard.median_errors(even)["median_t_m"] - 0.02) < 1e-15
