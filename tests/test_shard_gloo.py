"""CPU: the N>1 path (frames sharded over ranks, one gather at the end) on gloo, world_size 2."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_frames, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gs_localization_amd import shard
    mine = shard.shard_frames(n_frames, rank, world)
    local = torch.tensor([[float(f), 0.001 * f, 0.1 * f, float(rank)] for f in mine], dtype=torch.float64).reshape(-1, 4)
    res = shard.gather_results(local, n_frames, rank, world)
    if rank == 0:
        torch.save(res, out)
    else:
        assert res is None
    dist.destroy_process_group()


def test_shard_and_gather_world2(tmp_path):
    for n_frames in (7, 4, 1):
        out = str(tmp_path / f"res{n_frames}.pt")
        mp.spawn(_worker, args=(2, 29500 + n_frames, n_frames, out), nprocs=2, join=True)
        res = torch.load(out)
        assert res.shape == (n_frames, 4)
        assert torch.equal(res[:, 0], torch.arange(n_frames, dtype=torch.float64))
        assert torch.equal(res[:, 3], torch.arange(n_frames, dtype=torch.float64) % 2)   # round-robin owner
        assert torch.allclose(res[:, 1], 0.001 * torch.arange(n_frames, dtype=torch.float64))


def test_shard_is_a_partition():
    from gs_localization_amd import shard
    for n in (0, 1, 5, 16, 17):
        for w in (1, 2, 3, 8):
            allf = sorted(f for r in range(w) for f in shard.shard_frames(n, r, w))
            assert allf == list(range(n))


def test_median_errors_table():
    from gs_localization_amd import shard
    res = torch.tensor([[0, 0.004, 0.3], [1, 0.02, 0.8], [2, 0.3, 4.0]], dtype=torch.float64)
    m = shard.median_errors(res)
    assert m["median_t_m"] == 0.02 and m["median_R_deg"] == 0.8
    assert abs(m["recall"]["1cm,1deg"] - 1 / 3) < 1e-12 and m["recall"]["500cm,10deg"] == 1.0
