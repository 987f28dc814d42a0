"""Multi-GPU sharding of the localisation workload.

Query frames are independent (gs_localization/pipelines/7scenes_localize_full_dslam.py:352-377: one `gradient_decent` per
test image against a read-only map), so one process per GPU refines its own frames against its own replica of the map and
the only exchange on the data path is ONE gather of the per-frame result rows at the end (RCCL over xGMI on GPUs, gloo in
the CPU tests); medians and the recall table are computed on rank 0 exactly as :381-389.

Which rank refines which frame.  A refinement stops when `update_pose` reports convergence (:88-91), so a frame costs
anything between 1 and 50 iterations and a static split can leave ranks idle.  `FrameQueue` hands frames out on demand from
one shared counter -- an atomic add on the process group's rendezvous store (a control-plane TCP round trip per claim of
`chunk` frames, nothing on the GPUs' links) -- so that every rank, and every frame slot in flight on it, takes the next
unclaimed frames the moment it is free.  `assign="static"` is the round-robin split (no store traffic at all).
"""
import datetime
import os
import threading
import time

import torch
import torch.distributed as dist

_STORE = None      # the rendezvous store of the process group init_process_group() below created (the frame queue counts in it)


def init_process_group(backend, rank, world, device=None, timeout_s=600):
    """One process per GPU: creates the rendezvous store ITSELF (public API only: `dist.TCPStore` on MASTER_ADDR / MASTER_PORT,
    the way torch's own env:// rendezvous does -- under torchrun the elastic agent already serves that port and every worker
    connects as a client) and hands it to `dist.init_process_group`, so that `FrameQueue` can count in it without reaching
    into torch's private `_get_default_store()`.  backend "nccl" is RCCL on ROCm (`device` = this rank's GPU, bound eagerly
    through `device_id`); "gloo" runs the same code on host tensors.  A world of ONE rank is a valid group: the collectives
    below then still go through the backend (that is how the RCCL path is executed on a one-GPU box)."""
    global _STORE
    addr = os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ["MASTER_PORT"])
    timeout = datetime.timedelta(seconds=timeout_s)
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True":
        base = dist.TCPStore(addr, port, world, False, timeout)
        store = dist.PrefixStore(f"/worker/attempt_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}", base)
    else:
        store = dist.TCPStore(addr, port, world, rank == 0, timeout, multi_tenant=True)
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = torch.device(device)
    dist.init_process_group(backend, store=store, rank=rank, world_size=world, timeout=timeout, **kw)
    _STORE = store
    return store


def destroy_process_group():
    global _STORE
    _STORE = None
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def group_active():
    return dist.is_available() and dist.is_initialized()


def shard_frames(n_frames, rank, world):
    """round-robin frame ids of this rank"""
    return list(range(rank, n_frames, world))


_queue_generation = [0]      # queues created so far in this process (every rank creates them in the same order)


class FrameQueue:
    """Hands out frame ids [0, n_frames) exactly once across all ranks and threads.

    The shared counter lives in the process group's rendezvous store under `key` and is never reset, so every queue needs a
    key of its own.  key=None (the default) derives one from a per-process generation count: all ranks construct their queues
    in the same order (run_split is a collective in that sense), so the n-th queue gets the same fresh key everywhere -- a
    warm-up pass followed by a measured pass, or several sequences, just work.  A caller that passes `key` itself must make
    it unique per run.  `store` is the c10d store to count in (default: the default process group's)."""

    def __init__(self, n_frames, rank=0, world=1, assign="queue", chunk=1, key=None, store=None):
        self.n, self.rank, self.world, self.assign, self.chunk = int(n_frames), rank, world, assign, max(1, int(chunk))
        self._lock = threading.Lock()
        self._local = []            # frames claimed but not yet handed to a worker
        self._next_static = rank
        self._counter = 0
        self._store = None
        if key is None:
            _queue_generation[0] += 1
            key = f"gsr_frames/{_queue_generation[0]}"
        if assign == "queue" and (world > 1 or store is not None or (_STORE is not None and group_active())):
            base = store if store is not None else _STORE
            if base is None:
                raise RuntimeError("FrameQueue: no rendezvous store -- initialise the process group with shard.init_process_group() "
                                   "(which keeps the store it creates) or pass store=")
            self._store = dist.PrefixStore(key, base)
        elif assign not in ("queue", "static"):
            raise ValueError("assign must be 'queue' or 'static'")

    def claim(self):
        """next frame id for the calling worker, or None when everything has been handed out"""
        with self._lock:
            if self._local:
                return self._local.pop(0)
            if self.assign == "static":
                f = self._next_static
                if f >= self.n:
                    return None
                self._next_static += self.world
                return f
            if self._store is not None:
                end = int(self._store.add("next", self.chunk))          # atomic fetch-and-add on the rendezvous store
            else:
                self._counter += self.chunk
                end = self._counter
            lo, hi = end - self.chunk, min(end, self.n)
            if lo >= self.n:
                return None
            self._local = list(range(lo + 1, hi))
            return lo


def run_split(n_frames, refine_fn, rank=0, world=1, slots=1, assign="queue", chunk=1, row_width=5, key=None):
    """Every rank calls this.  refine_fn(slot, frame_id) -> sequence of floats (the frame's result row WITHOUT the leading
    frame id, at most row_width - 2 values); `slots` worker threads per rank call it concurrently (frames in flight on one
    GPU).  Returns (local rows [n_local, row_width] float64 with columns frame_id, values..., rank; busy seconds per slot)."""
    q = FrameQueue(n_frames, rank, world, assign, chunk, key)
    rows, busy, errors = [], [0.0] * slots, []
    lock = threading.Lock()

    def work(slot):
        try:
            while True:
                f = q.claim()
                if f is None:
                    return
                t0 = time.perf_counter()
                vals = list(refine_fn(slot, f))
                busy[slot] += time.perf_counter() - t0
                row = [float(f)] + [float(v) for v in vals][: row_width - 2]
                row += [0.0] * (row_width - 1 - len(row)) + [float(rank)]
                with lock:
                    rows.append(row)
        except Exception as ex:      # re-raised by the caller's thread
            errors.append(ex)
    ts = [threading.Thread(target=work, args=(s,)) for s in range(slots)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    if errors:
        raise errors[0]
    local = torch.tensor(rows, dtype=torch.float64).reshape(-1, row_width)
    return local, busy


def gather_results(local, n_frames, rank, world, group=None):
    """local: [n_local, K] rows (frame_id, ...) of this rank, any number of them.  Returns [n_frames, K] sorted by frame id
    on rank 0 (None elsewhere): one all_gather of the row counts and one of the rows, padded to the largest shard.  Without a
    process group (a plain one-GPU run) there is nothing to gather; WITH one the collectives run even for a world of one rank."""
    if world == 1 and not group_active():
        return local[torch.argsort(local[:, 0])]
    K = local.shape[1]
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    n_max = max(1, int(max(int(c.item()) for c in counts)))
    pad = torch.full((n_max, K), -1.0, dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    if rank != 0:
        return None
    allr = torch.cat([b[: int(c.item())] for b, c in zip(bufs, counts)], 0)
    assert allr.shape[0] == n_frames, f"{allr.shape[0]} result rows for {n_frames} frames"
    return allr[torch.argsort(allr[:, 0])]


def median_errors(results):
    """results [n,K] with columns (frame_id, trans_err_m, rot_err_deg, ...) -> medians + the recall
    table of 7scenes_localize_full_dslam.py:381-389"""
    te, re = results[:, 1], results[:, 2]
    # (numpy's median, as the reference computes it: the mean of the two middle values for an even count)
    out = {"median_t_m": float(te.double().quantile(0.5)), "median_R_deg": float(re.double().quantile(0.5)), "recall": {}}
    for th_t, th_R in zip([0.01, 0.02, 0.03, 0.05, 0.25, 0.5, 5.0], [1.0, 2.0, 3.0, 5.0, 2.0, 5.0, 10.0]):
        out["recall"][f"{th_t * 100:.0f}cm,{th_R:.0f}deg"] = float(((te < th_t) & (re < th_R)).double().mean())
    return out
