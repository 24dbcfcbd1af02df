"""Multi-GPU driver: streams shard across ranks, frames do not.

* :func:`lpt_partition` -- longest-processing-time-first assignment of ragged streams to ranks;
* :func:`broadcast_blobs` -- the ONE collective of the path: rank 0's packed (model, task set)
  bytes to every rank (RCCL over xGMI when the process group is "nccl"; gloo in the CPU tests);
* :func:`run_sharded` -- partition, run a compute function on the local shard, hand back the
  local results with their global stream ids.  No per-step collective, no all-reduce: every
  reduction of the pipeline (the per-clip min-z) is local to one stream, hence to one rank.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np

from .ik_config import MODEL_DTYPE, TASKSET_DTYPE


def lpt_partition(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Stream ids per rank; deterministic (ties by stream id), balanced on total frames."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(lengths[i])
    for p in parts:
        p.sort()
    return parts


def broadcast_blobs(model_blob, taskset_blob, rank: int, comm=None, device=None):
    """Rank 0 passes the packed structs, other ranks pass None; everybody gets both back.  ``comm`` is a
    :mod:`comm` communicator (RCCL through the library, or a torch process group); a ``torch.distributed`` module
    is accepted as well (callers that already live in a process group)."""
    nbytes = MODEL_DTYPE.itemsize + TASKSET_DTYPE.itemsize
    if comm is None:
        return model_blob, taskset_blob
    if hasattr(comm, "broadcast_bytes"):
        if comm.world == 1 and comm.backend == "none":
            return model_blob, taskset_blob
        buf = None
        if rank == 0:
            buf = np.concatenate([np.ascontiguousarray(model_blob).view(np.uint8).ravel(),
                                  np.ascontiguousarray(taskset_blob).view(np.uint8).ravel()])
        buf = comm.broadcast_bytes(buf, nbytes, 0)
    else:                                           # a torch.distributed module
        dist = comm
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return model_blob, taskset_blob
        import torch
        if rank == 0:
            buf = np.concatenate([np.ascontiguousarray(model_blob).view(np.uint8).ravel(),
                                  np.ascontiguousarray(taskset_blob).view(np.uint8).ravel()])
        else:
            buf = np.zeros(nbytes, dtype=np.uint8)
        t = torch.from_numpy(buf.copy())
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=0)
        buf = t.cpu().numpy()
    mb = buf[: MODEL_DTYPE.itemsize].view(MODEL_DTYPE).copy()
    ts = buf[MODEL_DTYPE.itemsize:].view(TASKSET_DTYPE).copy()
    return mb, ts


def run_sharded(compute: Callable, q0: np.ndarray, human: np.ndarray, lens: np.ndarray, rank: int,
                world: int) -> Tuple[List[int], tuple]:
    """Run ``compute(q0_local, human_local, lens_local)`` on this rank's streams.

    Returns ``(stream_ids, compute_result)``; concatenating the per-rank results ordered by stream
    id reproduces the single-rank result bit for bit (a stream's arithmetic does not depend on the
    batch it is launched in)."""
    ids = lpt_partition(lens, world)[rank]
    if len(ids) == 0:
        return ids, None
    return ids, compute(q0[ids], human[ids], np.asarray(lens)[ids])


def timed_steps(comm, step: Callable[[], None], steps: int, warmup: int) -> Tuple[float, List[float]]:
    """The timing protocol of every leg of bench.py: ``warmup`` untimed steps, a barrier (device-synchronising) on both
    sides of exactly ``steps`` timed ones, MAX over ranks.  Returns (seconds of the slowest rank, seconds per rank)."""
    import time
    for _ in range(warmup):
        step()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    comm.barrier()
    mine = time.perf_counter() - t0
    return comm.allreduce_max(mine), comm.allgather(mine)


def _device_sync():
    from . import _lib
    _lib.check(_lib.lib().gmr_stream_sync(None))


def strong_scaling_leg(comm, lens: Sequence[int], make_step: Callable[[List[int]], Callable[[], None]], steps: int,
                       warmup: int, single_step: Callable[[], Callable[[], None]] = None,
                       device_sync: Callable[[], None] = _device_sync) -> dict:
    """One fixed batch of streams (``lens[i]`` frames each) LPT-sharded over the ranks of ``comm``; no data-path
    collective.  ``make_step(stream_ids)`` prepares this rank's shard (inputs resident on its GPU) and returns the
    function that retargets it once.  With ``single_step`` rank 0 also runs the WHOLE batch alone on its GPU (the other
    ranks wait at a barrier), so that the efficiency ``value_N / (N * value_1)`` comes from one run on one set of
    devices.  Returns the numbers bench.py prints."""
    world, rank = comm.world, comm.rank
    parts = lpt_partition(lens, world)
    total = int(np.sum(np.asarray(lens, dtype=np.int64)))
    step = make_step(parts[rank])
    t_max, t_all = timed_steps(comm, step, steps, warmup)
    out = {"frames": total, "streams": len(lens), "world_size": world,
           "frames_per_rank": [int(np.sum(np.asarray(lens, dtype=np.int64)[p])) if len(p) else 0 for p in parts],
           "seconds": t_max, "per_rank_seconds": t_all, "value": total * steps / t_max}
    if single_step is not None:
        import time
        t1 = None
        if rank == 0:
            one = single_step()
            for _ in range(warmup):
                one()
            device_sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            device_sync()
            t1 = time.perf_counter() - t0
        comm.barrier()
        t1 = comm.allreduce_max(t1 if t1 is not None else 0.0)
        out["value_1gpu"] = total * steps / t1
        out["efficiency"] = out["value"] / (world * out["value_1gpu"])
    return out
