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


def broadcast_blobs(model_blob, taskset_blob, rank: int, dist=None, device=None):
    """Rank 0 passes the packed structs, other ranks pass None; everybody gets both back."""
    nbytes = MODEL_DTYPE.itemsize + TASKSET_DTYPE.itemsize
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
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
