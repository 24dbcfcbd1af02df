"""Job-level communication of the multi-GPU path (one process per GPU): ONE broadcast of the
packed robot model + task set from rank 0, a barrier and a few timing reductions for the drivers.  There is no
per-step collective.

Backends behind one small interface (:func:`create`):

* ``"rccl"``  -- the library's own communicator (``gmr_comm_*``: RCCL opened with dlopen, the ranks form a TCP control
  star at ``MASTER_ADDR`` over which the ncclUniqueId travels and every bring-up step is agreed on; no PyTorch).  The
  default when more than one rank runs on GPUs.  A failure is the same :class:`GmrHipError` on EVERY rank (there is no
  silent per-rank fallback: a rank that switched backends alone would leave its peers in a rendezvous nobody joins);
  ``GMR_COMM_FALLBACK=tcp`` lets the whole job continue on the control star instead, labelled as such.
* ``"tcp"``   -- the control star alone (job-level plumbing only; CPU rehearsal of the N>1 path without torch).
* ``"torch"`` -- ``torch.distributed`` (``nccl`` = RCCL on ROCm, or ``gloo`` for the CPU rehearsal of the N>1 path in
  the test-suite): optional plumbing for callers that already live in a torch process group.
* ``"none"``  -- a single rank.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def default_port() -> int:
    """Port of the ncclUniqueId exchange: ``GMR_COMM_PORT`` or MASTER_PORT + 1 (MASTER_PORT itself belongs to the
    launcher's own store under ``torch.distributed.run``)."""
    if os.environ.get("GMR_COMM_PORT"):
        return int(os.environ["GMR_COMM_PORT"])
    return int(os.environ.get("MASTER_PORT", "29500")) + 1


class SingleComm:
    backend, rank, world = "none", 0, 1

    def broadcast_bytes(self, buf: Optional[np.ndarray], nbytes: int, root: int = 0) -> np.ndarray:
        return np.ascontiguousarray(buf, dtype=np.uint8)

    def barrier(self):
        from . import _lib
        if _lib.lib().gmr_device_count() > 0:        # (a CPU-only rehearsal has nothing to wait for)
            _lib.check(_lib.lib().gmr_stream_sync(None))

    def allreduce_max(self, x: float) -> float:
        return float(x)

    def allgather(self, x: float):
        return [float(x)]

    def close(self):
        pass


class RcclComm:
    """``gmr_comm_*`` of libgmrhip.so.  Call after ``gmr_set_device(local_rank)``.  ``backend`` is what the library
    reports: ``"rccl-<version>"``, ``"tcp"`` or ``"tcp (fallback: ...)"``."""

    def __init__(self, rank: int, world: int, addr: Optional[str] = None, port: Optional[int] = None, tcp: bool = False):
        from . import _lib
        self._lib = _lib
        h = C.c_void_p()
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        saved = os.environ.get("GMR_COMM_BACKEND")
        if tcp:
            os.environ["GMR_COMM_BACKEND"] = "tcp"         # read by gmr_comm_create
        elif saved == "tcp":
            del os.environ["GMR_COMM_BACKEND"]
        try:
            _lib.check(_lib.lib().gmr_comm_create(rank, world, addr.encode(), int(port or default_port()), C.byref(h)))
        finally:
            if saved is None:
                os.environ.pop("GMR_COMM_BACKEND", None)
            else:
                os.environ["GMR_COMM_BACKEND"] = saved
        self.handle = h
        self.rank, self.world = rank, world
        self.backend = _lib.lib().gmr_comm_backend(h).decode()

    def broadcast_bytes(self, buf, nbytes, root=0):
        out = np.zeros(nbytes, dtype=np.uint8) if buf is None else np.ascontiguousarray(buf, dtype=np.uint8).copy()
        assert out.nbytes == nbytes
        self._lib.check(self._lib.lib().gmr_comm_broadcast(self.handle, out.ctypes.data_as(C.c_void_p), nbytes, root))
        return out

    def barrier(self):
        self._lib.check(self._lib.lib().gmr_comm_barrier(self.handle))

    def allreduce_max(self, x):
        a = np.array([x], dtype=np.float64)
        self._lib.check(self._lib.lib().gmr_comm_allreduce_max(self.handle, a.ctypes.data_as(C.c_void_p), 1))
        return float(a[0])

    def allgather(self, x):
        a = np.array([x], dtype=np.float64)
        out = np.zeros(self.world, dtype=np.float64)
        self._lib.check(self._lib.lib().gmr_comm_allgather(self.handle, a.ctypes.data_as(C.c_void_p),
                                                           out.ctypes.data_as(C.c_void_p), 1))
        return [float(v) for v in out]

    def close(self):
        if self.handle:
            self._lib.lib().gmr_comm_destroy(self.handle)
            self.handle = None


class TorchComm:
    """``torch.distributed`` process group (``nccl`` on GPUs, ``gloo`` on CPUs)."""

    def __init__(self, rank: int, local_rank: int, world: int, backend: str = "nccl"):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.backend = "torch-" + backend
        self._own = not dist.is_initialized()
        if self._own:
            if backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = "cuda" if backend == "nccl" else "cpu"

    def _sync(self):
        if self.dev == "cuda":
            self.torch.cuda.synchronize()

    def broadcast_bytes(self, buf, nbytes, root=0):
        arr = np.zeros(nbytes, dtype=np.uint8) if buf is None else np.ascontiguousarray(buf, dtype=np.uint8).copy()
        t = self.torch.from_numpy(arr).to(self.dev)
        self.dist.broadcast(t, src=root)
        return t.cpu().numpy()

    def barrier(self):
        from . import _lib
        if _lib.lib().gmr_device_count() > 0:        # the library's launches must be complete, whatever carries the barrier
            _lib.check(_lib.lib().gmr_stream_sync(None))
        self._sync()
        self.dist.barrier()
        self._sync()

    def allreduce_max(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allgather(self, x):
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def close(self):
        if self._own and self.dist.is_initialized():
            self.dist.destroy_process_group()


def create(backend: Optional[str] = None, force: bool = False):
    """Communicator of this process from the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    ``backend``: "rccl" (default for world > 1), "tcp", "torch" / "torch-nccl", "gloo" / "torch-gloo", "none".
    ``force``: build a real communicator even for a single rank (rehearsal of the N>1 path on one GPU).
    A failure raises on every rank (see the module docstring); start a fresh job with another backend if wanted."""
    rank, local_rank, world = env_rank_world()
    backend = (backend or os.environ.get("GMR_COMM_BACKEND") or "rccl").lower()
    if backend == "none" or (world == 1 and not force):
        return SingleComm()
    if backend == "rccl":
        return RcclComm(rank, world)
    if backend == "tcp":
        return RcclComm(rank, world, tcp=True)
    if backend in ("torch", "torch-nccl", "nccl"):
        return TorchComm(rank, local_rank, world, "nccl")
    if backend in ("gloo", "torch-gloo"):
        return TorchComm(rank, local_rank, world, "gloo")
    raise ValueError(f"unknown communication backend {backend!r}")
