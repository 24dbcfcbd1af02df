"""ctypes binding of libgmrhip.so (include/gmr_hip.h).  No torch, no fallback: if the library is
missing or no GPU is visible the product path raises -- there is deliberately no CPU path here."""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

from .ik_config import MODEL_DTYPE, TASKSET_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GMR_HIP_LIBRARY") or os.path.join(_HERE, "libgmrhip.so")   # override: A/B builds of the library
_lib = None

FLAG_OFFSET_TO_GROUND = 1
FLAG_EVAL_ONLY = 2
STATUS_OK, STATUS_QP_FAILED, STATUS_QP_MAXITER = 0, -1, -2


class GmrHipError(RuntimeError):
    pass


_SIGS = {
    "gmr_last_error": (C.c_char_p, []),
    "gmr_backend_info": (C.c_char_p, []),
    "gmr_device_count": (C.c_int, []),
    "gmr_set_device": (C.c_int, [C.c_int]),
    "gmr_sizeof_model": (C.c_size_t, []),
    "gmr_sizeof_taskset": (C.c_size_t, []),
    "gmr_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gmr_free": (C.c_int, [C.c_void_p]),
    "gmr_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "gmr_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gmr_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "gmr_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gmr_stream_destroy": (C.c_int, [C.c_void_p]),
    "gmr_stream_sync": (C.c_int, [C.c_void_p]),
    "gmr_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "gmr_event_destroy": (C.c_int, [C.c_void_p]),
    "gmr_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gmr_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "gmr_solver_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gmr_solver_destroy": (C.c_int, [C.c_void_p]),
    "gmr_solver_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gmr_solver_set_waves": (C.c_int, [C.c_void_p, C.c_int]),
    "gmr_solver_set_dispatch": (C.c_int, [C.c_void_p, C.c_int]),
    "gmr_retarget_lds_bytes": (C.c_int, [C.c_void_p]),
    "gmr_retarget_streams_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gmr_retarget_streams": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gmr_retarget_group_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "gmr_retarget_group": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "gmr_retarget_group_window_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "gmr_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "gmr_host_free": (C.c_int, [C.c_void_p]),
    "gmr_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "gmr_host_unregister": (C.c_int, [C.c_void_p]),
    "gmr_fk_create": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                C.POINTER(C.c_void_p)]),
    "gmr_fk_destroy": (C.c_int, [C.c_void_p]),
    "gmr_fk_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "gmr_fk_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    "gmr_fk_segment_min_z_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gmr_fk_batch_segments": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "gmr_smplx_create": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "gmr_smplx_destroy": (C.c_int, [C.c_void_p]),
    "gmr_smplx_rows": (C.c_int, [C.c_void_p]),
    "gmr_smplx_joints_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gmr_smplx_joints": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gmr_smplx_align_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "gmr_smplx_align": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    "gmr_smplx_frames": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gmr_smplx_compact_layout": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]),
    "gmr_smplx_align_compact_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "gmr_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "gmr_comm_destroy": (C.c_int, [C.c_void_p]),
    "gmr_comm_rank": (C.c_int, [C.c_void_p]),
    "gmr_comm_world": (C.c_int, [C.c_void_p]),
    "gmr_comm_backend": (C.c_char_p, [C.c_void_p]),
    "gmr_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "gmr_comm_broadcast_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "gmr_comm_barrier": (C.c_int, [C.c_void_p]),
    "gmr_comm_allreduce_max": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "gmr_comm_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "gmr_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "gmr_bootstrap_exchange": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_size_t, C.c_double]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


HIP_RUNTIME = None   # path of the HIP runtime preloaded by _share_hip_runtime(), if any


def _share_hip_runtime():
    """One HIP runtime per process.  A PyTorch-ROCm wheel bundles its own ``libamdhip64.so`` +
    ``libhsa-runtime64.so`` and asks the loader for them by their unversioned names, so when libgmrhip.so has
    already pulled in the system runtime (``/opt/rocm``, by SONAME ``libamdhip64.so.7``) a later ``import
    torch`` loads a SECOND runtime and finds "No HIP GPUs" (the first one owns the device).  The other order
    works (our SONAME request matches torch's copy).  So: if torch is installed but not yet imported, load ITS
    runtime first -- without importing torch -- and let libgmrhip.so bind to it.
    ``GMR_HIP_RUNTIME=system`` keeps the system runtime, ``GMR_HIP_RUNTIME=/path/libamdhip64.so`` picks one."""
    global HIP_RUNTIME
    mode = os.environ.get("GMR_HIP_RUNTIME", "auto")
    if mode == "system" or "torch" in sys.modules:
        return
    path = mode if mode not in ("auto", "torch") else None
    if path is None:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        for d in (spec.submodule_search_locations or []) if spec else []:
            cand = os.path.join(d, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                path = cand
                break
    if path:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        HIP_RUNTIME = path


def lib():
    """Load libgmrhip.so (raises GmrHipError when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GmrHipError(
                f"{LIB_PATH} not found: build it with `python -m general_motion_retargeting_amd.build` "
                "(there is no CPU fallback in the product path)")
        _share_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.gmr_sizeof_model() != MODEL_DTYPE.itemsize or L.gmr_sizeof_taskset() != TASKSET_DTYPE.itemsize:
            raise GmrHipError("ABI mismatch between libgmrhip.so and the Python struct layouts")
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise GmrHipError(f"libgmrhip error {rc}: {lib().gmr_last_error().decode()}")


def require_gpu() -> None:
    if lib().gmr_device_count() <= 0:
        raise GmrHipError("no HIP device visible: the product path needs an MI355X (no CPU fallback)")


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _s(stream):
    """hipStream_t of a :class:`Stream`, a raw ``c_void_p`` or ``None`` (the default stream)."""
    return stream.ptr if isinstance(stream, Stream) else stream


def _d(x):
    """device pointer of a :class:`DeviceBuffer`, a raw ``c_void_p`` or ``None``."""
    return x.ptr if isinstance(x, DeviceBuffer) else x


class DeviceBuffer:
    """Owned device allocation (hipMalloc through the C-ABI)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().gmr_malloc(C.byref(p), self.nbytes))
        self.ptr = p

    @classmethod
    def from_host(cls, a: np.ndarray, stream=None) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        s = _s(stream)
        check(lib().gmr_memcpy_h2d(b.ptr, _ptr(a), a.nbytes, s))
        check(lib().gmr_stream_sync(s))
        return b

    def to_host(self, shape, dtype, stream=None) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        s = _s(stream)
        check(lib().gmr_memcpy_d2h(_ptr(out), self.ptr, out.nbytes, s))
        check(lib().gmr_stream_sync(s))
        return out

    def zero(self, stream=None):
        check(lib().gmr_memset(self.ptr, 0, self.nbytes, _s(stream)))
        check(lib().gmr_stream_sync(_s(stream)))

    def free(self):
        if self.ptr:
            lib().gmr_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _PinnedOwner:
    """Keeps one gmr_host_alloc block alive for the NumPy arrays that view it."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib().gmr_host_alloc(C.byref(p), max(int(nbytes), 8)))
        self.ptr, self.nbytes = p, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().gmr_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64) -> np.ndarray:
    """``np.empty`` in page-locked host memory (gmr_host_alloc): H2D / D2H copies of such arrays are asynchronous and run
    at PCIe speed, which is what lets :func:`retarget_group` overlap them with the kernels.  The block is freed when the
    last view of the array dies."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    owner = _PinnedOwner(n * dtype.itemsize)
    buf = (C.c_char * max(owner.nbytes, 1)).from_address(owner.ptr.value)
    buf._gmr_owner = owner                               # the ctypes buffer is the ndarray's base: it carries the owner
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def pinned_copy(a: np.ndarray) -> np.ndarray:
    out = pinned_empty(a.shape, a.dtype)
    np.copyto(out, a)
    return out


class Job(C.Structure):
    """``gmr_job_t``: one (solver, batch) of a group launch."""
    _fields_ = [("solver", C.c_void_p), ("S", C.c_int32), ("T", C.c_int32), ("q0", C.c_void_p), ("human", C.c_void_p),
                ("len", C.c_void_p), ("q_out", C.c_void_p), ("nsolve", C.c_void_p), ("status", C.c_void_p),
                ("tgt_out", C.c_void_p), ("err_out", C.c_void_p)]


def _addr(x):
    """address of a host ndarray / DeviceBuffer / raw pointer / None"""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, DeviceBuffer):
        return x.ptr.value
    return x.value if isinstance(x, C.c_void_p) else int(x)


def group_outputs(jobs, pinned: bool = True):
    """Output arrays ``(q_out, nsolve, status)`` per job for :func:`retarget_group` (``outs=``): page-locking memory costs
    far more than retargeting a batch does, so a caller that runs many batches allocates them once."""
    empty = pinned_empty if pinned else (lambda shape, dtype: np.empty(shape, dtype))
    outs = []
    for j in jobs:
        sol, (S, T) = j["solver"], j["human"].shape[:2]
        outs.append((empty((S, T, sol.nq), np.float64), empty((S, T, 2), np.int32), empty((S,), np.int32)))
    return outs


def retarget_group(jobs, flags: int = 0, slices: int = 0, out_pinned: bool = False, outs=None):
    """Several (solver, batch) jobs as ONE scheduling domain through host buffers (``gmr_retarget_group``): the mixed-robot
    batch of BASELINE.json configs[3].  ``jobs`` = list of dicts ``{"solver": Solver, "human": f64[S,T,nhuman,7], optional
    "q0": f64[S,nq] (default the solver's qpos0), "lens": i32[S]}``.  Returns one ``(q_out, nsolve, status)`` per job.
    Inputs in pinned memory (:func:`pinned_empty`) and pinned outputs (``outs=`` from :func:`group_outputs`, reused across
    calls; or ``out_pinned=True``: allocated per call, which is slow) make the copies asynchronous, so that they overlap the
    kernels slice by slice."""
    arr = (Job * max(len(jobs), 1))()
    keep, given, outs = [], outs, []
    empty = pinned_empty if out_pinned else (lambda shape, dtype: np.empty(shape, dtype))
    for i, j in enumerate(jobs):
        sol = j["solver"]
        human = j["human"]
        if not (isinstance(human, np.ndarray) and human.dtype == np.float64 and human.flags.c_contiguous):
            human = np.ascontiguousarray(human, dtype=np.float64)
        if human.ndim != 4 or human.shape[2] != sol.nhuman or human.shape[3] != 7:
            raise ValueError(f"job {i}: human must be [S,T,{sol.nhuman},7], got {human.shape}")
        S, T = human.shape[:2]
        q0 = j.get("q0")
        if q0 is None:
            q0 = np.broadcast_to(sol.model_blob["qpos0"][0][: sol.nq], (S, sol.nq))
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        if q0.shape != (S, sol.nq):
            raise ValueError(f"job {i}: q0 must be [{S},{sol.nq}]")
        lens = j.get("lens")
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.int32)
            if lens.shape != (S,):
                raise ValueError(f"job {i}: lens must be [S]")
        if given is not None:
            q_out, nsolve, status = given[i]
            if q_out.shape != (S, T, sol.nq) or nsolve.shape != (S, T, 2) or status.shape != (S,) or q_out.dtype != np.float64 \
                    or nsolve.dtype != np.int32 or status.dtype != np.int32 or not (q_out.flags.c_contiguous and nsolve.flags.c_contiguous):
                raise ValueError(f"job {i}: outs do not match the batch")
        else:
            q_out = empty((S, T, sol.nq), np.float64)
            nsolve = empty((S, T, 2), np.int32)
            status = np.zeros(S, dtype=np.int32)
        keep.append((human, q0, lens))
        outs.append((q_out, nsolve, status))
        arr[i] = Job(sol.handle.value, S, T, _addr(q0), _addr(human), _addr(lens), _addr(q_out), _addr(nsolve), _addr(status),
                     None, None)
    check(lib().gmr_retarget_group(C.cast(arr, C.c_void_p), len(jobs), int(flags), int(slices)))
    return outs


def retarget_group_dev(jobs, flags: int = 0, stream=None, window=None):
    """``gmr_retarget_group_dev``: ``jobs`` = list of ``(solver, S, T, d_q0, d_human, d_len, d_q_out, d_nsolve, d_status)``
    with device pointers (DeviceBuffer or raw); asynchronous on ``stream``.  ``window=(t_begin, t_end)``: only those frames of
    every stream (``gmr_retarget_group_window_dev``; consecutive windows on one stream, starting at 0)."""
    arr = (Job * max(len(jobs), 1))()
    for i, (sol, S, T, d_q0, d_h, d_len, d_qo, d_ns, d_st) in enumerate(jobs):
        arr[i] = Job(sol.handle.value, int(S), int(T), _addr(d_q0), _addr(d_h), _addr(d_len), _addr(d_qo), _addr(d_ns),
                     _addr(d_st), None, None)
    if window is None:
        check(lib().gmr_retarget_group_dev(C.cast(arr, C.c_void_p), len(jobs), int(flags), _s(stream)))
    else:
        check(lib().gmr_retarget_group_window_dev(C.cast(arr, C.c_void_p), len(jobs), int(flags), int(window[0]), int(window[1]),
                                                  _s(stream)))


class Stream:
    """Non-blocking HIP stream (gmr_stream_create)."""

    def __init__(self):
        p = C.c_void_p()
        check(lib().gmr_stream_create(C.byref(p)))
        self.ptr = p

    def sync(self):
        check(lib().gmr_stream_sync(self.ptr))

    def __del__(self):
        try:
            if self.ptr:
                lib().gmr_stream_destroy(self.ptr)
        except Exception:
            pass


class Event:
    def __init__(self):
        p = C.c_void_p()
        check(lib().gmr_event_create(C.byref(p)))
        self.ptr = p

    def record(self, stream=None):
        check(lib().gmr_event_record(self.ptr, _s(stream)))

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        check(lib().gmr_event_elapsed_ms(self.ptr, stop.ptr, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self.ptr:
                lib().gmr_event_destroy(self.ptr)
        except Exception:
            pass


class Solver:
    """Device-resident (robot model, task set): handle behind gmr_solver_create."""

    def __init__(self, model_blob: np.ndarray, taskset_blob: np.ndarray):
        require_gpu()
        assert model_blob.dtype == MODEL_DTYPE and taskset_blob.dtype == TASKSET_DTYPE
        self.model_blob = np.ascontiguousarray(model_blob)
        self.taskset_blob = np.ascontiguousarray(taskset_blob)
        h = C.c_void_p()
        check(lib().gmr_solver_create(_ptr(self.model_blob), _ptr(self.taskset_blob), C.byref(h)))
        self.handle = h
        self.nq = int(model_blob["nq"][0])
        self.nv = int(model_blob["nv"][0])
        self.nhuman = int(taskset_blob["nhuman"][0])

    def set_waves(self, waves_per_stream: int) -> None:
        """0 = automatic, 1 = one wavefront per stream, 4 = main + 3 helper wavefronts per stream."""
        check(lib().gmr_solver_set_waves(self.handle, int(waves_per_stream)))

    def set_dispatch(self, frames_per_item: int) -> None:
        """Many-stream launches: > 0 = device-side FIFO of (stream, frames_per_item frames) items, 0 = one workgroup per stream."""
        check(lib().gmr_solver_set_dispatch(self.handle, int(frames_per_item)))

    @property
    def lds_bytes(self) -> int:
        return int(lib().gmr_retarget_lds_bytes(self.handle))

    def retarget_streams(self, q0, human, lens=None, flags: int = 0, want_targets: bool = False,
                         want_errors: bool = False):
        """Host arrays in/out: q0[S,nq], human[S,T,nhuman,7] -> q_out[S,T,nq], nsolve[S,T,2], status[S]
        (+ targets[S,T,nhuman,7] = the kernel's preprocessed frames and/or errors[S,T,2] = error1/error2 at the
        configuration every frame ends with, when asked for)."""
        human = np.ascontiguousarray(human, dtype=np.float64)
        if human.ndim != 4 or human.shape[2] != self.nhuman or human.shape[3] != 7:
            raise ValueError(f"human must be [S,T,{self.nhuman},7], got {human.shape}")
        S, T = human.shape[:2]
        q0 = np.ascontiguousarray(q0, dtype=np.float64)
        if q0.shape != (S, self.nq):
            raise ValueError(f"q0 must be [{S},{self.nq}], got {q0.shape}")
        if lens is not None:
            lens = np.ascontiguousarray(lens, dtype=np.int32)
            if lens.shape != (S,):
                raise ValueError("lens must be [S]")
        q_out = np.zeros((S, T, self.nq), dtype=np.float64)
        nsolve = np.zeros((S, T, 2), dtype=np.int32)
        status = np.zeros(S, dtype=np.int32)
        targets = np.zeros((S, T, self.nhuman, 7), dtype=np.float64) if want_targets else None
        errors = np.zeros((S, T, 2), dtype=np.float64) if want_errors else None
        check(lib().gmr_retarget_streams(self.handle, S, T, _ptr(q0), _ptr(human), _ptr(lens), int(flags),
                                         _ptr(q_out), _ptr(nsolve), _ptr(status), _ptr(targets), _ptr(errors)))
        if want_targets or want_errors:
            return q_out, nsolve, status, targets, errors
        return q_out, nsolve, status

    def retarget_streams_dev(self, S, T, d_q0, d_human, d_len, flags, d_q_out, d_nsolve, d_status, stream=None,
                             d_tgt_out=None, d_err_out=None):
        """Device pointers (DeviceBuffer or raw c_void_p); asynchronous on `stream`."""
        check(lib().gmr_retarget_streams_dev(self.handle, int(S), int(T), _d(d_q0), _d(d_human), _d(d_len), int(flags),
                                             _d(d_q_out), _d(d_nsolve), _d(d_status), _d(d_tgt_out), _d(d_err_out),
                                             _s(stream)))

    def close(self):
        if self.handle:
            lib().gmr_solver_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FkHandle:
    """Device-resident KinematicsModel tree: handle behind gmr_fk_create."""

    def __init__(self, tree: dict):
        require_gpu()
        self.nbody = int(len(tree["parent"]))
        self.ndof = int((np.asarray(tree["dof_dim"]) > 0).sum())
        self._keep = [np.ascontiguousarray(tree["parent"], dtype=np.int32),
                      np.ascontiguousarray(tree["local_translation"], dtype=np.float32),
                      np.ascontiguousarray(tree["local_rotation"], dtype=np.float32),
                      np.ascontiguousarray(tree["dof_idx"], dtype=np.int32),
                      np.ascontiguousarray(tree["axis"], dtype=np.float64)]
        h = C.c_void_p()
        check(lib().gmr_fk_create(self.nbody, *[_ptr(a) for a in self._keep], self.ndof, C.byref(h)))
        self.handle = h

    def fk(self, root_pos, root_rot, dof, want_rot=True, want_min_z=False):
        root_pos = np.ascontiguousarray(root_pos, dtype=np.float32)
        root_rot = np.ascontiguousarray(root_rot, dtype=np.float32)
        dof = np.ascontiguousarray(dof, dtype=np.float32)
        B = root_pos.shape[0]
        if root_pos.shape != (B, 3) or root_rot.shape != (B, 4) or dof.shape != (B, self.ndof):
            raise ValueError("shape mismatch in fk inputs")
        bp = np.zeros((B, self.nbody, 3), dtype=np.float32)
        br = np.zeros((B, self.nbody, 4), dtype=np.float32) if want_rot else None
        mz = np.zeros(1, dtype=np.float32) if want_min_z else None
        check(lib().gmr_fk_batch(self.handle, B, _ptr(root_pos), _ptr(root_rot), _ptr(dof), _ptr(bp), _ptr(br), _ptr(mz)))
        return bp, br, (float(mz[0]) if want_min_z else None)

    def fk_segments(self, root_pos, root_rot, dof, seg_start, want_pos=True):
        """Many clips in one launch: rows [seg_start[g], seg_start[g + 1]) are clip g.  Returns (body_pos [B, nb, 3] or None,
        seg_min_z [nseg]): the per-clip minimum of body_pos z (gmr_fk_batch_segments)."""
        root_pos = np.ascontiguousarray(root_pos, dtype=np.float32)
        root_rot = np.ascontiguousarray(root_rot, dtype=np.float32)
        dof = np.ascontiguousarray(dof, dtype=np.float32)
        seg_start = np.ascontiguousarray(seg_start, dtype=np.int32)
        B, nseg = root_pos.shape[0], len(seg_start) - 1
        if root_pos.shape != (B, 3) or root_rot.shape != (B, 4) or dof.shape != (B, self.ndof) or nseg < 0:
            raise ValueError("shape mismatch in fk inputs")
        bp = np.zeros((B, self.nbody, 3), dtype=np.float32) if want_pos else None
        mz = np.zeros(max(nseg, 0), dtype=np.float32)
        check(lib().gmr_fk_batch_segments(self.handle, B, _ptr(root_pos), _ptr(root_rot), _ptr(dof), nseg, _ptr(seg_start),
                                          _ptr(bp), _ptr(mz)))
        return bp, mz

    def fk_dev(self, B, d_root_pos, d_root_rot, d_dof, d_body_pos, d_body_rot=None, d_min_z=None, stream=None):
        check(lib().gmr_fk_batch_dev(self.handle, int(B), _d(d_root_pos), _d(d_root_rot), _d(d_dof), _d(d_body_pos),
                                     _d(d_body_rot), _d(d_min_z), _s(stream)))

    def close(self):
        if self.handle:
            lib().gmr_fk_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SmplxHandle:
    """Kinematic tree + joint selection behind gmr_smplx_create (N1: SMPL-X frame extraction)."""

    def __init__(self, parents, sel=None):
        require_gpu()
        self.parents = np.ascontiguousarray(parents, dtype=np.int32)
        self.J = int(len(self.parents))
        self.sel = None if sel is None else np.ascontiguousarray(sel, dtype=np.int32)
        h = C.c_void_p()
        check(lib().gmr_smplx_create(self.J, _ptr(self.parents), 0 if self.sel is None else len(self.sel),
                                     _ptr(self.sel), C.byref(h)))
        self.handle = h
        self.rows = int(lib().gmr_smplx_rows(h))

    def joints(self, j_rest, full_pose, transl):
        j_rest = np.ascontiguousarray(j_rest, dtype=np.float64)
        full_pose = np.ascontiguousarray(full_pose, dtype=np.float32).reshape(-1, self.J, 3)
        transl = np.ascontiguousarray(transl, dtype=np.float32).reshape(-1, 3)
        N = full_pose.shape[0]
        if j_rest.shape != (self.J, 3) or transl.shape[0] != N:
            raise ValueError("shape mismatch in smplx joints inputs")
        out = np.zeros((N, self.J, 3), dtype=np.float32)
        check(lib().gmr_smplx_joints(self.handle, N, _ptr(j_rest), _ptr(full_pose), _ptr(transl), _ptr(out)))
        return out

    def align(self, full_pose, joints, target_time=None):
        full_pose = np.ascontiguousarray(full_pose, dtype=np.float32).reshape(-1, self.J, 3)
        joints = np.ascontiguousarray(joints, dtype=np.float32)
        N = full_pose.shape[0]
        if joints.ndim != 3 or joints.shape[0] != N or joints.shape[1] < self.J or joints.shape[2] != 3:
            raise ValueError("joints must be [N, >=J, 3]")
        tt = None if target_time is None else np.ascontiguousarray(target_time, dtype=np.float64)
        nout = N if tt is None else int(len(tt))
        out = np.zeros((nout, self.rows, 7), dtype=np.float64)
        check(lib().gmr_smplx_align(self.handle, N, int(joints.shape[1]), _ptr(full_pose), _ptr(joints), nout, _ptr(tt),
                                    _ptr(out)))
        return out

    def frames(self, j_rest, full_pose, transl, target_time=None):
        """Joints-only body model + alignment of the selected rows in one call (``gmr_smplx_frames``): the packed frames
        ``f64[Nout, rows, 7]`` only; bit-identical to :meth:`joints` followed by :meth:`align`."""
        full_pose = np.ascontiguousarray(full_pose, dtype=np.float32).reshape(-1, self.J, 3)
        N = full_pose.shape[0]
        j_rest = np.ascontiguousarray(j_rest, dtype=np.float64)
        transl = np.ascontiguousarray(transl, dtype=np.float32).reshape(N, 3)
        if j_rest.shape != (self.J, 3):
            raise ValueError(f"j_rest must be [{self.J}, 3]")
        tt = None if target_time is None else np.ascontiguousarray(target_time, dtype=np.float64)
        nout = N if tt is None else len(tt)
        out = np.empty((nout, self.rows, 7), dtype=np.float64)
        check(lib().gmr_smplx_frames(self.handle, N, _ptr(j_rest), _ptr(full_pose), _ptr(transl), nout, _ptr(tt), _ptr(out)))
        return out

    def compact_layout(self):
        """(pose_joints, row_joints): the joints whose poses / positions the alignment reads, in the order of the compact
        inputs of :meth:`align_compact_dev`."""
        pj = np.zeros(self.J, dtype=np.int32)
        rj = np.zeros(self.J, dtype=np.int32)
        npose, nrow = C.c_int(), C.c_int()
        check(lib().gmr_smplx_compact_layout(self.handle, _ptr(pj), C.byref(npose), _ptr(rj), C.byref(nrow)))
        return pj[: npose.value].copy(), rj[: nrow.value].copy()

    def align_compact_dev(self, N, d_pose_c, d_joints_c, nout, d_target_time, d_out, stream=None):
        check(lib().gmr_smplx_align_compact_dev(self.handle, int(N), _d(d_pose_c), _d(d_joints_c), int(nout), _d(d_target_time),
                                                _d(d_out), _s(stream)))

    def align_dev(self, N, jstride, d_full_pose, d_joints, nout, d_target_time, d_out, stream=None):
        check(lib().gmr_smplx_align_dev(self.handle, int(N), int(jstride), _d(d_full_pose), _d(d_joints), int(nout),
                                        _d(d_target_time), _d(d_out), _s(stream)))

    def joints_dev(self, N, d_j_rest, d_full_pose, d_transl, d_joints, stream=None):
        check(lib().gmr_smplx_joints_dev(self.handle, int(N), _d(d_j_rest), _d(d_full_pose), _d(d_transl), _d(d_joints),
                                         _s(stream)))

    def close(self):
        if self.handle:
            lib().gmr_smplx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
