"""Dataset harness: what the reference's ``process_file`` does after loading a clip
(``scripts/smplx_to_robot_dataset.py:78-146`` and ``scripts/bvh_to_robot_dataset.py:86-152``;
SURVEY.md H10), for MANY clips per launch.

Per clip: retarget every frame from ``qpos0`` (a fresh ``GeneralMotionRetargeting`` per file,
:79-87) -> split qpos, root quaternion wxyz -> xyzw (:97-102) -> ``local_body_pos`` = FK with
identity root (:106-112) -> optional height adjust ``root_z -= min_{t,b} z`` over the world FK
(:118-126) -> optional ``root_xy -= root_xy[0]`` (:128-131) -> the pkl dict (:134-141).
The SMPL-X script has both adjustments on, the BVH script both off (``bvh_to_robot_dataset.py:128``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .data_loader import motion_dict
from .kinematics_model import KinematicsModel
from .motion_retarget import GeneralMotionRetargeting


def postprocess_clip(qpos: np.ndarray, km: KinematicsModel, fps: float, height_adjust: bool = True,
                     root_origin_offset: bool = True, ground_offset: float = 0.0) -> Dict:
    """qpos f64[T, nq] of one clip -> motion dict (App. D)."""
    qpos = np.array(qpos, dtype=np.float64, copy=True)
    root_pos = qpos[:, :3].copy()
    root_rot = qpos[:, [4, 5, 6, 3]].copy()          # wxyz -> xyzw
    dof_pos = qpos[:, 7:].copy()
    T = qpos.shape[0]
    ident_pos = np.zeros((T, 3), dtype=np.float32)
    ident_rot = np.zeros((T, 4), dtype=np.float32)
    ident_rot[:, 3] = 1.0
    local_body_pos, _ = km.forward_kinematics(ident_pos, ident_rot, dof_pos.astype(np.float32))
    if height_adjust and T > 0:
        _, _, lowest = km.forward_kinematics(root_pos.astype(np.float32), root_rot.astype(np.float32),
                                             dof_pos.astype(np.float32), return_min_z=True)
        root_pos[:, 2] = root_pos[:, 2] - lowest + ground_offset
    if root_origin_offset and T > 0:
        root_pos[:, :2] -= root_pos[0, :2]
    return motion_dict(fps, root_pos, root_rot, dof_pos, np.asarray(local_body_pos), km.body_names)


def postprocess_clips(qpos_list: Sequence[np.ndarray], km: KinematicsModel, fps: Sequence[float], height_adjust: bool = True,
                      root_origin_offset: bool = True, ground_offset: float = 0.0) -> List[Dict]:
    """:func:`postprocess_clip` for MANY clips with two FK launches in total instead of one or two per clip: the frames
    of all clips are concatenated, ``local_body_pos`` comes from one launch with identity roots, the per-clip minimum of
    the world FK's z from one launch + a segmented reduction on the device (``gmr_fk_batch_segments``).  Same values."""
    lens = np.array([len(q) for q in qpos_list], dtype=np.int64)
    if len(lens) == 0:
        return []
    seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    B = int(seg[-1])
    allq = np.concatenate([np.asarray(q, dtype=np.float64).reshape(-1, km.num_dof + 7) for q in qpos_list]) if B else np.zeros((0, km.num_dof + 7))
    root_pos = allq[:, :3].copy()
    root_rot = allq[:, [4, 5, 6, 3]].copy()           # wxyz -> xyzw
    dof_pos = allq[:, 7:].copy()
    dof32 = dof_pos.astype(np.float32)
    ident = np.zeros((B, 4), dtype=np.float32)
    ident[:, 3] = 1.0
    fk = km.hip_handle
    local_body_pos = fk.fk_segments(np.zeros((B, 3), dtype=np.float32), ident, dof32, np.array([0, B]))[0] if B else \
        np.zeros((0, km.num_joint, 3), np.float32)
    if height_adjust and B:
        _, lowest = fk.fk_segments(root_pos.astype(np.float32), root_rot.astype(np.float32), dof32, seg, want_pos=False)
        root_pos[:, 2] = root_pos[:, 2] - np.repeat(lowest.astype(np.float64), lens) + ground_offset
    out = []
    for i in range(len(lens)):
        a, b = int(seg[i]), int(seg[i + 1])
        rp = root_pos[a:b].copy()
        if root_origin_offset and b > a:
            rp[:, :2] -= rp[0, :2]
        out.append(motion_dict(fps[i], rp, root_rot[a:b].copy(), dof_pos[a:b].copy(), local_body_pos[a:b].copy(), km.body_names))
    return out


class ClipRetargeter:
    """Many clips of one (source, robot, height) per call, ONE IK launch each: the per-batch step of the dataset drivers.
    What does not depend on the batch is built once -- the solver, the ``KinematicsModel`` -- and the padded input /
    output arrays live in page-locked memory that is reused from batch to batch (locking pages costs more than
    retargeting them), so the copies of a large batch run asynchronously, slice by slice, under its kernels."""

    def __init__(self, src_human: str, tgt_robot: str, actual_human_height: Optional[float] = None, height_adjust: bool = True,
                 root_origin_offset: bool = True, offset_to_ground: bool = False):
        self.gmr = GeneralMotionRetargeting(src_human, tgt_robot, actual_human_height=actual_human_height)
        self.height_adjust, self.root_origin_offset, self.offset_to_ground = height_adjust, root_origin_offset, offset_to_ground
        self._km: Optional[KinematicsModel] = None
        self._pin: Dict[str, np.ndarray] = {}
        self.timing: Dict[str, float] = {}

    def _buf(self, name: str, shape, dtype) -> np.ndarray:
        """a [shape] view of this object's page-locked block ``name`` (grown by half when too small)"""
        from . import _lib
        need = int(np.prod(shape)) * np.dtype(dtype).itemsize
        blk = self._pin.get(name)
        if blk is None or blk.nbytes < need:
            self._pin.pop(name, None)
            blk = self._pin[name] = _lib.pinned_empty((need + need // 2,), np.uint8)
        return blk[:need].view(dtype).reshape(shape)

    # ---- staged protocol (DatasetPipeline): clips are copied into the padded, page-locked batch AS THEY ARRIVE from the
    # loaders -- the time this thread would otherwise spend waiting for them -- instead of when the batch is launched
    def reserve(self, frames_budget: int) -> None:
        """Lock the pages of a batch of ``frames_budget`` padded frames now (e.g. while the loaders are starting up)."""
        sol = self.gmr.hip_solver
        self._buf("human", (frames_budget, sol.nhuman, 7), np.float64)
        self._buf("q_out", (frames_budget, sol.nq), np.float64)
        self._buf("nsolve", (frames_budget, 2), np.int32)
        if self._km is None:
            self._km = KinematicsModel(self.gmr.xml_file)
            self._km.hip_handle

    def begin(self, longest: int, max_clips: int, frames_budget: int) -> None:
        """A batch whose longest clip has ``longest`` frames (the drivers hand the clips over largest first)."""
        sol = self.gmr.hip_solver
        self._T = max(int(longest), 1)
        self._cap = max(1, min(int(max_clips), int(frames_budget) // self._T))
        self._human = self._buf("human", (self._cap, self._T, sol.nhuman, 7), np.float64)
        self._lens = np.zeros(self._cap, dtype=np.int32)
        self._n = 0

    def add(self, clip) -> bool:
        """Copy one clip into the batch; False when it does not fit (batch full, or the clip is longer than the batch's
        rows): the caller finishes this batch and begins another."""
        import time
        t0 = time.perf_counter()
        p = clip if isinstance(clip, np.ndarray) else self.gmr.pack_frames(clip)
        if self._n >= self._cap or p.shape[0] > self._T:
            return False
        self._human[self._n, : p.shape[0]] = p          # (rows beyond a clip's length are never read by the kernel)
        self._lens[self._n] = p.shape[0]
        self._n += 1
        self.timing["pack"] = self.timing.get("pack", 0.0) + time.perf_counter() - t0
        return True

    def finish(self, fps: Sequence[float]) -> List[Dict]:
        import time
        from . import _lib
        S, T, gmr = self._n, self._T, self.gmr
        if S == 0:
            return []
        t1 = time.perf_counter()
        sol = gmr.hip_solver
        lens = self._lens[:S].copy()
        q0 = self._buf("q0", (S, sol.nq), np.float64)
        q0[:] = gmr.model.qpos0
        outs = [(self._buf("q_out", (S, T, sol.nq), np.float64), self._buf("nsolve", (S, T, 2), np.int32), np.zeros(S, np.int32))]
        (qpos, _, status), = _lib.retarget_group([{"solver": sol, "human": self._human[:S], "q0": q0, "lens": lens}],
                                                 gmr._flags(self.offset_to_ground), 0, outs=outs)
        t2 = time.perf_counter()
        self._n = 0
        if (status != 0).any():
            raise RuntimeError(f"IK failed for clips {np.nonzero(status)[0].tolist()}")
        if self._km is None:
            self._km = KinematicsModel(gmr.xml_file)
        out = postprocess_clips([qpos[i, : lens[i]] for i in range(S)], self._km, fps, self.height_adjust, self.root_origin_offset)
        t3 = time.perf_counter()
        for k, v in (("ik", t2 - t1), ("post", t3 - t2)):
            self.timing[k] = self.timing.get(k, 0.0) + v
        return out

    def __call__(self, clips: Sequence, fps: Sequence[float]) -> List[Dict]:
        packed = [c if isinstance(c, np.ndarray) else self.gmr.pack_frames(c) for c in clips]
        if not packed:
            return []
        self.begin(max(p.shape[0] for p in packed), len(packed), 1 << 62)
        for p in packed:
            assert self.add(p)
        return self.finish(fps)


def retarget_clips(src_human: str, tgt_robot: str, clips: Sequence, fps: Sequence[float],
                   actual_human_height: Optional[float] = None, height_adjust: bool = True,
                   root_origin_offset: bool = True, offset_to_ground: bool = False) -> List[Dict]:
    """Retarget many clips of one (source, robot, height) in ONE IK launch.

    ``clips[i]`` is a list of ``human_data`` dicts or an array ``[T_i, nhuman, 7]`` (ragged lengths
    are fine: streams are padded and the kernel stops each stream at its own length).
    Returns one motion dict per clip, identical to processing the clips one by one.
    """
    return ClipRetargeter(src_human, tgt_robot, actual_human_height, height_adjust, root_origin_offset, offset_to_ground)(clips, fps)


def retarget_mixed(groups: Sequence[Dict], offset_to_ground: bool = False, slices: int = 0, pinned_outputs: bool = False):
    """Mixed-robot batch (BASELINE.json configs[3]; SURVEY.md section 8d "one kernel with per-stream model index"): every
    group = one (source, robot, height) with its own streams; ALL groups form one scheduling domain on the device --
    the robots of the throughput kernel's size class (every shipped one) share a resident grid and one queue of
    (robot, stream, chunk) items (``gmr_retarget_group``) -- and the host buffers are cut into slices whose copies
    overlap the kernels.  Bit-identical to retargeting every group by itself.

    ``groups[i]`` = ``{"src_human", "tgt_robot", "human": f64[S,T,nhuman,7], optional "lens",
    "actual_human_height"}``; arrays allocated with ``_lib.pinned_empty`` are copied asynchronously.
    Returns a list of ``(qpos[S,T,nq], nsolve[S,T,2], status[S])``.
    """
    from . import _lib
    jobs, keep = [], []
    for g in groups:
        gmr = GeneralMotionRetargeting(g["src_human"], g["tgt_robot"], actual_human_height=g.get("actual_human_height"))
        keep.append(gmr)
        jobs.append({"solver": gmr.hip_solver, "human": g["human"], "lens": g.get("lens")})
    flags = _lib.FLAG_OFFSET_TO_GROUND if offset_to_ground else 0
    return _lib.retarget_group(jobs, flags, slices, out_pinned=pinned_outputs)   # (rows beyond a stream's length: zeros)


def retarget_bvh_files(bvh_files: Sequence[str], tgt_robot: str, fps: float = 30.0, height_adjust: bool = False,
                       root_origin_offset: bool = False) -> List[Dict]:
    """``scripts/bvh_to_robot_dataset.py:60-152`` for a list of files: every BVH clip becomes one
    stream of ONE launch (LAFAN1: 77 ragged clips).  Both adjustments default to off like that script
    (``HEIGHT_ADJUST = False``, :128); the loader's hard-coded height 1.75 is used (lafan1.py:39)."""
    from .utils.lafan1 import load_lafan1_packed
    gmr = GeneralMotionRetargeting("bvh", tgt_robot, actual_human_height=1.75)
    clips = [load_lafan1_packed(f, gmr.human_body_names)[0] for f in bvh_files]
    return retarget_clips("bvh", tgt_robot, clips, [fps] * len(clips), actual_human_height=1.75,
                          height_adjust=height_adjust, root_origin_offset=root_origin_offset)


def retarget_smplx_files(smplx_files: Sequence[str], smplx_body_model_path: str, tgt_robot: str, tgt_fps: int = 30,
                         height_adjust: bool = True, root_origin_offset: bool = True,
                         skip_errors: bool = True) -> List[Optional[Dict]]:
    """``scripts/smplx_to_robot_dataset.py:39-146`` (``process_file``) for a list of AMASS-style SMPL-X
    files, everything after the file read on the device: joints-only body model + fps alignment
    (utils/smpl.py) -> packed frames -> IK (one group launch for all heights) -> FK post-processing.
    Returns one motion dict per file in input order (``None`` for a file that failed to load when
    ``skip_errors`` -- the script prints and skips, :62-76)."""
    raws, ok = [], []
    for i, f in enumerate(smplx_files):
        try:
            raws.append(_load_smplx_raw(f))
            ok.append(i)
        except Exception as e:  # noqa: BLE001 - mirrors the script's print-and-skip
            if not skip_errors:
                raise
            print(f"Error loading {f}: {e}")
    out: List[Optional[Dict]] = [None] * len(smplx_files)
    for i, md in zip(ok, retarget_smplx_loaded(raws, smplx_body_model_path, tgt_robot, tgt_fps, height_adjust, root_origin_offset)):
        out[i] = md
    return out


def retarget_single_clip(retargeter: GeneralMotionRetargeting, frames: Sequence, fps: float,
                         skip_first_frame: bool = True) -> Dict:
    """The save path of the single-clip scripts (``scripts/smplx_to_robot.py:104-161``,
    ``scripts/bvh_to_robot.py`` alike): one ``retarget()`` per displayed frame, then a pkl dict with
    ``root_rot`` xyzw, ``local_body_pos = None`` and ``link_body_list = None`` (:146-158).  Their loop
    increments the index BEFORE the first use (:104-111), so frame 0 is never retargeted; that is the default
    here too.  One launch instead of one call per frame; continues from ``retargeter``'s configuration."""
    seq = frames[1:] if skip_first_frame else frames
    qpos = retargeter.retarget_clip(seq)
    return motion_dict(fps, qpos[:, :3].copy(), qpos[:, [4, 5, 6, 3]].copy(), qpos[:, 7:].copy(), None, None)


# ----------------------------------------------------------------------------------------------
# the dataset drivers with the reference's FILE semantics
#   python -m general_motion_retargeting_amd.dataset --source smplx|bvh --src_folder ... --tgt_folder ... --robot ...
# (scripts/smplx_to_robot_dataset.py:171-242, scripts/bvh_to_robot_dataset.py:16-157): same folder walk, same
# skip-if-exists / --override rule, same exclusion lists, one pkl per input -- but the files of a batch share ONE IK
# launch (and one FK launch per clip) instead of one process per file.
# ----------------------------------------------------------------------------------------------
EXCLUDE_FILE_CONTENT = ["BMLrub", "EKUT", "crawl", "_lie", "upstairs", "downstairs"]    # smplx_to_robot_dataset.py:218


def natural_key(name: str):
    """Sort key of ``natsort.natsorted`` (default algorithm: runs of digits compare as unsigned integers)."""
    import re
    parts = re.split(r"(\d+)", name)
    return [int(p) if p.isdigit() else p for p in parts if p != ""]


def _natsorted(names):
    try:                                   # the reference uses natsort; not installed here -> same order from natural_key
        from natsort import natsorted
        return natsorted(names)
    except ImportError:
        def key(n):
            return [(0, p, "") if isinstance(p, int) else (1, 0, p) for p in natural_key(n)]
        return sorted(names, key=key)


def load_hard_motions(paths) -> List[str]:
    """Names listed in the ``assets/hard_motions/*.txt`` reports (``Motion: <name>.pkl, Difficulty: ...``;
    smplx_to_robot_dataset.py:196-205)."""
    import os
    out = []
    for p in paths:
        if not os.path.exists(p):
            continue
        with open(p, "r") as f:
            for line in f:
                if "Motion:" not in line:
                    continue
                motion_path = line.split(":")[1].strip()
                out.append(motion_path.split(",")[0].strip().split(".")[0])
    return out


def list_smplx_jobs(src_folder: str, tgt_folder: str, override: bool = False, hard_motions: Sequence[str] = (),
                    exclude: Sequence[str] = tuple(EXCLUDE_FILE_CONTENT)):
    """(source file, target pkl) pairs in the reference's order and with its filters (:208-228): natsorted walk,
    ``*_stagei.npz`` skipped, ``.pkl`` / ``.npz`` accepted, existing targets skipped unless ``override``, hard and
    infeasible motions removed."""
    import os
    jobs = []
    for dirpath, _, filenames in os.walk(src_folder):
        for filename in _natsorted(filenames):
            if filename.endswith("_stagei.npz"):
                continue
            if filename.endswith((".pkl", ".npz")):
                src = os.path.join(dirpath, filename)
                tgt = src.replace(src_folder, tgt_folder).replace(".npz", ".pkl")
                if not os.path.exists(tgt) or override:
                    jobs.append((src, tgt))
    hard = set(hard_motions)
    kept = []
    for src, tgt in jobs:
        motion_name = src.split("/")[-1].split(".")[0]
        if motion_name in hard or any(c in motion_name for c in exclude):
            continue
        kept.append((src, tgt))
    return jobs, kept


def list_bvh_jobs(src_folder: str, tgt_folder: str, override: bool = False, verbose: bool = True):
    """(bvh file, target pkl) pairs of ``bvh_to_robot_dataset.py:60-73``: sorted walk, ``.bvh`` only, existing targets
    skipped (with the script's message) unless ``override``."""
    import os
    jobs = []
    for dirpath, _, filenames in os.walk(src_folder):
        for filename in sorted(filenames):
            if not filename.endswith(".bvh"):
                continue
            src = os.path.join(dirpath, filename)
            tgt = src.replace(src_folder, tgt_folder).replace(".bvh", ".pkl")
            if os.path.exists(tgt) and not override:
                if verbose:
                    print(f"Skipping {src} because {tgt} exists")
                continue
            jobs.append((src, tgt))
    return jobs


def _dump(tgt: str, motion: Dict, key_order: Sequence[str]) -> None:
    import os
    import pickle
    os.makedirs(os.path.dirname(tgt) or ".", exist_ok=True)
    with open(tgt, "wb") as f:
        pickle.dump({k: motion[k] for k in key_order}, f)


SMPLX_KEYS = ("fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list")      # :134-141
BVH_KEYS = ("root_pos", "root_rot", "dof_pos", "local_body_pos", "fps", "link_body_list")        # bvh script :141-148


# ----------------------------------------------------------------------------------------------
# The pipeline behind both drivers: loading, retargeting and writing overlap, and a batch
# is sized for the throughput shape of the kernel -- a budget of (padded) frames, thousands of clips -- instead of a
# fixed number of files.
#
#   loader pool (processes / threads)   load(src) -> clip          batch k + 1, k + 2 are being read and parsed ...
#   this thread (the GPU)               retarget(batch) -> dicts   ... while batch k is on the device ...
#   writer threads                      pickle.dump                ... and batch k - 1 is written.
#
# The GPU call releases the GIL (ctypes), so the pools keep running underneath it.  Output files are independent of
# each other, so the jobs may be processed in any order: the drivers sort them by cost (largest first), which keeps
# the padding of a ragged batch small and is also the order LPT sharding over ranks wants.
# ----------------------------------------------------------------------------------------------
class DatasetPipeline:
    def __init__(self, load, retarget, frames_of, key_order, frames_budget: int = 1 << 19, max_clips: int = 16384,
                 loader=None, prefetch_clips: int = 4096, writers: int = 4, verbose: bool = True, label: str = ""):
        """``load(src)`` -> clip (any object; runs in ``loader``, a ``concurrent.futures`` executor, or inline when None);
        ``frames_of(clip)`` -> frames of the clip; ``retarget(clips, srcs)`` -> one motion dict (or None) per clip, on
        this thread; ``frames_budget`` bounds ``len(batch) * longest clip`` (the padded batch the kernel is given).
        A ``retarget`` object with ``begin / add / finish`` (:class:`ClipRetargeter` via :class:`_Staged`) is handed every
        clip when it arrives, so that packing the batch happens while this thread would wait for the loaders anyway."""
        self.load, self.retarget, self.frames_of, self.key_order = load, retarget, frames_of, tuple(key_order)
        self.frames_budget, self.max_clips = max(int(frames_budget), 1), max(int(max_clips), 1)
        self.loader, self.prefetch_clips, self.writers = loader, max(int(prefetch_clips), 1), max(int(writers), 1)
        self.verbose, self.label = verbose, label
        self.stats = {"clips": 0, "frames": 0, "batches": 0, "load_errors": 0, "seconds_gpu": 0.0, "seconds_waiting_for_loads": 0.0,
                      "seconds_waiting_for_writes": 0.0, "seconds_total": 0.0}

    def _submit(self, src):
        import concurrent.futures as cf
        if self.loader is not None:
            return self.loader.submit(self.load, src)
        f = cf.Future()
        try:
            f.set_result(self.load(src))
        except Exception as e:  # noqa: BLE001
            f.set_exception(e)
        return f

    def run(self, jobs: Sequence, after_submit=None) -> int:
        """``jobs`` = (src, tgt) pairs.  Returns the number of files written.  ``after_submit()`` runs on this thread once
        the first window of loads has been handed to the pool (work that can hide under the loaders' start-up)."""
        import collections
        import concurrent.futures as cf
        import time
        t_start = time.perf_counter()
        jobs = list(jobs)
        todo = collections.deque(jobs)
        inflight = collections.deque()           # (src, tgt, future) in job order
        done = 0
        writes = collections.deque()
        with cf.ThreadPoolExecutor(self.writers) as wpool:
            batch, longest = [], 0

            staged = hasattr(self.retarget, "add")

            def flush():
                nonlocal batch, longest, done
                if not batch:
                    return
                t0 = time.perf_counter()
                if staged:
                    motions = self.retarget.finish([s for s, _, _ in batch])
                else:
                    motions = self.retarget([c for _, _, c in batch], [s for s, _, _ in batch])
                self.stats["seconds_gpu"] += time.perf_counter() - t0
                self.stats["batches"] += 1
                for (src, tgt, clip), md in zip(batch, motions):
                    if md is None:
                        continue
                    writes.append(wpool.submit(_dump, tgt, md, self.key_order))
                    done += 1
                    self.stats["clips"] += 1
                    self.stats["frames"] += int(clip) if staged else int(self.frames_of(clip))
                    if self.verbose:
                        print(f"Processed {done}/{len(jobs)}: {tgt}")
                while len(writes) > 4 * self.max_clips:          # bound the dicts waiting for the writers
                    writes.popleft().result()
                batch, longest = [], 0

            while todo or inflight:
                while todo and len(inflight) < self.prefetch_clips:
                    src, tgt = todo.popleft()
                    inflight.append((src, tgt, self._submit(src)))
                if after_submit is not None:
                    after_submit()
                    after_submit = None
                src, tgt, fut = inflight.popleft()
                try:
                    t_w = time.perf_counter()
                    clip = fut.result()
                    self.stats["seconds_waiting_for_loads"] += time.perf_counter() - t_w
                except Exception as e:  # noqa: BLE001 -- the scripts print and skip (smplx_to_robot_dataset.py:62-76)
                    print(f"Error loading {src}: {e}")
                    self.stats["load_errors"] += 1
                    continue
                n = int(self.frames_of(clip))
                if staged:
                    if not batch:
                        self.retarget.begin(n, self.max_clips, self.frames_budget)
                    if not self.retarget.add(clip):
                        flush()
                        self.retarget.begin(n, self.max_clips, self.frames_budget)
                        if not self.retarget.add(clip):
                            raise RuntimeError(f"{src}: clip does not fit an empty batch")
                    batch.append((src, tgt, n))           # (the frames now live in the staged batch)
                    longest = max(longest, n)
                    continue
                if batch and ((len(batch) + 1) * max(longest, n) > self.frames_budget or len(batch) >= self.max_clips):
                    flush()
                batch.append((src, tgt, clip))
                longest = max(longest, n)
            flush()
            t_w = time.perf_counter()
            for w in writes:
                w.result()                                        # a failed write raises here
            self.stats["seconds_waiting_for_writes"] = time.perf_counter() - t_w
        self.stats["seconds_total"] = time.perf_counter() - t_start
        return done


class _Staged:
    """:class:`ClipRetargeter` behind the staged protocol of :class:`DatasetPipeline` with a fixed fps per clip."""

    def __init__(self, rt: "ClipRetargeter", fps: float):
        self.rt, self.fps = rt, fps

    def begin(self, longest, max_clips, frames_budget):
        self.rt.begin(longest, max_clips, frames_budget)

    def add(self, clip):
        return self.rt.add(clip)

    def finish(self, files):
        return self.rt.finish([self.fps] * len(files))


def job_cost(src: str) -> int:
    """Cheap proxy of a file's frame count, the same on every rank: BVH -- the ``Frames:`` line of its header; anything
    else -- the file size.  Used to order the jobs and to shard them over ranks (LPT)."""
    import os
    if src.endswith(".bvh"):
        try:
            with open(src, "r", errors="replace") as f:
                for i, line in enumerate(f):
                    if line.lstrip().startswith("Frames:"):
                        return int(line.split(":")[1])
                    if i > 4000:
                        break
        except (OSError, ValueError):
            pass
    try:
        return int(os.path.getsize(src))
    except OSError:
        return 0


def shard_jobs(jobs: Sequence, rank: int = 0, world: int = 1, cost=job_cost):
    """This rank's jobs, largest first: ``sharding.lpt_partition`` on the cost of every job (deterministic: every rank
    computes the same partition from the same folder walk)."""
    from .sharding import lpt_partition
    jobs = list(jobs)
    costs = [int(cost(s)) for s, _ in jobs]
    mine = lpt_partition(costs, world)[rank] if world > 1 else list(range(len(jobs)))
    mine = sorted(mine, key=lambda i: (-costs[i], i))
    return [jobs[i] for i in mine]


def _skip_existing(jobs, override: bool, verbose: bool):
    """skip-if-exists of the scripts (bvh_to_robot_dataset.py:66-70, smplx_to_robot_dataset.py:213-215) on a job list"""
    import os
    if override:
        return list(jobs)
    kept = []
    for src, tgt in jobs:
        if os.path.exists(tgt):
            if verbose:
                print(f"Skipping {src} because {tgt} exists")
            continue
        kept.append((src, tgt))
    return kept


def _loader_pool(kind: str, workers: int):
    """Process pool (BVH: pure-Python parsing holds the GIL) or thread pool (npz: the reads release it).  Processes are
    SPAWNED, never forked: this process may already have initialised the GPU."""
    import concurrent.futures as cf
    import multiprocessing as mp
    if workers <= 0:
        return None
    if kind == "process":
        return cf.ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn"))
    return cf.ThreadPoolExecutor(workers)


def default_workers(world: int = 1) -> int:
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    return max(1, min(16, n // max(world, 1)))


# ---- BVH ---------------------------------------------------------------------------------------------------------------
def _load_bvh_clip(args):
    """(file, body names) -> packed frames f64[T, nhuman, 7]; module-level: runs in spawned worker processes."""
    from .utils.lafan1 import load_lafan1_packed
    f, names = args
    return load_lafan1_packed(f, names)[0]


def run_bvh_dataset(src_folder: str, tgt_folder: str, robot: str, override: bool = False, batch_files: int = 0,
                    retarget=None, verbose: bool = True, frames_budget: int = 1 << 19, loader_workers: int = -1,
                    rank: int = 0, world: int = 1, load=None, stats: Optional[Dict] = None) -> int:
    """``bvh_to_robot_dataset.py`` (:60-157) on the pipeline above.  ``batch_files`` > 0 additionally caps the clips of a
    launch (0: only the frames budget does).  ``retarget(clips, files)`` and ``load(file)`` are injectable (tests)."""
    # the partition is computed on ALL source files (a rank that starts later must not see another partition because
    # some targets exist by then); skip-if-exists is applied to the rank's own shard
    workers = default_workers(world) if loader_workers < 0 else loader_workers
    pool = _loader_pool("process" if load is None else "thread", workers)     # first: the workers start up under what follows
    jobs = _skip_existing(shard_jobs(list_bvh_jobs(src_folder, tgt_folder, True, False), rank, world), override, verbose)
    gmr = None
    if load is None or retarget is None:
        gmr = GeneralMotionRetargeting("bvh", robot, actual_human_height=1.75)
    names = gmr.human_body_names if gmr is not None else None
    rt = None
    if retarget is None:
        rt = ClipRetargeter("bvh", robot, 1.75, height_adjust=False, root_origin_offset=False)    # HEIGHT_ADJUST = False, :128
        retarget = _Staged(rt, 30)
    try:
        pipe = DatasetPipeline((lambda f: _load_bvh_clip((f, names))) if (load is None and pool is None) else (load or _BvhLoad(names)),
                               retarget, len, BVH_KEYS, frames_budget, batch_files if batch_files > 0 else 16384, pool,
                               verbose=verbose, label="bvh")
        # lock the pages of the first batch while the loader processes start up and parse the first files
        warm = (lambda: rt.reserve(min(frames_budget, 420 * len(jobs)))) if (rt is not None and len(jobs) > 64) else None
        done = pipe.run(jobs, warm)
    finally:
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
    if stats is not None:
        stats.update(pipe.stats)
        if rt is not None:
            stats["seconds_gpu_parts"] = dict(rt.timing)
    if verbose:
        print("Done. saved to ", tgt_folder)
    return done


class _BvhLoad:
    """Picklable ``load`` for the process pool."""

    def __init__(self, names):
        self.names = list(names)

    def __call__(self, f):
        return _load_bvh_clip((f, self.names))


# ---- SMPL-X ---------------------------------------------------------------------------------------------------------------
def _load_smplx_raw(f):
    """The arrays ``load_smplx_file`` reads from an AMASS-style file (utils/smpl.py:12-41), nothing computed: runs in the
    loader threads."""
    with np.load(f, allow_pickle=False) as z:
        return {k: np.asarray(z[k]) for k in ("gender", "betas", "root_orient", "pose_body", "trans", "mocap_frame_rate")}


def retarget_smplx_loaded(raws: Sequence[Dict], smplx_body_model_path: str, tgt_robot: str, tgt_fps: int = 30,
                          height_adjust: bool = True, root_origin_offset: bool = True) -> List[Optional[Dict]]:
    """``process_file`` (smplx_to_robot_dataset.py:39-146) for many already-read files: joints-only body model and fps
    alignment on the device per clip, the clips grouped by human height (a file's height comes from its betas, :36-39)
    into ONE group launch, two FK launches for the post-processing of all clips together."""
    from . import _lib
    from .utils import smpl
    packed, fps_of, height = {}, {}, {}
    gmr_of: Dict[float, GeneralMotionRetargeting] = {}
    for i, d in enumerate(raws):
        bm = smpl.body_model_for(smplx_body_model_path, str(d["gender"]))
        betas = np.asarray(d["betas"])
        h = float(1.66 + 0.1 * (betas[0] if betas.ndim == 1 else betas[0, 0]))
        g = gmr_of.get(h)
        if g is None:
            g = gmr_of[h] = GeneralMotionRetargeting("smplx", tgt_robot, actual_human_height=h)
        packed[i], fps_of[i] = smpl.smplx_frames_packed_fused(g, d, bm, tgt_fps=tgt_fps)      # body model + alignment, one call
        height[i] = h
    jobs, members = [], []
    for h, g in gmr_of.items():
        idxs = [i for i in range(len(raws)) if height[i] == h]
        lens = np.array([packed[i].shape[0] for i in idxs], dtype=np.int32)
        T = max(int(lens.max()), 1)
        human = np.zeros((len(idxs), T, len(g.human_body_names), 7))
        human[..., 3] = 1.0
        for k, i in enumerate(idxs):
            human[k, : lens[k]] = packed[i]
        jobs.append({"solver": g.hip_solver, "human": human, "lens": lens})
        members.append((idxs, lens))
    results = _lib.retarget_group(jobs) if jobs else []
    qpos: List[Optional[np.ndarray]] = [None] * len(raws)
    for (idxs, lens), (q, _, status) in zip(members, results):
        if (status != 0).any():
            raise RuntimeError(f"IK failed for clips {[idxs[k] for k in np.nonzero(status)[0]]}")
        for k, i in enumerate(idxs):
            qpos[i] = q[k, : lens[k]]
    if not raws:
        return []
    km = KinematicsModel(next(iter(gmr_of.values())).xml_file)
    return postprocess_clips(qpos, km, [fps_of[i] for i in range(len(raws))], height_adjust, root_origin_offset)


def run_smplx_dataset(src_folder: str, tgt_folder: str, robot: str, smplx_folder: str, override: bool = False,
                      hard_motion_files: Sequence[str] = (), batch_files: int = 0, retarget=None, verbose: bool = True,
                      frames_budget: int = 1 << 19, loader_workers: int = -1, rank: int = 0, world: int = 1, load=None,
                      stats: Optional[Dict] = None) -> int:
    """``smplx_to_robot_dataset.py:main`` (:171-242) on the pipeline above.  ``retarget(raws, files)`` defaults to
    :func:`retarget_smplx_loaded`, ``load(file)`` to the npz reader (tests inject stand-ins).  Returns the number of pkl
    files this rank wrote."""
    hard = load_hard_motions(hard_motion_files)
    all_jobs, jobs = list_smplx_jobs(src_folder, tgt_folder, override, hard)
    if verbose and rank == 0:
        print("full args_list:", len(all_jobs))
        print("new args_list:", len(jobs))
        print(f"Total number of files to process: {len(jobs)}")
    if world > 1:      # partition ALL files (see run_bvh_dataset), then drop what exists from this rank's shard
        jobs = _skip_existing(shard_jobs(list_smplx_jobs(src_folder, tgt_folder, True, hard)[1], rank, world), override, False)
    else:
        jobs = shard_jobs(jobs, rank, world)
    if retarget is None:
        def retarget(raws, files):
            return retarget_smplx_loaded(raws, smplx_folder, robot)
    workers = default_workers(world) if loader_workers < 0 else loader_workers
    pool = _loader_pool("thread", workers if len(jobs) > 1 else 0)
    try:
        pipe = DatasetPipeline(load or _load_smplx_raw, retarget, lambda d: len(d["trans"]) if isinstance(d, dict) else len(d),
                               SMPLX_KEYS, frames_budget, batch_files if batch_files > 0 else 16384, pool, verbose=verbose,
                               label="smplx")
        done = pipe.run(jobs)
    finally:
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
    if stats is not None:
        stats.update(pipe.stats)
    if verbose:
        print("Done. Saved to ", tgt_folder)
    return done


def _retarget_bvh_skipping_errors(bvh_files: Sequence[str], tgt_robot: str) -> List[Optional[Dict]]:
    """:func:`retarget_bvh_files` with the script's print-and-skip for files that fail to load (:77-82)."""
    from .utils.lafan1 import load_lafan1_packed
    gmr = GeneralMotionRetargeting("bvh", tgt_robot, actual_human_height=1.75)
    ok, clips = [], []
    for i, f in enumerate(bvh_files):
        try:
            clips.append(load_lafan1_packed(f, gmr.human_body_names)[0])
            ok.append(i)
        except Exception as e:  # noqa: BLE001 - mirrors the script
            print(f"Error loading {f}: {e}")
    out: List[Optional[Dict]] = [None] * len(bvh_files)
    if clips:
        res = retarget_clips("bvh", tgt_robot, clips, [30] * len(clips), actual_human_height=1.75,
                             height_adjust=False, root_origin_offset=False)
        for i, md in zip(ok, res):
            out[i] = md
    return out


def main(argv=None) -> int:
    import argparse
    import json
    import os
    import pathlib
    import sys
    import time
    ap = argparse.ArgumentParser(description="Retarget a folder of SMPL-X (AMASS) or BVH (LAFAN1) motions to a robot: "
                                             "the reference's dataset scripts on the MI355X kernels")
    ap.add_argument("--source", choices=["smplx", "bvh"], default="smplx")
    ap.add_argument("--robot", default="unitree_g1")
    ap.add_argument("--src_folder", type=str, required=True)
    ap.add_argument("--tgt_folder", type=str, required=True)
    ap.add_argument("--override", default=False, action="store_true")
    ap.add_argument("--num_cpus", default=-1, type=int, help="loader workers per rank (the reference's mp.Pool size; default: "
                                                             "this rank's share of the host's cores, at most 16)")
    ap.add_argument("--target_fps", default=30, type=int, help="accepted for compatibility (the BVH script ignores it too)")
    ap.add_argument("--smplx_folder", type=str, default=None, help="SMPL-X body models (default: <assets>/body_models)")
    ap.add_argument("--hard_motions", nargs="*", default=None, help="difficulty reports (default: <assets>/hard_motions/{0,1}.txt)")
    ap.add_argument("--batch_files", type=int, default=0, help="cap on the clips of one IK launch (0: the frames budget alone)")
    ap.add_argument("--frames_budget", type=int, default=1 << 19, help="padded frames (clips x longest clip) of one IK launch")
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; the files are LPT-sharded over them")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args(argv)
    from . import launcher
    if a.gpus > 1 and not launcher.is_rank_process():
        # become the launcher: one rank process per GPU, nothing of the GPU is touched here
        cmd = [sys.executable, "-m", "general_motion_retargeting_amd.dataset"] + list(sys.argv[1:] if argv is None else argv)
        return launcher.self_launch(a.gpus, cmd, "GMR_DATASET_TIMEOUT", 86400.0)
    from . import _lib, comm as gcomm, sharding
    rank, local_rank, world = gcomm.env_rank_world()
    cm = None
    if world > 1:
        L = _lib.lib()
        _lib.require_gpu()
        _lib.check(L.gmr_set_device(local_rank % max(L.gmr_device_count(), 1)))
        cm = gcomm.create()
        # the ONE collective of the path: rank 0's compiled (robot, task set) to every rank; a rank whose own plugin files
        # compile to other bytes would silently produce other motions -- that is an error, not a fallback
        g0 = GeneralMotionRetargeting(a.source, a.robot)
        mb, ts = sharding.broadcast_blobs(g0._model_blob if rank == 0 else None, g0._taskset_blob if rank == 0 else None, rank, cm)
        if not (np.array_equal(mb.view(np.uint8), g0._model_blob.view(np.uint8))):
            raise SystemExit(f"rank {rank}: the robot model compiled here differs from rank 0's")
    t0 = time.perf_counter()
    stats: Dict = {}
    if a.source == "bvh":
        n = run_bvh_dataset(a.src_folder, a.tgt_folder, a.robot, a.override, a.batch_files, verbose=not a.quiet,
                            frames_budget=a.frames_budget, loader_workers=a.num_cpus, rank=rank, world=world, stats=stats)
    else:
        from .params import ASSET_ROOT
        assets = pathlib.Path(ASSET_ROOT)
        smplx_folder = a.smplx_folder or str(assets / "body_models")
        hard = a.hard_motions if a.hard_motions is not None else [str(assets / "hard_motions" / "0.txt"),
                                                                    str(assets / "hard_motions" / "1.txt")]
        n = run_smplx_dataset(a.src_folder, a.tgt_folder, a.robot, smplx_folder, a.override, hard, a.batch_files,
                              verbose=not a.quiet, frames_budget=a.frames_budget, loader_workers=a.num_cpus, rank=rank,
                              world=world, stats=stats)
    dt = time.perf_counter() - t0
    frames = float(stats.get("frames", 0))
    if cm is not None:
        tot = cm.allgather(frames)
        files = cm.allgather(float(n))
        dt = cm.allreduce_max(dt)
        cm.barrier()
        cm.close()
        frames, n_all = sum(tot), int(sum(files))
    else:
        n_all = n
    if rank == 0:
        print(json.dumps({"dataset_summary": {"files_written": n_all, "frames": int(frames), "seconds": dt,
                                              "frames_per_s": frames / dt if dt > 0 else 0.0, "ranks": world,
                                              "rank0": {k: stats.get(k) for k in ("batches", "seconds_gpu", "seconds_waiting_for_loads",
                                                                                    "seconds_waiting_for_writes", "load_errors", "seconds_gpu_parts")}}}))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
