"""``GeneralMotionRetargeting`` with the reference's constructor / ``retarget()`` surface, backed by
the HIP kernels behind the C-ABI (no CPU fallback: without ``libgmrhip.so`` and a GPU every
numerical entry point raises).

Mirrors reference ``general_motion_retargeting/motion_retarget.py``:

* ``__init__``                      :13-72   (same signature, same public attributes)
* ``update_targets`` / ``retarget`` :117-185 (same in-place ``to_numpy`` mutation of the caller's
  dict, same ``KeyError`` behaviour, returns a fresh float64 ``qpos`` copy)
* ``scaled_human_data``, ``error1`` / ``error2`` :118-124, :188-200 -- served from what the kernel computed for the
  last frame (its ``tgt_out`` / ``err_out``); after ``update_targets()`` alone, from a solve-free launch
* ``scale_human_data`` / ``offset_human_data`` / ``offset_human_data_to_ground`` / ``to_numpy``
  :203-270 (NumPy helpers kept for callers; the retargeting path preprocesses on the device)

plus the batched entry points that the per-frame API cannot express (``retarget_clip``,
``retarget_streams``): many frames / many independent streams per launch, time loop on device.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .ik_config import TaskTables, build_task_tables, pack_model, pack_taskset
from .models import load_ik_config, load_robot
from .params import IK_CONFIG_DICT, ROBOT_XML_DICT


class TargetNotSet(Exception):
    """A frame task has no target (mirrors mink.TargetNotSet for duplicated human bodies)."""


class _Data:
    def __init__(self, qpos):
        self.qpos = qpos


class _Configuration:
    """Minimal stand-in for ``mink.Configuration``: holds ``q`` (``data.qpos``) and the model."""

    def __init__(self, model, q):
        self.model = model
        self.data = _Data(q)

    @property
    def q(self):
        return self.data.qpos.copy()

    def update(self, q=None):
        if q is not None:
            self.data.qpos = np.asarray(q, dtype=np.float64).copy()


def _quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _quat_rotate(q, v):
    qv = np.concatenate([[0.0], v])
    return _quat_mul(_quat_mul(q, qv), q * np.array([1.0, -1.0, -1.0, -1.0]))[1:]


class GeneralMotionRetargeting:
    """General Motion Retargeting (GMR), MI355X-native backend."""

    def __init__(
        self,
        src_human: str,
        tgt_robot: str,
        actual_human_height: float = None,
        solver: str = "daqp",
        damping: float = 5e-1,
        verbose: bool = False,
    ) -> None:
        self.xml_file = str(ROBOT_XML_DICT[tgt_robot])
        if verbose:
            print("Use robot model: ", self.xml_file)
        self.model = load_robot(self.xml_file)

        ik_config_path = IK_CONFIG_DICT[src_human][tgt_robot]
        ik_config = load_ik_config(ik_config_path)
        if verbose:
            print("Use IK config: ", ik_config_path)

        self._tables: TaskTables = build_task_tables(ik_config, actual_human_height, damping)
        tt = self._tables
        # same public attributes as the reference (:47-59)
        self.ik_match_table1 = ik_config["ik_match_table1"]
        self.ik_match_table2 = ik_config["ik_match_table2"]
        self.human_root_name = tt.human_root_name
        self.robot_root_name = tt.robot_root_name
        self.use_ik_match_table1 = tt.use_stage[0]
        self.use_ik_match_table2 = tt.use_stage[1]
        self.human_scale_table = dict(tt.scale_table)
        self.ground = tt.ground_height * np.array([0, 0, 1])
        self.max_iter = tt.max_iter
        self.solver = solver          # kept for signature compatibility; the device QP is exact
        self.damping = damping
        self.pos_offsets1 = {k: v.copy() for k, v in tt.pos_offsets1.items()}
        self.rot_offsets1 = {k: v.copy() for k, v in tt.rot_offsets1.items()}   # wxyz, normalised
        self._raw_frame = None
        self._scaled_cache = None

        self._human_names: List[str] = tt.human_names
        self._model_blob = pack_model(self.model)
        self._taskset_blob = pack_taskset(self.model, tt)
        self._solver: Optional[_lib.Solver] = None
        self.setup_retarget_configuration()

    # ------------------------------------------------------------------ #
    def setup_retarget_configuration(self):
        """Fresh configuration at ``qpos0`` (reference :74-75); tasks live in the packed task set."""
        self.configuration = _Configuration(self.model, self.model.qpos0.copy())

    @property
    def hip_solver(self) -> _lib.Solver:
        if self._solver is None:
            self._solver = _lib.Solver(self._model_blob, self._taskset_blob)   # raises without GPU/library
        return self._solver

    @property
    def human_body_names(self) -> List[str]:
        """Order of the bodies in the packed ``[nhuman, 7]`` frame layout."""
        return list(self._human_names)

    # ------------------------------------------------------------------ #
    # packing of the reference's dict format (App. D) into the C-ABI layout
    # ------------------------------------------------------------------ #
    def _check_names(self, human_data: Dict) -> None:
        tt = self._tables
        if tt.human_root_name not in human_data:
            raise KeyError(tt.human_root_name)
        # bodies that survive scale_human_data must have table-1 offsets (reference :241)
        for n in human_data.keys():
            if n in tt.scale_table and n not in tt.pos_offsets1:
                raise KeyError(n)
        for s in range(2):
            if not tt.use_stage[s]:
                continue
            hs = tt.stages[s].human_names
            if len(set(hs)) != len(hs):
                raise TargetNotSet("two robot frames are mapped to one human body: a task has no target")
            for n in hs:
                if n not in human_data or n not in tt.scale_table:
                    raise KeyError(n)

    def pack_frame(self, human_data: Dict) -> np.ndarray:
        """dict{name: (pos, quat_wxyz)} -> f64[nhuman, 7]; bodies absent from the dict become NaN rows."""
        self._check_names(human_data)
        out = np.full((len(self._human_names), 7), np.nan)
        for i, n in enumerate(self._human_names):
            if n in human_data:
                out[i, :3] = human_data[n][0]
                out[i, 3:] = human_data[n][1]
        return out

    def pack_frames(self, frames: Sequence[Dict]) -> np.ndarray:
        return np.stack([self.pack_frame(self.to_numpy(f)) for f in frames]) if len(frames) else \
            np.zeros((0, len(self._human_names), 7))

    # ------------------------------------------------------------------ #
    # reference API
    # ------------------------------------------------------------------ #
    def update_targets(self, human_data, offset_to_ground=False):
        human_data = self.to_numpy(human_data)
        self._set_frame(self.pack_frame(human_data), offset_to_ground, list(human_data.keys()))

    def _set_frame(self, frame, offset_to_ground, key_order=None):
        """New targets (a packed frame): everything derived from the previous ones is stale."""
        self._raw_frame = frame
        self._ground_flag = bool(offset_to_ground)
        self._key_order = key_order
        self._targets = None          # f64[nhuman, 7]: the kernel's preprocessed frame (tgt_out)
        self._errors = None           # (error1, error2) at self._errors_q
        self._errors_q = None
        self._scaled_cache = None

    def _flags(self, offset_to_ground):
        return _lib.FLAG_OFFSET_TO_GROUND if offset_to_ground else 0

    def _evaluate(self):
        """Targets and residual norms of the current (frame, configuration) WITHOUT a solve: the device runs the
        preprocessing and both tables' residuals (GMR_FLAG_EVAL_ONLY) -- update_targets() followed by
        scaled_human_data / error1() / error2() in the reference (:117-136, :188-200)."""
        if getattr(self, "_raw_frame", None) is None:
            raise TargetNotSet("retarget()/update_targets() has not been called")
        q = self.configuration.data.qpos
        _, _, status, tg, er = self.hip_solver.retarget_streams(
            q[None], self._raw_frame[None, None], flags=self._flags(self._ground_flag) | _lib.FLAG_EVAL_ONLY,
            want_targets=True, want_errors=True)
        if status[0] != 0:
            raise RuntimeError(f"evaluation failed (status {int(status[0])})")
        self._targets, self._errors, self._errors_q = tg[0, 0], er[0, 0], q.copy()
        self._scaled_cache = None

    @property
    def scaled_human_data(self):
        """``{name: [pos, quat_wxyz]}`` after scale / offset / ground (reference :118-124), exactly the values the
        IK kernel computed for the last frame (its ``tgt_out``), in the reference's dict order."""
        if self._scaled_cache is None:
            if getattr(self, "_raw_frame", None) is None:
                return None
            if self._targets is None:
                self._evaluate()
            names = self._human_names
            rows = {n: self._targets[i] for i, n in enumerate(names) if not np.isnan(self._raw_frame[i, 0])}
            order = [self.human_root_name] + [n for n in (self._key_order or names) if n != self.human_root_name]
            self._scaled_cache = {n: [rows[n][:3].copy(), rows[n][3:].copy()] for n in order if n in rows}
        return self._scaled_cache

    @scaled_human_data.setter
    def scaled_human_data(self, value):
        self._scaled_cache = value

    def _run(self, human, offset_to_ground, key_order=None):
        """One launch over ``human[T, nhuman, 7]`` continuing from the current configuration; keeps the last
        frame's targets and residual norms so that scaled_human_data / error1() / error2() refer to it."""
        q_out, nsolve, status, tg, er = self.hip_solver.retarget_streams(
            self.configuration.data.qpos[None], human[None], flags=self._flags(offset_to_ground),
            want_targets=True, want_errors=True)
        if status[0] != 0:
            raise RuntimeError(f"IK failed (status {int(status[0])}): QP not solvable / non-finite input")
        self._set_frame(human[-1], offset_to_ground, key_order)
        self.configuration.data.qpos = q_out[0, -1].copy()
        self._targets, self._errors, self._errors_q = tg[0, -1], er[0, -1], q_out[0, -1].copy()
        return q_out[0], nsolve[0]

    def retarget(self, human_data, offset_to_ground=False):
        """One frame (reference :139-185): warm-started from the previous call, returns qpos f64[nq]."""
        human_data = self.to_numpy(human_data)
        frame = self.pack_frame(human_data)
        q, ns = self._run(frame[None], offset_to_ground, list(human_data.keys()))
        self.last_num_solves = ns[0].copy()
        return q[0].copy()

    def retarget_packed(self, frame: np.ndarray, offset_to_ground=False) -> np.ndarray:
        """:meth:`retarget` for a frame that is already packed (``f64[nhuman, 7]``, rows in
        ``human_body_names`` order, NaN rows = absent bodies): the streaming path (utils/optitrack.py)."""
        frame = np.ascontiguousarray(frame, dtype=np.float64)
        if frame.shape != (len(self._human_names), 7):
            raise ValueError(f"packed frame must be [{len(self._human_names)}, 7], got {frame.shape}")
        q, ns = self._run(frame[None], offset_to_ground)
        self.last_num_solves = ns[0].copy()
        return q[0].copy()

    def retarget_clip(self, frames, offset_to_ground=False) -> np.ndarray:
        """All frames of one clip in ONE launch (time loop on device); continues from the current
        configuration exactly like calling :meth:`retarget` per frame.  ``frames`` is a sequence of
        ``human_data`` dicts or an array ``[T, nhuman, 7]``.  Returns ``qpos f64[T, nq]``."""
        key_order = None
        if isinstance(frames, np.ndarray):
            human = np.ascontiguousarray(frames, dtype=np.float64)
        else:
            human = self.pack_frames(frames)
            key_order = list(frames[-1].keys()) if len(frames) else None
        if human.shape[0] == 0:
            return np.zeros((0, self.model.nq))
        q, ns = self._run(human, offset_to_ground, key_order)
        self.last_num_solves = ns.copy()
        return q

    def retarget_streams(self, human: np.ndarray, q0: Optional[np.ndarray] = None, lens=None,
                         offset_to_ground=False):
        """Many independent streams ``human[S, T, nhuman, 7]`` (each from ``q0[s]``, default
        ``qpos0`` = a fresh object per clip as the dataset scripts do).  Does not touch this object's
        configuration.  Returns ``(qpos[S, T, nq], nsolve[S, T, 2], status[S])``."""
        S = human.shape[0]
        if q0 is None:
            q0 = np.broadcast_to(self.model.qpos0, (S, self.model.nq)).copy()
        return self.hip_solver.retarget_streams(q0, human, lens=lens, flags=self._flags(offset_to_ground))

    # ---- errors (reference :188-200): evaluated by the kernel ------------------------------
    def _error(self, stage: int) -> float:
        if getattr(self, "_raw_frame", None) is None:
            raise TargetNotSet("retarget()/update_targets() has not been called")
        if not self._tables.use_stage[stage]:
            raise ValueError("need at least one array to concatenate")    # the reference's empty task list (:190,:197)
        if self._errors is None or not np.array_equal(self._errors_q, self.configuration.data.qpos):
            self._evaluate()
        return float(self._errors[stage])

    def error1(self):
        return self._error(0)

    def error2(self):
        return self._error(1)

    # ---- preprocessing helpers (reference :203-270) ----------------------------------------------
    # Public-by-convention NumPy methods of the reference class, kept for callers that use them directly.  The
    # retargeting path does not: the kernel preprocesses the raw packed frame (and scaled_human_data above is
    # its output).
    def to_numpy(self, human_data):
        for body_name in human_data.keys():
            human_data[body_name] = [np.asarray(human_data[body_name][0]), np.asarray(human_data[body_name][1])]
        return human_data

    def scale_human_data(self, human_data, human_root_name, human_scale_table):
        root_pos, root_quat = human_data[human_root_name]
        scaled_root_pos = human_scale_table[human_root_name] * root_pos
        out = {human_root_name: (scaled_root_pos, root_quat)}
        for name in human_data.keys():
            if name not in human_scale_table or name == human_root_name:
                continue
            local = (human_data[name][0] - root_pos) * human_scale_table[name]
            out[name] = (local + scaled_root_pos, human_data[name][1])
        return out

    def offset_human_data(self, human_data, pos_offsets, rot_offsets):
        """Rotation offset first, then the position offset in the updated local frame."""
        out = {}
        for name in human_data.keys():
            pos, quat = human_data[name]
            q = np.asarray(quat, dtype=np.float64)
            q = q / np.linalg.norm(q)
            qo = np.asarray(rot_offsets[name], dtype=np.float64)
            uq = _quat_mul(q, qo / np.linalg.norm(qo))
            uq = uq / np.linalg.norm(uq)
            out[name] = [pos + _quat_rotate(uq, np.asarray(pos_offsets[name], dtype=np.float64)), uq]
        return out

    def offset_human_data_to_ground(self, human_data):
        ground_offset = self._tables.ground_offset
        lowest = np.inf
        for name in human_data.keys():
            if "Foot" not in name and "foot" not in name:
                continue
            if human_data[name][0][2] < lowest:
                lowest = human_data[name][0][2]
        out = {}
        for name in human_data.keys():
            pos, quat = human_data[name]
            out[name] = [pos - np.array([0, 0, lowest]) + np.array([0, 0, ground_offset]), quat]
        return out
