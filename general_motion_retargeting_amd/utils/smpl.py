"""SMPL-X frame extraction in front of the retargeting loop, the MI355X
counterpart of the reference's ``general_motion_retargeting/utils/smpl.py``:

=============================================  =====================================================
reference (utils/smpl.py)                      here
=============================================  =====================================================
``load_smpl_file`` (:8-10)                     ``load_smpl_file``
``load_smplx_file`` (:12-41)                   ``load_smplx_file`` -- joints-only body model
                                               (``SmplxBodyModel``; the 10 475-vertex mesh the
                                               reference evaluates for every frame is never built)
``get_smplx_data`` (:44-73)                    ``get_smplx_data``
``slerp`` (:76-107)                            ``slerp`` (quaternion arrays instead of Rotation objects)
``get_smplx_data_offline_fast`` (:109-197)     ``get_smplx_data_offline_fast`` (same return value) and
                                               ``smplx_frames_packed`` (straight to the packed
                                               ``human[T', nhuman, 7]`` array of the IK kernel)
=============================================  =====================================================

All arithmetic runs in the HIP kernels of ``csrc/gmr_smplx.hip`` through the C-ABI (``gmr_smplx_*``); there
is no CPU fallback.  Pose half (alignment + orientation chain): pinned by ``tests/golden/g_smplx.npz``.  Body
model half: the ``smplx`` package and its model files are third-party and absent here -> parity unpinned,
restated from the published formulation (joint regressor on the shaped template, Rodrigues, rigid chain).
"""
from __future__ import annotations

import os
import types
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .. import _lib

# SMPL-X joint order of the first 55 entries of smplx.joint_names.JOINT_NAMES (body 0-21, jaw, eyes, hands)
SMPLX_JOINT_NAMES: Tuple[str, ...] = (
    "pelvis", "left_hip", "right_hip", "spine1", "left_knee", "right_knee", "spine2", "left_ankle", "right_ankle",
    "spine3", "left_foot", "right_foot", "neck", "left_collar", "right_collar", "head", "left_shoulder",
    "right_shoulder", "left_elbow", "right_elbow", "left_wrist", "right_wrist", "jaw", "left_eye_smplhf",
    "right_eye_smplhf",
) + tuple(f"{side}_{finger}{k}" for side in ("left", "right")
          for finger in ("index", "middle", "pinky", "ring", "thumb") for k in (1, 2, 3))
JOINT_NAMES = SMPLX_JOINT_NAMES

# kinematic tree of those 55 joints (kintree_table[0] of the SMPL-X model files)
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15]
    + [20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38]
    + [21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)

_HANDLES: Dict[tuple, _lib.SmplxHandle] = {}


def _handle(parents, sel=None) -> _lib.SmplxHandle:
    key = (tuple(int(p) for p in parents), None if sel is None else tuple(int(s) for s in sel))
    h = _HANDLES.get(key)
    if h is None:
        h = _HANDLES[key] = _lib.SmplxHandle(np.asarray(parents), sel)
    return h


def _np(x, dtype=None):
    if hasattr(x, "detach"):                      # torch tensor from a genuine smplx body model
        x = x.detach().cpu().numpy()
    return np.asarray(x) if dtype is None else np.asarray(x, dtype=dtype)


class SmplxOutput(types.SimpleNamespace):
    """The three fields of the body model output the reference reads: ``global_orient [N, 3]``,
    ``full_pose [N, 3 J]``, ``joints [N, >=J, 3]`` (float32 like the reference's torch tensors)."""


class SmplxBodyModel:
    """Joints-only SMPL-X body model: what ``smplx.create(path, "smplx", gender=..., use_pca=False)`` is used
    for at utils/smpl.py:14-34.  Holds the joint regressor already applied to the template and to the shape
    directions, so a clip costs one [J,3,nb] x [nb] product + one kernel launch instead of a mesh per frame."""

    def __init__(self, j_template, j_shapedirs, parents, hand_mean=None, joint_names=SMPLX_JOINT_NAMES):
        self.j_template = np.asarray(j_template, dtype=np.float64)            # [J, 3]
        self.j_shapedirs = np.asarray(j_shapedirs, dtype=np.float64)          # [J, 3, nb]
        self.parents = np.asarray(parents, dtype=np.int64).copy()
        self.parents[0] = -1
        self.num_joints = len(self.parents)
        self.num_betas = self.j_shapedirs.shape[-1]
        self.joint_names = tuple(joint_names[: self.num_joints])
        # pose mean added to full_pose by the model (flat_hand_mean=False default): zeros except the hands
        self.pose_mean = np.zeros((self.num_joints, 3), dtype=np.float32)
        if hand_mean is not None:
            hm = np.asarray(hand_mean, dtype=np.float32).reshape(-1, 3)
            self.pose_mean[self.num_joints - len(hm):] = hm

    @classmethod
    def from_arrays(cls, v_template, shapedirs, j_regressor, parents, num_betas=10, hand_mean=None):
        jr = np.asarray(j_regressor, dtype=np.float64)
        n = len(parents)
        jr = jr[:n]
        sd = np.asarray(shapedirs, dtype=np.float64)[:, :, :num_betas]
        return cls(jr @ np.asarray(v_template, dtype=np.float64), np.einsum("jv,vcb->jcb", jr, sd), parents, hand_mean)

    @classmethod
    def from_model_path(cls, model_path, gender="neutral", num_betas=10, ext="npz"):
        """``model_path/smplx/SMPLX_<GENDER>.npz`` (the layout ``smplx.create`` expects) or a file path."""
        path = model_path
        if os.path.isdir(path):
            sub = os.path.join(path, "smplx")
            path = os.path.join(sub if os.path.isdir(sub) else path, f"SMPLX_{str(gender).upper()}.{ext}")
        with np.load(path, allow_pickle=False) as f:
            parents = np.asarray(f["kintree_table"])[0].astype(np.int64)[:55]
            hm = None
            if "hands_meanl" in f and "hands_meanr" in f:
                hm = np.concatenate([np.asarray(f["hands_meanl"]).reshape(-1, 3), np.asarray(f["hands_meanr"]).reshape(-1, 3)])
            return cls.from_arrays(f["v_template"], f["shapedirs"], f["J_regressor"], parents, num_betas, hm)

    def rest_joints(self, betas) -> np.ndarray:
        b = np.zeros(self.num_betas)
        bb = np.asarray(betas, dtype=np.float64).reshape(-1)[: self.num_betas]
        b[: len(bb)] = bb
        return self.j_template + self.j_shapedirs @ b

    def full_pose(self, global_orient, body_pose, left_hand_pose=None, right_hand_pose=None, jaw_pose=None, leye_pose=None,
                  reye_pose=None) -> np.ndarray:
        """``full_pose`` f32[N, J, 3] of the forward pass (absent parts are zero, the hands' mean pose added)."""
        go = _np(global_orient, np.float32).reshape(-1, 3)
        N = go.shape[0]
        J = self.num_joints

        def part(x, n):
            return np.zeros((N, n), np.float32) if x is None else _np(x, np.float32).reshape(N, n)
        full = np.concatenate([go, part(body_pose, 63), part(jaw_pose, 3), part(leye_pose, 3), part(reye_pose, 3),
                               part(left_hand_pose, 45), part(right_hand_pose, 45)], axis=1)[:, : 3 * J]
        return (full.reshape(N, J, 3) + self.pose_mean[None]).astype(np.float32)

    def __call__(self, betas, global_orient, body_pose, transl, left_hand_pose=None, right_hand_pose=None,
                 jaw_pose=None, leye_pose=None, reye_pose=None, return_full_pose=True, **_):
        full = self.full_pose(global_orient, body_pose, left_hand_pose, right_hand_pose, jaw_pose, leye_pose, reye_pose)
        N, J = full.shape[0], self.num_joints
        joints = _handle(self.parents).joints(self.rest_joints(_np(betas)), full, _np(transl, np.float32).reshape(N, 3))
        return SmplxOutput(global_orient=full[:, 0].copy(), full_pose=full.reshape(N, 3 * J), joints=joints)


_BODY_MODELS: Dict[tuple, "SmplxBodyModel"] = {}


def body_model_for(smplx_body_model_path, gender) -> "SmplxBodyModel":
    """The body model of (path, gender), read once per process (the reference calls ``smplx.create`` per file, :14-22;
    a dataset run opens thousands of files)."""
    key = (os.path.abspath(str(smplx_body_model_path)), str(gender))
    bm = _BODY_MODELS.get(key)
    if bm is None:
        bm = _BODY_MODELS[key] = SmplxBodyModel.from_model_path(smplx_body_model_path, gender=str(gender))
    return bm


def load_smpl_file(smpl_file):
    return np.load(smpl_file, allow_pickle=False)


def load_smplx_file(smplx_file, smplx_body_model_path):
    """Reference :12-41: returns ``(smplx_data, body_model, smplx_output, human_height)``."""
    smplx_data = np.load(smplx_file, allow_pickle=False)
    body_model = body_model_for(smplx_body_model_path, str(smplx_data["gender"]))
    smplx_output = body_model(betas=smplx_data["betas"], global_orient=smplx_data["root_orient"],
                              body_pose=smplx_data["pose_body"], transl=smplx_data["trans"])
    betas = np.asarray(smplx_data["betas"])
    human_height = 1.66 + 0.1 * (betas[0] if betas.ndim == 1 else betas[0, 0])
    return smplx_data, body_model, smplx_output, human_height


def _names(body_model) -> List[str]:
    n = len(body_model.parents)
    names = getattr(body_model, "joint_names", None) or SMPLX_JOINT_NAMES
    return list(names[:n])


def _frame_counts(smplx_data, num_frames: int, tgt_fps) -> Tuple[Optional[np.ndarray], float]:
    """Target times of the fps alignment (:118-127) and ``aligned_fps`` (:168-170)."""
    src_fps = np.asarray(smplx_data["mocap_frame_rate"]).item()
    if tgt_fps < src_fps:
        frame_skip = int(src_fps / tgt_fps)
        new_num_frames = num_frames // frame_skip
        return np.linspace(0, num_frames - 1, new_num_frames), new_num_frames / num_frames * src_fps
    return None, tgt_fps


def _poses(body_model, smplx_output):
    J = len(body_model.parents)
    full = _np(smplx_output.full_pose, np.float32)
    full = full.reshape(full.shape[0], -1, 3)[:, :J]
    joints = _np(smplx_output.joints, np.float32)
    if joints.ndim == 2:
        joints = joints[None]
    return np.ascontiguousarray(full), np.ascontiguousarray(joints)


def _to_dicts(names: Sequence[str], arr: np.ndarray) -> List[dict]:
    return [{n: (fr[i, :3].copy(), fr[i, 3:].copy()) for i, n in enumerate(names)} for fr in arr]


def get_smplx_data(smplx_data, body_model, smplx_output, curr_frame):
    """Reference :44-73: ``{joint name: (position, orientation wxyz)}`` of one source frame."""
    full, joints = _poses(body_model, smplx_output)
    arr = _handle(body_model.parents).align(full[curr_frame:curr_frame + 1], joints[curr_frame:curr_frame + 1])
    return _to_dicts(_names(body_model), arr)[0]


def get_smplx_data_offline_fast(smplx_data, body_model, smplx_output, tgt_fps=30):
    """Reference :109-197: ``(list of per-frame dicts, aligned_fps)``."""
    full, joints = _poses(body_model, smplx_output)
    tt, aligned_fps = _frame_counts(smplx_data, full.shape[0], tgt_fps)
    arr = _handle(body_model.parents).align(full, joints, tt)
    return _to_dicts(_names(body_model), arr), aligned_fps


def smplx_frames_packed(retargeter, smplx_data, body_model, smplx_output, tgt_fps=30):
    """The same frames, only the bodies ``retargeter`` uses, already in its packed layout:
    ``(human f64[T', nhuman, 7], aligned_fps)`` -- feed to ``retargeter.retarget_clip`` /
    ``retarget_streams``.  Raises ``KeyError`` for a body the body model does not have."""
    names = _names(body_model)
    idx = {n: i for i, n in enumerate(names)}
    sel = [idx[n] for n in retargeter.human_body_names]
    full, joints = _poses(body_model, smplx_output)
    tt, aligned_fps = _frame_counts(smplx_data, full.shape[0], tgt_fps)
    return _handle(body_model.parents, sel).align(full, joints, tt), aligned_fps


def slerp(q1_xyzw, q2_xyzw, t):
    """Reference :76-107 on plain quaternions (xyzw): shortest-path SLERP with the 0.9995 lerp switch;
    returns the normalised quaternion ``R.from_quat`` would hold."""
    q1 = np.asarray(q1_xyzw, dtype=np.float64); q2 = np.asarray(q2_xyzw, dtype=np.float64)
    q1 = q1 / np.linalg.norm(q1); q2 = q2 / np.linalg.norm(q2)
    dot = float(np.sum(q1 * q2))
    if dot < 0.0:
        q2, dot = -q2, -dot
    if dot > 0.9995:
        q = q1 + t * (q2 - q1)
    else:
        th0 = np.arccos(dot); th = th0 * t
        q = (np.cos(th) - dot * np.sin(th) / np.sin(th0)) * q1 + (np.sin(th) / np.sin(th0)) * q2
    return q / np.linalg.norm(q)


def smplx_frames_packed_fused(retargeter, smplx_data, body_model, tgt_fps=30):
    """:func:`smplx_frames_packed` straight from the file's arrays (``betas, root_orient, pose_body, trans,
    mocap_frame_rate``): body model and alignment in ONE library call (``gmr_smplx_frames``), the joints never leave the
    device.  Bit-identical to ``body_model(...)`` followed by :func:`smplx_frames_packed`."""
    names = _names(body_model)
    idx = {n: i for i, n in enumerate(names)}
    sel = [idx[n] for n in retargeter.human_body_names]
    full = body_model.full_pose(smplx_data["root_orient"], smplx_data["pose_body"])
    tt, aligned_fps = _frame_counts(smplx_data, full.shape[0], tgt_fps)
    frames = _handle(body_model.parents, sel).frames(body_model.rest_joints(_np(smplx_data["betas"])), full,
                                                     _np(smplx_data["trans"], np.float32).reshape(full.shape[0], 3), tt)
    return frames, aligned_fps
