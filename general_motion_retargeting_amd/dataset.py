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


def retarget_clips(src_human: str, tgt_robot: str, clips: Sequence, fps: Sequence[float],
                   actual_human_height: Optional[float] = None, height_adjust: bool = True,
                   root_origin_offset: bool = True, offset_to_ground: bool = False) -> List[Dict]:
    """Retarget many clips of one (source, robot, height) in ONE IK launch.

    ``clips[i]`` is a list of ``human_data`` dicts or an array ``[T_i, nhuman, 7]`` (ragged lengths
    are fine: streams are padded and the kernel stops each stream at its own length).
    Returns one motion dict per clip, identical to processing the clips one by one.
    """
    gmr = GeneralMotionRetargeting(src_human, tgt_robot, actual_human_height=actual_human_height)
    packed = [c if isinstance(c, np.ndarray) else gmr.pack_frames(c) for c in clips]
    S = len(packed)
    if S == 0:
        return []
    lens = np.array([p.shape[0] for p in packed], dtype=np.int32)
    T = max(int(lens.max()), 1)
    nh = len(gmr.human_body_names)
    human = np.zeros((S, T, nh, 7))
    human[..., 3] = 1.0
    for i, p in enumerate(packed):
        human[i, : p.shape[0]] = p
    qpos, nsolve, status = gmr.retarget_streams(human, lens=lens, offset_to_ground=offset_to_ground)
    if (status != 0).any():
        bad = np.nonzero(status)[0].tolist()
        raise RuntimeError(f"IK failed for clips {bad}")
    km = KinematicsModel(gmr.xml_file)
    return [postprocess_clip(qpos[i, : lens[i]], km, fps[i], height_adjust, root_origin_offset) for i in range(S)]
