"""Streaming ingest for the 120 Hz latency path: the shape adapter between a
motion-capture frame (rigid bodies ``id, pos, rot xyzw`` of one skeleton) and the hot path's inputs.  Only
the adapter: the vendor's network client (``optitrack_vendor/NatNetClient.py``) is out of scope.

Reference behaviour followed here:
* ``NatNetClient.get_frame`` (``optitrack_vendor/NatNetClient.py:2368-2383``): for every rigid body whose
  id is in the id map, ``frame[name] = [rb.pos, np.roll(rb.rot, 1)]`` (xyzw -> wxyz); unknown ids are
  reported and skipped;
* the id map (``:37-89``): ids ``1 + offset .. 51 + offset`` name the 51 bones of the FBX skeleton;
* the caller loop (``scripts/optitrack_to_robot.py:37-46``): ``GMR(src_human="fbx", tgt_robot=...,
  actual_human_height=1.6)`` then ``qpos = retarget.retarget(client.get_frame())`` per frame.

``RigidBodyPacker`` goes from the id/pos/rot arrays straight to the packed ``f64[nhuman, 7]`` frame of the
C-ABI (no dict, no per-body Python work); ``StreamingRetargeter`` is the per-frame loop on top of it.
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

# bone names of the motion-capture skeleton in id order (id = index + 1 + offset)
FBX_SKELETON_NAMES: Tuple[str, ...] = (
    "Hips", "Spine", "Spine1", "Neck", "Head",
    "LeftShoulder", "LeftArm", "LeftForeArm", "LeftHand",
    "RightShoulder", "RightArm", "RightForeArm", "RightHand",
    "LeftUpLeg", "LeftLeg", "LeftFoot", "LeftToeBase",
    "RightUpLeg", "RightLeg", "RightFoot", "RightToeBase",
) + tuple(f"{side}Hand{finger}{k}" for side in ("Left", "Right")
          for finger in ("Thumb", "Index", "Middle", "Ring", "Pinky") for k in (1, 2, 3))


def rigid_body_id_map(offset: int = 0) -> Dict[int, str]:
    """``{id_num: bone name}``; ``offset`` shifts the ids as the reference's module constant does."""
    return {i + 1 + offset: n for i, n in enumerate(FBX_SKELETON_NAMES)}


def frame_from_rigid_bodies(rigid_bodies: Iterable, id_map: Optional[Dict[int, str]] = None,
                            unknown: Optional[list] = None) -> Dict[str, list]:
    """``[(id_num, pos[3], rot_xyzw[4]), ...]`` -> ``{name: [pos, quat_wxyz]}`` (the dict ``retarget``
    takes).  Ids outside the map are skipped (appended to ``unknown`` when given)."""
    id_map = rigid_body_id_map() if id_map is None else id_map
    frame = {}
    for rb in rigid_bodies:
        id_num, pos, rot = rb
        name = id_map.get(int(id_num))
        if name is None:
            if unknown is not None:
                unknown.append(int(id_num))
            continue
        frame[name] = [pos, np.roll(rot, 1)]
    return frame


class RigidBodyPacker:
    """ids/pos/rot arrays of one skeleton -> packed ``f64[nhuman, 7]`` frame of ``retargeter``'s solver
    (rows in ``retargeter.human_body_names`` order, quaternion wxyz; a body the frame lacks gets NaN rows,
    which the kernel treats like the reference treats a missing dict entry it does not need)."""

    def __init__(self, retargeter, id_map: Optional[Dict[int, str]] = None):
        id_map = rigid_body_id_map() if id_map is None else id_map
        self.names = list(retargeter.human_body_names)
        row_of = {n: i for i, n in enumerate(self.names)}
        max_id = max(id_map) if id_map else 0
        self._row = np.full(max_id + 2, -1, dtype=np.int64)      # id -> packed row, -1 = not used
        for i, n in id_map.items():
            if i >= 0 and n in row_of:
                self._row[i] = row_of[n]
        tt = retargeter._tables
        need = {tt.human_root_name}
        for s in range(2):
            if tt.use_stage[s]:
                need.update(tt.stages[s].human_names)
        self._required = np.array(sorted(row_of[n] for n in need if n in row_of), dtype=np.int64)
        missing_in_map = [n for n in need if n not in set(id_map.values())]
        if missing_in_map:
            raise KeyError(f"id map has no id for the bodies the IK config needs: {missing_in_map}")

    def pack(self, ids: Sequence[int], pos: np.ndarray, rot_xyzw: np.ndarray) -> np.ndarray:
        ids = np.asarray(ids, dtype=np.int64)
        pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
        rot = np.asarray(rot_xyzw, dtype=np.float64).reshape(-1, 4)
        known = (ids >= 0) & (ids < len(self._row) - 1)
        rows = np.where(known, self._row[np.clip(ids, 0, len(self._row) - 1)], -1)
        sel = rows >= 0
        out = np.full((len(self.names), 7), np.nan)
        out[rows[sel], :3] = pos[sel]
        out[rows[sel], 3] = rot[sel, 3]
        out[rows[sel], 4:] = rot[sel, :3]
        if np.isnan(out[self._required, 0]).any():
            lacking = [self.names[r] for r in self._required if np.isnan(out[r, 0])]
            raise KeyError(lacking[0])                            # what retarget(dict) raises (:129,:135)
        return out


class StreamingRetargeter:
    """The loop of ``scripts/optitrack_to_robot.py:37-46`` without the dict: one packed frame per call,
    warm-started on the device-side state of ``retargeter`` exactly like ``retarget``."""

    def __init__(self, retargeter, id_map: Optional[Dict[int, str]] = None):
        self.retargeter = retargeter
        self.packer = RigidBodyPacker(retargeter, id_map)
        self.frame_number = -1

    def step(self, ids, pos, rot_xyzw, frame_number: Optional[int] = None, offset_to_ground: bool = False):
        frame = self.packer.pack(ids, pos, rot_xyzw)
        self.frame_number = self.frame_number + 1 if frame_number is None else int(frame_number)
        return self.retargeter.retarget_packed(frame, offset_to_ground)
