"""LAFAN1 / BVH source adapter: BVH file -> ``human_data`` frames.

Same results as the reference's ``load_lafan1_file`` (``general_motion_retargeting/utils/lafan1.py:8-41``
on top of ``utils/lafan_vendor/extract.py:43-166`` and ``utils/lafan_vendor/utils.py:42-162,251-268``),
restated with vectorised NumPy: BVH text -> Euler channels -> local quaternions (with the sign
de-flipping along time) -> global poses by FK over the skeleton -> Y-up centimetres to Z-up metres ->
per-frame dict ``{bone: (pos, quat_wxyz)}`` plus the two synthetic bodies ``LeftFootMod`` /
``RightFootMod`` (foot position with the toe's orientation).  Pinned by ``tests/golden/g_bvh.npz``
(the reference's own loader run on ``tests/golden/synthetic.bvh``).

:func:`load_lafan1_packed` additionally returns the frames already in the packed
``[T, nhuman, 7]`` layout of the C-ABI for a given body order (no per-frame dicts).
"""
from __future__ import annotations

import re
from typing import Dict, List, Sequence, Tuple

import numpy as np

_CHANNEL_AXIS = {"Xrotation": "x", "Yrotation": "y", "Zrotation": "z"}
_AXIS_VEC = {"x": np.array([1.0, 0.0, 0.0]), "y": np.array([0.0, 1.0, 0.0]), "z": np.array([0.0, 0.0, 1.0])}
# Y-up -> Z-up (lafan1.py:20): rows of the rotation matrix
_ROT = np.array([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 1.0, 0.0]])
_ROT_QUAT = np.array([np.sqrt(0.5), np.sqrt(0.5), 0.0, 0.0])   # R.from_matrix(_ROT).as_quat(scalar_first=True)


def _quat_mul(x, y):
    x0, x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    y0, y1, y2, y3 = y[..., 0], y[..., 1], y[..., 2], y[..., 3]
    return np.stack([y0 * x0 - y1 * x1 - y2 * x2 - y3 * x3,
                     y0 * x1 + y1 * x0 - y2 * x3 + y3 * x2,
                     y0 * x2 + y1 * x3 + y2 * x0 - y3 * x1,
                     y0 * x3 - y1 * x2 + y2 * x1 + y3 * x0], axis=-1)


def _quat_mul_vec(q, x):
    t = 2.0 * np.cross(q[..., 1:], x)
    return x + q[..., 0:1] * t + np.cross(q[..., 1:], t)


def _angle_axis_to_quat(angle, axis):
    c = np.cos(angle / 2.0)[..., None]
    s = np.sin(angle / 2.0)[..., None]
    return np.concatenate([c, s * axis], axis=-1)


class Bvh:
    """Parsed BVH: names, parents, offsets, per-frame local positions and Euler angles (degrees)."""

    def __init__(self, names, parents, offsets, positions, eulers, order, frametime):
        self.bones: List[str] = names
        self.parents = parents
        self.offsets = offsets
        self.pos = positions
        self.eulers = eulers
        self.order = order
        self.frametime = frametime
        self.quats = None


def read_bvh(filename) -> Bvh:
    """BVH text -> :class:`Bvh` (hierarchy walk and channel layouts of extract.py:43-166)."""
    names: List[str] = []
    offsets: List[List[float]] = []
    parents: List[int] = []
    active = -1
    end_site = False
    order = None
    channels = None
    frametime = None
    nframes = None
    rows: List[np.ndarray] = []
    with open(filename, "r") as f:
        lines = f.readlines()
    motion_at = None
    for li, line in enumerate(lines):
        if "HIERARCHY" in line or "MOTION" in line or "{" in line:
            continue
        m = re.match(r"\s*(ROOT|JOINT)\s+(\w+)", line)
        if m:
            names.append(m.group(2))
            offsets.append([0.0, 0.0, 0.0])
            parents.append(active)
            active = len(parents) - 1
            continue
        if "}" in line:
            if end_site:
                end_site = False
            else:
                active = parents[active]
            continue
        m = re.match(r"\s*OFFSET\s+([\-\d\.e]+)\s+([\-\d\.e]+)\s+([\-\d\.e]+)", line)
        if m:
            if not end_site:
                offsets[active] = [float(v) for v in m.groups()]
            continue
        m = re.match(r"\s*CHANNELS\s+(\d+)", line)
        if m:
            channels = int(m.group(1))
            if order is None:
                lo, hi = (0, 3) if channels == 3 else (3, 6)
                parts = line.split()[2 + lo: 2 + hi]
                if all(p in _CHANNEL_AXIS for p in parts):
                    order = "".join(_CHANNEL_AXIS[p] for p in parts)
            continue
        if "End Site" in line:
            end_site = True
            continue
        m = re.match(r"\s*Frames:\s+(\d+)", line)
        if m:
            nframes = int(m.group(1))
            continue
        m = re.match(r"\s*Frame Time:\s+([\d\.]+)", line)
        if m:
            frametime = float(m.group(1))
            if nframes is not None and frametime is not None:
                motion_at = li + 1      # everything below is the motion block: parsed in one go, not line by line
                break
            continue
        vals = line.strip().split(" ")
        if vals and vals != [""]:
            rows.append(np.array([float(v) for v in vals]))
    if motion_at is not None:
        # The motion block: one row of numbers per frame.  Parsed with one C call instead of a Python loop with five
        # regular-expression attempts per line (the loader's cost is what bounds the dataset driver: tools/dataset_probe.py);
        # the same correctly-rounded doubles as float().
        body = [ln for ln in lines[motion_at:] if ln.strip()]
        if body:
            ncol = len(body[0].split())
            flat = np.fromstring("".join(body), dtype=np.float64, sep=" ")
            if ncol == 0 or flat.size % ncol != 0:
                raise ValueError(f"{filename}: ragged motion block")
            rows = list(flat.reshape(-1, ncol))
    n = len(parents)
    off = np.array(offsets, dtype=np.float64).reshape(n, 3)
    data = np.stack(rows[:nframes]) if rows else np.zeros((0, 3 + 3 * n))
    T = data.shape[0]
    positions = np.repeat(off[None], T, axis=0)
    eulers = np.zeros((T, n, 3))
    if channels == 3:
        positions[:, 0] = data[:, 0:3]
        eulers[:] = data[:, 3:].reshape(T, n, 3)
    elif channels == 6:
        blk = data.reshape(T, n, 6)
        positions[:] = blk[..., 0:3]
        eulers[:] = blk[..., 3:6]
    elif channels == 9:
        positions[:, 0] = data[:, 0:3]
        blk = data[:, 3:].reshape(T, n - 1, 9)
        eulers[:, 1:] = blk[..., 3:6]
        positions[:, 1:] += blk[..., 0:3] * blk[..., 6:9]
    else:
        raise Exception("Too many channels! %s" % channels)
    bvh = Bvh(names, np.array(parents, dtype=int), off, positions, eulers, order, frametime)
    bvh.quats = remove_quat_discontinuities(euler_to_quat(np.radians(eulers), order))
    return bvh


def euler_to_quat(e, order="zyx"):
    q0 = _angle_axis_to_quat(e[..., 0], _AXIS_VEC[order[0]])
    q1 = _angle_axis_to_quat(e[..., 1], _AXIS_VEC[order[1]])
    q2 = _angle_axis_to_quat(e[..., 2], _AXIS_VEC[order[2]])
    return _quat_mul(q0, _quat_mul(q1, q2))


def remove_quat_discontinuities(rotations):
    """Flip q_t to -q_t whenever that is closer to the (already processed) q_{t-1}: the flip state is
    the running product of sign(<q_{t-1}, q_t>) over time (utils.py:251-268), vectorised."""
    if rotations.shape[0] < 2:
        return rotations
    dots = np.sum(rotations[:-1] * rotations[1:], axis=-1)
    step = np.where(dots < -dots, -1.0, 1.0)               # strict: ties keep the sign
    sign = np.concatenate([np.ones((1,) + step.shape[1:]), np.cumprod(step, axis=0)], axis=0)
    return rotations * sign[..., None]


def quat_fk(lrot, lpos, parents):
    """Global (rot, pos) from local ones, parents before children (utils.py:88-103).  The joints of one tree depth are
    composed in one vectorised step (the same operations per joint as the reference's joint-by-joint loop: same bits)."""
    J = len(parents)
    gr = np.empty_like(lrot)
    gp = np.empty_like(lpos)
    gr[..., 0, :] = lrot[..., 0, :]
    gp[..., 0, :] = lpos[..., 0, :]
    depth = np.zeros(J, dtype=int)
    for i in range(1, J):
        depth[i] = depth[int(parents[i])] + 1
    for d in range(1, int(depth.max()) + 1 if J > 1 else 1):
        idx = np.nonzero(depth == d)[0]
        par = np.asarray(parents)[idx].astype(int)
        gp[..., idx, :] = _quat_mul_vec(gr[..., par, :], lpos[..., idx, :]) + gp[..., par, :]
        gr[..., idx, :] = _quat_mul(gr[..., par, :], lrot[..., idx, :])
    return gr, gp


def _global_poses(bvh_file) -> Tuple[List[str], np.ndarray, np.ndarray]:
    data = read_bvh(bvh_file)
    grot, gpos = quat_fk(data.quats, data.pos, data.parents)
    orient = _quat_mul(np.broadcast_to(_ROT_QUAT, grot.shape), grot)
    position = gpos @ _ROT.T / 100                                           # cm -> m
    names = list(data.bones) + ["LeftFootMod", "RightFootMod"]
    li, lt = data.bones.index("LeftFoot"), data.bones.index("LeftToe")
    ri, rt = data.bones.index("RightFoot"), data.bones.index("RightToe")
    position = np.concatenate([position, position[:, [li, ri]]], axis=1)
    orient = np.concatenate([orient, orient[:, [lt, rt]]], axis=1)
    return names, position, orient


def load_lafan1_file(bvh_file):
    """``(frames, human_height)`` exactly like the reference: a list of per-frame dicts
    ``{bone: (position, orientation_wxyz)}`` and the hard-coded height 1.75 (lafan1.py:37-39)."""
    names, position, orient = _global_poses(bvh_file)
    frames = [{n: (position[t, i], orient[t, i]) for i, n in enumerate(names)} for t in range(position.shape[0])]
    return frames, 1.75


def load_lafan1_packed(bvh_file, body_names: Sequence[str]) -> Tuple[np.ndarray, float]:
    """``(human f64[T, len(body_names), 7], human_height)``: the same data in the packed layout of
    ``gmr_retarget_streams`` (pos xyz, quat wxyz), for ``GeneralMotionRetargeting.human_body_names``."""
    names, position, orient = _global_poses(bvh_file)
    idx = [names.index(n) for n in body_names]          # ValueError for an unknown body
    return np.concatenate([position[:, idx], orient[:, idx]], axis=-1), 1.75
