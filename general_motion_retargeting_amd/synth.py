"""Seeded synthetic human-motion streams (no motion data ships with the reference).

Generator of SURVEY.md section 8(d): per stream a smooth, reachable robot trajectory ``q*(t)`` is
drawn, the poses of the task frames are computed with a NumPy FK, GMR's scale/offset preprocessing
(reference ``motion_retarget.py:209-250``) is *inverted* to obtain raw ``human_data`` for the
bodies the ik_config consumes, and 1 cm / 2 deg noise is added so that targets are near- but not
exactly reachable.  Output layout is the packed batch layout of the C-ABI:
``human f64[S, T, nhuman, 7]`` (pos xyz, quat wxyz) and ``q0 f64[S, nq] = qpos0``.
"""
from __future__ import annotations

import numpy as np

from .ik_config import TaskTables
from .mjcf import RobotModel


def quat_mul(a, b):
    w1, x1, y1, z1 = np.moveaxis(a, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(b, -1, 0)
    return np.stack([
        w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
    ], axis=-1)


def quat_conj(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0])


def quat_rotate(q, v):
    qv = np.concatenate([np.zeros(v.shape[:-1] + (1,)), v], axis=-1)
    return quat_mul(quat_mul(q, qv), quat_conj(q))[..., 1:]


def axis_angle_quat(axis, angle):
    h = 0.5 * angle
    return np.concatenate([np.cos(h)[..., None], axis * np.sin(h)[..., None]], axis=-1)


def rotvec_quat(rv):
    ang = np.linalg.norm(rv, axis=-1)
    axis = rv / np.maximum(ang, 1e-300)[..., None]
    return axis_angle_quat(axis, ang)


def fk_numpy(model: RobotModel, q: np.ndarray):
    """MuJoCo-semantics FK, vectorised over leading dims: q[..., nq] -> (xpos[..., nb, 3], xquat[..., nb, 4])."""
    lead = q.shape[:-1]
    nb = model.nbody
    xpos = np.zeros(lead + (nb, 3))
    xquat = np.zeros(lead + (nb, 4))
    xpos[..., 0, :] = q[..., 0:3]
    rq = q[..., 3:7]
    xquat[..., 0, :] = rq / np.linalg.norm(rq, axis=-1, keepdims=True)
    for b in range(1, nb):
        p = int(model.parent[b])
        xpos[..., b, :] = xpos[..., p, :] + quat_rotate(xquat[..., p, :], np.broadcast_to(model.body_pos[b], lead + (3,)))
        qb = quat_mul(xquat[..., p, :], np.broadcast_to(model.body_quat[b], lead + (4,)))
        h = int(model.body_hinge[b])
        if h >= 0:
            qb = quat_mul(qb, axis_angle_quat(np.broadcast_to(model.hinge_axis[h], lead + (3,)), q[..., 7 + h]))
        xquat[..., b, :] = qb / np.linalg.norm(qb, axis=-1, keepdims=True)
    return xpos, xquat


def make_trajectory(model: RobotModel, rng: np.random.Generator, T: int, fps: float = 30.0) -> np.ndarray:
    """Smooth reachable q*(t), f64[T, nq]."""
    t = np.arange(T) / fps
    nh = model.nhinge
    q = np.zeros((T, model.nq))
    lo = np.where(model.limited > 0, model.range_lo, -1.0)
    hi = np.where(model.limited > 0, model.range_hi, 1.0)
    mid, half = 0.5 * (lo + hi), (hi - lo)
    f = rng.uniform(0.2, 1.5, size=(nh, 3))
    a = rng.dirichlet(np.ones(3), size=nh)
    ph = rng.uniform(0.0, 2 * np.pi, size=(nh, 3))
    wave = (a[None] * np.sin(2 * np.pi * f[None] * t[:, None, None] + ph[None])).sum(-1)
    q[:, 7:] = np.clip(mid[None] + 0.35 * half[None] * wave, lo[None], hi[None])
    # root: planar walk with heading drift, small roll/pitch
    v = rng.uniform(0.0, 1.2)
    yaw0 = rng.uniform(-np.pi, np.pi)
    yaw = yaw0 + 0.3 * np.sin(2 * np.pi * rng.uniform(0.05, 0.3) * t + rng.uniform(0, 2 * np.pi))
    dt = 1.0 / fps
    q[:, 0] = np.cumsum(v * np.cos(yaw) * dt)
    q[:, 1] = np.cumsum(v * np.sin(yaw) * dt)
    q[:, 2] = model.qpos0[2] + 0.05 * np.sin(2 * np.pi * rng.uniform(0.3, 1.0) * t + rng.uniform(0, 2 * np.pi))
    roll = 0.2 * np.sin(2 * np.pi * rng.uniform(0.2, 0.8) * t + rng.uniform(0, 2 * np.pi))
    pitch = 0.2 * np.sin(2 * np.pi * rng.uniform(0.2, 0.8) * t + rng.uniform(0, 2 * np.pi))
    z = np.zeros_like(t)
    qy = axis_angle_quat(np.stack([z, z, z + 1], -1), yaw)
    qp = axis_angle_quat(np.stack([z, z + 1, z], -1), pitch)
    qr = axis_angle_quat(np.stack([z + 1, z, z], -1), roll)
    q[:, 3:7] = quat_mul(quat_mul(qy, qp), qr)
    return q


def make_streams(model: RobotModel, tt: TaskTables, S: int, T: int, seed: int = 0,
                 pos_noise: float = 0.01, rot_noise_deg: float = 2.0, return_truth: bool = False):
    """Synthetic batch: human f64[S,T,nhuman,7], q0 f64[S,nq] (stream s uses seed ``seed + s``)."""
    names = tt.human_names
    nhum = len(names)
    # robot frame whose pose defines each human body (table 1 first, then table 2, else the root)
    frame_of = {}
    for st in (tt.stages[1], tt.stages[0]):
        for fr, hb in zip(st.frame_names, st.human_names):
            frame_of[hb] = model.body_id(fr)
    root_i = names.index(tt.human_root_name)
    human = np.zeros((S, T, nhum, 7))
    truth = np.zeros((S, T, model.nq))
    scale = np.array([tt.scale_table[n] for n in names])
    for s in range(S):
        rng = np.random.default_rng(seed + s)
        q = make_trajectory(model, rng, T)
        truth[s] = q
        xpos, xquat = fk_numpy(model, q)
        tp = np.zeros((T, nhum, 3))
        tq = np.zeros((T, nhum, 4))
        for i, n in enumerate(names):
            b = frame_of.get(n, 0)
            tp[:, i] = xpos[:, b] + rng.normal(0.0, pos_noise, size=(T, 3))
            nq_ = rotvec_quat(rng.normal(0.0, np.deg2rad(rot_noise_deg), size=(T, 3)))
            tq[:, i] = quat_mul(xquat[:, b], nq_)
        # invert offset_human_data: q_h = q_t * conj(q_off); p_scaled = p_t - R(q_t) off
        ps = np.zeros_like(tp)
        for i, n in enumerate(names):
            off = tt.pos_offsets1.get(n, np.zeros(3))
            qo = tt.rot_offsets1.get(n, np.array([1.0, 0, 0, 0]))
            human[s, :, i, 3:] = quat_mul(tq[:, i], np.broadcast_to(quat_conj(qo), (T, 4)))
            ps[:, i] = tp[:, i] - quat_rotate(tq[:, i], np.broadcast_to(off, (T, 3)))
        # invert scale_human_data
        raw_root = ps[:, root_i] / scale[root_i]
        for i in range(nhum):
            if i == root_i:
                human[s, :, i, :3] = raw_root
            else:
                human[s, :, i, :3] = (ps[:, i] - ps[:, root_i]) / scale[i] + raw_root
    q0 = np.broadcast_to(model.qpos0, (S, model.nq)).copy()
    if return_truth:
        return human, q0, truth
    return human, q0


def streams_to_dicts(tt: TaskTables, human_stream: np.ndarray):
    """One stream f64[T,nhuman,7] -> list of reference-style ``human_data`` dicts (App. D)."""
    out = []
    for t in range(human_stream.shape[0]):
        out.append({n: (human_stream[t, i, :3].copy(), human_stream[t, i, 3:].copy())
                    for i, n in enumerate(tt.human_names)})
    return out
