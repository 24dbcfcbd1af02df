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


def _draw_stream(model: RobotModel, rng: np.random.Generator, T: int, nhum: int, pos_noise: float, rot_noise: float):
    """The random draws of ONE stream, in the generator's fixed order: (q*[T, nq], position noise [nhum, T, 3],
    rotation-vector noise [nhum, T, 3])."""
    q = make_trajectory(model, rng, T)
    pn = np.empty((nhum, T, 3))
    rn = np.empty((nhum, T, 3))
    for i in range(nhum):
        pn[i] = rng.normal(0.0, pos_noise, size=(T, 3))
        rn[i] = rng.normal(0.0, rot_noise, size=(T, 3))
    return q, pn, rn


def _make_chunk(args):
    """Streams with the given seeds (one per stream) -> (human[n, T, nhum, 7], truth[n, T, nq])."""
    model, tt, seeds, T, pos_noise, rot_noise = args
    names = tt.human_names
    nhum = len(names)
    # robot frame whose pose defines each human body (table 1 first, then table 2, else the root)
    frame_of = {}
    for st in (tt.stages[1], tt.stages[0]):
        for fr, hb in zip(st.frame_names, st.human_names):
            frame_of[hb] = model.body_id(fr)
    root_i = names.index(tt.human_root_name)
    scale = np.array([tt.scale_table[n] for n in names])
    body = np.array([frame_of.get(n, 0) for n in names])
    off = np.array([tt.pos_offsets1.get(n, np.zeros(3)) for n in names])
    qoc = np.array([quat_conj(tt.rot_offsets1.get(n, np.array([1.0, 0, 0, 0]))) for n in names])
    n = len(seeds)
    human = np.zeros((n, T, nhum, 7))
    q = np.empty((n, T, model.nq))
    pn = np.empty((n, nhum, T, 3))
    rn = np.empty((n, nhum, T, 3))
    for j, sd in enumerate(seeds):
        q[j], pn[j], rn[j] = _draw_stream(model, np.random.default_rng(int(sd)), T, nhum, pos_noise, rot_noise)
    xpos, xquat = fk_numpy(model, q)                                   # [n, T, nb, 3 / 4]
    tp = xpos[:, :, body] + np.moveaxis(pn, 1, 2)                        # [n, T, nhum, 3]
    tq = quat_mul(xquat[:, :, body], rotvec_quat(np.moveaxis(rn, 1, 2)))
    # invert offset_human_data: q_h = q_t * conj(q_off); p_scaled = p_t - R(q_t) off
    human[..., 3:] = quat_mul(tq, np.broadcast_to(qoc, tq.shape))
    ps = tp - quat_rotate(tq, np.broadcast_to(off, tp.shape))
    # invert scale_human_data
    raw_root = ps[:, :, root_i] / scale[root_i]
    raw = (ps - ps[:, :, root_i:root_i + 1]) / scale[None, None, :, None] + raw_root[:, :, None]
    raw[:, :, root_i] = raw_root
    human[..., :3] = raw
    return human, q


def make_streams_ids(model: RobotModel, tt: TaskTables, ids, T: int, seed: int = 0, pos_noise: float = 0.01,
                     rot_noise_deg: float = 2.0, return_truth: bool = False, chunk: int = 1024, workers: int = 1):
    """The streams ``ids`` of the batch that :func:`make_streams` generates for ``seed`` (stream i uses seed
    ``seed + i``, whatever the set it is generated in: a rank's shard holds exactly the streams of the whole batch).
    The random draws run per stream, the arithmetic (FK, inverse preprocessing) over ``chunk`` streams at a time;
    ``workers`` > 1 spreads the chunks over forked processes (call before the GPU is initialised)."""
    ids = np.asarray(ids, dtype=np.int64)
    S = len(ids)
    rot_noise = np.deg2rad(rot_noise_deg)
    jobs = [(model, tt, seed + ids[s0:s0 + chunk], T, pos_noise, rot_noise) for s0 in range(0, S, chunk)]
    if workers > 1 and len(jobs) > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:
            parts = pool.map(_make_chunk, jobs)
    else:
        parts = [_make_chunk(j) for j in jobs]
    human = np.concatenate([p[0] for p in parts]) if parts else np.zeros((0, T, len(tt.human_names), 7))
    q0 = np.broadcast_to(model.qpos0, (S, model.nq)).copy()
    if return_truth:
        truth = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, T, model.nq))
        return human, q0, truth
    return human, q0


def make_streams(model: RobotModel, tt: TaskTables, S: int, T: int, seed: int = 0,
                 pos_noise: float = 0.01, rot_noise_deg: float = 2.0, return_truth: bool = False, chunk: int = 1024,
                 workers: int = 1):
    """Synthetic batch: human f64[S,T,nhuman,7], q0 f64[S,nq] (stream s uses seed ``seed + s``)."""
    return make_streams_ids(model, tt, np.arange(S), T, seed, pos_noise, rot_noise_deg, return_truth, chunk, workers)


def streams_to_dicts(tt: TaskTables, human_stream: np.ndarray):
    """One stream f64[T,nhuman,7] -> list of reference-style ``human_data`` dicts (App. D)."""
    out = []
    for t in range(human_stream.shape[0]):
        out.append({n: (human_stream[t, i, :3].copy(), human_stream[t, i, 3:].copy())
                    for i, n in enumerate(tt.human_names)})
    return out
