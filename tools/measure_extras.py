#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the headline bench): PCIe-inclusive rate of the
host-buffer entry point, FK kernel bandwidth, per-frame latency (BASELINE.json configs[4]),
large-batch throughput per robot (configs[3] shape on one GPU)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, KinematicsModel, _lib, synth  # noqa: E402

out = {}
L = _lib.lib()
g = GeneralMotionRetargeting("smplx", "unitree_g1")
sol = g.hip_solver
nq, nh = sol.nq, sol.nhuman


def dev_time(S, T, human, q0, reps=3):
    d_q0 = _lib.DeviceBuffer.from_host(q0); d_h = _lib.DeviceBuffer.from_host(human)
    d_qo = _lib.DeviceBuffer(S * T * nq * 8); d_ns = _lib.DeviceBuffer(S * T * 8); d_st = _lib.DeviceBuffer(S * 4)
    sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st)
    _lib.check(L.gmr_stream_sync(None))
    ms = []
    for _ in range(reps):
        a, b = _lib.Event(), _lib.Event()
        a.record(); sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st); b.record()
        ms.append(a.elapsed_ms(b))
    ns = d_ns.to_host((S, T, 2), np.int32)
    return float(np.median(ms)), float(ns.sum() / (S * T))


# 1. config 2, host-buffer entry point (H2D + kernel + D2H)
human, q0 = synth.make_streams(g.model, g._tables, 100, 100, seed=0)
sol.retarget_streams(q0, human)
t = time.perf_counter(); sol.retarget_streams(q0, human); dt = time.perf_counter() - t
k_ms, spf = dev_time(100, 100, human, q0)
out["config2_S100_T100"] = {"kernel_ms": k_ms, "kernel_fps": 1e4 / k_ms * 1e3, "host_api_ms": dt * 1e3,
                            "host_api_fps_pcie_inclusive": 1e4 / dt, "solves_per_frame": spf}
# 2. width sweep on one GPU (frames/s, device-resident)
sweep = {}
for S, T in ((256, 40), (1024, 40), (4096, 40), (16384, 16)):
    human, q0 = synth.make_streams(g.model, g._tables, min(S, 512), T, seed=1)
    reps = S // human.shape[0]
    human = np.tile(human, (reps, 1, 1, 1)); q0 = np.tile(q0, (reps, 1))
    k_ms, spf = dev_time(S, T, human, q0, reps=2)
    sweep[f"S{S}_T{T}"] = {"kernel_ms": k_ms, "fps": S * T / k_ms * 1e3, "solves_per_frame": spf}
out["width_sweep_g1"] = sweep
# 3. per-frame latency through the reference API (configs[4] shape: one frame per call)
human, q0 = synth.make_streams(g.model, g._tables, 1, 300, seed=3)
frames = synth.streams_to_dicts(g._tables, human[0])
g2 = GeneralMotionRetargeting("smplx", "unitree_g1")
lat = []
for f in frames:
    t = time.perf_counter(); g2.retarget(f); lat.append(time.perf_counter() - t)
lat = np.array(lat[20:]) * 1e3
out["per_frame_latency_ms"] = {"p50": float(np.percentile(lat, 50)), "p95": float(np.percentile(lat, 95)),
                               "mean": float(lat.mean())}
# 4. FK kernel bandwidth (positions + rotations), 1M frames
km = KinematicsModel(g.xml_file)
B = 1 << 20
rng = np.random.default_rng(0)
dof = rng.uniform(-1, 1, size=(B, 29)).astype(np.float32)
rp = rng.normal(size=(B, 3)).astype(np.float32)
rq = rng.normal(size=(B, 4)).astype(np.float32); rq /= np.linalg.norm(rq, axis=1, keepdims=True)
h = km.hip_handle
d = [_lib.DeviceBuffer.from_host(a) for a in (rp, rq, dof)]
d_bp = _lib.DeviceBuffer(B * 38 * 12); d_br = _lib.DeviceBuffer(B * 38 * 16); d_mz = _lib.DeviceBuffer(4)
for want_rot in (True, False):
    h.fk_dev(B, d[0], d[1], d[2], d_bp, d_br if want_rot else None, d_mz)
    _lib.check(L.gmr_stream_sync(None))
    ms = []
    for _ in range(5):
        a, b = _lib.Event(), _lib.Event()
        a.record(); h.fk_dev(B, d[0], d[1], d[2], d_bp, d_br if want_rot else None, d_mz); b.record()
        ms.append(a.elapsed_ms(b))
    ms = float(np.median(ms))
    byt = B * (116 + 12 + 16 + 456 + (608 if want_rot else 0))
    out["fk_batch_1M_" + ("pos_rot" if want_rot else "pos")] = {"ms": ms, "GBps": byt / ms / 1e6, "frames_per_s": B / ms * 1e3}
# 5. N1: SMPL-X frame extraction kernels (120 -> 30 fps alignment of 2^20 source frames; joints-only body model)
from general_motion_retargeting_amd.utils import smpl  # noqa: E402
N = 1 << 20
par = smpl.SMPLX_PARENTS
pose = (rng.normal(0, 0.5, size=(1, 55, 3)) + 0.05 * rng.normal(size=(N, 55, 3))).astype(np.float32)
jts = rng.normal(size=(N, 55, 3)).astype(np.float32)
tt = np.linspace(0, N - 1, N // 4)
names = list(smpl.SMPLX_JOINT_NAMES)
sel = [names.index(n) for n in g.human_body_names]
closure = set()
for j in sel:
    while j >= 0:
        closure.add(j); j = int(par[j])
d_pose = _lib.DeviceBuffer.from_host(pose); d_j = _lib.DeviceBuffer.from_host(jts); d_tt = _lib.DeviceBuffer.from_host(tt)
for tag, s in (("packed_rows", sel), ("all_55_joints", None)):
    hs = _lib.SmplxHandle(par, s)
    nrow = hs.rows
    d_o = _lib.DeviceBuffer(len(tt) * nrow * 56)
    hs.align_dev(N, 55, d_pose, d_j, len(tt), d_tt, d_o)
    _lib.check(L.gmr_stream_sync(None))
    ms = []
    for _ in range(5):
        a, b = _lib.Event(), _lib.Event()
        a.record(); hs.align_dev(N, 55, d_pose, d_j, len(tt), d_tt, d_o); b.record()
        ms.append(a.elapsed_ms(b))
    ms = float(np.median(ms))
    nj = len(closure) if s is not None else 55
    byt = len(tt) * (2 * 12 * nj + 2 * 12 * nrow + 8 + 56 * nrow)
    out["smplx_align_" + tag] = {"ms": ms, "out_frames_per_s": len(tt) / ms * 1e3, "algorithmic_GBps": byt / ms / 1e6,
                                 "bytes_per_out_frame": byt // len(tt), "rows": nrow}
hs = _lib.SmplxHandle(par)
d_jr = _lib.DeviceBuffer.from_host(rng.normal(size=(55, 3))); d_tr = _lib.DeviceBuffer.from_host(rng.normal(size=(N, 3)).astype(np.float32))
hs.joints_dev(N, d_jr, d_pose, d_tr, d_j)
_lib.check(L.gmr_stream_sync(None))
ms = []
for _ in range(5):
    a, b = _lib.Event(), _lib.Event()
    a.record(); hs.joints_dev(N, d_jr, d_pose, d_tr, d_j); b.record()
    ms.append(a.elapsed_ms(b))
ms = float(np.median(ms))
out["smplx_joints_1M"] = {"ms": ms, "frames_per_s": N / ms * 1e3, "algorithmic_GBps": N * (660 + 12 + 660) / ms / 1e6}
print(json.dumps(out, indent=1))
