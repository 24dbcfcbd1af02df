#!/usr/bin/env python3
"""The kernels beside the IK loop, alone (for rocprofv3 runs): post-hoc FK (positions / positions + rotations) and the
SMPL-X frame-extraction kernels (fps alignment to packed rows; joints-only body model), 2^20 frames each.

    python tools/aux_kernels.py [reps]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, KinematicsModel, ROBOT_XML_DICT, _lib  # noqa: E402
from general_motion_retargeting_amd.utils import smpl  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
L = _lib.lib()
out = {}


def timed(fn):
    fn()
    _lib.check(L.gmr_stream_sync(None))
    ms = []
    for _ in range(reps):
        a, b = _lib.Event(), _lib.Event()
        a.record(); fn(); b.record()
        ms.append(a.elapsed_ms(b))
    return float(np.median(ms))


km = KinematicsModel(ROBOT_XML_DICT["unitree_g1"])
h = km.hip_handle
B = 1 << 20
rng = np.random.default_rng(0)
dof = rng.uniform(-1, 1, size=(B, 29)).astype(np.float32)
rp = rng.normal(size=(B, 3)).astype(np.float32)
rq = rng.normal(size=(B, 4)).astype(np.float32)
rq /= np.linalg.norm(rq, axis=1, keepdims=True)
d = [_lib.DeviceBuffer.from_host(a) for a in (rp, rq, dof)]
d_bp = _lib.DeviceBuffer(B * 38 * 12)
d_br = _lib.DeviceBuffer(B * 38 * 16)
d_mz = _lib.DeviceBuffer(4)
for want_rot in (False, True):
    ms = timed(lambda: h.fk_dev(B, d[0], d[1], d[2], d_bp, d_br if want_rot else None, d_mz))
    byt = B * (116 + 12 + 16 + 456 + (608 if want_rot else 0))
    out["fk_batch_" + ("pos_rot" if want_rot else "pos")] = {"frames": B, "ms": ms, "algorithmic_bytes": byt, "GBps": byt / ms / 1e6,
                                                          "frac_of_8TBps": byt / ms / 1e6 / 8000.0}
g = GeneralMotionRetargeting("smplx", "unitree_g1")
N = B
par = smpl.SMPLX_PARENTS
pose = (rng.normal(0, 0.5, size=(1, 55, 3)) + 0.05 * rng.normal(size=(N, 55, 3))).astype(np.float32)
jts = rng.normal(size=(N, 55, 3)).astype(np.float32)
tt = np.linspace(0, N - 1, N // 4)
names = list(smpl.SMPLX_JOINT_NAMES)
sel = [names.index(n) for n in g.human_body_names]
closure = set()
for j in sel:
    while j >= 0:
        closure.add(j)
        j = int(par[j])
d_pose = _lib.DeviceBuffer.from_host(pose)
d_j = _lib.DeviceBuffer.from_host(jts)
d_tt = _lib.DeviceBuffer.from_host(tt)
hs = _lib.SmplxHandle(par, sel)
d_o = _lib.DeviceBuffer(len(tt) * hs.rows * 56)
ms = timed(lambda: hs.align_dev(N, 55, d_pose, d_j, len(tt), d_tt, d_o))
byt = len(tt) * (2 * 12 * len(closure) + 2 * 12 * hs.rows + 8 + 56 * hs.rows)
out["smplx_align_packed_rows"] = {"out_frames": len(tt), "ms": ms, "algorithmic_bytes": byt, "GBps": byt / ms / 1e6,
                                  "frac_of_8TBps": byt / ms / 1e6 / 8000.0}
# the same alignment on COMPACT inputs (what the host entry point uploads): only the closure's poses, only the rows' joints
pj, rj = hs.compact_layout()
d_pc = _lib.DeviceBuffer.from_host(np.ascontiguousarray(pose[:, pj].transpose(1, 2, 0)))
d_jc = _lib.DeviceBuffer.from_host(np.ascontiguousarray(jts[:, rj].transpose(1, 2, 0)))
d_o2 = _lib.DeviceBuffer(len(tt) * hs.rows * 56)
ms = timed(lambda: hs.align_compact_dev(N, d_pc, d_jc, len(tt), d_tt, d_o2))
same = bool(np.array_equal(d_o.to_host((len(tt), hs.rows, 7), np.float64), d_o2.to_host((len(tt), hs.rows, 7), np.float64)))
out["smplx_align_packed_rows_compact_inputs"] = {"out_frames": len(tt), "ms": ms, "algorithmic_bytes": byt, "GBps": byt / ms / 1e6,
                                                 "frac_of_8TBps": byt / ms / 1e6 / 8000.0, "bit_identical_to_full_rows": same}
hj = _lib.SmplxHandle(par)
d_jr = _lib.DeviceBuffer.from_host(rng.normal(size=(55, 3)))
d_tr = _lib.DeviceBuffer.from_host(rng.normal(size=(N, 3)).astype(np.float32))
ms = timed(lambda: hj.joints_dev(N, d_jr, d_pose, d_tr, d_j))
byt = N * (660 + 12 + 660)
out["smplx_joints"] = {"frames": N, "ms": ms, "algorithmic_bytes": byt, "GBps": byt / ms / 1e6, "frac_of_8TBps": byt / ms / 1e6 / 8000.0}
print(json.dumps(out, indent=1))
