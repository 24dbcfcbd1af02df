#!/usr/bin/env python3
"""FK kernel alone (for rocprofv3 runs): 2^20 frames, G1; argv[1] = 'pos' | 'posrot'."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import KinematicsModel, ROBOT_XML_DICT, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "pos"
L = _lib.lib()
km = KinematicsModel(ROBOT_XML_DICT["unitree_g1"]); h = km.hip_handle
B = 1 << 20
rng = np.random.default_rng(0)
amp = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
dof = rng.uniform(-amp, amp, size=(B, 29)).astype(np.float32)
rp = rng.normal(size=(B, 3)).astype(np.float32)
rq = rng.normal(size=(B, 4)).astype(np.float32); rq /= np.linalg.norm(rq, axis=1, keepdims=True)
d = [_lib.DeviceBuffer.from_host(a) for a in (rp, rq, dof)]
d_bp = _lib.DeviceBuffer(B * 38 * 12); d_br = _lib.DeviceBuffer(B * 38 * 16); d_mz = _lib.DeviceBuffer(4)
ms = []
for i in range(6):
    a, b = _lib.Event(), _lib.Event()
    a.record(); h.fk_dev(B, d[0], d[1], d[2], d_bp, d_br if mode == "posrot" else None, d_mz); b.record()
    ms.append(a.elapsed_ms(b))
print(mode, "ms", np.round(ms, 3))
