#!/usr/bin/env python3
"""Throughput-shape probe of the IK kernel (for rocprofv3 runs and A/B builds): S streams x T frames -> G1,
512 distinct seeded streams tiled to S.  Prints frames/s (median of `reps` launches, HIP events).

    python tools/wide_probe.py [S] [T] [waves] [reps]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, _lib, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
L = _lib.lib()
g = GeneralMotionRetargeting("smplx", "unitree_g1")
sol = g.hip_solver
nq = sol.nq
nb = min(S, 512)
base_h, base_q = synth.make_streams(g.model, g._tables, nb, T, seed=1)
r = (S + nb - 1) // nb
human = np.tile(base_h, (r, 1, 1, 1))[:S].copy()
q0 = np.tile(base_q, (r, 1))[:S].copy()
d_q0 = _lib.DeviceBuffer.from_host(q0)
d_h = _lib.DeviceBuffer.from_host(human)
d_qo = _lib.DeviceBuffer(S * T * nq * 8)
d_ns = _lib.DeviceBuffer(S * T * 8)
d_st = _lib.DeviceBuffer(S * 4)
sol.set_waves(waves)
sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st)
_lib.check(L.gmr_stream_sync(None))
ms = []
for _ in range(reps):
    a, b = _lib.Event(), _lib.Event()
    a.record(); sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st); b.record()
    ms.append(a.elapsed_ms(b))
ns = d_ns.to_host((S, T, 2), np.int32)
st = d_st.to_host((S,), np.int32)
q = d_qo.to_host((S, T, nq), np.float64)
print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "S": S, "T": T, "waves": waves, "ms": [round(x, 3) for x in ms],
                  "frames_per_s": S * T / float(np.median(ms)) * 1e3, "solves_per_frame": float(ns.sum()) / (S * T),
                  "failed_streams": int((st != 0).sum()), "q_checksum": float(np.abs(q).sum())}))
