#!/usr/bin/env python3
"""Launch-shape crossover: frames/s of the IK kernel with 1 and 4 wavefronts per stream over the number
of streams per launch (sets GMR_HELPER_MAX_STREAMS in gmr_abi.hip)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, _lib  # noqa: E402
from general_motion_retargeting_amd import synth  # noqa: E402

L = _lib.lib()
g = GeneralMotionRetargeting("smplx", sys.argv[1] if len(sys.argv) > 1 else "unitree_g1")
sol = g.hip_solver
nq = sol.nq
T = 40
base_h, base_q = synth.make_streams(g.model, g._tables, 512, T, seed=1)
out = {}
for S in (64, 128, 256, 384, 512, 768, 1024, 1536, 2048, 4096):
    reps = (S + 511) // 512
    human = np.tile(base_h, (reps, 1, 1, 1))[:S].copy(); q0 = np.tile(base_q, (reps, 1))[:S].copy()
    d_q0 = _lib.DeviceBuffer.from_host(q0); d_h = _lib.DeviceBuffer.from_host(human)
    d_qo = _lib.DeviceBuffer(S * T * nq * 8); d_ns = _lib.DeviceBuffer(S * T * 8); d_st = _lib.DeviceBuffer(S * 4)
    row = {}
    for nw in (1, 4):
        sol.set_waves(nw)
        sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st)
        _lib.check(L.gmr_stream_sync(None))
        ms = []
        for _ in range(3):
            a, b = _lib.Event(), _lib.Event()
            a.record(); sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st); b.record()
            ms.append(a.elapsed_ms(b))
        row[f"nw{nw}_fps"] = S * T / float(np.median(ms)) * 1e3
    out[f"S{S}"] = row
    print(S, {k: round(v) for k, v in row.items()}, flush=True)
sol.set_waves(0)
print(json.dumps(out))
