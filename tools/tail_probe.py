#!/usr/bin/env python3
"""How much of a wide launch is end-of-kernel tail?  Times S streams x T frames in the given order, in descending
order of the measured solve count (an oracle longest-first order no real caller has) and in ascending order.

    python tools/tail_probe.py [S] [T] [reps]
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
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = _lib.lib()
g = GeneralMotionRetargeting("smplx", "unitree_g1")
sol = g.hip_solver
nq = sol.nq
nb = min(S, int(os.environ.get("GMR_PROBE_DISTINCT", "512")))
base_h, base_q = synth.make_streams(g.model, g._tables, nb, T, seed=1, workers=16)
r = (S + nb - 1) // nb
human = np.tile(base_h, (r, 1, 1, 1))[:S].copy()
q0 = np.tile(base_q, (r, 1))[:S].copy()
sol.set_waves(1)


def run(order):
    d_q0 = _lib.DeviceBuffer.from_host(np.ascontiguousarray(q0[order]))
    d_h = _lib.DeviceBuffer.from_host(np.ascontiguousarray(human[order]))
    d_qo = _lib.DeviceBuffer(S * T * nq * 8)
    d_ns = _lib.DeviceBuffer(S * T * 8)
    d_st = _lib.DeviceBuffer(S * 4)
    sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st)
    _lib.check(L.gmr_stream_sync(None))
    ms = []
    for _ in range(reps):
        a, b = _lib.Event(), _lib.Event()
        a.record(); sol.retarget_streams_dev(S, T, d_q0, d_h, None, 0, d_qo, d_ns, d_st); b.record()
        ms.append(a.elapsed_ms(b))
    ns = d_ns.to_host((S, T, 2), np.int32)
    return float(np.median(ms)), ns


ident = np.arange(S)
ms0, ns = run(ident)
cost = ns.reshape(S, -1).sum(1)
desc = np.argsort(-cost, kind="stable")
ms1, _ = run(desc)
ms2, _ = run(desc[::-1])
print(json.dumps({"S": S, "T": T, "solves_per_stream": {"min": int(cost.min()), "median": float(np.median(cost)), "max": int(cost.max())},
                  "given_order_ms": ms0, "longest_first_ms": ms1, "shortest_first_ms": ms2,
                  "frames_per_s": {"given": S * T / ms0 * 1e3, "longest_first": S * T / ms1 * 1e3, "shortest_first": S * T / ms2 * 1e3}}))
