#!/usr/bin/env python3
"""One-frame calls of gmr_retarget_streams (S = T = 1, the 120 Hz per-frame API without the Python class around it):
wall time per call, and how many solves the frame took.   python tools/frame_call_probe.py [calls]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
g = GeneralMotionRetargeting("smplx", "unitree_g1")
human, q0 = synth.make_streams(g.model, g._tables, 1, 300, seed=4)
sol = g.hip_solver
q = q0.copy()
ts, ns = [], []
for t in range(n):
    f = human[:, t % 300][:, None]
    t0 = time.perf_counter()
    qo, nsv, st, tg, er = sol.retarget_streams(q, f, want_targets=True, want_errors=True)
    ts.append(time.perf_counter() - t0)
    ns.append(int(nsv.sum()))
    q = qo[:, -1]
ts = np.array(ts[50:]) * 1e3
print(json.dumps({"calls": n, "p50_ms": float(np.percentile(ts, 50)), "p95_ms": float(np.percentile(ts, 95)),
                  "mean_solves_per_frame": float(np.mean(ns[50:])), "zero_copy": "GMR_NO_ZERO_COPY" not in os.environ}))
