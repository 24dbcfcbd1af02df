#!/usr/bin/env python3
"""Edge cases of the queued dispatch (no hang, same bits as one workgroup per stream): one stream more than the resident
wavefronts, all-empty streams, T = chunk + 1, a chunk longer than T (direct), many tiny streams, a few very long ones."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, synth  # noqa: E402

g = GeneralMotionRetargeting("smplx", "unitree_g1")
sol = g.hip_solver
sol.set_waves(1)
bh, bq = synth.make_streams(g.model, g._tables, 64, 40, seed=4)
rng = np.random.default_rng(1)
out = {}
cases = {"S=2049,T=3,chunk=2": (2049, 3, 2, None), "S=2049,T=3,chunk=5 (direct)": (2049, 3, 5, None),
         "S=5000,T=1": (5000, 1, 1, None), "all empty": (3000, 4, 2, "zeros"), "mostly empty": (4096, 6, 2, "sparse"),
         "few long among short": (2500, 40, 3, "skew"), "S=2048 (not queued)": (2048, 5, 2, None)}
for name, (S, T, chunk, mode) in cases.items():
    pick = rng.integers(0, 64, size=S)
    human, q0 = np.ascontiguousarray(bh[pick][:, :T]), np.ascontiguousarray(bq[pick])
    lens = None
    if mode == "zeros":
        lens = np.zeros(S, np.int32)
    elif mode == "sparse":
        lens = (rng.random(S) < 0.02).astype(np.int32) * T
    elif mode == "skew":
        lens = np.where(rng.random(S) < 0.01, T, 1).astype(np.int32)
    sol.set_dispatch(0)
    ref = sol.retarget_streams(q0, human, lens=lens)
    sol.set_dispatch(chunk)
    got = sol.retarget_streams(q0, human, lens=lens)
    out[name] = bool(all(np.array_equal(a, b) for a, b in zip(ref, got)) and (ref[2] == 0).all())
print(json.dumps(out, indent=1))
assert all(out.values())
