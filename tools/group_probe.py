#!/usr/bin/env python3
"""Cost of the group instance of the throughput kernel (job fields from a table in device memory instead of kernel
arguments): the same S x T G1 batch as ONE job (plain instance) and as TWO jobs of S/2 streams on the same solver
(group instance, one scheduling domain).  Bit-identical by construction; prints both rates.

    python tools/group_probe.py [S] [T] [reps]
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
nq, nh = sol.nq, sol.nhuman
nb = min(S, 512)
base_h, base_q = synth.make_streams(g.model, g._tables, nb, T, seed=1)
r = (S + nb - 1) // nb
human = np.tile(base_h, (r, 1, 1, 1))[:S].copy()
q0 = np.tile(base_q, (r, 1))[:S].copy()
d_q0, d_h = _lib.DeviceBuffer.from_host(q0), _lib.DeviceBuffer.from_host(human)
d_qo, d_ns, d_st = _lib.DeviceBuffer(S * T * nq * 8), _lib.DeviceBuffer(S * T * 8), _lib.DeviceBuffer(S * 4)
H = S // 2


def off(buf, nbytes):
    import ctypes as C
    return C.c_void_p(buf.ptr.value + nbytes)


one = [(sol, S, T, d_q0, d_h, None, d_qo, d_ns, d_st)]
two = [(sol, H, T, d_q0, d_h, None, d_qo, d_ns, d_st),
       (sol, S - H, T, off(d_q0, H * nq * 8), off(d_h, H * T * nh * 56), None, off(d_qo, H * T * nq * 8), off(d_ns, H * T * 8),
        off(d_st, H * 4))]
out = {"S": S, "T": T}
qs = {}
for name, jobs in (("one_job_plain_instance", one), ("two_jobs_group_instance", two)):
    _lib.retarget_group_dev(jobs, 0, None)
    _lib.check(L.gmr_stream_sync(None))
    ms = []
    for _ in range(reps):
        a, b = _lib.Event(), _lib.Event()
        a.record(); _lib.retarget_group_dev(jobs, 0, None); b.record()
        ms.append(a.elapsed_ms(b))
    qs[name] = d_qo.to_host((S, T, nq), np.float64)
    out[name] = {"ms": [round(x, 3) for x in ms], "frames_per_s": S * T / float(np.median(ms)) * 1e3}
out["bit_identical"] = bool(np.array_equal(qs["one_job_plain_instance"], qs["two_jobs_group_instance"]))
out["group_over_plain"] = out["two_jobs_group_instance"]["frames_per_s"] / out["one_job_plain_instance"]["frames_per_s"]
print(json.dumps(out))
