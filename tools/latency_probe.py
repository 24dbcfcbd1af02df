#!/usr/bin/env python3
"""BASELINE.json configs[4] / SURVEY.md section 8(d) "Config 5": one stream, frames delivered one at a time
at 120 Hz from a host timer, ``fbx_to_g1.json`` (51-body skeleton, 14 bodies used).  Reports p50 / p95
per-frame milliseconds of (i) retarget(dict) through the reference-shaped API, (ii) the streaming adapter
(ids/pos/rot arrays -> packed frame -> kernel, utils/optitrack.py), (iii) the CPU oracle per frame
(measurement tool: like bench.py's cpu_baseline leg it may time the oracle)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, synth  # noqa: E402
from general_motion_retargeting_amd.utils.optitrack import FBX_SKELETON_NAMES, StreamingRetargeter, rigid_body_id_map  # noqa: E402

HZ, NFRAMES, WARM = 120.0, 360, 30
g = GeneralMotionRetargeting("fbx", "unitree_g1", actual_human_height=1.6)
human, q0 = synth.make_streams(g.model, g._tables, 1, NFRAMES, seed=3)
names = g.human_body_names
rng = np.random.default_rng(0)
id_of = {n: i for i, n in rigid_body_id_map().items()}
used = np.array([id_of[n] for n in names])
others = np.array([i for i in range(1, 52) if i not in set(used)])


def mocap_frame(t):
    """all 51 rigid bodies, the unused ones with arbitrary poses, in arrival order"""
    fr = human[0, t]
    ids = np.concatenate([used, others])
    pos = np.vstack([fr[:, :3], rng.normal(size=(len(others), 3))])
    rot = np.vstack([fr[:, [4, 5, 6, 3]], np.tile([0, 0, 0, 1.0], (len(others), 1))])
    return ids, pos, rot


def paced(fn):
    lat = []
    t_next = time.perf_counter()
    for t in range(NFRAMES):
        while time.perf_counter() < t_next:
            pass
        t_next += 1.0 / HZ
        a = time.perf_counter(); fn(t); lat.append(time.perf_counter() - a)
    lat = np.array(lat[WARM:]) * 1e3
    return {"p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)), "max_ms": float(lat.max())}


out = {"rate_hz": HZ, "frames": NFRAMES - WARM, "config": "fbx -> unitree_g1, 51-body frames"}
frames = []
for t in range(NFRAMES):
    ids, pos, rot = mocap_frame(t)
    frames.append({rigid_body_id_map()[i]: [p, np.roll(r, 1)] for i, p, r in zip(ids, pos, rot)})
out["hip_retarget_dict"] = paced(lambda t: g.retarget(frames[t]))
g2 = GeneralMotionRetargeting("fbx", "unitree_g1", actual_human_height=1.6)
st = StreamingRetargeter(g2)
mf = [mocap_frame(t) for t in range(NFRAMES)]
out["hip_streaming_adapter"] = paced(lambda t: st.step(*mf[t]))
try:
    from oracle import oracle as orc
    orc.build()
    state = {"q": q0[0].copy()}

    def cpu_step(t):
        q, ns, stt = orc.retarget_streams(g._model_blob, g._taskset_blob, state["q"][None], human[:, t:t + 1], nthreads=1)
        state["q"] = q[0, 0]
    out["cpu_oracle_per_frame"] = paced(cpu_step)
except Exception as e:  # noqa: BLE001
    out["cpu_oracle_per_frame"] = f"unavailable: {e}"
print(json.dumps(out, indent=1))
