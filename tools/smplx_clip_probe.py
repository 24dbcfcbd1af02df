#!/usr/bin/env python3
"""SMPL-X front end per clip (host buffers in, packed frames out): body model + alignment as two library calls
(joints to the host, gather, second upload) against gmr_smplx_frames (one call, joints stay on the device).
    python tools/smplx_clip_probe.py [frames_per_clip] [reps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting  # noqa: E402
from general_motion_retargeting_amd.utils import smpl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(9)
J, V = 55, 120
v = rng.normal(0, 0.3, size=(V, 3)) + np.array([0, 0, 1.0])
sd = rng.normal(0, 0.01, size=(V, 3, 20))
jr = rng.uniform(0, 1, size=(J, V)); jr /= jr.sum(1, keepdims=True)
bm = smpl.SmplxBodyModel.from_arrays(v, sd, jr, smpl.SMPLX_PARENTS.astype(np.int64), 16, None)
data = {"betas": rng.normal(0, 0.5, size=16), "root_orient": np.cumsum(rng.normal(0, 0.02, size=(N, 3)), 0),
        "pose_body": np.cumsum(rng.normal(0, 0.02, size=(N, 63)), 0),
        "trans": np.cumsum(rng.normal(0, 0.01, size=(N, 3)), 0) + np.array([0, 0, 0.9]), "mocap_frame_rate": np.array(120.0)}
g = GeneralMotionRetargeting("smplx", "unitree_g1", actual_human_height=1.7)


def two_step():
    so = bm(betas=data["betas"], global_orient=data["root_orient"], body_pose=data["pose_body"], transl=data["trans"])
    return smpl.smplx_frames_packed(g, data, bm, so, tgt_fps=30)[0]


def fused():
    return smpl.smplx_frames_packed_fused(g, data, bm, tgt_fps=30)[0]


a, b = two_step(), fused()
out = {"frames_per_clip": N, "output_frames": int(len(a)), "bit_identical": bool(np.array_equal(a, b))}
for name, fn in (("two_calls", two_step), ("gmr_smplx_frames", fused)):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    out[name + "_ms_per_clip"] = (time.perf_counter() - t0) / reps * 1e3
print(json.dumps(out))
