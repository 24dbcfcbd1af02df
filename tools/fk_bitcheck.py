#!/usr/bin/env python3
"""FK kernel variants against each other and against the CPU oracle on 2^20 random G1 frames (+ the golden inputs):

    GMR_HIP_LIBRARY=<variant>.so python tools/fk_bitcheck.py out.npz      # run a variant, save its outputs
    python tools/fk_bitcheck.py --compare a.npz b.npz                      # bit comparison of two runs
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in ("bp", "br"):
        same = np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f"{k}: bit-equal = {same}; differing floats = {int((a[k].view(np.uint32) != b[k].view(np.uint32)).sum())} of {a[k].size}; "
              f"max |diff| = {d.max():.3e}")
    sys.exit(0)

from general_motion_retargeting_amd import KinematicsModel, ROBOT_XML_DICT, _lib  # noqa: E402
from oracle import oracle  # noqa: E402

km = KinematicsModel(ROBOT_XML_DICT["unitree_g1"])
B = 1 << 20
rng = np.random.default_rng(0)
lo, hi = km.get_dof_limits()
dof = (np.asarray(lo) + (np.asarray(hi) - np.asarray(lo)) * rng.uniform(size=(B, km.num_dof))).astype(np.float32)
dof[: B // 8] = rng.uniform(-3.2, 3.2, size=(B // 8, km.num_dof)).astype(np.float32)      # beyond the limits too
rp = rng.normal(size=(B, 3)).astype(np.float32)
rq = rng.normal(size=(B, 4)).astype(np.float32)
rq /= np.linalg.norm(rq, axis=1, keepdims=True)
bp, br, _ = km.hip_handle.fk(rp, rq, dof, want_rot=True)
n = 1 << 14
obp, obr = oracle.fk_f32(km._tree, rp[:n], rq[:n], dof[:n])
print(os.path.basename(_lib.LIB_PATH), "vs CPU oracle on", n, "frames: max |dpos| = %.3e, max |drot| = %.3e" % (
    np.abs(bp[:n] - obp).max(), np.abs(br[:n] - obr).max()))
np.savez(sys.argv[1], bp=bp, br=br)
