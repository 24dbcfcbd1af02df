#!/usr/bin/env python3
"""Stress of the bound handling (block principal pivoting, Murty fallback, warm start) in all three QP code paths
(4-wavefront tree, 1-wavefront tree in DPP rows, dense) against the CPU oracle: targets far from reachable (large
position / orientation noise, scaled skeleton) push many joints onto their limits at once.  Measurement tool."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
out = {}
for src, robot in (("smplx", "unitree_g1"), ("smplx", "hightorque_hi"), ("bvh", "booster_t1"), ("smplx", "kuavo_s45")):
    g = GeneralMotionRetargeting(src, robot)
    for tag, (pn, rn, scale) in {"far": (0.30, 40.0, 1.0), "stretched": (0.05, 10.0, 1.35), "jumpy": (0.15, 25.0, 0.8)}.items():
        human, q0 = synth.make_streams(g.model, g._tables, 96, 24, seed=911, pos_noise=pn, rot_noise_deg=rn)
        root = human[:, :, :1, :3].copy()
        human[..., :3] = root + (human[..., :3] - root) * scale
        qc, nsc, stc = orc.retarget_streams(g._model_blob, g._taskset_blob, q0, human, nthreads=os.cpu_count())
        row = {"oracle_status_nonzero": int((stc != 0).sum()), "solves_per_frame": float(nsc.sum() / (96 * 24))}
        at_limit = 0
        lo, hi = g.model.range_lo, g.model.range_hi
        th = qc[..., 7:]
        at_limit = float(((th <= lo + 1e-9) | (th >= hi - 1e-9)).mean())
        row["fraction_of_joint_samples_on_a_limit"] = at_limit
        for waves in (4, 1):
            g.hip_solver.set_waves(waves)
            q, ns, st = g.retarget_streams(human, q0=q0)
            ok = (st == 0) & (stc == 0)
            row[f"nw{waves}"] = {"status_mismatch": int((st != stc).sum()),
                                 "max_joint_dev_rad": float(np.abs(q[ok][..., 7:] - qc[ok][..., 7:]).max()) if ok.any() else None,
                                 "frames_with_other_solve_count": int((ns[ok] != nsc[ok]).any(axis=-1).sum())}
        g.hip_solver.set_waves(0)
        out[f"{src}->{robot}:{tag}"] = row
        print(f"{src}->{robot}:{tag}", json.dumps(row), flush=True)
print(json.dumps(out, indent=1))
