#!/usr/bin/env python3
"""Large-scale parity soak (measurement tool; may call the oracle like bench.py's cpu_baseline leg):
every shipped (source, robot) config at S=2560 x T=64 on the GPU in the throughput shape (more streams than resident
wavefronts: the queued dispatch), a random sample of
96 streams per config re-computed by the CPU oracle; reports max joint / root deviation and solve-count
mismatches; plus shard-invariance (the sample launched alone gives bit-identical results)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import IK_CONFIG_DICT, GeneralMotionRetargeting, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
S, T, NS = 2560, 64, 96
rng = np.random.default_rng(0)
out = {}
worst = 0.0
for src, tbl in IK_CONFIG_DICT.items():
    for robot in tbl:
        g = GeneralMotionRetargeting(src, robot)
        base_h, base_q = synth.make_streams(g.model, g._tables, 256, T, seed=hash((src, robot)) % 10000)
        human = np.tile(base_h, (S // 256, 1, 1, 1))
        # make the copies distinct: a small per-stream offset of the whole skeleton
        human[..., :3] += rng.normal(0, 0.02, size=(S, 1, 1, 3))
        q0 = np.tile(base_q, (S // 256, 1))
        t0 = time.perf_counter()
        q, ns, st = g.retarget_streams(human, q0=q0)
        dt = time.perf_counter() - t0
        pick = np.sort(rng.choice(S, NS, replace=False))
        qc, nsc, stc = orc.retarget_streams(g._model_blob, g._taskset_blob, q0[pick], human[pick], nthreads=os.cpu_count())
        dj = float(np.abs(q[pick][..., 7:] - qc[..., 7:]).max())
        dr = float(np.abs(q[pick][..., :7] - qc[..., :7]).max())
        mism = int((ns[pick] != nsc).any(axis=-1).sum())
        q2, ns2, st2 = g.retarget_streams(human[pick], q0=q0[pick])           # latency shape, different launch
        g.hip_solver.set_waves(1)
        q3, _, _ = g.retarget_streams(human[pick], q0=q0[pick])               # same shape, different batch
        g.hip_solver.set_waves(0)
        key = f"{src}->{robot}"
        out[key] = {"frames": S * T, "wall_s": dt, "status_nonzero": int((st != 0).sum()), "max_joint_dev_rad": dj,
                    "max_root_dev": dr, "frames_with_other_solve_count": mism, "solves_per_frame": float(ns.sum() / (S * T)),
                    "shape_dev_rad": float(np.abs(q2 - q[pick]).max()), "batch_invariant_bitwise": bool(np.array_equal(q3, q[pick]))}
        worst = max(worst, dj)
        print(key, json.dumps(out[key]), flush=True)
out["worst_joint_dev_rad"] = worst
print(json.dumps(out, indent=1))
