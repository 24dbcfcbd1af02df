#!/usr/bin/env python3
"""BASELINE.json configs[2] and configs[3] (SURVEY.md section 8d "Config 3" / "Config 4") on ONE GPU.

Config 3: LAFAN1-shaped stand-in -- 77 ragged streams, ~496k frames in total, ``bvh_to_g1.json``; all
streams fit one launch (77 < 256 CUs), so the makespan is the longest clip; the LPT shards of an 8-GPU run
are listed beside it.  Config 4: 1 048 576 frames, 4 096 streams x 256 frames round-robin over six robots
with their smplx configs, one kernel per robot model on its own HIP stream (dataset.retarget_mixed)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, dataset, synth  # noqa: E402
from general_motion_retargeting_amd.sharding import lpt_partition  # noqa: E402

out = {}
# ---------------- config 3 ----------------
rng = np.random.default_rng(3)
lens = rng.integers(3000, 9500, size=77)
lens = (lens * (496000 / lens.sum())).astype(np.int32)
g = GeneralMotionRetargeting("bvh", "unitree_g1", actual_human_height=1.75)
T = int(lens.max())
base_h, base_q = synth.make_streams(g.model, g._tables, 77, 1200, seed=30)       # 1200-frame motifs, played back and forth
idx = np.arange(T) % 2398
idx = np.where(idx < 1200, idx, 2398 - idx)
human = np.ascontiguousarray(base_h[:, idx])
t0 = time.perf_counter()
q, ns, st = g.retarget_streams(human, lens=lens)
dt = time.perf_counter() - t0
assert (st == 0).all()
frames = int(lens.sum())
shards = lpt_partition(lens.tolist(), 8)
out["config3_lafan1_shape"] = {
    "streams": 77, "frames": frames, "longest_clip": T, "wall_s_pcie_inclusive": dt, "frames_per_s": frames / dt,
    "solves_per_frame": float(ns.sum() / frames),
    "lpt_8gpu_frames_per_rank": [int(lens[s].sum()) for s in shards],
    "lpt_8gpu_longest_clip_per_rank": [int(lens[s].max()) for s in shards],
    "note": "one launch, latency shape: the makespan is the longest clip x per-frame latency; sharding 77 streams over 8 "
            "GPUs cannot shorten it (each GPU would still wait for its longest clip)",
}
print(json.dumps(out["config3_lafan1_shape"]), flush=True)
del human, q, ns
# ---------------- config 4 ----------------
robots = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01", "hightorque_hi"]
S_total, T = 4096, 256
groups = []
for r, robot in enumerate(robots):
    gm = GeneralMotionRetargeting("smplx", robot)
    S = len(range(r, S_total, len(robots)))
    bh, _ = synth.make_streams(gm.model, gm._tables, 128, T, seed=1 + 1000 * r)
    groups.append({"src_human": "smplx", "tgt_robot": robot, "human": np.ascontiguousarray(np.tile(bh, ((S + 127) // 128, 1, 1, 1))[:S])})
dataset.retarget_mixed([{**gr, "human": gr["human"][:8, :8]} for gr in groups])       # warm-up (handles, code objects)
t0 = time.perf_counter()
res = dataset.retarget_mixed(groups)
dt = time.perf_counter() - t0
assert all((r[2] == 0).all() for r in res)
nfr = sum(gr["human"].shape[0] for gr in groups) * T
out["config4_mixed_1M"] = {"frames": nfr, "robots": robots, "streams": [int(gr["human"].shape[0]) for gr in groups],
                           "wall_s_pcie_inclusive": dt, "frames_per_s": nfr / dt,
                           "solves_per_frame": [float(r[1].sum() / (r[1].shape[0] * T)) for r in res]}
print(json.dumps(out, indent=1))
