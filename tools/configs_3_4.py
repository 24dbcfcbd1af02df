#!/usr/bin/env python3
"""BASELINE.json configs[2] and configs[3] (SURVEY.md section 8d "Config 3" / "Config 4") on ONE GPU.

Config 3 (``lafan``): LAFAN1-shaped stand-in -- 77 ragged streams, ~496k frames in total, ``bvh_to_g1.json``; all
streams fit one launch (77 < 256 CUs), so the makespan is the longest clip; the LPT shards of an 8-GPU run are listed
beside it.  Config 4 (``mixed``): 1 048 576 frames, 4 096 streams x 256 frames round-robin over six robots with their
smplx configs -- as ONE scheduling domain (group launch), device-resident and through host buffers (pinned, sliced;
and pageable), beside the round-2 way (one kernel per robot on its own HIP stream).

    python tools/configs_3_4.py [lafan] [mixed] [S_total T]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import GeneralMotionRetargeting, _lib, dataset, synth  # noqa: E402
from general_motion_retargeting_amd.sharding import lpt_partition  # noqa: E402

ROBOTS = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01", "hightorque_hi"]


def lafan_shape():
    rng = np.random.default_rng(3)
    lens = rng.integers(3000, 9500, size=77)
    lens = (lens * (496000 / lens.sum())).astype(np.int32)
    g = GeneralMotionRetargeting("bvh", "unitree_g1", actual_human_height=1.75)
    T = int(lens.max())
    base_h, _ = synth.make_streams(g.model, g._tables, 77, 1200, seed=30)       # 1200-frame motifs, played back and forth
    idx = np.arange(T) % 2398
    idx = np.where(idx < 1200, idx, 2398 - idx)
    human = np.ascontiguousarray(base_h[:, idx])
    t0 = time.perf_counter()
    q, ns, st = g.retarget_streams(human, lens=lens)
    dt = time.perf_counter() - t0
    assert (st == 0).all()
    frames = int(lens.sum())
    shards = lpt_partition(lens.tolist(), 8)
    return {
        "streams": 77, "frames": frames, "longest_clip": T, "wall_s_pcie_inclusive": dt, "frames_per_s": frames / dt,
        "solves_per_frame": float(ns.sum() / frames),
        "lpt_8gpu_frames_per_rank": [int(lens[s].sum()) for s in shards],
        "lpt_8gpu_longest_clip_per_rank": [int(lens[s].max()) for s in shards],
        "note": "one launch, latency shape: the makespan is the longest clip x per-frame latency; sharding 77 streams over 8 "
                "GPUs cannot shorten it (each GPU would still wait for its longest clip)",
    }


def mixed_batch(S_total=4096, T=256, motifs=128, seed=1):
    """One job per robot: `motifs` distinct streams tiled to the robot's share of S_total."""
    jobs = []
    for r, robot in enumerate(ROBOTS):
        gm = GeneralMotionRetargeting("smplx", robot)
        S = len(range(r, S_total, len(ROBOTS)))
        bh, _ = synth.make_streams(gm.model, gm._tables, min(motifs, S), T, seed=seed + 1000 * r)
        human = np.ascontiguousarray(np.tile(bh, ((S + len(bh) - 1) // len(bh), 1, 1, 1))[:S])
        q0 = np.broadcast_to(gm.model.qpos0, (S, gm.model.nq)).copy()
        jobs.append({"gmr": gm, "solver": gm.hip_solver, "human": human, "q0": q0, "robot": robot})
    return jobs


def measure_mixed(S_total=4096, T=256, reps=3):
    L = _lib.lib()
    jobs = mixed_batch(S_total, T)
    nfr = sum(j["human"].shape[0] for j in jobs) * T
    out = {"frames": nfr, "robots": ROBOTS, "streams": [int(j["human"].shape[0]) for j in jobs], "T": T}
    # ---- device-resident -------------------------------------------------------------------------------------------
    dev, bufs = [], []
    for j in jobs:
        sol, (S, _) = j["solver"], j["human"].shape[:2]
        b = (_lib.DeviceBuffer.from_host(j["q0"]), _lib.DeviceBuffer.from_host(j["human"]), _lib.DeviceBuffer(S * T * sol.nq * 8),
             _lib.DeviceBuffer(S * T * 8), _lib.DeviceBuffer(S * 4))
        bufs.append(b)
        dev.append((sol, S, T, b[0], b[1], None, b[2], b[3], b[4]))

    def timed(fn):
        fn()
        _lib.check(L.gmr_stream_sync(None))
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            _lib.check(L.gmr_stream_sync(None))
            ts.append(time.perf_counter() - t0)
        return min(ts), ts

    best, ts = timed(lambda: _lib.retarget_group_dev(dev, 0, None))
    out["device_resident_group"] = {"seconds": ts, "frames_per_s": nfr / best}
    ref = [(b[2].to_host((S, T, sol.nq), np.float64), b[3].to_host((S, T, 2), np.int32), b[4].to_host((S,), np.int32))
           for (sol, S, _, *_), b in zip(dev, bufs)]
    assert all((r[2] == 0).all() for r in ref)
    out["solves_per_frame"] = [float(r[1].sum() / (r[1].shape[0] * T)) for r in ref]
    streams = [_lib.Stream() for _ in jobs]

    def per_robot():
        for (sol, S, TT, q0, h, ln, qo, ns, st), hs in zip(dev, streams):
            sol.retarget_streams_dev(S, TT, q0, h, ln, 0, qo, ns, st, hs)

    best, ts = timed(per_robot)
    out["device_resident_per_robot_streams"] = {"seconds": ts, "frames_per_s": nfr / best,
                                                "note": "round 2: one kernel per robot on its own HIP stream"}
    same = all(np.array_equal(b[2].to_host((S, T, sol.nq), np.float64), r[0]) for (sol, S, _, *_), b, r in zip(dev, bufs, ref))
    out["group_bit_identical_to_per_robot"] = bool(same)
    for b in bufs:
        for x in b:
            x.free()
    # ---- host buffers ------------------------------------------------------------------------------------------------
    hj = [{"solver": j["solver"], "human": j["human"], "q0": j["q0"]} for j in jobs]
    pj = [{"solver": j["solver"], "human": _lib.pinned_copy(j["human"]), "q0": _lib.pinned_copy(j["q0"])} for j in jobs]
    for name, jj, pin, slices in (("host_pageable", hj, False, 1), ("host_pinned_1slice", pj, True, 1),
                                  ("host_pinned_2slices", pj, True, 2), ("host_pinned_4slices", pj, True, 4),
                                  ("host_pinned_2windows", pj, True, -2), ("host_pinned_4windows", pj, True, -4),
                                  ("host_pinned_8windows", pj, True, -8), ("host_pinned_16windows", pj, True, -16),
                                  ("host_pageable_4windows", hj, False, -4), ("host_pinned_auto", pj, True, 0)):
        res = None
        ts = []
        outs = _lib.group_outputs(jj, pinned=pin)
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            res = _lib.retarget_group(jj, 0, slices, outs=outs)
            ts.append(time.perf_counter() - t0)
        ok = all(np.array_equal(a[0], r[0]) and np.array_equal(a[1], r[1]) for a, r in zip(res, ref))
        out[name] = {"seconds": ts[1:], "frames_per_s": nfr / min(ts[1:]), "bit_identical": bool(ok)}
        del res
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    nums = [int(a) for a in args if a.isdigit()]
    which = [a for a in args if not a.isdigit()] or ["lafan", "mixed"]
    out = {}
    if "lafan" in which:
        out["config3_lafan1_shape"] = lafan_shape()
        print(json.dumps(out["config3_lafan1_shape"]), flush=True)
    if "mixed" in which:
        out["config4_mixed_1M"] = measure_mixed(*(nums[:2] if len(nums) >= 2 else (4096, 256)))
    print(json.dumps(out, indent=1))
