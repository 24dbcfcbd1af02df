#!/usr/bin/env python3
"""How far could the genuine reference stack be from this build?  (CPU only; uses the oracle = test infrastructure.)

The one KNOWN behavioural difference between the restatement (oracle and HIP kernel alike) and the stack the
reference runs on is the QP solver's accuracy: DAQP stops at a primal tolerance of ~1e-6, the restatement solves
every box-QP exactly.  A QP solution that is off by 1e-6 moves `next` by O(1e-6), and where a stop-rule decision
`curr - next > 0.001` (reference motion_retarget.py:153,172) sits within such a margin it can flip and with it a
whole extra (or missing) solve.  This tool measures that exposure on the synthetic workloads of SURVEY.md 8(d):

  * per frame, the margin |(curr - next) - tol| of every stop-rule decision, the distance of the nearest
    inactive joint bound and the smallest multiplier of an active one (exact run);
  * the whole batch re-run (a) with every QP solved by an emulation of DAQP's termination rule at primal
    tolerance 1e-6 (bounds violated by <= 1e-6 never enter the working set, x not clipped) and (b) with every
    component of every QP solution moved by independent uniform noise of +-1e-6 (pessimistic: any solver that is
    only accurate to 1e-6), reporting the fraction of frames whose solve count changes and the joint deviation.

    python tools/parity_risk.py [--quick] > profiles/r02_parity_risk.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import params, synth  # noqa: E402
from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset  # noqa: E402
from general_motion_retargeting_amd.models import load_ik_config, load_robot  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def audit(src, robot, S, T, seed, threads):
    model = load_robot(params.ROBOT_XML_DICT[robot])
    tt = build_task_tables(load_ik_config(params.IK_CONFIG_DICT[src][robot]), None)
    mb, ts = pack_model(model), pack_taskset(model, tt)
    human, q0 = synth.make_streams(model, tt, S, T, seed=seed)
    q_ex, ns_ex, st_ex, mg = orc.retarget_streams_audit(mb, ts, q0, human, nthreads=threads)
    q_plain, ns_plain, _ = orc.retarget_streams(mb, ts, q0, human, nthreads=threads)
    assert np.array_equal(q_ex, q_plain) and np.array_equal(ns_ex, ns_plain), "audit hooks changed the exact run"
    assert (st_ex == 0).all()
    nfr = S * T
    stop = mg[..., 0].ravel()
    res = {
        "frames": nfr, "solves_per_frame": float(ns_ex.sum()) / nfr,
        "stop_margin": {f"frac_below_{t:g}": float((stop < t).mean()) for t in (1e-4, 1e-5, 1e-6, 1e-7)},
        "stop_margin_min": float(stop.min()),
        "inactive_bound_gap": {f"frac_below_{t:g}": float((mg[..., 1].ravel() < t).mean()) for t in (1e-5, 1e-6, 1e-7)},
        "active_multiplier": {f"frac_below_{t:g}": float((mg[..., 2].ravel() < t).mean()) for t in (1e-5, 1e-6, 1e-7)},
        "frames_with_an_active_bound": float(np.isfinite(mg[..., 2]).mean()),
    }
    for name, kw in (("daqp_like_ptol_1e-6", dict(qp_ptol=1e-6)), ("noise_1e-6", dict(qp_noise=1e-6, seed=seed)),
                     ("noise_1e-8", dict(qp_noise=1e-8, seed=seed))):
        q_p, ns_p, st_p, _ = orc.retarget_streams_audit(mb, ts, q0, human, nthreads=threads, **kw)
        assert (st_p == 0).all()
        diff = (ns_p != ns_ex).any(axis=-1)                       # [S, T]
        dj = np.abs(q_p[..., 7:] - q_ex[..., 7:]).max(axis=-1)    # [S, T] max joint deviation per frame
        # frames of streams in which no solve count has differed SO FAR (deviation not caused by a flipped branch)
        clean = np.cumsum(diff, axis=1) == 0
        res[name] = {
            "frames_with_different_solve_count": int(diff.sum()),
            "frac_frames_with_different_solve_count": float(diff.mean()),
            "streams_with_a_flip": int(diff.any(axis=1).sum()),
            "max_joint_dev_rad": float(dj.max()),
            "p999_joint_dev_rad": float(np.quantile(dj, 0.999)),
            "p99_joint_dev_rad": float(np.quantile(dj, 0.99)),
            "median_joint_dev_rad": float(np.median(dj)),
            "max_joint_dev_rad_before_any_flip": float(dj[clean].max()) if clean.any() else None,
            "max_joint_dev_rad_on_flipped_frames": float(dj[diff].max()) if diff.any() else None,
            "frac_frames_above_1e-4_rad": float((dj > 1e-4).mean()),
            "max_root_pos_dev_m": float(np.abs(q_p[..., :3] - q_ex[..., :3]).max()),
        }
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="small sizes (CI)")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    a = ap.parse_args()
    orc.build()
    out = {"what": __doc__.split("\n\n")[0],
           "contract": "max per-joint deviation 1e-4 rad (BASELINE.json north_star)", "workloads": {}}
    S, T = (16, 20) if a.quick else (100, 100)
    out["workloads"]["configs[1] smplx->unitree_g1 S=%d T=%d seed 0" % (S, T)] = audit("smplx", "unitree_g1", S, T, 0, a.threads)
    S2, T2 = (4, 12) if a.quick else (64, 64)
    for src in sorted(params.IK_CONFIG_DICT):
        for robot in sorted(params.IK_CONFIG_DICT[src]):
            if a.quick and robot not in ("unitree_g1", "hightorque_hi"):
                continue
            out["workloads"]["soak %s->%s S=%d T=%d seed 7" % (src, robot, S2, T2)] = audit(src, robot, S2, T2, 7, a.threads)
    # totals
    tot = {}
    for name in ("daqp_like_ptol_1e-6", "noise_1e-6", "noise_1e-8"):
        fr = sum(w["frames"] for w in out["workloads"].values())
        fl = sum(w[name]["frames_with_different_solve_count"] for w in out["workloads"].values())
        tot[name] = {"frames": fr, "frames_with_different_solve_count": fl, "frac": fl / fr,
                     "max_joint_dev_rad": max(w[name]["max_joint_dev_rad"] for w in out["workloads"].values()),
                     "max_joint_dev_rad_before_any_flip": max((w[name]["max_joint_dev_rad_before_any_flip"] or 0.0)
                                                                for w in out["workloads"].values())}
    out["totals"] = tot
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
