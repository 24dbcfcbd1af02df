#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GMR retargeting hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (target preprocessing + two-stage IK, time loop on device) over one batch of
synthetic human-motion streams -> Unitree G1 (29-DoF).  Inputs are resident in HBM when a timed region starts; every
timed region is K steps after W warm-up steps, bracketed by a barrier + device synchronisation, MAX over ranks.

N = 1 (default)  `value` = BASELINE.json configs[1]: the 10k-frame batch, S=100 streams x T=100 frames (SURVEY.md
                 section 8d "Config 2"), with the roofline object of its kernel and the CPU baseline timed on this
                 box's host cores.  Beside it, `strong_1m`: the 1M-frame batch of the multi-GPU leg on this one GPU.
N > 1            one rank per GPU.  Either launched by `python -m torch.distributed.run` (the launcher only provides
                 RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), or plainly as `python bench.py --gpus N`: with WORLD_SIZE
                 unset the process becomes the LAUNCHER -- it never loads the library or touches a GPU, starts N fresh
                 rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / two free
                 ports), relays rank 0's JSON line and exits non-zero as soon as any rank does (the others are
                 killed; GMR_BENCH_TIMEOUT bounds the whole job).  The ranks talk through the library's own RCCL
                 communicator, no PyTorch.  `value` = the 1M-frame synthetic batch north_star names -- S=131072 streams x T=8
                 frames, the same seeds for every N (wide enough that one of 8 GPUs still holds 16 384 streams: measured
                 on one GPU, 16 384 streams run at 0.91 of the rate of 131 072, 8 192 x 16 only at 0.83 of 65 536 x 16:
                 the tail of the longest streams, no communication involved) -- LPT-sharded over the N GPUs after ONE
                 RCCL broadcast of the packed
                 robot model + task set; no per-step collective ("scaling": "strong").  Rank 0 then runs the whole
                 batch alone on its GPU in the same run: `value_1gpu`, `efficiency` = value / (N * value_1gpu).
                 `weak_leg`: configs[1]'s S=100 x T=100 per rank (its makespan is the slowest of 100 N streams: the
                 value falls with N from that statistical tail alone, no communication involved).

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from general_motion_retargeting_amd import _lib, comm as gcomm, params, sharding, synth  # noqa: E402
from general_motion_retargeting_amd.ik_config import (MODEL_DTYPE, TASKSET_DTYPE, build_task_tables,  # noqa: E402
                                                      pack_model, pack_taskset)
from general_motion_retargeting_amd.models import load_ik_config, load_robot  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (spec sheet)
def _strip_comments(src: str) -> str:
    """C / C++ source without comments and with runs of white space collapsed (string and character literals kept)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c in "\"'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def kernel_sources_sha256() -> str:
    """Hash of the CODE of every kernel / layout source of libgmrhip.so (comments and white space do not count) and of the
    compiler flags: profiles/traffic.json carries the hash it was measured on."""
    import hashlib
    from general_motion_retargeting_amd import build
    h = hashlib.sha256()
    d = os.path.join(ROOT, "general_motion_retargeting_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")):
            h.update(name.encode())
            with open(os.path.join(d, name), "r", errors="replace") as f:
                h.update(_strip_comments(f.read()).encode())
    h.update(repr((build.FLAGS, sorted(build.PER_SOURCE_FLAGS.items()))).encode())
    return h.hexdigest()


def _traffic_profile():
    """profiles/traffic.json if it was measured on THESE kernel sources, else (None, why)."""
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(prof):
        return None, "profiles/traffic.json missing"
    try:
        with open(prof) as f:
            pj = json.load(f)
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/traffic.json unreadable: {e}"
    if pj.get("kernel_sources_sha256") != kernel_sources_sha256():
        return None, "stale: profiles/traffic.json was measured on other kernel sources than HEAD's (re-run tools/profile_round3.sh + tools/publish_profiles.py)"
    return pj, None


def _wide_roofline(robot, frames, solves_per_frame, seconds):
    """HBM and FP64-VALU fractions of the throughput kernel over one step of the 1M-frame leg (wall clock of the step:
    queue initialisation + kernel), same accounting as the headline's `roofline`."""
    nsolve = solves_per_frame * frames
    flops = nsolve * F_ITER_DENSE.get(robot, 1.77e5) + (nsolve + 2 * frames) * F_ERR.get(robot, 2.1e3)
    gbs = BYTES_PER_FRAME.get(robot, 1360) * frames / seconds / 1e9
    return {"kernel": "ik_wide_kernel (queued dispatch)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": BYTES_PER_FRAME.get(robot, 1360),
            "fp64_valu": dict({"achieved_tflops": flops / seconds / 1e12, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                               "frac": flops / seconds / 1e12 / FP64_VALU_PEAK_TFLOPS,
                               "flops_model": "SURVEY.md 8(d) dense accounting, measured solve counts"},
                              **_executed_frac(nsolve, seconds))}


def _executed_frac(nsolve, seconds):
    """FP64 flop the throughput kernel EXECUTES (counter file: FMA / MUL / ADD / TRANS wave-instructions per solve x the mean
    active-lane fraction) x the solves measured here -- beside the dense-model `frac`, which counts flop a sparse kernel
    rightly never performs."""
    pj, why = _traffic_profile()
    w = (pj or {}).get("wide")
    if not w:
        return {"executed_frac": None, "executed_note": why or "no throughput-kernel counters in profiles/traffic.json"}
    tf = w["executed_fp64_flop_per_solve"] * nsolve / seconds / 1e12
    return {"executed_tflops": tf, "executed_frac": tf / FP64_VALU_PEAK_TFLOPS,
            "executed_frac_all_lanes_live": w["executed_fp64_flop_per_solve_all_lanes_live"] * nsolve / seconds / 1e12 / FP64_VALU_PEAK_TFLOPS,
            "mean_active_lane_fraction": w.get("mean_active_lane_fraction_of_valu"), "valu_per_solve": w.get("valu_per_solve"),
            "lds_bank_conflict_share": w.get("lds_bank_conflict_share"),
            "mfma_f64": {"insts_per_solve": w.get("mfma_f64_insts_per_solve"),
                         "achieved_tflops": (w.get("mfma_f64_flop_per_solve") or 0.0) * nsolve / seconds / 1e12,
                         "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                         "frac": (w.get("mfma_f64_flop_per_solve") or 0.0) * nsolve / seconds / 1e12 / FP64_VALU_PEAK_TFLOPS,
                         "note": "the Schur products of the tree solver (four 9x7 by 7x9 per factorisation) as v_mfma_f64_16x16x4; "
                                 "MI355X FP64 matrix peak = FP64 vector peak (spec sheet)"},
            "executed_source": "profiles/traffic.json `wide` (rocprofv3 PMC, same kernel sources as HEAD) x the solve count of this run"}


# algorithmic bytes and flops
BYTES_PER_FRAME = {"unitree_g1": 1360}
F_ITER_DENSE = {"unitree_g1": 1.77e5}
F_ERR = {"unitree_g1": 2.1e3}
METRIC = "retargeted frames/sec (whole node) + max joint-angle err vs CPU ref, G1 29-DoF"


def self_launch(n: int, argv) -> int:
    """`python bench.py --gpus N` without an external launcher: this process starts N rank processes of this script
    and does nothing else -- in particular it never loads libgmrhip.so and never initialises a GPU.  Rank 0's stdout
    (the ONE JSON line) is relayed; the first non-zero exit of any rank ends the job with that code."""
    from general_motion_retargeting_amd import launcher
    return launcher.self_launch(n, [sys.executable, os.path.abspath(__file__)] + list(argv), "GMR_BENCH_TIMEOUT", 1500.0,
                                require_stdout_prefix="{")


def load_standin():
    """GMR_BENCH_STANDIN=/path/to/file.py: a module that provides `init(local_rank)`, `Solver(model_blob, taskset_blob)`,
    `Shard(solver, q0, human)`, `Event()` and `device_sync()` in place of the HIP ones -- the CPU rehearsal of the
    launcher and of the N > 1 protocol in tests/ (no GPU there).  Never set on a GPU box; the line says `"standin"`."""
    path = os.environ.get("GMR_BENCH_STANDIN")
    if not path:
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("gmr_bench_standin", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class Shard:
    """One batch of streams resident on this rank's GPU + the function that retargets it once."""

    def __init__(self, solver, q0, human):
        self.solver = solver
        self.S, self.T = human.shape[0], human.shape[1]
        nq = solver.nq
        self.d_q0 = _lib.DeviceBuffer.from_host(q0)
        self.d_human = _lib.DeviceBuffer.from_host(human)
        self.d_qout = _lib.DeviceBuffer(max(self.S * self.T * nq * 8, 8))
        self.d_ns = _lib.DeviceBuffer(max(self.S * self.T * 2 * 4, 8))
        self.d_st = _lib.DeviceBuffer(max(self.S * 4, 8))

    def step(self, ev0=None, ev1=None):
        if ev0 is not None:
            ev0.record(None)
        self.solver.retarget_streams_dev(self.S, self.T, self.d_q0, self.d_human, None, 0, self.d_qout, self.d_ns, self.d_st, None)
        if ev1 is not None:
            ev1.record(None)

    def results(self):
        nq = self.solver.nq
        return (self.d_qout.to_host((self.S, self.T, nq), np.float64), self.d_ns.to_host((self.S, self.T, 2), np.int32),
                self.d_st.to_host((self.S,), np.int32))

    def free(self):
        for b in (self.d_q0, self.d_human, self.d_qout, self.d_ns, self.d_st):
            b.free()


# ---------------------------------------------------------------------------------------------------------------------
# Extra legs of the N = 1 line: the other BASELINE.json configs and the end-to-end (PCIe-
# inclusive) rates in the driver-run record.  Each leg carries its own `workload` string; a leg that fails reports
# {"error": ...} and never touches the headline fields.  Run after every timed region of the headline.
# ---------------------------------------------------------------------------------------------------------------------
def _wall(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def leg_e2e(solver, q0, human, q_all, h_all, steps):
    """configs[1] and the 1M-frame batch THROUGH HOST BUFFERS (SURVEY.md 8d timing protocol: "wall-clock end-to-end
    incl. H2D/D2H ... report both"): pageable NumPy arrays, and pinned arrays with the copies of a slice running under
    the kernels of its neighbours (gmr_retarget_group)."""
    out = {}

    def both(tag, q, h, reps, workload):
        frames = h.shape[0] * h.shape[1]
        jp = [{"solver": solver, "human": h, "q0": q}]
        ts_page = _wall(lambda: _lib.retarget_group(jp, 0, 0), reps + 1)[1:]
        jpin = [{"solver": solver, "human": _lib.pinned_copy(h), "q0": _lib.pinned_copy(q)}]
        outs = _lib.group_outputs(jpin)          # page-locked once: locking 300 MB costs more than retargeting 1M frames
        ts_pin = _wall(lambda: _lib.retarget_group(jpin, 0, 0, outs=outs), reps + 1)[1:]
        ts_pin1 = _wall(lambda: _lib.retarget_group(jpin, 0, 1, outs=outs), reps + 1)[1:]
        out[tag] = {"workload": workload, "unit": "frames/s", "steps": reps,
                    "pinned_overlapped": frames / float(np.mean(ts_pin)), "pinned_one_slice": frames / float(np.mean(ts_pin1)),
                    "pageable": frames / float(np.mean(ts_page)),
                    "ms_per_step": {"pinned_overlapped": float(np.mean(ts_pin)) * 1e3, "pinned_one_slice": float(np.mean(ts_pin1)) * 1e3,
                                    "pageable": float(np.mean(ts_page)) * 1e3},
                    "bytes_per_step": {"h2d": int(h.nbytes + q.nbytes), "d2h": int(frames * (solver.nq * 8 + 8) + h.shape[0] * 4)}}

    both("configs1", q0, human, max(2, min(steps, 10)),
         f"configs[1] S={human.shape[0]} x T={human.shape[1]} through host buffers: H2D + kernel + D2H per step, wall clock")
    if h_all is not None:
        both("strong_1m", q_all, h_all, 2,
             f"the 1M-frame batch S={h_all.shape[0]} x T={h_all.shape[1]} through host buffers, wall clock; `pinned_overlapped` = "
             f"sliced automatically (about 64 MB of input per slice on four HIP streams)")
    return out


def leg_config2(seed=30):
    """BASELINE.json configs[2] on this one GPU: LAFAN1-shaped stand-in (no BVH files offline; SURVEY.md 8d "Config 3")."""
    from general_motion_retargeting_amd import GeneralMotionRetargeting
    rng = np.random.default_rng(3)
    lens = rng.integers(3000, 9500, size=77)
    lens = (lens * (496000 / lens.sum())).astype(np.int32)
    g = GeneralMotionRetargeting("bvh", "unitree_g1", actual_human_height=1.75)
    T = int(lens.max())
    base_h, _ = synth.make_streams(g.model, g._tables, 77, 1200, seed=seed, workers=4)   # 1200-frame motifs, played back and forth
    idx = np.arange(T) % 2398
    idx = np.where(idx < 1200, idx, 2398 - idx)
    human = np.ascontiguousarray(base_h[:, idx])
    frames = int(lens.sum())
    sol = g.hip_solver
    S, nq = 77, sol.nq
    q0 = np.broadcast_to(g.model.qpos0, (S, nq)).copy()
    d = [_lib.DeviceBuffer.from_host(q0), _lib.DeviceBuffer.from_host(human), _lib.DeviceBuffer.from_host(lens),
         _lib.DeviceBuffer(S * T * nq * 8), _lib.DeviceBuffer(S * T * 8), _lib.DeviceBuffer(S * 4)]
    ms = []
    for i in range(2):
        a, b = _lib.Event(), _lib.Event()
        a.record(None)
        sol.retarget_streams_dev(S, T, d[0], d[1], d[2], 0, d[3], d[4], d[5], None)
        b.record(None)
        ms.append(a.elapsed_ms(b))
    st = d[5].to_host((S,), np.int32)
    ns = d[4].to_host((S, T, 2), np.int32)
    for x in d:
        x.free()
    t0 = time.perf_counter()
    q, ns2, st2 = g.retarget_streams(human, lens=lens)
    wall = time.perf_counter() - t0
    shards = sharding.lpt_partition(lens.tolist(), 8)
    return {"workload": f"BASELINE.json configs[2] stand-in: 77 ragged streams (LAFAN1's shape, bvh -> Unitree G1), {frames} frames, "
                        f"longest clip {T}, ONE launch on one GPU (latency shape: 77 < 256 CUs)",
            "unit": "frames/s", "value": frames / (min(ms) * 1e-3), "ms_per_step": float(min(ms)), "steps": 2,
            "pcie_inclusive_pageable": frames / wall, "failed_streams": int((st != 0).sum() + (st2 != 0).sum()),
            "mean_solves_per_frame": float(ns.sum()) / frames,
            "lpt_8gpu_frames_per_rank": [int(lens[s_].sum()) for s_ in shards],
            "lpt_8gpu_longest_clip_per_rank": [int(lens[s_].max()) for s_ in shards],
            "note": "the makespan is the longest clip x per-frame latency: sharding 77 streams over 8 GPUs cannot shorten it"}


SIX_ROBOTS = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01", "hightorque_hi"]


def leg_config3(S_total=4096, T=256, motifs=128, seed=1):
    """BASELINE.json configs[3] on this one GPU: 1M frames round-robin over six robots as ONE scheduling domain."""
    from general_motion_retargeting_amd import GeneralMotionRetargeting
    jobs, keep = [], []
    for r, robot in enumerate(SIX_ROBOTS):
        gm = GeneralMotionRetargeting("smplx", robot)
        S = len(range(r, S_total, len(SIX_ROBOTS)))
        bh, _ = synth.make_streams(gm.model, gm._tables, min(motifs, S), T, seed=seed + 1000 * r, workers=4)
        human = _lib.pinned_empty((S, T, bh.shape[2], 7))
        for s0 in range(0, S, len(bh)):
            human[s0:s0 + len(bh)] = bh[: S - s0]
        q0 = _lib.pinned_copy(np.broadcast_to(gm.model.qpos0, (S, gm.model.nq)))
        keep.append(gm)
        jobs.append({"solver": gm.hip_solver, "human": human, "q0": q0})
    nfr = sum(j["human"].shape[0] for j in jobs) * T
    dev, bufs = [], []
    for j in jobs:
        sol, S = j["solver"], j["human"].shape[0]
        b = (_lib.DeviceBuffer.from_host(j["q0"]), _lib.DeviceBuffer.from_host(j["human"]), _lib.DeviceBuffer(S * T * sol.nq * 8),
             _lib.DeviceBuffer(S * T * 8), _lib.DeviceBuffer(S * 4))
        bufs.append(b)
        dev.append((sol, S, T, b[0], b[1], None, b[2], b[3], b[4]))
    ms = []
    for i in range(3):
        a, b = _lib.Event(), _lib.Event()
        a.record(None)
        _lib.retarget_group_dev(dev, 0, None)
        b.record(None)
        ms.append(a.elapsed_ms(b))
    failed = sum(int((b[4].to_host((S,), np.int32) != 0).sum()) for (_, S, *_), b in zip(dev, bufs))
    spf = [float(b[3].to_host((S, T, 2), np.int32).sum()) / (S * T) for (_, S, *_), b in zip(dev, bufs)]
    for b in bufs:
        for x in b:
            x.free()
    outs = _lib.group_outputs(jobs)
    ts = _wall(lambda: _lib.retarget_group(jobs, 0, 0, outs=outs), 3)[1:]
    return {"workload": f"BASELINE.json configs[3] on ONE GPU: {nfr} frames = {S_total} streams x {T} frames round-robin over "
                        f"{', '.join(SIX_ROBOTS)} (smplx configs), one group launch = one resident grid + one device-side queue",
            "unit": "frames/s", "value": nfr / (min(ms[1:]) * 1e-3), "ms_per_step": float(min(ms[1:])), "steps": 2,
            "pcie_inclusive_pinned": nfr / float(np.mean(ts)), "failed_streams": failed, "mean_solves_per_frame": spf}


def leg_config4_hip(nframes=300, warm=30, hz=120.0):
    """BASELINE.json configs[4]: one stream, 51-body frames delivered one at a time at 120 Hz, retarget(dict) per frame."""
    from general_motion_retargeting_amd import GeneralMotionRetargeting
    g = GeneralMotionRetargeting("fbx", "unitree_g1", actual_human_height=1.6)
    human, q0 = synth.make_streams(g.model, g._tables, 1, nframes, seed=3)
    frames = synth.streams_to_dicts(g._tables, human[0])
    lat = []
    t_next = time.perf_counter()
    for t in range(nframes):
        while time.perf_counter() < t_next:
            pass
        t_next += 1.0 / hz
        a = time.perf_counter()
        g.retarget(frames[t])
        lat.append(time.perf_counter() - a)
    lat = np.array(lat[warm:]) * 1e3
    return ({"workload": f"BASELINE.json configs[4]: fbx -> Unitree G1, one stream, {nframes - warm} frames paced at {hz:.0f} Hz, "
                         f"one retarget(dict) call per frame (1 H2D + 1 launch + 1 D2H)",
             "unit": "ms per frame", "p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)),
             "max_ms": float(lat.max())}, (g, human, q0, nframes, warm, hz))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=100, help="configs[1] leg: streams per GPU")
    ap.add_argument("--frames", type=int, default=100, help="configs[1] leg: frames per stream")
    ap.add_argument("--strong-streams", type=int, default=131072, help="1M-frame leg: streams in the whole batch")
    ap.add_argument("--strong-frames", type=int, default=8, help="1M-frame leg: frames per stream")
    ap.add_argument("--no-strong", action="store_true", help="skip the 1M-frame leg")
    ap.add_argument("--robot", default="unitree_g1")
    ap.add_argument("--src", default="smplx")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="N = 1: skip the e2e / configs[2..4] legs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (before anything loads the library or touches a GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank, local_rank, world = gcomm.env_rank_world()
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: running {world}", file=sys.stderr)
        args.gpus = world
    standin = load_standin()

    model = load_robot(params.ROBOT_XML_DICT[args.robot])
    tt = build_task_tables(load_ik_config(params.IK_CONFIG_DICT[args.src][args.robot]), None)

    # ---- synthetic inputs (host; before anything touches the GPU: the generator may fork workers) -------------------
    S, T = args.streams, args.frames
    human, q0 = synth.make_streams(model, tt, S, T, seed=args.seed + rank * S)
    SS, ST = args.strong_streams, args.strong_frames
    strong = not args.no_strong
    strong_seed = args.seed + 1_000_003
    lens = np.full(SS, ST, dtype=np.int64)
    my_ids = sharding.lpt_partition(lens, world)[rank] if strong else []
    # generator processes of this rank: its share of the host's cores (every rank of the node generates at the same time)
    workers = max(1, min(16, len(os.sched_getaffinity(0)) // max(1, world))) if hasattr(os, "sched_getaffinity") else 4
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith("ROCPROF") for k in os.environ):
        workers = 1         # a profiler's preloaded tool library owns the GPU already: do not fork worker processes under it
    if strong:
        if rank == 0:       # rank 0 holds the whole batch (the one-GPU reference leg); its shard is a view of it
            h_all, q_all = synth.make_streams(model, tt, SS, ST, seed=strong_seed, workers=workers)
            h_mine, q_mine = h_all[my_ids], q_all[my_ids]
        else:
            h_mine, q_mine = synth.make_streams_ids(model, tt, my_ids, ST, seed=strong_seed, workers=max(1, workers // 2))

    # ---- device + communicator -------------------------------------------------------------------------------------
    # No torch in this process unless a torch backend is asked for: bind the library to the SYSTEM HIP runtime, the one
    # the system's librccl.so is linked against (with torch installed, _lib would otherwise preload torch's bundled copy
    # so that a later `import torch` shares it: two HIP runtimes in one process is what that avoids, here as there).
    if (os.environ.get("GMR_BENCH_BACKEND") or os.environ.get("GMR_COMM_BACKEND") or "rccl").lower() in ("rccl", "tcp"):
        os.environ.setdefault("GMR_HIP_RUNTIME", "system")
    # A job whose RCCL bring-up fails (agreed by all ranks over the control star) still yields a scaling curve: it
    # continues on the star and `comm_backend` says "tcp (fallback: <reason>)".  GMR_COMM_FALLBACK=none makes it fatal.
    os.environ.setdefault("GMR_COMM_FALLBACK", "tcp")
    if standin is None:
        L = _lib.lib()
        _lib.require_gpu()
        _lib.check(L.gmr_set_device(local_rank % max(L.gmr_device_count(), 1)))
        SolverCls, ShardCls, EventCls, device_sync = _lib.Solver, Shard, _lib.Event, sharding._device_sync
    else:
        standin.init(local_rank)
        SolverCls, ShardCls, EventCls, device_sync = standin.Solver, standin.Shard, standin.Event, standin.device_sync
    # GMR_BENCH_FORCE_DIST=1 with ONE rank walks the whole RCCL path (init, broadcast, barrier, reductions) on one GPU
    try:
        comm = gcomm.create(os.environ.get("GMR_BENCH_BACKEND"), force=os.environ.get("GMR_BENCH_FORCE_DIST") == "1")
    except Exception as e:      # the same error on every rank (comm.py): say it once per rank and leave non-zero
        print(f"[bench] rank {rank}: communicator could not be created: {e}", file=sys.stderr, flush=True)
        raise SystemExit(3)
    if comm.world != world or comm.rank != rank:
        print(f"[bench] rank {rank}: communicator reports rank {comm.rank} of {comm.world}, launcher said {rank} of {world}", file=sys.stderr)
        raise SystemExit(3)

    # ---- rank 0 compiles the robot + task set; ONE broadcast ships it to the peers -----------------------------------
    mb0 = ts0 = None
    if rank == 0:
        mb0, ts0 = pack_model(model), pack_taskset(model, tt)
    mb, ts = sharding.broadcast_blobs(mb0, ts0, rank, comm)      # RCCL over xGMI, ~24 KB, once
    assert mb.dtype == MODEL_DTYPE and ts.dtype == TASKSET_DTYPE
    solver = SolverCls(mb, ts)

    # ---- leg A: BASELINE.json configs[1] (per rank) -------------------------------------------------------------------
    shard = ShardCls(solver, q0, human)
    for _ in range(args.warmup):
        shard.step()
    comm.barrier()
    evs = [(EventCls(), EventCls()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        shard.step(*evs[i])
    comm.barrier()
    mine = time.perf_counter() - t0
    elapsed = comm.allreduce_max(mine)
    per_rank = comm.allgather(mine)
    kern_ms = [a.elapsed_ms(b) for a, b in evs]
    q_hip, ns_hip, st_hip = shard.results()
    assert (st_hip == 0).all(), "IK kernel reported a failed stream"
    shard.free()

    # ---- leg B: the 1M-frame batch, sharded over the ranks (and whole on rank 0) --------------------------------------
    sres = None
    if strong:
        keep = {}

        def make_step(ids):
            keep["mine"] = ShardCls(solver, q_mine, h_mine)
            return keep["mine"].step

        def single_step():
            keep["mine"].free()
            keep["all"] = ShardCls(solver, q_all, h_all)
            return keep["all"].step

        # N > 1: `value` -> exactly K steps; N = 1: an extra beside configs[1], kept short
        k_strong = args.steps if world > 1 else max(1, min(args.steps, 5))
        sres = sharding.strong_scaling_leg(comm, lens, make_step, k_strong, min(args.warmup, 1),
                                           single_step if world > 1 else None, device_sync=device_sync)
        last = keep.get("all") or keep["mine"]
        _, ns_s, st_s = last.results()
        assert (st_s == 0).all(), "IK kernel reported a failed stream (1M-frame leg)"
        sres["steps"] = k_strong
        sres["mean_solves_per_frame"] = float(ns_s.sum()) / (last.S * last.T)

    if rank == 0:
        frames_launch = S * T
        k_ms = float(np.mean(kern_ms))
        bpf = BYTES_PER_FRAME.get(args.robot, 1360)
        achieved_gbs = bpf * frames_launch / (k_ms * 1e-3) / 1e9
        nsolve_total = int(ns_hip.sum())
        flops_launch = nsolve_total * F_ITER_DENSE.get(args.robot, 1.77e5) + \
            (nsolve_total + 2 * frames_launch) * F_ERR.get(args.robot, 2.1e3)
        weak_value = frames_launch * world * args.steps / elapsed
        roofline = {
            "kernel": "ik_streams_kernel<36, 4, 1>",
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": None,
            "kernel_ms": k_ms,
            "algorithmic_bytes_per_frame": bpf,
            "note": "latency/FP64-bound by construction (SURVEY.md F9): HBM fraction is reported, not the limiter",
            "fp64_valu": {
                "achieved_tflops": flops_launch / (k_ms * 1e-3) / 1e12,
                "peak_tflops": FP64_VALU_PEAK_TFLOPS,
                "frac": flops_launch / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS,
                "flops_model": "SURVEY.md 8(d) dense accounting, measured solve counts",
            },
        }
        pj, why = _traffic_profile()
        if pj is not None and pj.get("streams") == S and pj.get("frames") == T:
            roofline["traffic"] = pj.get("hbm_bytes_per_launch")
            roofline["traffic_source"] = pj.get("source")
        else:
            roofline["traffic_note"] = why or "profiles/traffic.json holds another workload"
        weak_cfg = {
            "workload": f"10k-frame synthetic SMPL-X batch -> Unitree G1 (29-DoF): S={S} streams x T={T} "
                        f"frames per GPU, two-stage IK, time loop on device (BASELINE.json configs[1])",
            "robot": args.robot, "source": args.src, "streams_per_gpu": S, "frames_per_stream": T,
            "parallelism": f"streams sharded over {world} GPU(s), no per-step collective",
            "mean_solves_per_frame": nsolve_total / frames_launch,
        }
        out = {"metric": METRIC, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
        if world == 1 and comm.backend == "none" or sres is None:
            out.update({"value": weak_value, "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak", "config": weak_cfg,
                        "roofline": roofline})
            if sres is not None:
                out["strong_1m"] = {
                    "workload": f"1M-frame synthetic batch -> Unitree G1: S={SS} streams x T={ST} frames on this one GPU "
                                f"(the batch `--gpus N` shards; throughput shape: one wavefront per stream)",
                    "value": sres["value"], "unit": "frames/s", "steps": sres["steps"],
                    "ms_per_step": sres["seconds"] / sres["steps"] * 1e3, "mean_solves_per_frame": sres["mean_solves_per_frame"],
                    "roofline": _wide_roofline(args.robot, SS * ST, sres["mean_solves_per_frame"], sres["seconds"] / sres["steps"]),
                }
        else:
            out.update({
                "value": sres["value"], "ms_per_step": sres["seconds"] / sres["steps"] * 1e3, "scaling": "strong",
                "steps": sres["steps"],
                "config": {
                    "workload": f"1M-frame synthetic AMASS-shaped batch -> Unitree G1 (29-DoF): S={SS} streams x T={ST} frames "
                                f"= {SS * ST} frames in total, LPT-sharded over {world} GPU(s), the same seeds for every N "
                                f"(north_star; BASELINE.json configs[3] shape on one robot).  `value` at N > 1 is THIS strong-"
                                f"scaling batch; its one-GPU anchor is `value_1gpu` here (same run, rank 0's GPU) = "
                                f"`strong_1m.value` of the N = 1 line -- NOT the N = 1 line's `value`, which is configs[1] "
                                f"(10k frames, latency shape; repeated here per rank as `weak_leg`)",
                    "robot": args.robot, "source": args.src, "streams": SS, "frames_per_stream": ST,
                    "parallelism": f"streams sharded over {world} GPU(s): one {comm.backend} broadcast of the 24 KB model + "
                                   f"task set, no per-step collective",
                    "mean_solves_per_frame": sres["mean_solves_per_frame"],
                },
                "world_size": comm.world, "world_size_launcher": world, "comm_backend": comm.backend,
                "launcher": "self (bench.py --gpus N)" if os.environ.get("GMR_BENCH_SELF_LAUNCHED") == "1" else "external (RANK / WORLD_SIZE from the environment)",
                "frames_per_rank": sres["frames_per_rank"],
                "per_rank_ms_per_step": [t / sres["steps"] * 1e3 for t in sres["per_rank_seconds"]],
                "value_1gpu": sres.get("value_1gpu"), "efficiency": sres.get("efficiency"),
                "efficiency_note": "value / (N * value_1gpu); value_1gpu = the whole batch on rank 0's GPU alone, same run",
                "weak_leg": {
                    "workload": weak_cfg["workload"], "value": weak_value, "ms_per_step": elapsed / args.steps * 1e3,
                    "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank], "steps": args.steps,
                    "note": "every stream has a CU to itself: the time is the SLOWEST of the 100*N streams (per-solve latency x "
                            "its solve count), so this value falls with N from the statistical tail alone",
                    "roofline": roofline,
                },
            })
        # ---- the other configs and the end-to-end rates (N = 1 only; after every timed region of the headline) ------------
        leg4_ctx = None
        if world == 1 and comm.backend == "none" and standin is None and not args.no_extra_legs:
            if sres is not None:
                for k_ in ("mine", "all"):
                    if k_ in keep:
                        keep[k_].free()
            legs = {}

            def run_leg(name, fn):
                t_leg = time.perf_counter()
                try:
                    legs[name] = fn()
                except Exception as e:  # noqa: BLE001 -- an extra leg must never cost the headline
                    legs[name] = {"error": f"{type(e).__name__}: {e}"}
                if isinstance(legs[name], dict):
                    legs[name]["leg_seconds"] = round(time.perf_counter() - t_leg, 2)

            run_leg("e2e", lambda: leg_e2e(solver, q0, human, q_all if strong else None, h_all if strong else None, args.steps))
            run_leg("config2_lafan1_shape", leg_config2)
            run_leg("config3_mixed_1m", leg_config3)

            def _leg4():
                nonlocal leg4_ctx
                res, leg4_ctx = leg_config4_hip()
                return res
            run_leg("config4_latency", _leg4)
            out["legs"] = legs
        if standin is not None:
            out["standin"] = os.environ["GMR_BENCH_STANDIN"]
            out["data"] = "synthetic (STAND-IN compute function: rehearsal of the launcher / N > 1 protocol, not a measurement)"
        if not args.no_cpu_baseline:
            from oracle import oracle as orc   # CPU restatement: the checker and the timed CPU leg
            orc.build()
            cores = os.cpu_count() or 1
            try:
                cores = len(os.sched_getaffinity(0))
            except Exception:
                pass
            t1 = time.perf_counter()
            q_cpu, ns_cpu, st_cpu = orc.retarget_streams(mb, ts, q0, human, nthreads=cores)
            cpu_s = time.perf_counter() - t1
            s1 = min(S, 8)
            t1 = time.perf_counter()
            orc.retarget_streams(mb, ts, q0[:s1], human[:s1], nthreads=1)
            cpu1_s = time.perf_counter() - t1
            out["cpu_baseline"] = {
                "value": S * T / cpu_s,
                "unit": "frames/s",
                "cores": cores,
                "kind": "port",
                "sample": f"the same S={S}xT={T} batch (rank 0's shard of configs[1]), OpenMP over streams on {cores} host "
                          f"threads; 1-thread rate on the first {s1} streams: {s1 * T / cpu1_s:.0f} frames/s",
                "note": "C restatement of mink/MuJoCo/DAQP (oracle/gmr_oracle.c); the genuine reference cannot run offline "
                        "(published: 35-70 frames/s single stream, README.md:217-220).  Parity of the restatement with the "
                        "genuine stack is unpinned for the IK numerics; the one known difference (DAQP stops at ~1e-6 primal "
                        "tolerance, the QP here is solved exactly) was measured: 0 of 67 344 frames change their solve count "
                        "under a DAQP-like termination rule, max joint deviation 5.8e-7 rad (profiles/r02_parity_risk.json)",
            }
            if leg4_ctx is not None:       # configs[4] on the CPU: the same paced frames, one oracle call per frame, one thread
                g4, h4, q04, n4, w4, hz4 = leg4_ctx
                state = q04[0].copy()
                lat = []
                t_next = time.perf_counter()
                for t in range(n4):
                    while time.perf_counter() < t_next:
                        pass
                    t_next += 1.0 / hz4
                    a4 = time.perf_counter()
                    qq, _, _ = orc.retarget_streams(g4._model_blob, g4._taskset_blob, state[None], h4[:, t:t + 1], nthreads=1)
                    state = qq[0, 0]
                    lat.append(time.perf_counter() - a4)
                lat = np.array(lat[w4:]) * 1e3
                out["cpu_baseline"]["config4_latency"] = {"p50_ms": float(np.percentile(lat, 50)), "p95_ms": float(np.percentile(lat, 95)),
                                                          "max_ms": float(lat.max()), "threads": 1,
                                                          "sample": f"the {n4 - w4} paced frames of legs.config4_latency, one oracle call per frame"}
            out["max_joint_err_rad"] = float(np.abs(q_hip[..., 7:] - q_cpu[..., 7:]).max())
            out["max_root_pos_err_m"] = float(np.abs(q_hip[..., :3] - q_cpu[..., :3]).max())
            out["frames_with_different_solve_count"] = int((ns_hip != ns_cpu).any(axis=-1).sum())
        print(json.dumps(out), flush=True)

    comm.barrier()
    comm.close()


if __name__ == "__main__":
    main()
