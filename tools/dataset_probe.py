#!/usr/bin/env python3
"""Dataset driver on a synthetic tree of BVH clips: wall clock of
``python -m general_motion_retargeting_amd.dataset --source bvh`` (loader pool || one IK launch per frames budget || writer
threads) against the round-2 driver (256 files per launch, read -> launch -> dump in one thread) on the same tree, and a
sample of the written files compared between the two.

    python tools/dataset_probe.py [nclips] [--old DIR_OF_AN_OLDER_TREE] [--gpus N]
"""
import json
import os
import pickle
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "synthetic.bvh")


def _write_clip(args):
    path, header, nframes, seed = args
    rng = np.random.default_rng(seed)
    nch = 6 + 3 * 21
    t = np.arange(nframes)[:, None] / 30.0
    f = rng.uniform(0.2, 1.2, size=(1, nch))
    ph = rng.uniform(0, 2 * np.pi, size=(1, nch))
    amp = rng.uniform(2.0, 25.0, size=(1, nch))
    data = amp * np.sin(2 * np.pi * f * t + ph)
    data[:, 0] = 30.0 * np.sin(0.3 * t[:, 0])            # root translation (cm)
    data[:, 1] = 92.0 + 2.0 * np.sin(2.0 * t[:, 0])
    data[:, 2] = 40.0 * t[:, 0]
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        fh.write(header)
        fh.write(f"MOTION\nFrames: {nframes}\nFrame Time: 0.033333\n")
        np.savetxt(fh, data, fmt="%.6f")
    return nframes


def make_tree(src, nclips, seed=0):
    import multiprocessing as mp
    header = open(GOLD).read().split("MOTION")[0]
    rng = np.random.default_rng(seed)
    lens = rng.integers(80, 420, size=nclips)
    jobs = [(os.path.join(src, f"subject{i % 7}", f"clip_{i:05d}.bvh"), header, int(n), seed + 1 + i) for i, n in enumerate(lens)]
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 4)) as pool:      # (this process never touches the GPU)
        frames = sum(pool.map(_write_clip, jobs, chunksize=16))
    return frames


def run_cli(cwd, src, tgt, extra=()):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "general_motion_retargeting_amd.dataset", "--source", "bvh", "--src_folder", src,
                        "--tgt_folder", tgt, "--robot", "unitree_g1"] + list(extra), cwd=cwd, env=env, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"CLI failed in {cwd}: {r.stderr[-2000:]}")
    summary = None
    for ln in r.stdout.splitlines():
        if ln.startswith('{"dataset_summary"'):
            summary = json.loads(ln)["dataset_summary"]
    return dt, summary


def main():
    args = sys.argv[1:]
    nclips = int(args[0]) if args and args[0].isdigit() else 2400
    old = args[args.index("--old") + 1] if "--old" in args else None
    gpus = int(args[args.index("--gpus") + 1]) if "--gpus" in args else 1
    base = os.environ.get("GMR_DS_PROBE_DIR", "/tmp/gmr_ds_probe")
    shutil.rmtree(base, ignore_errors=True)
    src = os.path.join(base, "src")
    t0 = time.perf_counter()
    frames = make_tree(src, nclips)
    out = {"clips": nclips, "frames": frames, "tree_seconds": time.perf_counter() - t0}
    dt, summary = run_cli(ROOT, src, os.path.join(base, "new"), ["--quiet"] + (["--gpus", str(gpus)] if gpus > 1 else []))
    out["new"] = {"wall_seconds": dt, "frames_per_s_wall": frames / dt, "summary": summary}
    n_new = sum(len(f) for _, _, f in os.walk(os.path.join(base, "new")))
    out["new"]["files_written"] = n_new
    if old:
        dt_old, _ = run_cli(os.path.abspath(old), src, os.path.join(base, "old"))
        out["old"] = {"tree": old, "wall_seconds": dt_old, "frames_per_s_wall": frames / dt_old,
                      "files_written": sum(len(f) for _, _, f in os.walk(os.path.join(base, "old")))}
        out["speedup_wall"] = dt_old / dt
        worst = 0.0
        rng = np.random.default_rng(0)
        for i in rng.integers(0, nclips, size=8):
            rel = os.path.join(f"subject{i % 7}", f"clip_{i:05d}.pkl")
            a = pickle.load(open(os.path.join(base, "new", rel), "rb"))
            b = pickle.load(open(os.path.join(base, "old", rel), "rb"))
            assert list(a) == list(b)
            for k in ("root_pos", "root_rot", "dof_pos", "local_body_pos"):
                worst = max(worst, float(np.abs(np.asarray(a[k], dtype=np.float64) - np.asarray(b[k], dtype=np.float64)).max()))
        out["max_abs_difference_new_vs_old_on_8_files"] = worst
    shutil.rmtree(base, ignore_errors=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
