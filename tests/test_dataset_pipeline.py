"""The dataset pipeline without a GPU: loaders and the retargeting step are stand-ins; what is
tested is the machinery around them -- batching by a frames budget, largest-first order, skip-if-exists, a loader that
fails, deterministic LPT sharding over ranks, loading / retargeting / writing overlapped, every file written once."""
import os
import pickle
import threading
import time

import numpy as np
import pytest

from general_motion_retargeting_amd import dataset


def _md(n):
    return {"fps": 30, "root_pos": np.zeros((n, 3)), "root_rot": np.zeros((n, 4)), "dof_pos": np.full((n, 29), float(n)),
            "local_body_pos": np.zeros((n, 38, 3), np.float32), "link_body_list": ["pelvis"]}


def _tree(tmp_path, lengths):
    src = tmp_path / "src"
    for i, n in enumerate(lengths):
        f = src / f"d{i % 3}" / f"clip_{i:04d}.bvh"
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text("HIERARCHY\nROOT Hips\n{\n}\nMOTION\nFrames: %d\nFrame Time: 0.0333\n" % n)
    return src


def _length(path):
    return dataset.job_cost(path)


def test_batches_follow_the_frames_budget_and_largest_first_order(tmp_path):
    rng = np.random.default_rng(0)
    lengths = rng.integers(5, 200, size=120).tolist()
    src, tgt = _tree(tmp_path, lengths), tmp_path / "out"
    batches = []

    def retarget(clips, files):
        batches.append([len(c) for c in clips])
        return [_md(len(c)) for c in clips]

    stats = {}
    n = dataset.run_bvh_dataset(str(src), str(tgt), "unitree_g1", retarget=retarget, load=lambda f: np.zeros((_length(f), 15, 7)),
                                frames_budget=2000, loader_workers=3, verbose=False, stats=stats)
    assert n == 120 and stats["clips"] == 120 and stats["frames"] == sum(lengths) and stats["batches"] == len(batches)
    flat = [x for b in batches for x in b]
    assert flat == sorted(lengths, reverse=True)                         # largest first: little padding in a ragged batch
    assert all(len(b) * max(b) <= 2000 or len(b) == 1 for b in batches)   # the padded batch stays within the budget
    assert len(batches) > 3
    for i, nfr in enumerate(lengths):                                    # every file written once, with its own clip
        with open(tgt / f"d{i % 3}" / f"clip_{i:04d}.pkl", "rb") as f:
            assert pickle.load(f)["dof_pos"].shape == (nfr, 29)
    # second run: nothing left to do
    batches.clear()
    assert dataset.run_bvh_dataset(str(src), str(tgt), "unitree_g1", retarget=retarget, load=lambda f: np.zeros((3, 15, 7)),
                                   verbose=False) == 0 and not batches


def test_sharding_over_ranks_is_deterministic_and_complete(tmp_path):
    rng = np.random.default_rng(1)
    lengths = rng.integers(10, 3000, size=77).tolist()                    # LAFAN1's shape
    src = _tree(tmp_path, lengths)
    jobs = dataset.list_bvh_jobs(str(src), str(tmp_path / "out"), False, False)
    world = 8
    parts = [dataset.shard_jobs(jobs, r, world) for r in range(world)]
    assert [dataset.shard_jobs(jobs, r, world) for r in range(world)] == parts          # same partition whoever computes it
    allf = [s for p in parts for s, _ in p]
    assert sorted(allf) == sorted(s for s, _ in jobs) and len(set(allf)) == len(jobs)    # every file on exactly one rank
    loads = [sum(_length(s) for s, _ in p) for p in parts]
    assert max(loads) - min(loads) <= max(lengths)                                       # LPT balance
    for p in parts:
        c = [_length(s) for s, _ in p]
        assert c == sorted(c, reverse=True)
    # ranks write disjoint files: run all eight "ranks" one after the other into one target folder
    written = 0
    for r in range(world):
        written += dataset.run_bvh_dataset(str(src), str(tmp_path / "out"), "unitree_g1", retarget=lambda c, f: [_md(len(x)) for x in c],
                                           load=lambda f: np.zeros((_length(f), 15, 7)), rank=r, world=world, verbose=False,
                                           loader_workers=0)
    assert written == 77 and len(list((tmp_path / "out").rglob("*.pkl"))) == 77


def test_a_failing_loader_is_printed_and_skipped_and_a_failing_writer_raises(tmp_path, capsys):
    src = _tree(tmp_path, [10, 20, 30, 40])

    def load(f):
        if f.endswith("clip_0002.bvh"):
            raise ValueError("truncated file")
        return np.zeros((_length(f), 15, 7))

    n = dataset.run_bvh_dataset(str(src), str(tmp_path / "out"), "unitree_g1", retarget=lambda c, f: [_md(len(x)) for x in c], load=load,
                                verbose=False, loader_workers=2)
    assert n == 3 and "Error loading" in capsys.readouterr().out and not (tmp_path / "out" / "d2" / "clip_0002.pkl").exists()
    (tmp_path / "ro").mkdir()
    (tmp_path / "ro" / "d0").write_text("a file where a folder must be created")
    with pytest.raises(Exception):
        dataset.run_bvh_dataset(str(src), str(tmp_path / "ro"), "unitree_g1", retarget=lambda c, f: [_md(len(x)) for x in c],
                                load=lambda f: np.zeros((2, 15, 7)), verbose=False, loader_workers=0)


def test_loading_retargeting_and_writing_overlap(tmp_path):
    """While batch k is "on the GPU" the loader pool keeps working on later files, and the writers on earlier ones: some
    load runs strictly inside a retarget interval, and the whole run is shorter than the sum of its three stages."""
    src = _tree(tmp_path, [50] * 48)
    loads, gpu = [], []
    lock = threading.Lock()

    def load(f):
        t0 = time.perf_counter()
        time.sleep(0.05)
        with lock:
            loads.append((t0, time.perf_counter()))
        return np.zeros((50, 15, 7))

    def retarget(clips, files):
        t0 = time.perf_counter()
        time.sleep(0.1)
        gpu.append((t0, time.perf_counter()))
        return [_md(50) for _ in clips]

    pool = dataset._loader_pool("thread", 4)
    p = dataset.DatasetPipeline(load, retarget, len, dataset.BVH_KEYS, frames_budget=400, loader=pool, prefetch_clips=32, verbose=False)
    jobs = dataset.list_bvh_jobs(str(src), str(tmp_path / "out"), False, False)
    t0 = time.perf_counter()
    assert p.run(jobs) == 48
    dt = time.perf_counter() - t0
    pool.shutdown()
    assert p.stats["batches"] == 6 and len(gpu) == 6
    inside = sum(1 for a, b in loads for g0, g1 in gpu if a > g0 and b < g1)
    assert inside > 0, "no load ran while a batch was being retargeted"
    assert dt < 48 * 0.05 / 4 + 6 * 0.1 - 0.05, dt                        # shorter than loading, then retargeting


def test_cli_launches_its_own_ranks_and_shards_the_folder(tmp_path):
    """`python -m ...dataset --gpus 2` without a launcher: two rank processes, the folder LPT-sharded, one summary line from
    rank 0.  (No GPU here: the run stops at the communicator's device requirement -- what is checked is that the launcher
    starts the ranks and reports the failure of a rank cleanly instead of hanging.)"""
    import subprocess
    import sys
    src = _tree(tmp_path, [5, 6, 7])
    env = dict(os.environ, GMR_COMM_TIMEOUT="20", GMR_DATASET_TIMEOUT="120")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "general_motion_retargeting_amd.dataset", "--source", "bvh", "--src_folder", str(src),
                        "--tgt_folder", str(tmp_path / "out"), "--gpus", "2"], cwd=root, env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode != 0 and "[launcher] rank" in r.stderr and "no HIP device" in r.stderr
