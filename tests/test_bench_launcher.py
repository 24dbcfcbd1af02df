"""`python bench.py --gpus N` without an external launcher: the parent starts N rank processes,
relays rank 0's ONE JSON line and fails fast.  No GPU here: the ranks run tests/bench_standin.py as their compute
function and talk over the library's TCP control star ("tcp") or a gloo process group; the communicator, the one
broadcast, the LPT shards, the barriers and the JSON fields are the real ones."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STANDIN = os.path.join(ROOT, "tests", "bench_standin.py")


def _run(args, backend, extra_env=None, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GMR_COMM_PORT"):
        env.pop(k, None)
    env.update(GMR_BENCH_STANDIN=STANDIN, GMR_BENCH_BACKEND=backend, GMR_COMM_TIMEOUT="60")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


SMALL = ["--steps", "2", "--warmup", "1", "--streams", "6", "--frames", "5", "--strong-streams", "64", "--strong-frames", "4",
         "--no-cpu-baseline"]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["tcp", "gloo"])
def test_self_launch_two_ranks(backend):
    r = _run(["--gpus", "2"] + SMALL, backend)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["world_size_launcher"] == 2
    assert out["comm_backend"] == ("tcp" if backend == "tcp" else "torch-gloo")
    assert out["launcher"].startswith("self")
    assert out["scaling"] == "strong" and out["steps"] == 2 and out["warmup"] == 1
    assert sum(out["frames_per_rank"]) == 64 * 4 and len(out["per_rank_ms_per_step"]) == 2
    assert out["value"] > 0 and out["value_1gpu"] > 0
    assert abs(out["efficiency"] - out["value"] / (2 * out["value_1gpu"])) < 1e-9
    assert "strong_1m.value" in out["config"]["workload"]            # the N = 1 anchor of the curve is named
    assert out["weak_leg"]["value"] > 0 and "standin" in out


@pytest.mark.timeout(300)
def test_self_launch_fails_fast_when_a_rank_dies():
    t0 = time.time()
    r = _run(["--gpus", "2"] + SMALL, "tcp", {"GMR_STANDIN_FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert "rank 1 exited with code 7" in r.stderr and "[launcher]" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 120                                    # the surviving rank was stopped, not waited for


@pytest.mark.timeout(300)
def test_single_rank_line_is_the_configs1_line():
    r = _run(["--gpus", "1"] + SMALL, "tcp")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["scaling"] == "weak" and "configs[1]" in out["config"]["workload"]
    assert "strong_1m" in out and "world_size" not in out


@pytest.mark.timeout(120)
def test_rccl_failure_is_a_clean_error_on_every_rank(tmp_path):
    """no silent per-rank fallback.  With RCCL unavailable (no such library) both ranks report the same
    GmrHipError naming the failing rank; with GMR_COMM_FALLBACK=tcp both continue on the control star, labelled."""
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from general_motion_retargeting_amd import comm\n"
        "try:\n"
        "    c = comm.create('rccl')\n"
        "    print('BACKEND', c.backend); c.barrier(); print('MAX', c.allreduce_max(float(c.rank))); c.close()\n"
        "except Exception as e:\n"
        "    print('ERROR', e); sys.exit(5)\n")
    from test_distributed_gloo import _free_port
    for fallback, want_rc in (("none", 5), ("tcp", 0)):
        port = _free_port()
        procs = []
        for rank in range(2):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), GMR_RCCL_LIBRARY="/nonexistent/librccl.so", GMR_COMM_FALLBACK=fallback,
                       GMR_COMM_TIMEOUT="30", LD_LIBRARY_PATH="")
            env.pop("GMR_COMM_PORT", None)
            procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
        outs = [p.communicate(timeout=90) for p in procs]
        for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
            assert p.returncode == want_rc, (fallback, rank, so, se)
            if want_rc:
                assert "RCCL bring-up failed on rank" in so, so
            else:
                assert "BACKEND tcp (fallback: RCCL bring-up failed on rank" in so and "MAX 1.0" in so, so


@pytest.mark.timeout(300)
def test_external_launcher_route_still_works():
    """The driver's N > 1 form: `python -m torch.distributed.run ... bench.py --gpus 2` (the launcher provides RANK /
    WORLD_SIZE / MASTER_*; the ranks' control star sits at MASTER_PORT + 1)."""
    from test_distributed_gloo import _free_port
    env = dict(os.environ, GMR_BENCH_STANDIN=STANDIN, GMR_BENCH_BACKEND="tcp", GMR_COMM_TIMEOUT="60")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GMR_COMM_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL,
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["world_size"] == 2 and out["launcher"].startswith("external") and out["comm_backend"] == "tcp"
