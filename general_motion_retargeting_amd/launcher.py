"""One rank per GPU without an external launcher: ``python bench.py --gpus N`` and ``python -m
general_motion_retargeting_amd.dataset --gpus N`` become, when ``WORLD_SIZE`` is unset, a LAUNCHER process that starts N rank
processes of the same program (``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 /`` two free ports), relays rank 0's
stdout and exits non-zero as soon as any rank does (the others are terminated by process group).  The launcher never loads
libgmrhip.so and never initialises a GPU.  The reference's counterpart is ``mp.Pool(args.num_cpus)`` over files
(``scripts/smplx_to_robot_dataset.py:241-242``)."""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import List, Sequence


def free_ports(n: int) -> List[int]:
    socks, ports = [], []
    for _ in range(n):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        socks.append(s)
        ports.append(s.getsockname()[1])
    for s in socks:
        s.close()
    return ports


def is_rank_process() -> bool:
    """True inside a process some launcher (ours, torchrun) started as one rank of a job."""
    return "WORLD_SIZE" in os.environ


def self_launch(n: int, command: Sequence[str], timeout_env: str = "GMR_BENCH_TIMEOUT", default_timeout: float = 1500.0,
                require_stdout_prefix: str = "", relay_all_stderr: bool = True) -> int:
    """Start ``n`` rank processes of ``command`` (an argv list); returns the exit code of the job.
    ``require_stdout_prefix``: rank 0 must print a line starting with it (bench.py: "{") or the job counts as failed."""
    port, comm_port = free_ports(2)
    timeout = float(os.environ.get(timeout_env, str(default_timeout)))
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GMR_COMM_PORT=str(comm_port), GMR_SELF_LAUNCHED="1", GMR_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on this driver
        env.setdefault("GMR_COMM_TIMEOUT", "120")
        procs.append(subprocess.Popen(list(command), env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      start_new_session=True))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)        # each rank is its own session: its worker processes go with it
                except OSError:
                    pass
        t_kill = time.time() + 5.0
        for p in procs:
            try:
                p.wait(max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass

    lines: List[str] = []

    def relay():
        for raw in procs[0].stdout:
            line = raw.decode(errors="replace")
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    t_end = time.time() + timeout
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"[launcher] rank {r} exited with code {c}: stopping the other ranks", file=sys.stderr, flush=True)
                rc = c if c > 0 else 1
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > t_end:
                print(f"[launcher] no result within {timeout_env}={timeout:.0f} s: stopping all ranks", file=sys.stderr, flush=True)
                rc = 124
                break
            time.sleep(0.05)
    finally:
        stop_all()
    th.join(5.0)
    if rc == 0 and require_stdout_prefix and not any(ln.lstrip().startswith(require_stdout_prefix) for ln in lines):
        print(f"[launcher] every rank exited 0 but rank 0 printed no line starting with {require_stdout_prefix!r}", file=sys.stderr, flush=True)
        rc = 1
    return rc
