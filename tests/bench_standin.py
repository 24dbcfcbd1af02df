"""Stand-in for the HIP side of bench.py (GMR_BENCH_STANDIN): lets the CPU suite walk bench.py's launcher and its
N > 1 protocol (rank processes, communicator, one broadcast, LPT shards, barriers, the JSON line) where no GPU exists.
Test infrastructure: a step costs a time proportional to its frames, results are zeros of the right shapes.
GMR_STANDIN_FAIL_RANK=r makes rank r die right after start-up (the launcher must then stop the others)."""
import os
import sys
import time

import numpy as np

SECONDS_PER_FRAME = 2e-7


def init(local_rank):
    if os.environ.get("GMR_STANDIN_FAIL_RANK") == os.environ.get("RANK", "0"):
        print("stand-in: this rank fails on purpose", file=sys.stderr, flush=True)
        raise SystemExit(7)


def device_sync():
    pass


class Solver:
    def __init__(self, model_blob, taskset_blob):
        self.nq = int(model_blob["nq"][0])
        self.nhuman = int(taskset_blob["nhuman"][0])


class Event:
    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_ms(self, stop):
        return (stop.t - self.t) * 1e3


class Shard:
    def __init__(self, solver, q0, human):
        self.solver = solver
        self.S, self.T = human.shape[0], human.shape[1]

    def step(self, ev0=None, ev1=None):
        if ev0 is not None:
            ev0.record()
        time.sleep(self.S * self.T * SECONDS_PER_FRAME)
        if ev1 is not None:
            ev1.record()

    def results(self):
        return (np.zeros((self.S, self.T, self.solver.nq)), np.ones((self.S, self.T, 2), dtype=np.int32),
                np.zeros(self.S, dtype=np.int32))

    def free(self):
        pass
