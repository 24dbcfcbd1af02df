"""N>1 path on CPU: two gloo ranks, one broadcast of the packed (model, task set), LPT sharding of
ragged streams, shard results reassembled == single-rank result bit for bit.  The compute function
is the oracle here (no GPU in this suite); on the GPU box the same driver runs the HIP solver."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import get_setup
    from general_motion_retargeting_amd import synth
    from general_motion_retargeting_amd.sharding import broadcast_blobs, run_sharded
    from oracle import oracle
    su = get_setup()
    # only rank 0 "owns" the compiled robot; the others receive it through the one broadcast
    mb, ts = broadcast_blobs(su.mb if rank == 0 else None, su.ts if rank == 0 else None, rank, dist)
    assert np.array_equal(mb.view(np.uint8), su.mb.view(np.uint8)) and np.array_equal(ts.view(np.uint8), su.ts.view(np.uint8))
    S, T = 7, 9
    human, q0 = synth.make_streams(su.model, su.tt, S, T, seed=100)
    lens = np.array([9, 3, 7, 1, 9, 5, 2], dtype=np.int32)

    def compute(q0_l, human_l, lens_l):
        outs = []
        for i in range(len(lens_l)):
            q, ns, st = oracle.retarget_streams(mb, ts, q0_l[i:i + 1], human_l[i:i + 1, : lens_l[i]])
            outs.append((q[0], ns[0], int(st[0])))
        return outs

    ids, res = run_sharded(compute, q0, human, lens, rank, world)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), ids=np.array(ids, dtype=np.int64),
             **{f"q{i}": r[0] for i, r in zip(ids, res or [])}, **{f"n{i}": r[1] for i, r in zip(ids, res or [])})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_sharding_equals_single_rank(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import get_setup
    from general_motion_retargeting_amd import synth
    from oracle import oracle
    su = get_setup()
    S, T = 7, 9
    human, q0 = synth.make_streams(su.model, su.tt, S, T, seed=100)
    lens = [9, 3, 7, 1, 9, 5, 2]
    seen = set()
    for r in range(world):
        z = np.load(tmp_path / f"rank{r}.npz")
        for i in z["ids"]:
            i = int(i)
            assert i not in seen
            seen.add(i)
            q, ns, st = oracle.retarget_streams(su.mb, su.ts, q0[i:i + 1], human[i:i + 1, : lens[i]])
            assert np.array_equal(z[f"q{i}"], q[0]) and np.array_equal(z[f"n{i}"], ns[0])
    assert seen == set(range(S))


# ----------------------------------------------------------------------------------------------
# the library's own communicator: bootstrap half (plain TCP, no GPU) -- rank 0 hands the 128-byte
# ncclUniqueId to its peers at MASTER_ADDR : port
# ----------------------------------------------------------------------------------------------
def _bootstrap_worker(rank, world, port, outdir):
    import ctypes as C
    sys.path.insert(0, ROOT)
    from general_motion_retargeting_amd import _lib
    L = _lib.lib()
    buf = (C.c_ubyte * 128)()
    if rank == 0:
        for i in range(128):
            buf[i] = (7 * i + 3) % 251
    else:
        import time
        time.sleep(0.3 * rank)                       # peers may come up before or after rank 0 listens
    rc = L.gmr_bootstrap_exchange(rank, world, b"127.0.0.1", port, buf, 128, 30.0)
    with open(os.path.join(outdir, f"boot{rank}.txt"), "w") as f:
        f.write(f"{rc} " + " ".join(str(b) for b in buf))


@pytest.mark.timeout(120)
def test_bootstrap_exchange_over_tcp(tmp_path):
    import torch.multiprocessing as mp
    world, port = 3, _free_port()
    mp.spawn(_bootstrap_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = [(7 * i + 3) % 251 for i in range(128)]
    for r in range(world):
        vals = open(tmp_path / f"boot{r}.txt").read().split()
        assert int(vals[0]) == 0 and [int(v) for v in vals[1:]] == want, r


def test_bootstrap_reports_an_unreachable_rank0():
    import ctypes as C
    sys.path.insert(0, ROOT)
    from general_motion_retargeting_amd import _lib
    L = _lib.lib()
    buf = (C.c_ubyte * 16)()
    rc = L.gmr_bootstrap_exchange(1, 2, b"127.0.0.1", _free_port(), buf, 16, 0.5)
    assert rc != 0 and b"could not reach rank 0" in L.gmr_last_error()
    assert L.gmr_bootstrap_exchange(0, 1, b"127.0.0.1", 1, buf, 16, 0.5) == 0      # a single rank needs no exchange


# ----------------------------------------------------------------------------------------------
# bench.py's N > 1 leg on CPU: the strong-scaling driver over two gloo ranks, the oracle as the compute function
# ----------------------------------------------------------------------------------------------
def _strong_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    from conftest import get_setup
    from general_motion_retargeting_amd import comm as gcomm, sharding, synth
    from oracle import oracle
    comm = gcomm.create("gloo")
    assert comm.world == world and comm.rank == rank and comm.backend == "torch-gloo"
    su = get_setup()
    mb, ts = sharding.broadcast_blobs(su.mb if rank == 0 else None, su.ts if rank == 0 else None, rank, comm)
    assert np.array_equal(mb.view(np.uint8), su.mb.view(np.uint8)) and np.array_equal(ts.view(np.uint8), su.ts.view(np.uint8))
    S, T, seed = 10, 4, 77
    lens = np.full(S, T)
    ids = sharding.lpt_partition(lens, world)[rank]
    # a rank generates exactly its streams of the whole batch (same seeds for every N)
    human, q0 = synth.make_streams_ids(su.model, su.tt, ids, T, seed=seed)
    got = {}

    def make_step(my_ids):
        assert list(my_ids) == list(ids)

        def step():
            got["mine"] = oracle.retarget_streams(mb, ts, q0, human)
        return step

    def single_step():
        h_all, q_all = synth.make_streams(su.model, su.tt, S, T, seed=seed)

        def step():
            got["all"] = oracle.retarget_streams(mb, ts, q_all, h_all)
        return step

    res = sharding.strong_scaling_leg(comm, lens, make_step, 2, 1, single_step, device_sync=lambda: None)
    assert res["world_size"] == world and res["frames"] == S * T and sum(res["frames_per_rank"]) == S * T
    assert len(res["per_rank_seconds"]) == world and res["seconds"] == max(res["per_rank_seconds"])
    assert res["value"] > 0 and res["value_1gpu"] > 0 and abs(res["efficiency"] - res["value"] / (world * res["value_1gpu"])) < 1e-12
    np.savez(os.path.join(outdir, f"strong{rank}.npz"), ids=np.array(ids), q=got["mine"][0],
             q_all=got["all"][0] if rank == 0 else np.zeros(0))
    comm.barrier()
    comm.close()


@pytest.mark.timeout(300)
def test_two_rank_gloo_strong_scaling_leg(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_strong_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    z0, z1 = np.load(tmp_path / "strong0.npz"), np.load(tmp_path / "strong1.npz")
    q_all = z0["q_all"]
    assert sorted(list(z0["ids"]) + list(z1["ids"])) == list(range(10))
    for z in (z0, z1):                      # the shards reassemble the one-GPU result bit for bit
        assert np.array_equal(z["q"], q_all[z["ids"]])
