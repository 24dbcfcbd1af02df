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
