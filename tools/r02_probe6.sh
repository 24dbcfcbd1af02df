#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p6
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench.json 2>$O/bench.err; echo "bench rc=$?"; cat $O/bench.json; tail -3 $O/bench.err
GMR_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 600 python bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench_rccl.json 2>$O/bench_rccl.err; echo "bench rccl rc=$?"; cat $O/bench_rccl.json; tail -5 $O/bench_rccl.err
GMR_BENCH_FORCE_DIST=1 GMR_BENCH_BACKEND=torch timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 1 --steps 3 --warmup 1 > $O/bench_torch.json 2>$O/bench_torch.err; echo "bench torch rc=$?"; cat $O/bench_torch.json; tail -5 $O/bench_torch.err
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_tr1.json 2>$O/bench_tr1.err; echo "bench torchrun-1 rc=$?"; cut -c1-300 $O/bench_tr1.json; tail -3 $O/bench_tr1.err
echo done
