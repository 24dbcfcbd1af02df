#!/bin/bash
# bench.py's N > 1 code path with two ranks on the one-GPU box: gloo for the job-level collectives (RCCL cannot put two
# ranks on one device), both ranks computing on GPU 0
R=$PWD
O=$R/gpurun_out/r02p14
mkdir -p $O
GMR_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 3 --warmup 1 > $O/bench_2rank_gloo.json 2>$O/bench_2rank_gloo.err; echo "rc=$?"; cat $O/bench_2rank_gloo.json; tail -5 $O/bench_2rank_gloo.err
