#!/bin/bash
R=$PWD
O=$R/gpurun_out/r02p15
mkdir -p $O
GMR_BENCH_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 timeout -k 10 600 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_rccl.json 2>$O/bench_rccl.err; echo "bench rccl rc=$?"; head -c 120 $O/bench_rccl.json; echo; python -c "
import json; d=json.load(open('$O/bench_rccl.json')); print(d['value'], d['comm_backend'], d['world_size'], d['weak_leg']['value'])"; grep -i "version\|librccl" $O/bench_rccl.err
