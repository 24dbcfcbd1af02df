#!/bin/bash
# One GPU call that refreshes every judged artifact of a kernel version:  tools/profile_round.sh v6
# (run through gpurun from the repo root; writes under gpurun_out/, copy the summaries into profiles/)
set -o pipefail
V=${1:-vX}
R=$PWD
O=$R/gpurun_out
mkdir -p $O
timeout -k 10 600 python bench.py > $O/bench_$V.json 2> $O/bench_$V.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$V -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_$V.log 2>&1 || exit 2
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  name=${grp%%:*}; ctr=${grp#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${V}_$name -o runc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_${V}_$name.log 2>&1 || exit 3
done
cd $R
python tools/collect_traffic.py $V $O/pmc_${V}_fetch $O/pmc_${V}_write $O/pmc_${V}_sq > $O/traffic_$V.json || exit 4
python tools/phase_profile.py > $O/phase_$V.txt 2>/dev/null || exit 5
python tools/latency_probe.py > $O/latency_$V.json 2>/dev/null || exit 6
python tools/shape_sweep.py > $O/shape_sweep_$V.txt 2>/dev/null || exit 7
echo done
