#!/bin/bash
# One GPU call that refreshes the judged artifacts of round 3:  tools/profile_round3.sh v1 [AB]
# (run through gpurun from the repo root; writes under gpurun_out/r03_<tag>/, publish with tools/publish_profiles.py --round r03 <tag>)
set -o pipefail
V=${1:-vX}
PART=${2:-AB}
R=$PWD
O=$R/gpurun_out/r03_$V
mkdir -p $O
SQ1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
SQ3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64"
SQ4="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"          # lane utilisation = THREAD_CYCLES_VALU / (64 * ACTIVE_INST_VALU)
if [[ $PART == *A* ]]; then
echo "[1] bench (the driver's command)"; timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
echo "[2] kernel stats of the bench command (headline leg + the 1M-frame leg; the extra legs are measured by [1])"
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/bench_stats -o run --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $O/bench_stats.log 2>&1 || exit 2
echo "[3] counters of the headline kernel (bench.py, configs[1] leg only)"
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:$SQ1" "sq2:$SQ2" "sq3:$SQ3"; do
  name=${grp%%:*}; ctr=${grp#*:}; echo "  pmc head $name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_head_$name -o runc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong --no-extra-legs > $O/pmc_head_$name.log 2>&1 || exit 3
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ4 -d $O/pmc_head_sq4 -o runc --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong --no-extra-legs > $O/pmc_head_sq4.log 2>&1 || echo "  (lane-utilisation counters not available for the headline kernel)"
echo "[4] counters of the throughput kernel (S=16384 x T=16)"
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:$SQ1" "sq2:$SQ2" "sq3:$SQ3"; do
  name=${grp%%:*}; ctr=${grp#*:}; echo "  pmc wide $name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_wide_$name -o runc --output-format csv -- python3 $R/tools/wide_probe.py 16384 16 1 3 > $O/pmc_wide_$name.log 2>&1 || exit 4
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SQ4 -d $O/pmc_wide_sq4 -o runc --output-format csv -- python3 $R/tools/wide_probe.py 16384 16 1 3 > $O/pmc_wide_sq4.log 2>&1 || echo "  (lane-utilisation counters not available for the throughput kernel)"
echo "[5] the kernels beside the loop: stats + counters"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/aux_stats -o run --output-format csv -- python3 $R/tools/aux_kernels.py 5 > $O/aux_kernels.json 2>$O/aux_stats.log || exit 5
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
  name=${grp%%:*}; ctr=${grp#*:}; echo "  pmc aux $name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_aux_$name -o runc --output-format csv -- python3 $R/tools/aux_kernels.py 3 > $O/pmc_aux_$name.log 2>&1 || exit 5
done
cd $R
python tools/collect_counters.py ik_streams_kernel $O/pmc_head_fetch $O/pmc_head_write $O/pmc_head_sq $O/pmc_head_sq2 $O/pmc_head_sq3 $O/pmc_head_sq4 > $O/counters_headline.json || exit 6
python tools/collect_counters.py ik_wide_kernel $O/pmc_wide_fetch $O/pmc_wide_write $O/pmc_wide_sq $O/pmc_wide_sq2 $O/pmc_wide_sq3 $O/pmc_wide_sq4 > $O/counters_wide.json || exit 6
python tools/collect_counters.py "" $O/pmc_aux_fetch $O/pmc_aux_write $O/pmc_aux_sq > $O/counters_aux.json || exit 6
echo "[6] phase shares"
timeout -k 10 300 python tools/phase_profile.py 100 100 > $O/phase_shares.txt 2>$O/phase_shares.err || exit 7
timeout -k 10 300 python tools/phase_profile.py 16384 16 > $O/phase_shares_wide.txt 2>$O/phase_shares_wide.err || exit 7
fi
if [[ $PART == *B* ]]; then
echo "[7] extras, latency, launch-shape crossover"
timeout -k 10 600 python tools/measure_extras.py > $O/extras.json 2>/dev/null || exit 8
timeout -k 10 300 python tools/latency_probe.py > $O/latency_config5.json 2>/dev/null || exit 9
timeout -k 10 600 python tools/shape_sweep.py > $O/shape_sweep.txt 2>/dev/null || exit 10
echo "[8] parity evidence at HEAD"
timeout -k 10 900 python tools/soak_check.py > $O/soak_all_configs.json 2>/dev/null || exit 11
timeout -k 10 600 python tools/stress_bounds.py > $O/stress_bounds.json 2>/dev/null || exit 12
timeout -k 10 600 python tools/configs_3_4.py > $O/configs_3_4.json 2>/dev/null || exit 13
timeout -k 10 400 python tools/dataset_probe.py 2400 > $O/dataset_probe.json 2>/dev/null || exit 14
fi
rm -rf $O/pmc_*/ $O/bench_stats/*agent* 2>/dev/null
echo done
