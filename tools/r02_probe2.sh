#!/bin/bash
# round-2 probe 2: the throughput-shape kernel (gmr_ik_wide.hip): parity, A/B against the one-wavefront kernel
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -25 $O/pytest_gpu.log
[ $rc -ne 0 ] && echo "PYTEST FAILED rc=$rc"
for S in 16384 4096 1024; do
  GMR_IK_NO_WIDE=1 timeout -k 10 200 python tools/wide_probe.py $S 16 1 5 >> $O/wide_ab.txt 2>>$O/wide_ab.err || exit 3
  timeout -k 10 200 python tools/wide_probe.py $S 16 1 5 >> $O/wide_ab.txt 2>>$O/wide_ab.err || exit 3
done
cat $O/wide_ab.txt
echo done
