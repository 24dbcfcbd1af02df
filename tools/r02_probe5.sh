#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p5
mkdir -p $O
timeout -k 10 120 tools/micro/math_check > $O/math_check.txt 2>&1; echo "math_check rc=$?"; cat $O/math_check.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && echo "PYTEST FAILED rc=$rc" && tail -40 $O/pytest_gpu.log
for S in 16384 1024; do
  timeout -k 10 200 python tools/wide_probe.py $S 16 1 5 >> $O/wide_ab.txt 2>>$O/wide_ab.err || exit 3
done
timeout -k 10 200 python tools/wide_probe.py 100 100 4 10 >> $O/wide_ab.txt 2>>$O/wide_ab.err || exit 3
cat $O/wide_ab.txt
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; cat $O/bench.json | cut -c1-400
timeout -k 10 300 python tools/phase_profile.py 100 100 > $O/phase_100.txt 2>$O/phase.err; cat $O/phase_100.txt
timeout -k 10 300 python tools/phase_profile.py 16384 16 > $O/phase_16384.txt 2>$O/phase.err; cat $O/phase_16384.txt
echo done
