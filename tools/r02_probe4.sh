#!/bin/bash
# round-2 probe 4: lean device math + no machine-LICM build of the IK kernels
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p4
mkdir -p $O
timeout -k 10 120 tools/micro/math_check > $O/math_check.txt 2>&1; echo "math_check rc=$?"; cat $O/math_check.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && echo "PYTEST FAILED rc=$rc" && tail -40 $O/pytest_gpu.log
for lib in libgmrhip.so libgmrhip_nolicm.so; do
  for S in 16384 4096 1024; do
    GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/$lib timeout -k 10 200 python tools/wide_probe.py $S 16 1 5 >> $O/wide_ab.txt 2>>$O/wide_ab.err || exit 3
  done
  GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/$lib timeout -k 10 200 python tools/wide_probe.py 100 100 4 10 >> $O/wide_ab.txt 2>>$O/wide_ab.err || exit 3
done
cat $O/wide_ab.txt
GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/libgmrhip_nolicm.so timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_nolicm.log 2>&1; rc=$?
tail -3 $O/pytest_gpu_nolicm.log
echo done
