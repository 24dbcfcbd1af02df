#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p7
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && echo "PYTEST FAILED rc=$rc" && tail -40 $O/pytest_gpu.log
for v in fk_old fk_libm_poly; do
  GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/libgmrhip_$v.so timeout -k 10 300 python tools/fk_bitcheck.py $O/fk_$v.npz || exit 2
done
timeout -k 10 300 python tools/fk_bitcheck.py $O/fk_new.npz || exit 2
echo "old vs libm+poly:"; python tools/fk_bitcheck.py --compare $O/fk_fk_old.npz $O/fk_fk_libm_poly.npz
echo "old vs new:"; python tools/fk_bitcheck.py --compare $O/fk_fk_old.npz $O/fk_new.npz
rm -f $O/fk_*.npz
for v in libgmrhip_fk_old.so libgmrhip.so; do
  for m in pos posrot; do GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/$v timeout -k 10 120 python tools/fk_only.py $m; done
done
timeout -k 10 200 python tools/wide_probe.py 100 100 4 10
echo done
