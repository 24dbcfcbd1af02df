#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p8
mkdir -p $O
for k in 1 4; do GMR_FK_LISTS=$k timeout -k 10 300 python tools/fk_bitcheck.py $O/fk_$k.npz || exit 2; done
echo "lists 1 vs 4:"; python tools/fk_bitcheck.py --compare $O/fk_1.npz $O/fk_4.npz
rm -f $O/fk_*.npz
for k in 1 2 3 4; do
  for m in pos posrot; do echo "lists=$k"; GMR_FK_LISTS=$k timeout -k 10 120 python tools/fk_only.py $m 2>/dev/null; done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fk or kinematics or dataset or clip or bvh or smplx" > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && echo "PYTEST FAILED rc=$rc" && tail -40 $O/pytest_gpu.log
echo done
for lib in libgmrhip.so libgmrhip_mfma.so; do
  GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/$lib timeout -k 10 200 python tools/wide_probe.py 16384 16 1 5 || exit 3
done
GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/libgmrhip_mfma.so timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ik or shim or soak or full_size or streams" > $O/pytest_mfma.log 2>&1; tail -3 $O/pytest_mfma.log
echo done2
