#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p10
mkdir -p $O
GMR_FK_LISTS=1 timeout -k 10 300 python tools/fk_bitcheck.py $O/fk_1.npz || exit 2
cp $O/fk_1.npz $O/fk_4.npz
echo "lists 1 (dof parked) vs 4 (split, strided):"; python tools/fk_bitcheck.py --compare $O/fk_1.npz $O/fk_4.npz
rm -f $O/fk_*.npz
for k in 1; do
  for m in pos posrot; do echo "lists=$k"; GMR_FK_LISTS=$k timeout -k 10 120 python tools/fk_only.py $m 2>/dev/null; done
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fk or kinematics or dataset or clip or bvh or smplx" > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && echo "PYTEST FAILED rc=$rc" && tail -40 $O/pytest_gpu.log
echo done
