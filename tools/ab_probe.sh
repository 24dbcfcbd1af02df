#!/bin/bash
# A/B of two library builds: tools/ab_probe.sh libA.so libB.so   (throughput shape, latency shape, bench line)
R=$PWD
for lib in "$@"; do
  echo "== $lib"
  GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/$lib timeout -k 10 200 python tools/wide_probe.py 16384 16 1 5 2>/dev/null | cut -c1-200
  GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/$lib timeout -k 10 200 python tools/wide_probe.py 100 100 4 10 2>/dev/null | cut -c1-260
done
echo done
