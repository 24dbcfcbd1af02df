#!/bin/bash
R=$PWD
O=$R/gpurun_out/r02_v2
timeout -k 10 900 python tools/soak_check.py > $O/soak_all_configs.json 2>$O/soak.err; echo "soak rc=$?"; tail -5 $O/soak.err
timeout -k 10 600 python tools/stress_bounds.py > $O/stress_bounds.json 2>$O/stress.err; echo "stress rc=$?"; tail -5 $O/stress.err
timeout -k 10 600 python tools/configs_3_4.py > $O/configs_3_4.json 2>$O/configs.err; echo "configs rc=$?"; tail -5 $O/configs.err
for st in "8192 16" "16384 16" "32768 16" "65536 16" "32768 4" "262144 4" "2048 64" "16384 64" "4096 256"; do
  timeout -k 10 300 python tools/wide_probe.py $st 1 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['S'],d['T'],round(d['frames_per_s']), d['ms'])"
done
echo done
