#!/bin/bash
for st in "16384 8" "32768 8" "65536 8" "131072 8"; do
  timeout -k 10 300 python tools/wide_probe.py $st 1 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['S'],d['T'],round(d['frames_per_s']), d['ms'])"
done
echo done
