#!/bin/bash
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GMR_FK_LISTS=1
for grp in "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  name=${grp%%:*}; ctr=${grp#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr -d $O/fk_$name -o runc --output-format csv -- python3 $R/tools/fk_only.py pos > $O/fk_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $O/fk_$name.log; }
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join("gpurun_out", "r02p11")
for d in sorted(glob.glob(os.path.join(O, "*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fk_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(os.path.basename(d), k, sum(v) / len(v), len(v))
PY
echo done
