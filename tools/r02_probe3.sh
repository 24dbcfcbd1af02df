#!/bin/bash
# round-2 probe 3: where does the throughput-shape kernel spend its time at two wavefronts per SIMD?
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p3
mkdir -p $O
timeout -k 10 300 python tools/phase_profile.py 16384 16 > $O/phase_16384.txt 2>$O/phase.err || exit 4
cat $O/phase_16384.txt
cd /tmp && export TMPDIR=/tmp
for grp in "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "sq3:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_FLAT" \
           "grbm:GRBM_GUI_ACTIVE"; do
  name=${grp%%:*}; ctr=${grp#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_$name -o runc --output-format csv -- python3 $R/tools/wide_probe.py 16384 16 1 3 > $O/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $O/pmc_$name.log; }
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join("gpurun_out", "r02p3")
for d in sorted(glob.glob(os.path.join(O, "pmc_*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ik_" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = (r["Kernel_Name"][:40], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r.get("Scratch_Size"))
    for k, v in sorted(acc.items()):
        print(os.path.basename(d), k, sum(v) / len(v), len(v))
    if acc: print(" meta", meta)
PY
echo done
