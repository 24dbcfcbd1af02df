#!/bin/bash
# counters of the FK kernel (what stalls it?) and of the MFMA variant of the wide IK kernel
set -o pipefail
R=$PWD
O=$R/gpurun_out/r02p9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GMR_FK_LISTS=1
for grp in "sq1:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
           "sq2:SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "sq3:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
           "tcp:TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "fetch:FETCH_SIZE" "write:WRITE_SIZE" "grbm:GRBM_GUI_ACTIVE"; do
  name=${grp%%:*}; ctr=${grp#*:}
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctr -d $O/fk_$name -o runc --output-format csv -- python3 $R/tools/fk_only.py pos > $O/fk_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $O/fk_$name.log; }
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/fk_stats -o run --output-format csv -- python3 $R/tools/fk_only.py pos > $O/fk_stats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/fkrot_stats -o run --output-format csv -- python3 $R/tools/fk_only.py posrot > $O/fkrot_stats.log 2>&1
export GMR_HIP_LIBRARY=$R/general_motion_retargeting_amd/libgmrhip_mfma.so
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA -d $O/mfma -o runc --output-format csv -- python3 $R/tools/wide_probe.py 16384 16 1 3 > $O/mfma.log 2>&1 || { echo "pmc mfma failed"; tail -3 $O/mfma.log; }
unset GMR_HIP_LIBRARY
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join("gpurun_out", "r02p9")
for d in sorted(glob.glob(os.path.join(O, "*"))):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "fk_" in r["Kernel_Name"] or "ik_" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:28], r["Counter_Name"])].append(float(r["Counter_Value"]))
                meta = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r.get("Scratch_Size"))
    for k, v in sorted(acc.items()):
        print(os.path.basename(d), k, sum(v) / len(v), len(v))
    if acc: print(" meta vgpr/agpr/sgpr/lds/scratch", meta)
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        print(open(f).read())
PY
echo done
