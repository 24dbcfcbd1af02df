#!/usr/bin/env python3
"""Average every counter of the rocprofv3 --pmc passes found under the given directories, per kernel:

    python tools/collect_counters.py <kernel substring> <dir> [<dir> ...]   -> JSON on stdout

FETCH_SIZE / WRITE_SIZE are KiB on gfx950; FETCH_SIZE reports half of the bytes of wide coalesced reads
(MI355X_MICROARCH.md, HBM section): `hbm_bytes_per_launch` = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024."""
import csv
import glob
import json
import os
import sys

sub, dirs = sys.argv[1], sys.argv[2:]
acc, meta, durs = {}, {}, {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if sub not in r["Kernel_Name"]:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                meta[k] = {"VGPR_Count": int(r["VGPR_Count"]), "Accum_VGPR_Count": int(r["Accum_VGPR_Count"]),
                           "SGPR_Count": int(r["SGPR_Count"]), "Workgroup_Size": int(r["Workgroup_Size"]),
                           "LDS_Block_Size": int(r["LDS_Block_Size"]), "Scratch_Size": int(r.get("Scratch_Size") or 0),
                           "Grid_Size": int(r["Grid_Size"])}
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if sub in r["Kernel_Name"]:
                    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                    durs.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
out = {}
for k, cs in acc.items():
    o = dict(meta[k])
    o["note_registers"] = ("rocprofv3's VGPR_Count is HALF of the kernel's unified register count (arch VGPRs + AGPRs; code-object metadata .vgpr_count), rounded up to a granule of 8: ik_streams_kernel<36,4,1> 331 -> 168, ik_wide_kernel 224 -> 112; Accum_VGPR_Count is reported as 0 on gfx950. The compiler's own numbers per kernel (VGPRs, AGPRs, spills, occupancy) are in profiles/r02_v2_kernel_resource_usage.txt")
    for c, v in sorted(cs.items()):
        o[c + "_mean_per_launch"] = sum(v) / len(v)
    if k in durs:
        o["mean_ms_under_pmc"] = sum(durs[k]) / len(durs[k])
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rd = o["FETCH_SIZE_mean_per_launch"] * 1024
        wr = o["WRITE_SIZE_mean_per_launch"] * 1024
        o["hbm_read_bytes_raw"] = rd
        o["hbm_write_bytes"] = wr
        o["hbm_bytes_per_launch"] = 2 * rd + wr
    out[k] = o
print(json.dumps(out, indent=1))
