#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 --pmc passes of bench.py (one directory per counter group):

    python tools/collect_traffic.py <tag> <dir> [<dir> ...]

Averages every counter over the dispatches of the IK kernel, applies the gfx950 corrections of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE x2 for wide reads."""
import csv
import glob
import json
import os
import sys

tag, dirs = sys.argv[1], sys.argv[2:]
acc, meta = {}, {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "ik_streams_kernel" not in r["Kernel_Name"]:
                    continue
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                meta = {"kernel": r["Kernel_Name"].split("(")[0].replace("void ", ""), "VGPR_Count": int(r["VGPR_Count"]),
                        "Accum_VGPR_Count": int(r["Accum_VGPR_Count"]), "SGPR_Count": int(r["SGPR_Count"]),
                        "Workgroup_Size": int(r["Workgroup_Size"]), "LDS_Block_Size": int(r["LDS_Block_Size"]),
                        "Grid_Size": int(r["Grid_Size"])}
out = {"source": f"rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group) -- python3 bench.py --steps 3 "
                 f"--warmup 1 --no-cpu-baseline; MI355X, round 1, kernel {tag}", **meta,
       "streams": meta.get("Grid_Size", 0) // max(meta.get("Workgroup_Size", 1), 1), "frames": 100}
for k, v in sorted(acc.items()):
    out[k + "_mean_per_launch"] = sum(v) / len(v)
if "FETCH_SIZE" in acc and "WRITE_SIZE" in acc:
    rd = out["FETCH_SIZE_mean_per_launch"] * 1024
    wr = out["WRITE_SIZE_mean_per_launch"] * 1024
    out["hbm_read_bytes_raw"] = rd
    out["hbm_write_bytes"] = wr
    out["hbm_bytes_per_launch"] = 2 * rd + wr
out["note"] = ("FETCH_SIZE/WRITE_SIZE are KiB. MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a "
               "wide coalesced read and is uncalibrated for other widths (the IK kernel reads 8 B/lane, 784 contiguous bytes per "
               "frame, plus the constants image per stream with 16 B/lane); hbm_bytes_per_launch applies the guide's x2 read "
               "correction = an upper bound. Known algorithmic read: 100*100*784 B + 100*288 B (+ 100 x image, L2-resident after "
               "the first stream); WRITE_SIZE equals the algorithmic write (q_out 2.88 MB + nsolve 80 KB). Algorithmic total "
               "(SURVEY 8d) 1360 B/frame * 1e4 = 13.6 MB: no wasted re-reads.")
print(json.dumps(out, indent=1))
