#!/usr/bin/env python3
"""Copy one run of tools/profile_round<N>.sh from gpurun_out/<round>_<tag>/ into profiles/ as <round>_<tag>_* (JSON outputs
reduced to their JSON, progress lines dropped), regenerate the compiler's kernel resource report and profiles/traffic.json
(with the hash of the kernel sources it was measured on: bench.py prints `traffic: null` + "stale" when HEAD differs).

    python tools/publish_profiles.py v4 [--round r03] [--drop v3]
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
drop = sys.argv[sys.argv.index("--drop") + 1] if "--drop" in sys.argv else None
RND = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "r03"
src = os.path.join(ROOT, "gpurun_out", f"{RND}_{tag}")
dst = os.path.join(ROOT, "profiles")


def last_json(path):
    """The last line (or trailing block) of a file that parses as JSON."""
    text = open(path).read().strip()
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        pass
    for start in range(len(text)):
        if text[start] in "{[" and (start == 0 or text[start - 1] == "\n"):
            try:
                return json.loads(text[start:])
            except json.JSONDecodeError:
                continue
    raise SystemExit(f"no JSON in {path}")


COPY_JSON = {"bench.json": "bench.json", "counters_headline.json": "counters_headline_kernel.json",
             "counters_wide.json": "counters_wide_kernel.json", "counters_aux.json": "counters_aux_kernels.json",
             "aux_kernels.json": "aux_kernels.json", "extras.json": "extras.json", "latency_config5.json": "latency_config5.json",
             "soak_all_configs.json": "soak_all_configs.json", "stress_bounds.json": "stress_bounds.json",
             "configs_3_4.json": "configs_3_4.json", "dataset_probe.json": "dataset_probe.json"}
COPY_TEXT = {"phase_shares.txt": "phase_shares.txt", "phase_shares_wide.txt": "phase_shares_wide.txt", "shape_sweep.txt": "shape_sweep.txt"}
if drop:
    for f in glob.glob(os.path.join(dst, f"{RND}_{drop}_*")):
        base = os.path.basename(f)[len(f"{RND}_{drop}_"):]
        if base in COPY_JSON.values() or base in COPY_TEXT.values() or base in ("bench_kernel_stats.csv", "aux_kernel_stats.csv", "kernel_resource_usage.txt"):
            os.remove(f)
for a, b in COPY_JSON.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        json.dump(last_json(p), open(os.path.join(dst, f"{RND}_{tag}_{b}"), "w"), indent=1)
for a, b in COPY_TEXT.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{RND}_{tag}_{b}"))
for a, b in (("bench_stats/run_kernel_stats.csv", "bench_kernel_stats.csv"), ("aux_stats/run_kernel_stats.csv", "aux_kernel_stats.csv")):
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{RND}_{tag}_{b}"))

# the compiler's view of every kernel, with the flags build.py uses
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import build  # noqa: E402

lines = []
csrc = os.path.join(ROOT, "general_motion_retargeting_amd", "csrc")
for name in build.SOURCES:
    if not name.endswith(".hip"):
        continue
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", "-c", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", os.path.join(csrc, name)] + build.PER_SOURCE_FLAGS.get(name, [])
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    for ln in out.splitlines():
        if "remark:" in ln:
            lines.append(name + ": " + ln.split("remark:", 1)[1].replace("[-Rpass-analysis=kernel-resource-usage]", "").rstrip())
open(os.path.join(dst, f"{RND}_{tag}_kernel_resource_usage.txt"), "w").write(
    "hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (per-source flags of build.py)\n" + "\n".join(lines) + "\n")

# counter files: cite the resource report that exists
for name in ("counters_headline_kernel.json", "counters_wide_kernel.json", "counters_aux_kernels.json"):
    fp = os.path.join(dst, f"{RND}_{tag}_{name}")
    if os.path.exists(fp):
        j = json.load(open(fp))
        for v in j.values():
            if "note_registers" in v:
                v["note_registers"] = v["note_registers"].rsplit("The compiler's own numbers", 1)[0] + \
                    f"The compiler's own numbers per kernel (VGPRs, AGPRs, spills, occupancy) are in profiles/{RND}_{tag}_kernel_resource_usage.txt"
        json.dump(j, open(fp, "w"), indent=1)

# traffic.json: the headline kernel's counters in the form bench.py reads + what the throughput kernel EXECUTES
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_sources_sha256)

head = json.load(open(os.path.join(dst, f"{RND}_{tag}_counters_headline_kernel.json")))
kname, k = next(iter(head.items()))
old = json.load(open(os.path.join(dst, "traffic.json")))
t = {"source": f"rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group) -- python3 bench.py --steps 3 --warmup 1 "
               f"--no-cpu-baseline --no-strong --no-extra-legs; MI355X, tools/profile_round3.sh {tag}",
     "kernel_sources_sha256": bench.kernel_sources_sha256(), "kernel": kname, "streams": 100, "frames": 100}
t.update({a: b for a, b in k.items() if a != "note_registers"})
t["register_note"] = old["register_note"].rsplit("The compiler's own numbers", 1)[0] + \
    f"The compiler's own numbers per kernel (VGPRs, AGPRs, spills, occupancy) are in profiles/{RND}_{tag}_kernel_resource_usage.txt"
t["note"] = old["note"]
wide_fp = os.path.join(dst, f"{RND}_{tag}_counters_wide_kernel.json")
if os.path.exists(wide_fp):
    wname, w = next(iter(json.load(open(wide_fp)).items()))
    g = lambda n: w.get(n + "_mean_per_launch", 0.0)  # noqa: E731
    S, T = 16384, 16
    probe = os.path.join(src, "pmc_wide_sq3.log")
    spf = None
    if os.path.exists(probe):
        for ln in open(probe):
            if ln.startswith("{") and "solves_per_frame" in ln:
                spf = json.loads(ln)["solves_per_frame"]
    solves = S * T * (spf or 8.17)
    flop_all_lanes = 64.0 * (2 * g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64"))
    lane_util = g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU") else None
    t["wide"] = {
        "kernel": wname, "workload": f"S={S} x T={T} (tools/wide_probe.py)", "solves_per_launch": solves,
        "valu_per_solve": g("SQ_INSTS_VALU") / solves, "salu_per_solve": g("SQ_INSTS_SALU") / solves, "lds_per_solve": g("SQ_INSTS_LDS") / solves,
        "fp64_share_of_valu": (g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_TRANS_F64")) / g("SQ_INSTS_VALU"),
        "int32_share_of_valu": g("SQ_INSTS_VALU_INT32") / g("SQ_INSTS_VALU"),
        "lds_bank_conflict_share": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
        "mean_active_lane_fraction_of_valu": lane_util,
        "mfma_f64_insts_per_solve": g("SQ_INSTS_VALU_MFMA_F64") / solves,
        "mfma_f64_flop_per_solve": 2048.0 * g("SQ_INSTS_VALU_MFMA_F64") / solves,      # v_mfma_f64_16x16x4: 16 x 16 x 4 FMA
        "executed_fp64_flop_per_solve_all_lanes_live": flop_all_lanes / solves,
        "executed_fp64_flop_per_solve": flop_all_lanes / solves * (lane_util or 1.0),
        "note": "executed flop = 64 lanes x (2 FMA + MUL + ADD + TRANS) FP64 wave-instructions per launch; multiplied by the mean "
                "active-lane fraction of VALU instructions (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), all VALU types) for the "
                "second figure.  bench.py turns it into `fp64_valu.executed_frac` with the solve count it measures.  The Schur "
                "products of the tree solver run as v_mfma_f64_16x16x4 (SQ_INSTS_VALU_MFMA_F64, 2 048 flop each, 7 of 16 K-columns "
                "and 9 x 9 of the 16 x 16 outputs used): counted separately (mfma_f64_*), not in the VALU figures.",
    }
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("published", RND, tag)
