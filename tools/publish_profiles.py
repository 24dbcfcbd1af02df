#!/usr/bin/env python3
"""Copy one run of tools/profile_round2.sh from gpurun_out/r02_<tag>/ into profiles/ as r02_<tag>_* (JSON outputs reduced
to their JSON, progress lines dropped), regenerate the compiler's kernel resource report and profiles/traffic.json.

    python tools/publish_profiles.py v4 [--drop v3]
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
src = os.path.join(ROOT, "gpurun_out", f"r02_{tag}")
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
             "configs_3_4.json": "configs_3_4.json"}
COPY_TEXT = {"phase_shares.txt": "phase_shares.txt", "phase_shares_wide.txt": "phase_shares_wide.txt", "shape_sweep.txt": "shape_sweep.txt"}
if drop:
    for f in glob.glob(os.path.join(dst, f"r02_{drop}_*")):
        base = os.path.basename(f)[len(f"r02_{drop}_"):]
        if base in COPY_JSON.values() or base in COPY_TEXT.values() or base in ("bench_kernel_stats.csv", "aux_kernel_stats.csv", "kernel_resource_usage.txt"):
            os.remove(f)
for a, b in COPY_JSON.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        json.dump(last_json(p), open(os.path.join(dst, f"r02_{tag}_{b}"), "w"), indent=1)
for a, b in COPY_TEXT.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"r02_{tag}_{b}"))
for a, b in (("bench_stats/run_kernel_stats.csv", "bench_kernel_stats.csv"), ("aux_stats/run_kernel_stats.csv", "aux_kernel_stats.csv")):
    p = os.path.join(src, a)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"r02_{tag}_{b}"))

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
open(os.path.join(dst, f"r02_{tag}_kernel_resource_usage.txt"), "w").write(
    "hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage (per-source flags of build.py)\n" + "\n".join(lines) + "\n")

# traffic.json: the headline kernel's counters in the form bench.py reads
head = json.load(open(os.path.join(dst, f"r02_{tag}_counters_headline_kernel.json")))
kname, k = next(iter(head.items()))
old = json.load(open(os.path.join(dst, "traffic.json")))
t = {"source": old["source"], "kernel": kname, "streams": 100, "frames": 100}
t.update({a: b for a, b in k.items() if a != "note_registers"})
t["register_note"] = old["register_note"].rsplit("The compiler's own numbers", 1)[0] + f"The compiler's own numbers per kernel (VGPRs, AGPRs, spills, occupancy) are in profiles/r02_{tag}_kernel_resource_usage.txt"
t["note"] = old["note"]
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("published", tag)
