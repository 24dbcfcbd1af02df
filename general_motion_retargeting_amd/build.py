"""Build libgmrhip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to its own object (cached under build/obj by source + header mtimes and flags), then
linked: editing one kernel file rebuilds that file only.  ``build_variant(name, defines)`` produces
``libgmrhip_<name>.so`` for A/B measurements through ``GMR_HIP_LIBRARY`` (diagnostic builds; never loaded by default).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgmrhip.so")
OBJ = os.path.join(os.path.dirname(HERE), "build", "obj")
SOURCES = ["gmr_ik.hip", "gmr_ik_wide.hip", "gmr_fk.hip", "gmr_smplx.hip", "gmr_comm.hip", "gmr_abi.hip"]
HEADERS = ["gmr_ik_wide_item.inc", "gmr_device_math.h", "gmr_ik_layout.h", "gmr_ik_wide_layout.h", "gmr_ik_prof.h", "gmr_ik_tree.h",
           "gmr_fk_tree.h", "gmr_internal.h", "../../include/gmr_hip.h", "../../include/gmr_types.h"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
# The throughput kernel must stay within 256 registers (two wavefronts per SIMD; since round 3: 168, three).  Machine-LICM hoists every FP64
# literal and lane predicate of the (fully inlined) frame loop into registers that live for the whole kernel; they
# then spill to scratch and are RELOADED inside the loop (99 spilled VGPRs, 336 B of scratch per lane).  Without it
# and with sinking enabled: 226 VGPRs, no scratch, +13 % frames/s (profiles/r02_*).
PER_SOURCE_FLAGS = {"gmr_ik_wide.hip": ["-mllvm", "-disable-machine-licm", "-mllvm", "-sink-insts-to-avoid-spills=1"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def _object(src: str, defines, force: bool, verbose: bool) -> str:
    os.makedirs(OBJ, exist_ok=True)
    per_source = [] if os.environ.get("GMR_BUILD_NO_PER_SOURCE_FLAGS") else PER_SOURCE_FLAGS.get(src, [])   # (A/B builds)
    defines = per_source + list(defines)
    tag = hashlib.sha1(" ".join(FLAGS + defines).encode()).hexdigest()[:10]
    obj = os.path.join(OBJ, f"{os.path.splitext(src)[0]}.{tag}.o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    if force or verbose or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
        cmd = [_hipcc()] + FLAGS + list(defines) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd, cwd=CSRC)
    return obj


def _link(objs, out: str) -> str:
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", out + ".tmp"] + objs + ["-ldl"])
    os.replace(out + ".tmp", out)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    return _link([_object(s, [], force, verbose) for s in SOURCES], LIB)


def build_variant(name: str, defines=(), verbose: bool = False) -> str:
    """libgmrhip_<name>.so with extra -D flags on every source (objects cached per flag set)."""
    out = os.path.join(HERE, f"libgmrhip_{name}.so")
    return _link([_object(s, list(defines), False, verbose) for s in SOURCES], out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=False))
    else:
        print(build(force=True, verbose=True))
