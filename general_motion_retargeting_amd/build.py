"""Build libgmrhip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgmrhip.so")
SOURCES = ["gmr_ik.hip", "gmr_ik_wide.hip", "gmr_fk.hip", "gmr_smplx.hip", "gmr_abi.hip"]
HEADERS = ["gmr_device_math.h", "gmr_ik_layout.h", "gmr_ik_wide_layout.h", "gmr_ik_prof.h", "gmr_ik_tree.h", "gmr_fk_tree.h", "gmr_internal.h", "../../include/gmr_hip.h",
           "../../include/gmr_types.h"]


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


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-o", LIB + ".tmp"]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
