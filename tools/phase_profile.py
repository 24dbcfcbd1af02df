#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the IK kernel (GMR_IK_PROFILE build, s_memtime stamps).
Not part of the product or of the bench; read SHARES, not absolute time (stamps forbid overlap).

    python tools/phase_profile.py [S] [T]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from general_motion_retargeting_amd import _lib, params, synth  # noqa: E402
from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset  # noqa: E402
from general_motion_retargeting_amd.models import load_ik_config, load_robot  # noqa: E402

PH = ["PRE", "FK", "ERR", "JLOG", "PAIRS", "CVEC", "HACC", "KBUILD", "CHOL", "SUBST", "RATIO", "MULT", "INTEG", "IO",
      "NFACT", "NSOLVE", "TICKS", "REALTIME"]


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    pkg = os.path.join(ROOT, "general_motion_retargeting_amd")
    so = os.path.join(pkg, "libgmrhip_prof.so")
    if not os.path.exists(so) or os.environ.get("REBUILD"):
        from general_motion_retargeting_amd import build
        build.build_variant("prof", ["-DGMR_IK_PROFILE"])
    _lib.LIB_PATH = so
    L = _lib.lib()
    model = load_robot(params.ROBOT_XML_DICT["unitree_g1"])
    tt = build_task_tables(load_ik_config(params.IK_CONFIG_DICT["smplx"]["unitree_g1"]), None)
    sol = _lib.Solver(pack_model(model), pack_taskset(model, tt))
    human, q0 = synth.make_streams(model, tt, S, T, seed=0)
    d_q0 = _lib.DeviceBuffer.from_host(q0)
    d_h = _lib.DeviceBuffer.from_host(human)
    d_qo = _lib.DeviceBuffer(S * T * sol.nq * 8)
    d_ns = _lib.DeviceBuffer(S * T * 8)
    d_st = _lib.DeviceBuffer(S * 4)
    d_pr = _lib.DeviceBuffer(2 * S * len(PH) * 8)
    _lib.check(L.gmr_memset(d_pr.ptr, 0, 2 * S * len(PH) * 8, None))
    fn = L.gmr_retarget_streams_prof
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 4
    for _ in range(2):
        _lib.check(fn(sol.handle, S, T, d_q0.ptr, d_h.ptr, 0, d_qo.ptr, d_ns.ptr, d_st.ptr, d_pr.ptr))
    _lib.check(L.gmr_stream_sync(None))
    both = d_pr.to_host((2, S, len(PH)), np.uint64).astype(np.float64)
    pr, hp = both[0], both[1]
    tot = pr[:, :14].sum(axis=1).mean()
    nsolve = pr[:, 15].mean()
    nfact = pr[:, 14].mean()
    print(f"S={S} T={T}  solves/stream={nsolve:.0f}  factorizations/solve={nfact / nsolve:.2f}  "
          f"stamped cycles/solve={tot / nsolve:.0f} (100 MHz ticks if s_memtime is the constant clock)")
    clk = pr[:, 16].mean() / pr[:, 17].mean() * 100.0
    print(f"  kernel ticks/stream={pr[:, 16].mean():.3e}  realtime(100MHz)={pr[:, 17].mean():.3e}  => in-kernel clock {clk:.0f} MHz; "
          f"stream wall {pr[:, 17].mean() / 100.0:.0f} us")
    rt = pr[:, 17] / 100.0
    print("  per-stream wall us: min %.0f median %.0f max %.0f ; fact/solve per stream: min %.2f max %.2f" % (
        rt.min(), np.median(rt), rt.max(), (pr[:, 14] / pr[:, 15]).min(), (pr[:, 14] / pr[:, 15]).max()))
    for i, n in enumerate(PH[:14]):
        print(f"  {n:7s} {pr[:, i].mean() / tot * 100:6.2f} %   {pr[:, i].mean() / nsolve:10.0f} /solve")
    if S <= 512:     # latency shape: the batch's time is its slowest stream's -- the same breakdown for that stream alone
        k = int(np.argmax(pr[:, 17]))
        med = int(np.argsort(pr[:, 17])[S // 2])
        for name, i in (("slowest", k), ("median", med)):
            ns_i, tot_i = pr[i, 15], pr[i, :14].sum()
            print(f"  {name} stream {i}: wall {pr[i, 17] / 100.0:.0f} us, {ns_i:.0f} solves, {tot_i / ns_i:.0f} stamped cycles/solve: " +
                  " ".join(f"{n}={pr[i, j] / ns_i:.0f}" for j, n in enumerate(PH[:14])))
    if hp.sum() > 0:
        print("  helper wavefront 1 (cycles/solve): idle at B1 %.0f, Jacobian share %.0f, wait B2 %.0f, H share %.0f, wait B3 %.0f, tree-QP %.0f" % (
            hp[:, 0].mean() / nsolve, hp[:, 4].mean() / nsolve, hp[:, 5].mean() / nsolve, hp[:, 6].mean() / nsolve,
            hp[:, 3].mean() / nsolve, hp[:, 7:14].sum(axis=1).mean() / nsolve))


if __name__ == "__main__":
    main()
