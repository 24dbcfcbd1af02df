"""MJCF / ik_config compilers, packed-struct ABI, C-ABI symbol exports (no GPU compute)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ALL_CONFIGS, HAVE_REFERENCE, REFERENCE, ROOT, get_setup
from general_motion_retargeting_amd import params
from general_motion_retargeting_amd.ik_config import (MAX_DEPTH, MODEL_DTYPE, TASKSET_DTYPE, build_task_tables,
                                                      pack_model, pack_taskset, task_dofs)
from general_motion_retargeting_amd.models import (ik_config_to_arrays, load_ik_config, load_kinematics_tree,
                                                   load_robot, robot_pack_arrays)

# SURVEY.md Appendix B
ROBOT_FACTS = {
    "unitree_g1": (38, 29), "booster_t1": (32, 21), "booster_t1_4dof": (26, 21), "stanford_toddy": (33, 22),
    "fourier_n1": (29, 23), "engineai_pm01": (29, 24), "kuavo_s45": (29, 28), "hightorque_hi": (26, 25),
}


@pytest.mark.parametrize("robot", list(ROBOT_FACTS))
def test_robot_model_invariants(robot):
    m = load_robot(params.ROBOT_XML_DICT[robot])
    nb, nh = ROBOT_FACTS[robot]
    assert (m.nbody, m.nhinge, m.nq, m.nv) == (nb, nh, nh + 7, nh + 6)
    assert m.parent[0] == -1 and all(0 <= m.parent[b] < b for b in range(1, nb))       # DFS order
    assert np.allclose(np.linalg.norm(m.body_quat, axis=1), 1) and np.allclose(np.linalg.norm(m.hinge_axis, axis=1), 1)
    assert m.limited.all() and (m.range_lo < m.range_hi).all()
    assert np.array_equal(m.qpos0[:3], m.body_pos[0]) and np.array_equal(m.qpos0[3:7], m.body_quat[0])
    assert (m.qpos0[7:] == 0).all() and m.timestep == 0.002
    assert [m.body_hinge[b] for b in m.hinge_body] == list(range(nh))
    assert int(m.depth().max()) + 1 <= MAX_DEPTH
    assert params.ROBOT_BASE_DICT[robot] in m.body_names


def test_g1_known_values():
    m = load_robot(params.ROBOT_XML_DICT["unitree_g1"])
    assert m.body_names[0] == "pelvis" and m.body_names[1] == "left_hip_pitch_link"
    assert m.joint_names[0] == "left_hip_pitch_joint" and m.joint_names[3] == "left_knee_joint"
    assert np.allclose(m.qpos0[:7], [0, 0, 0.793, 1, 0, 0, 0])
    assert (m.range_lo[3], m.range_hi[3]) == (-0.087267, 2.8798)
    b = m.body_id("left_hip_roll_link")
    q = np.array([0.996179, 0, -0.0873386, 0])
    assert np.allclose(m.body_quat[b], q / np.linalg.norm(q), atol=1e-15)
    assert np.array_equal(m.body_pos[m.body_id("left_toe_link")], [0.1, 0, -0.02])
    assert m.body_hinge[m.body_id("left_toe_link")] == -1
    with pytest.raises(KeyError):
        m.body_id("nope")


def test_pm01_oblique_axes_are_normalised():
    m = load_robot(params.ROBOT_XML_DICT["engineai_pm01"])
    a = m.hinge_axis[m.joint_names.index("J00_HIP_PITCH_L")]
    raw = np.array([0, 0.965926, -0.258819])
    assert np.allclose(a, raw / np.linalg.norm(raw), atol=1e-15) and abs(np.linalg.norm(a) - 1) < 1e-15


@pytest.mark.skipif(not HAVE_REFERENCE, reason="reference checkout not mounted (packs are used instead)")
def test_bundled_packs_equal_fresh_compile_of_the_plugin_files():
    for robot, rel in params._ROBOT_XML_REL.items():
        xml = os.path.join(REFERENCE, "assets", *rel)
        fresh = robot_pack_arrays(xml)
        with np.load(params.DATA_ROOT / "robots" / f"{robot}.npz", allow_pickle=False) as z:
            assert set(z.files) == set(fresh)
            for k in z.files:
                assert np.array_equal(z[k], fresh[k]), (robot, k)
    for src, tbl in params._IK_REL.items():
        for robot, name in tbl.items():
            js = os.path.join(REFERENCE, "general_motion_retargeting", "ik_configs", name)
            cfg = load_ik_config(js)
            packed = load_ik_config(params.DATA_ROOT / "ik" / (name[:-5] + ".npz"))
            assert packed == cfg, name
            assert list(packed["ik_match_table1"]) == list(cfg["ik_match_table1"])     # order matters


def test_mjcf_defaults_includes_and_errors(tmp_path):
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "leg.xml").write_text(
        '<mujoco><body name="leg" pos="0 0 -1"><joint name="j1" class="wide"/>'
        '<body name="foot" childclass="narrow"><joint name="j2" axis="0 2 0"/></body></body></mujoco>')
    (tmp_path / "top.xml").write_text(
        '<mujoco model="t"><compiler angle="degree"/><option timestep="0.01"/>'
        '<default><joint axis="1 0 0" range="-90 90"/>'
        '<default class="wide"><joint range="-180 180"/></default>'
        '<default class="narrow"><joint range="0 0"/></default></default>'
        '<worldbody><body name="base" pos="0 0 1" quat="2 0 0 0"><freejoint/><include file="sub/leg.xml"/></body>'
        '</worldbody></mujoco>')
    from general_motion_retargeting_amd.mjcf import compile_mjcf
    m = compile_mjcf(str(tmp_path / "top.xml"))
    assert m.body_names == ["base", "leg", "foot"] and m.timestep == 0.01
    assert np.allclose(m.body_quat[0], [1, 0, 0, 0])
    assert np.allclose(m.range_hi, [np.pi, 0.0]) and list(m.limited) == [1, 0]           # autolimits: lo < hi
    assert np.allclose(m.hinge_axis, [[1, 0, 0], [0, 1, 0]])
    bad = tmp_path / "bad.xml"
    bad.write_text('<mujoco><worldbody><body name="b"><freejoint/><body name="c"><joint type="slide"/></body>'
                   '</body></worldbody></mujoco>')
    with pytest.raises(NotImplementedError):
        compile_mjcf(str(bad))
    bad.write_text('<mujoco><worldbody><body name="b"/></worldbody></mujoco>')
    with pytest.raises(NotImplementedError):
        compile_mjcf(str(bad))


@pytest.mark.parametrize("src,robot", ALL_CONFIGS)
def test_taskset_packing(src, robot):
    su = get_setup(src, robot, 1.7)
    ts, tt, m = su.ts[0], su.tt, su.model
    assert ts["nhuman"] == len(tt.human_names) and tt.human_names[ts["human_root"]] == tt.human_root_name
    ratio = 1.7 / su.cfg["human_height_assumption"]
    assert np.allclose(ts["scale"][: ts["nhuman"]], [v * ratio for v in su.cfg["human_scale_table"].values()])
    for s in range(2):
        st = tt.stages[s]
        tbl = su.cfg[f"ik_match_table{s + 1}"]
        kept = [(f, e) for f, e in tbl.items() if e[1] != 0 or e[2] != 0]
        assert st.frame_names == [f for f, _ in kept] and ts["ntask"][s] == len(kept)
        p = 0
        for k, (f, e) in enumerate(kept):
            assert m.body_names[ts["task_body"][s][k]] == f and tt.human_names[ts["task_human"][s][k]] == e[0]
            assert (ts["w_pos"][s][k], ts["w_rot"][s][k]) == (e[1], e[2])
            dofs = task_dofs(m, m.body_id(f))
            assert dofs[:6] == list(range(6))
            if e[1] == 0:               # no position cost: the base translations' columns are exactly zero, not listed
                dofs = dofs[3:]
            assert list(ts["pair_dof"][s][p: p + len(dofs)]) == dofs
            p += len(dofs)
        assert ts["npair"][s] == p
    g = su.cfg["ground_height"]
    for i, n in enumerate(tt.human_names):
        e = su.cfg["ik_match_table1"]
        ent = [v for v in e.values() if v[0] == n and (v[1] != 0 or v[2] != 0)]
        if ent:
            assert np.allclose(ts["pos_off"][i], np.array(ent[-1][3]) - [0, 0, g])
            q = np.array(ent[-1][4]); assert np.allclose(ts["quat_off"][i], q / np.linalg.norm(q))
        assert ts["is_foot"][i] == int("foot" in n or "Foot" in n)
    assert (ts["damping"], ts["lm_damping"], ts["tol"], ts["limit_gain"], ts["max_iter"]) == (0.5, 1.0, 0.001, 0.95, 10)


def test_unknown_names_raise_keyerror():
    from general_motion_retargeting_amd import GeneralMotionRetargeting
    with pytest.raises(KeyError):
        GeneralMotionRetargeting("smplx", "no_such_robot")
    with pytest.raises(KeyError):
        GeneralMotionRetargeting("no_such_source", "unitree_g1")
    su = get_setup()
    cfg = {k: (dict(v) if isinstance(v, dict) else v) for k, v in su.cfg.items()}
    cfg["ik_match_table1"] = dict(cfg["ik_match_table1"]); cfg["ik_match_table1"]["ghost_link"] = ["pelvis", 1, 1, [0, 0, 0], [1, 0, 0, 0]]
    with pytest.raises(KeyError):
        pack_taskset(su.model, build_task_tables(cfg))


def test_struct_layouts_match_the_c_headers(oracle):
    su = get_setup()
    oracle.check_abi(su.mb, su.ts)                                   # gcc's sizeof == numpy itemsize
    assert MODEL_DTYPE.fields["timestep"][1] % 8 == 0 and TASKSET_DTYPE.fields["damping"][1] % 8 == 0
    hdr = open(os.path.join(ROOT, "include", "gmr_types.h")).read()
    from general_motion_retargeting_amd import ik_config as ic
    for name in ("MAX_BODIES", "MAX_HINGES", "MAX_DOF", "MAX_NQ", "MAX_DEPTH", "MAX_TASKS", "MAX_HUMAN", "MAX_PAIRS"):
        assert int(re.search(rf"#define GMR_{name}\s+(\d+)", hdr).group(1)) == getattr(ic, name)


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    from general_motion_retargeting_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "gmr_hip.h")).read()
    declared = set(re.findall(r"\b(gmr_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = C.CDLL(_lib.LIB_PATH)
    for sym in sorted(declared):
        assert hasattr(L, sym), f"{sym} declared in include/gmr_hip.h but not exported"
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    lib = _lib.lib()                                                 # also checks struct sizes against the .so
    assert lib.gmr_sizeof_model() == MODEL_DTYPE.itemsize and lib.gmr_sizeof_taskset() == TASKSET_DTYPE.itemsize


def test_product_fails_loudly_without_gpu_or_library(monkeypatch):
    from general_motion_retargeting_amd import GeneralMotionRetargeting, KinematicsModel, ROBOT_XML_DICT, _lib
    if _lib.lib().gmr_device_count() > 0:
        pytest.skip("GPU present")
    su = get_setup()
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(su.model, su.tt, 1, 1, seed=0)
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    with pytest.raises(_lib.GmrHipError):
        g.retarget(synth.streams_to_dicts(su.tt, human[0])[0])
    with pytest.raises(_lib.GmrHipError):
        KinematicsModel(ROBOT_XML_DICT["unitree_g1"]).forward_kinematics(np.zeros((1, 3)), np.array([[0, 0, 0, 1.0]]), np.zeros((1, 29)))
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgmrhip.so")
    with pytest.raises(_lib.GmrHipError):
        _lib.lib()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "general_motion_retargeting_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} mentions the oracle"
