"""The C++ host logic behind the IK kernel (static H-assembly schedule, limb/trunk decomposition, LDS
layout + image) checked on the CPU for every (source, robot) pair: tests/cpp/layout_check.cpp is plain
C++ (g++), includes the product header csrc/gmr_ik_layout.h and reads the packed structs from a file."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ALL_CONFIGS, ROOT, get_setup

SRC = os.path.join(ROOT, "tests", "cpp", "layout_check.cpp")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp") / "layout_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, SRC])
    return exe


@pytest.mark.parametrize("src,robot", ALL_CONFIGS)
def test_schedule_tree_and_layout(checker, tmp_path, src, robot):
    su = get_setup(src, robot, 1.7)
    blob = tmp_path / "blob.bin"
    with open(blob, "wb") as f:
        f.write(su.mb.tobytes())
        f.write(su.ts.tobytes())
    out = subprocess.run([checker, str(blob)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("ok") and "tree=1" in out.stdout and "limbs=4" in out.stdout, out.stdout
    # shipped configs: both tables name the same tasks (3; 7 if also the same pairs -- not when a table's tasks without a
    # position cost have their base-translation pairs pruned), or there is no second table (0)
    assert out.stdout.strip().endswith(("use1=3", "use1=7", "use1=0")), out.stdout


def test_non_decomposable_robot_falls_back(checker, tmp_path):
    """Four 10-hinge chains: limbs longer than 8 would have to join the trunk, which then exceeds 10."""
    from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset
    from general_motion_retargeting_amd.mjcf import compile_mjcf
    axes = ["1 0 0", "0 1 0", "0 0 1"]

    def chain(prefix, n, pos):
        s, e = "", ""
        for i in range(n):
            s += f'<body name="{prefix}{i}" pos="{pos if i == 0 else "0 0 -0.08"}"><joint name="{prefix}j{i}" axis="{axes[i % 3]}" range="-1 1"/>'
            e += "</body>"
        return s + e
    xml = ('<mujoco><compiler angle="radian"/><worldbody><body name="base"><freejoint/>'
           + chain("a", 10, "0 0.1 0") + chain("b", 10, "0 -0.1 0") + "</body></worldbody></mujoco>")
    p = tmp_path / "r.xml"
    p.write_text(xml)
    model = compile_mjcf(str(p))
    tbl = {"base": ["root", 10, 10, [0, 0, 0], [1, 0, 0, 0]], "a9": ["ha", 10, 10, [0, 0, 0], [1, 0, 0, 0]]}
    cfg = {"robot_root_name": "base", "human_root_name": "root", "ground_height": 0.0, "human_height_assumption": 1.8,
           "use_ik_match_table1": True, "use_ik_match_table2": False, "human_scale_table": {"root": 1.0, "ha": 1.0},
           "ik_match_table1": tbl, "ik_match_table2": {}}
    tt = build_task_tables(cfg)
    blob = tmp_path / "blob.bin"
    with open(blob, "wb") as f:
        f.write(pack_model(model).tobytes())
        f.write(pack_taskset(model, tt).tobytes())
    out = subprocess.run([checker, str(blob)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "tree=0" in out.stdout


@pytest.fixture(scope="module")
def split_checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp") / "fk_split_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "fk_split_check.cpp")])
    return exe


def test_fk_tree_partition_for_the_split_walk(split_checker):
    """Every loadable robot tree, a chain, a star and a deep binary tree: all bodies covered, lists parent-closed."""
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    trees = {}
    for robot, xml in params.ROBOT_XML_DICT.items():
        try:
            trees[robot] = [int(x) for x in load_kinematics_tree(xml)["parent"]]
        except AssertionError:          # engineai_pm01: the reference's own parser rejects it (worldbody in an include)
            continue
    assert "unitree_g1" in trees
    trees["chain"] = [-1] + list(range(0, 19))
    trees["star"] = [-1] + [0] * 30
    trees["binary"] = [-1] + [(b - 1) // 2 for b in range(1, 63)]
    trees["single"] = [-1]
    for name, par in trees.items():
        out = subprocess.run([split_checker], input=f"{len(par)} " + " ".join(map(str, par)), capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (name, out.stdout, out.stderr)
        if name == "unitree_g1":        # four wavefronts walk at most 13 of the 38 bodies each
            assert "maxw=4 nw=4 longest=13" in out.stdout, out.stdout


def test_layout_code_is_clean_under_sanitizers(tmp_path):
    """The host-side schedule / layout builders (what gmr_solver_create runs before any launch) under
    AddressSanitizer + UBSan for every shipped (source, robot) pair: a table written out of bounds on the host would
    become an out-of-bounds LDS or global access on the GPU."""
    exe = str(tmp_path / "layout_check_asan")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                         "-o", exe, SRC], capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("sanitizer runtime not available: " + cc.stderr[-200:])
    for src, robot in ALL_CONFIGS:
        su = get_setup(src, robot, 1.7)
        blob = tmp_path / "blob.bin"
        with open(blob, "wb") as f:
            f.write(su.mb.tobytes())
            f.write(su.ts.tobytes())
        out = subprocess.run([exe, str(blob)], capture_output=True, text=True)
        assert out.returncode == 0 and "runtime error" not in out.stderr and "ERROR" not in out.stderr, (src, robot, out.stderr[-2000:])
