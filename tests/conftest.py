import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("GMR_REFERENCE_ROOT", "/root/reference")
HAVE_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "general_motion_retargeting"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Setup:
    """(robot model, task tables, packed blobs) for one (source, robot) pair."""

    def __init__(self, src, robot, height=None):
        from general_motion_retargeting_amd import params
        from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset
        from general_motion_retargeting_amd.models import load_ik_config, load_robot
        self.src, self.robot = src, robot
        self.model = load_robot(params.ROBOT_XML_DICT[robot])
        self.cfg = load_ik_config(params.IK_CONFIG_DICT[src][robot])
        self.tt = build_task_tables(self.cfg, height)
        self.mb = pack_model(self.model)
        self.ts = pack_taskset(self.model, self.tt)


_cache = {}


def get_setup(src="smplx", robot="unitree_g1", height=None) -> Setup:
    key = (src, robot, height)
    if key not in _cache:
        _cache[key] = Setup(src, robot, height)
    return _cache[key]


@pytest.fixture(scope="session")
def g1():
    return get_setup("smplx", "unitree_g1")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


ALL_CONFIGS = [
    ("smplx", "unitree_g1"), ("smplx", "booster_t1"), ("smplx", "stanford_toddy"), ("smplx", "fourier_n1"),
    ("smplx", "engineai_pm01"), ("smplx", "kuavo_s45"), ("smplx", "hightorque_hi"),
    ("bvh", "unitree_g1"), ("bvh", "booster_t1"), ("bvh", "booster_t1_4dof"), ("bvh", "fourier_n1"),
    ("bvh", "stanford_toddy"), ("bvh", "engineai_pm01"), ("fbx", "unitree_g1"),
]
