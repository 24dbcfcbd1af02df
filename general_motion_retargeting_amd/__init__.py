"""MI355X-native drop-in for the hot path of GMR (General Motion Retargeting).

Same package surface as the reference's ``general_motion_retargeting/__init__.py:2-6``:
registries, ``GeneralMotionRetargeting``, ``KinematicsModel``, ``load_robot_motion`` (the viewer is
out of scope: SURVEY.md section 2).  Importing this package needs neither a GPU nor the built
library; every numerical entry point does, and raises if either is missing.
"""
from .params import IK_CONFIG_ROOT, ASSET_ROOT, ROBOT_XML_DICT, IK_CONFIG_DICT, ROBOT_BASE_DICT, VIEWER_CAM_DISTANCE_DICT
from .motion_retarget import GeneralMotionRetargeting, TargetNotSet
from .data_loader import load_robot_motion, save_robot_motion
from .kinematics_model import KinematicsModel
from . import dataset, sharding, synth


class RobotMotionViewer:  # pragma: no cover - out of scope (GUI), kept so imports do not break
    def __init__(self, *a, **k):
        raise NotImplementedError("RobotMotionViewer (MuJoCo GUI) is outside the MI355X hot path")


__all__ = ["IK_CONFIG_ROOT", "ASSET_ROOT", "ROBOT_XML_DICT", "IK_CONFIG_DICT", "ROBOT_BASE_DICT",
           "VIEWER_CAM_DISTANCE_DICT", "GeneralMotionRetargeting", "TargetNotSet", "KinematicsModel",
           "RobotMotionViewer", "load_robot_motion", "save_robot_motion"]
