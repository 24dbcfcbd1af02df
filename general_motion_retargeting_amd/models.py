"""Loaders that accept either the reference's plugin files or the bundled compiled packs.

* robots: ``*.xml`` (MJCF, compiled by :func:`..mjcf.compile_mjcf`) or ``*.npz`` pack;
* ik configs: ``*.json`` (reference schema) or ``*.npz`` pack, both returned as
  the same ``dict`` the reference gets from ``json.load`` (motion_retarget.py:30-31).

Packs are written by ``tools/make_packs.py``; they hold numbers and names only.
"""
from __future__ import annotations

import json
from typing import Dict, Optional

import numpy as np

from .mjcf import RobotModel, compile_mjcf, parse_kinematics_tree

_KM_KEYS = ("body_names", "parent", "local_translation", "local_rotation", "dof_dim", "dof_idx", "axis",
            "lower", "upper")


def load_robot(path) -> RobotModel:
    path = str(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return RobotModel.from_arrays({k: z[k] for k in z.files if not k.startswith("km_")})
    return compile_mjcf(path)


def load_kinematics_tree(path) -> Dict[str, np.ndarray]:
    """Tree arrays with the reference KinematicsModel's own parsing semantics (H8)."""
    path = str(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            # mirrors the reference: a top file without <worldbody> (engineai_pm01) is rejected
            assert "km_parent" in z.files, "worldbody not found"
            return {k: z["km_" + k] for k in _KM_KEYS}
    if not path.endswith(".xml"):
        raise NotImplementedError("File type not supported")
    return parse_kinematics_tree(path)


def robot_pack_arrays(xml_path) -> Dict[str, np.ndarray]:
    arrays = compile_mjcf(xml_path).to_arrays()
    try:
        tree = parse_kinematics_tree(xml_path)
    except AssertionError:
        tree = None
    if tree is not None:
        arrays.update({"km_" + k: v for k, v in tree.items()})
    return arrays


def ik_config_to_arrays(cfg: dict) -> Dict[str, np.ndarray]:
    out = {
        "robot_root_name": np.array(cfg["robot_root_name"]),
        "human_root_name": np.array(cfg["human_root_name"]),
        "ground_height": np.array(float(cfg["ground_height"])),
        "human_height_assumption": np.array(float(cfg["human_height_assumption"])),
        "use": np.array([bool(cfg["use_ik_match_table1"]), bool(cfg["use_ik_match_table2"])]),
        "scale_names": np.array(list(cfg["human_scale_table"].keys())),
        "scale_vals": np.array([float(v) for v in cfg["human_scale_table"].values()]),
    }
    for s in (1, 2):
        tbl = cfg[f"ik_match_table{s}"]
        out[f"t{s}_frames"] = np.array(list(tbl.keys()))
        out[f"t{s}_humans"] = np.array([e[0] for e in tbl.values()])
        out[f"t{s}_w"] = np.array([[float(e[1]), float(e[2])] for e in tbl.values()]).reshape(-1, 2)
        out[f"t{s}_pos"] = np.array([e[3] for e in tbl.values()], dtype=np.float64).reshape(-1, 3)
        out[f"t{s}_quat"] = np.array([e[4] for e in tbl.values()], dtype=np.float64).reshape(-1, 4)
    return out


def ik_config_from_arrays(z) -> dict:
    cfg = {
        "robot_root_name": str(z["robot_root_name"]),
        "human_root_name": str(z["human_root_name"]),
        "ground_height": float(z["ground_height"]),
        "human_height_assumption": float(z["human_height_assumption"]),
        "use_ik_match_table1": bool(z["use"][0]),
        "use_ik_match_table2": bool(z["use"][1]),
        "human_scale_table": {str(n): float(v) for n, v in zip(z["scale_names"], z["scale_vals"])},
    }
    for s in (1, 2):
        tbl = {}
        for i, fr in enumerate(z[f"t{s}_frames"]):
            w = z[f"t{s}_w"][i]
            tbl[str(fr)] = [str(z[f"t{s}_humans"][i]), float(w[0]), float(w[1]),
                            [float(x) for x in z[f"t{s}_pos"][i]], [float(x) for x in z[f"t{s}_quat"][i]]]
        cfg[f"ik_match_table{s}"] = tbl
    return cfg


def load_ik_config(path) -> dict:
    path = str(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return ik_config_from_arrays(z)
    with open(path) as f:
        return json.load(f)
