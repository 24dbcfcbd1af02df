"""Plugin registries, same names and keys as the reference's
``general_motion_retargeting/params.py:1-62`` (ROBOT_XML_DICT, IK_CONFIG_DICT, ROBOT_BASE_DICT,
VIEWER_CAM_DISTANCE_DICT, IK_CONFIG_ROOT, ASSET_ROOT).

Where the files come from
-------------------------
The plugin files themselves (MJCF robots, ik_config JSON) are the *user's* data and are consumed
unchanged.  They are looked up, in order, in

1. ``$GMR_ASSET_ROOT`` / ``$GMR_IK_CONFIG_ROOT`` (directories laid out like the reference's
   ``assets/`` and ``general_motion_retargeting/ik_configs/``),
2. a reference checkout next to this package or at ``$GMR_REFERENCE_ROOT``
   (``<root>/assets`` and ``<root>/general_motion_retargeting/ik_configs``),
3. the compiled packs bundled under ``data/`` (``robots/<key>.npz``, ``ik/<src>_to_<key>.npz``):
   numeric arrays produced from the reference's plugin files by ``tools/make_packs.py``.  These make
   the package self-contained on machines that have no copy of the plugin files (the GPU test box).

A dict value is therefore either a ``*.xml`` / ``*.json`` path or a ``*.npz`` pack path; the
loaders (:mod:`.models`) accept both.
"""
import os
import pathlib

HERE = pathlib.Path(__file__).parent
DATA_ROOT = HERE / "data"


def _first_dir(cands):
    for c in cands:
        if c and pathlib.Path(c).is_dir():
            return pathlib.Path(c)
    return None


_ref = os.environ.get("GMR_REFERENCE_ROOT")
ASSET_ROOT = _first_dir([
    os.environ.get("GMR_ASSET_ROOT"),
    pathlib.Path(_ref) / "assets" if _ref else None,
    HERE / ".." / "assets",
])
IK_CONFIG_ROOT = _first_dir([
    os.environ.get("GMR_IK_CONFIG_ROOT"),
    pathlib.Path(_ref) / "general_motion_retargeting" / "ik_configs" if _ref else None,
    HERE / "ik_configs",
])

# key -> (asset sub-directory, MJCF file, base body, viewer camera distance)
_ROBOTS = {
    "unitree_g1": ("unitree_g1", "g1_mocap_29dof.xml", "pelvis", 2.0),
    "booster_t1": ("booster_t1", "t1_mocap.xml", "Waist", 2.0),
    "booster_t1_4dof": ("booster_t1", "t1_mocap_4dof.xml", "Waist", 2.0),
    "stanford_toddy": ("stanford_toddy", "toddy_mocap.xml", "waist_link", 1.0),
    "fourier_n1": ("fourier_n1", "n1_mocap.xml", "base_link", 2.0),
    "engineai_pm01": ("engineai_pm01", "pm_v2.xml", "LINK_BASE", 2.0),
    "kuavo_s45": ("kuavo_s45", "biped_s45_collision.xml", "base_link", 2.0),
    "hightorque_hi": ("hightorque_hi", "hi_25dof.xml", "base_link", 2.0),
}
_ROBOT_XML_REL = {k: v[:2] for k, v in _ROBOTS.items()}

_IK_REL = {
    "smplx": {
        "unitree_g1": "smplx_to_g1.json",
        "booster_t1": "smplx_to_t1.json",
        "stanford_toddy": "smplx_to_toddy.json",
        "fourier_n1": "smplx_to_n1.json",
        "engineai_pm01": "smplx_to_pm01.json",
        "kuavo_s45": "smplx_to_kuavo.json",
        "hightorque_hi": "smplx_to_hi.json",
    },
    "bvh": {
        "unitree_g1": "bvh_to_g1.json",
        "booster_t1": "bvh_to_t1.json",
        "booster_t1_4dof": "bvh_to_t1_4dof.json",
        "fourier_n1": "bvh_to_n1.json",
        "stanford_toddy": "bvh_to_toddy.json",
        "engineai_pm01": "bvh_to_pm01.json",
    },
    "fbx": {
        "unitree_g1": "fbx_to_g1.json",
    },
}


def _robot_path(key):
    rel = _ROBOT_XML_REL[key]
    if ASSET_ROOT is not None and (ASSET_ROOT / rel[0] / rel[1]).is_file():
        return ASSET_ROOT / rel[0] / rel[1]
    return DATA_ROOT / "robots" / f"{key}.npz"


def _ik_path(src, key):
    rel = _IK_REL[src][key]
    if IK_CONFIG_ROOT is not None and (IK_CONFIG_ROOT / rel).is_file():
        return IK_CONFIG_ROOT / rel
    return DATA_ROOT / "ik" / (rel[:-5] + ".npz")


ROBOT_XML_DICT = {k: _robot_path(k) for k in _ROBOT_XML_REL}
IK_CONFIG_DICT = {src: {k: _ik_path(src, k) for k in tbl} for src, tbl in _IK_REL.items()}

ROBOT_BASE_DICT = {k: v[2] for k, v in _ROBOTS.items()}
VIEWER_CAM_DISTANCE_DICT = {k: v[3] for k, v in _ROBOTS.items()}
