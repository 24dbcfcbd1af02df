#!/usr/bin/env python3
"""Compile the reference's plugin files (MJCF robots, ik_config JSON) into the numeric packs that
ship under general_motion_retargeting_amd/data/ (run in the build container, where the reference
checkout is mounted; the GPU box has no copy of those files).

    GMR_REFERENCE_ROOT=/root/reference python tools/make_packs.py
"""
import os
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GMR_REFERENCE_ROOT", "/root/reference")

from general_motion_retargeting_amd import params  # noqa: E402
from general_motion_retargeting_amd.models import (ik_config_to_arrays, load_ik_config,  # noqa: E402
                                                   robot_pack_arrays)


def main():
    out_r = params.DATA_ROOT / "robots"
    out_i = params.DATA_ROOT / "ik"
    out_r.mkdir(parents=True, exist_ok=True)
    out_i.mkdir(parents=True, exist_ok=True)
    for key, path in params.ROBOT_XML_DICT.items():
        assert str(path).endswith(".xml"), f"{key}: plugin files not found ({path})"
        np.savez_compressed(out_r / f"{key}.npz", **robot_pack_arrays(str(path)))
        print("robot", key, "<-", path)
    for src, tbl in params.IK_CONFIG_DICT.items():
        for key, path in tbl.items():
            assert str(path).endswith(".json"), f"{src}/{key}: plugin files not found ({path})"
            name = pathlib.Path(path).stem
            np.savez_compressed(out_i / f"{name}.npz", **ik_config_to_arrays(load_ik_config(path)))
            print("ik", src, key, "<-", path)


if __name__ == "__main__":
    main()
