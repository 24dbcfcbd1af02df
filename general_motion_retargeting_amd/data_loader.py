"""Robot-motion pkl reader/writer (reference ``general_motion_retargeting/data_loader.py:3-15`` and
the dict written by ``scripts/smplx_to_robot_dataset.py:134-146``)."""
import pickle

import numpy as np

MOTION_KEYS = ("fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list")


def load_robot_motion(motion_file):
    """Returns (motion_data, fps, root_pos, root_rot wxyz, dof_pos, local_body_pos, link_body_list)."""
    with open(motion_file, "rb") as f:
        motion_data = pickle.load(f)      # files written by this package / the caller
    root_rot = motion_data["root_rot"][:, [3, 0, 1, 2]]   # stored xyzw -> wxyz
    return (motion_data, motion_data["fps"], motion_data["root_pos"], root_rot, motion_data["dof_pos"],
            motion_data["local_body_pos"], motion_data["link_body_list"])


def save_robot_motion(motion_file, motion_data):
    missing = [k for k in MOTION_KEYS if k not in motion_data]
    if missing:
        raise KeyError(f"motion_data lacks {missing}")
    with open(motion_file, "wb") as f:
        pickle.dump(dict(motion_data), f)


def motion_dict(fps, root_pos, root_rot_xyzw, dof_pos, local_body_pos=None, link_body_list=None):
    return {
        "fps": fps,
        "root_pos": np.asarray(root_pos),
        "root_rot": np.asarray(root_rot_xyzw),
        "dof_pos": np.asarray(dof_pos),
        "local_body_pos": local_body_pos,
        "link_body_list": link_body_list,
    }
