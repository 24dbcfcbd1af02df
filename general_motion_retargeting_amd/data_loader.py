"""Robot-motion pkl reader/writer.

* ``load_robot_motion``: reference ``general_motion_retargeting/data_loader.py:3-15``;
* ``save_robot_motion`` / ``motion_dict``: the dict written by ``scripts/smplx_to_robot_dataset.py:134-146``;
* ``to_training_compatible`` / ``save_robot_motion(..., training_compatible=True)``: the list-valued,
  pickle-protocol-2 variant produced by ``booster_gym/utils/convert_pkl_for_training.py:44-74``;
* ``motion_arrays``: what the downstream consumer ``booster_gym/utils/motion_loader.py:42-98`` extracts
  from either variant (float32 arrays, ``local_body_pos`` optional, ``link_body_list`` defaulting to []).
"""
import pickle

import numpy as np

MOTION_KEYS = ("fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list")


def load_robot_motion(motion_file):
    """Returns (motion_data, fps, root_pos, root_rot wxyz, dof_pos, local_body_pos, link_body_list)."""
    with open(motion_file, "rb") as f:
        motion_data = pickle.load(f)      # files written by this package / the caller
    root_rot = np.asarray(motion_data["root_rot"])[:, [3, 0, 1, 2]]   # stored xyzw -> wxyz
    return (motion_data, motion_data["fps"], motion_data["root_pos"], root_rot, motion_data["dof_pos"],
            motion_data["local_body_pos"], motion_data["link_body_list"])


def to_training_compatible(motion_data):
    """ndarray values -> nested Python lists, everything else kept (convert_pkl_for_training.py:44-64)."""
    return {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in motion_data.items()}


def save_robot_motion(motion_file, motion_data, training_compatible=False):
    """Writes the App. D dict.  ``training_compatible`` = lists + pickle protocol 2 (:72-74), the form
    older Python / NumPy installations on training machines can read."""
    missing = [k for k in MOTION_KEYS if k not in motion_data]
    if missing:
        raise KeyError(f"motion_data lacks {missing}")
    with open(motion_file, "wb") as f:
        if training_compatible:
            pickle.dump(to_training_compatible(motion_data), f, protocol=2)
        else:
            pickle.dump(dict(motion_data), f)


def motion_arrays(motion_data):
    """The consumer's view (motion_loader.py:72-98): float32 root_pos / root_rot (xyzw) / dof_pos from
    arrays or lists, ``local_body_pos`` or None, ``link_body_list`` or [], plus fps-derived timing."""
    fps = motion_data["fps"]
    n = len(motion_data["root_pos"])
    out = {k: np.asarray(motion_data[k], dtype=np.float32) for k in ("root_pos", "root_rot", "dof_pos")}
    lbp = motion_data.get("local_body_pos")
    out["local_body_pos"] = None if lbp is None else np.asarray(lbp, dtype=np.float32)
    out["link_body_list"] = motion_data.get("link_body_list") or []
    out.update(fps=fps, dt=1.0 / fps, num_frames=n, motion_duration=n / fps)
    return out


def motion_dict(fps, root_pos, root_rot_xyzw, dof_pos, local_body_pos=None, link_body_list=None):
    return {
        "fps": fps,
        "root_pos": np.asarray(root_pos),
        "root_rot": np.asarray(root_rot_xyzw),
        "dof_pos": np.asarray(dof_pos),
        "local_body_pos": local_body_pos,
        "link_body_list": link_body_list,
    }
