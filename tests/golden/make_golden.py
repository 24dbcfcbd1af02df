#!/usr/bin/env python3
"""Generate the golden fixtures tests/golden/*.npz by RUNNING the reference's own Python where it
is importable in the build container:

* G-PRE  -- ``GeneralMotionRetargeting.__init__`` / ``setup_retarget_configuration`` /
  ``update_targets`` (reference ``general_motion_retargeting/motion_retarget.py:13-136,203-270``)
  for every (source, robot) ik_config: seeded random ``human_data`` in, ``scaled_human_data`` and
  the poses handed to ``task.set_target`` out.  ``mink`` and ``mujoco`` are not installed; the
  preprocessing never touches them, so inert placeholder modules are registered under those names
  (they only record constructor arguments).  This pins rows H1-H3.
* G-FK   -- ``KinematicsModel`` (reference ``general_motion_retargeting/kinematics_model.py``) on
  CPU torch for the 7 robots it can parse: parsed tree arrays and ``forward_kinematics`` outputs
  for seeded random inputs.  This pins rows H8-H9.

* G-FKAUX -- the remaining public methods of ``KinematicsModel`` (``dof_to_rot``, ``rot_to_dof``,
  ``convert_local_rot_to_global``, ``forward_kinematics(fitted_shape=...)``, :172-246) for two robots.

* G-FKLARGE -- ``forward_kinematics`` on unwrapped angles up to 1e5 rad and on NaN / Inf angles (ADVICE round 2:
  the kernel's small-argument sincos must not be used there).

* G-BVH  -- the reference's LAFAN1 loader (``utils/lafan1.py`` + ``utils/lafan_vendor``, NumPy/SciPy
  only) on ``tests/golden/synthetic.bvh`` (a 22-joint, 12-frame BVH authored by this repository).
  This pins the "next" row N2.

* G-SMPLX -- the reference's SMPL-X frame extraction (``utils/smpl.py:44-197``: ``get_smplx_data``, ``slerp``,
  ``get_smplx_data_offline_fast`` -- NumPy/SciPy arithmetic) on a synthetic ``smplx_output`` (seeded random
  ``joints`` / ``full_pose`` float32 tensors, a 55-joint parent table) for 120->30, 60->30, 50->30 fps and the
  no-alignment branch.  The ``smplx`` package is not installed and its body model is not needed by these
  functions; a placeholder module supplies only ``JOINT_NAMES`` as 55 opaque labels ("j00".."j54") that become
  the dict keys.  This pins the pose half of the "next" row N1 (the body-model half stays unpinned).

Nothing here pins rows H4-H7 (the mink/MuJoCo/DAQP numerics): "parity unpinned", see DESIGN.md.

Only numbers and names are stored (``np.savez_compressed``; loadable with allow_pickle=False).

    python tests/golden/make_golden.py            # needs /root/reference
"""
import importlib
import os
import pathlib
import sys
import types

import numpy as np

REF = pathlib.Path(os.environ.get("GMR_REFERENCE_ROOT", "/root/reference"))
OUT = pathlib.Path(__file__).resolve().parent


# --------------------------------------------------------------------------- #
# inert placeholders for the two uninstalled third-party packages
# --------------------------------------------------------------------------- #
class _Rec:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k


def _install_placeholders():
    mink = types.ModuleType("mink")

    class Configuration(_Rec):
        pass

    class FrameTask(_Rec):
        target = None

        def set_target(self, t):
            self.target = t

    class SO3(_Rec):
        pass

    class SE3(_Rec):
        @classmethod
        def from_rotation_and_translation(cls, rot, pos):
            return cls(rot, pos)

    mink.Configuration, mink.FrameTask, mink.SO3, mink.SE3 = Configuration, FrameTask, SO3, SE3
    sys.modules["mink"] = mink

    mujoco = types.ModuleType("mujoco")

    class MjModel(_Rec):
        @classmethod
        def from_xml_path(cls, path):
            return cls(path)

    mujoco.MjModel = MjModel
    sys.modules["mujoco"] = mujoco


def _load_ref_module(name):
    """Import general_motion_retargeting.<name> without running the package __init__."""
    pkg_name = "general_motion_retargeting"
    if pkg_name not in sys.modules:
        pkg = types.ModuleType(pkg_name)
        pkg.__path__ = [str(REF / pkg_name)]
        sys.modules[pkg_name] = pkg
    return importlib.import_module(f"{pkg_name}.{name}")


def _random_human(rng, names, extra=("head", "jaw_extra")):
    data = {}
    for n in list(names) + list(extra):
        pos = rng.normal(0.0, 0.6, size=3) + np.array([0.0, 0.0, 0.9])
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        q *= 1.0 + 1e-3 * rng.normal()          # slightly de-normalised, like real loader output
        data[n] = (pos, q)
    return data


def make_pre():
    _install_placeholders()
    params = _load_ref_module("params")
    mr = _load_ref_module("motion_retarget")
    out = {}
    idx = 0
    for src, tbl in params.IK_CONFIG_DICT.items():
        for robot in tbl:
            for height in (None, 1.62):
                rng = np.random.default_rng(1000 + idx)
                idx += 1
                g = mr.GeneralMotionRetargeting(src, robot, actual_human_height=height)
                names = list(g.human_scale_table.keys())
                tag = f"{src}__{robot}__{'none' if height is None else 'h162'}"
                for ground in (False, True):
                    hd = _random_human(rng, names)
                    raw = np.array([np.concatenate([hd[n][0], hd[n][1]]) for n in names])
                    try:
                        g.update_targets(hd, offset_to_ground=ground)
                    except KeyError as e:      # some configs name scale bodies without table-1 entry
                        out[f"{tag}__g{int(ground)}__keyerror"] = np.array(str(e))
                        continue
                    sc = g.scaled_human_data
                    arr = np.array([np.concatenate([sc[n][0], sc[n][1]]) for n in names])
                    out[f"{tag}__g{int(ground)}__names"] = np.array(names)
                    out[f"{tag}__g{int(ground)}__raw"] = raw
                    out[f"{tag}__g{int(ground)}__scaled"] = arr
                    for s, tasks in ((1, g.tasks1), (2, g.tasks2)):
                        use = g.use_ik_match_table1 if s == 1 else g.use_ik_match_table2
                        if not use:
                            continue
                        tg = []
                        for t in tasks:
                            se3 = t.target
                            rot = np.asarray(se3.args[0].args[0], dtype=np.float64)
                            pos = np.asarray(se3.args[1], dtype=np.float64)
                            tg.append(np.concatenate([pos, rot]))
                        out[f"{tag}__g{int(ground)}__targets{s}"] = np.array(tg)
                        out[f"{tag}__g{int(ground)}__frames{s}"] = np.array([t.kwargs["frame_name"] for t in tasks])
                        out[f"{tag}__g{int(ground)}__costs{s}"] = np.array(
                            [[t.kwargs["position_cost"], t.kwargs["orientation_cost"]] for t in tasks], dtype=np.float64)
    np.savez_compressed(OUT / "g_pre.npz", **out)
    print("g_pre.npz:", len(out), "arrays")


def make_fk():
    import torch
    params = _load_ref_module("params")
    km = _load_ref_module("kinematics_model")
    out = {}
    for i, (robot, xml) in enumerate(params.ROBOT_XML_DICT.items()):
        try:
            model = km.KinematicsModel(str(xml), device="cpu")
        except AssertionError as e:
            out[f"{robot}__error"] = np.array(str(e))
            print("  ", robot, "-> AssertionError:", e)
            continue
        rng = np.random.default_rng(2000 + i)
        B = 24
        lo, hi = model.get_dof_limits()
        lo, hi = lo.numpy(), hi.numpy()
        dof = (lo + (hi - lo) * rng.uniform(size=(B, model.num_dof))).astype(np.float32)
        root_pos = rng.normal(size=(B, 3)).astype(np.float32)
        rq = rng.normal(size=(B, 4))
        root_rot = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)   # xyzw
        bp, br = model.forward_kinematics(torch.from_numpy(root_pos), torch.from_numpy(root_rot), torch.from_numpy(dof))
        out[f"{robot}__body_names"] = np.array(model.body_names)
        out[f"{robot}__parent"] = model.parent_indices.numpy().astype(np.int32)
        out[f"{robot}__local_translation"] = model._local_translation.numpy()
        out[f"{robot}__local_rotation"] = model._local_rotation.numpy()
        out[f"{robot}__dof_idx"] = np.array(model.joint_dof_idx, dtype=np.int32)
        out[f"{robot}__lower"] = lo
        out[f"{robot}__upper"] = hi
        out[f"{robot}__dof"] = dof
        out[f"{robot}__root_pos"] = root_pos
        out[f"{robot}__root_rot"] = root_rot
        out[f"{robot}__body_pos"] = bp.numpy()
        out[f"{robot}__body_rot"] = br.numpy()
        # identity-root call, as the dataset scripts do for local_body_pos (smplx_to_robot_dataset.py:106-112)
        z = torch.zeros((B, 3))
        idq = torch.zeros((B, 4)); idq[:, -1] = 1.0
        lbp, _ = model.forward_kinematics(z, idq, torch.from_numpy(dof))
        out[f"{robot}__local_body_pos"] = lbp.numpy()
    np.savez_compressed(OUT / "g_fk.npz", **out)
    print("g_fk.npz:", len(out), "arrays")




def make_fk_large():
    """G-FKLARGE: ``forward_kinematics`` accepts ANY user angle (``rot_to_dof`` clamps, FK does not): unwrapped angles of
    a few hundred to a few thousand radians, and one frame with a NaN and an Inf angle.  torch.sin / torch.cos are
    accurate for all of them; a kernel with a small-argument sincos must take another path there."""
    import torch
    params = _load_ref_module("params")
    km = _load_ref_module("kinematics_model")
    out = {}
    for i, robot in enumerate(["unitree_g1", "fourier_n1"]):
        model = km.KinematicsModel(str(params.ROBOT_XML_DICT[robot]), device="cpu")
        rng = np.random.default_rng(2700 + i)
        B = 12
        scale = np.array([3.0, 30.0, 300.0, 900.0, 1100.0, 3000.0, 2.0e4, 1.0e5, 1.0, 400.0, 400.0, 400.0])[:, None]
        dof = (scale * rng.uniform(-1, 1, size=(B, model.num_dof))).astype(np.float32)
        dof[9, 2] = np.nan                      # the frames' bodies below joint 2 / 5 become NaN in the reference too
        dof[10, 5] = np.inf
        root_pos = rng.normal(size=(B, 3)).astype(np.float32)
        rq = rng.normal(size=(B, 4))
        root_rot = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)
        bp, br = model.forward_kinematics(torch.from_numpy(root_pos), torch.from_numpy(root_rot), torch.from_numpy(dof))
        out[f"{robot}__dof"], out[f"{robot}__root_pos"], out[f"{robot}__root_rot"] = dof, root_pos, root_rot
        out[f"{robot}__body_pos"], out[f"{robot}__body_rot"] = bp.numpy(), br.numpy()
    np.savez_compressed(OUT / "g_fk_large.npz", **out)
    print("g_fk_large.npz:", len(out), "arrays")


def make_fk_aux():
    """G-FKAUX: the other public methods of the reference's ``KinematicsModel`` (kinematics_model.py:172-246):
    ``dof_to_rot``, ``rot_to_dof``, ``convert_local_rot_to_global`` and ``forward_kinematics(fitted_shape=...)``,
    on CPU torch for two robots.  Pins those methods of row H9."""
    import torch
    params = _load_ref_module("params")
    km = _load_ref_module("kinematics_model")
    out = {}
    for i, robot in enumerate(["unitree_g1", "booster_t1"]):
        model = km.KinematicsModel(str(params.ROBOT_XML_DICT[robot]), device="cpu")
        rng = np.random.default_rng(2500 + i)
        B = 6
        lo, hi = model.get_dof_limits()
        lo, hi = lo.numpy(), hi.numpy()
        dof = (lo + (hi - lo) * rng.uniform(size=(B, model.num_dof))).astype(np.float32)
        jr = model.dof_to_rot(torch.from_numpy(dof))                       # [B, nb-1, 4] xyzw
        out[f"{robot}__dof"] = dof
        out[f"{robot}__joint_rot"] = jr.numpy()
        out[f"{robot}__dof_back"] = model.rot_to_dof(jr).numpy()
        # joints pushed beyond their limits are clamped by rot_to_dof
        wide = (dof + rng.normal(0.0, 1.0, size=dof.shape)).astype(np.float32)
        out[f"{robot}__dof_wide"] = wide
        out[f"{robot}__dof_wide_back"] = model.rot_to_dof(model.dof_to_rot(torch.from_numpy(wide))).numpy()
        rq = rng.normal(size=(B, 1, 4))
        root = (rq / np.linalg.norm(rq, axis=-1, keepdims=True)).astype(np.float32)
        local = np.concatenate([root, jr.numpy()], axis=1)                 # [B, nb, 4]: row 0 the root rotation
        out[f"{robot}__local_rot"] = local
        out[f"{robot}__global_rot"] = model.convert_local_rot_to_global(torch.from_numpy(local)).numpy()
        shape = (1.0 + 0.2 * rng.uniform(-1, 1, size=(model.num_joint, 3))).astype(np.float32)
        root_pos = rng.normal(size=(B, 3)).astype(np.float32)
        bp, br = model.forward_kinematics(torch.from_numpy(root_pos), torch.from_numpy(root[:, 0]), torch.from_numpy(dof),
                                          fitted_shape=torch.from_numpy(shape))
        out[f"{robot}__fitted_shape"] = shape
        out[f"{robot}__root_pos"] = root_pos
        out[f"{robot}__root_rot"] = root[:, 0]
        out[f"{robot}__fitted_body_pos"] = bp.numpy()
        out[f"{robot}__fitted_body_rot"] = br.numpy()
    np.savez_compressed(OUT / "g_fk_aux.npz", **out)
    print("g_fk_aux.npz:", len(out), "arrays")


def make_bvh():
    """G-BVH: the reference's LAFAN1 loader (utils/lafan1.py + lafan_vendor, NumPy/SciPy only) on the
    small synthetic BVH authored by this repository (tests/golden/synthetic.bvh)."""
    pkg = "general_motion_retargeting"
    if pkg not in sys.modules:
        p = types.ModuleType(pkg); p.__path__ = [str(REF / pkg)]; sys.modules[pkg] = p
    lafan1 = importlib.import_module(f"{pkg}.utils.lafan1")
    frames, height = lafan1.load_lafan1_file(str(OUT / "synthetic.bvh"))
    names = list(frames[0].keys())
    arr = np.array([[np.concatenate([f[n][0], f[n][1]]) for n in names] for f in frames])
    np.savez_compressed(OUT / "g_bvh.npz", names=np.array(names), poses=arr, height=np.array(height))
    print("g_bvh.npz:", arr.shape)


def make_smplx():
    """G-SMPLX: reference utils/smpl.py on synthetic body-model outputs (see the module docstring)."""
    import torch
    smplx = types.ModuleType("smplx")
    jn = types.ModuleType("smplx.joint_names")
    jn.JOINT_NAMES = [f"j{i:02d}" for i in range(60)]         # opaque labels; only the first len(parents) are used
    smplx.joint_names = jn
    sys.modules["smplx"], sys.modules["smplx.joint_names"] = smplx, jn
    smpl = _load_ref_module("utils.smpl")
    # a 55-joint tree with the branching of a humanoid + two 15-joint hands (parents precede children)
    parents = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15]
                       + [20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38]
                       + [21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)
    J = len(parents)
    out = {"parents": parents}
    body_model = types.SimpleNamespace(parents=parents)
    for case, (src_fps, N, tgt_fps) in enumerate(((120.0, 41, 30), (60.0, 23, 30), (50.0, 17, 30), (30.0, 9, 30),
                                                  (120.0, 8, 30))):
        rng = np.random.default_rng(7000 + case)
        # smooth random-walk poses with a few large rotations (angles up to ~pi) and sign flips across frames
        base = rng.normal(0, 0.8, size=(1, J, 3))
        walk = np.cumsum(rng.normal(0, 0.15, size=(N, J, 3)), axis=0)
        pose = base + walk
        pose[:, 5] *= 2.2                                   # one joint sweeps through large angles
        pose[N // 2:, 7] *= -1.0                            # one joint jumps to the antipodal side mid-clip
        pose[:, 9] *= 1e-5                                  # one joint stays in the small-angle branch
        pose[:, 11] = pose[0, 11]                           # one joint is constant (lerp branch of slerp)
        full_pose = torch.tensor(pose.reshape(N, J * 3), dtype=torch.float32)
        joints = torch.tensor(np.cumsum(rng.normal(0, 0.02, size=(N, J + 3, 3)), axis=0) + rng.normal(0, 0.5, size=(1, J + 3, 3)),
                              dtype=torch.float32)          # the model returns extra landmark joints after the 55
        so = types.SimpleNamespace(global_orient=full_pose[:, :3].clone(), full_pose=full_pose, joints=joints)
        data = {"mocap_frame_rate": np.array(src_fps), "pose_body": np.zeros((N, 63))}
        frames, aligned_fps = smpl.get_smplx_data_offline_fast(data, body_model, so, tgt_fps=tgt_fps)
        names = jn.JOINT_NAMES[:J]
        arr = np.array([[np.concatenate([np.asarray(f[n][0], dtype=np.float64), f[n][1]]) for n in names] for f in frames])
        tag = f"c{case}"
        out[f"{tag}__src_fps"] = np.array(src_fps)
        out[f"{tag}__tgt_fps"] = np.array(tgt_fps)
        out[f"{tag}__full_pose"] = full_pose.numpy()
        out[f"{tag}__joints"] = joints.numpy()
        out[f"{tag}__frames"] = arr                          # [N', 55, 7] pos xyz + quat wxyz
        out[f"{tag}__aligned_fps"] = np.array(float(aligned_fps))
        # the per-frame entry point on the source frames (no alignment)
        single = [smpl.get_smplx_data(data, body_model, so, t) for t in (0, N - 1)]
        out[f"{tag}__single"] = np.array([[np.concatenate([np.asarray(f[n][0], dtype=np.float64), f[n][1]]) for n in names]
                                          for f in single])
    # slerp() itself on a few hand-picked pairs
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(7100)
    q1 = rng.normal(size=(12, 4)); q2 = rng.normal(size=(12, 4))
    q2[:3] = q1[:3] + 1e-3 * rng.normal(size=(3, 4))        # nearly equal -> lerp branch
    q2[3:5] = -q1[3:5] + 1e-2 * rng.normal(size=(2, 4))     # nearly antipodal -> flipped
    tt = rng.uniform(0, 1, size=12)
    res = np.array([smpl.slerp(R.from_quat(a), R.from_quat(b), t).as_quat() for a, b, t in zip(q1, q2, tt)])
    out["slerp__q1"], out["slerp__q2"], out["slerp__t"], out["slerp__out"] = q1, q2, tt, res
    np.savez_compressed(OUT / "g_smplx.npz", **out)
    print("g_smplx.npz:", len(out), "arrays")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "smplx":
        make_smplx()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fk_aux":
        make_fk_aux()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "fk_large":
        make_fk_large()
        sys.exit(0)
    make_pre()
    make_fk()
    make_fk_aux()
    make_fk_large()
    make_bvh()
    make_smplx()
