"""Dataset harness: what the reference's ``process_file`` does after loading a clip
(``scripts/smplx_to_robot_dataset.py:78-146`` and ``scripts/bvh_to_robot_dataset.py:86-152``;
SURVEY.md H10), for MANY clips per launch.

Per clip: retarget every frame from ``qpos0`` (a fresh ``GeneralMotionRetargeting`` per file,
:79-87) -> split qpos, root quaternion wxyz -> xyzw (:97-102) -> ``local_body_pos`` = FK with
identity root (:106-112) -> optional height adjust ``root_z -= min_{t,b} z`` over the world FK
(:118-126) -> optional ``root_xy -= root_xy[0]`` (:128-131) -> the pkl dict (:134-141).
The SMPL-X script has both adjustments on, the BVH script both off (``bvh_to_robot_dataset.py:128``).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .data_loader import motion_dict
from .kinematics_model import KinematicsModel
from .motion_retarget import GeneralMotionRetargeting


def postprocess_clip(qpos: np.ndarray, km: KinematicsModel, fps: float, height_adjust: bool = True,
                     root_origin_offset: bool = True, ground_offset: float = 0.0) -> Dict:
    """qpos f64[T, nq] of one clip -> motion dict (App. D)."""
    qpos = np.array(qpos, dtype=np.float64, copy=True)
    root_pos = qpos[:, :3].copy()
    root_rot = qpos[:, [4, 5, 6, 3]].copy()          # wxyz -> xyzw
    dof_pos = qpos[:, 7:].copy()
    T = qpos.shape[0]
    ident_pos = np.zeros((T, 3), dtype=np.float32)
    ident_rot = np.zeros((T, 4), dtype=np.float32)
    ident_rot[:, 3] = 1.0
    local_body_pos, _ = km.forward_kinematics(ident_pos, ident_rot, dof_pos.astype(np.float32))
    if height_adjust and T > 0:
        _, _, lowest = km.forward_kinematics(root_pos.astype(np.float32), root_rot.astype(np.float32),
                                             dof_pos.astype(np.float32), return_min_z=True)
        root_pos[:, 2] = root_pos[:, 2] - lowest + ground_offset
    if root_origin_offset and T > 0:
        root_pos[:, :2] -= root_pos[0, :2]
    return motion_dict(fps, root_pos, root_rot, dof_pos, np.asarray(local_body_pos), km.body_names)


def retarget_clips(src_human: str, tgt_robot: str, clips: Sequence, fps: Sequence[float],
                   actual_human_height: Optional[float] = None, height_adjust: bool = True,
                   root_origin_offset: bool = True, offset_to_ground: bool = False) -> List[Dict]:
    """Retarget many clips of one (source, robot, height) in ONE IK launch.

    ``clips[i]`` is a list of ``human_data`` dicts or an array ``[T_i, nhuman, 7]`` (ragged lengths
    are fine: streams are padded and the kernel stops each stream at its own length).
    Returns one motion dict per clip, identical to processing the clips one by one.
    """
    gmr = GeneralMotionRetargeting(src_human, tgt_robot, actual_human_height=actual_human_height)
    packed = [c if isinstance(c, np.ndarray) else gmr.pack_frames(c) for c in clips]
    S = len(packed)
    if S == 0:
        return []
    lens = np.array([p.shape[0] for p in packed], dtype=np.int32)
    T = max(int(lens.max()), 1)
    nh = len(gmr.human_body_names)
    human = np.zeros((S, T, nh, 7))
    human[..., 3] = 1.0
    for i, p in enumerate(packed):
        human[i, : p.shape[0]] = p
    qpos, nsolve, status = gmr.retarget_streams(human, lens=lens, offset_to_ground=offset_to_ground)
    if (status != 0).any():
        bad = np.nonzero(status)[0].tolist()
        raise RuntimeError(f"IK failed for clips {bad}")
    km = KinematicsModel(gmr.xml_file)
    return [postprocess_clip(qpos[i, : lens[i]], km, fps[i], height_adjust, root_origin_offset) for i in range(S)]


def retarget_mixed(groups: Sequence[Dict], offset_to_ground: bool = False, slices: int = 0, pinned_outputs: bool = False):
    """Mixed-robot batch (BASELINE.json configs[3]; SURVEY.md section 8d "one kernel with per-stream model index"): every
    group = one (source, robot, height) with its own streams; ALL groups form one scheduling domain on the device --
    the robots of the throughput kernel's size class (every shipped one) share a resident grid and one queue of
    (robot, stream, chunk) items (``gmr_retarget_group``) -- and the host buffers are cut into slices whose copies
    overlap the kernels.  Bit-identical to retargeting every group by itself.

    ``groups[i]`` = ``{"src_human", "tgt_robot", "human": f64[S,T,nhuman,7], optional "lens",
    "actual_human_height"}``; arrays allocated with ``_lib.pinned_empty`` are copied asynchronously.
    Returns a list of ``(qpos[S,T,nq], nsolve[S,T,2], status[S])``.
    """
    from . import _lib
    jobs, keep = [], []
    for g in groups:
        gmr = GeneralMotionRetargeting(g["src_human"], g["tgt_robot"], actual_human_height=g.get("actual_human_height"))
        keep.append(gmr)
        jobs.append({"solver": gmr.hip_solver, "human": g["human"], "lens": g.get("lens")})
    flags = _lib.FLAG_OFFSET_TO_GROUND if offset_to_ground else 0
    return _lib.retarget_group(jobs, flags, slices, out_pinned=pinned_outputs)   # (rows beyond a stream's length: zeros)


def retarget_bvh_files(bvh_files: Sequence[str], tgt_robot: str, fps: float = 30.0, height_adjust: bool = False,
                       root_origin_offset: bool = False) -> List[Dict]:
    """``scripts/bvh_to_robot_dataset.py:60-152`` for a list of files: every BVH clip becomes one
    stream of ONE launch (LAFAN1: 77 ragged clips).  Both adjustments default to off like that script
    (``HEIGHT_ADJUST = False``, :128); the loader's hard-coded height 1.75 is used (lafan1.py:39)."""
    from .utils.lafan1 import load_lafan1_packed
    gmr = GeneralMotionRetargeting("bvh", tgt_robot, actual_human_height=1.75)
    clips = [load_lafan1_packed(f, gmr.human_body_names)[0] for f in bvh_files]
    return retarget_clips("bvh", tgt_robot, clips, [fps] * len(clips), actual_human_height=1.75,
                          height_adjust=height_adjust, root_origin_offset=root_origin_offset)


def retarget_smplx_files(smplx_files: Sequence[str], smplx_body_model_path: str, tgt_robot: str, tgt_fps: int = 30,
                         height_adjust: bool = True, root_origin_offset: bool = True,
                         skip_errors: bool = True) -> List[Optional[Dict]]:
    """``scripts/smplx_to_robot_dataset.py:39-146`` (``process_file``) for a list of AMASS-style SMPL-X
    files, everything after the file read on the device: joints-only body model + fps alignment
    (utils/smpl.py) -> packed frames -> IK -> FK post-processing.  A file's human height comes from its betas
    (smpl.py:36-39), so files are grouped by height: one task set and ONE ragged IK launch per group, the
    groups running concurrently on separate HIP streams.  Both adjustments default to on like that script.
    Returns one motion dict per file in input order (``None`` for a file that failed to load when
    ``skip_errors`` -- the script prints and skips, :62-76)."""
    from .utils import smpl
    loaded: Dict[int, tuple] = {}
    for i, f in enumerate(smplx_files):
        try:
            loaded[i] = smpl.load_smplx_file(f, smplx_body_model_path)
        except Exception as e:  # noqa: BLE001 - mirrors the script's print-and-skip
            if not skip_errors:
                raise
            print(f"Error loading {f}: {e}")
    by_height: Dict[float, List[int]] = {}
    for i, (_, _, _, h) in loaded.items():
        by_height.setdefault(float(h), []).append(i)
    groups, members, fps_of = [], [], {}
    for h, idxs in by_height.items():
        gmr = GeneralMotionRetargeting("smplx", tgt_robot, actual_human_height=h)
        clips = []
        for i in idxs:
            data, bm, so, _ = loaded[i]
            packed, fps_of[i] = smpl.smplx_frames_packed(gmr, data, bm, so, tgt_fps=tgt_fps)
            clips.append(packed)
        lens = np.array([c.shape[0] for c in clips], dtype=np.int32)
        T = max(int(lens.max()), 1)
        human = np.zeros((len(clips), T, len(gmr.human_body_names), 7))
        human[..., 3] = 1.0
        for k, c in enumerate(clips):
            human[k, : c.shape[0]] = c
        groups.append({"src_human": "smplx", "tgt_robot": tgt_robot, "actual_human_height": h, "human": human, "lens": lens})
        members.append((idxs, lens, gmr.xml_file))
    results = retarget_mixed(groups) if groups else []
    out: List[Optional[Dict]] = [None] * len(smplx_files)
    km = None
    for (idxs, lens, xml), (qpos, _, status) in zip(members, results):
        if (status != 0).any():
            raise RuntimeError(f"IK failed for files {[smplx_files[idxs[k]] for k in np.nonzero(status)[0]]}")
        km = km or KinematicsModel(xml)
        for k, i in enumerate(idxs):
            out[i] = postprocess_clip(qpos[k, : lens[k]], km, fps_of[i], height_adjust, root_origin_offset)
    return out


def retarget_single_clip(retargeter: GeneralMotionRetargeting, frames: Sequence, fps: float,
                         skip_first_frame: bool = True) -> Dict:
    """The save path of the single-clip scripts (``scripts/smplx_to_robot.py:104-161``,
    ``scripts/bvh_to_robot.py`` alike): one ``retarget()`` per displayed frame, then a pkl dict with
    ``root_rot`` xyzw, ``local_body_pos = None`` and ``link_body_list = None`` (:146-158).  Their loop
    increments the index BEFORE the first use (:104-111), so frame 0 is never retargeted; that is the default
    here too.  One launch instead of one call per frame; continues from ``retargeter``'s configuration."""
    seq = frames[1:] if skip_first_frame else frames
    qpos = retargeter.retarget_clip(seq)
    return motion_dict(fps, qpos[:, :3].copy(), qpos[:, [4, 5, 6, 3]].copy(), qpos[:, 7:].copy(), None, None)


# ----------------------------------------------------------------------------------------------
# the dataset drivers with the reference's FILE semantics
#   python -m general_motion_retargeting_amd.dataset --source smplx|bvh --src_folder ... --tgt_folder ... --robot ...
# (scripts/smplx_to_robot_dataset.py:171-242, scripts/bvh_to_robot_dataset.py:16-157): same folder walk, same
# skip-if-exists / --override rule, same exclusion lists, one pkl per input -- but the files of a batch share ONE IK
# launch (and one FK launch per clip) instead of one process per file.
# ----------------------------------------------------------------------------------------------
EXCLUDE_FILE_CONTENT = ["BMLrub", "EKUT", "crawl", "_lie", "upstairs", "downstairs"]    # smplx_to_robot_dataset.py:218


def natural_key(name: str):
    """Sort key of ``natsort.natsorted`` (default algorithm: runs of digits compare as unsigned integers)."""
    import re
    parts = re.split(r"(\d+)", name)
    return [int(p) if p.isdigit() else p for p in parts if p != ""]


def _natsorted(names):
    try:                                   # the reference uses natsort; not installed here -> same order from natural_key
        from natsort import natsorted
        return natsorted(names)
    except ImportError:
        def key(n):
            return [(0, p, "") if isinstance(p, int) else (1, 0, p) for p in natural_key(n)]
        return sorted(names, key=key)


def load_hard_motions(paths) -> List[str]:
    """Names listed in the ``assets/hard_motions/*.txt`` reports (``Motion: <name>.pkl, Difficulty: ...``;
    smplx_to_robot_dataset.py:196-205)."""
    import os
    out = []
    for p in paths:
        if not os.path.exists(p):
            continue
        with open(p, "r") as f:
            for line in f:
                if "Motion:" not in line:
                    continue
                motion_path = line.split(":")[1].strip()
                out.append(motion_path.split(",")[0].strip().split(".")[0])
    return out


def list_smplx_jobs(src_folder: str, tgt_folder: str, override: bool = False, hard_motions: Sequence[str] = (),
                    exclude: Sequence[str] = tuple(EXCLUDE_FILE_CONTENT)):
    """(source file, target pkl) pairs in the reference's order and with its filters (:208-228): natsorted walk,
    ``*_stagei.npz`` skipped, ``.pkl`` / ``.npz`` accepted, existing targets skipped unless ``override``, hard and
    infeasible motions removed."""
    import os
    jobs = []
    for dirpath, _, filenames in os.walk(src_folder):
        for filename in _natsorted(filenames):
            if filename.endswith("_stagei.npz"):
                continue
            if filename.endswith((".pkl", ".npz")):
                src = os.path.join(dirpath, filename)
                tgt = src.replace(src_folder, tgt_folder).replace(".npz", ".pkl")
                if not os.path.exists(tgt) or override:
                    jobs.append((src, tgt))
    hard = set(hard_motions)
    kept = []
    for src, tgt in jobs:
        motion_name = src.split("/")[-1].split(".")[0]
        if motion_name in hard or any(c in motion_name for c in exclude):
            continue
        kept.append((src, tgt))
    return jobs, kept


def list_bvh_jobs(src_folder: str, tgt_folder: str, override: bool = False, verbose: bool = True):
    """(bvh file, target pkl) pairs of ``bvh_to_robot_dataset.py:60-73``: sorted walk, ``.bvh`` only, existing targets
    skipped (with the script's message) unless ``override``."""
    import os
    jobs = []
    for dirpath, _, filenames in os.walk(src_folder):
        for filename in sorted(filenames):
            if not filename.endswith(".bvh"):
                continue
            src = os.path.join(dirpath, filename)
            tgt = src.replace(src_folder, tgt_folder).replace(".bvh", ".pkl")
            if os.path.exists(tgt) and not override:
                if verbose:
                    print(f"Skipping {src} because {tgt} exists")
                continue
            jobs.append((src, tgt))
    return jobs


def _dump(tgt: str, motion: Dict, key_order: Sequence[str]) -> None:
    import os
    import pickle
    os.makedirs(os.path.dirname(tgt) or ".", exist_ok=True)
    with open(tgt, "wb") as f:
        pickle.dump({k: motion[k] for k in key_order}, f)


SMPLX_KEYS = ("fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list")      # :134-141
BVH_KEYS = ("root_pos", "root_rot", "dof_pos", "local_body_pos", "fps", "link_body_list")        # bvh script :141-148


def run_smplx_dataset(src_folder: str, tgt_folder: str, robot: str, smplx_folder: str, override: bool = False,
                      hard_motion_files: Sequence[str] = (), batch_files: int = 256, retarget=None, verbose: bool = True) -> int:
    """``smplx_to_robot_dataset.py:main`` with ``batch_files`` files per IK launch.  ``retarget`` defaults to
    :func:`retarget_smplx_files` (tests inject a stand-in).  Returns the number of pkl files written."""
    retarget = retarget or retarget_smplx_files
    all_jobs, jobs = list_smplx_jobs(src_folder, tgt_folder, override, load_hard_motions(hard_motion_files))
    if verbose:
        print("full args_list:", len(all_jobs))
        print("new args_list:", len(jobs))
        print(f"Total number of files to process: {len(jobs)}")
    done = 0
    for b0 in range(0, len(jobs), max(batch_files, 1)):
        batch = jobs[b0:b0 + max(batch_files, 1)]
        motions = retarget([s for s, _ in batch], smplx_folder, robot)
        for (src, tgt), md in zip(batch, motions):
            if md is None:                      # the file failed to load: printed and skipped like :62-76
                continue
            _dump(tgt, md, SMPLX_KEYS)
            done += 1
            if verbose:
                print(f"Processed {done}/{len(jobs)}: {tgt}")
    if verbose:
        print("Done. Saved to ", tgt_folder)
    return done


def run_bvh_dataset(src_folder: str, tgt_folder: str, robot: str, override: bool = False, batch_files: int = 128,
                    retarget=None, verbose: bool = True) -> int:
    """``bvh_to_robot_dataset.py`` with ``batch_files`` files per IK launch (LAFAN1: all 77 clips in one)."""
    retarget = retarget or _retarget_bvh_skipping_errors
    jobs = list_bvh_jobs(src_folder, tgt_folder, override, verbose)
    done = 0
    for b0 in range(0, len(jobs), max(batch_files, 1)):
        batch = jobs[b0:b0 + max(batch_files, 1)]
        motions = retarget([s for s, _ in batch], robot)
        for (src, tgt), md in zip(batch, motions):
            if md is None:
                continue
            _dump(tgt, md, BVH_KEYS)
            done += 1
    if verbose:
        print("Done. saved to ", tgt_folder)
    return done


def _retarget_bvh_skipping_errors(bvh_files: Sequence[str], tgt_robot: str) -> List[Optional[Dict]]:
    """:func:`retarget_bvh_files` with the script's print-and-skip for files that fail to load (:77-82)."""
    from .utils.lafan1 import load_lafan1_packed
    gmr = GeneralMotionRetargeting("bvh", tgt_robot, actual_human_height=1.75)
    ok, clips = [], []
    for i, f in enumerate(bvh_files):
        try:
            clips.append(load_lafan1_packed(f, gmr.human_body_names)[0])
            ok.append(i)
        except Exception as e:  # noqa: BLE001 - mirrors the script
            print(f"Error loading {f}: {e}")
    out: List[Optional[Dict]] = [None] * len(bvh_files)
    if clips:
        res = retarget_clips("bvh", tgt_robot, clips, [30] * len(clips), actual_human_height=1.75,
                             height_adjust=False, root_origin_offset=False)
        for i, md in zip(ok, res):
            out[i] = md
    return out


def main(argv=None) -> int:
    import argparse
    import pathlib
    ap = argparse.ArgumentParser(description="Retarget a folder of SMPL-X (AMASS) or BVH (LAFAN1) motions to a robot: "
                                             "the reference's dataset scripts on the MI355X kernels")
    ap.add_argument("--source", choices=["smplx", "bvh"], default="smplx")
    ap.add_argument("--robot", default="unitree_g1")
    ap.add_argument("--src_folder", type=str, required=True)
    ap.add_argument("--tgt_folder", type=str, required=True)
    ap.add_argument("--override", default=False, action="store_true")
    ap.add_argument("--num_cpus", default=4, type=int, help="accepted for compatibility; files share GPU launches instead")
    ap.add_argument("--target_fps", default=30, type=int, help="accepted for compatibility (the BVH script ignores it too)")
    ap.add_argument("--smplx_folder", type=str, default=None, help="SMPL-X body models (default: <assets>/body_models)")
    ap.add_argument("--hard_motions", nargs="*", default=None, help="difficulty reports (default: <assets>/hard_motions/{0,1}.txt)")
    ap.add_argument("--batch_files", type=int, default=256, help="files per IK launch")
    a = ap.parse_args(argv)
    if a.source == "bvh":
        run_bvh_dataset(a.src_folder, a.tgt_folder, a.robot, a.override, a.batch_files)
        return 0
    from .params import ASSET_ROOT
    assets = pathlib.Path(ASSET_ROOT)
    smplx_folder = a.smplx_folder or str(assets / "body_models")
    hard = a.hard_motions if a.hard_motions is not None else [str(assets / "hard_motions" / "0.txt"),
                                                                str(assets / "hard_motions" / "1.txt")]
    run_smplx_dataset(a.src_folder, a.tgt_folder, a.robot, smplx_folder, a.override, hard, a.batch_files)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
