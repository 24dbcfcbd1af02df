"""Row N2: the BVH / LAFAN1 source adapter against the reference's own loader run on
tests/golden/synthetic.bvh (tests/golden/make_golden.py::make_bvh) -- pinned parity."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

BVH = os.path.join(GOLDEN, "synthetic.bvh")
G = np.load(os.path.join(GOLDEN, "g_bvh.npz"))


def test_load_lafan1_file_matches_reference():
    from general_motion_retargeting_amd.utils.lafan1 import load_lafan1_file
    frames, height = load_lafan1_file(BVH)
    names = [str(x) for x in G["names"]]
    assert height == float(G["height"]) == 1.75
    assert len(frames) == G["poses"].shape[0] and list(frames[0].keys()) == names
    arr = np.array([[np.concatenate([f[n][0], f[n][1]]) for n in names] for f in frames])
    assert np.abs(arr - G["poses"]).max() <= 1e-14
    # the synthetic file forces quaternion sign flips along time: the de-flipping must match too
    assert np.array_equal(np.sign(arr[..., 3:]), np.sign(G["poses"][..., 3:]))
    assert names[-2:] == ["LeftFootMod", "RightFootMod"]
    li, lt = names.index("LeftFoot"), names.index("LeftToe")
    assert np.array_equal(arr[:, -2, :3], arr[:, li, :3]) and np.array_equal(arr[:, -2, 3:], arr[:, lt, 3:])


def test_packed_layout_feeds_the_retargeter():
    from general_motion_retargeting_amd import GeneralMotionRetargeting
    from general_motion_retargeting_amd.utils.lafan1 import load_lafan1_file, load_lafan1_packed
    g = GeneralMotionRetargeting("bvh", "unitree_g1", actual_human_height=1.75)
    human, h = load_lafan1_packed(BVH, g.human_body_names)
    frames, _ = load_lafan1_file(BVH)
    assert human.shape == (12, 14, 7)
    assert np.array_equal(human, g.pack_frames(frames))          # dict route == packed route
    with pytest.raises(ValueError):
        load_lafan1_packed(BVH, ["NoSuchBone"])


def test_remove_quat_discontinuities_is_the_sequential_rule():
    from general_motion_retargeting_amd.utils.lafan1 import remove_quat_discontinuities
    rng = np.random.default_rng(0)
    q = rng.normal(size=(40, 5, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    ref = q.copy()
    for i in range(1, ref.shape[0]):                              # the reference's loop, restated
        flip = np.sum(ref[i - 1] * ref[i], axis=-1) < np.sum(ref[i - 1] * -ref[i], axis=-1)
        ref[i] = np.where(flip[:, None], -ref[i], ref[i])
    assert np.array_equal(remove_quat_discontinuities(q.copy()), ref)


@pytest.mark.gpu
def test_bvh_file_to_robot_motion_on_gpu(oracle):
    """bvh_to_robot_dataset flow: file -> packed frames -> one launch -> pkl dict; IK vs the oracle."""
    from conftest import get_setup
    from general_motion_retargeting_amd import dataset
    from general_motion_retargeting_amd.utils.lafan1 import load_lafan1_packed
    su = get_setup("bvh", "unitree_g1", 1.75)
    out = dataset.retarget_bvh_files([BVH, BVH], "unitree_g1")
    human, _ = load_lafan1_packed(BVH, su.tt.human_names)
    q_o, _, st = oracle.retarget_streams(su.mb, su.ts, su.model.qpos0[None], human[None])
    assert st[0] == 0
    for md in out:
        assert md["fps"] == 30.0 and md["dof_pos"].shape == (12, 29)
        assert np.abs(md["dof_pos"] - q_o[0, :, 7:]).max() <= 1e-8
        assert np.abs(md["root_pos"] - q_o[0, :, :3]).max() <= 1e-8          # BVH script: no height / origin adjust
        assert md["local_body_pos"].shape == (12, 38, 3)


@pytest.mark.gpu
def test_bvh_dataset_cli_on_gpu(tmp_path, oracle, capsys):
    """``python -m general_motion_retargeting_amd.dataset --source bvh`` end to end: folder walk -> ONE launch for the
    folder's clips -> one pkl per file (reference scripts/bvh_to_robot_dataset.py), skip-if-exists on the second run,
    a file that fails to load is printed and skipped."""
    import shutil
    from conftest import get_setup
    from general_motion_retargeting_amd import dataset, load_robot_motion
    from general_motion_retargeting_amd.utils.lafan1 import load_lafan1_packed
    src, tgt = tmp_path / "lafan1", tmp_path / "out"
    (src / "sub").mkdir(parents=True)
    shutil.copy(BVH, src / "walk1_subject1.bvh")
    shutil.copy(BVH, src / "sub" / "dance2_subject3.bvh")
    (src / "broken.bvh").write_text("not a bvh file")
    assert dataset.main(["--source", "bvh", "--src_folder", str(src), "--tgt_folder", str(tgt), "--robot", "unitree_g1"]) == 0
    assert "Error loading" in capsys.readouterr().out
    got = sorted(str(p.relative_to(tgt)) for p in tgt.rglob("*.pkl"))
    assert got == ["sub/dance2_subject3.pkl", "walk1_subject1.pkl"]
    su = get_setup("bvh", "unitree_g1", 1.75)
    human, _ = load_lafan1_packed(BVH, su.tt.human_names)
    q_o, _, _ = oracle.retarget_streams(su.mb, su.ts, su.model.qpos0[None], human[None])
    md, fps, root_pos, root_rot_wxyz, dof_pos, lbp, names = load_robot_motion(str(tgt / "walk1_subject1.pkl"))
    assert fps == 30 and list(md) == ["root_pos", "root_rot", "dof_pos", "local_body_pos", "fps", "link_body_list"]
    assert np.abs(dof_pos - q_o[0, :, 7:]).max() <= 1e-8 and np.abs(root_rot_wxyz - q_o[0, :, 3:7]).max() <= 1e-8
    assert lbp.shape == (12, 38, 3) and names[0] == "pelvis"
    before = (tgt / "walk1_subject1.pkl").stat().st_mtime_ns
    assert dataset.main(["--source", "bvh", "--src_folder", str(src), "--tgt_folder", str(tgt), "--robot", "unitree_g1"]) == 0
    assert "Skipping" in capsys.readouterr().out and (tgt / "walk1_subject1.pkl").stat().st_mtime_ns == before
