"""Host-side logic that needs no GPU: packing of human_data dicts, error conventions, LPT
sharding, the dataset post-processing (with the FK supplied by the oracle), synthetic generator."""
import os

import numpy as np
import pytest

from conftest import get_setup
from general_motion_retargeting_amd import GeneralMotionRetargeting, TargetNotSet, synth
from general_motion_retargeting_amd.sharding import lpt_partition


def test_pack_frame_and_keyerrors(g1):
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    human, _ = synth.make_streams(g1.model, g1.tt, 1, 2, seed=0)
    hd = synth.streams_to_dicts(g1.tt, human[0])[0]
    hd["extra_body"] = ([0, 0, 0], [1, 0, 0, 0])                      # extra bodies are dropped (:218-220)
    packed = g.pack_frame(g.to_numpy(hd))
    assert np.array_equal(packed, human[0, 0])
    missing = dict(hd); del missing["left_wrist"]
    with pytest.raises(KeyError, match="left_wrist"):
        g.pack_frame(missing)
    noroot = dict(hd); del noroot["pelvis"]
    with pytest.raises(KeyError, match="pelvis"):
        g.pack_frame(noroot)
    assert g.pack_frames([]).shape == (0, 14, 7)


def test_duplicate_human_body_is_target_not_set(g1):
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    g._tables.stages[0].human_names[1] = g._tables.stages[0].human_names[2]
    human, _ = synth.make_streams(g1.model, g1.tt, 1, 1, seed=0)
    with pytest.raises(TargetNotSet):
        g.pack_frame(synth.streams_to_dicts(g1.tt, human[0])[0])


def test_constructor_attributes_match_reference_surface():
    g = GeneralMotionRetargeting("smplx", "hightorque_hi", actual_human_height=1.7, damping=0.25)
    assert g.xml_file.endswith((".xml", ".npz")) and g.max_iter == 10 and g.damping == 0.25 and g.solver == "daqp"
    assert np.allclose(g.ground, [0, 0, -0.05]) and g.use_ik_match_table1 and g.use_ik_match_table2
    assert abs(g.human_scale_table["pelvis"] / (1.7 / 1.8) - g._tables.scale_table["pelvis"] / (1.7 / 1.8)) < 1e-15
    assert g.configuration.q.shape == (g.model.nq,) and np.array_equal(g.configuration.q, g.model.qpos0)
    # ground is subtracted from the LOCAL offset before rotation (:91): preserved quirk
    name = g._tables.stages[0].human_names[0]
    raw = np.array(g.ik_match_table1[g._tables.stages[0].frame_names[0]][3])
    assert np.allclose(g.pos_offsets1[name], raw - g.ground)
    assert g._taskset_blob["damping"][0] == 0.25


def test_error_functions_need_targets():
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    with pytest.raises(TargetNotSet):
        g.error1()
    assert g.scaled_human_data is None
    k = GeneralMotionRetargeting("smplx", "kuavo_s45")      # use_ik_match_table2 = false: an empty task list
    k._raw_frame = np.zeros((len(k.human_body_names), 7))
    with pytest.raises(ValueError):
        k.error2()


def test_binding_accepts_stream_objects(monkeypatch):
    """Every *_dev wrapper and Event.record take a _lib.Stream, a raw pointer or None (one helper)."""
    from general_motion_retargeting_amd import _lib
    import ctypes as C

    class S(_lib.Stream):
        def __init__(self):             # no device here: a Stream object around a fake handle
            self.ptr = C.c_void_p(0x1234)

        def __del__(self):
            pass
    s = S()
    assert _lib._s(s).value == 0x1234 and _lib._s(None) is None and _lib._s(s.ptr) is s.ptr
    seen = {}

    class FakeLib:
        def __getattr__(self, name):
            def f(*a):
                seen[name] = a
                return 0
            return f
    monkeypatch.setattr(_lib, "lib", lambda: FakeLib())
    ev = _lib.Event.__new__(_lib.Event)
    ev.ptr = C.c_void_p(1)
    ev.record(s)
    assert seen["gmr_event_record"][1].value == 0x1234
    fk = _lib.FkHandle.__new__(_lib.FkHandle)
    fk.handle = C.c_void_p(2)
    fk.fk_dev(4, None, None, None, None, stream=s)
    assert seen["gmr_fk_batch_dev"][-1].value == 0x1234
    sx = _lib.SmplxHandle.__new__(_lib.SmplxHandle)
    sx.handle = C.c_void_p(3)
    sx.align_dev(1, 55, None, None, 1, None, None, stream=s)
    sx.joints_dev(1, None, None, None, None, stream=s)
    assert seen["gmr_smplx_align_dev"][-1].value == 0x1234 and seen["gmr_smplx_joints_dev"][-1].value == 0x1234
    so = _lib.Solver.__new__(_lib.Solver)
    so.handle = C.c_void_p(4)
    so.retarget_streams_dev(1, 1, None, None, None, 0, None, None, None, s)
    assert seen["gmr_retarget_streams_dev"][-1].value == 0x1234
    for o in (ev, fk, sx, so):          # fake handles: nothing to release
        o.ptr = o.handle = None


def test_lpt_partition_properties():
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 9000, size=77)                              # LAFAN1-shaped: 77 ragged streams
    parts = lpt_partition(lens, 8)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(77))
    loads = [int(lens[p].sum()) for p in parts]
    assert max(loads) - min(loads) <= lens.max()
    assert lpt_partition(lens, 8) == parts                             # deterministic
    assert lpt_partition([5, 5], 4) == [[0], [1], [], []]
    assert lpt_partition([], 2) == [[], []]


def test_dataset_postprocess_matches_reference_recipe(oracle, g1, monkeypatch):
    """H10 with the FK kernel replaced by the oracle's float32 FK (CPU): checks the recipe itself."""
    from general_motion_retargeting_amd import KinematicsModel, ROBOT_XML_DICT, dataset

    class FakeHandle:
        def __init__(self, tree):
            self.tree = tree

        def fk(self, rp, rr, dof, want_rot=True, want_min_z=False):
            bp, br = oracle.fk_f32(self.tree, rp, rr, dof)
            return bp, br, (float(bp[..., 2].min()) if want_min_z else None)

    km = KinematicsModel(ROBOT_XML_DICT["unitree_g1"])
    km._handle = FakeHandle(km._tree)
    human, q0 = synth.make_streams(g1.model, g1.tt, 1, 8, seed=4)
    qpos, _, _ = oracle.retarget_streams(g1.mb, g1.ts, q0, human)
    qpos = qpos[0]
    md = dataset.postprocess_clip(qpos, km, fps=30.0)
    assert list(md) == ["fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list"]
    assert np.array_equal(md["root_rot"], qpos[:, [4, 5, 6, 3]]) and np.array_equal(md["dof_pos"], qpos[:, 7:])
    assert md["local_body_pos"].dtype == np.float32 and md["local_body_pos"].shape == (8, 38, 3)
    assert md["link_body_list"] == km.body_names and md["root_pos"].dtype == np.float64
    bp, _ = oracle.fk_f32(km._tree, qpos[:, :3].astype(np.float32), qpos[:, [4, 5, 6, 3]].astype(np.float32),
                          qpos[:, 7:].astype(np.float32))
    exp = qpos[:, :3].copy()
    exp[:, 2] -= float(bp[..., 2].min())
    exp[:, :2] -= exp[0, :2]
    assert np.array_equal(md["root_pos"], exp)
    assert np.array_equal(md["root_pos"][0, :2], [0, 0])
    md2 = dataset.postprocess_clip(qpos, km, fps=30.0, height_adjust=False, root_origin_offset=False)   # BVH script
    assert np.array_equal(md2["root_pos"], qpos[:, :3])
    # pkl round trip through the loader (root_rot back to wxyz)
    import os, tempfile
    from general_motion_retargeting_amd import load_robot_motion, save_robot_motion
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "m.pkl")
        save_robot_motion(p, md)
        data, fps, rp, rr, dp, lbp, names = load_robot_motion(p)
        assert fps == 30.0 and np.array_equal(rr, qpos[:, 3:7]) and names == km.body_names
        # N3: the list-valued, protocol-2 variant (convert_pkl_for_training.py:44-74) and what the
        # training-side MotionLoader extracts from either file (motion_loader.py:72-98)
        import pickle
        from general_motion_retargeting_amd.data_loader import motion_arrays
        p2 = os.path.join(d, "m2.pkl")
        save_robot_motion(p2, md, training_compatible=True)
        with open(p2, "rb") as f:
            raw = f.read()
        assert raw[:2] == b"\x80\x02"                                  # pickle protocol 2
        conv = pickle.loads(raw)
        assert all(not isinstance(v, np.ndarray) for v in conv.values())
        assert isinstance(conv["root_pos"], list) and conv["fps"] == 30.0 and conv["link_body_list"] == km.body_names
        a, b = motion_arrays(data), motion_arrays(conv)
        for k in ("root_pos", "root_rot", "dof_pos", "local_body_pos"):
            assert a[k].dtype == np.float32 and np.array_equal(a[k], b[k])
        assert a["num_frames"] == len(qpos) and abs(a["motion_duration"] - len(qpos) / 30.0) < 1e-12
        data2, _, _, rr2, _, _, _ = load_robot_motion(p2)              # the loader accepts the list form too
        assert np.array_equal(rr2, qpos[:, 3:7])


def test_synthetic_generator_is_seeded_and_invertible(oracle, g1):
    h1, q1 = synth.make_streams(g1.model, g1.tt, 3, 5, seed=42)
    h2, q2 = synth.make_streams(g1.model, g1.tt, 3, 5, seed=42)
    assert np.array_equal(h1, h2) and np.array_equal(q1, q2)
    h3, _ = synth.make_streams(g1.model, g1.tt, 2, 5, seed=43)
    assert np.array_equal(h3[0], h1[1])                                # stream s uses seed + s
    # without noise, preprocessing the generated raw data reproduces the task-frame poses of q*
    h, q0, truth = synth.make_streams(g1.model, g1.tt, 1, 3, seed=1, pos_noise=0, rot_noise_deg=0, return_truth=True)
    tgt = oracle.preprocess(g1.ts, h[0, 2])
    xpos, xquat = synth.fk_numpy(g1.model, truth[0, 2])
    st = g1.tt.stages[0]
    for fr, hb in zip(st.frame_names, st.human_names):
        i, b = g1.tt.human_names.index(hb), g1.model.body_id(fr)
        assert np.abs(tgt[i, :3] - xpos[b]).max() < 1e-12
        assert min(np.abs(tgt[i, 3:] - xquat[b]).max(), np.abs(tgt[i, 3:] + xquat[b]).max()) < 1e-12
    lo, hi = g1.model.range_lo, g1.model.range_hi
    assert (truth[..., 7:] >= lo).all() and (truth[..., 7:] <= hi).all()


def test_dataset_drivers_follow_the_reference_file_semantics(tmp_path, capsys):
    """Folder walk, filters, skip-if-exists / --override, one pkl per input (scripts/smplx_to_robot_dataset.py:171-242,
    scripts/bvh_to_robot_dataset.py:60-157); the retargeting itself is replaced by a stand-in (no GPU here)."""
    import pickle
    from general_motion_retargeting_amd import dataset
    src, tgt = tmp_path / "amass", tmp_path / "out"
    files = ["A/walk_10_stageii.npz", "A/walk_2_stageii.npz", "A/walk_2_stagei.npz", "A/notes.txt", "B/BMLrub_jump.npz",
             "B/crawl_1.npz", "B/run_1.pkl", "B/hard_one_stageii.npz", "B/broken.npz", "C/sub/upstairs_3.npz"]
    for f in files:
        (src / f).parent.mkdir(parents=True, exist_ok=True)
        (src / f).write_bytes(b"x")
    hard = tmp_path / "0.txt"
    hard.write_text("Motions with difficulty > 5:\nMotion: hard_one_stageii.pkl, Difficulty: 14.07\nsomething else\n")
    assert dataset.load_hard_motions([str(hard), str(tmp_path / "missing.txt")]) == ["hard_one_stageii"]
    calls = []

    def fake_load(path):
        if "broken" in path:
            raise ValueError("bad zip file")                      # a file that fails to load: printed + skipped (:62-76)
        return {"trans": np.zeros((2, 3)), "path": path}

    def fake_retarget(clips, paths):
        calls.append(list(paths))
        assert [c["path"] for c in clips] == list(paths)
        return [{"fps": 30.0, "root_pos": np.zeros((2, 3)), "root_rot": np.zeros((2, 4)), "dof_pos": np.zeros((2, 29)),
                 "local_body_pos": np.zeros((2, 38, 3), np.float32), "link_body_list": ["pelvis"], "extra": 1} for _ in paths]

    n = dataset.run_smplx_dataset(str(src), str(tgt), "unitree_g1", "models", False, [str(hard)], batch_files=2,
                                  retarget=fake_retarget, load=fake_load, loader_workers=2)
    got = sorted(str(p.relative_to(tgt)) for p in tgt.rglob("*.pkl"))
    assert got == ["A/walk_10_stageii.pkl", "A/walk_2_stageii.pkl", "B/run_1.pkl"] and n == 3
    flat = [os.path.relpath(p, src) for c in calls for p in c]
    assert all(len(c) <= 2 for c in calls)                        # clips grouped into launches of <= batch_files
    all_jobs, kept = dataset.list_smplx_jobs(str(src), str(tgt / "nothing_here"), False, ["hard_one_stageii"])
    a_files = [os.path.relpath(s_, src) for s_, _ in kept if "/A/" in s_]
    assert a_files == ["A/walk_2_stageii.npz", "A/walk_10_stageii.npz"]   # natsorted, *_stagei.npz and non-motion files skipped
    assert sorted(flat) == ["A/walk_10_stageii.npz", "A/walk_2_stageii.npz", "B/run_1.pkl"]
    assert not any(x in f for f in flat for x in ("BMLrub", "crawl", "upstairs", "hard_one", "stagei.npz"))
    with open(tgt / "B" / "run_1.pkl", "rb") as f:
        md = pickle.load(f)
    assert list(md) == ["fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list"]
    out = capsys.readouterr().out
    assert "full args_list: 8" in out and "new args_list: 4" in out and "Processed 3/4" in out and "Error loading" in out
    # second run: everything that exists is skipped; --override redoes it
    calls.clear()
    assert dataset.run_smplx_dataset(str(src), str(tgt), "unitree_g1", "models", False, [str(hard)], retarget=fake_retarget,
                                     load=fake_load, loader_workers=0) == 0
    assert calls == []                                            # only the broken file was left, and it does not load
    dataset.run_smplx_dataset(str(src), str(tgt), "unitree_g1", "models", True, [str(hard)], retarget=fake_retarget, load=fake_load)
    assert len([p for c in calls for p in c]) == 3
    # BVH driver: sorted walk, .bvh only, skip message, the BVH script's key order
    bsrc, btgt = tmp_path / "lafan", tmp_path / "lafan_out"
    for f in ["walk1_subject1.bvh", "aiming1_subject1.bvh", "readme.md", "sub/run2_subject4.bvh"]:
        (bsrc / f).parent.mkdir(parents=True, exist_ok=True)
        (bsrc / f).write_bytes(b"x")
    (btgt).mkdir()
    (btgt / "walk1_subject1.pkl").write_bytes(b"old")
    seen = []

    def fake_bvh(clips, paths):
        seen.extend(paths)
        return [{"fps": 30, "root_pos": np.zeros((1, 3)), "root_rot": np.zeros((1, 4)), "dof_pos": np.zeros((1, 29)),
                 "local_body_pos": np.zeros((1, 38, 3), np.float32), "link_body_list": ["pelvis"]} for _ in paths]

    assert dataset.run_bvh_dataset(str(bsrc), str(btgt), "unitree_g1", retarget=fake_bvh, load=lambda f: np.zeros((1, 15, 7))) == 2
    assert sorted(os.path.relpath(p, bsrc) for p in seen) == ["aiming1_subject1.bvh", "sub/run2_subject4.bvh"]
    assert "Skipping" in capsys.readouterr().out and (btgt / "walk1_subject1.pkl").read_bytes() == b"old"
    with open(btgt / "sub" / "run2_subject4.pkl", "rb") as f:
        assert list(pickle.load(f)) == ["root_pos", "root_rot", "dof_pos", "local_body_pos", "fps", "link_body_list"]
