"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs."""
import os
import numpy as np
import pytest

from conftest import ALL_CONFIGS, get_setup

pytestmark = pytest.mark.gpu

TOL_RAD = 1e-9      # FP64 kernel vs FP64 oracle; the contract (BASELINE.json) is 1e-4 rad
TOL_POS = 1e-9


def _quat_dist(a, b):
    d = np.abs(np.sum(a * b, axis=-1))
    return 2.0 * np.arccos(np.clip(d, -1.0, 1.0))


def _compare(q_hip, q_orc):
    joint = np.abs(q_hip[..., 7:] - q_orc[..., 7:]).max()
    pos = np.abs(q_hip[..., :3] - q_orc[..., :3]).max()
    # geodesic distance loses precision near 0 (acos); compare components up to sign instead
    qa, qb = q_hip[..., 3:7], q_orc[..., 3:7]
    sgn = np.sign(np.sum(qa * qb, axis=-1, keepdims=True))
    rot = np.abs(qa - sgn * qb).max()
    return joint, pos, rot


@pytest.fixture(scope="module")
def hip():
    from general_motion_retargeting_amd import _lib
    _lib.require_gpu()
    return _lib


def test_backend_is_gfx950(hip):
    info = hip.lib().gmr_backend_info().decode()
    assert "gfx950" in info, info


@pytest.mark.parametrize("S,T", [(1, 1), (3, 7), (16, 40)])
def test_ik_streams_g1_matches_oracle(hip, oracle, g1, S, T):
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(g1.model, g1.tt, S, T, seed=11)
    q_o, ns_o, st_o = oracle.retarget_streams(g1.mb, g1.ts, q0, human)
    sol = hip.Solver(g1.mb, g1.ts)
    q_h, ns_h, st_h = sol.retarget_streams(q0, human)
    assert (st_h == 0).all() and (st_o == 0).all()
    assert np.array_equal(ns_h, ns_o), "solver-iteration counts differ (branch fidelity)"
    joint, pos, rot = _compare(q_h, q_o)
    assert joint <= TOL_RAD and pos <= TOL_POS and rot <= TOL_RAD, (joint, pos, rot)


@pytest.mark.parametrize("src,robot", ALL_CONFIGS)
def test_ik_streams_all_configs(hip, oracle, src, robot):
    from general_motion_retargeting_amd import synth
    su = get_setup(src, robot, 1.7)
    human, q0 = synth.make_streams(su.model, su.tt, 4, 12, seed=5)
    q_o, ns_o, st_o = oracle.retarget_streams(su.mb, su.ts, q0, human)
    q_h, ns_h, st_h = hip.Solver(su.mb, su.ts).retarget_streams(q0, human)
    assert (st_h == 0).all()
    assert np.array_equal(ns_h, ns_o)
    joint, pos, rot = _compare(q_h, q_o)
    assert joint <= TOL_RAD and pos <= TOL_POS and rot <= TOL_RAD, (joint, pos, rot)


def test_ik_offset_to_ground_and_ragged(hip, oracle, g1):
    from general_motion_retargeting_amd import synth
    S, T = 5, 9
    human, q0 = synth.make_streams(g1.model, g1.tt, S, T, seed=3)
    lens = np.array([9, 1, 4, 0, 7], dtype=np.int32)
    sol = hip.Solver(g1.mb, g1.ts)
    q_h, ns_h, st_h = sol.retarget_streams(q0, human, lens=lens, flags=hip.FLAG_OFFSET_TO_GROUND)
    for s in range(S):
        n = int(lens[s])
        if n == 0:
            assert (ns_h[s] == 0).all()
            continue
        q_o, ns_o, st_o = oracle.retarget_streams(g1.mb, g1.ts, q0[s:s + 1], human[s:s + 1, :n], offset_to_ground=True)
        assert np.array_equal(ns_h[s, :n], ns_o[0])
        joint, pos, rot = _compare(q_h[s, :n], q_o[0])
        assert joint <= TOL_RAD and pos <= TOL_POS and rot <= TOL_RAD
        assert (ns_h[s, n:] == 0).all() and (q_h[s, n:] == 0).all()


def test_ik_joint_limits_active(hip, oracle, g1):
    """Targets far outside the reachable set drive many joints onto their limits (active-set path)."""
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 6, 10, seed=21)
    rng = np.random.default_rng(0)
    human[..., :3] += rng.normal(0, 0.3, size=human[..., :3].shape)     # scatter keypoints
    rv = rng.normal(0, 1.0, size=human.shape[:-1] + (3,))
    human[..., 3:] = synth.quat_mul(human[..., 3:], synth.rotvec_quat(rv))
    q_o, ns_o, st_o = oracle.retarget_streams(g1.mb, g1.ts, q0, human)
    q_h, ns_h, st_h = hip.Solver(g1.mb, g1.ts).retarget_streams(q0, human)
    assert (st_h == 0).all() and (st_o == 0).all()
    lo, hi = g1.model.range_lo, g1.model.range_hi
    th = q_h[..., 7:]
    assert (th >= lo - 1e-9).all() and (th <= hi + 1e-9).all(), "joint limits violated"
    at_limit = (np.abs(th - lo) < 1e-6) | (np.abs(th - hi) < 1e-6)
    assert at_limit.sum() > 20, "test did not exercise the active set"
    assert np.array_equal(ns_h, ns_o)
    joint, pos, rot = _compare(q_h, q_o)
    assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (joint, pos, rot)


def test_ik_bitwise_reproducible_and_shard_invariant(hip, g1):
    """Same stream => same bits, whatever batch it is launched in (sharding across ranks is exact)."""
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 8, 12, seed=7)
    sol = hip.Solver(g1.mb, g1.ts)
    q_all, ns_all, _ = sol.retarget_streams(q0, human)
    q_again, _, _ = sol.retarget_streams(q0, human)
    assert np.array_equal(q_all, q_again)
    q_a, _, _ = sol.retarget_streams(q0[:3], human[:3])
    q_b, _, _ = sol.retarget_streams(q0[3:], human[3:])
    assert np.array_equal(np.concatenate([q_a, q_b]), q_all)


def test_ik_nonfinite_input_sets_status(hip, g1):
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 2, 3, seed=1)
    human[1, 1, 0, 3:] = np.nan          # NaN orientation of the root body in frame 1 of stream 1
    q_h, ns_h, st_h = hip.Solver(g1.mb, g1.ts).retarget_streams(q0, human)
    assert st_h[0] == 0 and st_h[1] == hip.STATUS_QP_FAILED
    assert np.isfinite(q_h[0]).all()


def test_fk_batch_matches_oracle_and_golden(hip, oracle):
    import os
    from conftest import GOLDEN
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    g = np.load(os.path.join(GOLDEN, "g_fk.npz"))
    for robot in params.ROBOT_XML_DICT:
        if robot + "__error" in g.files:
            continue
        tree = load_kinematics_tree(params.ROBOT_XML_DICT[robot])
        fk = hip.FkHandle(tree)
        bp, br, mz = fk.fk(g[robot + "__root_pos"], g[robot + "__root_rot"], g[robot + "__dof"], want_min_z=True)
        assert np.abs(bp - g[robot + "__body_pos"]).max() <= 2e-6, robot
        assert np.abs(br - g[robot + "__body_rot"]).max() <= 2e-6, robot
        assert mz == bp[..., 2].min()
        obp, obr = oracle.fk_f32(tree, g[robot + "__root_pos"], g[robot + "__root_rot"], g[robot + "__dof"])
        assert np.abs(bp - obp).max() <= 2e-6 and np.abs(br - obr).max() <= 2e-6


@pytest.mark.parametrize("robot", ["unitree_g1", "fourier_n1"])
def test_fk_large_and_nonfinite_angles_match_reference(hip, robot):
    """Unwrapped angles up to 1e5 rad and NaN / Inf angles (g_fk_large.npz, reference outputs): the kernel's
    small-argument sincos hands such lanes to the library path; NaNs exactly where the reference has them."""
    import os
    from conftest import GOLDEN
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    G = np.load(os.path.join(GOLDEN, "g_fk_large.npz"))
    fk = hip.FkHandle(load_kinematics_tree(params.ROBOT_XML_DICT[robot]))
    for want_rot in (True, False):
        bp, br, _ = fk.fk(G[robot + "__root_pos"], G[robot + "__root_rot"], G[robot + "__dof"], want_rot=want_rot)
        for got, ref in ((bp, G[robot + "__body_pos"]),) + (((br, G[robot + "__body_rot"]),) if want_rot else ()):
            assert np.array_equal(np.isnan(got), np.isnan(ref))
            assert np.nanmax(np.abs(got - ref)) <= 4e-6


def test_fk_batch_large_and_edge_sizes(hip, oracle):
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    tree = load_kinematics_tree(params.ROBOT_XML_DICT["unitree_g1"])
    fk = hip.FkHandle(tree)
    rng = np.random.default_rng(0)
    for B in (1, 5, 6, 7, 1000, 20011):
        dof = rng.uniform(-1, 1, size=(B, fk.ndof)).astype(np.float32)
        rp = rng.normal(size=(B, 3)).astype(np.float32)
        rq = rng.normal(size=(B, 4)); rq = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)
        bp, br, mz = fk.fk(rp, rq, dof, want_min_z=True)
        obp, obr = oracle.fk_f32(tree, rp, rq, dof)
        assert np.abs(bp - obp).max() <= 5e-6 and np.abs(br - obr).max() <= 5e-6, B
        assert mz == bp[..., 2].min()
        bp2, br2, _ = fk.fk(rp, rq, dof, want_rot=False)
        assert br2 is None and np.array_equal(bp2, bp)


# ----------------------------------------------------------------------------------------------
# the reference-shaped host API on the GPU
# ----------------------------------------------------------------------------------------------
def test_shim_per_frame_equals_clip_equals_oracle(hip, oracle, g1):
    from general_motion_retargeting_amd import GeneralMotionRetargeting, synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 2, 10, seed=13)
    q_o, ns_o, _ = oracle.retarget_streams(g1.mb, g1.ts, q0, human)
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    frames = synth.streams_to_dicts(g1.tt, human[0])
    per_frame = np.array([g.retarget(f) for f in frames])            # scripts/smplx_to_robot.py loop
    assert per_frame.dtype == np.float64 and per_frame.shape == (10, 36)
    assert np.abs(per_frame - q_o[0]).max() <= TOL_RAD
    assert np.array_equal(g.last_num_solves, ns_o[0, -1])
    assert set(g.scaled_human_data) == set(g1.tt.human_names)
    g2 = GeneralMotionRetargeting("smplx", "unitree_g1")
    clip = g2.retarget_clip(frames[:6])
    clip2 = g2.retarget_clip(frames[6:])                              # continues from the stored configuration
    # the QP active set is warm-started inside a launch but not across launches: same minimiser,
    # different pivoting path => agreement to rounding, not to the bit
    assert np.abs(np.concatenate([clip, clip2]) - per_frame).max() <= 1e-12
    out = g2.retarget(frames[0])
    out[:] = 0                                                        # a fresh copy, not a view of the state
    assert np.abs(g2.configuration.q).max() > 0
    with pytest.raises(RuntimeError):
        bad = synth.streams_to_dicts(g1.tt, human[1])[0]
        bad["pelvis"] = (np.array([np.nan, 0, 0]), bad["pelvis"][1])
        GeneralMotionRetargeting("smplx", "unitree_g1").retarget(bad)


def test_hip_preprocess_matches_reference_golden(hip):
    """H2/H3 on the HIP path, pinned DIRECTLY: the kernel's own preprocessed targets (tgt_out = what it hands to
    the residuals, i.e. task.set_target's poses / scaled_human_data, reference motion_retarget.py:117-136,203-270)
    against the fixtures generated by running the reference's Python -- all 14 configs x 2 heights x 2 ground flags.
    Tolerance 2e-15 absolute on metres / unit quaternions (the fixture values are < 2 in magnitude: <= 9 ulp;
    the device contracts multiply-adds, NumPy does not)."""
    from conftest import ALL_CONFIGS, GOLDEN, get_setup
    import os
    G = np.load(os.path.join(GOLDEN, "g_pre.npz"))
    worst = 0.0
    for src, robot in ALL_CONFIGS:
        for hname, height in (("none", None), ("h162", 1.62)):
            su = get_setup(src, robot, height)
            sol = hip.Solver(su.mb, su.ts)
            q0 = su.model.qpos0[None].copy()
            for ground in (0, 1):
                tag = f"{src}__{robot}__{hname}__g{ground}"
                raw = G[tag + "__raw"]
                exp = G[tag + "__scaled"]
                flags = (hip.FLAG_OFFSET_TO_GROUND if ground else 0)
                # once without a solve (update_targets() alone) and once as part of a real retarget() launch
                for fl in (flags | hip.FLAG_EVAL_ONLY, flags):
                    q, ns, st, tg, er = sol.retarget_streams(q0, raw[None, None], flags=fl, want_targets=True,
                                                             want_errors=True)
                    got = tg[0, 0]
                    # (fbx has no body named *foot*: with offset_to_ground the reference's `lowest` stays +inf and
                    # every target z becomes -inf (:252-270); the fixture holds that, the kernel reproduces it and
                    # then reports the stream as failed instead of solving on non-finite targets)
                    both_inf = np.isinf(got) & np.isinf(exp) & (np.sign(got) == np.sign(exp))
                    assert np.array_equal(np.isfinite(got) | both_inf, np.ones_like(got, dtype=bool)), tag
                    assert np.array_equal(np.isinf(got), np.isinf(exp)), tag
                    with np.errstate(invalid="ignore"):
                        d = np.where(both_inf, 0.0, np.abs(got - exp))
                    worst = max(worst, float(d.max()))
                    assert d.max() <= 2e-15, (tag, float(d.max()))
                    assert st[0] == (0 if np.isfinite(exp).all() or (fl & hip.FLAG_EVAL_ONLY) else hip.STATUS_QP_FAILED), tag
                    if fl & hip.FLAG_EVAL_ONLY:
                        assert (ns == 0).all() and np.abs(q[0, 0] - q0[0]).max() <= 1e-15
            sol.close()
    print("max |tgt_out - reference scaled_human_data| =", worst)


def test_shim_scaled_human_data_and_errors_come_from_the_kernel(hip, oracle, g1):
    """scaled_human_data / error1() / error2() (reference :118-124, :188-200) follow the LAST retargeted frame for
    every entry point, and after update_targets() alone they come from a solve-free launch."""
    import os
    from conftest import GOLDEN
    from general_motion_retargeting_amd import GeneralMotionRetargeting, synth
    from general_motion_retargeting_amd.motion_retarget import TargetNotSet
    G = np.load(os.path.join(GOLDEN, "g_pre.npz"))
    g = GeneralMotionRetargeting("smplx", "unitree_g1", actual_human_height=1.62)
    tag = "smplx__unitree_g1__h162__g1"
    names = [str(x) for x in G[tag + "__names"]]
    raw = G[tag + "__raw"]
    hd = {n: (raw[i, :3].copy(), raw[i, 3:].copy()) for i, n in enumerate(names)}
    hd["not_in_scale_table"] = (np.zeros(3), np.array([1.0, 0, 0, 0]))
    g.update_targets(hd, offset_to_ground=True)
    shd = g.scaled_human_data
    assert "not_in_scale_table" not in shd and list(shd)[0] == g.human_root_name
    got = np.array([np.concatenate(shd[n]) for n in names])
    assert np.abs(got - G[tag + "__scaled"]).max() <= 2e-15
    # errors at an arbitrary configuration, no solve: the oracle's stage error on the oracle's targets
    human, q0, truth = synth.make_streams(g1.model, g1.tt, 1, 5, seed=9, return_truth=True)
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    with pytest.raises(TargetNotSet):
        g.error1()
    frames = synth.streams_to_dicts(g1.tt, human[0])
    g.update_targets(frames[1])
    g.configuration.update(truth[0, 0])
    tgt = oracle.preprocess(g1.ts, human[0, 1])
    for stage, fn in ((0, g.error1), (1, g.error2)):
        _, E = oracle.stage_error(g1.mb, g1.ts, stage, truth[0, 0], tgt)
        assert abs(fn() - E) < 1e-12
    assert np.array_equal(g.configuration.q, truth[0, 0])            # evaluation does not move the configuration
    # after retarget(): the errors of the configuration the frame ended with
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    q = None
    for t in range(3):
        q = g.retarget(frames[t])
    tgt = oracle.preprocess(g1.ts, human[0, 2])
    for stage, fn in ((0, g.error1), (1, g.error2)):
        _, E = oracle.stage_error(g1.mb, g1.ts, stage, q, tgt)
        assert abs(fn() - E) < 1e-10
    assert np.abs(np.concatenate(g.scaled_human_data["pelvis"]) - tgt[g1.tt.human_names.index("pelvis")]).max() < 1e-14
    # after retarget_clip / retarget_packed: still the last frame's (ADVICE: stale targets)
    qc = g.retarget_clip(frames[3:5])
    tgt = oracle.preprocess(g1.ts, human[0, 4])
    _, E = oracle.stage_error(g1.mb, g1.ts, 1, qc[-1], tgt)
    assert abs(g.error2() - E) < 1e-10
    assert np.abs(g.scaled_human_data["pelvis"][0] - tgt[g1.tt.human_names.index("pelvis"), :3]).max() < 1e-14
    qp = g.retarget_packed(human[0, 0])
    tgt = oracle.preprocess(g1.ts, human[0, 0])
    _, E = oracle.stage_error(g1.mb, g1.ts, 0, qp, tgt)
    assert abs(g.error1() - E) < 1e-10 and set(g.scaled_human_data) == set(g1.tt.human_names)


def test_shim_offset_to_ground(hip, oracle, g1):
    from general_motion_retargeting_amd import GeneralMotionRetargeting, synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 1, 4, seed=2)
    q_o, _, _ = oracle.retarget_streams(g1.mb, g1.ts, q0, human, offset_to_ground=True)
    g = GeneralMotionRetargeting("smplx", "unitree_g1")
    for t, f in enumerate(synth.streams_to_dicts(g1.tt, human[0])):
        assert np.abs(g.retarget(f, offset_to_ground=True) - q_o[0, t]).max() <= TOL_RAD
    feet = [g.scaled_human_data[n][0][2] for n in g.scaled_human_data if "foot" in n]
    assert abs(min(feet) - 0.1) < 1e-12


def test_kinematics_model_fitted_shape_matches_reference_golden(hip):
    """forward_kinematics(fitted_shape=...) (reference kinematics_model.py:213-246, :224): the FK kernel on a tree whose
    local translations are scaled per body, against the reference's own outputs."""
    import os
    from conftest import GOLDEN
    from general_motion_retargeting_amd import KinematicsModel, ROBOT_XML_DICT
    G = np.load(os.path.join(GOLDEN, "g_fk_aux.npz"))
    for robot in ("unitree_g1", "booster_t1"):
        km = KinematicsModel(ROBOT_XML_DICT[robot])
        bp, br = km.forward_kinematics(G[f"{robot}__root_pos"], G[f"{robot}__root_rot"], G[f"{robot}__dof"],
                                       fitted_shape=G[f"{robot}__fitted_shape"])
        assert np.abs(bp - G[f"{robot}__fitted_body_pos"]).max() <= 5e-6
        assert np.abs(br - G[f"{robot}__fitted_body_rot"]).max() <= 5e-6
        bp0, _ = km.forward_kinematics(G[f"{robot}__root_pos"], G[f"{robot}__root_rot"], G[f"{robot}__dof"])
        assert np.abs(bp0 - bp).max() > 1e-3                                # the plain handle is untouched and differs


def test_kinematics_model_numpy_and_torch_paths(hip, oracle):
    import os
    from conftest import GOLDEN
    from general_motion_retargeting_amd import KinematicsModel, ROBOT_XML_DICT
    g = np.load(os.path.join(GOLDEN, "g_fk.npz"))
    km = KinematicsModel(ROBOT_XML_DICT["unitree_g1"], device="cuda:0")
    assert km.num_dof == 29 and km.num_joint == 38 and km.body_names[0] == "pelvis"
    rp, rr, dof = g["unitree_g1__root_pos"], g["unitree_g1__root_rot"], g["unitree_g1__dof"]
    bp, br = km.forward_kinematics(rp, rr, dof)
    assert np.abs(bp - g["unitree_g1__body_pos"]).max() <= 2e-6
    bp3, br3 = km.forward_kinematics(rp.reshape(4, 6, 3), rr.reshape(4, 6, 4), dof.reshape(4, 6, 29))   # leading dims
    assert bp3.shape == (4, 6, 38, 3) and np.array_equal(bp3.reshape(bp.shape), bp)
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        t = lambda a: torch.from_numpy(a).to("cuda:0")
        tbp, tbr = km.forward_kinematics(t(rp), t(rr), t(dof))        # zero-copy on torch device memory
        assert tbp.is_cuda and tbp.dtype == torch.float32
        assert np.array_equal(tbp.cpu().numpy(), bp) and np.array_equal(tbr.cpu().numpy(), br)
        lowest = torch.min(tbp[..., 2]).item()                        # what the dataset script does (:125)
        _, _, mz = km.forward_kinematics(t(rp), t(rr), t(dof), return_min_z=True)
        assert mz == lowest
    cbp, _ = km.forward_kinematics(torch.from_numpy(rp), torch.from_numpy(rr), torch.from_numpy(dof))
    assert np.array_equal(cbp.numpy(), bp)


def test_dataset_harness_many_clips_one_launch(hip, oracle, g1):
    from general_motion_retargeting_amd import dataset, synth
    lens = [7, 12, 1, 9]
    human, q0 = synth.make_streams(g1.model, g1.tt, 4, 12, seed=31)
    clips = [human[i, :n] for i, n in enumerate(lens)]
    out = dataset.retarget_clips("smplx", "unitree_g1", clips, fps=[30.0] * 4)
    assert len(out) == 4
    for i, n in enumerate(lens):
        q_o, _, _ = oracle.retarget_streams(g1.mb, g1.ts, q0[i:i + 1], human[i:i + 1, :n])
        md = out[i]
        assert md["dof_pos"].shape == (n, 29) and np.abs(md["dof_pos"] - q_o[0, :, 7:]).max() <= TOL_RAD
        assert np.abs(md["root_rot"] - q_o[0][:, [4, 5, 6, 3]]).max() <= TOL_RAD
        assert md["local_body_pos"].shape == (n, 38, 3) and md["local_body_pos"].dtype == np.float32
        assert np.abs(md["root_pos"][0, :2]).max() == 0
        from general_motion_retargeting_amd import KinematicsModel, ROBOT_XML_DICT
        km = KinematicsModel(ROBOT_XML_DICT["unitree_g1"])
        bp, _ = km.forward_kinematics(md["root_pos"], md["root_rot"], md["dof_pos"])
        assert abs(float(bp[..., 2].min())) < 2e-6                    # lowest body part sits on the ground


def test_full_size_config2_properties(hip, oracle, g1):
    """BASELINE.json configs[1] at full size (S=100 x T=100): iteration counts and joint angles
    against the oracle on every frame, limits respected, unit root quaternions, streams independent."""
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 100, 100, seed=0)
    sol = hip.Solver(g1.mb, g1.ts)
    q_h, ns_h, st_h = sol.retarget_streams(q0, human)
    assert (st_h == 0).all()
    assert (ns_h >= 1).all() and (ns_h <= 11).all()
    assert np.abs(np.linalg.norm(q_h[..., 3:7], axis=-1) - 1).max() < 1e-14
    assert (q_h[..., 7:] >= g1.model.range_lo - 1e-12).all() and (q_h[..., 7:] <= g1.model.range_hi + 1e-12).all()
    import os
    q_o, ns_o, st_o = oracle.retarget_streams(g1.mb, g1.ts, q0, human, nthreads=os.cpu_count() or 1)
    assert np.array_equal(ns_h, ns_o)
    joint, pos, rot = _compare(q_h, q_o)
    assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (joint, pos, rot)   # contract: 1e-4 rad
    perm = np.random.default_rng(0).permutation(100)[:17]
    q_p, _, _ = sol.retarget_streams(q0[perm], human[perm])
    assert np.array_equal(q_p, q_h[perm])


def test_mixed_robot_batch_concurrent_streams(hip, oracle):
    """configs[3] shape: several robots in one call, one kernel per robot model on its own HIP stream."""
    from general_motion_retargeting_amd import dataset, synth
    specs = [("smplx", "unitree_g1", None), ("smplx", "booster_t1", 1.7), ("smplx", "hightorque_hi", None),
             ("bvh", "engineai_pm01", None), ("smplx", "kuavo_s45", 1.6), ("smplx", "stanford_toddy", None),
             ("smplx", "fourier_n1", None)]
    groups, setups = [], []
    for i, (src, robot, h) in enumerate(specs):
        su = get_setup(src, robot, h)
        human, q0 = synth.make_streams(su.model, su.tt, 3, 8, seed=50 + i)
        groups.append({"src_human": src, "tgt_robot": robot, "actual_human_height": h, "human": human})
        setups.append((su, human, q0))
    res = dataset.retarget_mixed(groups)
    for (su, human, q0), (q_h, ns_h, st_h) in zip(setups, res):
        q_o, ns_o, _ = oracle.retarget_streams(su.mb, su.ts, q0, human)
        assert (st_h == 0).all() and np.array_equal(ns_h, ns_o)
        joint, pos, rot = _compare(q_h, q_o)
        assert joint <= TOL_RAD and pos <= TOL_POS and rot <= TOL_RAD, (su.robot, joint, pos, rot)


def test_generic_large_robot_nvp48(hip, oracle, tmp_path):
    """A synthetic 40-hinge chain robot (nv = 46 = GMR_MAX_DOF) exercises the NVP=48 instantiation."""
    import json
    from general_motion_retargeting_amd import synth
    from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset
    from general_motion_retargeting_amd.mjcf import compile_mjcf
    axes = ["1 0 0", "0 1 0", "0 0 1"]
    def chain(prefix, n, pos):
        s, e = "", ""
        for i in range(n):
            s += f'<body name="{prefix}{i}" pos="{pos if i == 0 else "0 0 -0.08"}"><joint name="{prefix}j{i}" axis="{axes[i % 3]}" range="-1.2 1.2"/>'
            e += "</body>"
        return s + e
    xml = ('<mujoco model="many"><compiler angle="radian"/><worldbody><body name="base" pos="0 0 1"><freejoint/>'
           + chain("a", 10, "0 0.1 0") + chain("b", 10, "0 -0.1 0") + chain("c", 10, "0.1 0 0.2") + chain("d", 10, "-0.1 0 0.2")
           + "</body></worldbody></mujoco>")
    p = tmp_path / "many.xml"
    p.write_text(xml)
    model = compile_mjcf(str(p))
    assert model.nv == 46
    names = ["root", "ha", "hb", "hc", "hd", "ma", "mb"]
    frames = ["base", "a9", "b9", "c9", "d9", "a4", "b4"]
    tbl = {f: [h, 50, 10, [0, 0, 0], [1, 0, 0, 0]] for f, h in zip(frames, names)}
    cfg = {"robot_root_name": "base", "human_root_name": "root", "ground_height": 0.0, "human_height_assumption": 1.8,
           "use_ik_match_table1": True, "use_ik_match_table2": True, "human_scale_table": {n: 1.0 for n in names},
           "ik_match_table1": tbl, "ik_match_table2": json.loads(json.dumps(tbl))}
    tt = build_task_tables(cfg)
    mb, ts = pack_model(model), pack_taskset(model, tt)
    human, q0 = synth.make_streams(model, tt, 3, 6, seed=8)
    q_o, ns_o, st_o = oracle.retarget_streams(mb, ts, q0, human)
    q_h, ns_h, st_h = hip.Solver(mb, ts).retarget_streams(q0, human)
    assert (st_h == 0).all() and (st_o == 0).all() and np.array_equal(ns_h, ns_o)
    joint, pos, rot = _compare(q_h, q_o)
    assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (joint, pos, rot)


def test_launch_shapes_agree(hip, oracle, g1):
    """1 wave per stream (many streams) and main + 3 helper waves per stream (few streams) are the same
    algorithm: identical solve counts, results equal to rounding; automatic choice by stream count."""
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(g1.model, g1.tt, 12, 8, seed=77)
    sol = hip.Solver(g1.mb, g1.ts)
    sol.set_waves(1)
    q1, ns1, st1 = sol.retarget_streams(q0, human)
    sol.set_waves(4)
    q4, ns4, st4 = sol.retarget_streams(q0, human)
    sol.set_waves(0)
    qa, nsa, _ = sol.retarget_streams(q0, human)
    assert (st1 == 0).all() and (st4 == 0).all()
    assert np.array_equal(ns1, ns4) and np.abs(q1 - q4).max() <= 1e-12
    assert np.array_equal(qa, q4)                                # 12 streams -> helper shape
    q_o, ns_o, _ = oracle.retarget_streams(g1.mb, g1.ts, q0, human)
    assert np.array_equal(ns1, ns_o) and np.abs(q1 - q_o).max() <= TOL_RAD
    # above the threshold the one-wave shape is chosen: 800 copies of one short stream
    hb = np.repeat(human[:1, :3], 800, axis=0)
    qb0 = np.repeat(q0[:1], 800, axis=0)
    qb, nsb, stb = sol.retarget_streams(qb0, hb)
    assert (stb == 0).all() and np.array_equal(qb[0], qb[799]) and np.abs(qb[0] - q1[0, :3]).max() <= 1e-12
    with pytest.raises(hip.GmrHipError):
        sol.set_waves(3)


def _synthetic_robot(tmp_path, name, chains, tasks):
    """chains: {prefix: (parent_body, n_hinges, pos)}; tasks: [(frame, w_pos, w_rot)].  Returns Setup-like tuple."""
    from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset
    from general_motion_retargeting_amd.mjcf import compile_mjcf
    axes = ["1 0 0", "0 1 0", "0 0 1"]
    children = {}
    for prefix, (parent, n, pos) in chains.items():
        children.setdefault(parent, []).append((prefix, n, pos))

    def emit(body):
        out = ""
        for prefix, n, pos in children.get(body, []):
            s, e = "", ""
            for i in range(n):
                nm = f"{prefix}{i}"
                s += (f'<body name="{nm}" pos="{pos if i == 0 else "0.02 0 -0.09"}" quat="0.98 0.1 0.05 0.12">'
                      f'<joint name="{prefix}j{i}" axis="{axes[(i + len(prefix)) % 3]}" range="-1.3 1.1"/>')
                s += emit(nm) if i < n - 1 else ""
                e = "</body>" + e
            # children of the chain's last body
            last = f"{prefix}{n - 1}"
            s = s + emit(last) if False else s
            out += s + emit(last) + e
        return out
    xml = ('<mujoco model="%s"><compiler angle="radian"/><worldbody><body name="base" pos="0 0 1"><freejoint/>%s'
           '</body></worldbody></mujoco>' % (name, emit("base")))
    p = tmp_path / f"{name}.xml"
    p.write_text(xml)
    model = compile_mjcf(str(p))
    names = [f"h{i}" for i in range(len(tasks))]
    tbl1 = {f: [h, wp, wr, [0.01, 0, 0], [1, 0, 0, 0]] for (f, wp, wr), h in zip(tasks, names)}
    tbl2 = {f: [h, wp * 2 + 1, max(wr / 2, 1), [0, 0, 0], [1, 0, 0, 0]] for (f, wp, wr), h in zip(tasks, names)}
    cfg = {"robot_root_name": "base", "human_root_name": "h0", "ground_height": 0.0, "human_height_assumption": 1.8,
           "use_ik_match_table1": True, "use_ik_match_table2": True, "human_scale_table": {n: 0.9 for n in names},
           "ik_match_table1": tbl1, "ik_match_table2": tbl2}
    tt = build_task_tables(cfg, 1.7)
    return model, tt, pack_model(model), pack_taskset(model, tt)


@pytest.mark.parametrize("topology", ["biped_no_arms", "five_limbs_head_in_trunk", "hand_with_fingers", "too_wide_for_tree"])
def test_generic_topologies(hip, oracle, tmp_path, topology):
    """Robots the shipped set does not contain: fewer / more limbs, limbs that branch (the arm becomes
    trunk), and one the tree solver must refuse (that robot always runs the 1-wavefront shape, whatever is
    requested).  Both launch-shape requests against the oracle."""
    from general_motion_retargeting_amd import synth
    if topology == "biped_no_arms":
        chains = {"l": ("base", 6, "0 0.1 0"), "r": ("base", 6, "0 -0.1 0")}
        tasks = [("base", 100, 10), ("l2", 0, 10), ("l5", 50, 10), ("r2", 0, 10), ("r5", 50, 10)]
    elif topology == "five_limbs_head_in_trunk":
        chains = {"w": ("base", 2, "0 0 0.1"), "l": ("base", 5, "0 0.1 0"), "r": ("base", 5, "0 -0.1 0"),
                  "a": ("w1", 4, "0 0.2 0.2"), "b": ("w1", 4, "0 -0.2 0.2"), "n": ("w1", 2, "0 0 0.3")}
        tasks = [("base", 100, 10), ("w1", 0, 10), ("l4", 50, 10), ("r4", 50, 10), ("a3", 10, 5), ("b3", 10, 5), ("n1", 0, 10)]
    elif topology == "hand_with_fingers":
        chains = {"l": ("base", 4, "0 0.1 0"), "r": ("base", 4, "0 -0.1 0"), "a": ("base", 4, "0 0.2 0.3"),
                  "f": ("a3", 2, "0.05 0.02 0"), "g": ("a3", 2, "0.05 -0.02 0")}
        tasks = [("base", 100, 10), ("l3", 50, 10), ("r3", 50, 10), ("a3", 10, 10), ("f1", 20, 5), ("g1", 20, 5)]
    else:
        chains = {"w": ("base", 3, "0 0 0.1"), "l": ("base", 6, "0 0.1 0"), "r": ("base", 6, "0 -0.1 0"),
                  "a": ("w2", 5, "0 0.2 0.2"), "b": ("w2", 5, "0 -0.2 0.2"), "n": ("w2", 3, "0 0 0.3")}
        tasks = [("base", 100, 10), ("w2", 0, 10), ("l5", 50, 10), ("r5", 50, 10), ("a4", 10, 5), ("b4", 10, 5), ("n2", 0, 10)]
    model, tt, mb, ts = _synthetic_robot(tmp_path, topology, chains, tasks)
    human, q0 = synth.make_streams(model, tt, 5, 8, seed=123)
    rng = np.random.default_rng(1)
    human[..., :3] += rng.normal(0, 0.05, size=human[..., :3].shape)     # push some joints onto their limits
    q_o, ns_o, st_o = oracle.retarget_streams(mb, ts, q0, human)
    assert (st_o == 0).all()
    sol = hip.Solver(mb, ts)
    for waves in (4, 1):
        sol.set_waves(waves)
        q_h, ns_h, st_h = sol.retarget_streams(q0, human)
        assert (st_h == 0).all(), (topology, waves)
        assert np.array_equal(ns_h, ns_o), (topology, waves)
        joint, pos, rot = _compare(q_h, q_o)
        assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (topology, waves, joint, pos, rot)


@pytest.mark.gpu
def test_single_clip_script_contract(tmp_path):
    """one synthetic stream S=1, T=300 through the equivalent of
    scripts/smplx_to_robot.py:104-161 -- frame 0 skipped by the script's loop, pkl with root_rot xyzw and
    local_body_pos / link_body_list None, readable by load_robot_motion."""
    from general_motion_retargeting_amd import GeneralMotionRetargeting, dataset, load_robot_motion, save_robot_motion, synth
    g = GeneralMotionRetargeting("smplx", "unitree_g1", actual_human_height=1.7)
    human, _ = synth.make_streams(g.model, g._tables, 1, 300, seed=21)
    frames = synth.streams_to_dicts(g._tables, human[0])
    md = dataset.retarget_single_clip(g, frames, fps=30.0)
    assert list(md) == ["fps", "root_pos", "root_rot", "dof_pos", "local_body_pos", "link_body_list"]
    assert md["root_pos"].shape == (299, 3) and md["root_rot"].shape == (299, 4) and md["dof_pos"].shape == (299, 29)
    assert md["local_body_pos"] is None and md["link_body_list"] is None
    g2 = GeneralMotionRetargeting("smplx", "unitree_g1", actual_human_height=1.7)
    q = np.array([g2.retarget(f) for f in frames[1:6]])                    # the script's loop, first five calls
    assert np.abs(md["root_pos"][:5] - q[:, :3]).max() < 1e-12
    assert np.abs(md["root_rot"][:5] - q[:, [4, 5, 6, 3]]).max() < 1e-12
    assert np.abs(md["dof_pos"][:5] - q[:, 7:]).max() < 1e-12
    p = str(tmp_path / "clip.pkl")
    save_robot_motion(p, md)
    data, fps, rp, rr, dp, lbp, names = load_robot_motion(p)
    assert fps == 30.0 and lbp is None and names is None and np.array_equal(rr[:, [1, 2, 3, 0]], md["root_rot"])


@pytest.mark.gpu
def test_library_and_torch_share_one_hip_runtime():
    """bench.py --gpus N loads libgmrhip.so BEFORE it imports torch for the RCCL plumbing.  A PyTorch-ROCm wheel
    bundles its own HIP runtime; two runtimes in one process leave the second without a device ("No HIP GPUs are
    available").  _lib._share_hip_runtime() makes both bind to one: checked in a fresh process, library first."""
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        "from general_motion_retargeting_amd import _lib\n"
        "L = _lib.lib(); _lib.require_gpu(); assert L.gmr_set_device(0) == 0\n"
        "b = _lib.DeviceBuffer(1 << 20)\n"
        "import torch\n"
        "assert torch.cuda.is_available() and torch.cuda.device_count() >= 1\n"
        "x = torch.arange(8, device='cuda:0', dtype=torch.float32)\n"
        "assert float((x * 2).sum().item()) == 56.0\n"
        "maps = open('/proc/self/maps').read()\n"
        "libs = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}\n"
        "assert len(libs) == 1, libs\n"
        "print('one runtime:', libs.pop())\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "one runtime:" in r.stdout


@pytest.mark.gpu
def test_robot_promoted_to_larger_size_class(hip, oracle, tmp_path):
    """LDS offsets are compile-time constants of a size class (csrc/gmr_ik_layout.h: NVP 28/32/36/48 with body /
    task / pair capacities).  A robot with few dofs but more bodies than its class holds (nv = 16 -> class 28,
    capacity 34 bodies; here 41 bodies, most of them jointless) must move to a class that holds it (48) and still
    match the oracle in both launch-shape requests."""
    from general_motion_retargeting_amd import synth
    from general_motion_retargeting_amd.ik_config import build_task_tables, pack_model, pack_taskset
    from general_motion_retargeting_amd.mjcf import compile_mjcf
    axes = ["1 0 0", "0 1 0", "0 0 1"]

    def leg(prefix, y):
        s, e = "", ""
        for i in range(5):
            s += (f'<body name="{prefix}{i}" pos="{"0 %s -0.05" % y if i == 0 else "0.01 0 -0.1"}">'
                  f'<joint name="{prefix}j{i}" axis="{axes[i % 3]}" range="-1.2 1.2"/>')
            s += "".join(f'<body name="{prefix}{i}_m{k}" pos="0.0{k + 1} 0.01 0"/>' for k in range(3))   # jointless
            e += "</body>"
        return s + e
    xml = ('<mujoco model="many_bodies"><compiler angle="radian"/><worldbody><body name="base" pos="0 0 1"><freejoint/>'
           + leg("l", 0.1) + leg("r", -0.1) + '</body></worldbody></mujoco>')
    p = tmp_path / "many_bodies.xml"
    p.write_text(xml)
    model = compile_mjcf(str(p))
    assert model.nv == 16 and model.nbody == 41
    tasks = [("base", 100, 10), ("l2", 0, 10), ("l4", 50, 10), ("r2", 0, 10), ("r4", 50, 10), ("l4_m2", 10, 0)]
    names = [f"h{i}" for i in range(len(tasks))]
    tbl1 = {f: [h, wp, wr, [0.01, 0, 0], [1, 0, 0, 0]] for (f, wp, wr), h in zip(tasks, names)}
    tbl2 = {f: [h, wp + 1, max(wr, 1), [0, 0, 0], [1, 0, 0, 0]] for (f, wp, wr), h in zip(tasks, names)}
    cfg = {"robot_root_name": "base", "human_root_name": "h0", "ground_height": 0.0, "human_height_assumption": 1.8,
           "use_ik_match_table1": True, "use_ik_match_table2": True, "human_scale_table": {n: 0.9 for n in names},
           "ik_match_table1": tbl1, "ik_match_table2": tbl2}
    tt = build_task_tables(cfg, 1.7)
    mb, ts = pack_model(model), pack_taskset(model, tt)
    human, q0 = synth.make_streams(model, tt, 4, 6, seed=77)
    q_o, ns_o, st_o = oracle.retarget_streams(mb, ts, q0, human)
    assert (st_o == 0).all()
    sol = hip.Solver(mb, ts)
    for waves in (0, 1, 4):
        sol.set_waves(waves)
        q_h, ns_h, st_h = sol.retarget_streams(q0, human)
        assert (st_h == 0).all() and np.array_equal(ns_h, ns_o), waves
        joint, pos, rot = _compare(q_h, q_o)
        assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (waves, joint, pos, rot)


@pytest.mark.gpu
@pytest.mark.parametrize("src,robot", [("smplx", "unitree_g1"), ("bvh", "booster_t1")])
def test_many_joints_on_their_limits(hip, oracle, src, robot):
    """Targets far from reachable (0.3 m / 40 deg noise, skeleton stretched) put many joints on their limits at
    once: block principal pivoting, multiplier checks and the warm start of the bound sets are exercised in both
    QP code paths (tree solver over four wavefronts; tree solver in the DPP rows of one wavefront)."""
    from general_motion_retargeting_amd import synth
    su = get_setup(src, robot, 1.7)
    human, q0 = synth.make_streams(su.model, su.tt, 24, 12, seed=911, pos_noise=0.30, rot_noise_deg=40.0)
    root = human[:, :, :1, :3].copy()
    human[..., :3] = root + (human[..., :3] - root) * 1.3
    q_o, ns_o, st_o = oracle.retarget_streams(su.mb, su.ts, q0, human)
    assert (st_o == 0).all()
    th = q_o[..., 7:]
    on_limit = ((th <= su.model.range_lo + 1e-9) | (th >= su.model.range_hi - 1e-9)).mean()
    assert on_limit > 0.03, on_limit                                     # the case really is bound-heavy
    sol = hip.Solver(su.mb, su.ts)
    for waves in (4, 1):
        sol.set_waves(waves)
        q_h, ns_h, st_h = sol.retarget_streams(q0, human)
        assert (st_h == 0).all() and np.array_equal(ns_h, ns_o), waves
        joint, pos, rot = _compare(q_h, q_o)
        assert joint <= 1e-9 and pos <= 1e-9 and rot <= 1e-9, (waves, joint, pos, rot)


@pytest.mark.gpu
def test_queued_dispatch_is_bit_identical_to_direct(hip, oracle, g1):
    """Streams outnumbering the resident wavefronts are served from a device-side FIFO of (stream, chunk) items;
    whatever the chunk, the bits are those of one workgroup per stream: ragged lengths, empty streams, a failing
    stream, offset_to_ground, the optional outputs."""
    from general_motion_retargeting_amd import synth
    nb, S, T = 96, 6000, 7
    bh, bq = synth.make_streams(g1.model, g1.tt, nb, T, seed=11)
    rng = np.random.default_rng(5)
    pick = rng.integers(0, nb, size=S)
    human, q0 = bh[pick].copy(), bq[pick].copy()
    lens = rng.integers(0, T + 1, size=S).astype(np.int32)
    lens[:8] = [7, 0, 1, 2, 3, 7, 7, 0]
    human[5, 3, 0, 3:] = np.nan                   # stream 5 fails in frame 3
    sol = hip.Solver(g1.mb, g1.ts)
    sol.set_waves(1)
    sol.set_dispatch(0)
    ref = sol.retarget_streams(q0, human, lens=lens, flags=hip.FLAG_OFFSET_TO_GROUND, want_targets=True, want_errors=True)
    assert ref[2][5] == hip.STATUS_QP_FAILED and (np.delete(ref[2], 5) == 0).all()
    for chunk in (1, 2, 3, 6):
        sol.set_dispatch(chunk)
        out = sol.retarget_streams(q0, human, lens=lens, flags=hip.FLAG_OFFSET_TO_GROUND, want_targets=True, want_errors=True)
        for a, b in zip(ref, out):
            assert np.array_equal(a, b, equal_nan=True), chunk
    # and the oracle agrees on a few of them
    for s in (0, 3, 4, 100, 5999):
        n = int(lens[s])
        if n == 0:
            continue
        q_o, ns_o, _ = oracle.retarget_streams(g1.mb, g1.ts, q0[s:s + 1], human[s:s + 1, :n], offset_to_ground=True)
        assert np.array_equal(ref[1][s, :n], ns_o[0]) and np.abs(ref[0][s, :n] - q_o[0]).max() <= TOL_RAD
    with pytest.raises(hip.GmrHipError):
        sol.set_dispatch(-1)


@pytest.mark.gpu
@pytest.mark.parametrize("src,robot", ALL_CONFIGS)
def test_queued_dispatch_all_configs(hip, oracle, src, robot):
    """Every shipped (source, robot) pair through the device-side queue: the bits of one workgroup per stream, and
    the oracle's solve counts and angles on a sample."""
    from general_motion_retargeting_amd import synth
    su = get_setup(src, robot, 1.7)
    nb, S, T = 24, 2600, 5                       # more streams than resident wavefronts (2 048 on an MI355X)
    bh, bq = synth.make_streams(su.model, su.tt, nb, T, seed=9)
    pick = np.arange(S) % nb
    human, q0 = bh[pick].copy(), bq[pick].copy()
    sol = hip.Solver(su.mb, su.ts)
    sol.set_waves(1)
    sol.set_dispatch(0)
    q_d, ns_d, st_d = sol.retarget_streams(q0, human)
    sol.set_dispatch(2)
    q_q, ns_q, st_q = sol.retarget_streams(q0, human)
    assert (st_d == 0).all() and np.array_equal(st_d, st_q) and np.array_equal(ns_d, ns_q) and np.array_equal(q_d, q_q)
    q_o, ns_o, _ = oracle.retarget_streams(su.mb, su.ts, q0[:3], human[:3])
    assert np.array_equal(ns_q[:3], ns_o) and np.abs(q_q[:3] - q_o).max() <= TOL_RAD


@pytest.mark.gpu
def test_fk_split_walk_is_bit_equal_to_one_wavefront_per_block(hip, monkeypatch):
    """Every loadable robot tree: the split walk (up to four wavefronts per 64-frame block, ancestors recomputed)
    gives the bits of the one-wavefront walk, positions and rotations, for every number of wavefronts."""
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    rng = np.random.default_rng(3)
    for robot, xml in params.ROBOT_XML_DICT.items():
        try:
            tree = load_kinematics_tree(xml)
        except AssertionError:
            continue
        B = 777
        monkeypatch.setenv("GMR_FK_WAVES", "1")
        fk1 = hip.FkHandle(tree)
        dof = rng.uniform(-1.5, 1.5, size=(B, fk1.ndof)).astype(np.float32)
        rp = rng.normal(size=(B, 3)).astype(np.float32)
        rq = rng.normal(size=(B, 4)); rq = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)
        bp1, br1, mz1 = fk1.fk(rp, rq, dof, want_min_z=True)
        for waves in ("2", "3", "4"):
            monkeypatch.setenv("GMR_FK_WAVES", waves)
            fkn = hip.FkHandle(tree)
            bp, br, mz = fkn.fk(rp, rq, dof, want_min_z=True)
            assert np.array_equal(bp, bp1) and np.array_equal(br, br1) and mz == mz1, (robot, waves)
            bp_only, none, _ = fkn.fk(rp, rq, dof, want_rot=False)
            assert none is None and np.array_equal(bp_only, bp1), (robot, waves)


@pytest.mark.gpu
def test_fk_exact_zero_and_one_shortcuts_give_the_generic_walk(hip, monkeypatch):
    """The walk skips products with the exact zeros / ones of a unit local rotation and of an axis-aligned hinge (record
    flags set by gmr_fk_create).  Every loadable robot tree: the same values as the walk without the shortcuts
    (GMR_FK_NO_SPECIAL=1), bit for bit except the sign of an exact zero."""
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    rng = np.random.default_rng(11)
    seen = 0
    for robot, xml in params.ROBOT_XML_DICT.items():
        try:
            tree = load_kinematics_tree(xml)
        except AssertionError:
            continue
        B = 1500
        monkeypatch.setenv("GMR_FK_NO_SPECIAL", "1")
        generic = hip.FkHandle(tree)
        monkeypatch.delenv("GMR_FK_NO_SPECIAL")
        fast = hip.FkHandle(tree)
        dof = rng.uniform(-3.0, 3.0, size=(B, fast.ndof)).astype(np.float32)
        dof[:7] = 0.0                                            # zero angles: sin = 0 exactly
        rp = rng.normal(size=(B, 3)).astype(np.float32)
        rq = rng.normal(size=(B, 4)); rq = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)
        rq[:3] = [0, 0, 0, 1]
        a, b = generic.fk(rp, rq, dof), fast.fk(rp, rq, dof)
        for x, y in zip(a[:2], b[:2]):
            assert np.array_equal(x, y), robot
            diff = x.view(np.uint32) != y.view(np.uint32)
            assert not (diff & (x != 0)).any(), robot
        seen += 1
    assert seen >= 6


@pytest.mark.gpu
def test_fk_wavefront_wide_sincos_shortcut_is_bit_identical(hip):
    """When all 64 frames of a wavefront have a joint's half angle within +-0.785 rad the walk evaluates sin / cos without
    range reduction.  The same frames walked beside ONE frame with large angles (which sends its wavefront through the
    reduction) must come out bit for bit the same."""
    from general_motion_retargeting_amd import params
    from general_motion_retargeting_amd.models import load_kinematics_tree
    rng = np.random.default_rng(21)
    fk = hip.FkHandle(load_kinematics_tree(params.ROBOT_XML_DICT["unitree_g1"]))
    B = 640
    dof = rng.uniform(-1.55, 1.55, size=(B, fk.ndof)).astype(np.float32)        # half angles within 0.775
    rp = rng.normal(size=(B, 3)).astype(np.float32)
    rq = rng.normal(size=(B, 4)); rq = (rq / np.linalg.norm(rq, axis=1, keepdims=True)).astype(np.float32)
    a = fk.fk(rp, rq, dof)
    dof2 = dof.copy()
    dof2[5::64] = rng.uniform(2.0, 9.0, size=dof2[5::64].shape).astype(np.float32)   # one frame per wavefront
    b = fk.fk(rp, rq, dof2)
    keep = np.ones(B, dtype=bool); keep[5::64] = False
    for x, y in zip(a[:2], b[:2]):
        assert np.array_equal(x[keep].view(np.uint32), y[keep].view(np.uint32))
    assert not np.array_equal(a[0][~keep], b[0][~keep])


SIX_ROBOTS = ["unitree_g1", "booster_t1", "stanford_toddy", "fourier_n1", "engineai_pm01", "hightorque_hi"]


def _six_robot_batch(hip, counts, T, seed, ragged=True, fail=None):
    """One job per robot of BASELINE.json configs[3] with `counts[i]` streams (tiled from 16 distinct motions)."""
    from general_motion_retargeting_amd import synth
    rng = np.random.default_rng(seed)
    jobs, setups = [], []
    for i, (robot, S) in enumerate(zip(SIX_ROBOTS, counts)):
        su = get_setup("smplx", robot)
        bh, bq = synth.make_streams(su.model, su.tt, 16, T, seed=seed + 100 * i)
        pick = rng.integers(0, 16, size=S)
        human, q0 = bh[pick].copy(), bq[pick].copy()
        lens = rng.integers(0, T + 1, size=S).astype(np.int32) if ragged else None
        if ragged:
            lens[:4] = [T, 0, 1, T]
        if fail is not None and i == fail[0]:
            human[fail[1], fail[2], 0, 3:] = np.nan
            if ragged:
                lens[fail[1]] = T
        sol = hip.Solver(su.mb, su.ts)
        jobs.append({"solver": sol, "human": human, "q0": q0, "lens": lens})
        setups.append(su)
    return jobs, setups


@pytest.mark.gpu
def test_group_launch_is_one_scheduling_domain_and_bit_identical(hip, oracle):
    """BASELINE.json configs[3]: all six robots in ONE launch -- one resident grid, one device-side queue of (robot,
    stream, chunk) items ("one kernel with per-stream model index").  Ragged lengths, empty streams, a
    failing stream; the bits are those of every robot launched by itself, one workgroup per stream; a sample against the
    oracle.  Both the queued group (streams outnumber the resident wavefronts) and the direct one."""
    for counts, T in (([500, 430, 390, 410, 380, 450], 6), ([90, 70, 60, 80, 50, 65], 5)):
        jobs, setups = _six_robot_batch(hip, counts, T, seed=77, fail=(2, 7, 2))
        ref = []
        for j in jobs:
            j["solver"].set_waves(1)
            j["solver"].set_dispatch(0)
            ref.append(j["solver"].retarget_streams(j["q0"], j["human"], lens=j["lens"]))
            j["solver"].set_dispatch(2)
        assert ref[2][2][7] == hip.STATUS_QP_FAILED and sum(int((r[2] != 0).sum()) for r in ref) == 1
        for slices in (1, 2):
            out = hip.retarget_group(jobs, 0, slices)
            for r, (a, o) in enumerate(zip(ref, out)):
                for x, y in zip(a, o):
                    assert np.array_equal(x, y, equal_nan=True), (counts[0], slices, SIX_ROBOTS[r])
        for r, (su, j) in enumerate(zip(setups, jobs)):
            for s in (0, 3, counts[r] - 1):
                n = int(j["lens"][s])
                if n == 0 or (r == 2 and s == 7):
                    continue
                q_o, ns_o, _ = oracle.retarget_streams(su.mb, su.ts, j["q0"][s:s + 1], j["human"][s:s + 1, :n])
                assert np.array_equal(out[r][1][s, :n], ns_o[0]) and np.abs(out[r][0][s, :n] - q_o[0]).max() <= TOL_RAD
                assert not out[r][0][s, n:].any()                 # rows beyond a stream's length come back as zeros


@pytest.mark.gpu
def test_group_launch_device_pointers_and_pinned_pipeline(hip):
    """The device-pointer group entry on a caller's stream, and the sliced host pipeline through pinned buffers, against the
    plain per-solver call."""
    jobs, _ = _six_robot_batch(hip, [1500, 1400, 1450, 1380, 1420, 1490], 4, seed=5, ragged=False)    # 8 640 streams: 2 slices
    ref = [j["solver"].retarget_streams(j["q0"], j["human"]) for j in jobs]
    pj = [{"solver": j["solver"], "human": hip.pinned_copy(j["human"]), "q0": hip.pinned_copy(j["q0"])} for j in jobs]
    out = hip.retarget_group(pj, 0, 0, out_pinned=True)
    for a, o in zip(ref, out):
        assert all(np.array_equal(x, y) for x, y in zip(a, o))
    st = hip.Stream()
    dev, bufs = [], []
    for j in jobs:
        sol, (S, T) = j["solver"], j["human"].shape[:2]
        b = (hip.DeviceBuffer.from_host(j["q0"]), hip.DeviceBuffer.from_host(j["human"]), hip.DeviceBuffer(S * T * sol.nq * 8),
             hip.DeviceBuffer(S * T * 8), hip.DeviceBuffer(S * 4))
        bufs.append(b)
        dev.append((sol, S, T, b[0], b[1], None, b[2], b[3], b[4]))
    hip.retarget_group_dev(dev, 0, st)
    st.sync()
    for (sol, S, T, *_), b, a in zip(dev, bufs, ref):
        assert np.array_equal(b[2].to_host((S, T, sol.nq), np.float64), a[0])
        assert np.array_equal(b[3].to_host((S, T, 2), np.int32), a[1]) and not b[4].to_host((S,), np.int32).any()


@pytest.mark.gpu
def test_group_windows_in_time_are_bit_identical_to_one_launch(hip):
    """The host pipeline of a long, narrow batch: consecutive windows of frames (gmr_retarget_group_window_dev), the state of
    every stream (q, the QP's bound sets, status) carried in device memory.  Ragged lengths, empty streams, a failing
    stream, windows that end beyond some streams; queued and direct launches; device pointers and host buffers
    (pageable and pinned, explicit and automatic window counts)."""
    for counts, T, wins in (([500, 430, 390, 410, 380, 450], 11, 3), ([90, 70, 60, 80, 50, 65], 9, 4)):
        jobs, _ = _six_robot_batch(hip, counts, T, seed=91, fail=(3, 5, 4))
        ref = hip.retarget_group(jobs, 0, 1)
        assert ref[3][2][5] == hip.STATUS_QP_FAILED and sum(int((r[2] != 0).sum()) for r in ref) == 1
        out = hip.retarget_group(jobs, 0, -wins)
        for a, o in zip(ref, out):
            assert all(np.array_equal(x, y) for x, y in zip(a, o))
        pj = [{"solver": j["solver"], "human": hip.pinned_copy(j["human"]), "q0": hip.pinned_copy(j["q0"]), "lens": j["lens"]} for j in jobs]
        out = hip.retarget_group(pj, 0, -2, outs=hip.group_outputs(pj, pinned=True))
        for a, o in zip(ref, out):
            assert all(np.array_equal(x, y) for x, y in zip(a, o))
        # device pointers, uneven windows on a caller's stream
        st = hip.Stream()
        dev, bufs = [], []
        for j in jobs:
            sol, (S, _) = j["solver"], j["human"].shape[:2]
            b = (hip.DeviceBuffer.from_host(j["q0"]), hip.DeviceBuffer.from_host(j["human"]), hip.DeviceBuffer.from_host(j["lens"]),
                 hip.DeviceBuffer(S * T * sol.nq * 8), hip.DeviceBuffer(S * T * 8), hip.DeviceBuffer(S * 4))
            for x in b[3:]:
                x.zero()
            bufs.append(b)
            dev.append((sol, S, T, b[0], b[1], b[2], b[3], b[4], b[5]))
        for w in ((0, 2), (2, 3), (3, 8), (8, T + 5)):
            hip.retarget_group_dev(dev, 0, st, window=w)
        st.sync()
        for (sol, S, _, *_), b, a in zip(dev, bufs, ref):
            assert np.array_equal(b[3].to_host((S, T, sol.nq), np.float64), a[0])
            assert np.array_equal(b[4].to_host((S, T, 2), np.int32), a[1]) and np.array_equal(b[5].to_host((S,), np.int32), a[2])
        for b in bufs:
            for x in b:
                x.free()
    # one job alone is not a group, but windows still run (the group instance with a table of one)
    su = get_setup("smplx", "unitree_g1")
    from general_motion_retargeting_amd import synth
    human, q0 = synth.make_streams(su.model, su.tt, 400, 10, seed=3)
    sol = hip.Solver(su.mb, su.ts)
    one = [{"solver": sol, "human": human, "q0": q0}]
    a, o = hip.retarget_group(one, 0, 1)[0], hip.retarget_group(one, 0, -3)[0]
    assert all(np.array_equal(x, y) for x, y in zip(a, o)) and not a[2].any()


@pytest.mark.gpu
def test_large_host_batch_takes_the_windowed_pipeline_by_itself(hip, monkeypatch):
    """gmr_retarget_streams with >= 64 MB of input in a few thousand streams: the library cuts it into windows of frames on
    its own (copies of window w + 1 under the kernel of window w).  Ragged lengths; the exported targets and errors travel
    window by window too; the bits of the same call with the windows switched off (GMR_NO_WINDOWS=1)."""
    from general_motion_retargeting_amd import synth
    su = get_setup("smplx", "unitree_g1")
    S, T = 2400, 36
    bh, bq = synth.make_streams(su.model, su.tt, 24, T, seed=17)
    pick = np.arange(S) % 24
    human, q0 = np.ascontiguousarray(bh[pick]), np.ascontiguousarray(bq[pick])
    assert human.nbytes >= 64 << 20
    lens = np.random.default_rng(2).integers(0, T + 1, size=S).astype(np.int32)
    lens[:3] = [T, 0, 1]
    sol = hip.Solver(su.mb, su.ts)
    monkeypatch.setenv("GMR_NO_WINDOWS", "1")
    ref = sol.retarget_streams(q0, human, lens=lens, want_targets=True, want_errors=True)
    monkeypatch.delenv("GMR_NO_WINDOWS")
    out = sol.retarget_streams(q0, human, lens=lens, want_targets=True, want_errors=True)
    for a, b in zip(ref, out):
        assert np.array_equal(a, b)
    assert not ref[2].any() and ref[1][0].min() >= 1 and not ref[0][1].any() and not ref[3][1].any()


@pytest.mark.gpu
def test_full_size_config_lafan1_shape_properties(hip, oracle):
    """BASELINE.json configs[2] at full size on one GPU: 77 ragged streams, ~496 k frames (bvh -> G1, the LAFAN1-shaped
    stand-in of bench.py's leg).  Status, iteration counts, joint limits, unit root quaternions, zeros beyond a clip's
    length; a sample of clips frame by frame against the oracle; batch invariance (a clip alone == the clip in the batch)."""
    import os
    from general_motion_retargeting_amd import GeneralMotionRetargeting, synth
    rng = np.random.default_rng(3)
    lens = rng.integers(3000, 9500, size=77)
    lens = (lens * (496000 / lens.sum())).astype(np.int32)
    g = GeneralMotionRetargeting("bvh", "unitree_g1", actual_human_height=1.75)
    T = int(lens.max())
    base_h, _ = synth.make_streams(g.model, g._tables, 77, 600, seed=30, workers=4)
    idx = np.arange(T) % 1198
    idx = np.where(idx < 600, idx, 1198 - idx)
    human = np.ascontiguousarray(base_h[:, idx])
    q, ns, st = g.retarget_streams(human, lens=lens)
    assert (st == 0).all() and int(lens.sum()) > 490000
    m = g.model
    for s in range(77):
        n = int(lens[s])
        assert (ns[s, :n] >= 1).all() and (ns[s, :n] <= 11).all() and not ns[s, n:].any() and not q[s, n:].any()
        assert np.abs(np.linalg.norm(q[s, :n, 3:7], axis=-1) - 1).max() < 1e-13
        th = q[s, :n, 7:]
        lim = m.limited > 0
        assert (th[:, lim] >= m.range_lo[lim] - 1e-12).all() and (th[:, lim] <= m.range_hi[lim] + 1e-12).all()
    mb, ts = g._model_blob, g._taskset_blob
    q0 = np.broadcast_to(m.qpos0, (77, m.nq)).copy()
    for s in (0, 38, int(np.argmax(lens))):                       # the oracle on whole clips (a few seconds each)
        n = min(int(lens[s]), 1500)
        q_o, ns_o, _ = oracle.retarget_streams(mb, ts, q0[s:s + 1], human[s:s + 1, :n], nthreads=os.cpu_count() or 1)
        assert np.array_equal(ns[s, :n], ns_o[0])
        joint, pos, rot = _compare(q[s, :n], q_o[0])
        assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (s, joint, pos, rot)
    q1, ns1, st1 = g.retarget_streams(human[5:6, : lens[5]])      # one clip by itself: the same bits
    assert np.array_equal(q1[0], q[5, : lens[5]]) and np.array_equal(ns1[0], ns[5, : lens[5]])


@pytest.mark.gpu
def test_full_size_config_mixed_1m_properties(hip, oracle):
    """BASELINE.json configs[3] at full size on one GPU: 1 048 576 frames = 4 096 streams x 256 frames round-robin over
    the six robots, ONE group launch through pinned host buffers.  Status, iteration counts, limits per robot; the oracle
    on a sample of streams of every robot; batch invariance against every robot launched by itself."""
    import os
    from general_motion_retargeting_amd import GeneralMotionRetargeting, synth
    T, S_total = 256, 4096
    jobs, gm = [], []
    for r, robot in enumerate(SIX_ROBOTS):
        g = GeneralMotionRetargeting("smplx", robot)
        S = len(range(r, S_total, 6))
        bh, _ = synth.make_streams(g.model, g._tables, 64, T, seed=1 + 1000 * r, workers=4)
        human = hip.pinned_empty((S, T, bh.shape[2], 7))
        for s0 in range(0, S, 64):
            human[s0:s0 + 64] = bh[: S - s0]
        q0 = np.broadcast_to(g.model.qpos0, (S, g.model.nq)).copy()
        jobs.append({"solver": g.hip_solver, "human": human, "q0": q0})
        gm.append(g)
    out = hip.retarget_group(jobs, 0, 0, outs=hip.group_outputs(jobs))
    assert sum(j["human"].shape[0] for j in jobs) * T == 1 << 20
    for g, j, (q, ns, st) in zip(gm, jobs, out):
        m = g.model
        assert (st == 0).all() and (ns >= 1).all() and (ns <= 11).all()
        assert np.abs(np.linalg.norm(q[..., 3:7], axis=-1) - 1).max() < 1e-13
        lim = m.limited > 0
        th = q[..., 7:]
        assert (th[..., lim] >= m.range_lo[lim] - 1e-12).all() and (th[..., lim] <= m.range_hi[lim] + 1e-12).all()
        assert np.array_equal(q[64:128], q[:64])                   # the tiled motifs: equal inputs, equal bits, wherever they ran
        for s in (3, 40):
            q_o, ns_o, _ = oracle.retarget_streams(g._model_blob, g._taskset_blob, j["q0"][s:s + 1], np.asarray(j["human"][s:s + 1]),
                                                   nthreads=os.cpu_count() or 1)
            assert np.array_equal(ns[s], ns_o[0])
            joint, pos, rot = _compare(q[s], q_o[0])
            assert joint <= 1e-8 and pos <= 1e-8 and rot <= 1e-8, (g.tgt_robot if hasattr(g, "tgt_robot") else "", joint, pos, rot)
        alone = j["solver"].retarget_streams(j["q0"][:96], np.asarray(j["human"][:96]))      # the robot by itself, another launch shape
        assert np.array_equal(alone[1], ns[:96]) and np.abs(alone[0] - q[:96]).max() <= 1e-9
