"""-m gpu: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs."""
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
