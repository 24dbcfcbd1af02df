"""N4: the streaming-ingest shape adapter.  The reference's side of this boundary is
``NatNetClient.get_frame`` (optitrack_vendor/NatNetClient.py:2368-2383: ``frame[name] = [rb.pos,
np.roll(rb.rot, 1)]`` for ids in the map, unknown ids skipped) feeding ``retarget(frame)``
(scripts/optitrack_to_robot.py:37-46).  The vendor client itself needs a socket and is out of scope, so the
expected values here are that contract restated, not fixtures produced by the reference (parity unpinned
for the network part; the packed layout is checked against the dict route of the shim)."""
import numpy as np
import pytest

from general_motion_retargeting_amd import GeneralMotionRetargeting
from general_motion_retargeting_amd.utils.optitrack import (FBX_SKELETON_NAMES, RigidBodyPacker, StreamingRetargeter,
                                                            frame_from_rigid_bodies, rigid_body_id_map)


def _skeleton(rng, n=51, offset=0):
    ids = np.arange(1, n + 1) + offset
    pos = rng.normal(size=(n, 3))
    rot = rng.normal(size=(n, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    return ids, pos, rot


def test_id_map_and_dict_adapter():
    m = rigid_body_id_map()
    assert len(FBX_SKELETON_NAMES) == 51 and sorted(m) == list(range(1, 52))
    assert m[1] == "Hips" and m[9] == "LeftHand" and m[21] == "RightToeBase" and m[22] == "LeftHandThumb1"
    assert m[36] == "LeftHandPinky3" and m[37] == "RightHandThumb1" and m[51] == "RightHandPinky3"
    assert rigid_body_id_map(100)[101] == "Hips"
    rng = np.random.default_rng(0)
    ids, pos, rot = _skeleton(rng)
    unknown = []
    frame = frame_from_rigid_bodies(list(zip(ids, pos, rot)) + [(777, pos[0], rot[0])], unknown=unknown)
    assert unknown == [777] and set(frame) == set(FBX_SKELETON_NAMES)
    assert np.array_equal(frame["Head"][0], pos[4])
    assert np.array_equal(frame["Head"][1], rot[4][[3, 0, 1, 2]])          # xyzw -> wxyz


def test_packer_matches_dict_route():
    g = GeneralMotionRetargeting("fbx", "unitree_g1", actual_human_height=1.6)    # the only fbx config shipped
    rng = np.random.default_rng(1)
    ids, pos, rot = _skeleton(rng)
    perm = rng.permutation(51)                                             # arrival order is arbitrary
    packed = RigidBodyPacker(g).pack(ids[perm], pos[perm], rot[perm])
    via_dict = g.pack_frame(g.to_numpy(frame_from_rigid_bodies(zip(ids, pos, rot))))
    assert packed.shape == (len(g.human_body_names), 7) and np.array_equal(packed, via_dict)
    # unknown ids are skipped; a body the IK config needs raises KeyError like retarget(dict) does
    extra = RigidBodyPacker(g).pack(np.append(ids, 999), np.vstack([pos, pos[:1]]), np.vstack([rot, rot[:1]]))
    assert np.array_equal(extra, packed)
    root_id = 1 + FBX_SKELETON_NAMES.index(g.human_root_name)
    keep = ids != root_id
    with pytest.raises(KeyError):
        RigidBodyPacker(g).pack(ids[keep], pos[keep], rot[keep])
    with pytest.raises(KeyError):
        g.pack_frame(g.to_numpy(frame_from_rigid_bodies(zip(ids[keep], pos[keep], rot[keep]))))
    # shifted ids
    shifted = RigidBodyPacker(g, rigid_body_id_map(7)).pack(ids + 7, pos, rot)
    assert np.array_equal(shifted, packed)


@pytest.mark.gpu
def test_streaming_equals_dict_retarget():
    from general_motion_retargeting_amd import synth
    g1 = GeneralMotionRetargeting("fbx", "unitree_g1", actual_human_height=1.6)
    g2 = GeneralMotionRetargeting("fbx", "unitree_g1", actual_human_height=1.6)
    human, _ = synth.make_streams(g1.model, g1._tables, 1, 12, seed=5)
    names = g1.human_body_names
    id_of = {n: i for i, n in rigid_body_id_map().items()}
    st = StreamingRetargeter(g2)
    for t in range(human.shape[1]):
        fr = human[0, t]
        frame = {n: [fr[i, :3].copy(), fr[i, 3:].copy()] for i, n in enumerate(names)}
        q_ref = g1.retarget(frame)
        ids = np.array([id_of[n] for n in names])
        q = st.step(ids, fr[:, :3], fr[:, [4, 5, 6, 3]])                    # client delivers xyzw
        assert np.array_equal(q, q_ref)
    assert st.frame_number == human.shape[1] - 1
