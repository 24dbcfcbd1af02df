"""ik_config JSON -> TaskTables; (RobotModel, TaskTables) -> packed C structs.

The JSON schema is the reference's plugin surface and is consumed unchanged
(e.g. reference ``general_motion_retargeting/ik_configs/smplx_to_g1.json``).
What the reference does with it at construction time
(``general_motion_retargeting/motion_retarget.py:29-114``) is restated in
:func:`build_task_tables`; :func:`pack_model` / :func:`pack_taskset` flatten the
result into the PODs of ``include/gmr_types.h``.
"""
from __future__ import annotations

import dataclasses
import json
from typing import Dict, List, Optional, Tuple

import numpy as np

from .mjcf import RobotModel

# ---- must match include/gmr_types.h ---------------------------------------
GMR_MAGIC_MODEL = 0x474D524D
GMR_MAGIC_TASKSET = 0x474D5254
GMR_ABI_VERSION = 1
MAX_BODIES = 48
MAX_HINGES = 40
MAX_DOF = 46
MAX_NQ = 47
MAX_DEPTH = 20
MAX_TASKS = 16
MAX_HUMAN = 16
MAX_PAIRS = 384

MODEL_DTYPE = np.dtype(
    [
        ("magic", "<i4"), ("version", "<i4"),
        ("nbody", "<i4"), ("nhinge", "<i4"), ("nq", "<i4"), ("nv", "<i4"),
        ("parent", "<i4", (MAX_BODIES,)),
        ("depth", "<i4", (MAX_BODIES,)),
        ("body_hinge", "<i4", (MAX_BODIES,)),
        ("hinge_body", "<i4", (MAX_HINGES,)),
        ("limited", "<i4", (MAX_HINGES,)),
        ("chain", "<i4", (MAX_BODIES, MAX_DEPTH)),
        ("timestep", "<f8"),
        ("body_pos", "<f8", (MAX_BODIES, 3)),
        ("body_quat", "<f8", (MAX_BODIES, 4)),
        ("hinge_axis", "<f8", (MAX_HINGES, 3)),
        ("range_lo", "<f8", (MAX_HINGES,)),
        ("range_hi", "<f8", (MAX_HINGES,)),
        ("qpos0", "<f8", (MAX_NQ + 1,)),
    ],
    align=True,
)

TASKSET_DTYPE = np.dtype(
    [
        ("magic", "<i4"), ("version", "<i4"),
        ("nhuman", "<i4"), ("human_root", "<i4"), ("max_iter", "<i4"), ("_pad0", "<i4"),
        ("use_stage", "<i4", (2,)),
        ("ntask", "<i4", (2,)),
        ("npair", "<i4", (2,)),
        ("is_foot", "<i4", (MAX_HUMAN,)),
        ("task_body", "<i4", (2, MAX_TASKS)),
        ("task_human", "<i4", (2, MAX_TASKS)),
        ("task_col0", "<i4", (2, MAX_TASKS)),
        ("task_ncol", "<i4", (2, MAX_TASKS)),
        ("pair_task", "<i4", (2, MAX_PAIRS)),
        ("pair_dof", "<i4", (2, MAX_PAIRS)),
        ("pair_index", "<i4", (2, MAX_TASKS, MAX_DOF)),
        ("damping", "<f8"), ("lm_damping", "<f8"), ("tol", "<f8"),
        ("limit_gain", "<f8"), ("ground_offset", "<f8"),
        ("w_pos", "<f8", (2, MAX_TASKS)),
        ("w_rot", "<f8", (2, MAX_TASKS)),
        ("scale", "<f8", (MAX_HUMAN,)),
        ("pos_off", "<f8", (MAX_HUMAN, 3)),
        ("quat_off", "<f8", (MAX_HUMAN, 4)),
    ],
    align=True,
)


@dataclasses.dataclass
class StageTable:
    """One ``ik_match_table`` after dropping zero-weight entries (motion_retarget.py:80-96)."""

    frame_names: List[str]     # robot body names
    human_names: List[str]     # human body per task
    w_pos: List[float]
    w_rot: List[float]


@dataclasses.dataclass
class TaskTables:
    human_root_name: str
    robot_root_name: str
    ground_height: float
    use_stage: Tuple[bool, bool]
    scale_table: Dict[str, float]              # already multiplied by the height ratio
    stages: Tuple[StageTable, StageTable]
    pos_offsets1: Dict[str, np.ndarray]        # human body -> table-1 offset minus ground
    rot_offsets1: Dict[str, np.ndarray]        # human body -> wxyz, normalised
    human_names: List[str]                     # packed order == scale-table order
    max_iter: int = 10
    damping: float = 0.5
    lm_damping: float = 1.0
    tol: float = 0.001
    limit_gain: float = 0.95
    ground_offset: float = 0.1


def load_ik_config(path) -> dict:
    with open(path) as f:
        return json.load(f)


def build_task_tables(ik_config: dict, actual_human_height: Optional[float] = None,
                      damping: float = 0.5) -> TaskTables:
    """Restates motion_retarget.py:35-59 and :74-114 (table parsing, height ratio, offsets)."""
    if actual_human_height is not None:
        ratio = actual_human_height / ik_config["human_height_assumption"]
    else:
        ratio = 1.0
    scale_table = {k: v * ratio for k, v in ik_config["human_scale_table"].items()}
    ground = ik_config["ground_height"] * np.array([0.0, 0.0, 1.0])

    def parse(table: dict):
        st = StageTable([], [], [], [])
        pos_off: Dict[str, np.ndarray] = {}
        rot_off: Dict[str, np.ndarray] = {}
        for frame_name, entry in table.items():
            body_name, pos_weight, rot_weight, pos_offset, rot_offset = entry
            if pos_weight != 0 or rot_weight != 0:
                st.frame_names.append(frame_name)
                st.human_names.append(body_name)
                st.w_pos.append(float(pos_weight))
                st.w_rot.append(float(rot_weight))
                pos_off[body_name] = np.asarray(pos_offset, dtype=np.float64) - ground
                q = np.asarray(rot_offset, dtype=np.float64)
                rot_off[body_name] = q / np.linalg.norm(q)   # scipy R.from_quat normalises
        return st, pos_off, rot_off

    st1, pos1, rot1 = parse(ik_config["ik_match_table1"])
    st2, _pos2, _rot2 = parse(ik_config["ik_match_table2"])   # table-2 offsets are never applied
    return TaskTables(
        human_root_name=ik_config["human_root_name"],
        robot_root_name=ik_config["robot_root_name"],
        ground_height=float(ik_config["ground_height"]),
        use_stage=(bool(ik_config["use_ik_match_table1"]), bool(ik_config["use_ik_match_table2"])),
        scale_table=scale_table,
        stages=(st1, st2),
        pos_offsets1=pos1,
        rot_offsets1=rot1,
        human_names=list(scale_table.keys()),
        damping=float(damping),
    )


# --------------------------------------------------------------------------- #
# packing
# --------------------------------------------------------------------------- #
def pack_model(model: RobotModel) -> np.ndarray:
    """RobotModel -> one-element array of MODEL_DTYPE (== gmr_model_t)."""
    nb, nh = model.nbody, model.nhinge
    if nb > MAX_BODIES or nh > MAX_HINGES:
        raise ValueError(f"robot too large for the packed model: nbody={nb} nhinge={nh}")
    depth = model.depth()
    if int(depth.max()) + 1 > MAX_DEPTH:
        raise ValueError(f"kinematic tree too deep: {int(depth.max()) + 1} > {MAX_DEPTH}")
    m = np.zeros(1, dtype=MODEL_DTYPE)
    r = m[0]
    r["magic"], r["version"] = GMR_MAGIC_MODEL, GMR_ABI_VERSION
    r["nbody"], r["nhinge"], r["nq"], r["nv"] = nb, nh, model.nq, model.nv
    r["parent"][:] = -1
    r["parent"][:nb] = model.parent
    r["depth"][:nb] = depth
    r["body_hinge"][:] = -1
    r["body_hinge"][:nb] = model.body_hinge
    r["hinge_body"][:nh] = model.hinge_body
    r["limited"][:nh] = model.limited
    r["chain"][:] = -1
    for b in range(nb):
        path = []
        c = b
        while c >= 0:
            path.append(c)
            c = int(model.parent[c])
        path.reverse()
        r["chain"][b, : len(path)] = path
    r["timestep"] = model.timestep
    r["body_pos"][:nb] = model.body_pos
    r["body_quat"][:, 0] = 1.0
    r["body_quat"][:nb] = model.body_quat
    r["hinge_axis"][:, 2] = 1.0
    r["hinge_axis"][:nh] = model.hinge_axis
    r["range_lo"][:nh] = model.range_lo
    r["range_hi"][:nh] = model.range_hi
    r["qpos0"][: model.nq] = model.qpos0
    return m


def task_dofs(model: RobotModel, body: int) -> List[int]:
    """Velocity-space dofs that move ``body``: the 6 base dofs + the hinges on root -> body."""
    dofs = list(range(6))
    path = []
    c = body
    while c >= 0:
        path.append(c)
        c = int(model.parent[c])
    for c in reversed(path):
        h = int(model.body_hinge[c])
        if h >= 0:
            dofs.append(6 + h)
    return dofs


def pack_taskset(model: RobotModel, tt: TaskTables) -> np.ndarray:
    """(RobotModel, TaskTables) -> one-element array of TASKSET_DTYPE (== gmr_taskset_t).

    Raises ``KeyError`` for a task frame that is not a body of the robot (mink raises
    ``InvalidFrame`` at FrameTask evaluation) and for table bodies missing from the scale table
    (the reference raises ``KeyError`` inside ``update_targets``, motion_retarget.py:129,241).
    """
    names = tt.human_names
    if len(names) > MAX_HUMAN:
        raise ValueError(f"too many human bodies: {len(names)} > {MAX_HUMAN}")
    if tt.human_root_name not in names:
        raise KeyError(tt.human_root_name)
    t = np.zeros(1, dtype=TASKSET_DTYPE)
    r = t[0]
    r["magic"], r["version"] = GMR_MAGIC_TASKSET, GMR_ABI_VERSION
    r["nhuman"] = len(names)
    r["human_root"] = names.index(tt.human_root_name)
    r["max_iter"] = tt.max_iter
    r["use_stage"][:] = [int(tt.use_stage[0]), int(tt.use_stage[1])]
    r["damping"], r["lm_damping"], r["tol"] = tt.damping, tt.lm_damping, tt.tol
    r["limit_gain"], r["ground_offset"] = tt.limit_gain, tt.ground_offset
    r["quat_off"][:, 0] = 1.0
    for i, n in enumerate(names):
        r["is_foot"][i] = int(("Foot" in n) or ("foot" in n))
        r["scale"][i] = tt.scale_table[n]
        # bodies of the scale table without a table-1 entry make the reference raise KeyError at
        # retarget time (motion_retarget.py:241); the shim checks that before any launch.
        if n in tt.pos_offsets1:
            r["pos_off"][i] = tt.pos_offsets1[n]
            r["quat_off"][i] = tt.rot_offsets1[n]
    r["pair_index"][:] = -1
    for s in range(2):
        st = tt.stages[s]
        k = len(st.frame_names)
        if k > MAX_TASKS:
            raise ValueError(f"too many tasks in stage {s + 1}: {k} > {MAX_TASKS}")
        r["ntask"][s] = k
        p = 0
        for i in range(k):
            body = model.body_id(st.frame_names[i])
            if st.human_names[i] not in names:
                raise KeyError(st.human_names[i])
            r["task_body"][s, i] = body
            r["task_human"][s, i] = names.index(st.human_names[i])
            r["w_pos"][s, i] = st.w_pos[i]
            r["w_rot"][s, i] = st.w_rot[i]
            dofs = task_dofs(model, body)
            if st.w_pos[i] == 0.0:
                # A task without a position cost (11 of 14 in the first table of every shipped config) has an exactly
                # zero weighted Jacobian column for each base translation: body-frame column [R^T e_i; 0], so its
                # orientation rows are zero and its position rows are multiplied by the zero cost.  Such (task, dof)
                # pairs add +-0 to H and c: they are not listed (a fifth of the pairs and a third of the H terms of
                # that table).
                dofs = [d for d in dofs if d >= 3]
            if p + len(dofs) > MAX_PAIRS:
                raise ValueError("too many (task, dof) pairs")
            r["task_col0"][s, i] = p
            r["task_ncol"][s, i] = len(dofs)
            for d in dofs:
                r["pair_task"][s, p] = i
                r["pair_dof"][s, p] = d
                r["pair_index"][s, i, d] = p
                p += 1
        r["npair"][s] = p
    return t
