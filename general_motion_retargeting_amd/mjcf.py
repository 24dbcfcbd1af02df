"""MJCF-subset compiler: robot XML -> RobotModel (plain numpy arrays).

Replaces, for the retargeting hot path only, what the reference obtains from
``mj.MjModel.from_xml_path(self.xml_file)`` (reference
``general_motion_retargeting/motion_retarget.py:27``): the kinematic tree, the
joint axes / ranges / limited flags, ``qpos0`` and ``opt.timestep``.  Geoms,
meshes, inertials, actuators, sensors and keyframes are ignored (they do not
enter ``mj_kinematics`` / ``mj_jacBody`` / ``mj_integratePos``).

Semantics restated from MuJoCo's MJCF rules:

* bodies are numbered in document (DFS pre-) order, the free-joint body is 0
  (MuJoCo's world body is not represented; ``parent == -1`` for the root);
* ``pos`` defaults to ``0 0 0``; ``quat`` (wxyz) defaults to ``1 0 0 0`` and is
  normalised; hinge ``axis`` defaults to ``0 0 1`` and is normalised;
* ``<default>`` classes are resolved (``class`` attribute, else the nearest
  ancestor ``childclass``, else the top-level default);
* ``limited``: explicit ``true``/``false``; ``auto`` (the default) means
  ``range[0] < range[1]`` when ``compiler/autolimits`` is true (MuJoCo >= 3
  default);
* ``<include file=...>`` is spliced in place, path relative to the including
  file (engineai_pm01 relies on this);
* ``compiler/angle`` defaults to ``degree``; ranges are converted to radians.

Supported subset: one top-level body carrying one free joint, every other body
carrying zero or one hinge joint.  Anything else raises ``NotImplementedError``
so that an unsupported robot fails loudly instead of silently mis-compiling.
"""
from __future__ import annotations

import dataclasses
import math
import os
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional

import numpy as np

__all__ = ["RobotModel", "compile_mjcf", "parse_kinematics_tree"]


@dataclasses.dataclass
class RobotModel:
    """Kinematic subset of a compiled MJCF (all float64, MuJoCo conventions)."""

    name: str
    body_names: List[str]
    parent: np.ndarray        # int32 [nbody], -1 for the root body
    body_pos: np.ndarray      # f64 [nbody,3]  (root: its qpos0 translation)
    body_quat: np.ndarray     # f64 [nbody,4]  wxyz, normalised
    body_hinge: np.ndarray    # int32 [nbody]  hinge index on this body or -1
    joint_names: List[str]    # hinge joint names, qpos order
    hinge_body: np.ndarray    # int32 [nhinge]
    hinge_axis: np.ndarray    # f64 [nhinge,3] body-local, normalised
    range_lo: np.ndarray      # f64 [nhinge]  radians
    range_hi: np.ndarray      # f64 [nhinge]
    limited: np.ndarray       # int32 [nhinge]
    qpos0: np.ndarray         # f64 [7+nhinge]  xyz, wxyz, hinge refs (0)
    timestep: float

    @property
    def nbody(self) -> int:
        return len(self.body_names)

    @property
    def nhinge(self) -> int:
        return int(self.hinge_body.shape[0])

    @property
    def nq(self) -> int:
        return 7 + self.nhinge

    @property
    def nv(self) -> int:
        return 6 + self.nhinge

    def body_id(self, name: str) -> int:
        try:
            return self.body_names.index(name)
        except ValueError:
            raise KeyError(f"body '{name}' not found in robot model '{self.name}'") from None

    def depth(self) -> np.ndarray:
        d = np.zeros(self.nbody, dtype=np.int32)
        for b in range(1, self.nbody):
            d[b] = d[self.parent[b]] + 1
        return d

    # -- (de)serialisation to a flat dict of arrays (npz-friendly, no pickle) --
    def to_arrays(self) -> Dict[str, np.ndarray]:
        return {
            "name": np.array(self.name),
            "body_names": np.array(self.body_names),
            "parent": self.parent.astype(np.int32),
            "body_pos": self.body_pos,
            "body_quat": self.body_quat,
            "body_hinge": self.body_hinge.astype(np.int32),
            "joint_names": np.array(self.joint_names),
            "hinge_body": self.hinge_body.astype(np.int32),
            "hinge_axis": self.hinge_axis,
            "range_lo": self.range_lo,
            "range_hi": self.range_hi,
            "limited": self.limited.astype(np.int32),
            "qpos0": self.qpos0,
            "timestep": np.array(self.timestep, dtype=np.float64),
        }

    @classmethod
    def from_arrays(cls, a) -> "RobotModel":
        return cls(
            name=str(a["name"]),
            body_names=[str(x) for x in a["body_names"]],
            parent=np.asarray(a["parent"], dtype=np.int32),
            body_pos=np.asarray(a["body_pos"], dtype=np.float64),
            body_quat=np.asarray(a["body_quat"], dtype=np.float64),
            body_hinge=np.asarray(a["body_hinge"], dtype=np.int32),
            joint_names=[str(x) for x in a["joint_names"]],
            hinge_body=np.asarray(a["hinge_body"], dtype=np.int32),
            hinge_axis=np.asarray(a["hinge_axis"], dtype=np.float64),
            range_lo=np.asarray(a["range_lo"], dtype=np.float64),
            range_hi=np.asarray(a["range_hi"], dtype=np.float64),
            limited=np.asarray(a["limited"], dtype=np.int32),
            qpos0=np.asarray(a["qpos0"], dtype=np.float64),
            timestep=float(a["timestep"]),
        )


# --------------------------------------------------------------------------- #
# XML loading with <include> splicing
# --------------------------------------------------------------------------- #
def _load_with_includes(path: str, _depth: int = 0) -> ET.Element:
    if _depth > 16:
        raise ValueError(f"<include> nesting too deep at {path}")
    root = ET.parse(path).getroot()
    base = os.path.dirname(os.path.abspath(path))

    def splice(elem: ET.Element) -> None:
        i = 0
        while i < len(elem):
            child = elem[i]
            if child.tag == "include":
                inc_path = os.path.join(base, child.attrib["file"])
                inc_root = _load_with_includes(inc_path, _depth + 1)
                elem.remove(child)
                for k, sub in enumerate(list(inc_root)):
                    elem.insert(i + k, sub)
                i += len(inc_root)
            else:
                splice(child)
                i += 1

    splice(root)
    return root


def _floats(s: str, n: Optional[int] = None) -> np.ndarray:
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and v.shape[0] != n:
        raise ValueError(f"expected {n} numbers, got '{s}'")
    return v


class _Defaults:
    """Resolved <default> classes for the <joint> element only."""

    def __init__(self, root: ET.Element):
        self.joint: Dict[str, Dict[str, str]] = {"main": {}}
        for top in root.findall("default"):
            self._walk(top, top.attrib.get("class", "main"), {})

    def _walk(self, node: ET.Element, cls: str, inherited: Dict[str, str]) -> None:
        attrs = dict(inherited)
        j = node.find("joint")
        if j is not None:
            attrs.update(j.attrib)
        # several top-level <default> blocks may all describe "main"
        merged = dict(self.joint.get(cls, {}))
        merged.update(attrs)
        self.joint[cls] = merged
        for sub in node.findall("default"):
            self._walk(sub, sub.attrib["class"], merged)

    def joint_attrs(self, elem: ET.Element, childclass: Optional[str]) -> Dict[str, str]:
        cls = elem.attrib.get("class", childclass or "main")
        if cls not in self.joint:
            raise KeyError(f"unknown default class '{cls}'")
        out = dict(self.joint[cls])
        out.update(elem.attrib)
        return out


def compile_mjcf(xml_path: str) -> RobotModel:
    """Compile the kinematic subset of an MJCF file (see module docstring)."""
    xml_path = str(xml_path)
    root = _load_with_includes(xml_path)
    if root.tag != "mujoco":
        raise ValueError(f"{xml_path}: root element is <{root.tag}>, expected <mujoco>")

    angle_unit = "degree"
    autolimits = True
    for comp in root.findall("compiler"):
        angle_unit = comp.attrib.get("angle", angle_unit)
        if "autolimits" in comp.attrib:
            autolimits = comp.attrib["autolimits"] == "true"
        if comp.attrib.get("coordinate", "local") != "local":
            raise NotImplementedError("compiler/coordinate=global is not supported")
    if angle_unit not in ("degree", "radian"):
        raise ValueError(f"invalid compiler/angle '{angle_unit}'")
    to_rad = math.pi / 180.0 if angle_unit == "degree" else 1.0

    timestep = 0.002
    for opt in root.findall("option"):
        if "timestep" in opt.attrib:
            timestep = float(opt.attrib["timestep"])

    defaults = _Defaults(root)

    top_bodies: List[ET.Element] = []
    for wb in root.findall("worldbody"):
        top_bodies.extend(wb.findall("body"))
    if len(top_bodies) != 1:
        raise NotImplementedError(
            f"{xml_path}: expected exactly one top-level <body>, found {len(top_bodies)}")

    body_names: List[str] = []
    parent: List[int] = []
    body_pos: List[np.ndarray] = []
    body_quat: List[np.ndarray] = []
    body_hinge: List[int] = []
    joint_names: List[str] = []
    hinge_body: List[int] = []
    hinge_axis: List[np.ndarray] = []
    range_lo: List[float] = []
    range_hi: List[float] = []
    limited: List[int] = []

    def add_body(node: ET.Element, parent_id: int, childclass: Optional[str]) -> None:
        for bad in ("euler", "axisangle", "xyaxes", "zaxis"):
            if bad in node.attrib:
                raise NotImplementedError(f"body orientation attribute '{bad}' is not supported")
        bid = len(body_names)
        body_names.append(node.attrib.get("name", f"body{bid}"))
        parent.append(parent_id)
        pos = _floats(node.attrib.get("pos", "0 0 0"), 3)
        quat = _floats(node.attrib.get("quat", "1 0 0 0"), 4)
        nrm = float(np.linalg.norm(quat))
        if nrm < 1e-15:
            raise ValueError(f"body '{body_names[-1]}': zero quaternion")
        quat = quat / nrm
        body_pos.append(pos)
        body_quat.append(quat)
        cc = node.attrib.get("childclass", childclass)

        joints = [(j, False) for j in node.findall("joint")] + [(j, True) for j in node.findall("freejoint")]
        n_hinge_here = 0
        has_free = False
        body_hinge.append(-1)
        for j, is_freejoint in joints:
            attrs = dict(j.attrib) if is_freejoint else defaults.joint_attrs(j, cc)
            jtype = "free" if is_freejoint else attrs.get("type", "hinge")
            if jtype == "free":
                if bid != 0:
                    raise NotImplementedError("free joint on a non-root body")
                has_free = True
                continue
            if jtype != "hinge":
                raise NotImplementedError(f"joint type '{jtype}' is not supported")
            if bid == 0:
                raise NotImplementedError("hinge joint on the floating-base body")
            n_hinge_here += 1
            if n_hinge_here > 1:
                raise NotImplementedError(
                    f"body '{body_names[bid]}': more than one hinge per body is not supported")
            jpos = _floats(attrs.get("pos", "0 0 0"), 3)
            if np.any(jpos != 0.0):
                raise NotImplementedError(
                    f"joint '{attrs.get('name')}': non-zero joint pos is not supported")
            if float(attrs.get("ref", "0")) != 0.0:
                raise NotImplementedError("joint ref != 0 is not supported")
            axis = _floats(attrs.get("axis", "0 0 1"), 3)
            an = float(np.linalg.norm(axis))
            if an < 1e-15:
                raise ValueError(f"joint '{attrs.get('name')}': zero axis")
            axis = axis / an
            rng = _floats(attrs.get("range", "0 0"), 2) * to_rad
            lim_attr = attrs.get("limited", "auto")
            if lim_attr == "true":
                lim = 1
            elif lim_attr == "false":
                lim = 0
            elif lim_attr == "auto":
                if not autolimits and (rng[0] != 0.0 or rng[1] != 0.0):
                    raise ValueError(
                        f"joint '{attrs.get('name')}': range given, limited=auto and autolimits=false")
                lim = 1 if (autolimits and rng[0] < rng[1]) else 0
            else:
                raise ValueError(f"invalid limited='{lim_attr}'")
            hid = len(hinge_body)
            body_hinge[bid] = hid
            joint_names.append(attrs.get("name", f"joint{hid}"))
            hinge_body.append(bid)
            hinge_axis.append(axis)
            range_lo.append(float(rng[0]))
            range_hi.append(float(rng[1]))
            limited.append(lim)
        if bid == 0 and not has_free:
            raise NotImplementedError("the root body must carry a free joint")
        for child in node.findall("body"):
            add_body(child, bid, cc)

    add_body(top_bodies[0], -1, None)

    nh = len(hinge_body)
    qpos0 = np.zeros(7 + nh, dtype=np.float64)
    qpos0[0:3] = body_pos[0]
    qpos0[3:7] = body_quat[0]
    return RobotModel(
        name=root.attrib.get("model", os.path.basename(xml_path)),
        body_names=body_names,
        parent=np.array(parent, dtype=np.int32),
        body_pos=np.array(body_pos, dtype=np.float64).reshape(-1, 3),
        body_quat=np.array(body_quat, dtype=np.float64).reshape(-1, 4),
        body_hinge=np.array(body_hinge, dtype=np.int32),
        joint_names=joint_names,
        hinge_body=np.array(hinge_body, dtype=np.int32),
        hinge_axis=np.array(hinge_axis, dtype=np.float64).reshape(-1, 3),
        range_lo=np.array(range_lo, dtype=np.float64),
        range_hi=np.array(range_hi, dtype=np.float64),
        limited=np.array(limited, dtype=np.int32),
        qpos0=qpos0,
        timestep=timestep,
    )


# --------------------------------------------------------------------------- #
# The reference KinematicsModel's own (different) XML reading, H8
# --------------------------------------------------------------------------- #
def parse_kinematics_tree(xml_path: str) -> Dict[str, np.ndarray]:
    """Tree arrays with the semantics of the reference's post-hoc FK parser.

    Mirrors ``KinematicsModel._parse_xml`` (reference
    ``general_motion_retargeting/kinematics_model.py:101-164``), which is NOT
    MuJoCo's compiler: it reads only the top file (no ``<include>``: a file whose
    ``<worldbody>`` is missing raises ``AssertionError`` exactly like the
    reference, ``:104-105``), takes the first ``<worldbody><body>`` subtree, keeps
    body ``quat`` un-normalised (reordered wxyz -> xyzw, ``:119-123``), requires
    explicit ``axis`` and ``range`` attributes on the element itself (defaults
    classes are not consulted, ``:133,:136``), gives the root body ``dof_dim`` 0
    (``:125-126``) and stores translation/rotation/limits as float32 and the hinge
    axis as float64 (``:93-98,:133-134``).
    """
    tree = ET.parse(str(xml_path))
    doc = tree.getroot()
    world = doc.find("worldbody")
    assert world is not None, "worldbody not found"
    body_root = world.find("body")
    assert body_root is not None, "body not found"
    compiler = doc.find("compiler")
    rot_unit = compiler.attrib.get("angle", "degree")
    assert rot_unit in ["degree", "radian"], f"Invalid rotation unit: {rot_unit}"

    names: List[str] = []
    parents: List[int] = []
    trans: List[np.ndarray] = []
    rots: List[np.ndarray] = []
    dof_dim: List[int] = []
    axes: List[np.ndarray] = []
    lo: List[float] = []
    hi: List[float] = []

    def add(node: ET.Element, parent_index: int) -> None:
        idx = len(names)
        pos = _floats(node.attrib.get("pos", "0 0 0"))
        q = _floats(node.attrib.get("quat", "1 0 0 0"))
        rots.append(np.array([q[1], q[2], q[3], q[0]]))
        trans.append(pos)
        names.append(node.attrib.get("name"))
        parents.append(parent_index)
        if idx == 0:
            dof_dim.append(0)
            axes.append(np.zeros(3))
        else:
            js = node.findall("joint")
            if len(js) == 0:
                dof_dim.append(0)
                axes.append(np.zeros(3))
            elif len(js) == 1:
                dof_dim.append(1)
                axes.append(_floats(js[0].attrib.get("axis")))
                r = _floats(js[0].attrib.get("range"))
                lo.append(r[0])
                hi.append(r[1])
            elif len(js) == 3:
                raise NotImplementedError(
                    "3-joint (exp-map) bodies are parsed by the reference but used by none of its robots")
            else:
                raise ValueError(f"Invalid number of joints: {len(js)} of body: {names[-1]}")
        for child in node.findall("body"):
            add(child, idx)

    add(body_root, -1)
    lo_a = np.array(lo, dtype=np.float32)
    hi_a = np.array(hi, dtype=np.float32)
    if rot_unit == "degree":
        lo_a = np.deg2rad(lo_a).astype(np.float32)
        hi_a = np.deg2rad(hi_a).astype(np.float32)
    dof_dim_a = np.array(dof_dim, dtype=np.int32)
    dof_idx = np.full(len(names), -1, dtype=np.int32)
    k = 0
    for i, d in enumerate(dof_dim):
        if d > 0:
            dof_idx[i] = k
            k += d
    return {
        "body_names": np.array(names),
        "parent": np.array(parents, dtype=np.int32),
        "local_translation": np.array(trans, dtype=np.float64).astype(np.float32),
        "local_rotation": np.array(rots, dtype=np.float64).astype(np.float32),  # xyzw, un-normalised
        "dof_dim": dof_dim_a,
        "dof_idx": dof_idx,
        "axis": np.array(axes, dtype=np.float64),
        "lower": lo_a,
        "upper": hi_a,
    }
