"""``KinematicsModel`` with the reference's surface (``kinematics_model.py:69-278``), the batched
float32 forward kinematics running as ONE HIP kernel (``gmr_fk_batch``) instead of ~2.6k ATen
launches.

Inputs may be NumPy arrays (host path: copied through the C-ABI) or torch tensors.  CUDA/ROCm
torch tensors are consumed in place (``data_ptr()``, torch's current stream) and the outputs are
torch tensors on the same device, so the reference's dataset scripts run unchanged
(``scripts/smplx_to_robot_dataset.py:93-126``); torch is plumbing for device memory here, never
the compute path.  There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from .models import load_kinematics_tree


class KinematicsModel:
    def __init__(self, file_path, device="cuda:0"):
        self._device = device
        self._file_path = str(file_path)
        self._tree = load_kinematics_tree(self._file_path)     # AssertionError / NotImplementedError like the reference
        self._body_names = [str(x) for x in self._tree["body_names"]]
        self._parent_indices = np.asarray(self._tree["parent"], dtype=np.int64)
        self._dof_idx = [int(x) for x in self._tree["dof_idx"]]
        self._num_dof = int((np.asarray(self._tree["dof_dim"]) > 0).sum())
        self._handle = None

    # ---- reference properties ---------------------------------------------------------------
    @property
    def body_names(self):
        return self._body_names

    @property
    def num_dof(self):
        return self._num_dof

    @property
    def num_joint(self):
        return len(self._body_names)

    @property
    def joint_dof_idx(self):
        return list(self._dof_idx)

    @property
    def parent_indices(self):
        return self._parent_indices

    def get_parent_idx(self, idx):
        return self._parent_indices[idx]

    def get_body_idx(self, body_name):
        return self._body_names.index(body_name)

    def get_dof_limits(self):
        return self._tree["lower"], self._tree["upper"]

    @property
    def hip_handle(self) -> _lib.FkHandle:
        if self._handle is None:
            self._handle = _lib.FkHandle(self._tree)          # raises without GPU/library
        return self._handle

    # ---- joint-angle <-> rotation helpers (reference :172-211; host NumPy, float32 like torch eager) --------
    # No shipped script calls them; they are kept so that code written against the reference class keeps working.
    # torch tensors in -> torch tensors out (same device), NumPy in -> NumPy out.
    @staticmethod
    def _np(x):
        if type(x).__module__.startswith("torch"):
            return x.detach().cpu().numpy(), x
        return np.asarray(x), None

    @staticmethod
    def _like(a, proto):
        if proto is None:
            return a
        import torch
        return torch.from_numpy(np.ascontiguousarray(a)).to(device=proto.device)

    def dof_to_rot(self, dof):
        """dof [..., num_dof] -> joint rotations xyzw [..., num_joint - 1, 4] (identity for jointless bodies)."""
        d, proto = self._np(dof)
        d = d.astype(np.float32, copy=False)
        nb = self.num_joint
        rot = np.zeros(d.shape[:-1] + (nb - 1, 4), dtype=np.float32)
        axis = np.asarray(self._tree["axis"], dtype=np.float64)
        for j in range(1, nb):
            k = self._dof_idx[j]
            if k < 0:
                rot[..., j - 1, 3] = 1.0
                continue
            # axis_angle_to_quat (torch_utils.py:353-359): float32 half angle, float64 axis -> products and the
            # normalisation in float64, rounded to float32 on assignment (Joint.dof_to_rot, :32)
            theta = (d[..., k] / np.float32(2.0)).astype(np.float32)
            a = axis[j] / max(np.linalg.norm(axis[j]), 1e-9)
            q = np.concatenate([a * np.sin(theta)[..., None].astype(np.float64), np.cos(theta)[..., None].astype(np.float64)], axis=-1)
            q = q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-9)
            rot[..., j - 1, :] = q.astype(np.float32)
        return self._like(rot, proto)

    def rot_to_dof(self, rot):
        """joint rotations xyzw [..., num_joint - 1, 4] -> dof [..., num_dof], clamped to the joint limits."""
        r, proto = self._np(rot)
        r = r.astype(np.float32, copy=False)
        nb = self.num_joint
        dof = np.zeros(r.shape[:-2] + (self._num_dof,), dtype=np.float32)
        axis = np.asarray(self._tree["axis"], dtype=np.float64)
        for j in range(1, nb):
            k = self._dof_idx[j]
            if k < 0:
                continue
            q = r[..., j - 1, :]
            q = np.where(q[..., 3:] < 0, -q, q)                              # quat_pos (torch_utils.py:312-317)
            length = np.linalg.norm(q[..., :3], axis=-1).astype(np.float32)
            angle = (np.float32(2.0) * np.arctan2(length, q[..., 3])).astype(np.float32)
            with np.errstate(invalid="ignore", divide="ignore"):
                ax = q[..., :3] / length[..., None]
            mask = length > 1e-5
            angle = np.where(mask, angle, np.float32(0.0))
            ax = np.where(mask[..., None], ax, np.array([0.0, 0.0, 1.0], dtype=np.float32))
            dot = np.sum(ax.astype(np.float64) * axis[j], axis=-1)           # raw (un-normalised) joint axis, :47
            dof[..., k] = np.where(dot < 0, -angle, angle)
        lo, hi = self._tree["lower"], self._tree["upper"]
        dof = np.clip(dof, np.asarray(lo, dtype=np.float32), np.asarray(hi, dtype=np.float32))
        return self._like(dof, proto)

    @staticmethod
    def _quat_mul_xyzw(a, b):
        """torch_utils.quat_mul (:117-138): the 8-multiplication form, float32."""
        x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
        x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
        ww = (z1 + x1) * (x2 + y2)
        yy = (w1 - y1) * (w2 + z2)
        zz = (w1 + y1) * (w2 - z2)
        xx = ww + yy + zz
        qq = np.float32(0.5) * (xx + (z1 - x1) * (x2 - y2))
        w = qq - ww + (z1 - y1) * (y2 - z2)
        x = qq - xx + (x1 + w1) * (x2 + w2)
        y = qq - yy + (w1 - x1) * (y2 + z2)
        z = qq - zz + (z1 + y1) * (w2 - x2)
        return np.stack([x, y, z, w], axis=-1)

    def convert_local_rot_to_global(self, local_rot):
        """local rotations xyzw [..., num_joint, 4] (row 0 = the root's) -> global rotations, chained down the tree."""
        r, proto = self._np(local_rot)
        r = r.astype(np.float32, copy=False)
        g = np.zeros_like(r)
        g[..., 0, :] = r[..., 0, :]
        for j in range(1, self.num_joint):
            g[..., j, :] = self._quat_mul_xyzw(g[..., int(self._parent_indices[j]), :], r[..., j, :])
        return self._like(g, proto)

    def _shaped_handle(self, fitted_shape) -> _lib.FkHandle:
        """FK handle whose local translations are scaled per body (``local_translation[j] * fitted_shape[j]``, :224)."""
        fs, _ = self._np(fitted_shape)
        fs = np.asarray(fs, dtype=np.float32)
        lt = np.asarray(self._tree["local_translation"], dtype=np.float32)
        tree = dict(self._tree)
        tree["local_translation"] = (lt * fs.reshape(self.num_joint, -1)).astype(np.float32)
        return _lib.FkHandle(tree)

    # ---- H9 ------------------------------------------------------------------------------------
    def forward_kinematics(self, root_pos, root_rot, dof_pos, fitted_shape=None, return_min_z=False):
        """root_pos [..., 3], root_rot xyzw [..., 4], dof_pos [..., num_dof] ->
        (body_pos [..., nb, 3], body_rot xyzw [..., nb, 4]) float32.  ``return_min_z`` additionally
        returns min over everything of body_pos z (the dataset scripts' height-adjust reduction).
        ``fitted_shape`` [num_joint, 3 or 1] scales every body's local translation (reference :224): the same
        kernel on a handle built for the scaled tree."""
        if fitted_shape is not None:
            saved = self._handle
            self._handle = self._shaped_handle(fitted_shape)
            try:
                return self.forward_kinematics(root_pos, root_rot, dof_pos, None, return_min_z)
            finally:
                self._handle.close()
                self._handle = saved
        is_torch = type(root_pos).__module__.startswith("torch")
        if is_torch:
            return self._fk_torch(root_pos, root_rot, dof_pos, return_min_z)
        rp = np.asarray(root_pos, dtype=np.float32)
        lead = rp.shape[:-1]
        B = int(np.prod(lead)) if lead else 1
        bp, br, mz = self.hip_handle.fk(rp.reshape(B, 3), np.asarray(root_rot, dtype=np.float32).reshape(B, 4),
                                        np.asarray(dof_pos, dtype=np.float32).reshape(B, self._num_dof),
                                        want_rot=True, want_min_z=return_min_z)
        nb = self.num_joint
        out = (bp.reshape(lead + (nb, 3)), br.reshape(lead + (nb, 4)))
        return out + (mz,) if return_min_z else out

    def _fk_torch(self, root_pos, root_rot, dof_pos, return_min_z):
        import torch
        lead = tuple(root_pos.shape[:-1])
        nb = self.num_joint
        if not root_pos.is_cuda:
            res = self.forward_kinematics(root_pos.detach().numpy(), root_rot.detach().numpy(),
                                          dof_pos.detach().numpy(), return_min_z=return_min_z)
            out = (torch.from_numpy(res[0]), torch.from_numpy(res[1]))
            return out + (res[2],) if return_min_z else out
        dev = root_pos.device
        B = int(np.prod(lead)) if lead else 1
        rp = root_pos.detach().to(torch.float32).reshape(B, 3).contiguous()
        rr = root_rot.detach().to(device=dev, dtype=torch.float32).reshape(B, 4).contiguous()
        dp = dof_pos.detach().to(device=dev, dtype=torch.float32).reshape(B, self._num_dof).contiguous()
        bp = torch.empty((B, nb, 3), dtype=torch.float32, device=dev)
        br = torch.empty((B, nb, 4), dtype=torch.float32, device=dev)
        mz = torch.empty((1,), dtype=torch.float32, device=dev) if return_min_z else None
        import ctypes as C
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            self.hip_handle.fk_dev(B, C.c_void_p(rp.data_ptr()), C.c_void_p(rr.data_ptr()), C.c_void_p(dp.data_ptr()),
                                   C.c_void_p(bp.data_ptr()), C.c_void_p(br.data_ptr()),
                                   C.c_void_p(mz.data_ptr()) if mz is not None else None, stream)
        out = (bp.reshape(lead + (nb, 3)), br.reshape(lead + (nb, 4)))
        return out + (float(mz.item()),) if return_min_z else out
