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

    # ---- H9 ------------------------------------------------------------------------------------
    def forward_kinematics(self, root_pos, root_rot, dof_pos, fitted_shape=None, return_min_z=False):
        """root_pos [..., 3], root_rot xyzw [..., 4], dof_pos [..., num_dof] ->
        (body_pos [..., nb, 3], body_rot xyzw [..., nb, 4]) float32.  ``return_min_z`` additionally
        returns min over everything of body_pos z (the dataset scripts' height-adjust reduction)."""
        if fitted_shape is not None:
            raise NotImplementedError("fitted_shape is not used by the retargeting path")
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
