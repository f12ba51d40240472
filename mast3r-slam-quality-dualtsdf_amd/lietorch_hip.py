"""Minimal ``lietorch.Sim3`` surface for the hot path, backed by libmslam_hip.so.

The reference depends on the external package lietorch (princeton-vl; unpinned in pyproject.toml:15,
commit 0fa9ce8f cited at gn_kernels.cu:344) which is NOT vendored in the reference tree.  The op set
below is exactly what the hot path calls: ``Identity``, construction from an
(...,8) tensor ``[t, q(xyzw), s]``, ``.data``, ``.act``, ``.inv``, ``*``, ``.retr``, ``Sim3.exp``,
``.matrix``, indexing, ``to/cpu/clone``.  Forward only (the SLAM loop runs under inference mode).
Maths: the reference's own in-tree restatement, gn_kernels.cu:177-413.
"""
import torch

import mslam_hip as _m


class Sim3:
    embedded_dim = 8
    manifold_dim = 7

    def __init__(self, data):
        if isinstance(data, Sim3):
            data = data.data
        self.data = data

    # -- construction ------------------------------------------------------------------------
    @staticmethod
    def Identity(*batch, device="cpu", dtype=torch.float32):
        d = torch.zeros(*batch, 8, device=device, dtype=dtype)
        d[..., 6] = 1.0
        d[..., 7] = 1.0
        return Sim3(d)

    @staticmethod
    def exp(xi):
        flat = xi.reshape(-1, 7).contiguous().float()
        out = torch.empty((flat.shape[0], 8), dtype=torch.float32, device=flat.device)
        rc = _m.lib().mslam_sim3_op(2, _m.ptr(flat), 0, _m.ptr(out), flat.shape[0], 0, 0, _m.stream_ptr())
        _m.check(rc, "Sim3.exp")
        return Sim3(out.reshape(*xi.shape[:-1], 8))

    # -- plumbing ----------------------------------------------------------------------------
    @property
    def shape(self):
        return self.data.shape[:-1]

    @property
    def device(self):
        return self.data.device

    def __getitem__(self, idx):
        return Sim3(self.data[idx])

    def __setitem__(self, idx, other):
        self.data[idx] = other.data if isinstance(other, Sim3) else other

    def to(self, *args, **kwargs):
        return Sim3(self.data.to(*args, **kwargs))

    def cpu(self):
        return Sim3(self.data.cpu())

    def clone(self):
        return Sim3(self.data.clone())

    def view(self, *shape):
        return Sim3(self.data.view(*shape, 8))

    # -- group ops ---------------------------------------------------------------------------
    def _binary(self, op, a, b):
        a2 = a.reshape(-1, a.shape[-1]).contiguous()
        b2 = b.reshape(-1, 8).contiguous()
        n = max(a2.shape[0], b2.shape[0])
        if a2.shape[0] not in (1, n) or b2.shape[0] not in (1, n):
            raise RuntimeError(f"Sim3: cannot broadcast batch sizes {a2.shape[0]} and {b2.shape[0]}")
        out = torch.empty((n, 8), dtype=torch.float32, device=b2.device)
        rc = _m.lib().mslam_sim3_op(op, _m.ptr(a2), _m.ptr(b2), _m.ptr(out), n,
                                    int(a2.shape[0] == 1 and n > 1), int(b2.shape[0] == 1 and n > 1), _m.stream_ptr())
        _m.check(rc, "Sim3 op")
        lead = a.shape[:-1] if a2.shape[0] == n else b.shape[:-1]
        return Sim3(out.reshape(*lead, 8))

    def inv(self):
        flat = self.data.reshape(-1, 8).contiguous()
        out = torch.empty_like(flat)
        rc = _m.lib().mslam_sim3_op(0, _m.ptr(flat), 0, _m.ptr(out), flat.shape[0], 0, 0, _m.stream_ptr())
        _m.check(rc, "Sim3.inv")
        return Sim3(out.reshape(self.data.shape))

    def __mul__(self, other):
        return self._binary(1, self.data, other.data)

    def retr(self, xi):
        """exp(xi) * self  (left retraction; lietorch's .retr, tracker.py:247)."""
        return self._binary(3, xi.float(), self.data)

    def act(self, X):
        """s*R*X + t.  self.data (...,8) broadcast against X (...,3): either one pose for all
        points (tracker.py:150) or one pose per leading batch entry (tsdf_refine.py:864)."""
        poses = self.data.reshape(-1, 8).contiguous()
        Xc = X.contiguous().float()
        pts = Xc.reshape(-1, 3)
        Y = torch.empty_like(pts)
        npose = poses.shape[0]
        if npose == 1:
            rc = _m.lib().mslam_sim3_act(_m.ptr(poses), _m.ptr(pts), _m.ptr(Y), 1, pts.shape[0], 1, _m.stream_ptr())
        else:
            if pts.shape[0] % npose != 0:
                raise RuntimeError(f"Sim3.act: {npose} poses do not divide {pts.shape[0]} points")
            rc = _m.lib().mslam_sim3_act(_m.ptr(poses), _m.ptr(pts), _m.ptr(Y), npose, pts.shape[0] // npose, 0,
                                         _m.stream_ptr())
        _m.check(rc, "Sim3.act")
        return Y.reshape(Xc.shape)

    def matrix(self):
        """4x4 [sR t; 0 1] (tsdf_refine.py:860-861).  Tiny: composed from the pose data with torch
        indexing on the device tensor."""
        d = self.data.reshape(-1, 8)
        x, y, z, w = d[:, 3], d[:, 4], d[:, 5], d[:, 6]
        s = d[:, 7]
        R = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
        M = torch.zeros((d.shape[0], 4, 4), dtype=d.dtype, device=d.device)
        M[:, :3, :3] = s[:, None, None] * R
        M[:, :3, 3] = d[:, :3]
        M[:, 3, 3] = 1.0
        return M.reshape(*self.data.shape[:-1], 4, 4)
