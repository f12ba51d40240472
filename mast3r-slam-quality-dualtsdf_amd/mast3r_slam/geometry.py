"""Mirror of the pure-tensor helpers of mast3r_slam/geometry.py that the hot path uses outside the
fused kernels (constrain_points_to_ray :37-42, backproject :107-115, get_pixel_coords :118-123,
act_Sim3 without Jacobian :45-52).  Index/pixel-grid plumbing on device tensors; the Jacobian forms
live inside the GN kernels (csrc/gn.hip, csrc/tracker.hip)."""
import torch


def get_pixel_coords(b, img_size, device, dtype):
    h, w = img_size
    u, v = torch.meshgrid(torch.arange(w, device=device), torch.arange(h, device=device), indexing="xy")
    return torch.stack((u, v), dim=-1).unsqueeze(0).repeat(b, 1, 1, 1).to(dtype=dtype)


def backproject(p, z, K):
    tmp1 = (p[..., 0] - K[0, 2]) / K[0, 0]
    tmp2 = (p[..., 1] - K[1, 2]) / K[1, 1]
    dP_dz = torch.empty(p.shape[:-1] + (3, 1), device=z.device, dtype=K.dtype)
    dP_dz[..., 0, 0] = tmp1
    dP_dz[..., 1, 0] = tmp2
    dP_dz[..., 2, 0] = 1.0
    return torch.squeeze(z[..., None, :] * dP_dz, dim=-1)


def constrain_points_to_ray(img_size, Xs, K):
    uv = get_pixel_coords(Xs.shape[0], img_size, device=Xs.device, dtype=Xs.dtype).view(*Xs.shape[:-1], 2)
    return backproject(uv, Xs[..., 2:3], K)


def act_Sim3(X, pC, jacobian=False):
    if jacobian:
        raise NotImplementedError("Jacobians are computed inside the fused GN kernels")
    return X.act(pC)
