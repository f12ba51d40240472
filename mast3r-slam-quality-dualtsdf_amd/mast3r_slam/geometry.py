"""Mirror of mast3r_slam/geometry.py (every public function, same argument meaning and return shapes).
The tracker and the backend do NOT go through these: their residuals and Jacobians are fused into
csrc/tracker.hip / csrc/gn.hip.  The tensor forms below serve callers outside the fused kernels (keyframe
initialisation with calibration, evaluation, debugging) and the parity fixture tests/golden/geometry.npz."""
import torch


def skew_sym(x):
    """geometry.py:5-9: [x]_x, (...,3) -> (...,3,3)."""
    a, b, c = x[..., 0], x[..., 1], x[..., 2]
    z = torch.zeros_like(a)
    rows = (torch.stack((z, -c, b), -1), torch.stack((c, z, -a), -1), torch.stack((-b, a, z), -1))
    return torch.stack(rows, -2)


def point_to_dist(X):
    """geometry.py:12-14."""
    return torch.linalg.norm(X, dim=-1, keepdim=True)


def point_to_ray_dist(X, jacobian=False):
    """geometry.py:17-34: (unit ray, distance), optionally d(ray, dist)/dX (...,4,3)."""
    d = point_to_dist(X)
    inv = 1.0 / d
    ray = inv * X
    rd = torch.cat((ray, d), dim=-1)
    if not jacobian:
        return rd
    eye = torch.eye(3, device=X.device, dtype=X.dtype).expand(*X.shape[:-1], 3, 3)
    outer = X.unsqueeze(-1) @ X.unsqueeze(-2)
    d_ray = inv.unsqueeze(-1) * (eye - (inv ** 2).unsqueeze(-1) * outer)
    return rd, torch.cat((d_ray, ray.unsqueeze(-2)), dim=-2)


def decompose_K(K):
    """geometry.py:55-60."""
    return K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]


def project_calib(P, K, img_size, jacobian=False, border=0, z_eps=0.0):
    """geometry.py:63-104: pinhole pixel + log depth, validity (strictly inside the border, z > z_eps),
    optionally d(u, v, log z)/dP (...,3,3)."""
    x, y, z = P[..., 0:1], P[..., 1:2], P[..., 2:3]
    hom = (K.expand(*P.shape[:-1], 3, 3) @ P.unsqueeze(-1)).squeeze(-1)
    uv = (hom / hom[..., 2:3])[..., :2]
    u, v = uv[..., 0:1], uv[..., 1:2]
    valid_z = z > z_eps
    valid = (u > border) & (u < img_size[1] - 1 - border) & (v > border) & (v < img_size[0] - 1 - border) & valid_z
    logz = torch.log(z)
    logz[~valid_z] = 0.0
    pz = torch.cat((uv, logz), dim=-1)
    if not jacobian:
        return pz, valid
    fx, fy, _, _ = decompose_K(K)
    zi = 1.0 / z[..., 0]
    J = torch.zeros(*P.shape[:-1], 3, 3, device=P.device, dtype=P.dtype)
    J[..., 0, 0] = fx
    J[..., 1, 1] = fy
    J[..., 0, 2] = -fx * x[..., 0] * zi
    J[..., 1, 2] = -fy * y[..., 0] * zi
    J *= zi[..., None, None]
    J[..., 2, 2] = zi
    return pz, J, valid


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
    """geometry.py:45-52: world point and, optionally, its (...,3,7) derivative w.r.t. the left Sim3 perturbation
    [I, -[pW]_x, pW]."""
    pW = X.act(pC)
    if not jacobian:
        return pW
    eye = torch.eye(3, device=pW.device).expand(*pW.shape[:-1], 3, 3)
    return pW, torch.cat((eye, -skew_sym(pW), pW.unsqueeze(-1)), dim=-1)
