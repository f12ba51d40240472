"""ORACLE (test infrastructure): restatement of the local dense-block TSDF of the reference's refiner
  TSDFRefiner._build_tsdf_robust      /root/reference/mast3r_slam/tsdf_refine.py:837-940
  TSDFRefiner._extract_surface_safe   tsdf_refine.py:942-1021
  TSDFRefiner._sample_tsdf_trilinear  tsdf_refine.py:1023-1064
with the reference's dtype chain: float32 tensors, python-float (double) scalars obtained via .item().
Pinned by tests/golden/tsdf_refine.npz (produced by RUNNING the reference's tsdf_refine.py on CPU).

linspace: the reference calls torch.linspace on `self.device`.  On CPU torch's vectorised kernel differs
from the per-element formula `start + step*i | end - step*(n-1-i)` by one ulp in ~1.5 % of samples; on a
GPU torch evaluates exactly that per-element formula.  `linspace="torch"` reproduces the CPU fixture,
`linspace="scalar"` the device behaviour (what the HIP kernel implements)."""
import numpy as np
import torch

F = np.float32


def linspace_f32(t0, t1, n, mode):
    if mode == "torch":
        return torch.linspace(t0, t1, n).numpy()
    s, e = F(t0), F(t1)
    if n == 1:
        return np.array([s], F)
    step = F((e - s) / F(n - 1))
    half = n // 2
    return np.array([F(s + step * F(i)) if i < half else F(e - step * F(n - i - 1)) for i in range(n)], F)


def grid_dims(xyz_min, xyz_max, voxel_size=0.02, max_grid_dim=64):
    roi = (xyz_max.astype(F) - xyz_min.astype(F)).astype(F)
    d = np.minimum(np.ceil(roi / F(voxel_size)).astype(np.int64), max_grid_dim)
    return int(d[0]), int(d[1]), int(d[2]), roi


def build_tsdf(X_world, C, origin, xyz_min, xyz_max, voxel_size=0.02, trunc=0.08, max_grid_dim=64,
               min_confidence=0.2, linspace="scalar"):
    """X_world (n,3) f32 = T_WC.act(X_canon); origin (3,) f32 = camera centre.  -> tsdf, weights (nz,ny,nx) f32."""
    X_world, C = X_world.astype(F), C.astype(F)
    xyz_min, xyz_max, origin = xyz_min.astype(F), xyz_max.astype(F), origin.astype(F)
    nx, ny, nz, roi = grid_dims(xyz_min, xyz_max, voxel_size, max_grid_dim)
    tsdf = np.ones((nz, ny, nx), F)
    weights = np.zeros((nz, ny, nx), F)
    valid = (np.isfinite(X_world).all(1) & (X_world >= xyz_min).all(1) & (X_world <= xyz_max).all(1)
             & (C > F(min_confidence)))
    if valid.sum() < 5:
        return tsdf, weights
    actual = (roi / np.array([nx, ny, nz], F)).astype(F)
    for i in np.nonzero(valid)[0]:
        conf = float(C[i])
        ray = (X_world[i] - origin).astype(F)
        ray_length = float(np.sqrt(F(F(ray[0] * ray[0]) + F(ray[1] * ray[1])) + F(ray[2] * ray[2])))
        ray_dir = (ray / F(ray_length + 1e-8)).astype(F)
        if ray_length < 0.05:
            continue
        t_start = max(0.05, ray_length - trunc * 2.0)
        t_end = ray_length + trunc * 2.0
        n = min(32, int((t_end - t_start) / voxel_size) + 1)
        for t in linspace_f32(t_start, t_end, n, linspace):
            sample = (origin + (ray_dir * t).astype(F)).astype(F)
            if (sample < xyz_min).any() or (sample > xyz_max).any():
                continue
            gp = ((sample - xyz_min).astype(F) / actual).astype(F)
            gx = int(min(max(gp[0], F(0)), F(nx - 1)))
            gy = int(min(max(gp[1], F(0)), F(ny - 1)))
            gz = int(min(max(gp[2], F(0)), F(nz - 1)))
            sdf = (ray_length - float(t)) / trunc
            sdf = max(-1.0, min(1.0, sdf))
            weight = conf * max(0.0, 1.0 - abs(sdf))
            old_w = float(weights[gz, gy, gx])
            new_w = old_w + weight
            if new_w > 1e-6:
                tsdf[gz, gy, gx] = F((float(tsdf[gz, gy, gx]) * old_w + sdf * weight) / new_w)
                weights[gz, gy, gx] = F(new_w)
    return tsdf, weights


def sample_trilinear(vol, x, y, z):
    nz, ny, nx = vol.shape
    x, y, z = min(max(x, F(0)), F(nx - 1)), min(max(y, F(0)), F(ny - 1)), min(max(z, F(0)), F(nz - 1))
    x0, y0, z0 = int(np.floor(x)), int(np.floor(y)), int(np.floor(z))
    x1, y1, z1 = min(x0 + 1, nx - 1), min(y0 + 1, ny - 1), min(z0 + 1, nz - 1)
    xd, yd, zd = float(F(x - F(x0))), float(F(y - F(y0))), float(F(z - F(z0)))
    c = lambda a, b, d: float(vol[a, b, d])
    c00 = c(z0, y0, x0) * (1 - xd) + c(z0, y0, x1) * xd
    c01 = c(z0, y1, x0) * (1 - xd) + c(z0, y1, x1) * xd
    c10 = c(z1, y0, x0) * (1 - xd) + c(z1, y0, x1) * xd
    c11 = c(z1, y1, x0) * (1 - xd) + c(z1, y1, x1) * xd
    c0 = c00 * (1 - yd) + c01 * yd
    c1 = c10 * (1 - yd) + c11 * yd
    return c0 * (1 - zd) + c1 * zd


def extract_surface(tsdf, xyz_min, xyz_max, mask, X_original, order, n_samples=64, max_displacement=0.015,
                    linspace="scalar"):
    """order = positions (into the masked pixel list) to process, i.e. randperm(n_mask)[:100].
    -> X_refined (n,3) f32, hits (n_mask,) bool"""
    X_original = X_original.astype(F)
    xyz_min, xyz_max = xyz_min.astype(F), xyz_max.astype(F)
    X_ref = X_original.copy()
    pix = np.nonzero(mask)[0]
    hits = np.zeros(len(pix), bool)
    nz, ny, nx = tsdf.shape
    actual = ((xyz_max - xyz_min).astype(F) / np.array([nx, ny, nz], F)).astype(F)
    for k in order:
        p = X_original[pix[k]]
        depth = float(p[2])
        if depth < 0.05:
            continue
        ts = linspace_f32(max(0.05, depth - 0.1), depth + 0.1, n_samples, linspace)
        prev_sdf, prev_t = None, None
        for t in ts:
            sp = (p * F(t / F(depth))).astype(F) if depth > 0.01 else p    # 0-dim f32 tensor / python float
            if (sp < xyz_min).any() or (sp > xyz_max).any():
                prev_sdf = None
                continue
            gp = ((sp - xyz_min).astype(F) / actual).astype(F)
            sdf = sample_trilinear(tsdf, gp[0], gp[1], gp[2])
            if prev_sdf is not None and prev_sdf * sdf < 0:
                alpha = abs(prev_sdf) / (abs(prev_sdf) + abs(sdf) + 1e-8)
                t_surf = F(prev_t + F(F(alpha) * F(t - prev_t)))
                surf = (p * F(t_surf / F(depth))).astype(F) if depth > 0.01 else p
                d = (surf - p).astype(F)
                disp = float(np.sqrt(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2])))
                if disp <= max_displacement:
                    X_ref[pix[k]] = surf
                    hits[k] = True
                break
            prev_sdf, prev_t = sdf, t
    return X_ref, hits
