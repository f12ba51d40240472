"""ORACLE (test infrastructure): numpy restatement of the reference's frame-tracking GN
  FrameTracker.opt_pose_ray_dist_sim3 / opt_pose_calib_sim3 / solve   /root/reference/mast3r_slam/tracker.py:208-318
  act_Sim3 / point_to_ray_dist / project_calib                        /root/reference/mast3r_slam/geometry.py:17-104
  huber / check_convergence                                           /root/reference/mast3r_slam/nonlinear_optimizer.py:5-33
written in the reference's own form (explicit (n,4,7) Jacobians, A^T A, Cholesky solve, retr), float32
tensors / python-float scalars like the torch code.  The single-step normal equations are pinned by
tests/golden/tracker_formulae.npz (built from the reference's geometry.py); lietorch's retr is the
oracle's Sim3 restatement (lietorch itself is absent: parity unpinned for the group ops)."""
import math

import numpy as np

import oracle


def _skew(x):
    o = np.zeros(x.shape[:-1], x.dtype)
    return np.stack([o, -x[..., 2], x[..., 1], x[..., 2], o, -x[..., 0], -x[..., 1], x[..., 0], o], -1).reshape(*x.shape[:-1], 3, 3)


def act_sim3_jac(T, pC):
    pW = oracle.sim3_act(T, pC)
    n = pW.shape[0]
    J = np.concatenate([np.broadcast_to(np.eye(3, dtype=np.float32), (n, 3, 3)), -_skew(pW), pW[..., None]], -1)
    return pW, J.astype(np.float32)


def ray_dist(X, jac=False):
    d = np.linalg.norm(X, axis=-1, keepdims=True).astype(np.float32)
    r = X / d
    rd = np.concatenate([r, d], -1)
    if not jac:
        return rd
    I = np.eye(3, dtype=np.float32)
    dr = (1.0 / d)[..., None] * (I - (1.0 / d ** 2)[..., None] * (X[..., :, None] @ X[..., None, :]))
    return rd, np.concatenate([dr, r[:, None, :]], -2).astype(np.float32)


def project_calib(P, K, img_size, border, z_eps):
    z = P[:, 2]
    u = K[0, 0] * P[:, 0] / z + K[0, 2]
    v = K[1, 1] * P[:, 1] / z + K[1, 2]
    valid = (u > border) & (u < img_size[1] - 1 - border) & (v > border) & (v < img_size[0] - 1 - border) & (z > z_eps)
    with np.errstate(invalid="ignore", divide="ignore"):
        logz = np.where(z > z_eps, np.log(z), 0.0)
    pz = np.stack([u, v, logz], -1).astype(np.float32)
    zi = 1.0 / z
    J = np.zeros((P.shape[0], 3, 3), np.float32)
    J[:, 0, 0] = K[0, 0] * zi; J[:, 1, 1] = K[1, 1] * zi
    J[:, 0, 2] = -K[0, 0] * P[:, 0] * zi * zi; J[:, 1, 2] = -K[1, 1] * P[:, 1] * zi * zi
    J[:, 2, 2] = zi
    return pz, J, valid[:, None]


def huber(r, k):
    a = np.abs(r)
    with np.errstate(divide="ignore"):
        return np.where(a < k, 1.0, k / a).astype(np.float32)


def solve(sqrt_info, r, J, k):
    robust = sqrt_info * np.sqrt(huber(sqrt_info * r, k))
    A = (robust[..., None] * J).reshape(-1, 7).astype(np.float32)
    b = (robust * r).reshape(-1, 1).astype(np.float32)
    H = A.T @ A
    g = -A.T @ b
    cost = 0.5 * float((b.T @ b)[0, 0])
    L = np.linalg.cholesky(H.astype(np.float64))   # raises LinAlgError like torch.linalg.cholesky
    tau = np.linalg.solve(L.T, np.linalg.solve(L, g.astype(np.float64)))
    return tau.reshape(1, 7).astype(np.float32), cost


def check_convergence(rel_thr, dn_thr, old_cost, new_cost, delta):
    with np.errstate(invalid="ignore"):
        rel_dec = math.fabs((old_cost - new_cost) / old_cost) if old_cost != 0 else float("nan")
    return rel_dec < rel_thr or float(np.linalg.norm(delta)) < dn_thr


def track(use_calib, Xf_g, Xk, T_WCf, T_WCk, Qk, valid, cfg, K=None, img_size=None):
    """Xf_g = frame points already gathered by idx_f2k (tracker.py:206).  Returns (T_WCf, T_CkCf, iters)."""
    sa, sb = (cfg["sigma_pixel"], cfg["sigma_depth"]) if use_calib else (cfg["sigma_ray"], cfg["sigma_dist"])
    na = 2 if use_calib else 3
    sq = (valid.reshape(-1, 1) * np.sqrt(Qk.reshape(-1, 1))).astype(np.float32)
    sqrt_info = np.concatenate([np.repeat(np.float32(1 / sa) * sq, na, 1), np.float32(1 / sb) * sq], 1)
    T = oracle.sim3_rel(T_WCk, T_WCf)[0]
    if use_calib:
        h, w = img_size
        uu, vv = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32), indexing="xy")
        valid_meas = Xk[:, 2:3] > cfg["depth_eps"]
        with np.errstate(invalid="ignore", divide="ignore"):
            meas = np.concatenate([uu.reshape(-1, 1), vv.reshape(-1, 1), np.log(Xk[:, 2:3])], 1).astype(np.float32)
        meas[~np.repeat(valid_meas, 3, 1)] = 0.0
    else:
        rd_k = ray_dist(Xk)
    old = float("inf")
    it = 0
    for it in range(1, cfg["max_iters"] + 1):
        P, dP = act_sim3_jac(T, Xf_g)
        if use_calib:
            pz, dpz, vproj = project_calib(P, K, img_size, cfg["pixel_border"], cfg["depth_eps"])
            si = (vproj & valid_meas) * sqrt_info
            r = meas - pz
            J = -(dpz @ dP)
        else:
            rd_f, drd = ray_dist(P, jac=True)
            si = sqrt_info
            r = rd_k - rd_f
            J = -(drd @ dP)
        tau, cost = solve(si.astype(np.float32), r.astype(np.float32), J.astype(np.float32), cfg["huber"])
        T = oracle.sim3_retr(tau, T)[0]
        if check_convergence(cfg["rel_error"], cfg["delta_norm"], old, cost, tau):
            break
        old = cost
    # T_WCf = T_WCk * T_CkCf
    Tw = oracle.sim3_retr(np.zeros(7, np.float32), T_WCk)[0]  # identity retraction = copy
    import scipy.spatial.transform as sst
    def mat(t):
        M = np.eye(4); M[:3, :3] = t[7] * sst.Rotation.from_quat(t[3:7].astype(np.float64)).as_matrix(); M[:3, 3] = t[:3]; return M
    M = mat(T_WCk) @ mat(T)
    s = np.cbrt(np.linalg.det(M[:3, :3]))
    q = sst.Rotation.from_matrix(M[:3, :3] / s).as_quat()
    if q[3] < 0 and False:
        q = -q
    T_WCf_new = np.concatenate([M[:3, 3], q, [s]]).astype(np.float32)
    return T_WCf_new, T, it
