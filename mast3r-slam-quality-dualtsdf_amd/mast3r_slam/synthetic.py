"""Seeded synthetic inputs of the shapes the hot path sees (BASELINE.json configs 3/5):
a procedural box room seen by a pinhole camera (K = [[400,0,256],[0,400,192],[0,0,1]] at 512x384),
pointmaps = back-projected depth (+noise), confidences, 24-d fp16-friendly descriptors, Sim3 poses.
numpy only (runs on the host); callers move arrays to the device.
"""
import numpy as np

ROOM_HALF = np.array([3.0, 2.0, 1.5])  # 6 x 4 x 3 m box, camera inside


def intrinsics(h=384, w=512):
    s = w / 512.0
    return np.array([[400.0 * s, 0, w / 2.0], [0, 400.0 * s, h / 2.0], [0, 0, 1.0]], np.float32)


def pixel_rays(h, w, K):
    u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64), indexing="xy")
    d = np.stack(((u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1], np.ones_like(u)), -1)
    return d  # (h,w,3), z = 1


def quat_from_rotvec(r):
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.array([0, 0, 0, 1.0])
    a = r / th
    return np.concatenate((a * np.sin(th / 2), [np.cos(th / 2)]))


def quat_rotate(q, X):
    qv, qw = q[:3], q[3]
    uv = 2.0 * np.cross(np.broadcast_to(qv, X.shape), X)
    return X + qw * uv + np.cross(np.broadcast_to(qv, X.shape), uv)


def sim3_act(T, X):
    """T = [t(3), q(xyzw), s] ; X (...,3) -> s*R*X + t"""
    return T[7] * quat_rotate(T[3:7], X) + T[:3]


def sim3_inv(T):
    qi = np.array([-T[3], -T[4], -T[5], T[6]])
    si = 1.0 / T[7]
    ti = -si * quat_rotate(qi, T[:3][None])[0]
    return np.concatenate((ti, qi, [si]))


def camera_pose(k, n_frames=1000):
    """Smooth trajectory inside the room: world-from-camera Sim3 (scale 1) for frame k."""
    a = 2 * np.pi * k / max(n_frames, 1) * 3.0
    t = np.array([1.2 * np.cos(a), 0.6 * np.sin(0.7 * a), 0.4 * np.sin(a)])
    rot = np.array([0.15 * np.sin(0.9 * a), a * 0.35, 0.1 * np.cos(1.3 * a)])
    return np.concatenate((t, quat_from_rotvec(rot), [1.0]))


def ray_box_depth(origin, dirs_world):
    """Distance along each ray (||dir|| arbitrary) to the inside of the axis-aligned room box."""
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (ROOM_HALF - origin) / dirs_world
        t2 = (-ROOM_HALF - origin) / dirs_world
    t = np.where(dirs_world > 0, t1, t2)
    t = np.where(np.abs(dirs_world) < 1e-12, np.inf, t)
    return t.min(-1)


def render_pointmap(T_wc, h=384, w=512, K=None):
    """Camera-frame pointmap (h,w,3) f64 of the room seen from Sim3 pose T_wc."""
    K = intrinsics(h, w) if K is None else K
    d_cam = pixel_rays(h, w, K)
    d_world = quat_rotate(T_wc[3:7], d_cam.reshape(-1, 3)).reshape(h, w, 3)
    z = ray_box_depth(T_wc[:3], d_world)  # since d_cam has z=1 and scale 1, t == depth
    return d_cam * z[..., None]


def descriptor_field(P_world, fdim=24, seed=7):
    """Smooth pseudo-random unit descriptors of 3-D position: sin(W p + phase), L2-normalised."""
    rng = np.random.default_rng(seed)
    Wm = rng.normal(0, 6.0, (3, fdim))
    ph = rng.uniform(0, 2 * np.pi, fdim)
    D = np.sin(P_world @ Wm + ph)
    return D / np.linalg.norm(D, axis=-1, keepdims=True)


def make_pair(k_i, k_j, h=384, w=512, seed=0, noise=0.002, n_frames=1000):
    """One keyframe pair in MASt3R output convention: X11 = points of view i in frame i,
    X21 = points of view j expressed in frame i (same pixel grid as view j), + C, D, Q maps."""
    rng = np.random.default_rng(seed + 1000 * k_i + k_j)
    K = intrinsics(h, w)
    Ti, Tj = camera_pose(k_i, n_frames), camera_pose(k_j, n_frames)
    Xi_i = render_pointmap(Ti, h, w, K)
    Xj_j = render_pointmap(Tj, h, w, K)
    Pw_i = sim3_act(Ti, Xi_i.reshape(-1, 3))
    Pw_j = sim3_act(Tj, Xj_j.reshape(-1, 3))
    Xj_i = sim3_act(sim3_inv(Ti), Pw_j).reshape(h, w, 3)
    X11 = (Xi_i + rng.normal(0, noise, Xi_i.shape)).astype(np.float32)
    X21 = (Xj_i + rng.normal(0, noise, Xj_i.shape)).astype(np.float32)
    D11 = descriptor_field(Pw_i).reshape(h, w, -1).astype(np.float32)
    D21 = descriptor_field(Pw_j).reshape(h, w, -1).astype(np.float32)
    C11 = rng.uniform(1.0, 3.0, (h, w)).astype(np.float32)
    C21 = rng.uniform(1.0, 3.0, (h, w)).astype(np.float32)
    Q11 = rng.uniform(1.5, 4.0, (h, w)).astype(np.float32)
    Q21 = rng.uniform(1.5, 4.0, (h, w)).astype(np.float32)
    return dict(X11=X11, X21=X21, D11=D11, D21=D21, C11=C11, C21=C21, Q11=Q11, Q21=Q21, Ti=Ti, Tj=Tj, K=K)


def render_rgb(T_wc, h=384, w=512, K=None):
    """Procedural texture of the room walls -> (3,h,w) f32 in [-1,1] (ImgNorm range)."""
    X = render_pointmap(T_wc, h, w, K)
    Pw = sim3_act(T_wc, X.reshape(-1, 3)).reshape(h, w, 3)
    r = np.sin(3.1 * Pw[..., 0] + 1.7 * Pw[..., 1])
    g = np.sin(2.3 * Pw[..., 1] + 2.9 * Pw[..., 2])
    b = np.sin(4.1 * Pw[..., 2] + 1.3 * Pw[..., 0])
    return np.stack((r, g, b), 0).astype(np.float32)


def project(K, X):
    z = X[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = K[0, 0] * X[..., 0] / z + K[0, 2]
        v = K[1, 1] * X[..., 1] / z + K[1, 2]
    return u, v, z


def make_graph(n_kf=4, h=24, w=32, stride=8, extra_edges=2, seed=0, pose_noise=0.01, point_noise=0.0,
               n_frames=1000, pairs=None):
    """A small factor graph in the reference's layout (global_opt.py:106-121): keyframes every
    `stride` frames, edges (k-1,k) plus `extra_edges` random earlier ones, BOTH directions
    (prep_two_way_edges).  Correspondences are geometric nearest pixels with an occlusion test.
    Returns a dict of numpy arrays: Twc_gt/Twc (P,8) f32, Xs (P,HW,3), Cs (P,HW,1), K (3,3),
    ii/jj (E) i64 global ids, idx_ii2jj (E,HW) i64, valid_match (E,HW,1) bool, Q (E,HW,1) f32."""
    rng = np.random.default_rng(seed)
    K = intrinsics(h, w)
    ids = np.arange(n_kf) * 3 + 5  # non-contiguous global keyframe ids on purpose
    T_gt = np.stack([camera_pose(k * stride, n_frames) for k in range(n_kf)])
    Xs = np.stack([render_pointmap(T, h, w, K).reshape(-1, 3) for T in T_gt])
    Xs = Xs + rng.normal(0, point_noise, Xs.shape) if point_noise > 0 else Xs
    Cs = rng.uniform(1.0, 3.0, (n_kf, h * w, 1))
    und = [(k - 1, k) for k in range(1, n_kf)]
    for k in range(2, n_kf):
        for m in rng.choice(k - 1, size=min(extra_edges, k - 1), replace=False):
            und.append((int(m), k))
    if pairs is not None:      # explicit undirected keyframe pairs instead of chain + random earlier ones
        und = [(int(a), int(b)) for a, b in pairs]
    ii = np.array([a for a, b in und] + [b for a, b in und], np.int64)
    jj = np.array([b for a, b in und] + [a for a, b in und], np.int64)
    E = len(ii)
    idx = np.zeros((E, h * w), np.int64)
    valid = np.zeros((E, h * w, 1), bool)
    for e in range(E):
        i, j = ii[e], jj[e]
        Pw = sim3_act(T_gt[j], Xs[j])
        Xi_pred = sim3_act(sim3_inv(T_gt[i]), Pw)
        u, v, z = project(K, Xi_pred)
        ui, vi = np.rint(u).astype(np.int64), np.rint(v).astype(np.int64)
        inside = (z > 0.05) & (ui >= 0) & (ui < w) & (vi >= 0) & (vi < h)
        lin = np.where(inside, vi * w + ui, 0)
        occl = np.linalg.norm(Xs[i][lin] - Xi_pred, axis=-1) < 0.1
        idx[e] = lin
        valid[e, :, 0] = inside & occl
    Q = rng.uniform(1.0, 4.0, (E, h * w, 1))
    # noisy initial poses (first pose exact: it is the pinned one)
    T = T_gt.copy()
    for k in range(1, n_kf):
        xi = rng.normal(0, pose_noise, 7)
        dq = quat_from_rotvec(xi[3:6])
        ds = np.exp(xi[6] * 0.5)
        q = T[k, 3:7]
        # left-multiply a small Sim3
        qn = np.array([
            dq[3] * q[0] + dq[0] * q[3] + dq[1] * q[2] - dq[2] * q[1],
            dq[3] * q[1] - dq[0] * q[2] + dq[1] * q[3] + dq[2] * q[0],
            dq[3] * q[2] + dq[0] * q[1] - dq[1] * q[0] + dq[2] * q[3],
            dq[3] * q[3] - dq[0] * q[0] - dq[1] * q[1] - dq[2] * q[2]])
        T[k, :3] = ds * quat_rotate(dq, T[k, :3][None])[0] + xi[:3]
        T[k, 3:7] = qn
        T[k, 7] = ds * T[k, 7]
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    return dict(Twc_gt=f32(T_gt), Twc=f32(T), Xs=f32(Xs), Cs=f32(Cs), K=f32(K), ii=ids[ii], jj=ids[jj],
                idx_ii2jj=idx, valid_match=valid, Q=f32(Q), h=h, w=w)
