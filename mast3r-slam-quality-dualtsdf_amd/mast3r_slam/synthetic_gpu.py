"""The procedural room of mast3r_slam.synthetic rendered ON THE DEVICE (torch ops on HIP tensors, no host round trip),
and the two stand-ins a run without trained weights / retrieval codebook needs (BASELINE config 3):

* `RoomGeometryModel` - the model surface SlamSystem drives (`_encode_image`, `decode_pair`).  It runs the REAL network
  (a `Mast3rHIP` with random-init weights: every kernel of the two-view forward is launched and timed) and hands back
  the room's geometry in the heads' output convention instead of the meaningless pointmaps random weights produce,
  so that matching, tracking, keyframe selection, the factor graph and both TSDFs work on meaningful data and every
  decision of the product loop is real.  The frame's position on the camera path travels with the data: pixel
  [0,0,0] of the image and element [0,0] of the encoder tokens hold the path index (read on the device).
* `PoseProximityRetriever` - stand-in for RetrievalDatabase.update (retrieval_database.py:43-72): the reference
  retrieves the k most similar keyframes by ASMK image similarity; without the codebook the proxy for "similar image"
  is "similar view": the earlier keyframes whose ground-truth optical axis and position are closest.
"""
import math

import numpy as np
import torch

from mast3r_slam import synthetic

TAG_SCALE = 4096.0   # img[0,0,0] = path index / TAG_SCALE (exact in f32 for indices < 2^24)


def camera_pose_t(k, n_frames=1000):
    """synthetic.camera_pose for a tensor of path indices k (B,) -> (B,8) f64 [t, q(xyzw), s=1]."""
    k = k.to(torch.float64)
    a = 2 * math.pi * k / max(n_frames, 1) * 3.0
    t = torch.stack((1.2 * torch.cos(a), 0.6 * torch.sin(0.7 * a), 0.4 * torch.sin(a)), -1)
    rot = torch.stack((0.15 * torch.sin(0.9 * a), a * 0.35, 0.1 * torch.cos(1.3 * a)), -1)
    th = torch.linalg.norm(rot, dim=-1, keepdim=True)
    small = th < 1e-12
    ths = torch.where(small, torch.ones_like(th), th)
    qv = torch.where(small, torch.zeros_like(rot), rot / ths * torch.sin(ths / 2))
    qw = torch.where(small, torch.ones_like(th), torch.cos(ths / 2))
    return torch.cat((t, qv, qw, torch.ones_like(th)), -1)


def quat_rotate_t(q, X):
    """q (B,4) xyzw, X (B,N,3)."""
    qv, qw = q[:, None, :3], q[:, None, 3:4]
    uv = 2.0 * torch.linalg.cross(qv.expand_as(X), X)
    return X + qw * uv + torch.linalg.cross(qv.expand_as(X), uv)


def sim3_act_t(T, X):
    return T[:, None, 7:8] * quat_rotate_t(T[:, 3:7], X) + T[:, None, :3]


def sim3_inv_t(T):
    qi = torch.cat((-T[:, 3:6], T[:, 6:7]), -1)
    si = 1.0 / T[:, 7:8]
    ti = -si * quat_rotate_t(qi, T[:, None, :3])[:, 0]
    return torch.cat((ti, qi, si), -1)


class RoomRenderer:
    """Camera-frame pointmaps, descriptors and RGB of the box room for batches of path indices, on `device`."""

    def __init__(self, device, h=384, w=512, n_frames=1000, fdim=24, desc_seed=7):
        self.device, self.h, self.w, self.n_frames = device, h, w, n_frames
        K = synthetic.intrinsics(h, w)
        self.rays = torch.from_numpy(synthetic.pixel_rays(h, w, K).reshape(-1, 3)).to(device)      # (HW,3) f64, z = 1
        rng = np.random.default_rng(desc_seed)                                                     # = descriptor_field
        self.Wm = torch.from_numpy(rng.normal(0, 6.0, (3, fdim))).to(device)
        self.ph = torch.from_numpy(rng.uniform(0, 2 * np.pi, fdim)).to(device)
        self.half = torch.from_numpy(synthetic.ROOM_HALF).to(device)
        self.K = K
        self._Wm_host = np.ascontiguousarray(self.Wm.cpu().numpy(), np.float64)
        self._ph_host = np.ascontiguousarray(self.ph.cpu().numpy(), np.float64)

    def pointmap(self, T):
        """(B,8) f64 -> camera-frame points (B,HW,3) f64 (synthetic.render_pointmap)."""
        d_cam = self.rays[None].expand(T.shape[0], -1, -1)
        d_world = quat_rotate_t(T[:, 3:7], d_cam)
        org = T[:, None, :3]
        t1 = (self.half - org) / d_world
        t2 = (-self.half - org) / d_world
        t = torch.where(d_world > 0, t1, t2)
        t = torch.where(d_world.abs() < 1e-12, torch.full_like(t, float("inf")), t)
        return d_cam * t.min(-1, keepdim=True).values

    def descriptors(self, Pw):
        D = torch.sin(Pw @ self.Wm + self.ph)
        return D / torch.linalg.norm(D, dim=-1, keepdim=True)

    def rgb(self, k):
        """(B,) path indices -> (B,3,h,w) f32 in [-1,1] (synthetic.render_rgb), tagged with the path index."""
        T = camera_pose_t(k, self.n_frames)
        Pw = sim3_act_t(T, self.pointmap(T))
        r = torch.sin(3.1 * Pw[..., 0] + 1.7 * Pw[..., 1])
        g = torch.sin(2.3 * Pw[..., 1] + 2.9 * Pw[..., 2])
        b = torch.sin(4.1 * Pw[..., 2] + 1.3 * Pw[..., 0])
        img = torch.stack((r, g, b), 1).reshape(-1, 3, self.h, self.w).float()
        img[:, 0, 0, 0] = (k.to(torch.float64) / TAG_SCALE).float()
        return img

    def _hash01(self, ki, kj, salt, channels=1):
        """Deterministic pseudo-random field in [0,1) per (pair, pixel, channel): the stand-in's noise must not depend on
        the order or the batching of the calls (the sharded backend decodes an edge on whichever rank owns it)."""
        B = ki.shape[0]
        pix = torch.arange(self.h * self.w, device=self.device, dtype=torch.float64)
        ch = torch.arange(channels, device=self.device, dtype=torch.float64)
        x = (pix[None, :, None] * 0.618033988749895 + ch[None, None, :] * 0.754877666246693
             + (ki.double() * 12.9898 + kj.double() * 78.233 + salt * 37.719)[:, None, None])
        v = torch.sin(x * 12.9898) * 43758.5453
        return (v - torch.floor(v)).reshape(B, self.h * self.w, channels)

    def pair_fused(self, ki, kj, noise=0.002):
        """pair() as ONE kernel (csrc/room.hip): same formulas, fused; device tensors only."""
        import mslam_hip as _m

        B, h, w = ki.shape[0], self.h, self.w
        f32 = dict(dtype=torch.float32, device=self.device)
        out = [dict(pts3d=torch.empty((B, h, w, 3), **f32), conf=torch.empty((B, h, w), **f32),
                    desc=torch.empty((B, h, w, 24), **f32), desc_conf=torch.empty((B, h, w), **f32)) for _ in range(2)]
        kif, kjf = ki.float().contiguous(), kj.float().contiguous()
        K = self.K
        rc = _m.lib().mslam_room_pair(
            _m.ptr(kif), _m.ptr(kjf), B, h, w, self.n_frames, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
            float(K[1, 2]), float(noise), self._Wm_host.ctypes.data, self._ph_host.ctypes.data,
            *[_m.ptr(out[s_][k]) for s_ in range(2) for k in ("pts3d", "conf", "desc", "desc_conf")], _m.stream_ptr())
        _m.check(rc, "room_pair")
        return out[0], out[1]

    def pair(self, ki, kj, noise=0.002, generator=None):
        """Two-view geometry in MASt3R's output convention for view i (path index ki) and view j, both (B,):
        res1 = view i in frame i, res2 = view j expressed in frame i (synthetic.make_pair).  Point noise (Gaussian,
        `noise` metres) and the confidence maps are deterministic functions of (ki, kj, pixel); `generator` is unused
        and kept for the signature."""
        B, h, w = ki.shape[0], self.h, self.w
        Ti, Tj = camera_pose_t(ki, self.n_frames), camera_pose_t(kj, self.n_frames)
        Xi_i, Xj_j = self.pointmap(Ti), self.pointmap(Tj)
        Pw_i, Pw_j = sim3_act_t(Ti, Xi_i), sim3_act_t(Tj, Xj_j)
        Xj_i = sim3_act_t(sim3_inv_t(Ti), Pw_j)
        out = []
        for side, (X, Pw) in enumerate(((Xi_i, Pw_i), (Xj_i, Pw_j))):
            u1 = self._hash01(ki, kj, 4 * side + 0, 3).clamp_min(1e-12)
            u2 = self._hash01(ki, kj, 4 * side + 1, 3)
            gauss = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2 * math.pi * u2)          # Box-Muller
            conf = 1.0 + 2.0 * self._hash01(ki, kj, 4 * side + 2)[..., 0]
            dconf = 1.5 + 2.5 * self._hash01(ki, kj, 4 * side + 3)[..., 0]
            out.append(dict(pts3d=(X + noise * gauss).float().reshape(B, h, w, 3), conf=conf.float().reshape(B, h, w),
                            desc=self.descriptors(Pw).float().reshape(B, h, w, -1),
                            desc_conf=dconf.float().reshape(B, h, w)))
        return out[0], out[1]


class RoomGeometryModel:
    """See the module docstring.  `net` = a Mast3rHIP (or None: geometry only, for functional tests)."""

    def __init__(self, net, device, h=384, w=512, n_frames=1000, noise=0.002, seed=0):
        self.net, self.device = net, device
        self.room = RoomRenderer(device, h, w, n_frames)
        self.noise = noise
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.enc_calls = self.enc_rows = self.dec_calls = self.dec_rows = 0

    @torch.inference_mode()
    def _encode_image(self, img, true_shape=None):
        self.enc_calls += 1
        self.enc_rows += img.shape[0]
        k = torch.round(img[:, 0, 0, 0].double() * TAG_SCALE).float()
        if self.net is not None:
            feat, pos, _ = self.net._encode_image(img, true_shape)
        else:
            n = (img.shape[-2] // 16) * (img.shape[-1] // 16)
            feat = torch.zeros((img.shape[0], n, 1024), device=self.device)
            pos = torch.zeros((img.shape[0], n, 2), dtype=torch.long, device=self.device)
        feat[:, 0, 0] = k
        return feat, pos, None

    @torch.inference_mode()
    def decode_pair(self, feat1, feat2, h, w):
        self.dec_calls += 1
        self.dec_rows += feat1.shape[0]
        if self.net is not None:
            self.net.decode_pair(feat1, feat2, h, w)     # the real two-view forward; its outputs are dropped
        if feat1.is_cuda:
            return self.room.pair_fused(feat1[:, 0, 0], feat2[:, 0, 0], self.noise)
        return self.room.pair(feat1[:, 0, 0], feat2[:, 0, 0], self.noise, self.gen)


class PoseProximityRetriever:
    """update(frame, add_after_query, k, min_thresh) -> keyframe indices, like RetrievalDatabase.update.  The database
    index of a keyframe is the order of insertion, as in the reference (kf_counter, retrieval_database.py:66-70)."""

    def __init__(self, path_index_of, n_frames=1000, max_dist=1.2, min_cos=0.75, exclude_recent=1):
        self.path_index_of, self.n_frames = path_index_of, n_frames
        self.max_dist, self.min_cos, self.exclude_recent = max_dist, min_cos, exclude_recent
        self.pos, self.axis = [], []

    def _view(self, frame):
        T = synthetic.camera_pose(self.path_index_of(frame), self.n_frames)
        return T[:3], synthetic.quat_rotate(T[3:7], np.array([[0.0, 0.0, 1.0]]))[0]

    def update(self, frame, add_after_query=True, k=3, min_thresh=0.0):
        p, a = self._view(frame)
        out = []
        n = len(self.pos) - (self.exclude_recent if add_after_query else 0)   # the consecutive edge is added anyway
        if n > 0 and k > 0:
            P, A = np.stack(self.pos[:n]), np.stack(self.axis[:n])
            dist, cos = np.linalg.norm(P - p, axis=1), A @ a
            score = cos - 0.3 * dist
            ok = (dist < self.max_dist) & (cos > self.min_cos)
            order = np.argsort(-score)
            out = [int(i) for i in order if ok[i]][:k]
        if add_after_query:
            self.pos.append(p)
            self.axis.append(a)
        return out
