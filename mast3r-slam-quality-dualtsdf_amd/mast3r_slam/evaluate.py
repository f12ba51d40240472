"""Mirror of the reference's output writers, mast3r_slam/evaluate.py: TUM-format trajectory, coloured
point cloud (binary little-endian PLY, with or without the quality attributes) and keyframe images — the formats the
reference's `scripts/eval_*.sh` / `evo_ape` consume.  Same function names and arguments.

The reference writes PLY through `plyfile` and images / grid upsampling through `cv2`; neither is installed here, so
* `save_ply*` emit the file `plyfile`'s `PlyData([PlyElement.describe(arr, "vertex")], text=False)` is documented to
  produce (header lines + packed records) — byte parity with plyfile itself is **unpinned**;
* `save_keyframes` writes PNG through PIL (same RGB content; cv2's BGR swap is its own file convention);
* the quality grids are upsampled with cv2.resize's published INTER_LINEAR / INTER_NEAREST sampling rules — **unpinned**.
`save_traj` is pinned against the reference function (tests/golden/evaluate_traj.npz)."""
import pathlib

import numpy as np
import torch

from mast3r_slam.config import config
from mast3r_slam.geometry import constrain_points_to_ray
from mast3r_slam.lietorch_utils import as_SE3


def prepare_savedir(args, dataset):
    """evaluate.py:14-20."""
    save_dir = pathlib.Path("logs")
    if args.save_as != "default":
        save_dir = save_dir / args.save_as
    save_dir.mkdir(exist_ok=True, parents=True)
    seq_name = dataset.dataset_path.stem
    return save_dir, seq_name


def save_traj(logdir, logfile, timestamps, frames, intrinsics=None):
    """evaluate.py:23-45: one line per keyframe, `t x y z qx qy qz qw` (python float repr of the float32 values)."""
    logdir = pathlib.Path(logdir)
    logdir.mkdir(exist_ok=True, parents=True)
    logfile = logdir / logfile
    with open(logfile, "w") as f:
        for i in range(len(frames)):
            keyframe = frames[i]
            t = timestamps[keyframe.frame_id]
            if intrinsics is None:
                T_WC = as_SE3(keyframe.T_WC)
            else:
                T_WC = intrinsics.refine_pose_with_calibration(keyframe)
            x, y, z, qx, qy, qz, qw = T_WC.data.cpu().numpy().reshape(-1)
            f.write(f"{t} {x} {y} {z} {qx} {qy} {qz} {qw}\n")


def _world_points(keyframe):
    if config["use_calib"]:
        X_canon = constrain_points_to_ray(keyframe.img_shape.flatten()[:2], keyframe.X_canon[None], keyframe.K)
        keyframe.X_canon = X_canon.squeeze(0)
    pW = keyframe.T_WC.act(keyframe.X_canon).cpu().numpy().reshape(-1, 3)
    color = (keyframe.uimg.cpu().numpy() * 255).astype(np.uint8).reshape(-1, 3)
    return pW, color


def save_reconstruction(savedir, filename, keyframes, c_conf_threshold):
    """evaluate.py:48-71."""
    savedir = pathlib.Path(savedir)
    savedir.mkdir(exist_ok=True, parents=True)
    pointclouds, colors = [], []
    for i in range(len(keyframes)):
        keyframe = keyframes[i]
        pW, color = _world_points(keyframe)
        valid = keyframe.get_average_conf().cpu().numpy().astype(np.float32).reshape(-1) > c_conf_threshold
        pointclouds.append(pW[valid])
        colors.append(color[valid])
    save_ply(savedir / filename, np.concatenate(pointclouds, axis=0), np.concatenate(colors, axis=0))


def save_keyframes(savedir, timestamps, keyframes):
    """evaluate.py:74-87 (PIL instead of cv2.imwrite)."""
    import PIL.Image

    savedir = pathlib.Path(savedir)
    savedir.mkdir(exist_ok=True, parents=True)
    for i in range(len(keyframes)):
        keyframe = keyframes[i]
        t = timestamps[keyframe.frame_id]
        PIL.Image.fromarray((keyframe.uimg.cpu().numpy() * 255).astype(np.uint8)).save(str(savedir / f"{t}.png"))


_PLY_TYPES = {"f4": "float", "u1": "uchar", "f8": "double", "i4": "int", "u4": "uint", "i2": "short", "u2": "ushort", "i1": "char"}


def _write_ply(filename, pcd):
    """Binary little-endian PLY with one `vertex` element described by the structured array `pcd`."""
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(pcd)}"]
    for name in pcd.dtype.names:
        header.append(f"property {_PLY_TYPES[pcd.dtype[name].str[1:]]} {name}")
    header.append("end_header")
    with open(filename, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(pcd.astype(pcd.dtype.newbyteorder("<")).tobytes())


def save_ply(filename, points, colors):
    """evaluate.py:90-106."""
    colors = colors.astype(np.uint8)
    pcd = np.empty(len(points), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    pcd["x"], pcd["y"], pcd["z"] = points.T
    pcd["red"], pcd["green"], pcd["blue"] = colors.T
    _write_ply(filename, pcd)


def _resize_grid(g, H, W, mode):
    """cv2.resize(g, (W, H), INTER_NEAREST | INTER_LINEAR) for a 2-D float32 grid: nearest takes src = floor(dst * scale),
    linear samples at (dst + 0.5) * scale - 0.5 with the border replicated."""
    gh, gw = g.shape
    sy, sx = gh / H, gw / W
    if mode == "nearest":
        iy = np.minimum(np.floor(np.arange(H) * sy).astype(np.int64), gh - 1)
        ix = np.minimum(np.floor(np.arange(W) * sx).astype(np.int64), gw - 1)
        return g[iy][:, ix]
    fy = (np.arange(H, dtype=np.float32) + 0.5) * np.float32(sy) - 0.5
    fx = (np.arange(W, dtype=np.float32) + 0.5) * np.float32(sx) - 0.5
    y0, x0 = np.floor(fy).astype(np.int64), np.floor(fx).astype(np.int64)
    wy, wx = (fy - y0).astype(np.float32), (fx - x0).astype(np.float32)
    y0c, y1c = np.clip(y0, 0, gh - 1), np.clip(y0 + 1, 0, gh - 1)
    x0c, x1c = np.clip(x0, 0, gw - 1), np.clip(x0 + 1, 0, gw - 1)
    top = g[y0c][:, x0c] * (1 - wx) + g[y0c][:, x1c] * wx
    bot = g[y1c][:, x0c] * (1 - wx) + g[y1c][:, x1c] * wx
    return (top * (1 - wy)[:, None] + bot * wy[:, None]).astype(np.float32)


def save_ply_with_quality(savedir, filename, keyframes, c_conf_threshold, quality_service, patch_size=16):
    """evaluate.py:108-186: the reconstruction with the per-patch quality grids (r, delta_cov, u, class_id, priority)
    upsampled to the image and attached to every vertex."""
    savedir = pathlib.Path(savedir)
    savedir.mkdir(exist_ok=True, parents=True)
    cols = {k: [] for k in ("points", "colors", "r", "delta_cov", "u", "class_id", "priority")}
    for i in range(len(keyframes)):
        kf = keyframes[i]
        pW, col = _world_points(kf)
        valid = kf.get_average_conf().cpu().numpy().astype(np.float32).reshape(-1) > c_conf_threshold
        H, W = int(kf.img_shape.flatten()[0]), int(kf.img_shape.flatten()[1])
        res = quality_service.get(kf.frame_id) if quality_service is not None else None
        if res is not None:
            def up(g, mode):
                gnp = g.detach().cpu().numpy() if torch.is_tensor(g) else np.asarray(g)
                return _resize_grid(gnp.astype(np.float32).reshape(gnp.shape[-2], gnp.shape[-1]), H, W, mode).reshape(-1)
            cid = res["class_id"].float() if torch.is_tensor(res["class_id"]) else np.asarray(res["class_id"]).astype(np.float32)
            q = dict(delta_cov=up(res["delta_cov"], "linear"), r=up(res["r"], "linear"), u=up(res["u"], "linear"),
                     class_id=up(cid, "nearest").astype(np.uint8), priority=up(res["priority"], "linear"))
        else:
            n = H * W
            q = dict(delta_cov=np.zeros(n, np.float32), r=np.zeros(n, np.float32), u=np.zeros(n, np.float32),
                     class_id=np.zeros(n, np.uint8), priority=np.zeros(n, np.float32))
        cols["points"].append(pW[valid])
        cols["colors"].append(col[valid])
        for k, v in q.items():
            cols[k].append(v[valid])
    points, colors = np.concatenate(cols["points"], 0), np.concatenate(cols["colors"], 0)
    pcd = np.empty(points.shape[0], dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1"),
                                           ("r", "f4"), ("delta_cov", "f4"), ("u", "f4"), ("class_id", "u1"), ("priority", "f4")])
    pcd["x"], pcd["y"], pcd["z"] = points.T
    pcd["red"], pcd["green"], pcd["blue"] = colors.T
    for k in ("r", "delta_cov", "u", "priority"):
        pcd[k] = np.concatenate(cols[k], 0).astype(np.float32)
    pcd["class_id"] = np.concatenate(cols["class_id"], 0).astype(np.uint8)
    _write_ply(savedir / filename, pcd)
