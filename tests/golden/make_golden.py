"""Generates the committed golden fixtures by RUNNING the importable pieces of the reference
(read-only at /root/reference) in the build container.  The reference never travels to the GPU
box; only these small .npz data files do.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]

Sections: prep  tracker  tsdf_global  tsdf_refine  network  resize  geometry  quality  frame  factor_graph  track_logic  utils_wrappers  refine_block  refine_schedule  retrieval_quantize  retrieval_asmk  evaluate  dataloader
Every fixture records numpy/torch versions (the global TSDF arithmetic depends on NumPy's
promotion rules: the container has NumPy 2.x (NEP 50), the reference pins numpy==1.26.4).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd"))


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def meta():
    return dict(numpy_version=np.__version__, torch_version=torch.__version__)


def section_prep():
    """prep_for_iter_proj (matching.py:25-49) = F.normalize + img_gradient (image.py:5-38)."""
    import torch.nn.functional as F
    from mast3r_slam import synthetic

    image = load_by_path("ref_image", f"{REF}/mast3r_slam/image.py")
    pair = synthetic.make_pair(0, 6, h=48, w=64, seed=3)
    X11 = torch.from_numpy(pair["X11"])[None]
    X21 = torch.from_numpy(pair["X21"])[None]
    # --- the reference lines, verbatim in behaviour ---
    rays_img = F.normalize(X11, dim=-1).permute(0, 3, 1, 2)
    gx, gy = image.img_gradient(rays_img)
    rays_with_grad = torch.cat((rays_img, gx, gy), dim=1).permute(0, 2, 3, 1).contiguous()
    pts3d_norm = F.normalize(X21.view(1, -1, 3), dim=-1)
    np.savez_compressed(
        os.path.join(HERE, "prep_iter_proj.npz"), X11=X11.numpy(), X21=X21.numpy(),
        rays_with_grad=rays_with_grad.numpy(), pts3d_norm=pts3d_norm.numpy(), **meta(),
    )
    print("prep_iter_proj.npz", rays_with_grad.shape)


SECTIONS = {"prep": section_prep}



def section_tracker():
    """Normal equations of ONE relative-pose GN step built with the reference's own pure-torch
    geometry (mast3r_slam/geometry.py: act_Sim3, point_to_ray_dist, project_calib, backproject;
    mast3r_slam/nonlinear_optimizer.py: huber) exactly as tracker.py:208-318 composes them, in
    float64.  lietorch is not installed: geometry.py only needs the module object for a type
    annotation and a duck-typed pose with .act(), supplied here."""
    import types

    from mast3r_slam import synthetic

    sys.modules.setdefault("lietorch", types.SimpleNamespace(Sim3=object))
    geom = load_by_path("ref_geometry", f"{REF}/mast3r_slam/geometry.py")
    nlo = load_by_path("ref_nlo", f"{REF}/mast3r_slam/nonlinear_optimizer.py")

    class Pose:  # duck-typed lietorch.Sim3: X -> s R X + t
        def __init__(self, T):
            self.T = np.asarray(T, np.float64)

        def act(self, X):
            return torch.from_numpy(synthetic.sim3_act(self.T, X.numpy()))

    g = synthetic.make_graph(n_kf=2, h=12, w=16, seed=0, pose_noise=0.02, extra_edges=0)
    sys.path.insert(0, ROOT)
    import oracle

    rng = np.random.default_rng(0)
    Tj = oracle.sim3_exp(rng.normal(0, 0.05, 7))[0]
    Twc = np.stack([np.array([0, 0, 0, 0, 0, 0, 1, 1], np.float32), Tj])
    idx = torch.from_numpy(g["idx_ii2jj"][0])
    valid = torch.from_numpy(g["valid_match"][0]).double()
    Q = torch.from_numpy(g["Q"][0]).double()
    Cs = torch.from_numpy(g["Cs"]).double()
    K = torch.from_numpy(g["K"]).double()
    h, w = g["h"], g["w"]
    out = dict(Twc=Twc)
    for kind in ("rays", "calib"):
        Xs = torch.from_numpy(g["Xs"]).double()
        if kind == "calib":  # global_opt.py:180-182
            Xs = geom.constrain_points_to_ray((h, w), Xs.view(2, h, w, 3), K).view(2, h * w, 3)
            out["Xs_calib"] = Xs.numpy().astype(np.float32)
        Xi = Xs[0][idx]           # measurement in frame i (gathered)   gn_kernels.cu:916
        Xj = Xs[1]                # points of j
        valid_all = valid * (Q > 1.5) * (Cs[0][idx] > 0.0) * (Cs[1] > 0.0)
        P, dP_dT = geom.act_Sim3(Pose(Tj.astype(np.float64)), Xj, jacobian=True)
        if kind == "rays":
            rd_i = geom.point_to_ray_dist(Xi, jacobian=False)
            rd_j, drd_dP = geom.point_to_ray_dist(P, jacobian=True)
            r = rd_i - rd_j                                   # tracker.py:241 (z - h(x))
            J = -drd_dP @ dP_dT                               # tracker.py:243
            s_a, s_b, na = 0.003, 10.0, 3
        else:
            pz, dpz_dP, valid_proj = geom.project_calib(P, K, (h, w), jacobian=True, border=-10, z_eps=1e-6)
            uv = geom.get_pixel_coords(1, (h, w), device="cpu", dtype=torch.float64).view(-1, 2)[idx]
            valid_i = Xi[:, 2:3] > 1e-6
            meas = torch.cat((uv, torch.log(Xi[:, 2:3])), -1)
            meas[~valid_i.repeat(1, 3)] = 0.0                 # tracker.py:199-203
            r = meas - pz                                     # tracker.py:300
            J = -dpz_dP @ dP_dT
            valid_all = valid_all * valid_proj * valid_i
            s_a, s_b, na = 1.0, 10.0, 2
        sqrt_info = torch.cat(((1 / s_a * valid_all * torch.sqrt(Q)).repeat(1, na),
                               1 / s_b * valid_all * torch.sqrt(Q)), 1)   # tracker.py:227-229
        whitened = sqrt_info * r
        robust = sqrt_info * torch.sqrt(nlo.huber(whitened, k=1.345))      # tracker.py:209-212
        A = (robust[..., None] * J).view(-1, 7)
        b = (robust * r).view(-1, 1)
        out[f"H_{kind}"] = (A.T @ A).numpy()
        # kernel convention: err = -r, raw rows = -J  =>  gs_j = sum w err x = A^T b
        out[f"g_{kind}"] = (A.T @ b).numpy()[:, 0]
    np.savez_compressed(os.path.join(HERE, "tracker_formulae.npz"), **out, **meta())
    print("tracker_formulae.npz", {k: v.shape for k, v in out.items()})


SECTIONS["tracker"] = section_tracker


def _tsdf_inputs():
    """Two keyframes of a small synthetic room, world-space points + confidences + camera origins."""
    from mast3r_slam import synthetic

    out = []
    for kf, k in enumerate((0, 12)):
        T = synthetic.camera_pose(k)
        X = synthetic.render_pointmap(T, 24, 32).reshape(-1, 3)
        rng = np.random.default_rng(100 + kf)
        sel = rng.permutation(X.shape[0])[:400]
        pw = synthetic.sim3_act(T, X[sel]).astype(np.float32)
        conf = rng.uniform(0.5, 30.0, sel.shape[0]).astype(np.float32).astype(np.float64)  # big: saturates 100
        out.append((pw, conf, T[:3].astype(np.float32)))
    return out


def section_tsdf_global():
    """TSDFVolume.integrate / query / gradient and the TSDF pose normal equations, by running the
    reference's tsdf/global_volume.py and tsdf/tsdf_optimizer.py (loaded by file path)."""
    import types

    sys.modules.setdefault("lietorch", types.SimpleNamespace(Sim3=object))
    gv = load_by_path("ref_global_volume", f"{REF}/mast3r_slam/tsdf/global_volume.py")
    to = load_by_path("ref_tsdf_optimizer", f"{REF}/mast3r_slam/tsdf/tsdf_optimizer.py")
    vol = gv.TSDFVolume(voxel_size=0.03, truncation=0.12, max_weight=100.0, min_weight=1.0e-3)
    data = _tsdf_inputs()
    save = {}
    for kf, (pw, conf, org) in enumerate(data):
        fused = vol.integrate(pw, conf, org)
        keys = np.array(sorted(vol._voxels.keys()), np.int64)
        save[f"kf{kf}_points"] = pw; save[f"kf{kf}_conf"] = conf; save[f"kf{kf}_origin"] = org
        save[f"kf{kf}_fused"] = fused
        save[f"kf{kf}_keys"] = keys
        save[f"kf{kf}_tsdf"] = np.array([float(vol._voxels[tuple(k)].tsdf) for k in keys])
        save[f"kf{kf}_weight"] = np.array([float(vol._voxels[tuple(k)].weight) for k in keys])
    # queries: the integrated points themselves, jittered, plus far-away misses
    rng = np.random.default_rng(5)
    q = np.concatenate([data[0][0][:150] + rng.normal(0, 0.02, (150, 3)).astype(np.float32),
                        data[1][0][:150], np.full((4, 3), 50.0, np.float32)]).astype(np.float32)
    val = np.zeros(len(q)); grad = np.zeros((len(q), 3)); st = np.zeros(len(q), np.uint8)
    for i, p in enumerate(q):
        v, g = vol.query(p)
        if v is not None:
            val[i] = float(v); st[i] = 1
            if g is not None:
                grad[i] = g; st[i] = 2
    save.update(query_points=q, query_value=val, query_grad=grad, query_status=st)
    opt = to.TSDFPoseOptimizer(vol, None, {"lambda": 0.15}, False, "cpu")
    qconf = rng.uniform(0.05, 3.0, len(q)).astype(np.float32)
    res, jac, wts = opt._build_linear_system(q, qconf)
    H, b = opt._accumulate_system(res, jac, wts)
    save.update(pose_conf=qconf, pose_H=H, pose_b=b, pose_used=len(res))
    np.savez_compressed(os.path.join(HERE, "tsdf_global.npz"), **save, **meta())
    print("tsdf_global.npz voxels:", len(keys), "saturated:", int((save["kf1_weight"] >= 100).sum()),
          "query status counts:", np.bincount(st, minlength=3))


SECTIONS["tsdf_global"] = section_tsdf_global


def section_network():
    """MASt3R two-view forward of the REFERENCE model classes (thirdparty/mast3r, importable on CPU)
    at a reduced width/depth, with weights overwritten by oracle.mast3r_ref.init_state_dict(seed) so
    that no weight file needs to be committed: the fixture holds inputs + reference outputs only."""
    sys.path.insert(0, f"{REF}/thirdparty/mast3r")
    import mast3r.utils.path_to_dust3r  # noqa: F401
    from mast3r.model import AsymmetricMASt3R

    sys.path.insert(0, ROOT)
    from oracle import mast3r_ref as R

    inf = float("inf")
    cfg = R.Mast3rConfig(enc_dim=128, enc_depth=2, enc_heads=2, dec_dim=128, dec_depth=12, dec_heads=2)
    model = AsymmetricMASt3R(
        pos_embed="RoPE100", patch_embed_cls="PatchEmbedDust3R", img_size=(512, 512), head_type="catmlp+dpt",
        output_mode="pts3d+desc24", depth_mode=("exp", -inf, inf), conf_mode=("exp", 1, inf),
        enc_embed_dim=cfg.enc_dim, enc_depth=cfg.enc_depth, enc_num_heads=cfg.enc_heads,
        dec_embed_dim=cfg.dec_dim, dec_depth=cfg.dec_depth, dec_num_heads=cfg.dec_heads, two_confs=True,
        desc_conf_mode=("exp", 0, inf), landscape_only=False).eval()
    sd = R.init_state_dict(cfg, seed=1234)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # layer{n}_rn are aliases of layer_rn.{n-1} (same Parameter objects, dpt_block.py:68-73)
    assert all(k == "mask_token" or "_rn.weight" in k for k in missing), missing
    H, W = 64, 96   # 4 x 6 = 24 tokens (the HIP attention kernel wants a multiple of 8)
    g = torch.Generator().manual_seed(7)
    img1 = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    img2 = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    ts = torch.tensor([[H, W]])
    with torch.inference_mode():
        f1, p1, _ = model._encode_image(img1, ts)
        f2, p2, _ = model._encode_image(img2, ts)
        dec1, dec2 = model._decoder(f1, p1, f2, p2)
        dec1, dec2 = list(dec1), list(dec2)
        r1 = model._downstream_head(1, [t.float() for t in dec1], ts)
        r2 = model._downstream_head(2, [t.float() for t in dec2], ts)
    out = dict(img1=img1.numpy(), img2=img2.numpy(), feat1=f1.numpy(), feat2=f2.numpy(), pos1=p1.numpy(),
               dec1_last=dec1[-1].numpy(), dec2_last=dec2[-1].numpy(), dec1_6=dec1[6].numpy(),
               cfg=np.array([cfg.enc_dim, cfg.enc_depth, cfg.enc_heads, cfg.dec_dim, cfg.dec_depth, cfg.dec_heads]),
               seed=1234)
    for h, r in ((1, r1), (2, r2)):
        for k in ("pts3d", "conf", "desc", "desc_conf"):
            out[f"head{h}_{k}"] = r[k].numpy()
    np.savez_compressed(os.path.join(HERE, "mast3r_small.npz"), **out, **meta())
    print("mast3r_small.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


SECTIONS["network"] = section_network


def section_tsdf_refine():
    """Local dense-block TSDF of the dual-TSDF refiner: _build_tsdf_robust (tsdf_refine.py:837-940) and
    _extract_surface_safe + _sample_tsdf_trilinear (:942-1064), by running the reference's
    tsdf_refine.py (loaded by file path; its optional geometry import falls back, :21-29) with a
    duck-typed Sim3 pose exposing .matrix() and .act()."""
    import contextlib
    import io

    from mast3r_slam import synthetic

    tr = load_by_path("ref_tsdf_refine", f"{REF}/mast3r_slam/tsdf_refine.py")
    cfg = dict(enabled=True, window_size=5, voxel_size=0.02, trunc_dist=0.08, max_grid_dim=64, roi_size=0.4,
               ray_samples=64, max_displacement=0.015, min_weight_threshold=0.01, confidence_boost=0.08,
               confidence_max=1.3, min_hit_rate=0.05, max_rois_per_kf=3, min_confidence=0.2)

    class Pose:
        def __init__(self, T):
            self.T = np.asarray(T, np.float64)

        def matrix(self):
            import scipy.spatial.transform as sst
            M = np.eye(4, dtype=np.float32)
            M[:3, :3] = (self.T[7] * sst.Rotation.from_quat(self.T[3:7]).as_matrix()).astype(np.float32)
            M[:3, 3] = self.T[:3].astype(np.float32)
            return torch.from_numpy(M)[None]

        def act(self, X):
            return torch.from_numpy(synthetic.sim3_act(self.T, X.numpy().astype(np.float64)).astype(np.float32))

    with contextlib.redirect_stdout(io.StringIO()):
        ref = tr.TSDFRefiner(dict(cfg), None, None, "cpu")
    H, W = 48, 64
    Tcam = synthetic.camera_pose(2)
    X = synthetic.render_pointmap(Tcam, H, W).reshape(-1, 3)
    rng = np.random.default_rng(4)
    X = (X + rng.normal(0, 0.003, X.shape)).astype(np.float32)
    C = rng.uniform(0.1, 1.2, H * W).astype(np.float32)
    save = dict(X=X, C=C, H=H, W=W)
    # case A: identity pose (camera frame == world frame: the only setting in which the reference's
    # ray cast, done in camera coordinates, meets the world-frame grid);  case B: a small Sim3 pose
    for name, T in (("A", np.array([0, 0, 0, 0, 0, 0, 1, 1.0])), ("B", np.array([0.05, -0.02, 0.03, 0.01, 0.02, -0.015, 0.9996, 1.02]))):
        T[3:7] /= np.linalg.norm(T[3:7])
        pose = Pose(T)
        ys, xs = np.meshgrid(np.arange(16, 32), np.arange(24, 40), indexing="ij")   # one 16x16 patch
        mask = np.zeros(H * W, bool)
        mask[(ys * W + xs).reshape(-1)] = True
        Xw = pose.act(torch.from_numpy(X)).numpy()
        xyz_min = torch.from_numpy(Xw[mask].min(0) - 0.04)
        xyz_max = torch.from_numpy(Xw[mask].max(0) + 0.04)
        with contextlib.redirect_stdout(io.StringIO()):
            tsdf, weights = ref._build_tsdf_robust(torch.from_numpy(X), torch.from_numpy(C), None, xyz_min, xyz_max, H, W, pose)
            torch.manual_seed(123)
            perm = torch.randperm(int(mask.sum()))[:100]
            torch.manual_seed(123)
            X_ref, hits = ref._extract_surface_safe(tsdf, xyz_min, xyz_max, None, torch.from_numpy(mask), H, W, torch.from_numpy(X))
        save.update({f"{name}_pose": T.astype(np.float32), f"{name}_mask": mask, f"{name}_xyz_min": xyz_min.numpy(),
                     f"{name}_xyz_max": xyz_max.numpy(), f"{name}_tsdf": tsdf.numpy(), f"{name}_weights": weights.numpy(),
                     f"{name}_perm": perm.numpy(), f"{name}_X_refined": X_ref.numpy(), f"{name}_hits": hits.numpy()})
        print(name, "grid", tuple(tsdf.shape), "touched voxels", int((weights > 0).sum()), "hits", int(hits.sum()))
    np.savez_compressed(os.path.join(HERE, "tsdf_refine.npz"), **save, **meta())


SECTIONS["tsdf_refine"] = section_tsdf_refine


def section_resize():
    """a1: resize_img (mast3r_utils.py:236-278).  The module itself cannot be imported (torchvision, cv2,
    the retrieval stack); its two PIL functions are taken from the reference file's AST and executed with
    ImgNorm restated in numpy (dust3r/utils/image.py:23: ToTensor + Normalize(0.5, 0.5))."""
    import ast

    import PIL.Image

    path = os.path.join(REF, "mast3r_slam/mast3r_utils.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("_resize_pil_image", "resize_img")]
    assert len(keep) == 2
    ns = {"PIL": PIL, "np": np,
          "ImgNorm": lambda im: torch.from_numpy(((np.asarray(im, np.float32) / 255.0 - 0.5) / 0.5).transpose(2, 0, 1).copy())}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    rng = np.random.default_rng(7)
    out = {}
    # inputs are stored as uint8 (the function quantises to uint8 first: np.uint8(img * 255)); the float image
    # handed to both implementations is (u8 + 0.5) / 255, which truncates back to u8 exactly.  Outputs: the
    # uint8 crop; `img` is its ImgNorm and is checked against that formula in the test.
    for name, (h, w) in {"lanczos": (530, 700), "bicubic": (200, 320), "square": (300, 300)}.items():
        u8 = rng.integers(0, 256, (h // 4 + 1, w // 4 + 1, 3)).astype(np.uint8)
        u8 = np.repeat(np.repeat(u8, 4, 0), 4, 1)[:h, :w]          # 4x4 blocks: structure for the filters, compressible
        img = ((u8.astype(np.float32) + 0.5) / 255.0).astype(np.float32)
        assert np.array_equal(np.uint8(img * 255), u8)
        res, tf = ns["resize_img"](img, 512, return_transformation=True)
        out[f"{name}_in_u8"] = u8
        out[f"{name}_true_shape"] = res["true_shape"]
        out[f"{name}_uimg"] = res["unnormalized_img"]
        out[f"{name}_img_sum"] = np.float64(res["img"].double().sum().item())
        out[f"{name}_tf"] = np.array(tf, np.float64)
    np.savez_compressed(os.path.join(HERE, "resize_img.npz"), **out, **meta())
    print("resize:", {k: v.shape for k, v in out.items() if k.endswith("_uimg")})


def section_geometry():
    """geometry.py / nonlinear_optimizer.py: every tensor helper, float64, seeded inputs (lietorch stubbed as in
    the tracker section: geometry.py only needs the name for an annotation and a pose with .act())."""
    import types

    from mast3r_slam import synthetic

    sys.modules.setdefault("lietorch", types.SimpleNamespace(Sim3=object))
    geom = load_by_path("ref_geometry2", f"{REF}/mast3r_slam/geometry.py")
    nlo = load_by_path("ref_nlo2", f"{REF}/mast3r_slam/nonlinear_optimizer.py")
    g = torch.Generator().manual_seed(12)
    X = torch.randn(5, 7, 3, generator=g, dtype=torch.float64) + torch.tensor([0.0, 0.0, 2.5], dtype=torch.float64)
    X[0, 0, 2] = -0.3   # behind the camera
    K = torch.tensor([[400.0, 0, 31.5], [0, 380.0, 23.5], [0, 0, 1]], dtype=torch.float64)
    T = np.array([0.1, -0.2, 0.05, 0.1, -0.05, 0.2, 0.97, 1.3], np.float64)
    T[3:7] /= np.linalg.norm(T[3:7])

    class Pose:
        def act(self, P):
            return torch.from_numpy(synthetic.sim3_act(T, P.numpy()))

    out = dict(X=X.numpy(), K=K.numpy(), T=T)
    out["skew"] = geom.skew_sym(X).numpy()
    out["dist"] = geom.point_to_dist(X).numpy()
    rd, J = geom.point_to_ray_dist(X, jacobian=True)
    out["rd"], out["rd_J"] = rd.numpy(), J.numpy()
    pz, Jp, valid = geom.project_calib(X, K, (48, 64), jacobian=True, border=2, z_eps=1e-6)
    out["pz"], out["pz_J"], out["pz_valid"] = pz.numpy(), Jp.numpy(), valid.numpy()
    pW, Ja = geom.act_Sim3(Pose(), X, jacobian=True)
    out["act"], out["act_J"] = pW.numpy(), Ja.numpy()
    out["bp"] = geom.backproject(pz[..., :2], X[..., 2:3], K).numpy()
    out["ray"] = geom.constrain_points_to_ray((5, 7), X[None], K).numpy()
    r = torch.linspace(-6, 6, 41, dtype=torch.float64)
    out["r"], out["huber"], out["tukey"] = r.numpy(), nlo.huber(r).numpy(), nlo.tukey(r).numpy()
    out["conv"] = np.array([nlo.check_convergence(0, 1e-3, 1e-3, 10.0, c, torch.tensor([d, 0.0]))
                            for c, d in ((9.0, 1.0), (9.9999, 1.0), (5.0, 1e-4))])
    np.savez_compressed(os.path.join(HERE, "geometry.npz"), **out, **meta())
    print("geometry.npz", sorted(out))


def section_quality():
    """quality_core.py (pure torch, importable as is): compute_batch on two seeded jobs plus the individual reducers."""
    qc = load_by_path("ref_quality_core", f"{REF}/mast3r_slam/quality_core.py")
    out = {}
    for name, (h, w, ps) in {"a": (96, 128, 16), "b": (64, 64, 8)}.items():
        g = torch.Generator().manual_seed(len(name) + h)
        valid = torch.rand(h * w, generator=g) > 0.35
        valid.view(h, w)[:ps, :ps] = False                      # one empty patch: nanmedian -> NaN -> 0
        r_pix = torch.rand(h * w, generator=g) * 0.05
        r_pix[::97] = float("nan")                              # NaNs inside valid pixels are ignored too
        Ck = torch.rand(h * w, 1, generator=g) * 4.0
        Qk = torch.rand(h * w, 1, generator=g) * 3.0
        job = dict(kf_id=3, H=h, W=w, valid_kf=valid, r_pix=r_pix, Ck=Ck, Qk=Qk, t_norm=torch.tensor(0.04),
                   theta=torch.tensor(0.12))
        if name == "b":
            job["cov_ewma"] = torch.rand(h // ps, w // ps, generator=g)
        res = qc.compute_batch([dict(job)], ps, 0.8, 0.1, 0.26, 2.0, 1.5, 1.0, 1.0, 0.02, "cpu")[0]
        for k in ("valid_kf", "r_pix", "Ck", "Qk"):
            out[f"{name}_{k}"] = job[k].numpy()
        if "cov_ewma" in job:
            out[f"{name}_prev"] = job["cov_ewma"].numpy()
        for k in ("delta_cov", "r", "u", "class_id", "priority", "cov_ewma"):
            out[f"{name}_out_{k}"] = np.asarray(res[k])
        out[f"{name}_mean"] = qc.reduce_grid(r_pix.nan_to_num(0.0), h, w, ps, valid=valid, method="mean").numpy()
        out[f"{name}_hwps"] = np.array([h, w, ps])
    np.savez_compressed(os.path.join(HERE, "quality_core.npz"), **out, **meta())
    print("quality_core.npz", {k: v.shape for k, v in out.items() if "_out_" in k})


def section_frame():
    """Frame.update_pointmap / get_average_conf (frame.py:17-108) in every filtering mode, and the reference's default
    configuration (config/base.yaml through its own loader).  frame.py imports lietorch and mast3r_utils (neither
    importable here): both are stubbed for the duration of the section - the Frame arithmetic does not touch them."""
    import json
    import types

    saved = {k: sys.modules.get(k) for k in ("lietorch", "mast3r_slam", "mast3r_slam.config", "mast3r_slam.mast3r_utils")}
    try:
        class _Sim3:
            embedded_dim = 8

            @staticmethod
            def Identity(*a, **k):
                return None

        sys.modules["lietorch"] = types.SimpleNamespace(Sim3=_Sim3)
        ref_config = load_by_path("ref_config", f"{REF}/mast3r_slam/config.py")
        pkg = types.ModuleType("mast3r_slam")
        pkg.__path__ = []
        sys.modules["mast3r_slam"] = pkg
        sys.modules["mast3r_slam.config"] = ref_config
        sys.modules["mast3r_slam.mast3r_utils"] = types.SimpleNamespace(resize_img=None)
        cwd = os.getcwd()
        os.chdir(REF)
        try:
            ref_config.load_config("config/base.yaml")
        finally:
            os.chdir(cwd)
        base = json.loads(json.dumps(ref_config.config))          # plain dict / list / numbers
        frame = load_by_path("ref_frame", f"{REF}/mast3r_slam/frame.py")
        g = torch.Generator().manual_seed(21)
        n = 40
        Xs = [torch.randn(n, 3, generator=g) + torch.tensor([0.0, 0.0, 3.0]) for _ in range(4)]
        Cs = [torch.rand(n, 1, generator=g) * 3 for _ in range(4)]
        out = dict(X=torch.stack(Xs).numpy(), C=torch.stack(Cs).numpy())
        for mode in ("first", "recent", "best_score", "indep_conf", "weighted_pointmap", "weighted_spherical"):
            ref_config.config["tracking"]["filtering_mode"] = mode
            for score in (("median", "mean") if mode == "best_score" else ("median",)):
                ref_config.config["tracking"]["filtering_score"] = score
                f = frame.Frame(0, None, None, None, None)
                for k in range(4):
                    f.update_pointmap(Xs[k], Cs[k])
                    tag = f"{mode}_{score}_{k}"
                    out[tag + "_X"], out[tag + "_C"] = f.X_canon.numpy().copy(), f.C.numpy().copy()
                    out[tag + "_N"] = np.array([f.N, f.N_updates])
                    out[tag + "_avg"] = f.get_average_conf().numpy().copy()
        np.savez_compressed(os.path.join(HERE, "frame_update.npz"), **out, **meta())
        with open(os.path.join(HERE, "reference_base_config.json"), "w") as fh:
            json.dump(base, fh, indent=1, sort_keys=True)
        print("frame_update.npz", len(out), "arrays; reference_base_config.json", sorted(base))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def section_factor_graph():
    """FactorGraph.add_factors / prep_two_way_edges / get_unique_kf_idx (global_opt.py:12-112) on the host: the
    reference module is loaded with its unimportable dependencies stubbed (lietorch, the compiled backends) and
    mast3r_match_symmetric replaced by a function that returns seeded match tensors, so what is recorded is the
    edge-acceptance and bookkeeping logic itself."""
    import types

    names = ("lietorch", "mast3r_slam", "mast3r_slam.config", "mast3r_slam.mast3r_utils", "mast3r_slam.frame",
             "mast3r_slam.geometry", "mast3r_slam_backends")
    saved = {k: sys.modules.get(k) for k in names}
    try:
        sys.modules["lietorch"] = types.SimpleNamespace(Sim3=object)
        sys.modules["mast3r_slam_backends"] = types.SimpleNamespace()
        ref_config = load_by_path("ref_config_fg", f"{REF}/mast3r_slam/config.py")
        ref_config.config.update({"local_opt": {"Q_conf": 1.5, "window_size": 1e6}})
        pkg = types.ModuleType("mast3r_slam"); pkg.__path__ = []
        sys.modules["mast3r_slam"] = pkg
        sys.modules["mast3r_slam.config"] = ref_config
        sys.modules["mast3r_slam.frame"] = types.SimpleNamespace(SharedKeyframes=object)
        sys.modules["mast3r_slam.geometry"] = types.SimpleNamespace(constrain_points_to_ray=None)
        store = {}
        sys.modules["mast3r_slam.mast3r_utils"] = types.SimpleNamespace(
            mast3r_match_symmetric=lambda model, fi, pi, fj, pj, si, sj: store["res"])
        go = load_by_path("ref_global_opt", f"{REF}/mast3r_slam/global_opt.py")
        HWn, out = 60, {}

        class KF:
            def __init__(self):
                self.feat = torch.zeros(1, 4, 8); self.pos = torch.zeros(1, 4, 2, dtype=torch.long)
                self.img_true_shape = torch.tensor([[6, 10]])

        frames = [KF() for _ in range(8)]
        g = torch.Generator().manual_seed(33)
        fg = go.FactorGraph(None, frames, device="cpu")
        calls = [([0], [1], False), ([0, 1, 1], [2, 2, 3], False), ([0, 2], [4, 4], True), ([3], [4], False)]
        for c, (ii, jj, is_reloc) in enumerate(calls):
            b = len(ii)
            idx_i2j = torch.randint(0, HWn, (b, HWn), generator=g)
            idx_j2i = torch.randint(0, HWn, (b, HWn), generator=g)
            # per-edge valid fractions around min_match_frac = 0.3: (1,2) is consecutive and kept although poor, (1,3) is
            # rejected, the relocalisation call contains one poor edge and must add nothing and return False
            fr = torch.tensor([[0.95], [0.95, 0.1, 0.2], [0.95, 0.15], [0.95]][c])
            vj = torch.rand(b, HWn, 1, generator=g) < fr[:, None, None]
            vi = torch.rand(b, HWn, 1, generator=g) < fr[:, None, None]
            Qs = [torch.rand(b, HWn, 1, generator=g) * 4 for _ in range(4)]
            store["res"] = (idx_i2j, idx_j2i, vj, vi, *Qs)
            ret = fg.add_factors(ii, jj, 0.3, is_reloc=is_reloc)
            for k, v in zip(("idx_i2j", "idx_j2i", "vj", "vi", "Qii", "Qjj", "Qji", "Qij"), store["res"]):
                out[f"call{c}_{k}"] = v.numpy()
            out[f"call{c}_ii"], out[f"call{c}_jj"] = np.array(ii), np.array(jj)
            out[f"call{c}_reloc"], out[f"call{c}_ret"] = np.array(is_reloc), np.array(bool(ret))
            for k in ("ii", "jj", "idx_ii2jj", "idx_jj2ii", "valid_match_j", "valid_match_i", "Q_ii2jj", "Q_jj2ii"):
                out[f"state{c}_{k}"] = getattr(fg, k).numpy().copy()
        two = fg.prep_two_way_edges()
        for k, v in zip(("ii", "jj", "idx", "valid", "Q"), two):
            out[f"two_way_{k}"] = v.numpy()
        out["unique"] = fg.get_unique_kf_idx().numpy()
        np.savez_compressed(os.path.join(HERE, "factor_graph.npz"), **out, **meta())
        print("factor_graph.npz: edges kept", out["state3_ii"].tolist(), out["state3_jj"].tolist(),
              "returns", [bool(out[f"call{c}_ret"]) for c in range(4)])
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def section_track_logic():
    """FrameTracker.track (tracker.py:28-179) around the pose solver: confidence products, validity masks, the
    match-fraction gate, keyframe pointmap fusion through the solved relative pose, the new-keyframe rule.  The
    reference class runs here with mast3r_match_asymmetric and opt_pose_ray_dist_sim3 replaced by functions that
    return seeded tensors / fixed poses (the solver itself is pinned elsewhere)."""
    import types

    from mast3r_slam import synthetic as syn

    names = ("lietorch", "mast3r_slam", "mast3r_slam.config", "mast3r_slam.mast3r_utils", "mast3r_slam.frame",
             "mast3r_slam.geometry", "mast3r_slam.nonlinear_optimizer")
    saved = {k: sys.modules.get(k) for k in names}
    try:
        class _Sim3:
            embedded_dim = 8

            @staticmethod
            def Identity(*a, **k):
                return None

        sys.modules["lietorch"] = types.SimpleNamespace(Sim3=_Sim3)
        ref_config = load_by_path("ref_config_tr", f"{REF}/mast3r_slam/config.py")
        ref_config.config.update({"use_calib": False, "tracking": {
            "C_conf": 0.0, "Q_conf": 1.5, "min_match_frac": 0.05, "match_frac_thresh": 0.333, "filtering_mode": "weighted_pointmap",
            "filtering_score": "median"}})
        pkg = types.ModuleType("mast3r_slam"); pkg.__path__ = []
        sys.modules["mast3r_slam"] = pkg
        sys.modules["mast3r_slam.config"] = ref_config
        store = {}
        sys.modules["mast3r_slam.mast3r_utils"] = types.SimpleNamespace(
            resize_img=None, mast3r_match_asymmetric=lambda model, fi, fj, idx_i2j_init=None: store["match"])
        sys.modules["mast3r_slam.frame"] = load_by_path("ref_frame_tr", f"{REF}/mast3r_slam/frame.py")
        sys.modules["mast3r_slam.geometry"] = load_by_path("ref_geometry_tr", f"{REF}/mast3r_slam/geometry.py")
        sys.modules["mast3r_slam.nonlinear_optimizer"] = load_by_path("ref_nlo_tr", f"{REF}/mast3r_slam/nonlinear_optimizer.py")
        trk = load_by_path("ref_tracker", f"{REF}/mast3r_slam/tracker.py")
        Frame = sys.modules["mast3r_slam.frame"].Frame

        class Pose:
            def __init__(self, T):
                self.T = np.asarray(T, np.float64)
                self.data = torch.from_numpy(self.T.astype(np.float32)).reshape(1, 8)

            def act(self, X):
                return torch.from_numpy(syn.sim3_act(self.T, X.double().numpy()).astype(np.float32))

        class Store:
            def __init__(self, kf):
                self.kfs = [kf]

            def last_keyframe(self):
                return self.kfs[-1]

            def __len__(self):
                return len(self.kfs)

            def __setitem__(self, i, v):
                self.kfs[i] = v

        n, out = 80, {}
        T_rel = np.array([0.02, -0.01, 0.03, 0.01, 0.02, -0.01, 0.9997, 1.01]); T_rel[3:7] /= np.linalg.norm(T_rel[3:7])
        T_new = np.array([0.1, 0.2, 0.3, 0.0, 0.0, 0.0, 1.0, 1.0])
        out["T_rel"], out["T_new"] = T_rel, T_new
        g = torch.Generator().manual_seed(77)
        cases = {"normal": dict(valid_p=0.9, idx="perm"), "skipped": dict(valid_p=0.02, idx="perm"),
                 "new_kf_unique": dict(valid_p=0.9, idx="few"), "solver_fails": dict(valid_p=0.9, idx="perm", fail=True)}
        for name, c in cases.items():
            kf = Frame(0, torch.zeros(1, 3, 8, 10), None, None, None, Pose([0, 0, 0, 0, 0, 0, 1, 1]))
            kf.update_pointmap(torch.randn(n, 3, generator=g) + 3, torch.rand(n, 1, generator=g) * 2 + 0.1)
            fr = Frame(1, torch.zeros(1, 3, 8, 10), None, None, None, Pose([0, 0, 0, 0, 0, 0, 1, 1]))
            idx = torch.randperm(n, generator=g)[None] if c["idx"] == "perm" else torch.randint(0, 5, (1, n), generator=g)
            vm = (torch.rand(1, n, 1, generator=g) < c["valid_p"])
            Xff, Xkf = torch.randn(n, 3, generator=g) + 3, torch.randn(n, 3, generator=g) + 3
            Cff, Ckf = torch.rand(n, 1, generator=g) * 2 + 0.1, torch.rand(n, 1, generator=g) * 2 + 0.1
            Qff, Qkf = torch.rand(n, 1, generator=g) * 4, torch.rand(n, 1, generator=g) * 4
            store["match"] = (idx, vm, Xff, Cff, Qff, Xkf, Ckf, Qkf)
            for k, v in zip(("idx", "vm", "Xff", "Cff", "Qff", "Xkf", "Ckf", "Qkf"), store["match"]):
                out[f"{name}_{k}"] = v.numpy()
            out[f"{name}_kfX0"], out[f"{name}_kfC0"] = kf.X_canon.numpy().copy(), kf.C.numpy().copy()
            tr = trk.FrameTracker(None, Store(kf), "cpu")
            seen = {}

            def solver(Xf, Xk, T_WCf, T_WCk, Qk, valid, _c=c, _seen=seen):
                _seen["Qk"], _seen["valid"], _seen["Xf"] = Qk.numpy().copy(), valid.numpy().copy(), Xf.numpy().copy()
                if _c.get("fail"):
                    raise RuntimeError("cholesky")
                return Pose(T_new), Pose(T_rel)

            tr.opt_pose_ray_dist_sim3 = solver
            new_kf, info, skipped = tr.track(fr)
            out[f"{name}_ret"] = np.array([bool(new_kf), bool(skipped), tr.idx_f2k is None])
            for k, v in seen.items():
                out[f"{name}_seen_{k}"] = v
            if not skipped:
                for k, v in zip(("Xk", "Ck", "Xf", "Cf", "Qkf", "Qff"), info):
                    out[f"{name}_info_{k}"] = v.numpy()
                out[f"{name}_kfN"] = np.array([tr.keyframes.kfs[0].N, tr.keyframes.kfs[0].N_updates])
        np.savez_compressed(os.path.join(HERE, "track_logic.npz"), **out, **meta())
        print("track_logic.npz", {k: out[f"{k}_ret"].tolist() for k in cases})
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def fake_heads(code1, code2, H, W):
    """Deterministic stand-in for decoder + heads: two result dicts that encode which (frame, other frame, head)
    produced them.  Shared by this generator and tests/test_utils_wrappers_oracle.py."""
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    res = []
    for head in (1, 2):
        base = 100.0 * head + 10.0 * float(code1) + float(code2)
        pts = torch.stack((base + yy, base + xx * 0.5, base + yy * xx * 0.01), -1)[None]
        desc = torch.stack([base * 0.1 + yy * (k + 1) - xx for k in range(6)], -1)[None]
        res.append(dict(pts3d=pts, conf=(base + xx)[None], desc=desc, desc_conf=(base - yy)[None]))
    return res[0], res[1]


def fake_match(X11, X21, D11, D21, idx_1_to_2_init=None):
    """Stand-in for matching.match: index and validity derived from the inputs (so that argument order matters)."""
    b, h, w = X21.shape[:3]
    key = (X11.reshape(b, h * w, 3)[..., 0] * 7 + X21.reshape(b, h * w, 3)[..., 1] * 3 + D11.reshape(b, h * w, -1)[..., 0]
           + 2 * D21.reshape(b, h * w, -1)[..., 1])
    idx = (key.abs() * 13).long() % (h * w)
    if idx_1_to_2_init is not None:
        idx = (idx + idx_1_to_2_init) % (h * w)
    return idx, ((key.long() % 3) != 0)[..., None]


def section_utils_wrappers():
    """mast3r_utils.py:34-231: the inference / matching wrappers (stack orders, reshapes, batch splitting,
    downsample) of the reference module, run with its unimportable imports stubbed, a stand-in model and a
    stand-in matching.match."""
    import types

    names = ("mast3r", "mast3r.utils", "mast3r.utils.path_to_dust3r", "dust3r", "dust3r.utils", "dust3r.utils.image",
             "mast3r.model", "mast3r_slam", "mast3r_slam.retrieval_database", "mast3r_slam.config", "mast3r_slam.matching")
    saved = {k: sys.modules.get(k) for k in names}
    try:
        for n in ("mast3r", "mast3r.utils", "dust3r", "dust3r.utils", "mast3r_slam"):
            m = types.ModuleType(n); m.__path__ = []
            sys.modules[n] = m
        sys.modules["mast3r.utils.path_to_dust3r"] = types.ModuleType("mast3r.utils.path_to_dust3r")
        sys.modules["dust3r.utils.image"] = types.SimpleNamespace(ImgNorm=None)
        sys.modules["mast3r.model"] = types.SimpleNamespace(AsymmetricMASt3R=object)
        sys.modules["mast3r_slam.retrieval_database"] = types.SimpleNamespace(RetrievalDatabase=object)
        ref_config = load_by_path("ref_config_uw", f"{REF}/mast3r_slam/config.py")
        ref_config.config.update({"dataset": {"img_downsample": 1}})
        sys.modules["mast3r_slam.config"] = ref_config
        mm = types.ModuleType("mast3r_slam.matching"); mm.match = fake_match
        sys.modules["mast3r_slam.matching"] = mm
        sys.modules["mast3r_slam"].matching = mm
        mu = load_by_path("ref_mast3r_utils", f"{REF}/mast3r_slam/mast3r_utils.py")
        H, W = 8, 12

        class Model:
            def _encode_image(self, img, shape):
                code = float(img.reshape(-1)[0])
                return torch.full((1, 4, 3), code), torch.zeros(1, 4, 2, dtype=torch.long), None

            def _decoder(self, f1, p1, f2, p2):
                c1, c2 = float(f1.reshape(-1)[0]), float(f2.reshape(-1)[0])
                return [torch.tensor([c1, c2])], [torch.tensor([c1, c2])]

            def _downstream_head(self, k, toks, shape):
                c1, c2 = float(toks[0][0]), float(toks[0][1])
                return fake_heads(c1, c2, H, W)[k - 1]

        class F:
            def __init__(self, code):
                self.img = torch.full((1, 3, H, W), float(code)); self.img_true_shape = torch.tensor([[H, W]])
                self.feat = None; self.pos = None

        model, out = Model(), {}
        fa, fb, fc = F(1), F(2), F(3)
        X, C = mu.mast3r_inference_mono(model, fa)
        out["mono_X"], out["mono_C"] = X.numpy(), C.numpy()
        for k, v in zip("XCDQ", mu.mast3r_symmetric_inference(model, fa, fb)):
            out[f"sym_{k}"] = v.numpy()
        for k, v in zip("XCDQ", mu.mast3r_asymmetric_inference(model, fb, fc)):
            out[f"asym_{k}"] = v.numpy()
        feat_i, feat_j = torch.cat((fa.feat, fb.feat)), torch.cat((fb.feat, fc.feat))
        pos = torch.cat((fa.pos, fb.pos))
        shp = [fa.img_true_shape, fb.img_true_shape]
        for k, v in zip("XCDQ", mu.mast3r_decode_symmetric_batch(model, feat_i, pos, feat_j, pos, shp, shp)):
            out[f"batch_{k}"] = v.numpy()
        for k, v in enumerate(mu.mast3r_match_symmetric(model, feat_i, pos, feat_j, pos, shp, shp)):
            out[f"msym_{k}"] = v.numpy()
        init = torch.arange(H * W)[None] % 7
        for k, v in enumerate(mu.mast3r_match_asymmetric(model, fa, fc, idx_i2j_init=init)):
            out[f"masym_{k}"] = v.numpy()
        ref_config.config["dataset"]["img_downsample"] = 2
        for k, v in zip("XCDQ", mu.mast3r_asymmetric_inference(model, fb, fc)):
            out[f"asym_ds2_{k}"] = v.numpy()
        np.savez_compressed(os.path.join(HERE, "utils_wrappers.npz"), **out, **meta())
        print("utils_wrappers.npz", {k: v.shape for k, v in out.items() if k.endswith("_X") or k.startswith("msym")})
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def section_refine_block():
    """TSDFRefiner._refine_block_enhanced (tsdf_refine.py:667-835) on a one-keyframe store, identity pose: the gates
    (valid pixels, ROI, voxel count, displacement, hit ratio), the confidence boost / clamp and the version counter.
    torch.manual_seed(123) right before each call fixes the <= 100 ray-cast pixels (torch.randperm, :965)."""
    import contextlib
    import io
    import threading

    from mast3r_slam import synthetic

    tr = load_by_path("ref_tsdf_refine_rb", f"{REF}/mast3r_slam/tsdf_refine.py")
    base = dict(enabled=True, window_size=5, voxel_size=0.02, trunc_dist=0.08, max_grid_dim=64, roi_size=0.4, ray_samples=64,
                max_displacement=0.015, min_weight_threshold=0.01, confidence_boost=0.08, confidence_max=1.3, min_hit_rate=0.05,
                max_rois_per_kf=3, min_confidence=0.2)

    class Pose:
        def clone(self):
            return self

        def matrix(self):
            return torch.eye(4)[None]

        def act(self, X):
            return X.clone()

    H, W = 48, 64
    Tcam = synthetic.camera_pose(2)
    rng = np.random.default_rng(4)
    X = (synthetic.render_pointmap(Tcam, H, W).reshape(-1, 3) + rng.normal(0, 0.003, (H * W, 3))).astype(np.float32)
    C = rng.uniform(0.1, 1.2, (H * W, 1)).astype(np.float32)
    C[:16 * W] = 0.01                                             # the top band is unusable
    out = dict(X=X, C=C, H=H, W=W)

    def mask_of(y0, x0):
        m = np.zeros((H, W), bool); m[y0:y0 + 16, x0:x0 + 16] = True
        return m.reshape(-1)

    cases = {"accepted": (dict(min_hit_rate=0.02), mask_of(16, 24)), "hit_ratio_reject": (dict(), mask_of(16, 24)),
             "too_few_valid": (dict(), mask_of(0, 8))}
    for name, (over, mask) in cases.items():
        class KF:
            pass

        kf = KF()
        kf.img_shape = torch.tensor([[H, W]]); kf.X_canon = torch.from_numpy(X.copy()); kf.C = torch.from_numpy(C.copy())
        kf.K = torch.eye(3); kf.T_WC = Pose()

        class Store:
            lock = threading.RLock()
            version = torch.zeros(4, dtype=torch.long)

            def __getitem__(self, i):
                return kf

        with contextlib.redirect_stdout(io.StringIO()):
            ref = tr.TSDFRefiner(dict(base, **over), Store(), None, "cpu")
            block = tr.PatchBlock(0, 7, [], torch.from_numpy(mask), 1.0, 1.0)
            torch.manual_seed(123)
            ok, score = ref._refine_block_enhanced(block)
        out[f"{name}_mask"] = mask
        out[f"{name}_ret"] = np.array([float(ok), float(score)])
        out[f"{name}_C_after"] = kf.C.numpy().copy()
        out[f"{name}_version"] = ref.keyframes.version.numpy().copy()
        d = ref.stats["debug_info"]
        out[f"{name}_stats"] = np.array([d["tsdf_constructions"], d["surface_extractions"], d["displacement_rejects"], d["hit_ratio_rejects"]])
        out[f"{name}_min_hit_rate"] = np.array(dict(base, **over)["min_hit_rate"])
        print(name, bool(ok), round(float(score), 4), "changed C:", int((kf.C.numpy() != C).sum()), out[f"{name}_stats"].tolist())
    np.savez_compressed(os.path.join(HERE, "refine_block.npz"), **out, **meta())


def section_evaluate():
    """evaluate.py:23-106 (save_traj, save_reconstruction -> save_ply) of the reference module on three synthetic
    keyframes.  cv2 / dataloader / frame are stubbed (unused by these functions), lietorch is a data holder so that the
    reference's own lietorch_utils.as_SE3 runs, and plyfile is a stub that captures the structured vertex array the
    reference hands to PlyElement.describe (file bytes of plyfile itself: unpinned)."""
    import tempfile
    import types

    from mast3r_slam import synthetic

    names = ("lietorch", "cv2", "plyfile", "mast3r_slam", "mast3r_slam.config", "mast3r_slam.dataloader", "mast3r_slam.frame",
             "mast3r_slam.lietorch_utils", "mast3r_slam.geometry")
    saved = {k: sys.modules.get(k) for k in names}
    captured = {}
    try:
        class _G:
            def __init__(self, data):
                self.data = data

        sys.modules["lietorch"] = types.SimpleNamespace(SE3=type("SE3", (_G,), {}), Sim3=type("Sim3", (_G,), {}))
        sys.modules["cv2"] = types.SimpleNamespace()

        class _PlyElement:
            @staticmethod
            def describe(arr, name):
                captured["pcd"], captured["name"] = arr.copy(), name
                return arr

        class _PlyData:
            def __init__(self, elements, text=True):
                captured["text"] = text

            def write(self, filename):
                captured["filename"] = str(filename)

        sys.modules["plyfile"] = types.SimpleNamespace(PlyData=_PlyData, PlyElement=_PlyElement)
        ref_config = load_by_path("ref_config_ev", f"{REF}/mast3r_slam/config.py")
        cwd = os.getcwd()
        os.chdir(REF)
        try:
            ref_config.load_config("config/base.yaml")
        finally:
            os.chdir(cwd)
        pkg = types.ModuleType("mast3r_slam")
        pkg.__path__ = []
        sys.modules["mast3r_slam"] = pkg
        sys.modules["mast3r_slam.config"] = ref_config
        sys.modules["mast3r_slam.dataloader"] = types.SimpleNamespace(Intrinsics=None)
        sys.modules["mast3r_slam.frame"] = types.SimpleNamespace(SharedKeyframes=None)
        sys.modules["mast3r_slam.lietorch_utils"] = load_by_path("ref_lietorch_utils_ev", f"{REF}/mast3r_slam/lietorch_utils.py")
        sys.modules["mast3r_slam.geometry"] = types.SimpleNamespace(constrain_points_to_ray=None)
        ev = load_by_path("ref_evaluate", f"{REF}/mast3r_slam/evaluate.py")

        H, W = 24, 32
        rng = np.random.default_rng(8)

        class Pose:
            def __init__(self, T):
                self.T = T
                self.data = torch.from_numpy(T.astype(np.float32)).reshape(1, 8)

            def act(self, X):
                return torch.from_numpy(synthetic.sim3_act(self.T, X.numpy().astype(np.float64)).astype(np.float32))

        class KF:
            pass

        kfs, out = [], {}
        for i, k in enumerate((0, 9, 21)):
            T = synthetic.camera_pose(k)
            T[7] = 1.0 + 0.05 * i
            kf = KF()
            kf.frame_id = 3 * i + 1
            kf.T_WC = Pose(T)
            kf.X_canon = torch.from_numpy(synthetic.render_pointmap(synthetic.camera_pose(k), H, W).reshape(-1, 3).astype(np.float32))
            kf.uimg = torch.from_numpy(rng.uniform(0, 1, (H, W, 3)).astype(np.float32))
            C = torch.from_numpy(rng.uniform(0.5, 3.0, (H * W, 1)).astype(np.float32))
            kf.get_average_conf = (lambda c: (lambda: c))(C)
            kf.img_shape = torch.tensor([[H, W]])
            kf.K = None
            kfs.append(kf)
            out[f"T_{i}"], out[f"X_{i}"], out[f"uimg_{i}"], out[f"C_{i}"] = T.astype(np.float32), kf.X_canon.numpy(), kf.uimg.numpy(), C.numpy()
            out[f"Xw_{i}"] = kf.T_WC.act(kf.X_canon).numpy()
        timestamps = [1305031102.175304 + 0.0333 * j for j in range(8)]
        with tempfile.TemporaryDirectory() as d:
            ev.save_traj(d, "traj.txt", timestamps, kfs)
            out["traj_txt"] = np.array(open(os.path.join(d, "traj.txt")).read())
            ev.save_reconstruction(d, "rec.ply", kfs, 1.5)
        out["timestamps"] = np.array(timestamps)
        out["frame_ids"] = np.array([kf.frame_id for kf in kfs])
        pcd = captured["pcd"]
        out["ply_names"] = np.array(list(pcd.dtype.names))
        out["ply_types"] = np.array([pcd.dtype[n].str for n in pcd.dtype.names])
        for n in pcd.dtype.names:
            out["ply_" + n] = pcd[n]
        out["ply_text"] = np.array(captured["text"])
        out["c_conf_threshold"] = np.array(1.5)
        np.savez_compressed(os.path.join(HERE, "evaluate.npz"), **out, **meta())
        print("evaluate.npz:", str(out["traj_txt"]).splitlines()[0], "| ply vertices", len(pcd), captured["name"], captured["text"])
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


sys.path.insert(0, HERE)
from dataset_tree import build_dataset_tree  # noqa: E402  (tests/golden/dataset_tree.py, shared with tests/test_dataloader.py)


def section_dataloader():
    """The file-based readers of dataloader.py (TUM, 7-Scenes, RGBFiles, ETH3D, Replica; use_calib False) and load_dataset
    on a small directory tree.  cv2.imread / cvtColor are a PIL shim (same decoded bytes for PNG), natsort.natsorted is a
    digit-aware sort written here (natsort itself: unpinned), pyrealsense2 an empty stub, resize_img the mirror's (pinned
    separately by resize_img.npz)."""
    import tempfile
    import types

    import PIL.Image
    from mast3r_slam.mast3r_utils import resize_img as my_resize_img

    names = ("cv2", "natsort", "pyrealsense2", "mast3r_slam", "mast3r_slam.config", "mast3r_slam.mast3r_utils")
    saved = {k: sys.modules.get(k) for k in names}
    try:
        def imread(path, flag=1):
            im = PIL.Image.open(path)
            return np.asarray(im.convert("L")) if flag == 0 else np.asarray(im.convert("RGB"))[..., ::-1].copy()

        def cvtColor(img, code):
            return img[..., ::-1].copy() if code == "BGR2RGB" else np.stack([img] * 3, -1)

        sys.modules["cv2"] = types.SimpleNamespace(imread=imread, cvtColor=cvtColor, COLOR_BGR2RGB="BGR2RGB",
                                                   COLOR_GRAY2BGR="GRAY2BGR", IMREAD_GRAYSCALE=0)
        nkey = lambda p: [int(t) if t.isdigit() else t.lower() for t in __import__("re").split(r"(\d+)", str(p))]
        sys.modules["natsort"] = types.SimpleNamespace(natsorted=lambda seq: sorted(seq, key=nkey))
        sys.modules["pyrealsense2"] = types.SimpleNamespace()
        ref_config = load_by_path("ref_config_dl", f"{REF}/mast3r_slam/config.py")
        cwd = os.getcwd()
        os.chdir(REF)
        try:
            ref_config.load_config("config/base.yaml")
        finally:
            os.chdir(cwd)
        pkg = types.ModuleType("mast3r_slam")
        pkg.__path__ = []
        sys.modules["mast3r_slam"] = pkg
        sys.modules["mast3r_slam.config"] = ref_config
        sys.modules["mast3r_slam.mast3r_utils"] = types.SimpleNamespace(resize_img=my_resize_img)
        if not hasattr(np, "unicode_"):
            np.unicode_ = np.str_          # the reference predates NumPy 2
        dl = load_by_path("ref_dataloader", f"{REF}/mast3r_slam/dataloader.py")
        rng = np.random.default_rng(12)
        imgs = rng.integers(0, 256, (5, 48, 64, 3), dtype=np.uint8)
        out = dict(imgs=imgs)
        with tempfile.TemporaryDirectory() as root:
            layout = build_dataset_tree(root, imgs)
            for cls, rel in layout.items():
                ds = dl.load_dataset(os.path.join(root, rel))
                assert type(ds).__name__ == cls, (type(ds).__name__, cls)
                out[f"{cls}_len"] = np.array(len(ds))
                out[f"{cls}_timestamps"] = np.array([str(t) for t in ds.timestamps])
                out[f"{cls}_files"] = np.array([os.path.relpath(str(f), root) for f in ds.rgb_files])
                t0, im0 = ds[0]
                t1, im1 = ds[len(ds) - 1]
                out[f"{cls}_t0"], out[f"{cls}_img0"], out[f"{cls}_imgN"] = np.array(str(t0)), im0, im1
                shp, raw = ds.get_img_shape()
                out[f"{cls}_shape"] = np.array([*shp, *raw])
                out[f"{cls}_flags"] = np.array([ds.has_calib(), ds.use_calibration, ds.save_results])
                ds.subsample(2)
                out[f"{cls}_sub_files"] = np.array([os.path.relpath(str(f), root) for f in ds.rgb_files])
                print(cls, len(out[f"{cls}_files"]), list(out[f"{cls}_timestamps"][:2]), out[f"{cls}_shape"].tolist(), out[f"{cls}_flags"].tolist())
        out["layout"] = np.array(json_dumps(layout))
        np.savez_compressed(os.path.join(HERE, "dataloader.npz"), **out, **meta())
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def json_dumps(o):
    import json

    return json.dumps(o, sort_keys=True)


SECTIONS["dataloader"] = section_dataloader
SECTIONS["evaluate"] = section_evaluate
SECTIONS["refine_block"] = section_refine_block


def section_refine_schedule():
    """TSDFRefiner block selection and sliding-window scheduling (tsdf_refine.py:246-601) of the reference class itself
    (never started as a thread) on a store of 9 small keyframes: `_select_blocks_enhanced` on a seeded priority grid
    (with clustering on and off), `_schedule_refinement`'s confidence fallback, and the queue after every
    `maybe_schedule_sliding_window(k)` of a run plus `schedule_final_pass` (time.sleep stubbed)."""
    import contextlib
    import io
    import threading
    import time as _time

    from mast3r_slam import synthetic

    tr = load_by_path("ref_tsdf_refine_rs", f"{REF}/mast3r_slam/tsdf_refine.py")
    base = dict(enabled=True, window_size=5, voxel_size=0.02, trunc_dist=0.08, max_grid_dim=64, roi_size=0.4, ray_samples=64,
                max_displacement=0.015, min_weight_threshold=0.01, confidence_boost=0.08, confidence_max=1.3, min_hit_rate=0.05,
                max_rois_per_kf=3, min_confidence=0.2, max_pending_tasks=50)
    H, W, NKF = 96, 128, 9
    rng = np.random.default_rng(12)
    Xs, Cs = [], []
    for k in range(NKF):
        X = synthetic.render_pointmap(synthetic.camera_pose(5 * k), H, W).reshape(-1, 3).astype(np.float32)
        C = rng.uniform(0.35, 1.2, (H * W, 1)).astype(np.float32)
        lo = rng.uniform(0.06, 0.29, (H, W)).astype(np.float32)
        C2 = C.reshape(H, W).copy()
        if k != 3:                                            # keyframe 3 has no low-confidence region: scheduling fails
            y0, x0 = 16 * (k % 4), 16 * (k % 6)
            C2[y0:y0 + 32, x0:x0 + 48] = lo[y0:y0 + 32, x0:x0 + 48]
        C2[40:44, :] = 0.01
        if k in (2, 5):      # the fallback map is per PIXEL but read with patch_size 16 (tsdf_refine.py:381-392, 437-470):
            C2[0:3, 0:4] = rng.uniform(0.06, 0.08, (3, 4))     # only pixels (y < H/16, x < W/16) can name a valid patch
        Xs.append(X); Cs.append(C2.reshape(-1, 1))
    out = dict(H=H, W=W, X=np.stack(Xs), C=np.stack(Cs))

    class KF:
        pass

    kfs = []
    for k in range(NKF):
        kf = KF()
        kf.frame_id = 10 * k; kf.img_shape = torch.tensor([[H, W]])
        kf.X_canon = torch.from_numpy(Xs[k].copy()); kf.C = torch.from_numpy(Cs[k].copy())
        kfs.append(kf)

    class Store:
        lock = threading.RLock()

        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return kfs[i]

    def dump_blocks(prefix, blocks):
        out[prefix + "_n"] = np.array(len(blocks))
        for b_i, b in enumerate(blocks):
            out[f"{prefix}_{b_i}_ids"] = np.array([b.kf_id, b.block_id])
            out[f"{prefix}_{b_i}_patches"] = np.array(b.patch_indices, np.int64).reshape(-1, 2)
            out[f"{prefix}_{b_i}_mask"] = np.flatnonzero(b.pixel_mask.numpy())
            out[f"{prefix}_{b_i}_vals"] = np.array([b.depth_median, b.priority, b.depth_variance])

    prio = rng.uniform(0, 1, (H // 16, W // 16)).astype(np.float32)
    prio[2, 3], prio[2, 4], prio[3, 4] = 0.99, 0.98, 0.97      # three adjacent top patches
    out["priority"] = prio
    real_sleep = _time.sleep
    tr.time.sleep = lambda s: None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            for name, over in (("select_single", {}), ("select_cluster", dict(max_block_edge=2, z_rel=0.5))):
                ref = tr.TSDFRefiner(dict(base, **over), Store(NKF), None, "cpu")
                dump_blocks(name, ref._select_blocks_enhanced(4, {"priority": torch.from_numpy(prio), "patch_size": 16}))
            ref = tr.TSDFRefiner(dict(base), Store(NKF), None, "cpu")
            ok = ref._schedule_refinement(2)
            items = list(ref.queue.queue)
            out["fallback_ok"] = np.array(ok)
            dump_blocks("fallback", [b for _, b in items])
            out["fallback_fail_ok"] = np.array(ref._schedule_refinement(3))
            ok5 = ref._schedule_refinement(5)
            out["fallback5_ok"] = np.array(ok5)
            dump_blocks("fallback5", [b for _, b in list(ref.queue.queue)[len(items):]])

            class FakeQuality:     # AsynchronousQualityService surface used by the refiner: get(frame_id), poll()
                def get(self, frame_id):
                    if frame_id == 30:
                        return None          # keyframe 3: no result yet -> confidence fallback -> fails -> retried
                    g = np.random.default_rng(frame_id).uniform(0, 1, (H // 16, W // 16)).astype(np.float32)
                    return {"priority": g, "patch_size": 16, "kf_id": frame_id}

                def poll(self):
                    pass

            out["quality_grids"] = np.stack([np.random.default_rng(10 * k).uniform(0, 1, (H // 16, W // 16)).astype(np.float32)
                                             for k in range(NKF)])
            # a run: keyframes arrive one by one (the store grows), then the final pass
            store = Store(0)
            ref = tr.TSDFRefiner(dict(base), store, FakeQuality(), "cpu")
            trace = []
            for cur in range(NKF):
                store.n = cur + 1
                ref.maybe_schedule_sliding_window(cur)
                trace.append([[k.kf_id, k.block_id] for k, _ in ref.queue.queue])
            ref.schedule_final_pass(NKF - 1)
            trace.append([[k.kf_id, k.block_id] for k, _ in ref.queue.queue])
    finally:
        tr.time.sleep = real_sleep
    for i, t in enumerate(trace):
        out[f"trace_{i}"] = np.array(t, np.int64).reshape(-1, 2)
    out["trace_n"] = np.array(len(trace))
    print("refine_schedule", {k: (v.tolist() if v.size < 24 else v.shape) for k, v in out.items() if k.startswith(("trace", "select_single_0", "fallback_ok", "fallback_n", "fallback_fail", "fallback5_n", "fallback5_0_p"))})
    np.savez_compressed(os.path.join(HERE, "refine_schedule.npz"), **out, **meta())


SECTIONS["refine_schedule"] = section_refine_schedule


def section_retrieval_quantize():
    """RetrievalDatabase.quantize_custom (retrieval_database.py:96-105): the method's own source (taken from the file's
    AST: the module cannot be imported without asmk) run on a seeded random codebook (65 536 x 1024) and 768 seeded
    features, multiple_assignment 5 (query) and 1 (build).  Only the indices and the seeds are stored; the test
    regenerates the inputs."""
    import ast
    import textwrap

    src = open(f"{REF}/mast3r_slam/retrieval_database.py").read()
    fn = [n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.FunctionDef) and n.name == "quantize_custom"][0]
    code = textwrap.dedent(ast.get_source_segment(src, fn))
    ns = {"torch": torch}
    exec(code, ns)

    class Self:
        pass

    g = torch.Generator().manual_seed(1234)
    me = Self()
    me.centroids = torch.randn(65536, 1024, generator=g)
    q = torch.randn(768, 1024, generator=g)
    q[:64] = me.centroids[1000:1064] + 0.05 * torch.randn(64, 1024, generator=g)      # some features sit near a centroid
    out = {}
    for name, k in (("query", 5), ("build", 1)):
        out[name] = ns["quantize_custom"](me, q, {"quantize": {"multiple_assignment": k}}).numpy()
    d = torch.cdist(q[:8], me.centroids)
    out["gap_check"] = torch.topk(d, 6, dim=1, largest=False).values.numpy()
    print("retrieval_quantize", out["query"][:3], out["query"][64:66])
    np.savez_compressed(os.path.join(HERE, "retrieval_quantize.npz"), seed=np.array(1234), **out, **meta())


SECTIONS["retrieval_quantize"] = section_retrieval_quantize

def _ref_asmk_modules():
    """The reference's own asmk modules (thirdparty/mast3r/asmk/asmk/{kernel,inverted_file,functional,io_helpers}.py) with
    its Cython extension built by oracle/Makefile's `ref` target into oracle/_ref/asmk_ext (asmk/__init__.py itself pulls
    faiss through asmk_method -> index, so the package object is made here and the four modules are loaded under it)."""
    import types

    ext = os.path.join(ROOT, "oracle", "_ref", "asmk_ext")
    sys.path.insert(0, ext)
    import hamming

    pkg = types.ModuleType("asmk")
    pkg.__path__ = [f"{REF}/thirdparty/mast3r/asmk/asmk"]
    sys.modules["asmk"] = pkg
    sys.modules["asmk.hamming"] = hamming
    pkg.hamming = hamming
    from asmk import functional, inverted_file, io_helpers, kernel

    return hamming, kernel, inverted_file, functional, io_helpers


def section_retrieval_asmk():
    """RetrievalDatabase.update / query / add_to_database / accumulate_scores / add_to_ivf_custom / prep_features
    (retrieval_database.py:24-166; the methods' own source from the file's AST - the module needs faiss and torchvision to
    import) driving the reference's ASMKKernel + IVF + compiled hamming extension, with Whitener / how_select_local taken
    the same way from thirdparty/mast3r/mast3r/retrieval/model.py.  Small seeded sizes: backbone dim 32, 48 tokens,
    descriptor dim 64, 24 local features per image, 512 centroids, 12 images of which three revisit earlier ones."""
    import ast
    import textwrap
    import types

    hamming, kernel, inverted_file, functional, io_helpers = _ref_asmk_modules()
    msrc = open(f"{REF}/thirdparty/mast3r/mast3r/retrieval/model.py").read()
    mtree = ast.parse(msrc)
    ns_model = {"torch": torch, "nn": torch.nn, "np": np}
    for node in mtree.body:
        if (isinstance(node, ast.ClassDef) and node.name == "Whitener") or \
           (isinstance(node, ast.FunctionDef) and node.name == "how_select_local"):
            exec(ast.get_source_segment(msrc, node), ns_model)
    Whitener, how_select_local = ns_model["Whitener"], ns_model["how_select_local"]

    src = open(f"{REF}/mast3r_slam/retrieval_database.py").read()
    cls = [n for n in ast.walk(ast.parse(src)) if isinstance(n, ast.ClassDef) and n.name == "RetrievalDatabase"][0]
    ns = {"torch": torch, "np": np, "io_helpers": io_helpers, "how_select_local": how_select_local}
    body = "\n".join(textwrap.dedent(ast.get_source_segment(src, f)) for f in cls.body
                     if isinstance(f, ast.FunctionDef) and f.name != "__init__")
    exec(body, ns)
    RefDB = type("RefDB", (), {k: v for k, v in ns.items() if isinstance(v, types.FunctionType) and k != "how_select_local"})

    g = torch.Generator().manual_seed(77)
    BD, NT, D, NF, K, NIMG = 32, 48, 64, 24, 512, 12
    model = types.SimpleNamespace()
    model.prewhiten = Whitener(BD)
    model.postwhiten = Whitener(D)
    with torch.no_grad():
        model.prewhiten.m.copy_(0.1 * torch.randn(1, BD, generator=g, dtype=torch.float64))
        model.prewhiten.p.copy_(torch.eye(BD, dtype=torch.float64) + 0.2 * torch.randn(BD, BD, generator=g, dtype=torch.float64))
        model.postwhiten.m.copy_(0.1 * torch.randn(1, D, generator=g, dtype=torch.float64))
        model.postwhiten.p.copy_(torch.eye(D, dtype=torch.float64) + 0.2 * torch.randn(D, D, generator=g, dtype=torch.float64))
    lin = torch.nn.Linear(BD, D)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(D, BD, generator=g) / BD ** 0.5)
        lin.bias.copy_(0.05 * torch.randn(D, generator=g))
    model.projector = torch.nn.Sequential(lin)         # build_projector(hdims=[D]) (model.py:139-151)
    model.residual = False
    model.attention = lambda x: x.norm(dim=-1)         # featweights == 'l2norm' (model.py:131-132)
    model.nfeat = NF
    centroids = torch.randn(K, D, generator=g).numpy().astype(np.float32)
    params = {"build_ivf": {"kernel": {"binary": True}, "ivf": {"use_idf": False},          # processor.py:84-89
                            "quantize": {"multiple_assignment": 1}, "aggregate": {}},
              "query_ivf": {"quantize": {"multiple_assignment": 5}, "aggregate": {}, "search": {"topk": None},
                            "similarity": {"similarity_threshold": 0.0, "alpha": 3.0}}}
    codebook = types.SimpleNamespace(centroids=centroids, size=K)
    kern = kernel.ASMKKernel(codebook, binary=True)
    ivf = inverted_file.IVF.initialize_empty(use_idf=False, codebook_size=K)
    me = RefDB()
    me.model = model
    me.asmk = types.SimpleNamespace(params=params, codebook=codebook)
    me.ivf_builder = types.SimpleNamespace(kernel=kern, ivf=ivf, step_params=params["build_ivf"])
    me.kf_counter, me.kf_ids = 0, []
    me.query_dtype, me.query_device = torch.float32, "cpu"
    me.centroids = torch.from_numpy(centroids)

    feats = torch.randn(NIMG, 1, NT, BD, generator=g)
    for new, old in ((5, 1), (8, 2), (11, 5)):           # revisits: an earlier frame's tokens, slightly changed
        feats[new] = feats[old] + 0.05 * torch.randn(1, NT, BD, generator=g)
    seen = {}
    orig_query = RefDB.query

    def spy_query(self, feat, id):
        ranks, scores, topk = orig_query(self, feat, id)
        seen["ranks"], seen["scores"], seen["topk"] = ranks.copy(), scores.copy(), topk.copy()
        return ranks, scores, topk

    RefDB.query = spy_query
    out = {}
    with torch.no_grad():
        for i in range(NIMG):
            frame = types.SimpleNamespace(feat=feats[i])
            out[f"local_{i}"] = me.prep_features(frame.feat)[0].numpy()
            inds = me.update(frame, add_after_query=True, k=3, min_thresh=0.005)
            out[f"inds_{i}"] = np.array(inds, dtype=np.int64)
            if i > 0:
                sc = np.empty_like(seen["scores"])
                sc[np.arange(sc.shape[0])[:, None], seen["ranks"]] = seen["scores"]
                out[f"scores_{i}"], out[f"topk_{i}"] = sc[0], seen["topk"]
        probe = types.SimpleNamespace(feat=feats[2] + 0.02 * torch.randn(1, NT, BD, generator=g))
        out["probe_feat"] = probe.feat.numpy()
        out["probe_inds"] = np.array(me.update(probe, add_after_query=False, k=4, min_thresh=0.0), dtype=np.int64)   # relocalisation use
        sc = np.empty_like(seen["scores"])
        sc[np.arange(sc.shape[0])[:, None], seen["ranks"]] = seen["scores"]
        out["probe_scores"] = sc[0]
    assert me.kf_counter == NIMG and ivf.n_images == NIMG
    words = np.array([w for w in range(K) if ivf.counts[w] > 0], dtype=np.int64)
    out["ivf_words"] = np.concatenate([np.full(ivf.counts[w], w) for w in words])
    out["ivf_imids"] = np.concatenate([ivf.ivf_image_ids[w][:ivf.counts[w]] for w in words])
    out["ivf_vecs"] = np.concatenate([ivf.ivf_vecs[w][:ivf.counts[w]] for w in words])
    out["norm_factor"] = np.asarray(ivf.norm_factor, dtype=np.float64)
    print("retrieval_asmk", [out[f"inds_{i}"].tolist() for i in range(NIMG)], out["probe_inds"], out["ivf_vecs"].shape)
    np.savez_compressed(os.path.join(HERE, "retrieval_asmk.npz"), feats=feats.numpy(), centroids=centroids,
                        pre_m=model.prewhiten.m.detach().numpy(), pre_p=model.prewhiten.p.detach().numpy(),
                        post_m=model.postwhiten.m.detach().numpy(), post_p=model.postwhiten.p.detach().numpy(),
                        proj_w=lin.weight.detach().numpy(), proj_b=lin.bias.detach().numpy(), nfeat=np.array(NF), **out, **meta())
    # the known answers the reference's own hamming tests / docstrings hold (asmk/test/test_hamming.py, hamming.pyx:60,88,117,135)
    r = np.random.default_rng(5)
    kat = {}
    for d1 in (1, 7, 31, 32, 33, 64, 100, 128, 139):
        a = (r.random((10, d1)) - 0.5).astype(np.float32)
        b = (r.random((10, d1)) - 0.5).astype(np.float32)
        kat[f"a_{d1}"], kat[f"b_{d1}"] = a, b
        kat[f"pack_a_{d1}"] = hamming.binarize_and_pack_2D(a)
        kat[f"cdist_{d1}"] = hamming.hamming_cdist_packed(hamming.binarize_and_pack_2D(a), hamming.binarize_and_pack_2D(b), d1)
    np.savez_compressed(os.path.join(HERE, "asmk_hamming.npz"), **kat, **meta())


SECTIONS["retrieval_asmk"] = section_retrieval_asmk
SECTIONS["utils_wrappers"] = section_utils_wrappers
SECTIONS["track_logic"] = section_track_logic
SECTIONS["factor_graph"] = section_factor_graph
SECTIONS["frame"] = section_frame
SECTIONS["quality"] = section_quality
SECTIONS["geometry"] = section_geometry
SECTIONS["resize"] = section_resize


if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
