"""Mirror of the compute part of mast3r_slam/tsdf_refine.py (TSDFRefiner): local dense-block TSDF build,
ray-cast surface extraction and the block refinement decision (lines 667-1064), same method names and
config keys.  The two python double loops run as HIP kernels (csrc/tsdf_local.hip).  The thread, queue,
retry registry and quality-service plumbing of the reference class (lines 33-665) are out of scope
("scheduling/threads OUT OF SCOPE"); `refine_block` below is called synchronously."""
import dataclasses

import torch

import mslam_hip as _m
from mast3r_slam.config import config


@dataclasses.dataclass
class PatchBlock:
    kf_id: int
    block_id: int
    patch_indices: list        # [(py, px)] 16x16 patch coordinates
    pixel_mask: torch.Tensor   # (HW,) bool
    depth_median: float
    priority: float
    depth_variance: float = 0.0


class TSDFRefiner:
    def __init__(self, cfg=None, shared_keyframes=None, quality_service=None, device="cuda"):
        self.cfg = dict(config["tsdf_refine"]) if cfg is None else dict(cfg)
        for k in ("voxel_size", "trunc_dist", "max_grid_dim", "roi_size"):
            if k not in self.cfg:
                raise ValueError(f"Missing required TSDF config parameter: {k}")   # tsdf_refine.py:146-150
        self.device = torch.device(device)
        self.keyframes = shared_keyframes
        self.versions = {}
        self._ws = None
        self.stats = {"total_blocks": 0, "successful_blocks": 0,
                      "debug_info": {"tsdf_constructions": 0, "surface_extractions": 0, "displacement_rejects": 0,
                                     "hit_ratio_rejects": 0}}

    # ------------------------------------------------------------------
    def _grid_dims(self, xyz_min, xyz_max):
        roi = xyz_max - xyz_min
        dims = torch.clamp((roi / self.cfg["voxel_size"]).ceil().long(), max=self.cfg["max_grid_dim"])
        return [int(v) for v in dims.detach().cpu().tolist()]   # the one host sync the reference also has (:846)

    def _build_tsdf_robust(self, X_canon, C_flat, K, xyz_min, xyz_max, H, W, T_WC):
        """tsdf_refine.py:837-940 -> (tsdf, weights) f32 [nz,ny,nx]."""
        nx, ny, nz = self._grid_dims(xyz_min, xyz_max)
        if C_flat.ndim > 1:
            C_flat = C_flat.squeeze(-1)
        X_world = T_WC.act(X_canon.contiguous()).contiguous()
        origin = T_WC.data.reshape(-1, 8)[0, :3].contiguous()     # matrix()[:3,3] == translation
        n = X_world.shape[0]
        tsdf = torch.empty((nz, ny, nx), dtype=torch.float32, device=self.device)
        weights = torch.empty_like(tsdf)
        L = _m.lib()
        need = L.mslam_tsdf_local_workspace_bytes(n)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        C32 = C_flat.float().contiguous()
        mn, mx = xyz_min.float().contiguous(), xyz_max.float().contiguous()
        rc = L.mslam_tsdf_local_build(
            _m.ptr(X_world), _m.ptr(C32), _m.ptr(origin), _m.ptr(mn), _m.ptr(mx), n, nx, ny, nz,
            float(self.cfg["voxel_size"]), float(self.cfg["trunc_dist"]), float(self.cfg["min_confidence"]),
            _m.ptr(tsdf), _m.ptr(weights), _m.ptr(self._ws), self._ws.numel(), _m.stream_ptr())
        _m.check(rc, "tsdf_local_build")
        return tsdf, weights

    def _extract_surface_safe(self, tsdf, xyz_min, xyz_max, K, mask, H, W, X_original, order=None):
        """tsdf_refine.py:942-1021 -> (X_refined (HW,3), hits (n_mask,) bool).  `order` = positions into the
        masked pixel list to march (default: torch.randperm(n_mask)[:100] like the reference, :965)."""
        X_original = X_original.contiguous()
        X_refined = X_original.clone()
        pixel_indices = torch.where(mask)[0]
        n_mask = int(pixel_indices.numel())
        hits = torch.zeros(n_mask, dtype=torch.bool, device=self.device)
        if n_mask == 0:
            return X_refined, hits
        if order is None:
            order = torch.randperm(n_mask, device=self.device)[: min(n_mask, 100)]
        order = order.to(self.device)
        sel_pix = pixel_indices[order].contiguous()
        n_sel = int(sel_pix.numel())
        nz, ny, nx = tsdf.shape
        surf = torch.empty((n_sel, 3), dtype=torch.float32, device=self.device)
        hit = torch.zeros(n_sel, dtype=torch.uint8, device=self.device)
        mn, mx = xyz_min.float().contiguous(), xyz_max.float().contiguous()
        rc = _m.lib().mslam_tsdf_local_raycast(
            _m.ptr(tsdf.contiguous()), nx, ny, nz, _m.ptr(mn), _m.ptr(mx), _m.ptr(X_original), _m.ptr(sel_pix), n_sel,
            int(self.cfg["ray_samples"]), float(self.cfg["max_displacement"]), _m.ptr(surf), _m.ptr(hit), _m.stream_ptr())
        _m.check(rc, "tsdf_local_raycast")
        hb = hit.bool()
        X_refined[sel_pix[hb]] = surf[hb]
        hits[order[hb]] = True
        return X_refined, hits

    # ------------------------------------------------------------------
    def _refine_block_enhanced(self, block: PatchBlock, order=None):
        """tsdf_refine.py:667-835 (decision logic unchanged): returns (success, score)."""
        cfg = self.cfg
        kf = self.keyframes[block.kf_id]
        start_version = self._version(block.kf_id)
        H, W = int(kf.img_shape[0, 0]), int(kf.img_shape[0, 1])
        X_canon = kf.X_canon.clone().to(self.device)
        C_flat = (kf.C.clone() if kf.C.ndim == 1 else kf.C[..., 0].clone()).to(self.device)
        mask = block.pixel_mask.to(self.device)
        X_block, C_block = X_canon[mask], C_flat[mask]
        valid = (C_block > 0.03) & torch.isfinite(X_block).all(dim=1) & (X_block[:, 2] > 0.03)
        valid_count, total_pixels = int(valid.sum()), int(mask.sum())
        if valid_count < max(3, total_pixels * 0.05):
            return False, 0.0
        X_valid = X_block[valid]
        roi_margin = max(0.01, min(0.05, 0.1 * float(cfg.get("roi_size", 0.4))))
        xyz_min = X_valid.min(dim=0)[0] - roi_margin
        xyz_max = X_valid.max(dim=0)[0] + roi_margin
        roi_size = xyz_max - xyz_min
        if torch.any(roi_size <= 0) or torch.any(roi_size > 15.0):
            return False, 0.0
        tsdf, w = self._build_tsdf_robust(X_canon, C_flat, kf.K, xyz_min, xyz_max, H, W, kf.T_WC)
        self.stats["debug_info"]["tsdf_constructions"] += 1
        if int((w > float(cfg.get("min_weight_threshold", 0.01))).sum()) < 3:
            return False, 0.0
        X_refined, hits = self._extract_surface_safe(tsdf, xyz_min, xyz_max, kf.K, mask, H, W, X_canon, order=order)
        self.stats["debug_info"]["surface_extractions"] += 1
        hit_count = int(hits.sum())
        hit_ratio = hit_count / max(1, total_pixels)
        geometric_gain = 0.0
        mask_hits = mask.clone()
        mask_hits[mask] = hits
        if hit_count > 0:
            disp = (X_refined[mask_hits] - X_canon[mask_hits]).norm(dim=1)
            geometric_gain = float(disp.mean())
            if float(disp.max()) > float(cfg.get("max_displacement", 0.015)):
                self.stats["debug_info"]["displacement_rejects"] += 1
                return False, hit_ratio
        if not (hit_ratio >= float(cfg.get("min_hit_rate", 0.05)) and hit_count >= 1):
            self.stats["debug_info"]["hit_ratio_rejects"] += 1
            return False, hit_ratio
        if self._version(block.kf_id) != start_version:   # optimistic version check (:790-794)
            return False, hit_ratio
        boost, cmax = float(cfg.get("confidence_boost", 0.08)), float(cfg.get("confidence_max", 1.3))
        if kf.C.ndim == 2:
            kf.C[mask_hits, 0] += boost
        else:
            kf.C[mask_hits] += boost
        kf.C.clamp_(max=cmax)
        gw = float(cfg.get("geometric_weight", 0.0))
        if gw > 0:
            kf.X_canon[mask_hits] = ((1 - gw) * X_canon + gw * X_refined)[mask_hits]
        self._bump_version(block.kf_id, start_version + 1)
        return True, max(hit_ratio, geometric_gain)

    # per-keyframe version counter: SharedKeyframes.version (frame.py:251) when the store has one, else a dict
    def _version(self, kf_id):
        v = getattr(self.keyframes, "version", None)
        return int(v[kf_id]) if v is not None else self.versions.get(kf_id, 0)

    def _bump_version(self, kf_id, value):
        v = getattr(self.keyframes, "version", None)
        if v is not None:
            v[kf_id] = value
        self.versions[kf_id] = value

    refine_block = _refine_block_enhanced
