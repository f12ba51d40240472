"""Mirror of mast3r_slam/tsdf_refine.py (TSDFRefiner): local dense-block TSDF build, ray-cast surface extraction and
the block refinement decision (lines 667-1064), block selection / clustering (lines 431-601) and the sliding-window
scheduling surface main.py drives (lines 246-429: `.start/.queue/.stats/.stop_flag/.registry/
.maybe_schedule_sliding_window/.schedule_final_pass`), same method names and config keys.  The two python double
loops run as HIP kernels (csrc/tsdf_local.hip).  The reference class is a daemon thread that drains `.queue`; here the
same loop body is `process_queue()`, called synchronously by the owner (SlamSystem's backend) - no thread, no sleeps,
no time-based retry back-off (a deferred keyframe is retried at the next call)."""
import contextlib
import dataclasses
import math
import queue
import threading

import numpy as np
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


@dataclasses.dataclass(frozen=True)
class BlockKey:
    kf_id: int
    block_id: int


class RefineRegistry:
    """tsdf_refine.py:50-137: per-block state machine (IDLE -> QUEUED -> RUNNING -> COOLDOWN) and history."""
    IDLE, QUEUED, RUNNING, COOLDOWN = 0, 1, 2, 3

    def __init__(self, cfg):
        self.lock = threading.Lock()
        self.state, self.history = {}, {}
        self.cooldown_frames = cfg["cooldown_frames"]
        self.frame_count = 0
        self.success_rate = 0.0

    def tick(self):
        with self.lock:
            self.frame_count += 1

    def get_stats(self):
        with self.lock:
            attempts = sum(h.get("attempts", 0) for h in self.history.values())
            successes = sum(h.get("successes", 0) for h in self.history.values())
            return {"active_blocks": len([k for k, s in self.state.items() if s != self.IDLE]),
                    "total_attempts": attempts, "success_rate": successes / max(1, attempts),
                    "frame_count": self.frame_count}

    def try_enqueue(self, key):
        with self.lock:
            if self.state.get(key, self.IDLE) != self.IDLE:
                return False
            self.state[key] = self.QUEUED
            return True

    def begin_run(self, key):
        with self.lock:
            if self.state.get(key) != self.QUEUED:
                return False
            self.state[key] = self.RUNNING
            return True

    def finish_run(self, key, success, gain):
        with self.lock:
            hist = self.history.setdefault(key, {"attempts": 0, "successes": 0, "last_gain": 0.0, "best_gain": 0.0})
            hist["attempts"] += 1
            if success:
                hist["successes"] += 1
            hist["last_gain"] = gain
            hist["best_gain"] = max(hist["best_gain"], gain)
            hist["last_frame"] = self.frame_count
            self.state[key] = self.COOLDOWN
            attempts = sum(h.get("attempts", 0) for h in self.history.values())
            successes = sum(h.get("successes", 0) for h in self.history.values())
            self.success_rate = successes / max(1, attempts)

    def cooldown_expired(self, key):
        with self.lock:
            hist = self.history.get(key)
            if hist and self.frame_count - hist["last_frame"] > self.cooldown_frames:
                self.state[key] = self.IDLE
                return True
            return False


class TSDFRefiner:
    def __init__(self, cfg=None, shared_keyframes=None, quality_service=None, device="cuda"):
        self.cfg = dict(config["tsdf_refine"]) if cfg is None else dict(cfg)
        for k in ("voxel_size", "trunc_dist", "max_grid_dim", "roi_size"):
            if k not in self.cfg:
                raise ValueError(f"Missing required TSDF config parameter: {k}")   # tsdf_refine.py:146-150
        self.device = torch.device(device)
        self.keyframes = shared_keyframes
        self.quality_service = quality_service
        self.edit_section = contextlib.nullcontext     # context manager factory around the in-place edit of a keyframe
        self.versions = {}
        self._ws = None
        self.registry = RefineRegistry({"cooldown_frames": int(self.cfg.get("cooldown_frames", 35))})
        self.queue = queue.Queue(maxsize=int(self.cfg.get("max_pending_tasks", 50)))
        self.stop_flag = threading.Event()
        self._pending_map = {}
        self.stats = {"total_blocks": 0, "successful_blocks": 0, "total_processing_time": 0.0,
                      "debug_info": {"tsdf_constructions": 0, "surface_extractions": 0, "displacement_rejects": 0,
                                     "hit_ratio_rejects": 0}}

    # ------------------------------------------------------------------ thread surface of the reference class
    def start(self):
        """The reference starts the worker thread here (main.py:277); the synchronous form has nothing to start."""

    def is_alive(self):
        return not self.stop_flag.is_set()

    def join(self, timeout=None):
        self.process_queue()

    # ------------------------------------------------------------------ scheduling (tsdf_refine.py:246-429)
    def schedule_final_pass(self, final_kf_id):
        """tsdf_refine.py:246-258 (the 1 s sleep between the two passes only waits for the worker thread)."""
        self.maybe_schedule_sliding_window(final_kf_id, is_final_pass=True)
        self.maybe_schedule_sliding_window(final_kf_id, is_final_pass=True)

    def maybe_schedule_sliding_window(self, current_kf_id, is_final_pass=False):
        """tsdf_refine.py:260-346: keyframe current - window_size is scheduled once it leaves the window; a keyframe
        that could not be scheduled is retried (<= max_retry_attempts_per_kf) at later calls; the final pass
        schedules the keyframes still inside the window.  Retry order: fewest attempts first."""
        if not self.cfg["enabled"]:
            return
        window = int(self.cfg.get("window_size", 3))
        retry_slack = int(self.cfg.get("retry_slack_frames", 2))
        max_pending = int(self.cfg.get("max_pending_kf", 64))
        max_attempts = int(self.cfg.get("max_retry_attempts_per_kf", 3))
        min_keep_id = max(0, current_kf_id - window - retry_slack)
        for k in [k for k in self._pending_map if k < min_keep_id]:
            self._pending_map.pop(k, None)
        if is_final_pass:
            for kf_id in range(max(0, current_kf_id - window + 1), current_kf_id + 1):
                if kf_id in self._pending_map:
                    continue
                if not self._schedule_refinement(kf_id):
                    self._pending_map[kf_id] = {"attempts": 1, "final_pass": True}
            return
        due = [(k, v) for k, v in self._pending_map.items() if k >= min_keep_id and not v.get("final_pass", False)]
        if current_kf_id >= window:
            target = current_kf_id - window
            if not self._schedule_refinement(target):
                ent = self._pending_map.get(target, {"attempts": 0})
                ent["attempts"] += 1
                if ent["attempts"] <= max_attempts:
                    self._pending_map[target] = ent
                else:
                    self._pending_map.pop(target, None)
        due.sort(key=lambda x: (x[1].get("attempts", 0), -x[1].get("best_gain", 0.0)))
        for k, ent in due[:3]:
            if self._schedule_refinement(k):
                self._pending_map.pop(k, None)
            else:
                ent["attempts"] += 1
                if ent["attempts"] <= max_attempts:
                    self._pending_map[k] = ent
                else:
                    self._pending_map.pop(k, None)
        if len(self._pending_map) > max_pending:
            for k in sorted(self._pending_map.keys())[:len(self._pending_map) - max_pending]:
                self._pending_map.pop(k, None)

    def _schedule_refinement(self, kf_id):
        """tsdf_refine.py:348-429: priority map from the quality service when it has one for the keyframe, else the
        confidence heuristic (0.05 < C < 0.3 => priority 0.3 - C, normalised); top blocks enqueued."""
        quality_result = None
        if self.quality_service is not None and kf_id < len(self.keyframes):
            quality_result = self.quality_service.get(self.keyframes[kf_id].frame_id)
        if quality_result is None:
            if kf_id >= len(self.keyframes):
                return False
            kf = self.keyframes[kf_id]
            H, W = int(kf.img_shape[0, 0]), int(kf.img_shape[0, 1])
            C = kf.C.reshape(H, W) if kf.C.ndim == 2 else kf.C.reshape(H, W, 1)[..., 0]
            low_conf_mask = (C < 0.3) & (C > 0.05)
            if int(low_conf_mask.sum()) < 100:
                return False
            priority = (0.3 - C) * low_conf_mask.float()
            priority = priority / (priority.max() + 1e-8)
            quality_result = {"priority": priority.cpu(), "patch_size": 16}
        blocks = self._select_blocks_enhanced(kf_id, quality_result)
        if len(blocks) == 0:
            return True
        blocks.sort(key=lambda b: b.priority, reverse=True)
        for block in blocks[:int(self.cfg.get("max_rois_per_kf", 3))]:
            key = BlockKey(block.kf_id, block.block_id)
            if self.registry.try_enqueue(key):
                try:
                    self.queue.put_nowait((key, block))
                except queue.Full:
                    break
        return True

    def _select_blocks_enhanced(self, kf_id, quality_result):
        """tsdf_refine.py:431-517: top-5 % patches of the priority grid, the max_rois_per_kf best of them, per-patch
        depth median / variance over pixels with C > 0.05 (> 2 such pixels), then clustering."""
        priority = quality_result.get("priority")
        if priority is None:
            return []
        if isinstance(priority, np.ndarray):
            priority = torch.from_numpy(priority)
        elif isinstance(priority, list):
            priority = torch.tensor(priority, dtype=torch.float32)
        patch_size = quality_result.get("patch_size", 16)
        if priority.ndim != 2:
            return []
        Gh, Gw = priority.shape
        flat_priority = priority.flatten()
        if flat_priority.numel() == 0:
            return []
        threshold = torch.quantile(flat_priority, 0.95)
        candidate_indices = torch.where(flat_priority >= threshold)[0]
        K = min(int(self.cfg.get("max_rois_per_kf", 3)), len(candidate_indices))
        if K == 0:
            return []
        topk_values, relative_indices = torch.topk(flat_priority[candidate_indices], k=K)
        topk_indices = candidate_indices[relative_indices]
        patch_coords = [(int(idx // Gw), int(idx % Gw)) for idx in topk_indices]
        patch_priorities = [float(v) for v in topk_values]
        kf = self.keyframes[min(kf_id, len(self.keyframes) - 1)]
        H, W = int(kf.img_shape[0, 0]), int(kf.img_shape[0, 1])
        X_canon = kf.X_canon.reshape(H, W, 3)
        C = kf.C.reshape(H, W) if kf.C.ndim == 2 else kf.C.reshape(H, W, 1)[..., 0]
        # one device pass + one D2H read for all K patches (the reference calls .item() twice per patch)
        stats = []
        for gh, gw in patch_coords:
            y0, y1 = gh * patch_size, min((gh + 1) * patch_size, H)
            x0, x1 = gw * patch_size, min((gw + 1) * patch_size, W)
            valid_z = X_canon[y0:y1, x0:x1, 2][C[y0:y1, x0:x1] > 0.05]
            n = valid_z.numel()        # shape of a masked select: the reference syncs here too
            if n > 2:
                stats.append(torch.stack((torch.median(valid_z), torch.var(valid_z))))
            else:
                stats.append(None)
        have = [s for s in stats if s is not None]
        vals = torch.stack(have).cpu().tolist() if have else []
        valid_coords, valid_depths, valid_variances, valid_priorities = [], [], [], []
        it = iter(vals)
        for i, s_ in enumerate(stats):
            if s_ is None:
                continue
            med, var = next(it)
            valid_coords.append(patch_coords[i]); valid_depths.append(med); valid_variances.append(var)
            valid_priorities.append(patch_priorities[i])
        if not valid_coords:
            return []
        return self._cluster_patches_enhanced(kf_id, valid_coords, valid_depths, valid_variances, valid_priorities, H, W,
                                              patch_size)

    def _cluster_patches_enhanced(self, kf_id, patch_coords, patch_depths, patch_variances, patch_priorities, H, W,
                                  patch_size):
        """tsdf_refine.py:519-601: greedy 8-connected clustering by depth similarity, <= max_block_edge^2 patches
        per block, seeds in order of priority."""
        blocks = []
        used = [False] * len(patch_coords)
        block_id = 0
        z_abs = float(self.cfg.get("z_abs_m", 1e9))
        z_rel = float(self.cfg.get("z_rel", 1e9))
        max_edge = int(self.cfg.get("max_block_edge", 1))
        for seed_idx in sorted(range(len(patch_coords)), key=lambda i: patch_priorities[i], reverse=True):
            if used[seed_idx] or not math.isfinite(patch_depths[seed_idx]):
                continue
            cluster = [seed_idx]
            used[seed_idx] = True
            todo = [seed_idx]
            while todo and len(cluster) < max_edge * max_edge:
                cur = todo.pop(0)
                cgh, cgw = patch_coords[cur]
                for j, (gh2, gw2) in enumerate(patch_coords):
                    if used[j] or not math.isfinite(patch_depths[j]):
                        continue
                    if max(abs(cgh - gh2), abs(cgw - gw2)) > 1:
                        continue
                    z1, z2 = patch_depths[cur], patch_depths[j]
                    depth_diff = abs(z1 - z2)
                    lo = min(abs(z1), abs(z2))
                    rel_diff = depth_diff / lo if lo > 0 else float("inf")
                    if depth_diff <= z_abs or rel_diff <= z_rel:
                        cluster.append(j)
                        used[j] = True
                        todo.append(j)
            mask2d = torch.zeros((H, W), dtype=torch.bool, device=self.device)
            for idx in cluster:
                gh, gw = patch_coords[idx]
                mask2d[gh * patch_size:min((gh + 1) * patch_size, H), gw * patch_size:min((gw + 1) * patch_size, W)] = True
            depths = [patch_depths[i] for i in cluster if math.isfinite(patch_depths[i])]
            blocks.append(PatchBlock(kf_id=kf_id, block_id=block_id, patch_indices=[patch_coords[i] for i in cluster],
                                     pixel_mask=mask2d.reshape(-1),
                                     depth_median=float(np.median(depths)),
                                     priority=float(np.mean([patch_priorities[i] for i in cluster])),
                                     depth_variance=float(np.var(depths))))
            block_id += 1
        return blocks

    def process_queue(self, max_blocks=None):
        """The body of the reference's worker loop (tsdf_refine.py:603-640) for the queued blocks; returns the number
        of blocks processed."""
        done = 0
        while (max_blocks is None or done < max_blocks) and not self.queue.empty():
            key, block = self.queue.get_nowait()
            if not self.registry.begin_run(key):
                continue
            try:
                success, gain = self._refine_block_enhanced(block)
            except Exception as e:   # the reference prints and counts the block as failed (:627-631)
                print(f"[TSDF-ERROR] Refinement error for block {key.block_id}: {e}")
                success, gain = False, 0.0
            self.stats["total_blocks"] += 1
            if success:
                self.stats["successful_blocks"] += 1
            self.registry.finish_run(key, success, gain)
            done += 1
        return done

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
        with self.edit_section():       # the in-place edit of the keyframe (a threaded owner makes it a hand-over section)
            if kf.C.ndim == 2:
                kf.C[mask_hits, 0] += boost
            else:
                kf.C[mask_hits] += boost
            kf.C.clamp_(max=cmax)
            gw = float(cfg.get("geometric_weight", 0.0))
            if gw > 0:
                kf.X_canon[mask_hits] = ((1 - gw) * X_canon + gw * X_refined)[mask_hits]
            self._bump_version(block.kf_id, start_version + 1)
            touch = getattr(self.keyframes, "touch", None)      # edited in place: whoever caches the tensors re-reads them
            if touch is not None:
                touch(block.kf_id)
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
