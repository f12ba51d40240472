"""Global parameter dict, same access pattern as the reference (``from mast3r_slam.config import
config``; mast3r_slam/config.py:4,50-54).  The YAML ``inherit:`` loader is out of scope; the values
below are the hot-path numerical contract of config/base.yaml (lines 1-125) and are what the
kernels receive as arguments.  ``set_global_config`` overlays user values exactly like the
reference's merge (config.py:40-48)."""
import copy

_BASE = {
    "use_calib": False,
    "single_thread": False,
    "dataset": {"subsample": 1, "img_downsample": 1, "center_principle_point": True},
    "matching": {
        "max_iter": 10, "lambda_init": 1e-8, "convergence_thresh": 1e-6, "dist_thresh": 1e-1,
        "radius": 3, "dilation_max": 5,
    },
    "tracking": {
        "min_match_frac": 0.05, "max_iters": 50, "C_conf": 0.0, "Q_conf": 1.5, "rel_error": 1e-3,
        "delta_norm": 1e-3, "huber": 1.345, "match_frac_thresh": 0.333, "sigma_ray": 0.003,
        "sigma_dist": 1e1, "sigma_pixel": 1.0, "sigma_depth": 1e1, "sigma_point": 0.05,
        "pixel_border": -10, "depth_eps": 1e-6, "filtering_mode": "weighted_pointmap",
        "filtering_score": "median",
    },
    "local_opt": {
        "pin": 1, "window_size": 1e6, "C_conf": 0.0, "Q_conf": 1.5, "min_match_frac": 0.1,
        "pixel_border": -10, "depth_eps": 1e-6, "max_iters": 10, "sigma_ray": 0.003, "sigma_dist": 1e1,
        "sigma_pixel": 1.0, "sigma_depth": 1e1, "sigma_point": 0.05, "delta_norm": 1e-8, "use_cuda": True,
    },
    "retrieval": {"k": 3, "min_thresh": 5e-3},
    "reloc": {"min_match_frac": 0.3, "strict": True},
    "tsdf_refine": {
        "enabled": True, "window_size": 5, "voxel_size": 0.02, "trunc_dist": 0.08, "max_grid_dim": 64,
        "roi_size": 0.4, "ray_samples": 64, "max_displacement": 0.015, "min_weight_threshold": 0.01,
        "confidence_boost": 0.08, "confidence_max": 1.3, "min_hit_rate": 0.05, "max_rois_per_kf": 3,
        "min_confidence": 0.2, "pose_stable_thresh": 0.01,
    },
    "tsdf_global": {
        "enabled": False, "voxel_size": 0.03, "trunc_dist": 0.12, "max_weight": 100.0,
        "min_tsdf_weight": 1.0e-3, "max_points_per_kf": 40000, "min_confidence": 0.05,
        "samples_per_kf": 2000, "lambda": 0.15, "max_iterations": 3, "pre_icp_iters": 2, "damping": 1.0e-4,
    },
}

config = copy.deepcopy(_BASE)


def merge_config(dict1, dict2):
    for k, v in dict2.items():
        if isinstance(v, dict):
            merge_config(dict1.setdefault(k, {}), v)
        else:
            dict1[k] = v
    return dict1


def set_global_config(cfg):
    merge_config(config, cfg)
    return config


def reset_config():
    config.clear()
    config.update(copy.deepcopy(_BASE))
    return config
