"""Frame.update_pointmap / get_average_conf in every filtering mode against the reference's own Frame class
(tests/golden/frame_update.npz), and the default configuration against the reference's config/base.yaml as its
loader reads it (tests/golden/reference_base_config.json).  Host tensors; exact."""
import json
import os

import numpy as np
import pytest
import torch

from mast3r_slam import config as cfgmod
from mast3r_slam.frame import Frame


@pytest.fixture()
def restore_config():
    import copy
    saved = copy.deepcopy(cfgmod.config)
    yield
    cfgmod.config.clear()
    cfgmod.config.update(saved)


def test_update_pointmap_modes(golden_dir, restore_config):
    fx = np.load(os.path.join(golden_dir, "frame_update.npz"))
    Xs, Cs = torch.from_numpy(fx["X"]), torch.from_numpy(fx["C"])
    for mode in ("first", "recent", "best_score", "indep_conf", "weighted_pointmap", "weighted_spherical"):
        for score in (("median", "mean") if mode == "best_score" else ("median",)):
            cfgmod.config["tracking"]["filtering_mode"] = mode
            cfgmod.config["tracking"]["filtering_score"] = score
            f = Frame(0, torch.zeros(1, 3, 4, 4), None, None, None)
            for k in range(4):
                f.update_pointmap(Xs[k], Cs[k])
                tag = f"{mode}_{score}_{k}"
                np.testing.assert_array_equal(f.X_canon.numpy(), fx[tag + "_X"], err_msg=tag)
                np.testing.assert_array_equal(f.C.numpy(), fx[tag + "_C"], err_msg=tag)
                assert [f.N, f.N_updates] == fx[tag + "_N"].tolist(), tag
                np.testing.assert_array_equal(f.get_average_conf().numpy(), fx[tag + "_avg"], err_msg=tag)


def test_default_config_equals_reference_base_yaml(golden_dir):
    """Every key of the reference's base.yaml that the mirrored modules read has the same default here."""
    ref = json.load(open(os.path.join(golden_dir, "reference_base_config.json")))
    cfgmod.reset_config()
    ours = cfgmod.config
    for section in ("tracking", "local_opt", "matching", "tsdf_refine", "tsdf_global", "dataset"):
        for key, val in ref[section].items():
            if key in ours.get(section, {}):
                assert ours[section][key] == val, (section, key, ours[section][key], val)
    for section, keys in (("tracking", ("C_conf", "Q_conf", "max_iters", "rel_error", "delta_norm", "huber", "min_match_frac",
                                        "match_frac_thresh", "filtering_mode", "filtering_score", "sigma_ray", "sigma_dist")),
                          ("local_opt", ("sigma_ray", "sigma_dist", "C_conf", "Q_conf", "max_iters", "delta_norm", "pin",
                                         "min_match_frac", "pixel_border", "depth_eps")),
                          ("matching", ("max_iter", "lambda_init", "convergence_thresh", "dist_thresh", "radius", "dilation_max"))):
        for key in keys:
            if key in ref[section]:
                assert key in ours[section], (section, key)
    assert ours["use_calib"] == ref["use_calib"]
