"""FactorGraph edge bookkeeping (global_opt.py:32-112) against states recorded from the reference's own
class (tests/golden/factor_graph.npz): acceptance by match fraction in both directions, the consecutive-edge
exemption, all-or-nothing relocalisation calls, two-way edge preparation.  Host tensors; exact."""
import os

import numpy as np
import torch

from mast3r_slam.config import config
from mast3r_slam.global_opt import FactorGraph


def test_add_factors_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "factor_graph.npz"))
    assert config["local_opt"]["Q_conf"] == 1.5
    fg = FactorGraph(None, None, device="cpu")
    t = lambda k: torch.from_numpy(fx[k])
    rets = []
    for c in range(4):
        ret = fg.add_matched_factors(fx[f"call{c}_ii"].tolist(), fx[f"call{c}_jj"].tolist(), t(f"call{c}_idx_i2j"),
                                     t(f"call{c}_idx_j2i"), t(f"call{c}_vj"), t(f"call{c}_vi"), t(f"call{c}_Qii"), t(f"call{c}_Qjj"),
                                     t(f"call{c}_Qji"), t(f"call{c}_Qij"), min_match_frac=0.3, is_reloc=bool(fx[f"call{c}_reloc"]))
        rets.append(bool(ret))
        assert bool(ret) == bool(fx[f"call{c}_ret"]), c
        for k in ("ii", "jj", "idx_ii2jj", "idx_jj2ii", "valid_match_j", "valid_match_i", "Q_ii2jj", "Q_jj2ii"):
            np.testing.assert_array_equal(getattr(fg, k).numpy(), fx[f"state{c}_{k}"], err_msg=f"{c} {k}")
    assert rets == [True, True, False, True]                      # the relocalisation call with a poor edge adds nothing
    assert fg.ii.tolist() == [0, 0, 1, 3] and fg.jj.tolist() == [1, 2, 2, 4]     # (1,3) rejected, consecutive (1,2) kept
    for k, v in zip(("ii", "jj", "idx", "valid", "Q"), fg.prep_two_way_edges()):
        np.testing.assert_array_equal(v.numpy(), fx[f"two_way_{k}"])
    np.testing.assert_array_equal(fg.get_unique_kf_idx().numpy(), fx["unique"])
