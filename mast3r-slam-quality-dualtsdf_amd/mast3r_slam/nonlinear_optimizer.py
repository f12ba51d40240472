"""Mirror of mast3r_slam/nonlinear_optimizer.py.  The fused tracker kernel applies the same Huber weight and
convergence rule on the device (csrc/tracker.hip); these tensor forms are for callers outside it."""
import math

import torch


def check_convergence(iter, rel_error_threshold, delta_norm_threshold, old_cost, new_cost, delta, verbose=False):
    """nonlinear_optimizer.py:5-25: relative cost decrease or step norm below threshold."""
    rel_dec = math.fabs((old_cost - new_cost) / old_cost)
    delta_norm = torch.linalg.norm(delta)
    converged = rel_dec < rel_error_threshold or delta_norm < delta_norm_threshold
    if verbose:
        print(f"iter={iter} | new_cost={new_cost} rel_dec={rel_dec} delta_norm={delta_norm} | converged={converged}")
    return converged


def huber(r, k=1.345):
    """nonlinear_optimizer.py:28-33: IRLS weight 1 inside k, k/|r| outside."""
    a = torch.abs(r)
    return torch.where(a < k, torch.ones((1), dtype=r.dtype, device=r.device), k / a)


def tukey(r, t=4.6851):
    """nonlinear_optimizer.py:36-42: biweight (1 - (r/t)^2)^2 inside t, 0 outside."""
    a = torch.abs(r)
    w = (1 - torch.square(a / t)) ** 2
    return torch.where(a < t, w, torch.zeros((), dtype=r.dtype, device=r.device))
