"""Mirror of mast3r_slam/lietorch_utils.py: Sim3 -> SE3 for trajectory output (drops the scale)."""
import torch


class SE3:
    """Minimal lietorch.SE3 stand-in: `.data` (..., 7) = [t(3), q(xyzw)]."""

    embedded_dim = 7

    def __init__(self, data):
        self.data = data


def as_SE3(X):
    """lietorch_utils.py:6-13: flattens the batch dims, moves to the host, keeps translation and rotation."""
    if isinstance(X, SE3):
        return X
    d = X.data.detach().cpu().reshape(-1, X.data.shape[-1])
    return SE3(torch.cat((d[:, :3], d[:, 3:7]), dim=-1))
