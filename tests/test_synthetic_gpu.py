"""The synthetic-data source of the product-loop bench (mast3r_slam/synthetic_gpu.py): the device renderer against the
numpy room of mast3r_slam/synthetic.py, the fused kernel (csrc/room.hip) against the torch formulation, and the
property the sharded backend relies on: a pair's output does not depend on the batch it is rendered in."""
import numpy as np
import pytest
import torch

from mast3r_slam import synthetic

pytestmark = pytest.mark.gpu


def test_fused_room_pair_matches_torch_and_numpy(device):
    from mast3r_slam.synthetic_gpu import RoomRenderer

    H, W = 96, 128
    R = RoomRenderer(device, H, W)
    ki, kj = torch.tensor([3.0, 27.0, 250.0], device=device), torch.tensor([0.0, 9.0, 244.0], device=device)
    a, b = R.pair(ki, kj, noise=0.002)
    fa, fb = R.pair_fused(ki, kj, noise=0.002)
    for k in ("pts3d", "conf", "desc", "desc_conf"):
        np.testing.assert_allclose(fa[k].cpu().numpy(), a[k].cpu().numpy(), atol=2e-5, err_msg=k)
        np.testing.assert_allclose(fb[k].cpu().numpy(), b[k].cpu().numpy(), atol=2e-5, err_msg=k)
    pr = synthetic.make_pair(27, 9, h=H, w=W, noise=0.0)
    fa0, fb0 = R.pair_fused(ki[1:2], kj[1:2], noise=0.0)
    np.testing.assert_allclose(fa0["pts3d"][0].cpu().numpy(), pr["X11"], atol=2e-6)
    np.testing.assert_allclose(fb0["pts3d"][0].cpu().numpy(), pr["X21"], atol=2e-6)
    np.testing.assert_allclose(fb0["desc"][0].cpu().numpy(), pr["D21"], atol=2e-6)
    # batch independence, bit for bit
    one_a, one_b = R.pair_fused(ki[2:3], kj[2:3], noise=0.002)
    assert torch.equal(one_a["pts3d"][0], fa["pts3d"][2]) and torch.equal(one_b["desc"][0], fb["desc"][2])
    assert torch.equal(one_b["conf"][0], fb["conf"][2])
