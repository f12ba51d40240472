"""geometry.py / nonlinear_optimizer.py / lietorch_utils.py mirrors against values produced by the reference's
own functions (tests/golden/geometry.npz, make_golden.py geometry).  float64 on the host: 1e-12."""
import os

import numpy as np
import torch

from mast3r_slam import geometry as G
from mast3r_slam import nonlinear_optimizer as N
from mast3r_slam import synthetic
from mast3r_slam.lietorch_utils import as_SE3


def test_geometry_matches_reference(golden_dir):
    fx = np.load(os.path.join(golden_dir, "geometry.npz"))
    X, K = torch.from_numpy(fx["X"]), torch.from_numpy(fx["K"])
    eq = lambda a, b: np.testing.assert_allclose(a.numpy() if torch.is_tensor(a) else a, b, rtol=1e-12, atol=1e-12)
    eq(G.skew_sym(X), fx["skew"])
    eq(G.point_to_dist(X), fx["dist"])
    rd, J = G.point_to_ray_dist(X, jacobian=True)
    eq(rd, fx["rd"]); eq(J, fx["rd_J"]); eq(G.point_to_ray_dist(X), fx["rd"])
    pz, Jp, valid = G.project_calib(X, K, (48, 64), jacobian=True, border=2, z_eps=1e-6)
    eq(pz, fx["pz"]); eq(Jp, fx["pz_J"])
    np.testing.assert_array_equal(valid.numpy(), fx["pz_valid"])
    assert not valid[0, 0, 0] and pz[0, 0, 2] == 0.0        # behind the camera: log z forced to 0, invalid

    class Pose:
        def act(self, P):
            return torch.from_numpy(synthetic.sim3_act(fx["T"], P.numpy()))

    pW, Ja = G.act_Sim3(Pose(), X, jacobian=True)
    eq(pW, fx["act"]); eq(Ja, fx["act_J"])
    eq(G.backproject(pz[..., :2], X[..., 2:3], K), fx["bp"])
    eq(G.constrain_points_to_ray((5, 7), X[None], K), fx["ray"])
    r = torch.from_numpy(fx["r"])
    eq(N.huber(r), fx["huber"]); eq(N.tukey(r), fx["tukey"])
    conv = [N.check_convergence(0, 1e-3, 1e-3, 10.0, c, torch.tensor([d, 0.0])) for c, d in ((9.0, 1.0), (9.9999, 1.0), (5.0, 1e-4))]
    np.testing.assert_array_equal(np.array(conv), fx["conv"])


def test_as_SE3_drops_scale():
    from lietorch_hip import Sim3

    d = torch.tensor([[[1.0, 2, 3, 0, 0, 0, 1, 1.7]], [[4.0, 5, 6, 0, 1, 0, 0, 0.5]]])
    se3 = as_SE3(Sim3(d))
    assert se3.data.shape == (2, 7) and torch.equal(se3.data, d.reshape(2, 8)[:, :7]) and as_SE3(se3) is se3
