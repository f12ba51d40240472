"""RetrievalDatabase.quantize_custom (retrieval_database.py:96-105) on the MFMA GEMM against indices
recorded from the reference method's own source (tests/golden/retrieval_quantize.npz; inputs regenerated from the seed):
65 536 x 1024 codebook, 768 features.  Index-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_quantize_custom_matches_reference_indices(device, golden_dir):
    from mast3r_slam.retrieval_database import RetrievalDatabase

    fx = np.load(os.path.join(golden_dir, "retrieval_quantize.npz"))
    g = torch.Generator().manual_seed(int(fx["seed"]))
    centroids = torch.randn(65536, 1024, generator=g)
    q = torch.randn(768, 1024, generator=g)
    q[:64] = centroids[1000:1064] + 0.05 * torch.randn(64, 1024, generator=g)
    db = RetrievalDatabase(centroids, device=device)
    for name, k in (("query", 5), ("build", 1)):
        idx = db.quantize_custom(q.to(device), {"quantize": {"multiple_assignment": k}})
        assert idx.shape == (768, k) and idx.dtype == torch.int64
        np.testing.assert_array_equal(idx.cpu().numpy(), fx[name])
    with pytest.raises(RuntimeError, match="asmk"):
        db.update(None, True, 3)
