"""RetrievalDatabase.from_checkpoint's loaders execute nothing from the files they read (the reference unpickles both:
thirdparty/mast3r/mast3r/retrieval/processor.py:64-98): the codebook goes through an unpickler that resolves numpy's array
reconstructors only, the checkpoint through torch's weights-only loader.  CPU only (no device call)."""
import argparse
import os
import pickle

import numpy as np
import pytest
import torch

from mast3r_slam.retrieval_database import load_codebook


class _Payload:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_codebook_pickle_roundtrip_and_npz(tmp_path):
    cen = np.random.default_rng(0).standard_normal((32, 16)).astype(np.float32)
    state = {"type": "codebook", "params": {"index": {"gpu_id": 0}}, "state": {"centroids": cen, "idf": None}}   # Codebook.state_dict()
    p = tmp_path / "x_codebook.pkl"
    p.write_bytes(pickle.dumps(state))
    assert np.array_equal(load_codebook(str(p)), cen)
    np.savez(tmp_path / "x_codebook.npz", centroids=cen)
    assert np.array_equal(load_codebook(str(tmp_path / "x_codebook.npz")), cen)
    np.save(tmp_path / "x_codebook.npy", cen)
    assert np.array_equal(load_codebook(str(tmp_path / "x_codebook.npy")), cen)


def test_codebook_with_a_reduce_payload_is_rejected(tmp_path):
    p = tmp_path / "evil_codebook.pkl"
    p.write_bytes(pickle.dumps({"state": {"centroids": _Payload()}}))
    with pytest.raises(pickle.UnpicklingError):
        load_codebook(str(p))


def test_checkpoint_with_a_reduce_payload_is_rejected(tmp_path):
    """The weights-only loader with argparse.Namespace allow-listed: a normal checkpoint loads, one that carries a
    callable payload raises (the class is imported lazily: from_checkpoint needs the device library only at the end)."""
    good = tmp_path / "m_retrieval.pth"
    torch.save({"args": argparse.Namespace(nfeat=300, residual=False), "model": {"projector.0.weight": torch.zeros(4, 4)}}, good)
    with torch.serialization.safe_globals([argparse.Namespace]):
        ck = torch.load(str(good), "cpu", weights_only=True)
    assert ck["args"].nfeat == 300
    bad = tmp_path / "e_retrieval.pth"
    torch.save({"args": _Payload(), "model": {}}, bad)
    from mast3r_slam.retrieval_database import RetrievalDatabase

    with pytest.raises(pickle.UnpicklingError):
        RetrievalDatabase.from_checkpoint(str(bad), device="cpu")      # raises inside the loader, before any device call
