"""Generates the committed golden fixtures by RUNNING the importable pieces of the reference
(read-only at /root/reference) in the build container.  The reference never travels to the GPU
box; only these small .npz data files do.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [section ...]

Sections: prep  tsdf_global  tsdf_refine  network
Every fixture records numpy/torch versions (the global TSDF arithmetic depends on NumPy's
promotion rules: the container has NumPy 2.x (NEP 50), the reference pins numpy==1.26.4).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "mast3r-slam-quality-dualtsdf_amd"))


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def meta():
    return dict(numpy_version=np.__version__, torch_version=torch.__version__)


def section_prep():
    """prep_for_iter_proj (matching.py:25-49) = F.normalize + img_gradient (image.py:5-38)."""
    import torch.nn.functional as F
    from mast3r_slam import synthetic

    image = load_by_path("ref_image", f"{REF}/mast3r_slam/image.py")
    pair = synthetic.make_pair(0, 6, h=48, w=64, seed=3)
    X11 = torch.from_numpy(pair["X11"])[None]
    X21 = torch.from_numpy(pair["X21"])[None]
    # --- the reference lines, verbatim in behaviour ---
    rays_img = F.normalize(X11, dim=-1).permute(0, 3, 1, 2)
    gx, gy = image.img_gradient(rays_img)
    rays_with_grad = torch.cat((rays_img, gx, gy), dim=1).permute(0, 2, 3, 1).contiguous()
    pts3d_norm = F.normalize(X21.view(1, -1, 3), dim=-1)
    np.savez_compressed(
        os.path.join(HERE, "prep_iter_proj.npz"), X11=X11.numpy(), X21=X21.numpy(),
        rays_with_grad=rays_with_grad.numpy(), pts3d_norm=pts3d_norm.numpy(), **meta(),
    )
    print("prep_iter_proj.npz", rays_with_grad.shape)


SECTIONS = {"prep": section_prep}

if __name__ == "__main__":
    todo = sys.argv[1:] or list(SECTIONS)
    for s in todo:
        SECTIONS[s]()
