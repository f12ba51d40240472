"""Mirror of the reference's mast3r_slam/mast3r_utils.py (lines 14-278): same function names, argument
order and return tuples; the model object is mast3r_slam.mast3r_model.Mast3rHIP and matching is
mast3r_slam.matching (both libmslam_hip.so); the retrieval database is mast3r_slam.retrieval_database."""
import numpy as np
import torch

import mast3r_slam.matching as matching
from mast3r_slam.config import config
from mast3r_slam.mast3r_model import Mast3rConfig, Mast3rHIP, load_mast3r_state_dict


def load_mast3r(path=None, device="cuda"):
    """mast3r_utils.py:14-21."""
    weights_path = "checkpoints/MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric.pth" if path is None else path
    return Mast3rHIP(load_mast3r_state_dict(weights_path), Mast3rConfig(), device=device)


def load_retriever(mast3r_model, retriever_path=None, device="cuda"):
    """mast3r_utils.py:24-31.  The backbone argument is kept for the signature: the database only ever consumes
    `frame.feat` (retrieval_database.py:24-41), never the encoder itself."""
    from mast3r_slam.retrieval_database import RetrievalDatabase

    retriever_path = ("checkpoints/MASt3R_ViTLarge_BaseDecoder_512_catmlpdpt_metric_retrieval_trainingfree.pth"
                      if retriever_path is None else retriever_path)
    return RetrievalDatabase.from_checkpoint(retriever_path, device=device)


def _hw(shape):
    s = shape[0] if (hasattr(shape, "ndim") and shape.ndim == 2) or isinstance(shape[0], (list, tuple)) else shape
    return int(s[0]), int(s[1])


@torch.inference_mode()
def decoder(model, feat1, feat2, pos1, pos2, shape1, shape2):
    """mast3r_utils.py:34-40: _decoder + both _downstream_head calls (one native call here)."""
    H, W = _hw(shape1)
    res1, res2 = model.decode_pair(feat1, feat2, H, W)
    return res1, res2


def downsample(X, C, D, Q):
    """mast3r_utils.py:43-52."""
    ds = config["dataset"]["img_downsample"]
    if ds > 1:
        X = X[..., ::ds, ::ds, :].contiguous()
        C = C[..., ::ds, ::ds].contiguous()
        D = D[..., ::ds, ::ds, :].contiguous()
        Q = Q[..., ::ds, ::ds].contiguous()
    return X, C, D, Q


def _ensure_feat(model, frame):
    if frame.feat is None:
        frame.feat, frame.pos, _ = model._encode_image(frame.img, frame.img_true_shape)


def _stack(res):
    X, C, D, Q = zip(*[(r["pts3d"][0], r["conf"][0], r["desc"][0], r["desc_conf"][0]) for r in res])
    return torch.stack(X), torch.stack(C), torch.stack(D), torch.stack(Q)


@torch.inference_mode()
def mast3r_symmetric_inference(model, frame_i, frame_j):
    """mast3r_utils.py:55-79 -> X,C,D,Q stacked [ii, ji, jj, ij]."""
    _ensure_feat(model, frame_i)
    _ensure_feat(model, frame_j)
    res11, res21 = decoder(model, frame_i.feat, frame_j.feat, frame_i.pos, frame_j.pos, frame_i.img_true_shape,
                           frame_j.img_true_shape)
    res22, res12 = decoder(model, frame_j.feat, frame_i.feat, frame_j.pos, frame_i.pos, frame_j.img_true_shape,
                           frame_i.img_true_shape)
    return downsample(*_stack([res11, res21, res22, res12]))


@torch.inference_mode()
def mast3r_decode_symmetric_batch(model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j, cached=None):
    """mast3r_utils.py:83-115.  The reference loops over the B edges in Python (one decoder call per
    edge and direction); here all edges and both directions are ONE batched native call.

    `cached` (optional): {row: (X, C, D, Q)} - two-view results (res11, res21 stacked, as mast3r_asymmetric_inference
    returns them) that already exist for rows of the batch: row e < B is decoder(feat_i[e], feat_j[e]), row B + e is
    decoder(feat_j[e], feat_i[e]).  Tracking has computed exactly the second kind for the consecutive edge of a new
    keyframe (the frame against the keyframe it was tracked on); rows of a batch do not depend on the batch, so taking
    them over changes no bit and saves one of the ~8 decoder + head rows of a keyframe's backend task."""
    H, W = _hw(shape_i[0])
    B = feat_i.shape[0]
    # both directions of all B edges in ONE native call of batch 2B: rows [0, B) decode (i, j), rows [B, 2B) decode
    # (j, i); results are bitwise those of separate calls (no arithmetic depends on the batch size)
    f1, f2 = torch.cat((feat_i, feat_j)), torch.cat((feat_j, feat_i))
    keys = ("pts3d", "conf", "desc", "desc_conf")
    if not cached:
        ra, rb = model.decode_pair(f1, f2, H, W)
        pick = lambda k: torch.stack((ra[k][:B], rb[k][:B], ra[k][B:], rb[k][B:]))      # order [ii, ji, jj, ij]
        return downsample(pick("pts3d"), pick("conf"), pick("desc"), pick("desc_conf"))  # X: (4, B, H, W, 3)
    keep = [r for r in range(2 * B) if r not in cached]
    out = []
    if f1.is_cuda:      # the cached rows were produced (and allocated) on the tracking stream; the caller drops them right
        cur = torch.cuda.current_stream(f1.device)   # after this call, while the copies below are still queued on this stream
        for res in cached.values():
            for t in res:
                t.record_stream(cur)
    if keep:
        sel = torch.tensor(keep, device=f1.device)
        ra, rb = model.decode_pair(f1[sel].contiguous(), f2[sel].contiguous(), H, W)
        part = downsample(*(torch.stack((ra[k], rb[k])) for k in keys))                  # each (2, n_keep, H', W'[, c])
    some = cached[next(iter(cached))]
    for q in range(4):
        full = torch.empty((2, 2 * B) + tuple(some[q].shape[1:]), dtype=some[q].dtype, device=some[q].device)
        if keep:
            full[:, sel] = part[q]
        for r, res in cached.items():
            full[:, r] = res[q]
        out.append(torch.stack((full[0, :B], full[1, :B], full[0, B:], full[1, B:])))    # [ii, ji, jj, ij]
    return tuple(out)


@torch.inference_mode()
def mast3r_inference_mono(model, frame):
    """mast3r_utils.py:118-139 -> (Xii (HW,3), Cii (HW,1))."""
    _ensure_feat(model, frame)
    res11, res21 = decoder(model, frame.feat, frame.feat, frame.pos, frame.pos, frame.img_true_shape,
                           frame.img_true_shape)
    X, C, D, Q = downsample(*_stack([res11, res21]))
    return X[0].reshape(-1, 3), C[0].reshape(-1, 1)


def mast3r_match_symmetric(model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j, cached=None):
    """mast3r_utils.py:142-180 (`cached`: see mast3r_decode_symmetric_batch)."""
    X, C, D, Q = mast3r_decode_symmetric_batch(model, feat_i, pos_i, feat_j, pos_j, shape_i, shape_j, cached=cached)
    b = X.shape[1]
    Xii, Xji, Xjj, Xij = X[0], X[1], X[2], X[3]
    Dii, Dji, Djj, Dij = D[0], D[1], D[2], D[3]
    Qii, Qji, Qjj, Qij = Q[0], Q[1], Q[2], Q[3]
    X11 = torch.cat((Xii, Xjj), dim=0)
    X21 = torch.cat((Xji, Xij), dim=0)
    D11 = torch.cat((Dii, Djj), dim=0)
    D21 = torch.cat((Dji, Dij), dim=0)
    idx_1_to_2, valid_match_2 = matching.match(X11, X21, D11, D21)
    match_b = X11.shape[0] // 2
    return (idx_1_to_2[:match_b], idx_1_to_2[match_b:], valid_match_2[:match_b], valid_match_2[match_b:],
            Qii.reshape(b, -1, 1), Qjj.reshape(b, -1, 1), Qji.reshape(b, -1, 1), Qij.reshape(b, -1, 1))


@torch.inference_mode()
def encode_frames(model, frames):
    """Frame-group extension (no reference counterpart; the reference encodes lazily, one frame per call,
    mast3r_utils.py:186-193): ONE batch encoder call fills .feat / .pos of every frame that has none.  The
    encoder does not depend on any pose, so a sequence reader may run it any number of frames ahead; rows of a
    batch are computed exactly as in a batch of one (tests/test_mast3r_gpu.py::test_frame_group…)."""
    todo = [f for f in frames if f.feat is None]
    if not todo:
        return
    feat, pos, _ = model._encode_image(torch.cat([f.img for f in todo]), todo[0].img_true_shape)
    for k, f in enumerate(todo):
        f.feat, f.pos = feat[k:k + 1], pos[k:k + 1]


@torch.inference_mode()
def mast3r_asymmetric_inference_group(model, frames, keyframe):
    """Frame-group extension: the two-view forward of `frames` (consecutive, not yet tracked) against the current
    keyframe in ONE batch call.  Each frame keeps its (X, C, D, Q) tagged with the keyframe it was decoded against;
    mast3r_asymmetric_inference(frame, keyframe) consumes it if the keyframe is still the same one and recomputes
    otherwise (speculation: a frame of the group may become a keyframe, which invalidates the rest)."""
    frames = list(frames)
    encode_frames(model, frames)
    _ensure_feat(model, keyframe)
    B = len(frames)
    feat_i = torch.cat([f.feat for f in frames])
    res11, res21 = decoder(model, feat_i, keyframe.feat.expand(B, -1, -1).contiguous(), None, None,
                           frames[0].img_true_shape, keyframe.img_true_shape)
    for k, f in enumerate(frames):
        pick = lambda key: torch.stack((res11[key][k], res21[key][k]))
        f.decoded = (int(keyframe.frame_id), downsample(pick("pts3d"), pick("conf"), pick("desc"), pick("desc_conf")))


@torch.inference_mode()
def mast3r_asymmetric_inference(model, frame_i, frame_j):
    """mast3r_utils.py:183-206.  A result left by mast3r_asymmetric_inference_group for this very pair is used
    once instead of being recomputed.  The result stays attached to frame_i (`pair_decode`, tagged with frame_j's id): if
    frame_i becomes a keyframe, the backend's symmetric inference of the edge (frame_j, frame_i) takes this direction over."""
    stash = getattr(frame_i, "decoded", None)
    if stash is not None:
        frame_i.decoded = None
        if stash[0] == int(frame_j.frame_id):   # a keyframe's features never change once encoded
            frame_i.pair_decode = stash
            return stash[1]
    _ensure_feat(model, frame_i)
    _ensure_feat(model, frame_j)
    res11, res21 = decoder(model, frame_i.feat, frame_j.feat, frame_i.pos, frame_j.pos, frame_i.img_true_shape,
                           frame_j.img_true_shape)
    out = downsample(*_stack([res11, res21]))
    if hasattr(frame_j, "frame_id"):
        frame_i.pair_decode = (int(frame_j.frame_id), out)
    return out


def mast3r_match_asymmetric(model, frame_i, frame_j, idx_i2j_init=None):
    """mast3r_utils.py:209-231."""
    X, C, D, Q = mast3r_asymmetric_inference(model, frame_i, frame_j)
    b, h, w = X.shape[:-1]
    b = b // 2
    Xii, Xji = X[:b], X[b:]
    Dii, Dji = D[:b], D[b:]
    idx_i2j, valid_match_j = matching.match(Xii, Xji, Dii, Dji, idx_1_to_2_init=idx_i2j_init)
    Xii, Xji = X.reshape(2, h * w, 3)
    Cii, Cji = C.reshape(2, h * w, 1)
    Qii, Qji = Q.reshape(2, h * w, 1)
    return idx_i2j, valid_match_j, Xii, Cii, Qii, Xji, Cji, Qji


def _resize_pil_image(img, long_edge_size):
    import PIL.Image

    S = max(img.size)
    interp = PIL.Image.LANCZOS if S > long_edge_size else PIL.Image.BICUBIC
    new_size = tuple(int(round(x * long_edge_size / S)) for x in img.size)
    return img.resize(new_size, interp)


def resize_img(img, size, square_ok=False, return_transformation=False):
    """mast3r_utils.py:244-278 (CPU, PIL): long side -> 512, centre crop to multiples of 16, ImgNorm."""
    import PIL.Image

    assert size == 224 or size == 512
    img = PIL.Image.fromarray(np.uint8(img * 255))
    W1, H1 = img.size
    if size == 224:
        img = _resize_pil_image(img, round(size * max(W1 / H1, H1 / W1)))
    else:
        img = _resize_pil_image(img, size)
    W, H = img.size
    cx, cy = W // 2, H // 2
    if size == 224:
        half = min(cx, cy)
        img = img.crop((cx - half, cy - half, cx + half, cy + half))
    else:
        halfw, halfh = ((2 * cx) // 16) * 8, ((2 * cy) // 16) * 8
        if not square_ok and W == H:
            halfh = 3 * halfw / 4
        img = img.crop((cx - halfw, cy - halfh, cx + halfw, cy + halfh))
    arr = np.asarray(img)
    norm = (torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5   # ImgNorm
    res = dict(img=norm[None], true_shape=np.int32([img.size[::-1]]), unnormalized_img=arr)
    if return_transformation:
        return res, (W1 / W, H1 / H, (W - img.size[0]) / 2, (H - img.size[1]) / 2)
    return res
