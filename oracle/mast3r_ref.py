"""ORACLE (test infrastructure, NOT product code): plain PyTorch fp32 restatement of the MASt3R
two-view forward, written from the reference's module definitions, operating directly on a
state_dict with the upstream key names.

  patch embed      /root/reference/thirdparty/mast3r/dust3r/dust3r/patch_embed.py:19-29
                   dust3r/croco/models/blocks.py:195-236 (PositionGetter, PatchEmbed)
  encoder block    dust3r/croco/models/blocks.py:81-130 (Attention, Block), :58-79 (Mlp)
  RoPE2D           dust3r/croco/models/pos_embed.py:112-158 (the PyTorch fallback = the spec)
  _encode_image    dust3r/dust3r/model.py:127-139
  _decoder         dust3r/dust3r/model.py:171-190 ; DecoderBlock/CrossAttention blocks.py:132-191
  DPT head         dust3r/croco/models/dpt_block.py:264-450 ; dust3r/dust3r/heads/dpt_head.py:34-65
  MLP + postproc   mast3r/catmlp_dpt_head.py:25-96 ; dust3r/dust3r/heads/postprocess.py:22-58

Pinned against tests/golden/mast3r_small.safetensors-style fixtures produced by instantiating the
reference's AsymmetricMASt3R (reduced depth, seeded weights) in the build container
(tests/golden/make_golden.py section `network`).  Pre-trained weights are not available offline, so
parity on the real checkpoint is "unpinned here" (SURVEY §8c); the key layout is the upstream one.
"""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class Mast3rConfig:
    enc_dim: int = 1024
    enc_depth: int = 24
    enc_heads: int = 16
    dec_dim: int = 768
    dec_depth: int = 12
    dec_heads: int = 12
    patch: int = 16
    desc_dim: int = 24
    feature_dim: int = 256
    rope_base: float = 100.0
    ln_eps: float = 1e-6

    @property
    def hooks(self):
        l2 = self.dec_depth
        return [0, l2 * 2 // 4, l2 * 3 // 4, l2]


def positions(b, h, w, device):
    y, x = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack((y.reshape(-1), x.reshape(-1)), -1)[None].expand(b, -1, 2).clone()


def rope2d(tokens, pos, base):
    """tokens (B,H,N,D), pos (B,N,2) int64 [y,x]  (pos_embed.py:142-158)"""
    D = tokens.shape[-1] // 2
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2, device=tokens.device).float() / D))
    seq = int(pos.max()) + 1
    t = torch.arange(seq, device=tokens.device, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    freqs = torch.cat((freqs, freqs), -1)
    cos, sin = freqs.cos(), freqs.sin()

    def rot_half(x):
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
        return torch.cat((-x2, x1), -1)

    def rope1d(tok, p):
        c = F.embedding(p, cos)[:, None]
        s = F.embedding(p, sin)[:, None]
        return tok * c + rot_half(tok) * s

    y, x = tokens.chunk(2, -1)
    return torch.cat((rope1d(y, pos[:, :, 0]), rope1d(x, pos[:, :, 1])), -1)


def _ln(x, sd, pre, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[pre + ".weight"], sd[pre + ".bias"], eps)


def _lin(x, sd, pre):
    return F.linear(x, sd[pre + ".weight"], sd.get(pre + ".bias"))


def _mlp(x, sd, pre):
    return _lin(F.gelu(_lin(x, sd, pre + ".fc1")), sd, pre + ".fc2")


def _self_attn(x, pos, sd, pre, heads, base):
    B, N, C = x.shape
    qkv = _lin(x, sd, pre + ".qkv").reshape(B, N, 3, heads, C // heads).transpose(1, 3)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    q, k = rope2d(q, pos, base), rope2d(k, pos, base)
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(-1)
    return _lin((attn @ v).transpose(1, 2).reshape(B, N, C), sd, pre + ".proj")


def _cross_attn(xq, y, qpos, kpos, sd, pre, heads, base):
    B, Nq, C = xq.shape
    Nk = y.shape[1]
    q = _lin(xq, sd, pre + ".projq").reshape(B, Nq, heads, C // heads).permute(0, 2, 1, 3)
    k = _lin(y, sd, pre + ".projk").reshape(B, Nk, heads, C // heads).permute(0, 2, 1, 3)
    v = _lin(y, sd, pre + ".projv").reshape(B, Nk, heads, C // heads).permute(0, 2, 1, 3)
    q, k = rope2d(q, qpos, base), rope2d(k, kpos, base)
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(-1)
    return _lin((attn @ v).transpose(1, 2).reshape(B, Nq, C), sd, pre + ".proj")


def encode_image(sd, cfg: Mast3rConfig, img):
    """img (B,3,H,W) -> feat (B,N,enc_dim), pos (B,N,2)   (_encode_image, model.py:127-139)"""
    B, _, H, W = img.shape
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=cfg.patch)
    pos = positions(B, H // cfg.patch, W // cfg.patch, img.device)
    x = x.flatten(2).transpose(1, 2)
    for i in range(cfg.enc_depth):
        p = f"enc_blocks.{i}"
        x = x + _self_attn(_ln(x, sd, p + ".norm1", cfg.ln_eps), pos, sd, p + ".attn", cfg.enc_heads, cfg.rope_base)
        x = x + _mlp(_ln(x, sd, p + ".norm2", cfg.ln_eps), sd, p + ".mlp")
    return _ln(x, sd, "enc_norm", cfg.ln_eps), pos


def _dec_block(x, y, xpos, ypos, sd, p, cfg):
    x = x + _self_attn(_ln(x, sd, p + ".norm1", cfg.ln_eps), xpos, sd, p + ".attn", cfg.dec_heads, cfg.rope_base)
    y_ = _ln(y, sd, p + ".norm_y", cfg.ln_eps)
    x = x + _cross_attn(_ln(x, sd, p + ".norm2", cfg.ln_eps), y_, xpos, ypos, sd, p + ".cross_attn", cfg.dec_heads,
                        cfg.rope_base)
    return x + _mlp(_ln(x, sd, p + ".norm3", cfg.ln_eps), sd, p + ".mlp")


def decoder(sd, cfg: Mast3rConfig, f1, pos1, f2, pos2):
    """-> (dec1, dec2): two lists of dec_depth+1 tensors [enc feat, block outputs...]  (model.py:171-190)"""
    outs = [(f1, f2)]
    g1, g2 = _lin(f1, sd, "decoder_embed"), _lin(f2, sd, "decoder_embed")
    prev = (g1, g2)
    for i in range(cfg.dec_depth):
        n1 = _dec_block(prev[0], prev[1], pos1, pos2, sd, f"dec_blocks.{i}", cfg)
        n2 = _dec_block(prev[1], prev[0], pos2, pos1, sd, f"dec_blocks2.{i}", cfg)
        prev = (n1, n2)
        outs.append(prev)
    outs[-1] = (_ln(outs[-1][0], sd, "dec_norm", cfg.ln_eps), _ln(outs[-1][1], sd, "dec_norm", cfg.ln_eps))
    return [o[0] for o in outs], [o[1] for o in outs]


def _conv(x, sd, pre, **kw):
    return F.conv2d(x, sd[pre + ".weight"], sd.get(pre + ".bias"), **kw)


def _rcu(x, sd, pre):
    out = _conv(F.relu(x), sd, pre + ".conv1", padding=1)
    out = _conv(F.relu(out), sd, pre + ".conv2", padding=1)
    return out + x


def _fusion(sd, pre, *xs):
    out = xs[0]
    if len(xs) == 2:
        out = out + _rcu(xs[1], sd, pre + ".resConfUnit1")
    out = _rcu(out, sd, pre + ".resConfUnit2")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return _conv(out, sd, pre + ".out_conv")


def dpt_logits(sd, cfg: Mast3rConfig, head, toks, H, W):
    """DPTOutputAdapter_fix.forward (dpt_head.py:34-65) -> (B,4,H,W) raw xyz + conf logit."""
    p = f"downstream_head{head}.dpt"
    nh, nw = H // cfg.patch, W // cfg.patch
    layers = [toks[h].transpose(1, 2).reshape(toks[h].shape[0], -1, nh, nw) for h in cfg.hooks]
    a = p + ".act_postprocess"
    l0 = F.conv_transpose2d(_conv(layers[0], sd, a + ".0.0"), sd[a + ".0.1.weight"], sd[a + ".0.1.bias"], stride=4)
    l1 = F.conv_transpose2d(_conv(layers[1], sd, a + ".1.0"), sd[a + ".1.1.weight"], sd[a + ".1.1.bias"], stride=2)
    l2 = _conv(layers[2], sd, a + ".2.0")
    l3 = _conv(_conv(layers[3], sd, a + ".3.0"), sd, a + ".3.1", stride=2, padding=1)
    ls = [_conv(l, sd, f"{p}.scratch.layer_rn.{i}", padding=1) for i, l in enumerate((l0, l1, l2, l3))]
    path4 = _fusion(sd, p + ".scratch.refinenet4", ls[3])[:, :, : ls[2].shape[2], : ls[2].shape[3]]
    path3 = _fusion(sd, p + ".scratch.refinenet3", path4, ls[2])
    path2 = _fusion(sd, p + ".scratch.refinenet2", path3, ls[1])
    path1 = _fusion(sd, p + ".scratch.refinenet1", path2, ls[0])
    out = _conv(path1, sd, p + ".head.0", padding=1)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    out = F.relu(_conv(out, sd, p + ".head.2", padding=1))
    return _conv(out, sd, p + ".head.4")


def downstream_head(sd, cfg: Mast3rConfig, head, toks, H, W):
    """Cat_MLP_LocalFeatures_DPT_Pts3d.forward + postprocess (catmlp_dpt_head.py:25-96):
    -> dict(pts3d (B,H,W,3), conf (B,H,W), desc (B,H,W,desc_dim), desc_conf (B,H,W))"""
    pts = dpt_logits(sd, cfg, head, toks, H, W)
    cat = torch.cat((toks[0], toks[-1]), -1)
    B = cat.shape[0]
    lf = _mlp(cat, sd, f"downstream_head{head}.head_local_features")
    lf = lf.transpose(-1, -2).reshape(B, -1, H // cfg.patch, W // cfg.patch)
    lf = F.pixel_shuffle(lf, cfg.patch)
    fmap = torch.cat((pts, lf), 1).permute(0, 2, 3, 1)
    xyz = fmap[..., 0:3]
    d = xyz.norm(dim=-1, keepdim=True)
    pts3d = xyz / d.clip(min=1e-8) * torch.expm1(d)            # depth_mode ('exp', -inf, inf)
    conf = 1 + fmap[..., 3].exp()                               # conf_mode ('exp', 1, inf)
    desc = fmap[..., 4:4 + cfg.desc_dim]
    desc = desc / desc.norm(dim=-1, keepdim=True)
    desc_conf = fmap[..., 4 + cfg.desc_dim].exp()               # desc_conf_mode ('exp', 0, inf)
    return dict(pts3d=pts3d, conf=conf, desc=desc, desc_conf=desc_conf)


def init_state_dict(cfg: Mast3rConfig, seed=0, scale=1.0):
    """Seeded random weights with the upstream key layout (xavier-like, small biases) for benches
    and tests when no checkpoint is available."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * (6.0 / (i + o)) ** 0.5 * scale
        if bias:
            sd[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * 0.02

    def ln(name, d):
        sd[name + ".weight"] = 1 + 0.1 * (torch.rand(d, generator=g) * 2 - 1)
        sd[name + ".bias"] = 0.05 * (torch.rand(d, generator=g) * 2 - 1)

    def conv(name, o, i, k, bias=True, transpose=False):
        shape = (i, o, k, k) if transpose else (o, i, k, k)
        fan = i * k * k
        sd[name + ".weight"] = (torch.rand(*shape, generator=g) * 2 - 1) * (3.0 / fan) ** 0.5 * scale
        if bias:
            sd[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * 0.02

    E, Dd = cfg.enc_dim, cfg.dec_dim
    conv("patch_embed.proj", E, 3, cfg.patch)
    for i in range(cfg.enc_depth):
        p = f"enc_blocks.{i}"
        ln(p + ".norm1", E); lin(p + ".attn.qkv", 3 * E, E); lin(p + ".attn.proj", E, E)
        ln(p + ".norm2", E); lin(p + ".mlp.fc1", 4 * E, E); lin(p + ".mlp.fc2", E, 4 * E)
    ln("enc_norm", E)
    lin("decoder_embed", Dd, E)
    for blocks in ("dec_blocks", "dec_blocks2"):
        for i in range(cfg.dec_depth):
            p = f"{blocks}.{i}"
            ln(p + ".norm1", Dd); lin(p + ".attn.qkv", 3 * Dd, Dd); lin(p + ".attn.proj", Dd, Dd)
            ln(p + ".norm2", Dd); ln(p + ".norm_y", Dd)
            for q in ("projq", "projk", "projv", "proj"):
                lin(f"{p}.cross_attn.{q}", Dd, Dd)
            ln(p + ".norm3", Dd); lin(p + ".mlp.fc1", 4 * Dd, Dd); lin(p + ".mlp.fc2", Dd, 4 * Dd)
    ln("dec_norm", Dd)
    fd = cfg.feature_dim
    dims = [96, 192, 384, 768]
    for h in (1, 2):
        p = f"downstream_head{h}.dpt"
        conv(p + ".act_postprocess.0.0", dims[0], E, 1); conv(p + ".act_postprocess.0.1", dims[0], dims[0], 4, transpose=True)
        conv(p + ".act_postprocess.1.0", dims[1], Dd, 1); conv(p + ".act_postprocess.1.1", dims[1], dims[1], 2, transpose=True)
        conv(p + ".act_postprocess.2.0", dims[2], Dd, 1)
        conv(p + ".act_postprocess.3.0", dims[3], Dd, 1); conv(p + ".act_postprocess.3.1", dims[3], dims[3], 3)
        for i, d in enumerate(dims):
            conv(f"{p}.scratch.layer_rn.{i}", fd, d, 3, bias=False)
        for r in (1, 2, 3, 4):
            q = f"{p}.scratch.refinenet{r}"
            conv(q + ".out_conv", fd, fd, 1)
            for u in (1, 2):
                conv(f"{q}.resConfUnit{u}.conv1", fd, fd, 3); conv(f"{q}.resConfUnit{u}.conv2", fd, fd, 3)
        conv(p + ".head.0", fd // 2, fd, 3); conv(p + ".head.2", fd // 2, fd // 2, 3); conv(p + ".head.4", 4, fd // 2, 1)
        # keep the xyz / conf logits O(1) like a trained network's (depth ~ expm1(|xyz|)): random
        # weights otherwise drive |xyz| ~ 10 and expm1 amplifies every rounding error by e^10
        sd[p + ".head.4.weight"] *= 0.1
        idim = E + Dd
        lin(f"downstream_head{h}.head_local_features.fc1", 4 * idim, idim)
        lin(f"downstream_head{h}.head_local_features.fc2", (cfg.desc_dim + 1) * cfg.patch ** 2, 4 * idim)
    return sd
