"""HIP-backed stand-in for the reference's ``AsymmetricMASt3R`` model object, exposing exactly the three
methods the SLAM code calls (mast3r_slam/mast3r_utils.py:36-39,57-64,74):

    model._encode_image(img, true_shape) -> (feat, pos, None)
    model._decoder(feat1, pos1, feat2, pos2) -> (dec1, dec2)        13-entry token lists
    model._downstream_head(k, tokens, shape) -> dict(pts3d, conf, desc, desc_conf)

The whole forward runs in libmslam_hip.so (csrc/mast3r.hip).  Decoder and both heads are ONE native
call; `_decoder` therefore performs that call and `_downstream_head` hands out the matching cached
result, so the reference's call sequence works unchanged while nothing is computed twice.
Weights use the upstream state-dict key layout (thirdparty/mast3r README checkpoint).
"""
import ctypes
import os

import torch

import mslam_hip as _m


class Mast3rConfig:
    def __init__(self, enc_dim=1024, enc_depth=24, enc_heads=16, dec_dim=768, dec_depth=12, dec_heads=12,
                 patch=16, desc_dim=24, feature_dim=256):
        self.enc_dim, self.enc_depth, self.enc_heads = enc_dim, enc_depth, enc_heads
        self.dec_dim, self.dec_depth, self.dec_heads = dec_dim, dec_depth, dec_heads
        self.patch, self.desc_dim, self.feature_dim = patch, desc_dim, feature_dim

    def as_list(self):
        return [self.enc_dim, self.enc_depth, self.enc_heads, self.dec_dim, self.dec_depth, self.dec_heads,
                self.patch, self.desc_dim, self.feature_dim]


def canonical_weights(sd, cfg, device):
    """Flatten an upstream state dict into the order csrc/mast3r.hip::mslam_mast3r_create consumes.
    Matrices -> bf16 [out, in]; 3x3 convs -> [out, (ky,kx,cin)]; ConvTranspose(k=s) -> [(cout,i,j), cin];
    biases, LayerNorm parameters and the last 1x1 conv stay fp32."""
    out = []

    def get(name):
        if name not in sd:
            alt = name.replace("dec_blocks2.", "dec_blocks.")          # dust3r/model.py:90-97
            for i in range(4):                                           # dpt_block.py:68-73 aliases
                alt = alt.replace(f"scratch.layer_rn.{i}.", f"scratch.layer{i + 1}_rn.")
            if alt not in sd:
                raise KeyError(f"checkpoint is missing {name}")
            name = alt
        return sd[name]

    def mat(name):
        w = get(name + ".weight").float()
        if w.ndim == 4:
            w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)           # conv: tap-major, channel-minor
        out.append(w.to(device=device, dtype=torch.bfloat16).contiguous())

    def vec(name):
        out.append(get(name).to(device=device, dtype=torch.float32).contiguous())

    def lin(name, bias=True):
        mat(name)
        if bias:
            vec(name + ".bias")

    def norm(name):
        vec(name + ".weight")
        vec(name + ".bias")

    def convT(name):
        w = get(name + ".weight").float()                               # (cin, cout, s, s)
        out.append(w.permute(1, 2, 3, 0).reshape(-1, w.shape[0]).to(device=device, dtype=torch.bfloat16).contiguous())
        vec(name + ".bias")

    w = get("patch_embed.proj.weight").float()
    out.append(w.reshape(w.shape[0], -1).to(device=device, dtype=torch.bfloat16).contiguous())
    vec("patch_embed.proj.bias")
    for i in range(cfg.enc_depth):
        p = f"enc_blocks.{i}"
        norm(p + ".norm1"); lin(p + ".attn.qkv"); lin(p + ".attn.proj")
        norm(p + ".norm2"); lin(p + ".mlp.fc1"); lin(p + ".mlp.fc2")
    norm("enc_norm")
    lin("decoder_embed")
    for blocks in ("dec_blocks", "dec_blocks2"):
        for i in range(cfg.dec_depth):
            p = f"{blocks}.{i}"
            norm(p + ".norm1"); lin(p + ".attn.qkv"); lin(p + ".attn.proj")
            norm(p + ".norm2"); norm(p + ".norm_y")
            lin(f"{p}.cross_attn.projq")
            # projk and projv read the same memory tokens: one [2D, D] matrix, one launch
            wk, wv = get(f"{p}.cross_attn.projk.weight").float(), get(f"{p}.cross_attn.projv.weight").float()
            out.append(torch.cat((wk, wv), 0).to(device=device, dtype=torch.bfloat16).contiguous())
            out.append(torch.cat((get(f"{p}.cross_attn.projk.bias"), get(f"{p}.cross_attn.projv.bias")), 0)
                       .to(device=device, dtype=torch.float32).contiguous())
            lin(f"{p}.cross_attn.proj")
            norm(p + ".norm3"); lin(p + ".mlp.fc1"); lin(p + ".mlp.fc2")
    norm("dec_norm")
    for h in (1, 2):
        p = f"downstream_head{h}.dpt"
        a = p + ".act_postprocess"
        lin(a + ".0.0"); convT(a + ".0.1")
        lin(a + ".1.0"); convT(a + ".1.1")
        lin(a + ".2.0")
        lin(a + ".3.0"); lin(a + ".3.1")
        for i in range(4):
            lin(f"{p}.scratch.layer_rn.{i}", bias=False)
        for r in (4, 3, 2, 1):
            q = f"{p}.scratch.refinenet{r}"
            if r != 4:  # refinenet4 is called with one input: resConfUnit1 is never used (dpt_block.py:194-204)
                lin(q + ".resConfUnit1.conv1"); lin(q + ".resConfUnit1.conv2")
            lin(q + ".resConfUnit2.conv1"); lin(q + ".resConfUnit2.conv2")
            lin(q + ".out_conv")
        lin(p + ".head.0"); lin(p + ".head.2")
        w4 = get(p + ".head.4.weight").float()
        out.append(w4.reshape(w4.shape[0], -1).to(device=device, dtype=torch.float32).contiguous())
        vec(p + ".head.4.bias")
        lin(f"downstream_head{h}.head_local_features.fc1"); lin(f"downstream_head{h}.head_local_features.fc2")
    return out


class Mast3rHIP:
    def __init__(self, state_dict, cfg: Mast3rConfig = None, device="cuda", use_graphs=False):
        """use_graphs: capture each (call, batch, H, W) once into a HIP graph and replay it (the forward is
        ~800 short launches per frame, so the host launch path matters); inputs are copied into the graph's
        static buffers and results are returned as fresh tensors, so call semantics do not change."""
        self.cfg = cfg or Mast3rConfig()
        self.device = torch.device(device)
        self.use_graphs = bool(use_graphs)
        self._graphs = {}
        self._pos_cache = {}
        self._weights = canonical_weights(state_dict, self.cfg, self.device)
        n = len(self._weights)
        ptrs = (ctypes.c_void_p * n)(*[w.data_ptr() for w in self._weights])
        numels = (ctypes.c_longlong * n)(*[w.numel() for w in self._weights])
        cfg9 = (ctypes.c_int * 9)(*self.cfg.as_list())
        handle = ctypes.c_void_p()
        rc = _m.lib().mslam_mast3r_create(ctypes.byref(handle), cfg9, ptrs, numels, n, _m.stream_ptr())
        _m.check(rc, "mast3r_create")
        self._h = handle
        self._ws = None
        self._head_cache = {}

    def __del__(self):
        try:
            _m.lib().mslam_mast3r_destroy(self._h)
        except Exception:
            pass

    # reference API plumbing -----------------------------------------------------------------
    def share_memory(self):   # main.py:217 (weights already live in device memory)
        return self

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def _workspace(self, B, H, W, kind="dec"):
        """Arena for one call, one per (kind, calling stream): the encoder of the next frame and the
        backend's batched decode may run on other streams while the current frame is decoded."""
        need = _m.lib().mslam_mast3r_workspace_bytes(self._h, B, H, W)
        if need == 0:
            _m.check(-1, "mast3r_workspace_bytes")
        if self._ws is None:
            self._ws = {}
        key = (kind, _m.stream_ptr())
        if key not in self._ws or self._ws[key].numel() < need:
            self._ws[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws[key]

    def positions(self, B, H, W):
        """PositionGetter (croco/models/blocks.py:195-207): (B, N, 2) int64 [y, x]."""
        key = (B, H, W)
        if key not in self._pos_cache:   # constant per shape: built once, handed out as a fresh copy
            nh, nw = H // self.cfg.patch, W // self.cfg.patch
            y, x = torch.meshgrid(torch.arange(nh, device=self.device), torch.arange(nw, device=self.device), indexing="ij")
            self._pos_cache[key] = torch.stack((y.reshape(-1), x.reshape(-1)), -1)[None].expand(B, -1, 2).contiguous()
        return self._pos_cache[key].clone()

    @torch.inference_mode()
    def _encode_image(self, image, true_shape=None):
        """dust3r/model.py:127-139 -> (feat f32[B,N,E], pos i64[B,N,2], None)."""
        image = image.to(self.device, torch.float32).contiguous()
        B, _, H, W = image.shape
        N = (H // self.cfg.patch) * (W // self.cfg.patch)
        self._last_hw = (H, W)
        if self.use_graphs:
            return self._replay(("enc", B, H, W), (image,))[0].clone(), self.positions(B, H, W), None
        feat = torch.empty((B, N, self.cfg.enc_dim), dtype=torch.float32, device=self.device)
        self._run_encode(image, feat, self._workspace(B, H, W, "enc"))
        return feat, self.positions(B, H, W), None

    def _run_encode(self, image, feat, ws):
        B, _, H, W = image.shape
        rc = _m.lib().mslam_mast3r_encode(self._h, _m.ptr(image), B, H, W, _m.ptr(feat), _m.ptr(ws), ws.numel(),
                                         _m.stream_ptr())
        _m.check(rc, "mast3r_encode")

    def _run_decode(self, feat1, feat2, H, W, outs, d1, d2, ws):
        a, b = outs
        rc = _m.lib().mslam_mast3r_decode(
            self._h, _m.ptr(feat1), _m.ptr(feat2), feat1.shape[0], H, W, _m.ptr(a["pts3d"]), _m.ptr(a["conf"]),
            _m.ptr(a["desc"]), _m.ptr(a["desc_conf"]), _m.ptr(b["pts3d"]), _m.ptr(b["conf"]), _m.ptr(b["desc"]),
            _m.ptr(b["desc_conf"]), _m.ptr(d1), _m.ptr(d2), _m.ptr(ws), ws.numel(), _m.stream_ptr())
        _m.check(rc, "mast3r_decode")

    def _decode_buffers(self, B, H, W, N):
        dd = self.cfg.desc_dim
        f32 = dict(dtype=torch.float32, device=self.device)
        outs = [dict(pts3d=torch.empty((B, H, W, 3), **f32), conf=torch.empty((B, H, W), **f32),
                     desc=torch.empty((B, H, W, dd), **f32), desc_conf=torch.empty((B, H, W), **f32)) for _ in range(2)]
        return outs, torch.empty((B, N, self.cfg.dec_dim), **f32), torch.empty((B, N, self.cfg.dec_dim), **f32)

    def _replay(self, key, inputs):
        """Graph cache: first use runs once eagerly (one-time kernel attribute calls), then captures."""
        ent = self._graphs.get(key)
        if ent is None:
            kind, B, H, W = key
            N = (H // self.cfg.patch) * (W // self.cfg.patch)
            need = _m.lib().mslam_mast3r_workspace_bytes(self._h, B, H, W)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)   # private: graphs may replay concurrently
            static_in = [torch.empty_like(x) for x in inputs]
            if kind == "enc":
                feat = torch.empty((B, N, self.cfg.enc_dim), dtype=torch.float32, device=self.device)
                run = lambda: self._run_encode(static_in[0], feat, ws)
                outs = (feat,)
            else:
                o, d1, d2 = self._decode_buffers(B, H, W, N)
                run = lambda: self._run_decode(static_in[0], static_in[1], H, W, o, d1, d2, ws)
                outs = (o, d1, d2)
            for dst, src in zip(static_in, inputs):
                dst.copy_(src)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                run()
            torch.cuda.current_stream(self.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):   # other host threads keep launching
                run()
            ent = self._graphs[key] = (g, static_in, outs, ws)
        g, static_in, outs, _ = ent
        for dst, src in zip(static_in, inputs):
            dst.copy_(src)
        g.replay()
        return outs

    @torch.inference_mode()
    def decode_pair(self, feat1, feat2, H, W, return_tokens=False):
        """_decoder + both heads in one native call.  Returns (res1, res2[, dec_last1, dec_last2])."""
        feat1 = feat1.to(self.device, torch.float32).contiguous()
        feat2 = feat2.to(self.device, torch.float32).contiguous()
        B, N = feat1.shape[0], feat1.shape[1]
        if self.use_graphs:
            (a, b), d1, d2 = self._replay(("dec", B, H, W), (feat1, feat2))
            a, b = {k: v.clone() for k, v in a.items()}, {k: v.clone() for k, v in b.items()}
            d1, d2 = d1.clone(), d2.clone()
        else:
            (a, b), d1, d2 = self._decode_buffers(B, H, W, N)
            self._run_decode(feat1, feat2, H, W, (a, b), d1, d2, self._workspace(B, H, W))
        if return_tokens:
            return a, b, d1, d2
        return a, b

    def _decoder(self, f1, pos1, f2, pos2):
        """dust3r/model.py:171-190.  Returns two lists of dec_depth+1 tensors like the reference; only
        entries 0 (encoder tokens) and -1 (dec_norm'ed last tokens) hold distinct data, which is all
        the SLAM code reads before passing the lists to _downstream_head."""
        nw = int(pos1[..., 1].max().item()) + 1
        nh = int(pos1[..., 0].max().item()) + 1
        H, W = nh * self.cfg.patch, nw * self.cfg.patch
        r1, r2, d1, d2 = self.decode_pair(f1, f2, H, W, return_tokens=True)
        self._head_cache = {d1.data_ptr(): r1, d2.data_ptr(): r2}
        n = self.cfg.dec_depth + 1
        return [f1] + [d1] * (n - 1), [f2] + [d2] * (n - 1)

    def _downstream_head(self, head_num, decout, img_shape):
        """dust3r/model.py:192-196: hands out the result computed by the fused decode."""
        key = decout[-1].data_ptr()
        if key not in self._head_cache:
            raise RuntimeError("_downstream_head: tokens do not come from the last _decoder call of this model")
        return self._head_cache[key]


def load_mast3r_state_dict(path):
    """Checkpoint dict {'args': Namespace, 'model': state_dict} (mast3r/model.py:24-34); only the tensor
    part is read, with a loader that executes nothing from the file."""
    if not os.path.isfile(path):
        raise FileNotFoundError(
            f"{path}: MASt3R checkpoint not found (the reference downloads it, README.md:63-65; no network here)")
    # weights_only=True; the only non-tensor object of an upstream checkpoint is ckpt["args"], an argparse.Namespace
    # (a plain attribute container), which is allow-listed explicitly - nothing from the file is executed
    import argparse

    with torch.serialization.safe_globals([argparse.Namespace]):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
    return ckpt["model"] if "model" in ckpt else ckpt


def random_state_dict(cfg: Mast3rConfig, seed=0):
    """Seeded random weights with the upstream key layout (no checkpoint is available offline; the
    bench states this in its `data`/`config` fields).  Xavier-style scales keep activations O(1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * (6.0 / (i + o)) ** 0.5
        if bias:
            sd[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * 0.02

    def ln(name, d):
        sd[name + ".weight"] = 1 + 0.1 * (torch.rand(d, generator=g) * 2 - 1)
        sd[name + ".bias"] = 0.05 * (torch.rand(d, generator=g) * 2 - 1)

    def conv(name, o, i, k, bias=True, transpose=False):
        shape = (i, o, k, k) if transpose else (o, i, k, k)
        sd[name + ".weight"] = (torch.rand(*shape, generator=g) * 2 - 1) * (3.0 / (i * k * k)) ** 0.5
        if bias:
            sd[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * 0.02

    E, D, P = cfg.enc_dim, cfg.dec_dim, cfg.patch
    conv("patch_embed.proj", E, 3, P)
    for i in range(cfg.enc_depth):
        p = f"enc_blocks.{i}"
        ln(p + ".norm1", E); lin(p + ".attn.qkv", 3 * E, E); lin(p + ".attn.proj", E, E)
        ln(p + ".norm2", E); lin(p + ".mlp.fc1", 4 * E, E); lin(p + ".mlp.fc2", E, 4 * E)
    ln("enc_norm", E)
    lin("decoder_embed", D, E)
    for blocks in ("dec_blocks", "dec_blocks2"):
        for i in range(cfg.dec_depth):
            p = f"{blocks}.{i}"
            ln(p + ".norm1", D); lin(p + ".attn.qkv", 3 * D, D); lin(p + ".attn.proj", D, D)
            ln(p + ".norm2", D); ln(p + ".norm_y", D)
            for q in ("projq", "projk", "projv", "proj"):
                lin(f"{p}.cross_attn.{q}", D, D)
            ln(p + ".norm3", D); lin(p + ".mlp.fc1", 4 * D, D); lin(p + ".mlp.fc2", D, 4 * D)
    ln("dec_norm", D)
    fd, dims = cfg.feature_dim, [96, 192, 384, 768]
    for h in (1, 2):
        p = f"downstream_head{h}.dpt"
        a = p + ".act_postprocess"
        conv(a + ".0.0", dims[0], E, 1); conv(a + ".0.1", dims[0], dims[0], 4, transpose=True)
        conv(a + ".1.0", dims[1], D, 1); conv(a + ".1.1", dims[1], dims[1], 2, transpose=True)
        conv(a + ".2.0", dims[2], D, 1)
        conv(a + ".3.0", dims[3], D, 1); conv(a + ".3.1", dims[3], dims[3], 3)
        for i, d in enumerate(dims):
            conv(f"{p}.scratch.layer_rn.{i}", fd, d, 3, bias=False)
        for r in (1, 2, 3, 4):
            q = f"{p}.scratch.refinenet{r}"
            conv(q + ".out_conv", fd, fd, 1)
            for u in (1, 2):
                conv(f"{q}.resConfUnit{u}.conv1", fd, fd, 3); conv(f"{q}.resConfUnit{u}.conv2", fd, fd, 3)
        conv(p + ".head.0", fd // 2, fd, 3); conv(p + ".head.2", fd // 2, fd // 2, 3); conv(p + ".head.4", 4, fd // 2, 1)
        sd[p + ".head.4.weight"] *= 0.1   # keep xyz / conf logits O(1) (expm1 / exp follow)
        idim = E + D
        lin(f"downstream_head{h}.head_local_features.fc1", 4 * idim, idim)
        lin(f"downstream_head{h}.head_local_features.fc2", (cfg.desc_dim + 1) * P * P, 4 * idim)
    return sd
