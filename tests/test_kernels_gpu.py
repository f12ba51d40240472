"""GPU kernel-level numerics tests (-m gpu): each hand-written HIP building block of the MASt3R forward
against a plain PyTorch fp32 reference of the same op on the same bf16-rounded inputs.
Tolerance: fp32 accumulation of bf16 products -> rel-L2 <= 2e-3 (bf16 output rounding 4e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _lib():
    import mslam_hip as m

    return m


@pytest.mark.parametrize("M,N,K", [(24, 128, 128), (160, 6400, 1792), (768, 3072, 1024), (768, 1024, 4096),
                                   (6144, 4096, 1024), (100, 96, 1024), (2048, 2304, 768), (768, 7168, 1792)])
@pytest.mark.parametrize("act,out_bf16", [(0, 0), (1, 1), (2, 1)])
def test_gemm_bf16(device, M, N, K, act, out_bf16):
    m = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(device)
    Wt = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16).to(device)
    bias = (torch.rand(N, generator=g) - 0.5).to(device)
    res = torch.rand(M, N, generator=g).to(device)
    out = torch.empty((M, N), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=device)
    rc = m.lib().mslam_gemm_bf16(m.ptr(A), m.ptr(Wt), m.ptr(bias), m.ptr(res), m.ptr(out), M, N, K, act, out_bf16,
                                 m.stream_ptr())
    m.check(rc, "gemm")
    ref = A.float() @ Wt.float().T + bias
    ref = F.gelu(ref) if act == 1 else (F.relu(ref) if act == 2 else ref)
    ref = ref + res
    assert _rel(out.float(), ref) <= (5e-3 if out_bf16 else 2e-3), _rel(out.float(), ref)


@pytest.mark.parametrize("M,N,K", [(3072, 3072, 1024), (3072, 768, 3072), (1000, 200, 520), (3072, 4096, 1024)])
def test_gemm_result_does_not_depend_on_the_tile(device, M, N, K):
    """Every tile configuration accumulates an output element over K in the same order, so the choice (heuristic,
    measured table, mslam_gemm_tile_override) never changes a bit of the result - what makes batched network calls
    bit-identical to one-frame calls."""
    m = _lib()
    g = torch.Generator().manual_seed(5)
    A = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).to(device)
    Wt = ((torch.rand(N, K, generator=g) * 2 - 1) / K ** 0.5).to(torch.bfloat16).to(device)
    bias = (torch.rand(N, generator=g) - 0.5).to(device)
    outs = []
    try:
        for cfg in (0, 642, 643, 644, 1262, 1263, 1242, 1282, 1283, 2128, 2256, 2192, 8256):   # 8256: gemm8p.hip
            m.check(m.lib().mslam_gemm_tile_override(M, N, K, cfg), "override")
            out = torch.empty((M, N), dtype=torch.float32, device=device)
            m.check(m.lib().mslam_gemm_bf16(m.ptr(A), m.ptr(Wt), m.ptr(bias), 0, m.ptr(out), M, N, K, 1, 0, m.stream_ptr()), "gemm")
            outs.append(out)
    finally:
        m.lib().mslam_gemm_tile_override(M, N, K, 0)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks,stride", [(1, 24, 32, 768, 256, 3, 1), (2, 12, 16, 96, 256, 3, 1),
                                                       (1, 24, 32, 768, 768, 3, 2), (2, 24, 32, 1024, 96, 1, 1),
                                                       (1, 96, 128, 256, 128, 3, 1), (1, 9, 13, 64, 40, 3, 1)])
def test_conv2d_implicit_gemm(device, B, H, W, Cin, Cout, ks, stride):
    m = _lib()
    g = torch.Generator().manual_seed(H * W + Cin)
    x = (torch.rand(B, Cin, H, W, generator=g) * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(Cout, Cin, ks, ks, generator=g) * 2 - 1) / (Cin * ks * ks) ** 0.5).to(torch.bfloat16)
    bias = torch.rand(Cout, generator=g) - 0.5
    ref = F.conv2d(F.relu(x.float()), w.float(), bias, stride=stride, padding=ks // 2)
    Ho, Wo = ref.shape[-2:]
    res = (torch.rand(B, Ho, Wo, Cout, generator=g)).to(torch.bfloat16)
    ref = (F.relu(ref).permute(0, 2, 3, 1) + res.float())
    xin = x.permute(0, 2, 3, 1).contiguous().to(device)
    wk = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(device)
    out = torch.empty((B, Ho, Wo, Cout), dtype=torch.bfloat16, device=device)
    bias_d, res_d = bias.to(device), res.to(device)   # keep the device tensors alive across the launch
    rc = m.lib().mslam_conv2d_nhwc_bf16(m.ptr(xin), m.ptr(wk), m.ptr(bias_d), m.ptr(res_d), m.ptr(out),
                                        B, H, W, Cin, Cout, ks, stride, 1, 2, m.stream_ptr())
    m.check(rc, "conv2d")
    assert _rel(out.float().cpu(), ref) <= 6e-3, _rel(out.float().cpu(), ref)


@pytest.mark.parametrize("B,Hh,Nq,Nk", [(1, 2, 24, 24), (2, 3, 80, 80), (1, 16, 768, 768), (2, 12, 768, 768),
                                        (1, 4, 200, 72)])
def test_attention(device, B, Hh, Nq, Nk):
    m = _lib()
    g = torch.Generator().manual_seed(Nq + Nk)
    q = (torch.randn(B, Hh, Nq, 64, generator=g) * 0.125 * 1.5).to(torch.bfloat16)   # pre-scaled by d^-1/2
    k = (torch.randn(B, Hh, Nk, 64, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(B, Hh, Nk, 64, generator=g).to(torch.bfloat16)
    k[0, 0, 3] *= 6.0   # a spiked key: exercises the online-softmax rescale across tiles
    ref = torch.softmax(q.float() @ k.float().transpose(-1, -2), -1) @ v.float()
    ref = ref.transpose(1, 2).reshape(B, Nq, Hh * 64)
    vt = v.transpose(-1, -2).contiguous()
    out = torch.empty((B, Nq, Hh * 64), dtype=torch.bfloat16, device=device)
    qd, kd, vd = q.to(device), k.to(device), vt.to(device)
    rc = m.lib().mslam_attention_bf16(m.ptr(qd), m.ptr(kd), m.ptr(vd), m.ptr(out), B, Hh,
                                      Nq, Nk, m.stream_ptr())
    m.check(rc, "attention")
    assert _rel(out.float().cpu(), ref) <= 1e-2, _rel(out.float().cpu(), ref)


@pytest.mark.parametrize("rows,D", [(768, 1024), (160, 256), (7, 768), (33, 2048)])
def test_layernorm(device, rows, D):
    m = _lib()
    g = torch.Generator().manual_seed(D)
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    w, b = torch.rand(D, generator=g) + 0.5, torch.rand(D, generator=g) - 0.5
    of = torch.empty((rows, D), dtype=torch.float32, device=device)
    ob = torch.empty((rows, D), dtype=torch.bfloat16, device=device)
    xd, wd, bd = x.to(device), w.to(device), b.to(device)
    rc = m.lib().mslam_layernorm_f32(m.ptr(xd), m.ptr(wd), m.ptr(bd), m.ptr(ob), m.ptr(of),
                                     rows, D, 1e-6, m.stream_ptr())
    m.check(rc, "layernorm")
    ref = F.layer_norm(x, (D,), w, b, 1e-6)
    np.testing.assert_allclose(of.cpu().numpy(), ref.numpy(), atol=2e-5, rtol=1e-5)
    assert _rel(ob.float().cpu(), ref) <= 4e-3
