// MASt3R two-view forward on gfx950: encoder (ViT-L, RoPE2D), asymmetric cross-attention decoder,
// DPT pointmap head + local-feature MLP head + post-processing, orchestrated natively so that one
// C-ABI call enqueues a whole stage (no per-op Python / framework overhead, graph-capturable).
//
// Reference behaviour (thirdparty/mast3r): _encode_image dust3r/dust3r/model.py:127-139;
// _decoder :171-190; Block/DecoderBlock dust3r/croco/models/blocks.py:114-191;
// DPTOutputAdapter_fix.forward dust3r/dust3r/heads/dpt_head.py:34-65 (+ croco/models/dpt_block.py);
// Cat_MLP_LocalFeatures_DPT_Pts3d.forward + postprocess mast3r/catmlp_dpt_head.py:25-96.
//
// Numerics: bf16 MFMA operands, fp32 accumulation; the residual stream, LayerNorm statistics,
// softmax and the whole post-processing are fp32.  Layout: tokens row-major [B*N, C] (== NHWC of the
// 24x32 token grid), DPT feature maps NHWC bf16.  The 1x1 out_conv of each fusion block is applied
// BEFORE the bilinear x2 upsample (both are linear and commute; 4x fewer FLOPs).
#include <stdlib.h>
#include <map>
#include <mutex>
#include <vector>
#include "common.h"
#include "gemm.h"

namespace mslam {

int launch_attention(const bf16* Q, const bf16* K, const bf16* VT, bf16* O, int batch, int heads, int nq, int nk,
                     hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (eps inside the sqrt, biased variance: torch.nn.LayerNorm), one wave
// per row, fp32 in; bf16 and/or fp32 out.
template <typename TIN>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, bf16* __restrict__ out_bf,
                                                        float* __restrict__ out_f, int rows, int D, float eps) {
  constexpr int MAXV = 32;  // D <= 2048; fully unrolled so v[] stays in registers
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const TIN* xr = x + (size_t)row * D;
  float v[MAXV];
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < MAXV; c++) {
    const int i = lane + 64 * c;
    v[c] = (i < D) ? (float)xr[i] : 0.0f;
    s += v[c];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < MAXV; c++) {
    const float d = (lane + 64 * c < D) ? v[c] - mean : 0.0f;
    q += d * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
  for (int c = 0; c < MAXV; c++) {
    const int i = lane + 64 * c;
    if (i < D) {
      const float y = (v[c] - mean) * rstd * w[i] + b[i];
      if (out_bf) out_bf[(size_t)row * D + i] = (bf16)y;
      if (out_f) out_f[(size_t)row * D + i] = y;
    }
  }
}

// Vectorised form for D == NV * 256 (the 1024 / 768 wide streams of the real model): one wave per row,
// 16-byte loads, gamma/beta fetched before the reductions so their latency overlaps them.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, bf16* __restrict__ out_bf,
                                                            float* __restrict__ out_f, int rows, float eps) {
  constexpr int D = NV * 256;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[NV], wv[NV], bv[NV];
#pragma unroll
  for (int c = 0; c < NV; c++) v[c] = xr[lane + 64 * c];
#pragma unroll
  for (int c = 0; c < NV; c++) {
    wv[c] = reinterpret_cast<const float4*>(w)[lane + 64 * c];
    bv[c] = reinterpret_cast<const float4*>(b)[lane + 64 * c];
  }
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NV; c++) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < NV; c++) {
    const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
  for (int c = 0; c < NV; c++) {
    float4 y;
    y.x = (v[c].x - mean) * rstd * wv[c].x + bv[c].x;
    y.y = (v[c].y - mean) * rstd * wv[c].y + bv[c].y;
    y.z = (v[c].z - mean) * rstd * wv[c].z + bv[c].z;
    y.w = (v[c].w - mean) * rstd * wv[c].w + bv[c].w;
    if (out_bf) {
      bf16x4 o;
      o[0] = (bf16)y.x; o[1] = (bf16)y.y; o[2] = (bf16)y.z; o[3] = (bf16)y.w;
      reinterpret_cast<bf16x4*>(out_bf + (size_t)row * D)[lane + 64 * c] = o;
    }
    if (out_f) reinterpret_cast<float4*>(out_f + (size_t)row * D)[lane + 64 * c] = y;
  }
}

// Two LayerNorms of the SAME rows with different affine parameters (decoder: norm_y of one side and norm1
// of the other both normalise the previous layer's tokens): statistics once, two bf16 outputs.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_dual_vec_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                                 const float* __restrict__ b1, bf16* __restrict__ out1,
                                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                                 bf16* __restrict__ out2, int rows, float eps) {
  constexpr int D = NV * 256;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[NV];
#pragma unroll
  for (int c = 0; c < NV; c++) v[c] = xr[lane + 64 * c];
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NV; c++) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < NV; c++) {
    const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const float* w = k ? w2 : w1;
    const float* b = k ? b2 : b1;
    bf16* out = k ? out2 : out1;
#pragma unroll
    for (int c = 0; c < NV; c++) {
      const float4 wv = reinterpret_cast<const float4*>(w)[lane + 64 * c];
      const float4 bv = reinterpret_cast<const float4*>(b)[lane + 64 * c];
      bf16x4 o;
      o[0] = (bf16)((v[c].x - mean) * rstd * wv.x + bv.x);
      o[1] = (bf16)((v[c].y - mean) * rstd * wv.y + bv.y);
      o[2] = (bf16)((v[c].z - mean) * rstd * wv.z + bv.z);
      o[3] = (bf16)((v[c].w - mean) * rstd * wv.w + bv.w);
      reinterpret_cast<bf16x4*>(out + (size_t)row * D)[lane + 64 * c] = o;
    }
  }
}

// Grouped forms for the decoder, whose two sides are stacked as rows [0, M) and [M, 2M) with their own affine
// parameters.  `single`: one output per row.  `cross`: every row yields norm1 of its own side (-> out_self, same
// row) AND norm_y of the OTHER side (-> out_mem at the other side's row block): both normalise the same tokens.
struct LnSet { const float* w; const float* b; };

template <int NV, bool CROSS>
__global__ __launch_bounds__(256) void layernorm_group_vec_kernel(const float* __restrict__ x, LnSet self0, LnSet self1,
                                                                  LnSet mem0, LnSet mem1, bf16* __restrict__ out_self,
                                                                  bf16* __restrict__ out_mem, int M, float eps) {
  constexpr int D = NV * 256;
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= 2 * M) return;
  const int side = row >= M;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
  float4 v[NV];
#pragma unroll
  for (int c = 0; c < NV; c++) v[c] = xr[lane + 64 * c];
  float s = 0.0f;
#pragma unroll
  for (int c = 0; c < NV; c++) s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)D;
  float q = 0.0f;
#pragma unroll
  for (int c = 0; c < NV; c++) {
    const float dx = v[c].x - mean, dy = v[c].y - mean, dz = v[c].z - mean, dw = v[c].w - mean;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
  for (int k = 0; k < (CROSS ? 2 : 1); k++) {
    // k = 0: this side's own norm, same row; k = 1: the other side's norm_y, written to the other side's memory block
    const LnSet ps = (k == 0) ? (side ? self1 : self0) : (side ? mem0 : mem1);
    bf16* out = (k == 0) ? out_self + (size_t)row * D : out_mem + (size_t)(side ? row - M : row + M) * D;
#pragma unroll
    for (int c = 0; c < NV; c++) {
      const float4 wv = reinterpret_cast<const float4*>(ps.w)[lane + 64 * c];
      const float4 bv = reinterpret_cast<const float4*>(ps.b)[lane + 64 * c];
      bf16x4 o;
      o[0] = (bf16)((v[c].x - mean) * rstd * wv.x + bv.x);
      o[1] = (bf16)((v[c].y - mean) * rstd * wv.y + bv.y);
      o[2] = (bf16)((v[c].z - mean) * rstd * wv.z + bv.z);
      o[3] = (bf16)((v[c].w - mean) * rstd * wv.w + bv.w);
      reinterpret_cast<bf16x4*>(out)[lane + 64 * c] = o;
    }
  }
}

static void launch_layernorm(const float* x, const float* w, const float* b, bf16* out_bf, float* out_f, int rows, int D,
                             float eps, hipStream_t s) {
  const dim3 grid((rows + 3) / 4), block(256);
  if (D == 1024) hipLaunchKernelGGL(layernorm_vec_kernel<4>, grid, block, 0, s, x, w, b, out_bf, out_f, rows, eps);
  else if (D == 768) hipLaunchKernelGGL(layernorm_vec_kernel<3>, grid, block, 0, s, x, w, b, out_bf, out_f, rows, eps);
  else hipLaunchKernelGGL(layernorm_kernel<float>, grid, block, 0, s, x, w, b, out_bf, out_f, rows, D, eps);
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 o;
    o[0] = (bf16)v.x; o[1] = (bf16)v.y; o[2] = (bf16)v.z; o[3] = (bf16)v.w;
    *reinterpret_cast<bf16x4*>(y + i) = o;
  } else {
    for (size_t k = i; k < n; k++) y[k] = (bf16)x[k];
  }
}

// img f32 [B,3,H,W] -> patches bf16 [B*nh*nw, 3*P*P], k = c*P*P + ky*P + kx (Conv2d weight order)
__global__ void patchify_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B, int H, int W, int P) {
  const int nh = H / P, nw = W / P, K = 3 * P * P;
  const size_t total = (size_t)B * nh * nw * K;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % K);
  const size_t tok = i / K;
  const int tx = (int)(tok % nw), ty = (int)((tok / nw) % nh), b = (int)(tok / ((size_t)nw * nh));
  const int c = k / (P * P), ky = (k / P) % P, kx = k % P;
  out[i] = (bf16)img[(((size_t)b * 3 + c) * H + ty * P + ky) * W + tx * P + kx];
}

// cat(enc_tok, dec_tok) along channels, bf16
__global__ void concat2_kernel(const bf16* __restrict__ a, int ca, const bf16* __restrict__ b, int cb,
                               bf16* __restrict__ out, size_t rows) {
  const int C = ca + cb;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  const size_t r = i / C;
  const int c = (int)(i % C);
  out[i] = c < ca ? a[r * ca + c] : b[r * cb + (c - ca)];
}

// bilinear x2, align_corners=True, NHWC bf16 (F.interpolate(scale_factor=2, mode='bilinear', align_corners=True))
__global__ void upsample2x_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int B, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W;
  const int c8 = C / 8;
  const size_t total = (size_t)B * Ho * Wo * c8;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cc = (int)(i % c8) * 8;
  const size_t pix = i / c8;
  const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.0f;
  const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.0f;
  const float fy = sy * oy, fx = sx * ox;
  const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
  const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
  const float wy = fy - y0, wx = fx - x0;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  const bf16* base = in + (size_t)b * H * W * C + cc;
  const bf16x8 v00 = *reinterpret_cast<const bf16x8*>(base + ((size_t)y0 * W + x0) * C);
  const bf16x8 v01 = *reinterpret_cast<const bf16x8*>(base + ((size_t)y0 * W + x1) * C);
  const bf16x8 v10 = *reinterpret_cast<const bf16x8*>(base + ((size_t)y1 * W + x0) * C);
  const bf16x8 v11 = *reinterpret_cast<const bf16x8*>(base + ((size_t)y1 * W + x1) * C);
  bf16x8 o;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float top = (float)v00[k] + wx * ((float)v01[k] - (float)v00[k]);
    const float bot = (float)v10[k] + wx * ((float)v11[k] - (float)v10[k]);
    o[k] = (bf16)(top + wy * (bot - top));
  }
  *reinterpret_cast<bf16x8*>(out + pix * C + cc) = o;
}

// Final 1x1 conv (fc -> 4) + pixel_shuffle(P) of the local-feature MLP + postprocess
// (catmlp_dpt_head.py:25-39, postprocess.py:22-58), all fp32.  One block per token = P x P pixel patch
// (P = 16, 256 threads): every global access is a full-line one.
//   phase 1: the patch's feature rows (16 px x fc bf16 = 4 KiB contiguous per image row) are read 16 B per
//            lane; the 16 lanes that share a pixel reduce their partial dot products with shuffles
//   phase 2: thread = pixel: reads its 25 local-feature values (contiguous across the patch per channel),
//            normalises, and the four outputs leave through LDS row buffers as 16-byte stores.
constexpr int kHP = 16;   // patch size the kernel is built for
__global__ __launch_bounds__(256) void head_post_kernel(const bf16* __restrict__ feat, int fc,
                                                        const float* __restrict__ w4, const float* __restrict__ b4,
                                                        const float* __restrict__ lf, int lf_ld, int desc_dim,
                                                        int B, int H, int W, float* __restrict__ X,
                                                        float* __restrict__ Cf, float* __restrict__ D,
                                                        float* __restrict__ Q) {
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  __shared__ float l_s[kHP * kHP][4];
  __shared__ __attribute__((aligned(16))) float d_s[kHP * kHP * 24];   // [row][px][desc] = the patch rows of D
  __shared__ __attribute__((aligned(16))) float x_s[kHP * kHP * 3];
  const int t = threadIdx.x;
  const int nw = W / kHP, nh = H / kHP;
  const int tok = blockIdx.x;                       // b * nh * nw + ty * nw + tx
  const int b = tok / (nh * nw), ty = (tok / nw) % nh, tx = tok % nw;
  const size_t pix00 = ((size_t)b * H + (size_t)ty * kHP) * W + (size_t)tx * kHP;   // top-left pixel of the patch
  // ---- phase 1: 1x1 conv, fc = 128: 16 lanes x 8 channels per pixel --------------------------------
  const int cg = t & 15, ppx = t >> 4;              // channel group, pixel within the patch row
  float wr[4][8];
#pragma unroll
  for (int o = 0; o < 4; o++)
#pragma unroll
    for (int k = 0; k < 8; k++) wr[o][k] = w4[o * fc + cg * 8 + k];
  for (int r = 0; r < kHP; r++) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(feat + (pix00 + (size_t)r * W + ppx) * fc + cg * 8);
    float p[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float fv = (float)v[k];
#pragma unroll
      for (int o = 0; o < 4; o++) p[o] = fmaf(wr[o][k], fv, p[o]);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
      for (int o = 0; o < 4; o++) p[o] += __shfl_xor(p[o], off, 64);
    if (cg == 0) {
#pragma unroll
      for (int o = 0; o < 4; o++) l_s[r * kHP + ppx][o] = p[o] + b4[o];
    }
  }
  __syncthreads();
  // ---- phase 2: thread = pixel (py, px) of the patch ----------------------------------------------------
  const int py = t >> 4, px = t & 15;
  const float l0 = l_s[t][0], l1 = l_s[t][1], l2 = l_s[t][2], l3 = l_s[t][3];
  const float d = sqrtf(l0 * l0 + l1 * l1 + l2 * l2);
  const float sc = expm1f(d) / fmaxf(d, 1e-8f);
  x_s[t * 3 + 0] = l0 * sc; x_s[t * 3 + 1] = l1 * sc; x_s[t * 3 + 2] = l2 * sc;
  const size_t pix = pix00 + (size_t)py * W + px;
  Cf[pix] = 1.0f + expf(l3);                                    // 16 consecutive floats per patch row
  const float* lp = lf + (size_t)tok * lf_ld + t;               // channel c of this pixel: lp[c * 256]
  float dv[24];
  float nn = 0.0f;
#pragma unroll
  for (int c = 0; c < 24; c++) { dv[c] = lp[c * kHP * kHP]; nn = fmaf(dv[c], dv[c], nn); }
  const float inv = 1.0f / sqrtf(nn);
#pragma unroll
  for (int c = 0; c < 24; c++) d_s[t * 24 + c] = dv[c] * inv;
  Q[pix] = expf(lp[24 * kHP * kHP]);
  __syncthreads();
  // patch row r of D = 16 px x 24 floats = 96 float4, contiguous in memory; 16 rows -> 1536 float4 / 256 threads
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int e = t + 256 * k, r = e / 96, q = e - r * 96;
    *reinterpret_cast<float4*>(D + (pix00 + (size_t)r * W) * 24 + q * 4) = *reinterpret_cast<const float4*>(d_s + r * 384 + q * 4);
  }
  if (t < 192) {   // X: 16 rows x 12 float4
    const int r = t / 12, q = t - r * 12;
    *reinterpret_cast<float4*>(X + (pix00 + (size_t)r * W) * 3 + q * 4) = *reinterpret_cast<const float4*>(x_s + r * 48 + q * 4);
  }
}

// Generic form (any patch size / channel count), one thread per pixel: used when the shape is not the
// production one the patch kernel above is built for.
// (catmlp_dpt_head.py:25-39, postprocess.py:22-58): one thread per pixel, all fp32.
__global__ __launch_bounds__(256) void head_post_generic_kernel(const bf16* __restrict__ feat, int fc,
                                                        const float* __restrict__ w4, const float* __restrict__ b4,
                                                        const float* __restrict__ lf, int lf_ld, int desc_dim, int P,
                                                        int B, int H, int W, float* __restrict__ X,
                                                        float* __restrict__ Cf, float* __restrict__ D,
                                                        float* __restrict__ Q) {
  const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (size_t)B * H * W) return;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
  float l[4] = {b4[0], b4[1], b4[2], b4[3]};
  const bf16* f = feat + pix * fc;
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  for (int c = 0; c < fc; c += 8) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(f + c);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const float fv = (float)v[k];
#pragma unroll
      for (int o = 0; o < 4; o++) l[o] = fmaf(w4[o * fc + c + k], fv, l[o]);
    }
  }
  const float d = sqrtf(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
  const float sc = expm1f(d) / fmaxf(d, 1e-8f);
  X[pix * 3 + 0] = l[0] * sc; X[pix * 3 + 1] = l[1] * sc; X[pix * 3 + 2] = l[2] * sc;
  Cf[pix] = 1.0f + expf(l[3]);
  const int nw = W / P;
  const size_t tok = (size_t)b * (H / P) * nw + (size_t)(y / P) * nw + x / P;
  const float* lp = lf + tok * lf_ld + (y % P) * P + (x % P);
  float dv[32];
  float nn = 0.0f;
  for (int c = 0; c < desc_dim; c++) { dv[c] = lp[c * P * P]; nn = fmaf(dv[c], dv[c], nn); }
  const float inv = 1.0f / sqrtf(nn);
  for (int c = 0; c < desc_dim; c++) D[pix * desc_dim + c] = dv[c] * inv;
  Q[pix] = expf(lp[desc_dim * P * P]);
}

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
struct Lin { const bf16* W; const float* b; int out, in; };
struct Norm { const float* w; const float* b; int d; };

struct EncBlock { Norm n1; Lin qkv, proj; Norm n2; Lin fc1, fc2; };
struct DecBlock { Norm n1; Lin qkv, proj; Norm n2, ny; Lin pq, pkv, cproj; Norm n3; Lin fc1, fc2; };
struct Rcu { Lin c1, c2; };
struct Fusion { Rcu r1, r2; Lin out; bool has_r1; };
struct Head {
  Lin a00, a01, a10, a11, a20, a30, a31;
  Lin rn[4];
  Fusion fus[4];  // index 0 = refinenet1 ... 3 = refinenet4
  Lin h0, h2;
  const float* h4w; const float* h4b;
  Lin fc1, fc2;
};

struct Mast3rModel {
  int E, enc_depth, enc_heads, Dd, dec_depth, dec_heads, P, desc_dim, fd;
  Lin pe;
  std::vector<EncBlock> enc;
  Norm enc_norm;
  Lin dec_embed;
  std::vector<DecBlock> dec[2];
  Norm dec_norm;
  Head head[2];
  float* rope_cos = nullptr;
  float* rope_sin = nullptr;
  int rope_len = 0;
  int hooks[4];
  // Second queue of a decode call (decoder side 2 / head 2 run beside side 1 / head 1) with its fork/join
  // events.  One per CALLER stream, created on first use: decode calls issued on different streams (the
  // frontend and the backend of the SLAM system run concurrently) never share a queue or an event.
  struct Fork {
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> events;
    size_t ev_next = 0;
  };
  unsigned* pf_sink = nullptr;           // 4 scratch bytes behind the RoPE tables (gemm.h: pf_sink)
  bool prefetch = true;                  // MSLAM_PREFETCH=0: no weight prefetch blocks
  // The two heads (and, when the decoder is not grouped, its two sides) of a decode call on two queues.  Pays with the
  // call ALONE on the chip (round 1: 7.3 vs 9.5 ms for four frames); inside the loop the other streams fill the chip and
  // one queue measures +4 % frames/s at 60-120 keyframes, neutral at 64 (profiles/r03_decoder_grouping_ab.log): off by
  // default since round 3 (MSLAM_TWO_STREAMS=1 turns it on; results do not depend on it)
  bool two_streams = false;
  int fork_max_rows = 1 << 30;           // MSLAM_FORK_MAX_M: calls with more token rows per side stay on one queue
  bool dec_grouped = true;               // both decoder sides per launch (MSLAM_DEC_GROUPED=0: one queue per side)
  // ... up to this many token rows per side (MSLAM_GROUP_MAX_M), two queues above.  Round 1 / 2 measured the switch-over
  // at 1 024 rows with the stage ALONE on the chip (two queues hide each other's launch bubbles: 7.3 vs 7.8 ms for a decode
  // of four frames); inside the loop, where the other streams keep the chip busy anyway, what counts is the work a launch
  // costs, and one launch over both sides does the same FLOP in ~20 % less GEMM time (6144x768x768 16 us against
  // 2 x 10 us, ...): always grouped measures +4-5 % frames/s end to end (profiles/r03_decoder_grouping_ab.log)
  int group_max_rows = 1 << 30;
  mutable std::mutex fork_mu;
  mutable std::map<hipStream_t, Fork*> forks;
  Fork* fork_for(hipStream_t caller, int& rc) const {
    std::lock_guard<std::mutex> lock(fork_mu);
    auto it = forks.find(caller);
    if (it != forks.end()) return it->second;
    Fork* f = new Fork();
    rc = check_hip(hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking), "side stream");
    for (int k = 0; k < 64 && !rc; k++) {
      hipEvent_t ev;
      rc = check_hip(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "event");
      if (!rc) f->events.push_back(ev);
    }
    if (rc) { delete f; return nullptr; }
    forks[caller] = f;
    return f;
  }
};

struct PtrFeed {
  void* const* ptrs;
  const long long* numels;
  int count, pos = 0;
  bool ok = true;
  void* take(long long expect, const char* what) {
    if (pos >= count) { if (ok) set_error("mast3r_create: ran out of weights at %s (#%d)", what, pos); ok = false; return nullptr; }
    if (numels[pos] != expect) {
      if (ok) set_error("mast3r_create: weight #%d (%s) has %lld elements, expected %lld", pos, what, numels[pos], expect);
      ok = false;
    }
    return ptrs[pos++];
  }
  Lin lin(int out, int in, bool bias, const char* what) {
    Lin l; l.out = out; l.in = in;
    l.W = (const bf16*)take((long long)out * in, what);
    l.b = bias ? (const float*)take(out, what) : nullptr;
    return l;
  }
  Norm norm(int d, const char* what) {
    Norm n; n.d = d;
    n.w = (const float*)take(d, what);
    n.b = (const float*)take(d, what);
    return n;
  }
};

// bump allocator over the caller's workspace; in dry mode only measures
struct Arena {
  char* base; size_t off = 0, cap; bool dry; bool overflow = false;
  template <typename T> T* get(size_t n) {
    const size_t bytes = (n * sizeof(T) + 255) / 256 * 256;
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += bytes;
    if (!dry && off > cap) { overflow = true; return reinterpret_cast<T*>(base); }
    return p;
  }
};

struct Ctx {
  const Mast3rModel* m;
  Arena ar;
  hipStream_t s;
  int rc = MSLAM_OK;
  Mast3rModel::Fork* fk = nullptr;
  int side = 0;   // which queue of the call c.s currently is
  bool dry() const { return ar.dry; }
  void fail(int r) { if (rc == MSLAM_OK) rc = r; }
};

// MSLAM_DEBUG=1: synchronise after every launch and name it on stderr (fault localisation only)
static void dbg(Ctx& c, const char* what, int a = 0, int b = 0, int d = 0) {
  static const bool on = getenv("MSLAM_DEBUG") != nullptr;
  if (!on || c.dry()) return;
  fprintf(stderr, "[mslam] %s %d %d %d ...", what, a, b, d);
  fflush(stderr);
  hipError_t e = hipStreamSynchronize(c.s);
  fprintf(stderr, " %s\n", hipGetErrorString(e));
}

static void run_gemm(Ctx& c, GemmArgs& g) {
  if (c.dry() || c.rc) return;
  c.fail(launch_gemm(g, c.s));
  dbg(c, g.epi == EPI_ATTN ? "gemm_attn" : (g.a_conv ? "gemm_conv" : "gemm"), g.M, g.N, g.K);
}

// weights of a later launch to stream into the Infinity Cache beside this one (gemm.h: pf_ptr)
struct Pf { const Lin* a = nullptr; const Lin* b = nullptr; };
static void set_pf(const Ctx& c, GemmArgs& g, const Pf& pf) {
  if (!c.m->prefetch) return;
  const Lin* l[2] = {pf.a, pf.b};
  for (int k = 0; k < 2; k++)
    if (l[k]) { g.pf_ptr[k] = l[k]->W; g.pf_bytes[k] = (size_t)l[k]->out * l[k]->in * sizeof(bf16); }
  g.pf_sink = c.m->pf_sink;
}

static GemmArgs dense_args(const bf16* A, int M, const Lin& l) {
  GemmArgs g = {};
  g.A = A; g.W = l.W; g.M = M; g.N = l.out; g.K = l.in; g.lda = l.in; g.bias = l.b;
  g.ldc = l.out; g.ldr1 = l.out; g.ldr2 = l.out;
  return g;
}

static void layernorm(Ctx& c, const float* x, const Norm& n, int rows, bf16* out_bf, float* out_f) {
  if (c.dry() || c.rc) return;
  launch_layernorm(x, n.w, n.b, out_bf, out_f, rows, n.d, 1e-6f, c.s);
  dbg(c, "layernorm", rows, n.d);
}

static void layernorm2(Ctx& c, const float* x, const Norm& n1, bf16* out1, const Norm& n2, bf16* out2, int rows) {
  if (c.dry() || c.rc) return;
  const dim3 grid((rows + 3) / 4), block(256);
  if (n1.d == 768 && n2.d == 768)
    hipLaunchKernelGGL(layernorm_dual_vec_kernel<3>, grid, block, 0, c.s, x, n1.w, n1.b, out1, n2.w, n2.b, out2, rows, 1e-6f);
  else if (n1.d == 1024 && n2.d == 1024)
    hipLaunchKernelGGL(layernorm_dual_vec_kernel<4>, grid, block, 0, c.s, x, n1.w, n1.b, out1, n2.w, n2.b, out2, rows, 1e-6f);
  else {
    launch_layernorm(x, n1.w, n1.b, out1, nullptr, rows, n1.d, 1e-6f, c.s);
    launch_layernorm(x, n2.w, n2.b, out2, nullptr, rows, n2.d, 1e-6f, c.s);
  }
  dbg(c, "layernorm2", rows, n1.d);
}

static void cast_bf16(Ctx& c, const float* x, bf16* y, size_t n) {
  if (c.dry() || c.rc) return;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, c.s, x, y, n);
  dbg(c, "cast", (int)n);
}

struct AttnBufs { bf16 *q, *k, *vt, *o; };

static void attn_project(Ctx& c, const bf16* A, int M, const Lin& l, int sec_base, int heads, int ntok, int kv_ntok,
                         int tok_w, const AttnBufs& ab, const Pf& pf = Pf()) {
  GemmArgs g = dense_args(A, M, l);
  set_pf(c, g, pf);
  g.epi = EPI_ATTN; g.sec_base = sec_base; g.sec_dim = heads * 64; g.heads = heads; g.ntok = ntok; g.kv_ntok = kv_ntok;
  g.tok_w = tok_w; g.q_out = ab.q; g.k_out = ab.k; g.vt_out = ab.vt; g.rope_cos = c.m->rope_cos;
  g.rope_sin = c.m->rope_sin; g.q_scale = 0.125f;  // head_dim 64 ** -0.5
  run_gemm(c, g);
}

static void attention(Ctx& c, const AttnBufs& ab, int B, int heads, int nq, int nk) {
  if (c.dry() || c.rc) return;
  c.fail(launch_attention(ab.q, ab.k, ab.vt, ab.o, B, heads, nq, nk, c.s));
  dbg(c, "attention", B * heads, nq, nk);
}

// x (f32 residual stream, [M,D]) += Linear(A) (+bias)
static void linear_residual(Ctx& c, const bf16* A, int M, const Lin& l, float* x, const Pf& pf = Pf()) {
  GemmArgs g = dense_args(A, M, l);
  set_pf(c, g, pf);
  g.res1 = x; g.res1_kind = KIND_F32; g.out = x; g.out_kind = KIND_F32;
  run_gemm(c, g);
}

static void linear_bf16(Ctx& c, const bf16* A, int M, const Lin& l, bf16* out, int act, const Pf& pf = Pf()) {
  GemmArgs g = dense_args(A, M, l);
  set_pf(c, g, pf);
  g.out = out; g.out_kind = KIND_BF16; g.act = act;
  run_gemm(c, g);
}

static void linear_f32(Ctx& c, const bf16* A, int M, const Lin& l, float* out) {
  GemmArgs g = dense_args(A, M, l);
  g.out = out; g.out_kind = KIND_F32;
  run_gemm(c, g);
}

struct BlockScratch { bf16* h; bf16* u; AttnBufs ab; };

static BlockScratch block_scratch(Ctx& c, int M, int D) {
  BlockScratch s;
  s.h = c.ar.get<bf16>((size_t)M * D);
  s.u = c.ar.get<bf16>((size_t)M * 4 * D);
  s.ab.q = c.ar.get<bf16>((size_t)M * D);
  s.ab.k = c.ar.get<bf16>((size_t)M * D);
  s.ab.vt = c.ar.get<bf16>((size_t)M * D);
  s.ab.o = c.ar.get<bf16>((size_t)M * D);
  return s;
}

static void mlp_residual(Ctx& c, float* x, int M, const Norm& n, const Lin& fc1, const Lin& fc2, BlockScratch& s,
                         const Pf& pf1 = Pf(), const Pf& pf2 = Pf()) {
  layernorm(c, x, n, M, s.h, nullptr);
  linear_bf16(c, s.h, M, fc1, s.u, ACT_GELU, pf1);
  linear_residual(c, s.u, M, fc2, x, pf2);
}

// feat_out f32 [B*N, E] (enc_norm output)
static void encode(Ctx& c, const float* img, int B, int H, int W, float* feat_out) {
  const Mast3rModel& m = *c.m;
  const int nh = H / m.P, nw = W / m.P, N = nh * nw, M = B * N, KP = 3 * m.P * m.P;
  bf16* patches = c.ar.get<bf16>((size_t)M * KP);
  float* x = c.ar.get<float>((size_t)M * m.E);
  BlockScratch s = block_scratch(c, M, m.E);
  if (!c.dry() && !c.rc) {
    const size_t total = (size_t)M * KP;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.s, img, patches, B, H, W,
                       m.P);
    dbg(c, "patchify", B, H, W);
  }
  linear_f32(c, patches, M, m.pe, x);
  for (int i = 0; i < m.enc_depth; i++) {
    const EncBlock& b = m.enc[i];
    const EncBlock* nb = i + 1 < m.enc_depth ? &m.enc[i + 1] : nullptr;   // each GEMM streams its successor's weights
    layernorm(c, x, b.n1, M, s.h, nullptr);
    attn_project(c, s.h, M, b.qkv, 0, m.enc_heads, N, N, nw, s.ab, Pf{nb ? &nb->qkv : nullptr});
    attention(c, s.ab, B, m.enc_heads, N, N);
    linear_residual(c, s.ab.o, M, b.proj, x, Pf{nb ? &nb->proj : nullptr});
    mlp_residual(c, x, M, b.n2, b.fc1, b.fc2, s, Pf{nb ? &nb->fc1 : nullptr}, Pf{nb ? &nb->fc2 : nullptr});
  }
  layernorm(c, x, m.enc_norm, M, nullptr, feat_out);
}

// ---- DPT pieces (NHWC bf16) --------------------------------------------------------------------
static GemmArgs conv_args(const bf16* in, int B, int H, int W, int C, const Lin& l, int ks, int stride) {
  GemmArgs g = {};
  const int pad = ks / 2;
  const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
  g.A = in; g.W = l.W; g.M = B * Ho * Wo; g.N = l.out; g.K = ks * ks * C; g.bias = l.b;
  g.a_conv = 1; g.cB = B; g.cH = H; g.cW = W; g.cC = C; g.cKs = ks; g.cStride = stride; g.cPad = pad; g.cHo = Ho; g.cWo = Wo;
  g.ldc = l.out; g.ldr1 = l.out; g.ldr2 = l.out; g.out_kind = KIND_BF16;
  return g;
}

static bf16* conv(Ctx& c, const bf16* in, int B, int H, int W, int C, const Lin& l, int ks, int stride, int a_relu,
                  int act, const bf16* res1, const bf16* res2) {
  GemmArgs g = conv_args(in, B, H, W, C, l, ks, stride);
  bf16* out = c.ar.get<bf16>((size_t)g.M * l.out);
  g.out = out; g.a_relu = a_relu; g.act = act;
  if (res1) { g.res1 = res1; g.res1_kind = KIND_BF16; }
  if (res2) { g.res2 = res2; g.res2_kind = KIND_BF16; }
  run_gemm(c, g);
  return out;
}

static bf16* conv_transpose(Ctx& c, const bf16* in, int B, int H, int W, int C, const Lin& l, int s, int cout) {
  GemmArgs g = {};
  g.A = in; g.W = l.W; g.M = B * H * W; g.N = cout * s * s; g.K = C; g.lda = C; g.bias = l.b;
  g.epi = EPI_CONVT; g.ct_s = s; g.ct_cout = cout; g.ct_h = H; g.ct_w = W; g.out_kind = KIND_BF16;
  bf16* out = c.ar.get<bf16>((size_t)B * H * s * W * s * cout);
  g.out = out;
  run_gemm(c, g);
  return out;
}

static bf16* upsample2x(Ctx& c, const bf16* in, int B, int H, int W, int C) {
  bf16* out = c.ar.get<bf16>((size_t)B * 4 * H * W * C);
  if (!c.dry() && !c.rc) {
    const size_t total = (size_t)B * 4 * H * W * (C / 8);
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.s, in, out, B, H, W, C);
    dbg(c, "upsample2x", H, W, C);
  }
  return out;
}

// ResidualConvUnit_custom (dpt_block.py:79-141): conv2(relu(conv1(relu(x)))) + x  [+ extra]
static bf16* rcu(Ctx& c, const bf16* x, int B, int H, int W, int C, const Rcu& r, const bf16* extra) {
  bf16* t = conv(c, x, B, H, W, C, r.c1, 3, 1, /*a_relu*/ 1, ACT_RELU, nullptr, nullptr);
  return conv(c, t, B, H, W, C, r.c2, 3, 1, 0, ACT_NONE, x, extra);
}

// FeatureFusionBlock_custom (dpt_block.py:144-218): returns the x2-upsampled, out_conv'ed map
static bf16* fusion(Ctx& c, const Fusion& f, const bf16* path, const bf16* layer, int B, int H, int W, int C) {
  const bf16* s = path;
  // NB: test the model flag, not the pointer - in the dry (sizing) pass every arena pointer is null
  if (f.has_r1) s = rcu(c, layer, B, H, W, C, f.r1, path);  // path + resConfUnit1(layer)
  bf16* r = rcu(c, s, B, H, W, C, f.r2, nullptr);
  bf16* o = conv(c, r, B, H, W, C, f.out, 1, 1, 0, ACT_NONE, nullptr, nullptr);
  return upsample2x(c, o, B, H, W, C);
}

struct HeadOut { float *X, *C, *D, *Q; };

// toks[4]: bf16 token tensors of hooks [0, 6, 9, 12]: [B*N, E], [B*N, Dd] x3
static void run_head(Ctx& c, const Head& hd, const bf16* const toks[4], int B, int H, int W, const HeadOut& out) {
  const Mast3rModel& m = *c.m;
  const int nh = H / m.P, nw = W / m.P, M = B * nh * nw;
  // act_postprocess (dpt_block.py:356-410)
  bf16* t0 = conv(c, toks[0], B, nh, nw, m.E, hd.a00, 1, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* l0 = conv_transpose(c, t0, B, nh, nw, hd.a00.out, hd.a01, 4, hd.a00.out);          // 4nh x 4nw
  bf16* t1 = conv(c, toks[1], B, nh, nw, m.Dd, hd.a10, 1, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* l1 = conv_transpose(c, t1, B, nh, nw, hd.a10.out, hd.a11, 2, hd.a10.out);          // 2nh x 2nw
  bf16* l2 = conv(c, toks[2], B, nh, nw, m.Dd, hd.a20, 1, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* t3 = conv(c, toks[3], B, nh, nw, m.Dd, hd.a30, 1, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* l3 = conv(c, t3, B, nh, nw, hd.a30.out, hd.a31, 3, 2, 0, ACT_NONE, nullptr, nullptr);  // ceil(nh/2)
  const int h3 = (nh + 2 - 3) / 2 + 1, w3 = (nw + 2 - 3) / 2 + 1;
  // scratch.layer_rn (3x3, no bias)
  bf16* r0 = conv(c, l0, B, 4 * nh, 4 * nw, hd.a00.out, hd.rn[0], 3, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* r1 = conv(c, l1, B, 2 * nh, 2 * nw, hd.a10.out, hd.rn[1], 3, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* r2 = conv(c, l2, B, nh, nw, hd.a20.out, hd.rn[2], 3, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* r3 = conv(c, l3, B, h3, w3, hd.a30.out, hd.rn[3], 3, 1, 0, ACT_NONE, nullptr, nullptr);
  if (2 * h3 != nh || 2 * w3 != nw) {
    if (!c.dry()) set_error("mast3r head: token grid %dx%d must be even", nh, nw);
    c.fail(MSLAM_EINVAL);
    return;
  }
  const int fd = m.fd;
  bf16* p4 = fusion(c, hd.fus[3], r3, nullptr, B, h3, w3, fd);          // -> nh x nw
  bf16* p3 = fusion(c, hd.fus[2], p4, r2, B, nh, nw, fd);               // -> 2nh
  bf16* p2 = fusion(c, hd.fus[1], p3, r1, B, 2 * nh, 2 * nw, fd);       // -> 4nh
  bf16* p1 = fusion(c, hd.fus[0], p2, r0, B, 4 * nh, 4 * nw, fd);       // -> 8nh
  // head (dpt_block.py:316-324)
  bf16* h0 = conv(c, p1, B, 8 * nh, 8 * nw, fd, hd.h0, 3, 1, 0, ACT_NONE, nullptr, nullptr);
  bf16* hu = upsample2x(c, h0, B, 8 * nh, 8 * nw, hd.h0.out);           // H x W
  bf16* h2 = conv(c, hu, B, H, W, hd.h0.out, hd.h2, 3, 1, 0, ACT_RELU, nullptr, nullptr);
  // local features MLP on cat(enc, dec_last)
  const int idim = m.E + m.Dd;
  bf16* cat = c.ar.get<bf16>((size_t)M * idim);
  bf16* hid = c.ar.get<bf16>((size_t)M * hd.fc1.out);
  float* lf = c.ar.get<float>((size_t)M * hd.fc2.out);
  if (!c.dry() && !c.rc) {
    const size_t total = (size_t)M * idim;
    hipLaunchKernelGGL(concat2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c.s, toks[0], m.E, toks[3],
                       m.Dd, cat, (size_t)M);
    dbg(c, "concat2", M);
  }
  linear_bf16(c, cat, M, hd.fc1, hid, ACT_GELU);
  linear_f32(c, hid, M, hd.fc2, lf);
  if (!c.dry() && !c.rc) {
    if (m.P == kHP && hd.h2.out == 128 && m.desc_dim == 24) {
      hipLaunchKernelGGL(head_post_kernel, dim3((unsigned)(B * (H / m.P) * (W / m.P))), dim3(256), 0, c.s, h2, hd.h2.out,
                         hd.h4w, hd.h4b, lf, hd.fc2.out, m.desc_dim, B, H, W, out.X, out.C, out.D, out.Q);
    } else {
      const size_t npix = (size_t)B * H * W;
      hipLaunchKernelGGL(head_post_generic_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, c.s, h2, hd.h2.out,
                         hd.h4w, hd.h4b, lf, hd.fc2.out, m.desc_dim, m.P, B, H, W, out.X, out.C, out.D, out.Q);
    }
    dbg(c, "head_post", B, H, W);
  }
}

static void dec_block(Ctx& c, const DecBlock& b, const DecBlock* nb, float* x, const bf16* yn, int B, int N, int Nk, int nw,
                      int nw_k, BlockScratch& s) {
  const Mast3rModel& m = *c.m;
  const int M = B * N, Mk = B * Nk;
  // s.h already holds norm1(x): computed together with the other side's norm_y (layernorm2 in decode())
  attn_project(c, s.h, M, b.qkv, 0, m.dec_heads, N, N, nw, s.ab, Pf{nb ? &nb->qkv : nullptr});
  attention(c, s.ab, B, m.dec_heads, N, N);
  linear_residual(c, s.ab.o, M, b.proj, x, Pf{nb ? &nb->proj : nullptr});
  layernorm(c, x, b.n2, M, s.h, nullptr);
  attn_project(c, s.h, M, b.pq, 0, m.dec_heads, N, Nk, nw, s.ab, Pf{nb ? &nb->pq : nullptr});
  attn_project(c, yn, Mk, b.pkv, 1, m.dec_heads, N, Nk, nw_k, s.ab, Pf{nb ? &nb->pkv : nullptr});   // [projk; projv] stacked
  attention(c, s.ab, B, m.dec_heads, N, Nk);
  linear_residual(c, s.ab.o, M, b.cproj, x, Pf{nb ? &nb->cproj : nullptr});
  mlp_residual(c, x, M, b.n3, b.fc1, b.fc2, s, Pf{nb ? &nb->fc1 : nullptr}, Pf{nb ? &nb->fc2 : nullptr});
}

// feat1/feat2 f32 [B*N, E]; outputs for both sides; dec_last (optional) f32 [2][B*N, Dd]
// fork/join between the caller's stream and the model's side stream (no-ops in the sizing pass)
static void stream_wait(Ctx& c, hipStream_t waiter, hipStream_t on) {
  if (c.dry() || c.rc) return;
  hipEvent_t ev = c.fk->events[c.fk->ev_next++ % c.fk->events.size()];
  c.fail(check_hip(hipEventRecord(ev, on), "hipEventRecord"));
  c.fail(check_hip(hipStreamWaitEvent(waiter, ev, 0), "hipStreamWaitEvent"));
}

// ---- grouped decoder layers: both sides of a layer in ONE launch per operation --------------------
// Activations of side s live at rows [s*M, (s+1)*M) of every buffer (x, h, u, q/k/vt/o, yn).
static void group2(GemmArgs& g, const Lin& l1, size_t a_gstride, size_t out_gbytes) {
  g.groups = 2; g.W1 = l1.W; g.bias1 = l1.b; g.a_gstride = a_gstride; g.out_gbytes = out_gbytes;
}

static void g_layernorm(Ctx& c, const float* x, const Norm& n0, const Norm& n1, int M, bf16* out) {
  if (c.dry() || c.rc) return;
  const LnSet s0{n0.w, n0.b}, s1{n1.w, n1.b};
  const dim3 grid((2 * M + 3) / 4), block(256);
  if (n0.d == 768) hipLaunchKernelGGL((layernorm_group_vec_kernel<3, false>), grid, block, 0, c.s, x, s0, s1, s0, s1, out, out, M, 1e-6f);
  else if (n0.d == 1024) hipLaunchKernelGGL((layernorm_group_vec_kernel<4, false>), grid, block, 0, c.s, x, s0, s1, s0, s1, out, out, M, 1e-6f);
  else {
    launch_layernorm(x, n0.w, n0.b, out, nullptr, M, n0.d, 1e-6f, c.s);
    launch_layernorm(x + (size_t)M * n0.d, n1.w, n1.b, out + (size_t)M * n0.d, nullptr, M, n0.d, 1e-6f, c.s);
  }
  dbg(c, "g_layernorm", M, n0.d);
}

// h[s] = norm1_s(x[s]);  yn[s] = norm_y_s(x[1-s])  (the memory side s attends to)
static void g_layernorm_cross(Ctx& c, const float* x, const DecBlock& b0, const DecBlock& b1, int M, bf16* h, bf16* yn) {
  if (c.dry() || c.rc) return;
  const int D = b0.n1.d;
  const dim3 grid((2 * M + 3) / 4), block(256);
  const LnSet self0{b0.n1.w, b0.n1.b}, self1{b1.n1.w, b1.n1.b}, mem0{b0.ny.w, b0.ny.b}, mem1{b1.ny.w, b1.ny.b};
  if (D == 768) hipLaunchKernelGGL((layernorm_group_vec_kernel<3, true>), grid, block, 0, c.s, x, self0, self1, mem0, mem1, h, yn, M, 1e-6f);
  else if (D == 1024) hipLaunchKernelGGL((layernorm_group_vec_kernel<4, true>), grid, block, 0, c.s, x, self0, self1, mem0, mem1, h, yn, M, 1e-6f);
  else {
    const size_t MD = (size_t)M * D;
    launch_layernorm(x, b0.n1.w, b0.n1.b, h, nullptr, M, D, 1e-6f, c.s);
    launch_layernorm(x + MD, b1.n1.w, b1.n1.b, h + MD, nullptr, M, D, 1e-6f, c.s);
    launch_layernorm(x + MD, b0.ny.w, b0.ny.b, yn, nullptr, M, D, 1e-6f, c.s);        // memory of side 0 = side 1's tokens
    launch_layernorm(x, b1.ny.w, b1.ny.b, yn + MD, nullptr, M, D, 1e-6f, c.s);
  }
  dbg(c, "g_layernorm_cross", M, D);
}

static void g_attn_project(Ctx& c, const bf16* A, int M, const Lin& l0, const Lin& l1, int sec_base, int heads, int B,
                           int ntok, int kv_ntok, int tok_w, const AttnBufs& ab, const Pf& pf = Pf()) {
  GemmArgs g = dense_args(A, M, l0);
  set_pf(c, g, pf);
  g.epi = EPI_ATTN; g.sec_base = sec_base; g.sec_dim = heads * 64; g.heads = heads; g.ntok = ntok; g.kv_ntok = kv_ntok;
  g.tok_w = tok_w; g.q_out = ab.q; g.k_out = ab.k; g.vt_out = ab.vt; g.rope_cos = c.m->rope_cos;
  g.rope_sin = c.m->rope_sin; g.q_scale = 0.125f;
  group2(g, l1, (size_t)M * l0.in, 0);
  g.qkv_gstride = (size_t)B * heads * 64 * (sec_base == 0 ? ntok : kv_ntok);
  run_gemm(c, g);
}

static void g_linear_residual(Ctx& c, const bf16* A, int M, const Lin& l0, const Lin& l1, float* x, const Pf& pf = Pf()) {
  GemmArgs g = dense_args(A, M, l0);
  set_pf(c, g, pf);
  g.res1 = x; g.res1_kind = KIND_F32; g.out = x; g.out_kind = KIND_F32;
  group2(g, l1, (size_t)M * l0.in, (size_t)M * l0.out * sizeof(float));
  g.res1_gbytes = g.out_gbytes;
  run_gemm(c, g);
}

static void g_linear_bf16(Ctx& c, const bf16* A, int M, const Lin& l0, const Lin& l1, bf16* out, int act,
                          const Pf& pf = Pf()) {
  GemmArgs g = dense_args(A, M, l0);
  set_pf(c, g, pf);
  g.out = out; g.out_kind = KIND_BF16; g.act = act;
  group2(g, l1, (size_t)M * l0.in, (size_t)M * l0.out * sizeof(bf16));
  run_gemm(c, g);
}

// one decoder layer for both sides; x [2M, Dd] f32; scratch s sized for 2M rows
static void dec_layer_grouped(Ctx& c, const DecBlock& b0, const DecBlock& b1, const DecBlock* n0, const DecBlock* n1,
                              float* x, bf16* yn, int B, int N, int nw, BlockScratch& s) {
  const Mast3rModel& m = *c.m;
  const int M = B * N;
#define MSLAM_PF(field) (n0 ? Pf{&n0->field, &n1->field} : Pf{})   /* the next layer's matrices of both sides */
  g_layernorm_cross(c, x, b0, b1, M, s.h, yn);
  g_attn_project(c, s.h, M, b0.qkv, b1.qkv, 0, m.dec_heads, B, N, N, nw, s.ab, MSLAM_PF(qkv));
  attention(c, s.ab, 2 * B, m.dec_heads, N, N);
  g_linear_residual(c, s.ab.o, M, b0.proj, b1.proj, x, MSLAM_PF(proj));
  g_layernorm(c, x, b0.n2, b1.n2, M, s.h);
  g_attn_project(c, s.h, M, b0.pq, b1.pq, 0, m.dec_heads, B, N, N, nw, s.ab, MSLAM_PF(pq));
  g_attn_project(c, yn, M, b0.pkv, b1.pkv, 1, m.dec_heads, B, N, N, nw, s.ab, MSLAM_PF(pkv));
  attention(c, s.ab, 2 * B, m.dec_heads, N, N);
  g_linear_residual(c, s.ab.o, M, b0.cproj, b1.cproj, x, MSLAM_PF(cproj));
  g_layernorm(c, x, b0.n3, b1.n3, M, s.h);
  g_linear_bf16(c, s.h, M, b0.fc1, b1.fc1, s.u, ACT_GELU, MSLAM_PF(fc1));
  g_linear_residual(c, s.u, M, b0.fc2, b1.fc2, x, MSLAM_PF(fc2));
#undef MSLAM_PF
}

// feat1/feat2 f32 [B*N, E]; outputs for both sides; dec_last (optional) f32 [2][B*N, Dd].
// The two sides of a decoder layer are independent (both read the PREVIOUS layer's outputs,
// dust3r/model.py:178-183) and at one image per side neither fills 256 CUs, so side 1 runs on the
// caller's stream and side 2 on the model's side stream, joined once per layer; the two heads likewise.
static void decode(Ctx& c, const float* feat1, const float* feat2, int B, int H, int W, const HeadOut out[2],
                   float* dec_last1, float* dec_last2) {
  const Mast3rModel& m = *c.m;
  const int nh = H / m.P, nw = W / m.P, N = nh * nw, M = B * N;
  const float* feat[2] = {feat1, feat2};
  float* dec_last[2] = {dec_last1, dec_last2};
  if (!c.dry() && m.two_streams && !c.rc && M <= m.fork_max_rows) {
    int frc = MSLAM_OK;
    c.fk = m.fork_for(c.s, frc);
    c.fail(frc);
  }
  hipStream_t sA = c.s, sB = c.fk ? c.fk->side : c.s;
  hipStream_t st[2] = {sA, sB};
  // every per-side buffer is one allocation with side s at rows [s*M, (s+1)*M)
  const size_t ME = (size_t)M * m.E, MD = (size_t)M * m.Dd;
  bf16* fb_all = c.ar.get<bf16>(2 * ME);
  float* x_all = c.ar.get<float>(2 * MD);
  bf16* yn_all = c.ar.get<bf16>(2 * MD);
  bf16* tok_all[4] = {fb_all, c.ar.get<bf16>(2 * MD), c.ar.get<bf16>(2 * MD), c.ar.get<bf16>(2 * MD)};
  BlockScratch bs_all = block_scratch(c, 2 * M, m.Dd);
  bf16* fb[2];
  float* x[2];
  bf16* yn[2];
  bf16* tok[2][4];
  BlockScratch bs[2];
  for (int s = 0; s < 2; s++) {
    fb[s] = c.dry() ? nullptr : fb_all + s * ME;
    x[s] = c.dry() ? nullptr : x_all + s * MD;
    yn[s] = c.dry() ? nullptr : yn_all + s * MD;
    tok[s][0] = fb[s];
    for (int k = 1; k < 4; k++) tok[s][k] = c.dry() ? nullptr : tok_all[k] + s * MD;
    bs[s] = bs_all;
    if (!c.dry()) {
      bs[s].h += s * MD; bs[s].u += s * 4 * MD;
      bs[s].ab.q += s * MD; bs[s].ab.k += s * MD; bs[s].ab.vt += s * MD; bs[s].ab.o += s * MD;
    }
  }
  // grouping pays while one side alone does not fill the chip (measured: 3.42 -> 3.29 ms at one image per side,
  // 7.7 -> 8.1 ms at four)
  if (m.dec_grouped && M <= m.group_max_rows) {
    // ---- both sides per launch, one queue (heads fork below) ----------------------------------------
    cast_bf16(c, feat[0], fb[0], ME);
    cast_bf16(c, feat[1], fb[1], ME);
    linear_f32(c, fb_all, 2 * M, m.dec_embed, x_all);          // decoder_embed is shared by the two sides
    for (int l = 0; l < m.dec_depth; l++) {
      const bool more = l + 1 < m.dec_depth;
      dec_layer_grouped(c, m.dec[0][l], m.dec[1][l], more ? &m.dec[0][l + 1] : nullptr, more ? &m.dec[1][l + 1] : nullptr,
                        x_all, yn_all, B, N, nw, bs_all);
      for (int k = 1; k < 3; k++)
        if (l + 1 == m.hooks[k]) cast_bf16(c, x_all, tok_all[k], 2 * MD);
    }
    if (sB != sA) stream_wait(c, sB, sA);  // fork for the heads
  } else {
  if (sB != sA) stream_wait(c, sB, sA);  // fork: side stream starts after everything already queued
  for (int s = 0; s < 2; s++) {
    c.s = st[s]; c.side = s;
    cast_bf16(c, feat[s], fb[s], (size_t)M * m.E);
    linear_f32(c, fb[s], M, m.dec_embed, x[s]);
  }
  for (int l = 0; l < m.dec_depth; l++) {
    c.s = sA; c.side = 0;
    if (sB != sA) stream_wait(c, sA, sB);                     // x[1] of the previous layer is final
    layernorm2(c, x[1], m.dec[0][l].ny, yn[0], m.dec[1][l].n1, bs[1].h, M);   // memory for side 1 = norm_y(f2); side 2's norm1
    layernorm2(c, x[0], m.dec[1][l].ny, yn[1], m.dec[0][l].n1, bs[0].h, M);   // memory for side 2 = norm_y(f1); side 1's norm1
    if (sB != sA) stream_wait(c, sB, sA);                     // memories ready; x[0] final for side 2's reads
    for (int s = 0; s < 2; s++) {
      c.s = st[s]; c.side = s;
      dec_block(c, m.dec[s][l], l + 1 < m.dec_depth ? &m.dec[s][l + 1] : nullptr, x[s], yn[s], B, N, N, nw, nw, bs[s]);
      for (int k = 1; k < 3; k++)
        if (l + 1 == m.hooks[k]) cast_bf16(c, x[s], tok[s][k], (size_t)M * m.Dd);
    }
  }
  }
  const size_t mark = c.ar.off;
  size_t end = mark;
  for (int s = 0; s < 2; s++) {
    c.s = st[s]; c.side = s;
    layernorm(c, x[s], m.dec_norm, M, tok[s][3], dec_last[s]);
    // each head gets its own scratch region when the two run concurrently
    c.ar.off = (sB != sA || c.dry()) ? end : mark;
    run_head(c, m.head[s], tok[s], B, H, W, out[s]);
    end = c.ar.off;
  }
  c.s = sA; c.side = 0;
  if (sB != sA) stream_wait(c, sA, sB);  // join
}

}  // namespace mslam

using namespace mslam;

extern "C" int mslam_mast3r_create(void** handle_out, const int* cfg9, void* const* weight_ptrs,
                                   const long long* weight_numels, int n_weights, void* stream) {
  MSLAM_REQUIRE(handle_out && cfg9 && weight_ptrs && weight_numels, "mast3r_create: null pointer");
  Mast3rModel* m = new Mast3rModel();
  m->E = cfg9[0]; m->enc_depth = cfg9[1]; m->enc_heads = cfg9[2]; m->Dd = cfg9[3]; m->dec_depth = cfg9[4];
  m->dec_heads = cfg9[5]; m->P = cfg9[6]; m->desc_dim = cfg9[7]; m->fd = cfg9[8];
  if (m->E != m->enc_heads * 64 || m->Dd != m->dec_heads * 64 || m->desc_dim > 31 || m->P < 1 || m->fd % 8) {
    set_error("mast3r_create: head_dim must be 64 (enc %d/%d, dec %d/%d), desc_dim <= 31", m->E, m->enc_heads, m->Dd,
              m->dec_heads);
    delete m;
    return MSLAM_EINVAL;
  }
  const int l2 = m->dec_depth;
  m->hooks[0] = 0; m->hooks[1] = l2 * 2 / 4; m->hooks[2] = l2 * 3 / 4; m->hooks[3] = l2;
  PtrFeed f{weight_ptrs, weight_numels, n_weights};
  const int E = m->E, D = m->Dd, P = m->P;
  m->pe = f.lin(E, 3 * P * P, true, "patch_embed");
  for (int i = 0; i < m->enc_depth; i++) {
    EncBlock b;
    b.n1 = f.norm(E, "enc.norm1"); b.qkv = f.lin(3 * E, E, true, "enc.qkv"); b.proj = f.lin(E, E, true, "enc.proj");
    b.n2 = f.norm(E, "enc.norm2"); b.fc1 = f.lin(4 * E, E, true, "enc.fc1"); b.fc2 = f.lin(E, 4 * E, true, "enc.fc2");
    m->enc.push_back(b);
  }
  m->enc_norm = f.norm(E, "enc_norm");
  m->dec_embed = f.lin(D, E, true, "decoder_embed");
  for (int s = 0; s < 2; s++)
    for (int i = 0; i < m->dec_depth; i++) {
      DecBlock b;
      b.n1 = f.norm(D, "dec.norm1"); b.qkv = f.lin(3 * D, D, true, "dec.qkv"); b.proj = f.lin(D, D, true, "dec.proj");
      b.n2 = f.norm(D, "dec.norm2"); b.ny = f.norm(D, "dec.norm_y");
      b.pq = f.lin(D, D, true, "dec.projq"); b.pkv = f.lin(2 * D, D, true, "dec.projkv");
      b.cproj = f.lin(D, D, true, "dec.cross_proj");
      b.n3 = f.norm(D, "dec.norm3"); b.fc1 = f.lin(4 * D, D, true, "dec.fc1"); b.fc2 = f.lin(D, 4 * D, true, "dec.fc2");
      m->dec[s].push_back(b);
    }
  m->dec_norm = f.norm(D, "dec_norm");
  const int dims[4] = {96, 192, 384, 768};
  const int fd = m->fd;
  for (int s = 0; s < 2; s++) {
    Head& h = m->head[s];
    h.a00 = f.lin(dims[0], E, true, "act0.0"); h.a01 = f.lin(dims[0] * 16, dims[0], false, "act0.1");
    h.a01.b = (const float*)f.take(dims[0], "act0.1.bias");
    h.a10 = f.lin(dims[1], D, true, "act1.0"); h.a11 = f.lin(dims[1] * 4, dims[1], false, "act1.1");
    h.a11.b = (const float*)f.take(dims[1], "act1.1.bias");
    h.a20 = f.lin(dims[2], D, true, "act2.0");
    h.a30 = f.lin(dims[3], D, true, "act3.0"); h.a31 = f.lin(dims[3], 9 * dims[3], true, "act3.1");
    for (int k = 0; k < 4; k++) h.rn[k] = f.lin(fd, 9 * dims[k], false, "layer_rn");
    for (int k = 3; k >= 0; k--) {  // refinenet4 first
      Fusion& fu = h.fus[k];
      fu.has_r1 = (k != 3);
      if (fu.has_r1) { fu.r1.c1 = f.lin(fd, 9 * fd, true, "rcu1.conv1"); fu.r1.c2 = f.lin(fd, 9 * fd, true, "rcu1.conv2"); }
      fu.r2.c1 = f.lin(fd, 9 * fd, true, "rcu2.conv1"); fu.r2.c2 = f.lin(fd, 9 * fd, true, "rcu2.conv2");
      fu.out = f.lin(fd, fd, true, "out_conv");
    }
    h.h0 = f.lin(fd / 2, 9 * fd, true, "head.0"); h.h2 = f.lin(fd / 2, 9 * (fd / 2), true, "head.2");
    h.h4w = (const float*)f.take(4 * (fd / 2), "head.4.weight"); h.h4b = (const float*)f.take(4, "head.4.bias");
    const int idim = E + D;
    h.fc1 = f.lin(4 * idim, idim, true, "lf.fc1"); h.fc2 = f.lin((m->desc_dim + 1) * P * P, 4 * idim, true, "lf.fc2");
  }
  if (!f.ok) { delete m; return MSLAM_EINVAL; }
  if (f.pos != n_weights) {
    set_error("mast3r_create: %d weights supplied, %d consumed", n_weights, f.pos);
    delete m;
    return MSLAM_EINVAL;
  }
  // RoPE2D tables (pos_embed.py:120-130): inv_freq_i = base^(-2i/32), i < 16; base fixed at 100
  m->rope_len = 1024;
  std::vector<float> hc((size_t)m->rope_len * 16), hs((size_t)m->rope_len * 16);
  for (int p = 0; p < m->rope_len; p++)
    for (int i = 0; i < 16; i++) {
      const float inv_freq = 1.0f / powf(100.0f, (float)(2 * i) / 32.0f);
      const float fr = (float)p * inv_freq;
      hc[(size_t)p * 16 + i] = cosf(fr);
      hs[(size_t)p * 16 + i] = sinf(fr);
    }
  int rc = check_hip(hipMalloc(&m->rope_cos, hc.size() * 4 + 16), "rope hipMalloc");
  if (!rc) m->pf_sink = reinterpret_cast<unsigned*>(m->rope_cos + hc.size());
  if (!rc) rc = check_hip(hipMalloc(&m->rope_sin, hs.size() * 4), "rope hipMalloc");
  if (!rc) rc = check_hip(hipMemcpyAsync(m->rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream), "rope copy");
  if (!rc) rc = check_hip(hipMemcpyAsync(m->rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, (hipStream_t)stream), "rope copy");
  if (!rc) rc = check_hip(hipStreamSynchronize((hipStream_t)stream), "rope sync");
  if (const char* e = getenv("MSLAM_TWO_STREAMS")) m->two_streams = atoi(e) != 0;
  if (getenv("MSLAM_SINGLE_STREAM")) m->two_streams = false;
  if (const char* e = getenv("MSLAM_FORK_MAX_M")) m->fork_max_rows = atoi(e);
  if (const char* e = getenv("MSLAM_PREFETCH")) m->prefetch = atoi(e) != 0;
  if (const char* e = getenv("MSLAM_DEC_GROUPED")) m->dec_grouped = atoi(e) != 0;
  if (const char* e = getenv("MSLAM_GROUP_MAX_M")) m->group_max_rows = atoi(e);
  if (rc) { delete m; return rc; }
  *handle_out = m;
  return MSLAM_OK;
}

extern "C" int mslam_mast3r_destroy(void* handle) {
  Mast3rModel* m = (Mast3rModel*)handle;
  if (!m) return MSLAM_OK;
  if (m->rope_cos) (void)hipFree(m->rope_cos);
  if (m->rope_sin) (void)hipFree(m->rope_sin);
  for (auto& kv : m->forks) {
    for (hipEvent_t ev : kv.second->events) (void)hipEventDestroy(ev);
    (void)hipStreamDestroy(kv.second->side);
    delete kv.second;
  }
  delete m;
  return MSLAM_OK;
}

static int check_shape(const Mast3rModel* m, int B, int H, int W, const char* who) {
  MSLAM_REQUIRE(m, "%s: null model", who);
  MSLAM_REQUIRE(B >= 1 && H >= m->P && W >= m->P && H % m->P == 0 && W % m->P == 0,
                "%s: image %dx%d must be a positive multiple of the patch size %d", who, H, W, m->P);
  const int N = (H / m->P) * (W / m->P);
  MSLAM_REQUIRE(N % 8 == 0, "%s: token count %d must be a multiple of 8", who, N);
  MSLAM_REQUIRE(m->E <= 2048 && m->Dd <= 2048, "%s: embedding dims above 2048 unsupported", who);
  MSLAM_REQUIRE(H / m->P < m->rope_len && W / m->P < m->rope_len, "%s: token grid exceeds the RoPE table", who);
  return MSLAM_OK;
}

extern "C" size_t mslam_mast3r_workspace_bytes(void* handle, int batch, int H, int W) {
  const Mast3rModel* m = (const Mast3rModel*)handle;
  if (!m || check_shape(m, batch, H, W, "mast3r_workspace_bytes")) return 0;
  Ctx c{m, Arena{nullptr, 0, 0, true}, nullptr};
  encode(c, nullptr, batch, H, W, nullptr);
  const size_t enc = c.ar.off;
  c.ar.off = 0;
  HeadOut out[2] = {};
  decode(c, nullptr, nullptr, batch, H, W, out, nullptr, nullptr);
  return (enc > c.ar.off ? enc : c.ar.off) + 4096;
}

extern "C" int mslam_mast3r_encode(void* handle, const float* img, int batch, int H, int W, float* feat_out,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  const Mast3rModel* m = (const Mast3rModel*)handle;
  int rc = check_shape(m, batch, H, W, "mast3r_encode");
  if (rc) return rc;
  MSLAM_REQUIRE(img && feat_out && workspace, "mast3r_encode: null pointer");
  MSLAM_REQUIRE(workspace_bytes >= mslam_mast3r_workspace_bytes(handle, batch, H, W), "mast3r_encode: workspace too small");
  Ctx c{m, Arena{(char*)workspace, 0, workspace_bytes, false}, (hipStream_t)stream};
  encode(c, img, batch, H, W, feat_out);
  MSLAM_REQUIRE(!c.ar.overflow, "mast3r_encode: workspace arena overflow (%zu > %zu)", c.ar.off, c.ar.cap);
  if (c.rc) return c.rc;
  return check_hip(hipGetLastError(), "mast3r_encode launch");
}

extern "C" int mslam_mast3r_decode(void* handle, const float* feat1, const float* feat2, int batch, int H, int W,
                                   float* X1, float* C1, float* D1, float* Q1, float* X2, float* C2, float* D2,
                                   float* Q2, float* dec_last1, float* dec_last2, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  const Mast3rModel* m = (const Mast3rModel*)handle;
  int rc = check_shape(m, batch, H, W, "mast3r_decode");
  if (rc) return rc;
  MSLAM_REQUIRE(feat1 && feat2 && X1 && C1 && D1 && Q1 && X2 && C2 && D2 && Q2 && workspace, "mast3r_decode: null pointer");
  MSLAM_REQUIRE(workspace_bytes >= mslam_mast3r_workspace_bytes(handle, batch, H, W), "mast3r_decode: workspace too small");
  Ctx c{m, Arena{(char*)workspace, 0, workspace_bytes, false}, (hipStream_t)stream};
  HeadOut out[2] = {{X1, C1, D1, Q1}, {X2, C2, D2, Q2}};
  decode(c, feat1, feat2, batch, H, W, out, dec_last1, dec_last2);
  MSLAM_REQUIRE(!c.ar.overflow, "mast3r_decode: workspace arena overflow (%zu > %zu)", c.ar.off, c.ar.cap);
  if (c.rc) return c.rc;
  return check_hip(hipGetLastError(), "mast3r_decode launch");
}

// ---- kernel-level entry points (unit tests, roofline measurement) -------------------------------
extern "C" int mslam_gemm_tile_override(int M, int N, int K, int cfg) {
  MSLAM_REQUIRE(M != 0 && N > 0 && K > 0 && cfg >= 0, "gemm_tile_override: bad arguments");
  return gemm_tile_override(M < 0 ? -M : M, N, K, cfg, M < 0);
}

extern "C" int mslam_gemm_profile_begin(int M, int N, int K, int max_samples) {
  return gemm_profile_begin(M, N, K, max_samples);
}

extern "C" int mslam_gemm_profile_end(double* avg_us, double* min_us, int* samples) {
  return gemm_profile_end(avg_us, min_us, samples);
}

extern "C" int mslam_gemm_bf16(const void* A, const void* Wt, const float* bias, const void* residual_f32, void* out,
                               int M, int N, int K, int act, int out_is_bf16, void* stream) {
  MSLAM_REQUIRE(A && Wt && out, "gemm_bf16: null pointer");
  GemmArgs g = {};
  g.A = (const bf16*)A; g.W = (const bf16*)Wt; g.M = M; g.N = N; g.K = K; g.lda = K; g.bias = bias; g.act = act;
  g.out = out; g.out_kind = out_is_bf16 ? KIND_BF16 : KIND_F32; g.ldc = N; g.ldr1 = N; g.ldr2 = N;
  if (residual_f32) { g.res1 = residual_f32; g.res1_kind = KIND_F32; }
  return launch_gemm(g, (hipStream_t)stream);
}

// NHWC bf16 convolution (ks in {1,3}, stride in {1,2}, zero pad ks/2) as implicit GEMM; W [Cout, ks*ks*Cin]
extern "C" int mslam_conv2d_nhwc_bf16(const void* in, const void* Wt, const float* bias, const void* residual_bf16,
                                      void* out_bf16, int B, int H, int Wd, int Cin, int Cout, int ks, int stride,
                                      int relu_in, int act, void* stream) {
  MSLAM_REQUIRE(in && Wt && out_bf16, "conv2d: null pointer");
  MSLAM_REQUIRE((ks == 1 || ks == 3) && (stride == 1 || stride == 2), "conv2d: unsupported ks/stride");
  Lin l; l.W = (const bf16*)Wt; l.b = bias; l.out = Cout; l.in = ks * ks * Cin;
  GemmArgs g = conv_args((const bf16*)in, B, H, Wd, Cin, l, ks, stride);
  g.out = out_bf16; g.a_relu = relu_in; g.act = act;
  if (residual_bf16) { g.res1 = residual_bf16; g.res1_kind = KIND_BF16; }
  return launch_gemm(g, (hipStream_t)stream);
}

// softmax(Q K^T) V with Q,K [B,H,N,64] (q pre-scaled), V^T [B,H,64,Nk] -> O [B,Nq,H*64], all bf16
extern "C" int mslam_attention_bf16(const void* Q, const void* K, const void* VT, void* O, int batch, int heads,
                                    int nq, int nk, void* stream) {
  MSLAM_REQUIRE(Q && K && VT && O, "attention: null pointer");
  return launch_attention((const bf16*)Q, (const bf16*)K, (const bf16*)VT, (bf16*)O, batch, heads, nq, nk,
                          (hipStream_t)stream);
}

extern "C" int mslam_layernorm_f32(const float* x, const float* w, const float* b, void* out_bf16, float* out_f32,
                                   int rows, int D, float eps, void* stream) {
  MSLAM_REQUIRE(x && w && b && (out_bf16 || out_f32), "layernorm: null pointer");
  MSLAM_REQUIRE(D <= 2048 && rows > 0, "layernorm: D=%d must be <= 2048", D);
  launch_layernorm(x, w, b, (bf16*)out_bf16, out_f32, rows, D, eps, (hipStream_t)stream);
  MSLAM_LAUNCH_CHECK("layernorm");
  return MSLAM_OK;
}
