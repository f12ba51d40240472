// Fused softmax(Q K^T) V for the MASt3R encoder / decoder (head_dim 64, self and cross attention).
//
// Reference: Attention.forward / CrossAttention.forward, dust3r/croco/models/blocks.py:95-112,149-169
// (materialises the (heads, N, N) fp32 score tensor).  Here: flash-style, one pass over K/V tiles staged
// through LDS, online softmax, nothing of size N x N ever reaches HBM.
//
// Wave64 / MFMA mapping (v_mfma_f32_32x32x16_bf16), one wave = 32 query rows:
//   S^T = K . Q^T   -> accumulator has the QUERY on the lane axis and 16 keys in registers, so the
//                      row-wise max / sum is 16 in-lane ops + one exchange with lane^32
//   O^T = V^T . P^T -> P^T is consumed straight from the S^T accumulator registers as the MFMA
//                      B operand (accumulator-as-operand, k order permuted: key 16s+8(j>>2)+4h+(j&3)),
//                      the matching A operand comes from the V^T tile with two 8-byte LDS reads
//   q arrives RoPE-rotated and pre-scaled by d^-1/2, k RoPE-rotated, v transposed per head: all three
//   are written in that form by the projection GEMM's epilogue (gemm.hip, EPI_ATTN).
#include "common.h"
#include "gemm.h"

namespace mslam {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int KV_TILE = 64;
constexpr int K_ROW = 72;   // bf16 per LDS row of the K tile (ds_read_b128, 144-byte stride)
constexpr int VT_ROW = 68;  // bf16 per LDS row of the V^T tile (ds_read_b64, 136-byte stride)

__global__ __launch_bounds__(128) void attention_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                        const bf16* __restrict__ VT, bf16* __restrict__ O,
                                                        int heads, int nq, int nk) {
  __shared__ __attribute__((aligned(16))) bf16 Ks[2][KV_TILE][K_ROW];
  __shared__ __attribute__((aligned(16))) bf16 VTs[2][64][VT_ROW];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int h = lane >> 5, lq = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const size_t bh = (size_t)b * heads + head;
  const int q_row = blockIdx.x * 64 + wid * 32 + lq;
  const bool q_ok = q_row < nq;
  const bf16* Kb = K + bh * (size_t)nk * 64;
  const bf16* VTb = VT + bh * 64 * (size_t)nk;

  // Q^T fragments (B operand): lane = query, 8 consecutive features per k-step
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (q_ok) v = *reinterpret_cast<const u32x4*>(Q + (bh * nq + q_row) * 64 + 16 * s + 8 * h);
    qf[s] = __builtin_bit_cast(bf16x8, v);
  }

  u32x4 kreg[4], vreg[4];
  auto load_kv = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = t + i * 128;          // 512 chunks of 16 bytes per tile
      const int row = c >> 3, part = c & 7;
      u32x4 kv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
      if (kv0 + row < nk) kv = *reinterpret_cast<const u32x4*>(Kb + (size_t)(kv0 + row) * 64 + part * 8);
      if (kv0 + part * 8 < nk) vv = *reinterpret_cast<const u32x4*>(VTb + (size_t)row * nk + kv0 + part * 8);
      kreg[i] = kv;
      vreg[i] = vv;
    }
  };
  auto store_kv = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = t + i * 128;
      const int row = c >> 3, part = c & 7;
      *reinterpret_cast<u32x4*>(&Ks[buf][row][part * 8]) = kreg[i];
      // 136-byte rows are only 8-byte aligned: two 8-byte stores
      const uint2 lo = make_uint2(vreg[i][0], vreg[i][1]), hi = make_uint2(vreg[i][2], vreg[i][3]);
      *reinterpret_cast<uint2*>(&VTs[buf][row][part * 8]) = lo;
      *reinterpret_cast<uint2*>(&VTs[buf][row][part * 8 + 4]) = hi;
    }
  };

  f32x16 o_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) o_acc[dt][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;
  const float kLog2e = 1.4426950408889634f;

  const int ntiles = (nk + KV_TILE - 1) / KV_TILE;
  load_kv(0);
  store_kv(0);
  __syncthreads();
  for (int kt = 0; kt < ntiles; kt++) {
    const int cur = kt & 1, kv0 = kt * KV_TILE;
    if (kt + 1 < ntiles) load_kv(kv0 + KV_TILE);

    // ---- S^T = K . Q^T for the two 32-key sub-tiles -----------------------------------------
    f32x16 s_acc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s_acc[u][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&Ks[cur][32 * u + lq][16 * s + 8 * h]);
        s_acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_acc[u], 0, 0, 0);
      }
    }
    // ---- online softmax: the lane owns one query; keys are in registers ------------------------
    float m_tile = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int kv = kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float sv = (kv < nk) ? s_acc[u][r] * kLog2e : -INFINITY;
        s_acc[u][r] = sv;
        m_tile = fmaxf(m_tile, sv);
      }
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = (m_run == -INFINITY) ? 0.0f : __builtin_amdgcn_exp2f(m_run - m_new);
    float l_tile = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(s_acc[u][r] - m_new);  // exp2(-inf) = 0 for masked keys
        s_acc[u][r] = p;
        l_tile += p;
      }
    l_tile += __shfl_xor(l_tile, 32, 64);
    l_run = l_run * alpha + l_tile;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int r = 0; r < 16; r++) o_acc[dt][r] *= alpha;

    // ---- O^T += V^T . P^T  (P^T straight from the S^T accumulator registers) --------------------
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; j++) pf[j] = (bf16)s_acc[u][8 * s2 + j];
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          const bf16* vrow = &VTs[cur][32 * dt + lq][32 * u + 16 * s2 + 4 * h];
          const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow);
          const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + 8);
          bf16x8 vf;
          vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
          vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[dt], 0, 0, 0);
        }
      }
    }
    if (kt + 1 < ntiles) store_kv(cur ^ 1);
    __syncthreads();
  }

  // ---- O[b, q, head*64 + d] = O^T / l ----------------------------------------------------------
  if (q_ok) {
    const float inv = 1.0f / l_run;
    bf16* orow = O + ((size_t)b * nq + q_row) * ((size_t)heads * 64) + (size_t)head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; dt++)
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        bf16x4 pk;
#pragma unroll
        for (int j = 0; j < 4; j++) pk[j] = (bf16)(o_acc[dt][4 * gq + j] * inv);
        *reinterpret_cast<bf16x4*>(orow + 32 * dt + 8 * gq + 4 * h) = pk;
      }
  }
}

int launch_attention(const bf16* Q, const bf16* K, const bf16* VT, bf16* O, int batch, int heads, int nq, int nk,
                     hipStream_t stream) {
  MSLAM_REQUIRE(nq > 0 && nk > 0 && batch > 0 && heads > 0, "attention: empty problem");
  MSLAM_REQUIRE(nk % 8 == 0, "attention: key count %d must be a multiple of 8", nk);
  dim3 grid((nq + 63) / 64, heads, batch);
  hipLaunchKernelGGL(attention_kernel, grid, dim3(128), 0, stream, Q, K, VT, O, heads, nq, nk);
  return check_hip(hipGetLastError(), "attention launch");
}

}  // namespace mslam
