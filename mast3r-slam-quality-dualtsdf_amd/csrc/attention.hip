// Fused softmax(Q K^T) V for the MASt3R encoder / decoder (head_dim 64, self and cross attention).
//
// Reference: Attention.forward / CrossAttention.forward, dust3r/croco/models/blocks.py:95-112,149-169
// (materialises the (heads, N, N) fp32 score tensor).  Here: flash-style, one pass over K/V tiles staged
// through LDS, online softmax, nothing of size N x N ever reaches HBM.
//
// Work decomposition.  At 512x384 there are only heads x 768 query rows per image (12 288 for the encoder):
// with 32 query rows per wave that is 384 waves for 1024 SIMDs, so the KEY range is split as well
// (flash-decoding): a block = 64 query rows x 4 key splits = 8 waves; wave (qg, sp) runs the online
// softmax of query group qg over the K/V tiles of split sp, the four partial (m, l, O) triples are merged
// through LDS at the end.  1536 waves for the 16-head encoder layer.
//
// Staging: K and V^T tiles (64 keys) arrive by LDS-DMA (buffer_load ... lds; rows beyond nk are zero-
// filled by the buffer range check) into a per-split two-stage ring, lane-linear 128-byte rows with the
// bank swizzle applied on the source side (slot s of row r holds 16-byte chunk s ^ ((r >> 1) & 7)), counted
// vmcnt + raw s_barrier so the next tile stays in flight while the current one is consumed.
//
// Wave64 / MFMA mapping (v_mfma_f32_32x32x16_bf16), one wave = 32 query rows:
//   S^T = K . Q^T   -> accumulator has the QUERY on the lane axis and 16 keys in registers, so the
//                      row-wise max / sum is 16 in-lane ops + one exchange with lane^32
//   O^T = V^T . P^T -> P^T is consumed straight from the S^T accumulator registers as the MFMA
//                      B operand (accumulator-as-operand, k order permuted: key 16s+8(j>>2)+4h+(j&3)),
//                      the matching A operand comes from the V^T tile with two 8-byte LDS reads
//   q arrives RoPE-rotated and pre-scaled by d^-1/2, k RoPE-rotated, v transposed per head: all three
//   are written in that form by the projection GEMM's epilogue (gemm_kernel.h, EPI_ATTN).
#include <stdlib.h>
#include "common.h"
#include "gemm.h"

namespace mslam {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int KV_TILE = 64;
constexpr int AT_STAGE = 16384;           // K tile (64 x 128 B) + V^T tile (64 x 128 B)
constexpr int OM_ROW = 68;                // floats per (wave, query) row of the merge buffer
constexpr unsigned kAtOob = 0x80000000u;

#define MSLAM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N>
__device__ __forceinline__ void at_wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

// AT_SPLIT key splits per block (2 * AT_SPLIT waves): 4 when the grid would otherwise leave SIMDs idle,
// fewer (less LDS, several blocks per CU) for the batched backend calls.
template <int AT_SPLIT>
__global__ __launch_bounds__(128 * AT_SPLIT) void attention_kernel(const bf16* __restrict__ Q, const bf16* __restrict__ K,
                                                        const bf16* __restrict__ VT, bf16* __restrict__ O,
                                                        int heads, int nq, int nk) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // ring, later the merge buffer
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qg = wid & 1, sp = wid >> 1;
  const int h = lane >> 5, lq = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const size_t bh = (size_t)b * heads + head;
  const int q_row = blockIdx.x * 64 + qg * 32 + lq;
  const bool q_ok = q_row < nq;

  // Q^T fragments (B operand): lane = query, 8 consecutive features per k-step
  bf16x8 qf[4];
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const size_t qi = (bh * nq + (q_ok ? q_row : 0)) * 64 + 16 * s + 8 * h;
    qf[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Q + qi));   // rows >= nq are never stored
  }

  // this wave's tiles: split sp owns tiles [t_begin, t_begin + my_n); every wave runs n_iter barriers
  const int ntiles = (nk + KV_TILE - 1) / KV_TILE;
  const int n_iter = (ntiles + AT_SPLIT - 1) / AT_SPLIT;
  const int t_begin = sp * n_iter;
  const int my_n = max(0, min(ntiles, t_begin + n_iter) - t_begin);

  // ---- LDS-DMA source addressing: the two waves of a split share its tiles, 4 + 4 pieces each ------
  const __amdgpu_buffer_rsrc_t rsK =
      __builtin_amdgcn_make_buffer_rsrc((void*)(K + bh * (size_t)nk * 64), 0, nk * 128, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsV =
      __builtin_amdgcn_make_buffer_rsrc((void*)(VT + bh * 64 * (size_t)nk), 0, nk * 128, 0x00020000);
  const int chunk = (lane & 7) ^ (4 * qg + (lane >> 4));   // piece p = qg + 2i: (row >> 1) & 7 = 4*qg + (lane >> 4)
  unsigned k_off[4], v_off[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (qg + 2 * i) * 8 + (lane >> 3);
    k_off[i] = (unsigned)(row * 64 + chunk * 8) * 2u;        // + kv0 * 128; keys >= nk fall outside the buffer
    v_off[i] = (unsigned)(row * nk + chunk * 8) * 2u;        // + kv0 * 2
  }
  auto issue = [&](int stage, int tile) {
    unsigned char* base = smem + (sp * 2 + stage) * AT_STAGE + qg * 1024;
    const int kv0 = tile * KV_TILE;
    const unsigned v_bad = (unsigned)(nk - 1 - (kv0 + chunk * 8)) & kAtOob;   // key chunk beyond nk: zero fill
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, MSLAM_LDS_PTR(base + i * 2048), 16, k_off[i] + (unsigned)kv0 * 128u, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, MSLAM_LDS_PTR(base + 8192 + i * 2048), 16,
                                               (v_off[i] + (unsigned)kv0 * 2u) | v_bad, 0, 0, 0);
  };

  f32x16 o_acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; dt++)
#pragma unroll
    for (int r = 0; r < 16; r++) o_acc[dt][r] = 0.0f;
  float m_run = -INFINITY, l_run = 0.0f;   // m in score units; probabilities are exp2((s - m) * log2 e)
  const float kLog2e = 1.4426950408889634f;
  const int kh = h ^ ((lq >> 1) & 7);      // fragment slot = (2s) ^ kh, see the swizzle above

  auto compute = [&](int stage, int tile) {
    const unsigned char* sK = smem + (sp * 2 + stage) * AT_STAGE + lq * 128;
    const unsigned char* sV = sK + 8192;
    const int kv0 = tile * KV_TILE;
    // ---- S^T = K . Q^T for the two 32-key sub-tiles -------------------------------------------
    f32x16 s_acc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s_acc[u][r] = 0.0f;
#pragma unroll
      for (int s = 0; s < 4; s++) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + u * 4096 + (((2 * s) ^ kh) * 16));
        s_acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], s_acc[u], 0, 0, 0);
      }
    }
    // ---- online softmax: the lane owns one query; keys are in registers --------------------------
    if (kv0 + KV_TILE > nk) {   // partial last tile: zero-filled K rows must not take part
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < 16; r++)
          if (kv0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * h >= nk) s_acc[u][r] = -INFINITY;
    }
    float m_tile = -INFINITY;
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) m_tile = fmaxf(m_tile, s_acc[u][r]);
    m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32, 64));
    const float m_new = fmaxf(m_run, m_tile);   // finite: every tile holds at least one valid key
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * kLog2e);   // exp2(-inf) = 0 on the first tile
    const float neg_m = -m_new * kLog2e;
    float l_tile = 0.0f;
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s_acc[u][r], kLog2e, neg_m));   // masked keys: exp2(-inf) = 0
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
    // ---- O^T += V^T . P^T  (P^T straight from the S^T accumulator registers) ----------------------
#pragma unroll
    for (int u = 0; u < 2; u++) {
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; j++) pf[j] = (bf16)s_acc[u][8 * s2 + j];
        const int c0 = 4 * u + 2 * s2;   // 16-byte chunk of keys 32u + 16 s2 .. +7; lane half h takes bytes 8h..8h+7
#pragma unroll
        for (int dt = 0; dt < 2; dt++) {
          const unsigned char* vrow = sV + dt * 4096 + 8 * h;
          const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow + ((c0 ^ ((lq >> 1) & 7)) * 16));
          const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + (((c0 + 1) ^ ((lq >> 1) & 7)) * 16));
          bf16x8 vf;
          vf[0] = v0[0]; vf[1] = v0[1]; vf[2] = v0[2]; vf[3] = v0[3];
          vf[4] = v1[0]; vf[5] = v1[1]; vf[6] = v1[2]; vf[7] = v1[3];
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[dt], 0, 0, 0);
        }
      }
    }
  };

  // ---- main loop: tile i+1 stays in flight while tile i is consumed ------------------------------
  if (my_n > 0) issue(0, t_begin);
  if (my_n > 1) issue(1, t_begin + 1);
  for (int i = 0; i < n_iter; i++) {
    if (i == 0 && my_n > 1) at_wait_vmcnt<8>();
    else at_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();   // both waves' pieces of tile i landed; everyone is done with tile i-1
    if (i >= 1 && i + 1 < my_n) issue((i + 1) & 1, t_begin + i + 1);
    if (i < my_n) compute(i & 1, t_begin + i);
  }

  // ---- merge the four key splits ---------------------------------------------------------------
  __builtin_amdgcn_s_barrier();     // the ring is free
  constexpr int NWV = 2 * AT_SPLIT;
  float* Om = reinterpret_cast<float*>(smem);                       // [NWV waves][32 q][OM_ROW]
  float* ml = Om + NWV * 32 * OM_ROW;                               // [NWV waves][32 q][2]
  if (h == 0) {
    ml[(wid * 32 + lq) * 2] = m_run;
    ml[(wid * 32 + lq) * 2 + 1] = l_run;
  }
#pragma unroll
  for (int dt = 0; dt < 2; dt++)
#pragma unroll
    for (int gq = 0; gq < 4; gq++)
      *reinterpret_cast<float4*>(Om + (wid * 32 + lq) * OM_ROW + 32 * dt + 8 * gq + 4 * h) =
          make_float4(o_acc[dt][4 * gq], o_acc[dt][4 * gq + 1], o_acc[dt][4 * gq + 2], o_acc[dt][4 * gq + 3]);
  __syncthreads();
  for (int e = t; e < 512; e += 128 * AT_SPLIT) {
    const int q = e >> 3, d0 = (e & 7) * 8;     // 64 queries x 8 feature octets
    const int fq = q >> 5, ql = q & 31;
    float m_s[AT_SPLIT], m_max = -INFINITY;
#pragma unroll
    for (int s = 0; s < AT_SPLIT; s++) {
      m_s[s] = ml[((s * 2 + fq) * 32 + ql) * 2];
      m_max = fmaxf(m_max, m_s[s]);
    }
    float acc8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, l_sum = 0.0f;
#pragma unroll
    for (int s = 0; s < AT_SPLIT; s++) {
      const float w = __builtin_amdgcn_exp2f((m_s[s] - m_max) * kLog2e);   // empty split: exp2(-inf) = 0
      l_sum = fmaf(w, ml[((s * 2 + fq) * 32 + ql) * 2 + 1], l_sum);
      const float* src = Om + ((s * 2 + fq) * 32 + ql) * OM_ROW + d0;
      const float4 p0 = *reinterpret_cast<const float4*>(src), p1 = *reinterpret_cast<const float4*>(src + 4);
      acc8[0] = fmaf(w, p0.x, acc8[0]); acc8[1] = fmaf(w, p0.y, acc8[1]);
      acc8[2] = fmaf(w, p0.z, acc8[2]); acc8[3] = fmaf(w, p0.w, acc8[3]);
      acc8[4] = fmaf(w, p1.x, acc8[4]); acc8[5] = fmaf(w, p1.y, acc8[5]);
      acc8[6] = fmaf(w, p1.z, acc8[6]); acc8[7] = fmaf(w, p1.w, acc8[7]);
    }
    const int row = blockIdx.x * 64 + q;
    if (row < nq) {
      const float inv = 1.0f / l_sum;
      bf16x8 pk;
#pragma unroll
      for (int k = 0; k < 8; k++) pk[k] = (bf16)(acc8[k] * inv);
      *reinterpret_cast<bf16x8*>(O + ((size_t)b * nq + row) * ((size_t)heads * 64) + (size_t)head * 64 + d0) = pk;
    }
  }
}

template <int AT_SPLIT>
static int launch_attention_split(const bf16* Q, const bf16* K, const bf16* VT, bf16* O, int batch, int heads, int nq,
                                  int nk, hipStream_t stream) {
  constexpr int ring = AT_SPLIT * 2 * AT_STAGE;
  constexpr int merge = (2 * AT_SPLIT * 32 * OM_ROW + 2 * AT_SPLIT * 32 * 2) * (int)sizeof(float);
  constexpr int shmem = ring > merge ? ring : merge;
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void*)attention_kernel<AT_SPLIT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, shmem), "attention: attr");
    if (rc) return rc;
    attr_set = true;
  }
  dim3 grid((nq + 63) / 64, heads, batch);
  hipLaunchKernelGGL(attention_kernel<AT_SPLIT>, grid, dim3(128 * AT_SPLIT), shmem, stream, Q, K, VT, O, heads, nq, nk);
  return check_hip(hipGetLastError(), "attention launch");
}

int launch_attention(const bf16* Q, const bf16* K, const bf16* VT, bf16* O, int batch, int heads, int nq, int nk,
                     hipStream_t stream) {
  MSLAM_REQUIRE(nq > 0 && nk > 0 && batch > 0 && heads > 0, "attention: empty problem");
  MSLAM_REQUIRE(nk % 8 == 0, "attention: key count %d must be a multiple of 8", nk);
  MSLAM_REQUIRE((size_t)nk * 128 < (1ull << 31), "attention: key count %d too large", nk);
  static int forced = -2;   // MSLAM_ATTN_SPLIT=1|2|4 forces the key split (experiments)
  if (forced == -2) {
    const char* e = getenv("MSLAM_ATTN_SPLIT");
    forced = e ? atoi(e) : -1;
  }
  const long blocks = (long)((nq + 63) / 64) * heads * batch;
  const int split = forced > 0 ? forced : (blocks <= 256 ? 4 : 2);   // measured: tools/attn_tune.py
  switch (split) {
    case 4: return launch_attention_split<4>(Q, K, VT, O, batch, heads, nq, nk, stream);
    case 2: return launch_attention_split<2>(Q, K, VT, O, batch, heads, nq, nk, stream);
    default: return launch_attention_split<1>(Q, K, VT, O, batch, heads, nq, nk, stream);
  }
}

}  // namespace mslam
