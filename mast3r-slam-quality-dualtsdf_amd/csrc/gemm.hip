// bf16 MFMA GEMM kernel (see gemm.h).  256 threads = 4 waves in a 2x2 arrangement; each wave owns
// MI x NI tiles of 32x32 (v_mfma_f32_32x32x16_bf16, fp32 accumulators in registers).
//   block tile (2*MI*32) x (2*NI*32), BK = 64, LDS double buffered, rows padded to 72 bf16 (144 B):
//   the 16 rows a ds_read_b128 lane group touches then start on 16 disjoint 4-bank slots.
//   global -> registers (16 B per lane, issued before the MFMA block of the current tile) ->
//   ds_write_b128 after it: one barrier per K tile.
#include "common.h"
#include "gemm.h"

namespace mslam {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int BK = 64;
constexpr int LDS_ROW = BK + 8;  // bf16 elements per LDS row

__device__ __forceinline__ float bf16_to_f32(bf16 v) { return (float)v; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int MI, int NI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int BM = 2 * MI * 32, BN = 2 * NI * 32;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;  // 16-byte chunks per thread per tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16* As = reinterpret_cast<bf16*>(smem);                       // [2][BM][LDS_ROW]
  bf16* Bs = As + 2 * BM * LDS_ROW;                               // [2][BN][LDS_ROW]

  // XCD-aware tile order: blocks sharing an XCD get neighbouring tiles (same A rows / W panel in L2)
  const unsigned nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  const unsigned tile = xcd_remap(blockIdx.x, nbm * nbn);
  const int m0 = (tile / nbn) * BM, n0 = (tile % nbn) * BN;

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // ---- per-thread load slots ---------------------------------------------------------------
  const int part = t & 7;  // 16-byte chunk inside the 128-byte K slice
  int a_row[A_CH];
  bool a_ok[A_CH];
  const bf16* a_base[A_CH];
  int a_iy0[A_CH], a_ix0[A_CH];
#pragma unroll
  for (int i = 0; i < A_CH; i++) {
    a_row[i] = (t >> 3) + i * 32;
    const int m = m0 + a_row[i];
    a_ok[i] = m < g.M;
    if (g.a_conv) {
      const int hw = g.cHo * g.cWo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
      a_iy0[i] = oy * g.cStride - g.cPad;
      a_ix0[i] = ox * g.cStride - g.cPad;
      a_base[i] = g.A + (size_t)b * g.cH * g.cW * g.cC;
    } else {
      a_base[i] = g.A + (size_t)m * g.lda;
      a_iy0[i] = a_ix0[i] = 0;
    }
  }
  int b_row[B_CH];
  bool b_ok[B_CH];
#pragma unroll
  for (int i = 0; i < B_CH; i++) {
    b_row[i] = (t >> 3) + i * 32;
    b_ok[i] = (n0 + b_row[i]) < g.N;
  }

  u32x4 a_reg[A_CH], b_reg[B_CH];
  auto load_tiles = [&](int k0) {
    const int kk = k0 + part * 8;
    const bool k_ok = kk < g.K;
    int tap_dy = 0, tap_dx = 0, cc = kk;
    if (g.a_conv) {
      const int tap = kk / g.cC;
      cc = kk - tap * g.cC;
      tap_dy = tap / g.cKs;
      tap_dx = tap - tap_dy * g.cKs;
    }
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (a_ok[i] && k_ok) {
        if (g.a_conv) {
          const int iy = a_iy0[i] + tap_dy, ix = a_ix0[i] + tap_dx;
          if (iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW)
            v = *reinterpret_cast<const u32x4*>(a_base[i] + ((size_t)iy * g.cW + ix) * g.cC + cc);
        } else {
          v = *reinterpret_cast<const u32x4*>(a_base[i] + kk);
        }
      }
      a_reg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (b_ok[i] && k_ok) v = *reinterpret_cast<const u32x4*>(g.W + (size_t)(n0 + b_row[i]) * g.K + kk);
      b_reg[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
      u32x4 v = a_reg[i];
      if (g.a_relu) {
        short8 s = __builtin_bit_cast(short8, v);
        const short8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        s = __builtin_elementwise_max(s, z);  // bf16 sign bit == int16 sign bit
        v = __builtin_bit_cast(u32x4, s);
      }
      *reinterpret_cast<u32x4*>(As + ((size_t)buf * BM + a_row[i]) * LDS_ROW + part * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++)
      *reinterpret_cast<u32x4*>(Bs + ((size_t)buf * BN + b_row[i]) * LDS_ROW + part * 8) = b_reg[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.0f;

  const int nk = (g.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  const int frag_row = lane & 31, frag_k = (lane >> 5) * 8;
  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles((kt + 1) * BK);
    const bf16* Ab = As + ((size_t)cur * BM + wm * MI * 32 + frag_row) * LDS_ROW + frag_k;
    const bf16* Bb = Bs + ((size_t)cur * BN + wn * NI * 32 + frag_row) * LDS_ROW + frag_k;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ks++) {
      bf16x8 af[MI], bfr[NI];
#pragma unroll
      for (int mi = 0; mi < MI; mi++) af[mi] = *reinterpret_cast<const bf16x8*>(Ab + (size_t)mi * 32 * LDS_ROW + ks * 16);
#pragma unroll
      for (int ni = 0; ni < NI; ni++) bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + (size_t)ni * 32 * LDS_ROW + ks * 16);
#pragma unroll
      for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------
  const int half = lane >> 5, lcol = lane & 31;
#pragma unroll
  for (int mi = 0; mi < MI; mi++) {
#pragma unroll
    for (int ni = 0; ni < NI; ni++) {
      const int col = n0 + wn * NI * 32 + ni * 32 + lcol;
      const int row_base = m0 + wm * MI * 32 + mi * 32 + 4 * half;
      const bool col_ok = col < g.N;
      float bias = 0.0f;
      if (g.bias && col_ok) bias = g.bias[g.epi == EPI_CONVT ? col / (g.ct_s * g.ct_s) : col];
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float x = acc[mi][ni][r] + bias;
        if (g.act == ACT_GELU) x = gelu_erf(x);
        else if (g.act == ACT_RELU) x = fmaxf(x, 0.0f);
        v[r] = x;
      }
      if (g.epi == EPI_PLAIN) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = row_base + (r & 3) + 8 * (r >> 2);
          if (row < g.M && col_ok) {
            float x = v[r];
            if (g.res1_kind == KIND_F32) x += reinterpret_cast<const float*>(g.res1)[(size_t)row * g.ldr1 + col];
            else if (g.res1_kind == KIND_BF16) x += bf16_to_f32(reinterpret_cast<const bf16*>(g.res1)[(size_t)row * g.ldr1 + col]);
            if (g.res2_kind == KIND_F32) x += reinterpret_cast<const float*>(g.res2)[(size_t)row * g.ldr2 + col];
            else if (g.res2_kind == KIND_BF16) x += bf16_to_f32(reinterpret_cast<const bf16*>(g.res2)[(size_t)row * g.ldr2 + col]);
            if (g.out_kind == KIND_F32) reinterpret_cast<float*>(g.out)[(size_t)row * g.ldc + col] = x;
            else reinterpret_cast<bf16*>(g.out)[(size_t)row * g.ldc + col] = (bf16)x;
          }
        }
      } else if (g.epi == EPI_ATTN) {
        // column -> (section, head, feature); a 32-wide MFMA tile is exactly one RoPE half of one head
        const int sec = g.sec_base + col / g.sec_dim;
        const int cs = col % g.sec_dim;
        const int head = cs >> 6, f = cs & 63;
        const int ntok = (sec == 0) ? g.ntok : g.kv_ntok;
        if (sec < 2) {
          const bool use_y = f < 32;
          const bool lo = (f & 31) < 16;
          bf16* dst = (sec == 0) ? g.q_out : g.k_out;
          const float scale = (sec == 0) ? g.q_scale : 1.0f;
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const float partner = __shfl_xor(v[r], 16, 64);
            if (row < g.M && col_ok) {
              const int b = row / ntok, n = row - b * ntok;
              const int p = use_y ? n / g.tok_w : n % g.tok_w;
              const float c = g.rope_cos[p * 16 + (f & 15)], s = g.rope_sin[p * 16 + (f & 15)];
              const float x = lo ? (v[r] * c - partner * s) : (v[r] * c + partner * s);
              dst[(((size_t)b * g.heads + head) * ntok + n) * 64 + f] = (bf16)(x * scale);
            }
          }
        } else {
          // v: transposed per head, 4 consecutive tokens per 8-byte store
#pragma unroll
          for (int gq = 0; gq < 4; gq++) {
            const int row = row_base + 8 * gq;  // rows row..row+3 (registers 4gq..4gq+3)
            if (row < g.M && col_ok) {
              const int b = row / ntok, n = row - b * ntok;
              typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
              bf16x4 pk;
              pk[0] = (bf16)v[4 * gq + 0]; pk[1] = (bf16)v[4 * gq + 1];
              pk[2] = (bf16)v[4 * gq + 2]; pk[3] = (bf16)v[4 * gq + 3];
              *reinterpret_cast<bf16x4*>(g.vt_out + (((size_t)b * g.heads + head) * 64 + f) * ntok + n) = pk;
            }
          }
        }
      } else {  // EPI_CONVT: n = co*s*s + i*s + j ; m = (b, y, x) -> out[b, y*s+i, x*s+j, co]
        const int ss = g.ct_s * g.ct_s;
        const int co = col / ss, ij = col - co * ss, i = ij / g.ct_s, j = ij - i * g.ct_s;
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int row = row_base + (r & 3) + 8 * (r >> 2);
          if (row < g.M && col_ok) {
            const int hw = g.ct_h * g.ct_w;
            const int b = row / hw, rem = row - b * hw, y = rem / g.ct_w, x = rem - y * g.ct_w;
            const size_t o = (((size_t)b * g.ct_h * g.ct_s + (size_t)y * g.ct_s + i) * (g.ct_w * g.ct_s) +
                              (size_t)x * g.ct_s + j) * g.ct_cout + co;
            reinterpret_cast<bf16*>(g.out)[o] = (bf16)v[r];
          }
        }
      }
    }
  }
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  MSLAM_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem %dx%dx%d", a.M, a.N, a.K);
  MSLAM_REQUIRE(a.K % 8 == 0, "gemm: K=%d must be a multiple of 8", a.K);
  MSLAM_REQUIRE(!a.a_conv || a.cC % 8 == 0, "gemm: conv channels %d must be a multiple of 8", a.cC);
  MSLAM_REQUIRE(a.a_conv || a.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", a.lda);
  MSLAM_REQUIRE(a.epi != EPI_ATTN || (a.sec_dim % 64 == 0 && a.ntok % 4 == 0 && a.kv_ntok % 4 == 0),
                "gemm: attention epilogue needs 64-wide heads and token counts divisible by 4");
  // large tiles when they still fill the chip, small tiles otherwise (M = 768 per image)
  const long big = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (big >= 192) {
    const size_t shmem = (size_t)2 * (128 + 128) * LDS_ROW * sizeof(bf16);
    static bool attr_set = false;
    if (!attr_set) {
      int rc = check_hip(hipFuncSetAttribute((const void*)gemm_bf16_kernel<2, 2>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                         "gemm: hipFuncSetAttribute");
      if (rc) return rc;
      attr_set = true;
    }
    hipLaunchKernelGGL((gemm_bf16_kernel<2, 2>), dim3((unsigned)big), dim3(256), shmem, stream, a);
  } else {
    const long small = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    const size_t shmem = (size_t)2 * (64 + 64) * LDS_ROW * sizeof(bf16);
    hipLaunchKernelGGL((gemm_bf16_kernel<1, 1>), dim3((unsigned)small), dim3(256), shmem, stream, a);
  }
  return check_hip(hipGetLastError(), "gemm launch");
}

}  // namespace mslam
