// bf16 MFMA GEMM kernel (see gemm.h).  256 threads = 4 waves in a 2x2 arrangement; each wave owns
// MI x NI tiles of 32x32 (v_mfma_f32_32x32x16_bf16, fp32 accumulators in registers).
//   block tile (2*MI*32) x (2*NI*32), BK = 64, LDS double buffered, rows padded to 72 bf16 (144 B):
//   the 16 rows a ds_read_b128 lane group touches then start on 16 disjoint 4-bank slots.
//   global -> registers (16 B per lane, issued before the MFMA block of the current tile) ->
//   ds_write_b128 after it: one barrier per K tile.
#include <stdlib.h>
#include <type_traits>
#include <utility>
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

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int MI, int NI, int R, bool CONV>
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
    if constexpr (CONV) {
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

  // Register ring of R stages: while tile kt is consumed from LDS, tiles kt+1 .. kt+R are in flight
  // from HBM/L2 into registers (a stage is 16 B x (A_CH + B_CH) per lane).  Stage indices are
  // compile-time constants (the K loop is unrolled by R), so the ring stays in VGPRs.
  const short relu_floor = g.a_relu ? (short)0 : (short)-32768;
  u32x4 a_reg[R][A_CH], b_reg[R][B_CH];
  unsigned ld_mask[R];  // bit i: A chunk i valid, bit 8+i: B chunk i valid (invalid chunks are zeroed at store)
  // Loads are UNCONDITIONAL (out-of-range chunks read a clamped, valid address and are masked when
  // they are written to LDS): a branch around a load would make the compiler drain the whole ring
  // with s_waitcnt vmcnt(0) instead of waiting only for the oldest stage.
  auto load_tiles = [&](auto st_c, int k0) {
    constexpr int st = decltype(st_c)::value;
    const int kk = k0 + part * 8;
    const bool k_ok = kk < g.K;
    int tap_dy = 0, tap_dx = 0, cc = kk;
    if constexpr (CONV) {
      const int tap = kk / g.cC;
      cc = kk - tap * g.cC;
      tap_dy = tap / g.cKs;
      tap_dx = tap - tap_dy * g.cKs;
    }
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
      bool ok = a_ok[i] && k_ok;
      const bf16* p;
      if constexpr (CONV) {
        const int iy = a_iy0[i] + tap_dy, ix = a_ix0[i] + tap_dx;
        ok = ok && iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW;
        p = a_base[i] + ((size_t)iy * g.cW + ix) * g.cC + cc;
      } else {
        p = a_base[i] + kk;
      }
      p = ok ? p : g.A;
      a_reg[st][i] = *reinterpret_cast<const u32x4*>(p);
      mask |= ok ? (1u << i) : 0u;
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
      const bool ok = b_ok[i] && k_ok;
      const bf16* p = ok ? g.W + (size_t)(n0 + b_row[i]) * g.K + kk : g.W;
      b_reg[st][i] = *reinterpret_cast<const u32x4*>(p);
      mask |= ok ? (1u << (8 + i)) : 0u;
    }
    ld_mask[st] = mask;
  };
  auto store_tiles = [&](auto st_c, int buf) {
    constexpr int st = decltype(st_c)::value;
    const unsigned mask = ld_mask[st];
    const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < A_CH; i++) {
      u32x4 v = ((mask >> i) & 1u) ? a_reg[st][i] : zero;
      {  // optional ReLU on A: max(x, 0) on the int16 view (bf16 sign bit == int16 sign bit); floor is
         // INT16_MIN when disabled, so the same instruction is a no-op (branch-free)
        short8 sv = __builtin_bit_cast(short8, v);
        const short fl = relu_floor;
        const short8 z = {fl, fl, fl, fl, fl, fl, fl, fl};
        sv = __builtin_elementwise_max(sv, z);
        v = __builtin_bit_cast(u32x4, sv);
      }
      *reinterpret_cast<u32x4*>(As + ((size_t)buf * BM + a_row[i]) * LDS_ROW + part * 8) = v;
    }
#pragma unroll
    for (int i = 0; i < B_CH; i++) {
      const u32x4 v = ((mask >> (8 + i)) & 1u) ? b_reg[st][i] : zero;
      *reinterpret_cast<u32x4*>(Bs + ((size_t)buf * BN + b_row[i]) * LDS_ROW + part * 8) = v;
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.0f;

  const int nk = (g.K + BK - 1) / BK;
  // prologue: tile 0 -> LDS[0]; tiles 1..R -> register stages (tile t lives in stage t % R)
  load_tiles(std::integral_constant<int, 0>{}, 0);
  store_tiles(std::integral_constant<int, 0>{}, 0);
  static_for<R>([&](auto j_c) {
    constexpr int t = decltype(j_c)::value + 1;
    load_tiles(std::integral_constant<int, t % R>{}, t * BK);  // beyond K: zero fill, no memory access
  });
  __syncthreads();

  const int frag_row = lane & 31, frag_k = (lane >> 5) * 8;
  // nk is rounded up to a multiple of R: the padding tiles are all-zero (masked loads), so the loop
  // body carries no data-dependent branch and the compiler keeps counted vmcnt waits.
  const int nk_pad = (nk + R - 1) / R * R;
  for (int kt0 = 0; kt0 < nk_pad; kt0 += R) {
    static_for<R>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      const int kt = kt0 + j;
      const int cur = kt & 1;
      using next_stage = std::integral_constant<int, (j + 1) % R>;  // stage of tile kt+1 (kt0 % R == 0)
      store_tiles(next_stage{}, cur ^ 1);            // waits only for the OLDEST stage in flight
      load_tiles(next_stage{}, (kt + 1 + R) * BK);   // refill the stage just drained
      const bf16* Ab = As + ((size_t)cur * BM + wm * MI * 32 + frag_row) * LDS_ROW + frag_k;
      const bf16* Bb = Bs + ((size_t)cur * BN + wn * NI * 32 + frag_row) * LDS_ROW + frag_k;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ks++) {
        bf16x8 af[MI], bfr[NI];
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
          af[mi] = *reinterpret_cast<const bf16x8*>(Ab + (size_t)mi * 32 * LDS_ROW + ks * 16);
#pragma unroll
        for (int ni = 0; ni < NI; ni++)
          bfr[ni] = *reinterpret_cast<const bf16x8*>(Bb + (size_t)ni * 32 * LDS_ROW + ks * 16);
#pragma unroll
        for (int mi = 0; mi < MI; mi++)
#pragma unroll
          for (int ni = 0; ni < NI; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
    });
  }

  // ---- epilogue ------------------------------------------------------------------------------
  // Every accumulator index below is a compile-time constant (static_for): a runtime-indexed
  // ext_vector array would be demoted to scratch memory and spilled inside the K loop.
  const int half = lane >> 5, lcol = lane & 31;
  static_for<MI>([&](auto mi_c) {
    static_for<NI>([&](auto ni_c) {
      constexpr int mi = decltype(mi_c)::value, ni = decltype(ni_c)::value;
      const f32x16 accv = acc[mi][ni];
      const int col = n0 + wn * NI * 32 + ni * 32 + lcol;
      const int row_base = m0 + wm * MI * 32 + mi * 32 + 4 * half;
      const bool col_ok = col < g.N;
      float bias = 0.0f;
      if (g.bias && col_ok) bias = g.bias[g.epi == EPI_CONVT ? col / (g.ct_s * g.ct_s) : col];
      float v[16];
      static_for<16>([&](auto r_c) {
        constexpr int r = decltype(r_c)::value;
        float x = accv[r] + bias;
        if (g.act == ACT_GELU) x = gelu_erf(x);
        else if (g.act == ACT_RELU) x = fmaxf(x, 0.0f);
        v[r] = x;
      });
      if (g.epi == EPI_PLAIN) {
        static_for<16>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
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
        });
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
          static_for<16>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            const float partner = __shfl_xor(v[r], 16, 64);
            if (row < g.M && col_ok) {
              const int b = row / ntok, n = row - b * ntok;
              const int p = use_y ? n / g.tok_w : n % g.tok_w;
              const float c = g.rope_cos[p * 16 + (f & 15)], sn = g.rope_sin[p * 16 + (f & 15)];
              const float x = lo ? (v[r] * c - partner * sn) : (v[r] * c + partner * sn);
              dst[(((size_t)b * g.heads + head) * ntok + n) * 64 + f] = (bf16)(x * scale);
            }
          });
        } else {
          // v: transposed per head, 4 consecutive tokens per 8-byte store
          static_for<4>([&](auto gq_c) {
            constexpr int gq = decltype(gq_c)::value;
            const int row = row_base + 8 * gq;  // rows row..row+3 (registers 4gq..4gq+3)
            if (row < g.M && col_ok) {
              const int b = row / ntok, n = row - b * ntok;
              typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
              bf16x4 pk;
              pk[0] = (bf16)v[4 * gq + 0]; pk[1] = (bf16)v[4 * gq + 1];
              pk[2] = (bf16)v[4 * gq + 2]; pk[3] = (bf16)v[4 * gq + 3];
              *reinterpret_cast<bf16x4*>(g.vt_out + (((size_t)b * g.heads + head) * 64 + f) * ntok + n) = pk;
            }
          });
        }
      } else {  // EPI_CONVT: n = co*s*s + i*s + j ; m = (b, y, x) -> out[b, y*s+i, x*s+j, co]
        const int ss = g.ct_s * g.ct_s;
        const int co = col / ss, ij = col - co * ss, i = ij / g.ct_s, j = ij - i * g.ct_s;
        static_for<16>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
          const int row = row_base + (r & 3) + 8 * (r >> 2);
          if (row < g.M && col_ok) {
            const int hw = g.ct_h * g.ct_w;
            const int b = row / hw, rem = row - b * hw, y = rem / g.ct_w, x = rem - y * g.ct_w;
            const size_t o = (((size_t)b * g.ct_h * g.ct_s + (size_t)y * g.ct_s + i) * (g.ct_w * g.ct_s) +
                              (size_t)x * g.ct_s + j) * g.ct_cout + co;
            reinterpret_cast<bf16*>(g.out)[o] = (bf16)v[r];
          }
        });
      }
    });
  });
}

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  MSLAM_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem %dx%dx%d", a.M, a.N, a.K);
  MSLAM_REQUIRE(a.K % 8 == 0, "gemm: K=%d must be a multiple of 8", a.K);
  MSLAM_REQUIRE(!a.a_conv || a.cC % 8 == 0, "gemm: conv channels %d must be a multiple of 8", a.cC);
  MSLAM_REQUIRE(a.a_conv || a.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", a.lda);
  MSLAM_REQUIRE(a.epi != EPI_ATTN || (a.sec_dim % 64 == 0 && a.ntok % 4 == 0 && a.kv_ntok % 4 == 0),
                "gemm: attention epilogue needs 64-wide heads and token counts divisible by 4");
  // Tile / prefetch-depth selection.  MSLAM_GEMM="<big_threshold>,<r_big>,<r_small>" overrides the
  // defaults for experiments (tools/bench_kernels.py).
  static int cfg_thr = 200, cfg_rbig = 2, cfg_rsmall = 4;
  static bool cfg_init = false;
  if (!cfg_init) {
    if (const char* e = getenv("MSLAM_GEMM")) sscanf(e, "%d,%d,%d", &cfg_thr, &cfg_rbig, &cfg_rsmall);
    cfg_init = true;
  }
  const long big = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const long small = (long)((a.M + 63) / 64) * ((a.N + 63) / 64);
  const size_t shmem_big = (size_t)2 * (128 + 128) * LDS_ROW * sizeof(bf16);
  const size_t shmem_small = (size_t)2 * (64 + 64) * LDS_ROW * sizeof(bf16);
#define MSLAM_GEMM_LAUNCH(MI_, NI_, R_, blocks, shmem)                                                            \
  do {                                                                                                            \
    if (a.a_conv) hipLaunchKernelGGL((gemm_bf16_kernel<MI_, NI_, R_, true>), dim3((unsigned)(blocks)), dim3(256), \
                                     shmem, stream, a);                                                           \
    else hipLaunchKernelGGL((gemm_bf16_kernel<MI_, NI_, R_, false>), dim3((unsigned)(blocks)), dim3(256), shmem,  \
                            stream, a);                                                                           \
  } while (0)
  if (big >= cfg_thr) {
    static bool attr_set = false;
    if (!attr_set) {
      const void* fns[4] = {(const void*)gemm_bf16_kernel<2, 2, 1, false>, (const void*)gemm_bf16_kernel<2, 2, 1, true>,
                            (const void*)gemm_bf16_kernel<2, 2, 2, false>, (const void*)gemm_bf16_kernel<2, 2, 2, true>};
      for (const void* fn : fns) {
        int rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_big), "gemm: attr");
        if (rc) return rc;
      }
      attr_set = true;
    }
    if (cfg_rbig >= 2) MSLAM_GEMM_LAUNCH(2, 2, 2, big, shmem_big);
    else MSLAM_GEMM_LAUNCH(2, 2, 1, big, shmem_big);
  } else {
    if (cfg_rsmall >= 4) MSLAM_GEMM_LAUNCH(1, 1, 4, small, shmem_small);
    else if (cfg_rsmall >= 2) MSLAM_GEMM_LAUNCH(1, 1, 2, small, shmem_small);
    else MSLAM_GEMM_LAUNCH(1, 1, 1, small, shmem_small);
  }
#undef MSLAM_GEMM_LAUNCH
  return check_hip(hipGetLastError(), "gemm launch");
}

}  // namespace mslam
