// bf16 MFMA GEMM kernel template (see gemm.h); instantiated per tile shape in gemm_t*.hip.
// WM x WN waves per block; each wave owns MI x NI tiles of 32x32 (v_mfma_f32_32x32x16_bf16, fp32
// accumulators in registers).
//   block tile BM x BN = (32*MI*WM) x (32*NI*WN), BK = 64 (128-byte rows), S-stage ring in LDS.
//   Staging is LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass, out-of-range
//   rows / K tail / conv padding are zero-filled by the buffer range check (no masking ALU).
//   One wave-instruction writes 8 rows x 128 B lane-linearly, so the bank swizzle is applied on the
//   SOURCE side: LDS slot s of row r holds K-chunk s ^ ((r >> 1) & 7); the fragment ds_read_b128 of 16
//   consecutive rows then covers all 16 sixteen-byte slots of the 256-byte bank row (conflict-free).
//   Pipeline: tiles kt+1 .. kt+S-2 stay in flight across the barrier (counted s_waitcnt vmcnt(N), raw
//   s_barrier - a __syncthreads() fence would drain the DMA queue); one barrier per K tile.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.h"
#include "gemm.h"

namespace mslam {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short short8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int BK = 64;
// Measurement builds only (tools/probes/build_gemm_ablate.sh): 1 = no DMA behind the prologue (MFMA + fragment reads +
// barriers), 2 = no MFMA (DMA + waits + barriers + fragment reads), 3 = no fragment reads (MFMA on stale registers),
// 4 = the same FLOP count through v_mfma_f32_16x16x32_bf16 (two per 32x32x16 instruction, quarter accumulators): a clock
// / issue probe for the other MFMA shape, not a GEMM.
#ifndef MSLAM_GEMM_ABLATE
#define MSLAM_GEMM_ABLATE 0
#endif
constexpr unsigned kOob = 0x80000000u;  // byte offset beyond any buffer (all operands are < 2 GiB): reads as zero

#define MSLAM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16 v) { return (float)v; }

template <int... Is, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)) with erfc from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far
// inside the bf16 output rounding): 1 rcp + 1 exp2 + 8 fma instead of the ~40-instruction libm erff.
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float erfc_z = p * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);   // erfc(|x| / sqrt 2)
  const float one_plus_erf = (x >= 0.0f) ? 2.0f - erfc_z : erfc_z;
  return 0.5f * x * one_plus_erf;
}

// The same function on two values at once: every multiply / fma is the packed instruction (v_pk_mul_f32, v_pk_fma_f32:
// the IEEE operation per half, so results are bit-identical to gelu_erf), the two transcendental ops stay per value.
// The GELU epilogue of the encoder's fc1 is ~8 us of pure VALU time per launch (128 values per lane).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  const f32x2 z = ax * 0.70710678118654752f;
  const f32x2 den = __builtin_elementwise_fma(f32x2{0.3275911f, 0.3275911f}, z, f32x2{1.0f, 1.0f});
  const f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  f32x2 p = __builtin_elementwise_fma(f32x2{1.061405429f, 1.061405429f}, t, f32x2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, f32x2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, f32x2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, f32x2{0.254829592f, 0.254829592f});
  const f32x2 a = (z * -1.4426950408889634f) * z;          // (-1.4427 z) z, as gelu_erf orders it
  const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2 erfc_z = (p * t) * e;
  const f32x2 two_minus = f32x2{2.0f, 2.0f} - erfc_z;
  const f32x2 one_plus_erf = {x[0] >= 0.0f ? two_minus[0] : erfc_z[0], x[1] >= 0.0f ? two_minus[1] : erfc_z[1]};
  return (x * 0.5f) * one_plus_erf;
}

// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = simm16[15:14]:[3:0], expcnt [6:4], lgkmcnt [11:8])
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

template <int WM, int WN, int MI, int NI, int S, bool CONV>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmArgs g) {
  constexpr int NW = WM * WN;                       // waves per block
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  static_assert(NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "DMA pieces must divide over the waves");
  constexpr int A_PC = BM / (8 * NW), B_PC = BN / (8 * NW);   // 1-KiB DMA pieces (8 rows x 128 B) per wave per tile
  constexpr int N_DMA = A_PC + B_PC;                // DMA instructions per wave per tile
  constexpr int STAGE = (BM + BN) * 128;            // bytes per ring stage
  static_assert(S >= 2 && S <= 8, "ring depth");
  static_assert(N_DMA * (S - 1) < 64, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];  // [S][BM + BN][128 B]

  // Tile order.  (1) xcd_remap: the blocks the dispatcher deals to one XCD get a contiguous range of
  // logical ids.  (2) Inside that order tiles are walked in groups of GM row-panels (~1024 rows),
  // column-major inside a group: the blocks that run together on an XCD then share a few A panels AND
  // a few W panels, so every W panel crosses the fabric into that XCD's L2 once per group instead of
  // once per row-panel; for the small-M shapes (all rows in one group) each XCD reads only its own
  // slice of W.
  const unsigned nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  const unsigned ngrp = g.groups > 1 ? 2u : 1u;
  if (blockIdx.x >= nbm * nbn * ngrp) {
    // ---- prefetch role: the blocks behind the tile grid only stream the next weights through the fabric ----
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const unsigned pb = blockIdx.x - nbm * nbn * ngrp, npb = gridDim.x - nbm * nbn * ngrp;
    u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const char* base = reinterpret_cast<const char*>(g.pf_ptr[r]);
      const size_t bytes = g.pf_bytes[r] & ~(size_t)15;
      const size_t stride = (size_t)npb * (64 * NW) * 16;
      size_t off = ((size_t)pb * (64 * NW) + threadIdx.x) * 16;
      for (; off + 15 * stride < bytes; off += 16 * stride) {   // 16 independent 16-byte loads in flight per lane
        u32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = *reinterpret_cast<const u32x4*>(base + off + k * stride);   // default policy: allocates in the Infinity Cache
#pragma unroll
        for (int k = 0; k < 16; k++) acc ^= v[k];
      }
      for (; off < bytes; off += stride) acc ^= *reinterpret_cast<const u32x4*>(base + off);
    }
    if ((acc[0] ^ acc[1]) == 0x9e3779b9u && (acc[2] ^ acc[3]) == 0x7f4a7c15u && g.pf_sink) *g.pf_sink = acc[0];
    return;
  }
  const unsigned logical = xcd_remap(blockIdx.x, nbm * nbn * ngrp);   // group 0 on the first XCDs, group 1 on the rest
  const unsigned prob = logical / (nbm * nbn), tile = logical - prob * (nbm * nbn);
  if (prob) {   // second problem of a grouped launch (block-uniform)
    g.A += g.a_gstride; g.W = g.W1; g.bias = g.bias1;
    g.out = reinterpret_cast<char*>(g.out) + g.out_gbytes;
    if (g.res1) g.res1 = reinterpret_cast<const char*>(g.res1) + g.res1_gbytes;
    if (g.res2) g.res2 = reinterpret_cast<const char*>(g.res2) + g.res2_gbytes;
    if (g.epi == EPI_ATTN) { g.q_out += g.qkv_gstride; g.k_out += g.qkv_gstride; g.vt_out += g.qkv_gstride; }
  }
  constexpr unsigned GM = 1024 / BM;
  const unsigned per_group = GM * nbn, grp = tile / per_group, in_grp = tile - grp * per_group;
  const unsigned gsz = min(nbm - grp * GM, GM);
  const int m0 = (grp * GM + in_grp % gsz) * BM, n0 = (in_grp / gsz) * BN;

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WN, wn = wid % WN;

  // early, latency-hidden loads for the epilogue
  const int lcol = lane & 31;
  float bias_v[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ni++) {
    const int col = n0 + wn * NI * 32 + ni * 32 + lcol;
    bias_v[ni] = (g.bias && col < g.N) ? g.bias[g.epi == EPI_CONVT ? col / (g.ct_s * g.ct_s) : col] : 0.0f;
  }

  // ---- DMA source addressing -----------------------------------------------------------------
  // piece q = wid + NW*i covers tile rows 8q .. 8q+7; lane -> row 8q + (lane >> 3), LDS slot lane & 7,
  // which holds K-chunk c = slot ^ ((row >> 1) & 7) = slot ^ (4*(wid & 1) + (lane >> 4)).
  const int chunk = (lane & 7) ^ (4 * (wid & 1) + (lane >> 4));
  const size_t a_bytes = CONV ? (size_t)g.cB * g.cH * g.cW * g.cC * 2 : ((size_t)(g.M - 1) * g.lda + g.K) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
  unsigned a_off[A_PC];   // dense: byte offset of (row, chunk) at k0 = 0, or kOob; conv: byte offset of the image
  int a_iy0[A_PC], a_ix0[A_PC];
#pragma unroll
  for (int i = 0; i < A_PC; i++) {
    const int m = m0 + (wid + NW * i) * 8 + (lane >> 3);
    if constexpr (CONV) {
      const int hw = g.cHo * g.cWo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / g.cWo, ox = rem - oy * g.cWo;
      a_iy0[i] = (m < g.M) ? oy * g.cStride - g.cPad : -(1 << 20);   // invalid row: every tap out of range
      a_ix0[i] = ox * g.cStride - g.cPad;
      a_off[i] = (unsigned)b * (unsigned)(g.cH * g.cW * g.cC) * 2u;
    } else {
      a_off[i] = (m < g.M) ? ((unsigned)m * (unsigned)g.lda + (unsigned)chunk * 8u) * 2u : kOob;
      a_iy0[i] = a_ix0[i] = 0;
    }
  }
  unsigned b_off[B_PC];
#pragma unroll
  for (int i = 0; i < B_PC; i++) {
    const int n = n0 + (wid + NW * i) * 8 + (lane >> 3);
    b_off[i] = (n < g.N) ? ((unsigned)n * (unsigned)g.K + (unsigned)chunk * 8u) * 2u : kOob;
  }
  // conv: (tap, channel) of this lane's chunk in the NEXT tile to be issued, advanced by 64 per tile
  int c_cc = 0, c_dy = 0, c_dx = 0;
  if constexpr (CONV) {
    const int kk = chunk * 8, tap = kk / g.cC;
    c_cc = kk - tap * g.cC;
    c_dy = tap / g.cKs;
    c_dx = tap - c_dy * g.cKs;
  }

  const int nk = (g.K + BK - 1) / BK;
  // issue tile `kt` into ring stage `st` (tiles are issued in increasing kt order, exactly once each)
  auto issue = [&](int st, int kt) {
    if (MSLAM_GEMM_ABLATE == 1 && kt >= S - 1) return;
    unsigned char* sbase = smem + st * STAGE + wid * 1024;
    const int k0 = kt * BK;
    // Validity is folded into the offset with sign-bit arithmetic (bit 31 set = out of range = zero fill).
    // A select here would be turned into divergent control flow around the DMA instruction, which both
    // doubles the instruction count and breaks the vmcnt bookkeeping below.
    const unsigned k_bad = (unsigned)(g.K - 1 - (k0 + chunk * 8)) & kOob;   // set only in a partial last tile
#pragma unroll
    for (int i = 0; i < A_PC; i++) {
      unsigned off;
      if constexpr (CONV) {
        const int iy = a_iy0[i] + c_dy, ix = a_ix0[i] + c_dx;
        const unsigned bad = (unsigned)(iy | (g.cH - 1 - iy) | ix | (g.cW - 1 - ix)) & kOob;
        off = (a_off[i] + (unsigned)((iy * g.cW + ix) * g.cC + c_cc) * 2u) | bad | k_bad;
      } else {
        off = (a_off[i] + (unsigned)k0 * 2u) | k_bad;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, MSLAM_LDS_PTR(sbase + i * (NW * 1024)), 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_PC; i++) {
      const unsigned off = (b_off[i] + (unsigned)k0 * 2u) | k_bad;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, MSLAM_LDS_PTR(sbase + BM * 128 + i * (NW * 1024)), 16, off, 0, 0, 0);
    }
    if constexpr (CONV) {
      c_cc += BK;
      while (c_cc >= g.cC) {
        c_cc -= g.cC;
        if (++c_dx == g.cKs) { c_dx = 0; c_dy++; }
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = 0.0f;
#if MSLAM_GEMM_ABLATE == 4
  typedef __attribute__((ext_vector_type(4))) float f32x4q;
  f32x4q acc4[MI][NI][4];
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc4[mi][ni][r] = f32x4q{0.0f, 0.0f, 0.0f, 0.0f};
#endif

  // fragment addressing: row lr of the wave's 32-row slab, K half h; slot = (2*ks + h) ^ ((lr >> 1) & 7)
  const int lr = lane & 31, kh = (lane >> 5) ^ ((lr >> 1) & 7);
  const unsigned fragA = (wm * MI * 32 + lr) * 128, fragB = BM * 128 + (wn * NI * 32 + lr) * 128;
  const bool relu_a = g.a_relu != 0;
  auto compute = [&](int st) {
    const unsigned char* sA = smem + st * STAGE + fragA;
    const unsigned char* sB = smem + st * STAGE + fragB;
    // The fragment reads of KG 16-deep steps are issued before the first MFMA that uses them: a wave pays
    // the LDS latency once per group instead of once per step (the MFMAs wait on counted lgkmcnt).  KG is
    // the whole tile unless the block runs 4 waves per SIMD (128-VGPR budget).
    constexpr int KS = BK / 16, KG = (NW >= 16) ? 2 : KS;
#pragma unroll
    for (int kg = 0; kg < KS; kg += KG) {
      bf16x8 af[KG][MI], bfr[KG][NI];
#pragma unroll
      for (int k = 0; k < KG; k++) {
        const int so = ((2 * (kg + k)) ^ kh) * 16;
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
          if (MSLAM_GEMM_ABLATE == 3) af[k][mi] = __builtin_bit_cast(bf16x8, u32x4_t{(unsigned)so, 1u, 2u, 3u});
          else af[k][mi] = *reinterpret_cast<const bf16x8*>(sA + mi * 4096 + so);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ni++) {
          if (MSLAM_GEMM_ABLATE == 3) bfr[k][ni] = __builtin_bit_cast(bf16x8, u32x4_t{(unsigned)so, 5u, 6u, 7u});
          else bfr[k][ni] = *reinterpret_cast<const bf16x8*>(sB + ni * 4096 + so);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs (the scheduler would re-serialise them)
#pragma unroll
      for (int k = 0; k < KG; k++) {
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
          if (CONV && relu_a) {  // pre-activation of the DPT residual units: max(x, 0) on the int16 view
            const short8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            af[k][mi] = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(short8, af[k][mi]), z));
          }
#pragma unroll
          for (int ni = 0; ni < NI; ni++) {
            if (MSLAM_GEMM_ABLATE == 2) {   // keep the fragments alive without the matrix instruction
              asm volatile("" ::"v"(af[k][mi]), "v"(bfr[k][ni]));
#if MSLAM_GEMM_ABLATE == 4
            } else if (true) {
              acc4[mi][ni][2 * (k & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k][mi], bfr[k][ni], acc4[mi][ni][2 * (k & 1)], 0, 0, 0);
              acc4[mi][ni][2 * (k & 1) + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k][mi], bfr[k][ni], acc4[mi][ni][2 * (k & 1) + 1], 0, 0, 0);
#endif
            } else {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[k][mi], bfr[k][ni], acc[mi][ni], 0, 0, 0);
            }
          }
        }
      }
      if (kg + KG < KS) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- main loop -----------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < S - 1; s++)
    if (s < nk) issue(s, s);
  int kt = 0, st = 0, st_issue = S - 1;   // st = kt % S, st_issue = (kt + S - 1) % S
  for (; kt < nk - (S - 2); kt++) {
    wait_vmcnt<N_DMA * (S - 2)>();         // tile kt has landed (this wave's pieces); kt+1 .. kt+S-2 in flight
    __builtin_amdgcn_s_barrier();          // everyone's pieces landed; everyone is done reading stage (kt-1) % S
    if (kt + S - 1 < nk) issue(st_issue, kt + S - 1);
    compute(st);
    st = (st + 1 == S) ? 0 : st + 1;
    st_issue = (st_issue + 1 == S) ? 0 : st_issue + 1;
  }
  for (; kt < nk; kt++) {                  // drain: nk-1-kt tiles still in flight behind this one
    const int behind = nk - 1 - kt;        // < S-2 here
    bool waited = false;
    if constexpr (S >= 4) {
      static_for<S - 3>([&](auto j_c) {
        constexpr int b = decltype(j_c)::value + 1;
        if (behind == b) { wait_vmcnt<N_DMA * b>(); waited = true; }
      });
    }
    if (!waited) wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    compute(st);
    st = (st + 1 == S) ? 0 : st + 1;
  }

#if MSLAM_GEMM_ABLATE == 4
#pragma unroll
  for (int mi = 0; mi < MI; mi++)
#pragma unroll
    for (int ni = 0; ni < NI; ni++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[mi][ni][r] = acc4[mi][ni][r >> 2][r & 3];
#endif
  // ---- epilogue ------------------------------------------------------------------------------
  // Every accumulator index below is a compile-time constant (static_for): a runtime-indexed
  // ext_vector array would be demoted to scratch memory and spilled inside the K loop.
  // The MFMA C layout has one COLUMN per lane, i.e. 2-byte scattered stores; a 32x32 tile is therefore
  // turned through a per-wave LDS patch (the ring is free now) so that every lane owns 8 consecutive
  // columns of a row: 16-byte loads of the residuals, 16-byte stores of the result.
  __builtin_amdgcn_s_barrier();   // all waves are done reading the ring (every DMA was waited for above)
  constexpr int TBS = 40;         // patch row stride in floats (16-byte aligned rows)
  float* tb = reinterpret_cast<float*>(smem) + wid * (32 * TBS);
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8s;
  const int half = lane >> 5;
  const int lrow0 = lane >> 2, lc = (lane & 3) * 8;
  const bool wide_plain = g.epi == EPI_PLAIN && (g.N & 7) == 0 && (g.ldc & 7) == 0 && (g.ldr1 & 7) == 0 && (g.ldr2 & 7) == 0;
  const bool wide_attn = g.epi == EPI_ATTN && (g.ntok & 31) == 0 && (g.kv_ntok & 31) == 0;
  auto load8 = [&](const void* base, int kind, size_t idx, float (&o)[8]) {
    if (kind == KIND_F32) {
      const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
      const float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx + 4);
      o[0] += p.x; o[1] += p.y; o[2] += p.z; o[3] += p.w; o[4] += q.x; o[5] += q.y; o[6] += q.z; o[7] += q.w;
    } else if (kind == KIND_BF16) {
      const bf16x8s p = *reinterpret_cast<const bf16x8s*>(reinterpret_cast<const bf16*>(base) + idx);
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] += (float)p[e];
    }
  };
  static_for<MI>([&](auto mi_c) {
    static_for<NI>([&](auto ni_c) {
      constexpr int mi = decltype(mi_c)::value, ni = decltype(ni_c)::value;
      const f32x16 accv = acc[mi][ni];
      const int col_base = n0 + wn * NI * 32 + ni * 32;
      const int col = col_base + lcol;
      const int row_base0 = m0 + wm * MI * 32 + mi * 32;
      const int row_base = row_base0 + 4 * half;
      const bool col_ok = col < g.N;
      const float bias = bias_v[ni];
      float v[16];
      static_for<8>([&](auto r_c) {
        constexpr int r = 2 * decltype(r_c)::value;
        f32x2 x = {accv[r] + bias, accv[r + 1] + bias};
        if (g.act == ACT_GELU) x = gelu_erf2(x);
        else if (g.act == ACT_RELU) x = __builtin_elementwise_max(x, f32x2{0.0f, 0.0f});
        v[r] = x[0]; v[r + 1] = x[1];
      });
      if (wide_plain) {
        static_for<16>([&](auto r_c) {
          constexpr int r = decltype(r_c)::value;
          tb[((r & 3) + 8 * (r >> 2) + 4 * half) * TBS + lcol] = v[r];
        });
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int lrow = lrow0 + 16 * j;
          const float4 p = *reinterpret_cast<const float4*>(tb + lrow * TBS + lc);
          const float4 q = *reinterpret_cast<const float4*>(tb + lrow * TBS + lc + 4);
          const int row = row_base0 + lrow, c0 = col_base + lc;
          if (row < g.M && c0 < g.N) {
            float o[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
            load8(g.res1, g.res1_kind, (size_t)row * g.ldr1 + c0, o);
            load8(g.res2, g.res2_kind, (size_t)row * g.ldr2 + c0, o);
            if (g.out_kind == KIND_F32) {
              float* dst = reinterpret_cast<float*>(g.out) + (size_t)row * g.ldc + c0;
              *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
              *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
              bf16x8s pk;
#pragma unroll
              for (int e = 0; e < 8; e++) pk[e] = (bf16)o[e];
              *reinterpret_cast<bf16x8s*>(reinterpret_cast<bf16*>(g.out) + (size_t)row * g.ldc + c0) = pk;
            }
          }
        }
      } else if (wide_attn) {
        // a 32-wide tile is one RoPE half of one head; a 32-row tile lies inside one image (ntok % 32 == 0)
        const int sec = g.sec_base + col_base / g.sec_dim;
        const int cs = col_base % g.sec_dim;
        const int head = cs >> 6, f0 = cs & 63;   // f0 = 0 or 32
        const int ntok = (sec == 0) ? g.ntok : g.kv_ntok;
        const int b = row_base0 / ntok, nb = row_base0 - b * ntok;
        if (sec < 2) {
          const bool use_y = f0 < 32;
          const bool lo = lcol < 16;
          const float scale = (sec == 0) ? g.q_scale : 1.0f;
          static_for<16>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            const int n = nb + 4 * half + (r & 3) + 8 * (r >> 2);
            const float partner = __shfl_xor(v[r], 16, 64);
            const int p = use_y ? n / g.tok_w : n % g.tok_w;
            const float c = g.rope_cos[p * 16 + (lcol & 15)], sn = g.rope_sin[p * 16 + (lcol & 15)];
            const float x = lo ? (v[r] * c - partner * sn) : (v[r] * c + partner * sn);
            tb[((r & 3) + 8 * (r >> 2) + 4 * half) * TBS + lcol] = x * scale;
          });
          bf16* dst = (sec == 0) ? g.q_out : g.k_out;
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const int lrow = lrow0 + 16 * j;
            const float4 p = *reinterpret_cast<const float4*>(tb + lrow * TBS + lc);
            const float4 q = *reinterpret_cast<const float4*>(tb + lrow * TBS + lc + 4);
            if (row_base0 + lrow < g.M && col_base < g.N) {
              bf16x8s pk;
              pk[0] = (bf16)p.x; pk[1] = (bf16)p.y; pk[2] = (bf16)p.z; pk[3] = (bf16)p.w;
              pk[4] = (bf16)q.x; pk[5] = (bf16)q.y; pk[6] = (bf16)q.z; pk[7] = (bf16)q.w;
              *reinterpret_cast<bf16x8s*>(dst + (((size_t)b * g.heads + head) * ntok + nb + lrow) * 64 + f0 + lc) = pk;
            }
          }
        } else {
          // v, transposed per head: lane = feature f, 16 consecutive tokens -> two 16-byte stores
          static_for<16>([&](auto r_c) {
            constexpr int r = decltype(r_c)::value;
            tb[((r & 3) + 8 * (r >> 2) + 4 * half) * TBS + lcol] = v[r];
          });
          const int tok0 = half * 16;
          if (row_base0 + tok0 < g.M && col_ok) {
            bf16x8s pk0, pk1;
#pragma unroll
            for (int e = 0; e < 8; e++) {
              pk0[e] = (bf16)tb[(tok0 + e) * TBS + lcol];
              pk1[e] = (bf16)tb[(tok0 + 8 + e) * TBS + lcol];
            }
            bf16* dst = g.vt_out + (((size_t)b * g.heads + head) * 64 + f0 + lcol) * ntok + nb + tok0;
            *reinterpret_cast<bf16x8s*>(dst) = pk0;
            *reinterpret_cast<bf16x8s*>(dst + 8) = pk1;
          }
        }
      } else {
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
      }
    });
  });
}


// ---- launcher for one instantiation ----------------------------------------------------------
template <int WM, int WN, int MI, int NI, int S>
int launch_cfg(const GemmArgs& a, hipStream_t stream) {
  constexpr int NW = WM * WN, BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr size_t ring = (size_t)S * (BM + BN) * 128, patch = (size_t)NW * 32 * 40 * sizeof(float);
  constexpr size_t shmem = ring > patch ? ring : patch;
  static_assert(shmem <= 160 * 1024, "LDS budget");
  const unsigned blocks = (unsigned)(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN)) * (a.groups > 1 ? 2u : 1u) +
                          ((a.pf_bytes[0] + a.pf_bytes[1]) ? (unsigned)kGemmPrefetchBlocks : 0u);
  static bool attr_set = false;   // > 64 KiB of dynamic LDS needs the opt-in, once per instantiation
  if (!attr_set) {
    const void* fns[2] = {(const void*)gemm_bf16_kernel<WM, WN, MI, NI, S, false>,
                          (const void*)gemm_bf16_kernel<WM, WN, MI, NI, S, true>};
    for (const void* fn : fns) {
      int rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "gemm: attr");
      if (rc) return rc;
    }
    attr_set = true;
  }
  if (!a.a_conv)
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, MI, NI, S, false>), dim3(blocks), dim3(64 * NW), shmem, stream, a);
  else
    hipLaunchKernelGGL((gemm_bf16_kernel<WM, WN, MI, NI, S, true>), dim3(blocks), dim3(64 * NW), shmem, stream, a);
  return check_hip(hipGetLastError(), "gemm launch");
}

}  // namespace mslam
