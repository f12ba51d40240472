// 256x256 bf16 MFMA GEMM with a phase-interleaved K loop for the large plain-epilogue shapes of the network (encoder
// fc1 / fc2 / proj at a frame group, backend batches):  C[M,N] = epilogue(A[M,K] . W[N,K]^T), fp32 accumulate.
//
// Why a second kernel: gemm_kernel.h synchronises once per 64-deep K tile and every wave then reads ALL its fragments and
// issues ALL its MFMAs - the eight waves of a block hit the LDS pipe together and the matrix pipe together, one after the
// other (profiles/r02_gemm_ablation.log: the K loop takes 2.3x its MFMA time and still 24.7 of 34.7 us with the MFMAs
// removed).  Here the K tile is cut into four phases of one C quadrant each (8 x v_mfma_f32_32x32x16_bf16 per wave: the same
// instruction and the same K order per output element as gemm_kernel.h, so results are BIT-IDENTICAL to every other tile), and
// the two wave rows run ONE BARRIER APART: while the four waves of one row issue their MFMAs (one wave per SIMD), the
// other four - the second wave of each SIMD - read their next fragments from LDS and issue the DMA of a later half
// tile, so the LDS pipe and the matrix pipe work at the same time instead of in turns.
//
//   block 256 x 256, 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = quadrants (A-half mh, B-half nh) of 64 x 32
//   LDS 128 KiB = 2 K-tile buffers x 4 half-tile slots of 128 rows x 128 B:
//       G0 = the A rows every wave needs in phases 0/1 (first 64 rows of both wave rows),  G1 = those of phases 2/3,
//       G2 = the W rows (output columns) of phases 0 and 3 (first 32 columns of the four wave columns),  G3 = the others
//   phase p of K tile t:   p0 reads A0 + B0 -> Q(0,0);  p1 reads B1 -> Q(0,1);  p2 reads A1 -> Q(1,1);  p3 -> Q(1,0)
//   every phase stages ONE half tile (2 LDS-DMA instructions per wave), sequence number g + 6 of the stream
//   [G0, G2, G3, G1](tile 0), [G0, G2, G3, G1](tile 1), ...: each half tile is issued >= 4 phases (a whole K tile) before
//   its first read and never earlier than 2 phases after the last read of the slot it overwrites; a uniform counted
//   s_waitcnt vmcnt(8) (four half tiles stay in flight) in front of the phase's first barrier retires what the NEXT phase
//   reads.  Staging is LDS-DMA with the bank swizzle on the source side, as in gemm_kernel.h.
// Epilogue: bias / GELU / ReLU in the accumulator layout, then 32 x 32 patches through LDS so that a lane owns 8
// consecutive columns of a row (16-byte residual loads and result stores), as gemm_kernel.h does.  Plain epilogue only (no RoPE / conv-transpose
// scatter, no implicit-conv view, no grouped launch): launch_gemm routes those to gemm_kernel.h.
#include "gemm_kernel.h"

namespace mslam {

namespace {

constexpr int P8_BM = 256, P8_BN = 256;
constexpr int P8_SLOT = 128 * 128;            // bytes per half-tile slot
constexpr int P8_BUF = 4 * P8_SLOT;           // bytes per K-tile buffer
constexpr int P8_LDS = 2 * P8_BUF;            // 128 KiB
constexpr int P8_PATCH_STRIDE = 40;           // floats per patch row (16-byte aligned)

__global__ __launch_bounds__(512) void gemm8p_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const unsigned nbm = (g.M + P8_BM - 1) / P8_BM, nbn = (g.N + P8_BN - 1) / P8_BN;
  const unsigned tile = xcd_remap(blockIdx.x, nbm * nbn);
  constexpr unsigned GM = 4;                  // row panels per group (~1024 rows): see gemm_kernel.h
  const unsigned per_group = GM * nbn, grp = tile / per_group, in_grp = tile - grp * per_group;
  const unsigned gsz = min(nbm - grp * GM, GM);
  const int m0 = (grp * GM + in_grp % gsz) * P8_BM, n0 = (in_grp / gsz) * P8_BN;

  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wr = wid >> 2, wc = wid & 3;

  // ---- DMA source addressing: piece q = wid + 8 i covers slot rows 8q .. 8q+7; lane -> row 8q + (lane >> 3), 16-byte
  // position lane & 7, which holds K chunk (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ (4 (wid & 1) + (lane >> 4))
  const int chunk = (lane & 7) ^ (4 * (wid & 1) + (lane >> 4));
  const size_t a_bytes = ((size_t)(g.M - 1) * g.lda + g.K) * 2;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW =
      __builtin_amdgcn_make_buffer_rsrc((void*)g.W, 0, (int)((size_t)g.N * g.K * 2), 0x00020000);
  unsigned a_off[2][2], b_off[2][2];          // [half][piece]: byte offset of (row, chunk) at k0 = 0, or kOob
#pragma unroll
  for (int hf = 0; hf < 2; hf++)
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int sr = (wid + 8 * i) * 8 + (lane >> 3);                        // slot row 0..127
      const int m = m0 + (sr & 63) + 128 * (sr >> 6) + 64 * hf;             // A: first / second 64 rows of both wave rows
      a_off[hf][i] = (m < g.M) ? ((unsigned)m * (unsigned)g.lda + (unsigned)chunk * 8u) * 2u : kOob;
      const int n = n0 + (sr >> 5) * 64 + (sr & 31) + 32 * hf;              // W: first / second 32 columns of the wave columns
      b_off[hf][i] = (n < g.N) ? ((unsigned)n * (unsigned)g.K + (unsigned)chunk * 8u) * 2u : kOob;
    }
  const int nk = (g.K + BK - 1) / BK;
  // half tile KIND of K tile kt; the stream is [G0, G2, G3, G1](tile 0), [G0, G2, G3, G1](tile 1), ...
  //   KIND 0 = G0 (slot 0, A first halves), 1 = G2 (slot 2, W first halves), 2 = G3 (slot 3), 3 = G1 (slot 1)
  auto stage = [&](auto kind_c, int kt) {
    constexpr int kind = decltype(kind_c)::value;
    constexpr int slot = (kind == 0) ? 0 : (kind == 1) ? 2 : (kind == 2) ? 3 : 1;
    constexpr bool is_a = (kind == 0) || (kind == 3);
    constexpr int hf = (kind == 0 || kind == 1) ? 0 : 1;
    const int k0 = kt * BK;
    // beyond the last K tile (and in the K tail) the offsets are out of range: the DMA zero-fills, the instruction
    // count - which the vmcnt bookkeeping relies on - stays uniform
    const unsigned k_bad = ((unsigned)(g.K - 1 - (k0 + chunk * 8)) & kOob) | (kt >= nk ? kOob : 0u);
    unsigned char* dst = smem + (kt & 1) * P8_BUF + slot * P8_SLOT + wid * 1024;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const unsigned off = ((is_a ? a_off[hf][i] : b_off[hf][i]) + (unsigned)k0 * 2u) | k_bad;
      if constexpr (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, MSLAM_LDS_PTR(dst + i * 8192), 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, MSLAM_LDS_PTR(dst + i * 8192), 16, off, 0, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  // ---- fragment addressing (32x32x16 operands): lane -> row lr of a 32-row slab, K half h = lane >> 5 of a 16-deep step;
  // 16-byte position of step ks: (2 ks + h) ^ ((lr >> 1) & 7)
  const int lr = lane & 31, kh = (lane >> 5) ^ ((lr >> 1) & 7);
  const unsigned fa = (unsigned)(wr * 64 + lr) * 128u, fb = (unsigned)(wc * 32 + lr) * 128u;
  bf16x8 a[2][4], b0[4], b1[4];
  f32x16 acc[2][2][2];                        // [A half][W half][32-row tile of the half]
#pragma unroll
  for (int mh = 0; mh < 2; mh++)
#pragma unroll
    for (int nh = 0; nh < 2; nh++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mh][nh][i][r] = 0.0f;

  auto read_a = [&](const unsigned char* slot) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int ks = 0; ks < 4; ks++) a[i][ks] = *reinterpret_cast<const bf16x8*>(slot + fa + i * 4096 + ((2 * ks) ^ kh) * 16);
  };
  auto read_b = [&](const unsigned char* slot, bf16x8 (&bb)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++) bb[ks] = *reinterpret_cast<const bf16x8*>(slot + fb + ((2 * ks) ^ kh) * 16);
  };
  auto quadrant = [&](f32x16 (&c)[2], const bf16x8 (&bb)[4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
#pragma unroll
      for (int i = 0; i < 2; i++) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], bb[ks], c[i], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
  // one phase: [reads of this phase] [stage] [counted wait] barrier | lgkmcnt(0) MFMAs | barrier
#define P8_PHASE_HEAD(KIND, KT)                        \
  stage(KIND{}, KT);                                   \
  wait_vmcnt<8>();                                     \
  __builtin_amdgcn_s_barrier();                        \
  __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ \
  __builtin_amdgcn_sched_barrier(0);
#define P8_PHASE_TAIL()            \
  __builtin_amdgcn_sched_barrier(0); \
  __builtin_amdgcn_s_barrier();

  // ---- prologue: the first six half tiles of the stream; the first two (G0, G2 of tile 0) must have landed
  stage(K0{}, 0); stage(K1{}, 0); stage(K2{}, 0); stage(K3{}, 0); stage(K0{}, 1); stage(K1{}, 1);
  wait_vmcnt<8>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();    // the second wave row runs one barrier behind the first

  for (int kt = 0; kt < nk; kt++) {
    const unsigned char* buf = smem + (kt & 1) * P8_BUF;
    // phase 0: B0 (G2), A0 (G0) -> Q(0,0)
    read_b(buf + 2 * P8_SLOT, b0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf + 0 * P8_SLOT);
    P8_PHASE_HEAD(K2, kt + 1)
    quadrant(acc[0][0], b0);
    P8_PHASE_TAIL()
    // phase 1: B1 (G3) -> Q(0,1)
    read_b(buf + 3 * P8_SLOT, b1);
    P8_PHASE_HEAD(K3, kt + 1)
    quadrant(acc[0][1], b1);
    P8_PHASE_TAIL()
    // phase 2: A1 (G1) -> Q(1,1)
    read_a(buf + 1 * P8_SLOT);
    P8_PHASE_HEAD(K0, kt + 2)
    quadrant(acc[1][1], b1);
    P8_PHASE_TAIL()
    // phase 3: -> Q(1,0)
    P8_PHASE_HEAD(K1, kt + 2)
    quadrant(acc[1][0], b0);
    P8_PHASE_TAIL()
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();    // re-align the two wave rows
  wait_vmcnt<0>();                              // the zero-fill DMAs of the tail have landed: the ring is free
  __builtin_amdgcn_s_barrier();
#undef P8_PHASE_HEAD
#undef P8_PHASE_TAIL

  // ---- epilogue ------------------------------------------------------------------------------
  float* tb = reinterpret_cast<float*>(smem) + wid * (32 * P8_PATCH_STRIDE);
  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8s;
  const int lcol = lane & 31, half = lane >> 5;
  const int lrow0 = lane >> 2, lc = (lane & 3) * 8;
  float bias_v[2];
#pragma unroll
  for (int nh = 0; nh < 2; nh++) {
    const int col = n0 + wc * 64 + nh * 32 + lcol;
    bias_v[nh] = (g.bias && col < g.N) ? g.bias[col] : 0.0f;
  }
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
  // G0 holds tile rows [0,64) and [128,192), G1 rows [64,128) and [192,256); wave row wr reads slot rows wr*64 .. +63 of
  // each, i.e. it owns tile rows wr*128 + mh*64 .. +63; wave column wc owns tile columns wc*64 + nh*32 .. +31
  static_for<2>([&](auto mh_c) {
    static_for<2>([&](auto i_c) {
      static_for<2>([&](auto nh_c) {
        constexpr int mh = decltype(mh_c)::value, i = decltype(i_c)::value, nh = decltype(nh_c)::value;
        const f32x16 accv = acc[mh][nh][i];
        const int row_base0 = m0 + wr * 128 + mh * 64 + i * 32;
        const int col_base = n0 + wc * 64 + nh * 32;
        static_for<8>([&](auto r_c) {
          constexpr int r = 2 * decltype(r_c)::value;
          f32x2 x = {accv[r] + bias_v[nh], accv[r + 1] + bias_v[nh]};
          if (g.act == ACT_GELU) x = gelu_erf2(x);
          else if (g.act == ACT_RELU) x = __builtin_elementwise_max(x, f32x2{0.0f, 0.0f});
          tb[((r & 3) + 8 * (r >> 2) + 4 * half) * P8_PATCH_STRIDE + lcol] = x[0];
          tb[(((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * half) * P8_PATCH_STRIDE + lcol] = x[1];
        });
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int lrow = lrow0 + 16 * j;
          const float4 p = *reinterpret_cast<const float4*>(tb + lrow * P8_PATCH_STRIDE + lc);
          const float4 q = *reinterpret_cast<const float4*>(tb + lrow * P8_PATCH_STRIDE + lc + 4);
          const int row = row_base0 + lrow, c0g = col_base + lc;
          if (row < g.M && c0g < g.N) {
            float o[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
            load8(g.res1, g.res1_kind, (size_t)row * g.ldr1 + c0g, o);
            load8(g.res2, g.res2_kind, (size_t)row * g.ldr2 + c0g, o);
            if (g.out_kind == KIND_F32) {
              float* dst = reinterpret_cast<float*>(g.out) + (size_t)row * g.ldc + c0g;
              *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
              *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
              bf16x8s pk;
#pragma unroll
              for (int e = 0; e < 8; e++) pk[e] = (bf16)o[e];
              *reinterpret_cast<bf16x8s*>(reinterpret_cast<bf16*>(g.out) + (size_t)row * g.ldc + c0g) = pk;
            }
          }
        }
      });
    });
  });
}

}  // namespace

bool gemm8p_supports(const GemmArgs& a) {
  return !a.a_conv && a.groups <= 1 && a.epi == EPI_PLAIN && (a.N & 7) == 0 && (a.ldc & 7) == 0 && (a.ldr1 & 7) == 0 &&
         (a.ldr2 & 7) == 0 && ((uintptr_t)a.out & 15) == 0 && ((uintptr_t)a.res1 & 15) == 0 && ((uintptr_t)a.res2 & 15) == 0;
}

int launch_gemm_8p(const GemmArgs& a, hipStream_t stream) {
  MSLAM_REQUIRE(gemm8p_supports(a), "gemm8p: unsupported problem (plain epilogue, dense A, N %% 8 == 0 only)");
  static bool attr_set = false;
  if (!attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void*)gemm8p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS),
                       "gemm8p: attr");
    if (rc) return rc;
    attr_set = true;
  }
  const unsigned blocks = (unsigned)(((a.M + P8_BM - 1) / P8_BM) * ((a.N + P8_BN - 1) / P8_BN));
  hipLaunchKernelGGL(gemm8p_kernel, dim3(blocks), dim3(512), P8_LDS, stream, a);
  return check_hip(hipGetLastError(), "gemm8p launch");
}

}  // namespace mslam
