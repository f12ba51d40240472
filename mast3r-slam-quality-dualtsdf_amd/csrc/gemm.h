// bf16 MFMA GEMM for gfx950 with fused prologue/epilogue, the work-horse of the MASt3R forward.
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )      fp32 accumulate (v_mfma_f32_32x32x16_bf16)
// A is either a dense row-major bf16 matrix or an implicit im2col view of an NHWC bf16 tensor
// (3x3 / 1x1 convolution, stride 1 or 2, zero padding) - no im2col buffer is ever materialised.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mslam {

typedef __bf16 bf16;

enum GemmAct { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };
enum GemmKind { KIND_NONE = 0, KIND_F32 = 1, KIND_BF16 = 2 };
enum GemmEpi {
  EPI_PLAIN = 0,      // out[m*ldc + n]
  EPI_ATTN = 1,       // attention projection: RoPE2D + head split (q,k -> [B,H,Ntok,64]; v -> [B,H,64,Ntok])
  EPI_CONVT = 2,      // ConvTranspose2d with kernel == stride: pixel scatter into NHWC
};

struct GemmArgs {
  const bf16* A;
  const bf16* W;  // [N, K] row-major
  int M, N, K;
  int lda;
  // implicit-conv view of A (a_conv != 0): NHWC input [B, cH, cW, cC], k = tap*cC + c
  int a_conv, cB, cH, cW, cC, cKs, cStride, cPad, cHo, cWo;
  int a_relu;  // ReLU applied to A on load (pre-activation of the DPT residual units)
  // epilogue
  const float* bias;
  int act;
  const void* res1; int res1_kind; int ldr1;
  const void* res2; int res2_kind; int ldr2;
  void* out; int out_kind; int ldc;
  int epi;
  // EPI_ATTN
  bf16* q_out; bf16* k_out; bf16* vt_out;
  int sec_base;      // section id of column 0 (0 q, 1 k, 2 v); sections are `sec_dim` columns wide
  int sec_dim;       // heads * 64
  int heads;
  int ntok;          // tokens per image (rows per batch element)
  int tok_w;         // tokens per image row (RoPE x period)
  int kv_ntok;       // tokens per image of the k/v side (== ntok for self attention)
  const float* rope_cos; const float* rope_sin;  // [max_pos][16]
  float q_scale;
  // EPI_CONVT
  int ct_s, ct_cout, ct_h, ct_w;
  // Grouped launch (groups == 2): a second, independent problem of the SAME shape runs in the same grid
  // (the two sides of a decoder layer).  Activations of the second problem sit at fixed strides behind the
  // first one's; its weights and bias are separate allocations.
  int groups;                    // 0 / 1 = single problem
  const bf16* W1; const float* bias1;
  size_t a_gstride;              // elements of A
  size_t out_gbytes, res1_gbytes, res2_gbytes;   // bytes (the kinds differ)
  size_t qkv_gstride;            // elements of q_out / k_out / vt_out
  // Weight prefetch (optional): up to two byte ranges (the weights a LATER launch will need) are streamed by
  // a few extra blocks of this grid, so that they sit in the memory-side Infinity Cache when their GEMM starts
  // (measured: 15.3 us with HBM-cold weights vs 11.3 us with cache-resident ones, tools/cold_probe.py).
  const void* pf_ptr[2]; size_t pf_bytes[2];
  unsigned* pf_sink;             // 4 writable bytes; written only if a checksum hits a magic value (keeps the loads alive)
};

constexpr int kGemmPrefetchBlocks = 64;   // x 256+ lanes x 16 loads x 16 B = 4 MiB in flight

int launch_gemm(const GemmArgs& a, hipStream_t stream);
int gemm_profile_begin(int M, int N, int K, int max_samples);   // see gemm.hip
int gemm_profile_end(double* avg_us, double* min_us, int* samples);
int gemm_tile_override(int M, int N, int K, int cfg, bool conv);   // cfg codes of gemm.hip; 0 removes the entry

}  // namespace mslam
