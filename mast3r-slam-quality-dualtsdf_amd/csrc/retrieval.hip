// Retrieval database (ASMK with binarised residuals) for gfx950: aggregation of one image's local descriptors into
// binary signatures, and the Hamming-kernel search of the inverted file.
//
// Reference behaviour being reproduced (never copied; the per-word python lists of the reference become flat device
// arrays, the search one launch with a block per database image):
//   ASMKKernel.aggregate_image   thirdparty/mast3r/asmk/asmk/kernel.py:28-42
//   binarize_and_pack_2D         thirdparty/mast3r/asmk/cython/hamming.pyx:93-127 (c_binarize_and_pack_uint32 :24-30)
//   IVF.search                   thirdparty/mast3r/asmk/asmk/inverted_file.py:90-114 (use_idf False, processor.py:85)
//   ASMKKernel.similarity        kernel.py:59-71, hamming_cdist_packed hamming.pyx:135-152 (c_hamming_dist :33-41)
//   asmk_kernel                  thirdparty/mast3r/asmk/asmk/functional.py:10-15
//
// Arithmetic contract (shared with oracle/asmk_py.py): residual sums are sequential fp32 in descriptor order (numpy's
// axis-0 reduction), the signature bit is `sum > 0`; the normalised Hamming distance is int / float in fp32,
// sim = -2 h + 1 in fp32, sim^alpha rounded to fp32, divided by sqrt(entries of the image) in fp64 and rounded to fp32,
// accumulated in fp64 in ascending word order (the order the reference visits the query words in; an image's entries are
// stored words-ascending), and the total divided by the fp32 square root of the number of query words.
#include "common.h"

namespace mslam {

// One block per unique visual word of the image.  flag[f] = descriptor f is assigned to the word (any of its m
// assignments); thread -> descriptor dimension; a wave's ballot is 64 consecutive signature bits.
__global__ __launch_bounds__(256) void asmk_aggregate_kernel(
    const float* __restrict__ des, const float* __restrict__ cent, const int64_t* __restrict__ assign,
    const int64_t* __restrict__ uniq, uint32_t* __restrict__ sig, int n, int m, int dim, int n_cent) {
  extern __shared__ unsigned char flag[];
  const int u = blockIdx.x, W = dim >> 5;
  const int64_t word = uniq[u];
  const bool word_ok = word >= 0 && word < n_cent;      // block-uniform
  for (int f = threadIdx.x; f < n; f += 256) {
    bool hit = false;
    for (int j = 0; j < m; j++) hit |= assign[(size_t)f * m + j] == word;
    flag[f] = hit;
  }
  __syncthreads();
  const float* c = cent + (size_t)(word_ok ? word : 0) * dim;
  for (int d0 = 0; d0 < dim; d0 += 256) {
    const int d = d0 + threadIdx.x;
    float acc = 0.0f;
    if (d < dim && word_ok) {
      const float cd = c[d];
      bool first = true;
      for (int f = 0; f < n; f++) {
        if (!flag[f]) continue;
        const float r = des[(size_t)f * dim + d] - cd;
        acc = first ? r : acc + r;
        first = false;
      }
    }
    const unsigned long long b = __ballot(d < dim && acc > 0.0f);
    if ((threadIdx.x & 63) == 0) {
      const int w0 = (d0 + threadIdx.x) >> 5;            // dimension 32*w0 is this wave's lane 0
      if (w0 < W) sig[(size_t)u * W + w0] = __brev((unsigned)(b & 0xffffffffull));        // first dimension -> MSB
      if (w0 + 1 < W) sig[(size_t)u * W + w0 + 1] = __brev((unsigned)(b >> 32));
    }
  }
}

__device__ __forceinline__ int find_word(const int32_t* __restrict__ q_words, int nq, int w) {
  int lo = 0, hi = nq;                                   // first index with q_words[i] >= w
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (q_words[mid] < w) lo = mid + 1; else hi = mid;
  }
  return (lo < nq && q_words[lo] == w) ? lo : -1;
}

// One block per database image; its entries are contiguous (images are added whole) and word-ascending.
__global__ __launch_bounds__(256) void asmk_search_kernel(
    const int32_t* __restrict__ e_word, const uint32_t* __restrict__ e_sig, const int32_t* __restrict__ img_start,
    const int32_t* __restrict__ q_words, const uint32_t* __restrict__ q_sig, int nq, int W, float thr, float alpha,
    double* __restrict__ scores) {
  __shared__ float contrib[256];
  const int img = blockIdx.x;
  const int s = img_start[img], e = img_start[img + 1];
  const double root_nf = sqrt((double)(e - s));          // norm_factor[image] = its number of entries without idf
  const float nbits = (float)(W * 32);
  double total = 0.0;
  for (int base = s; base < e; base += 256) {
    const int idx = base + (int)threadIdx.x;
    float c = 0.0f;
    if (idx < e) {
      const int qi = find_word(q_words, nq, e_word[idx]);
      if (qi >= 0) {
        const uint32_t* a = e_sig + (size_t)idx * W;
        const uint32_t* b = q_sig + (size_t)qi * W;
        int cnt = 0;
        if ((W & 3) == 0) {
          for (int k = 0; k < W; k += 4) {
            const uint4 x = *reinterpret_cast<const uint4*>(a + k), y = *reinterpret_cast<const uint4*>(b + k);
            cnt += __popc(x.x ^ y.x) + __popc(x.y ^ y.y) + __popc(x.z ^ y.z) + __popc(x.w ^ y.w);
          }
        } else {
          for (int k = 0; k < W; k++) cnt += __popc(a[k] ^ b[k]);
        }
        const float h = (float)cnt / nbits;
        const float sim = -2.0f * h + 1.0f;
        if (sim >= thr) {
          const float p = (alpha == 3.0f) ? (float)((double)sim * (double)sim * (double)sim) : powf(sim, alpha);
          c = (float)((double)p / root_nf);
        }
      }
    }
    contrib[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int cntb = min(256, e - base);
      for (int k = 0; k < cntb; k++) total += (double)contrib[k];      // entry order = ascending word order
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) scores[img] = total / (double)sqrtf((float)nq);
}

// ---------------------------------------------------------------------------------------------------------------------
// fp64 GEMM of the retrieval head on the f64 matrix cores (the same v_mfma_f64_16x16x4_f64 tile loop as the global GN's
// LL^T, gn.hip): out[M,N] = (A[M,K] - centre[K]) . B[K,N] (+ bias[N]), A and out row-major, B row-major [K,N].
//   Whitener.forward (thirdparty/mast3r/mast3r/retrieval/model.py:62-77): fp64 centre + matmul with p (d, d')
//   the projector's Linear layers (model.py:108-151) with fp64 accumulation of the fp32 operands (exact products)
// 64x64 tile per workgroup (4 waves, 32x32 each = 2x2 MFMA tiles), K in chunks of 32 through LDS; the centre is
// subtracted while the A chunk is staged.  768 x 1024 x 1024: 192 workgroups, 3.2 GFLOP.
// ---------------------------------------------------------------------------------------------------------------------
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_f64_kernel(const TA* __restrict__ A, const TB* __restrict__ B,
                                                       const double* __restrict__ centre, const double* __restrict__ bias,
                                                       double* __restrict__ out, int M, int N, int K, int lda, int ldb,
                                                       int b_transposed) {
  constexpr int T = 64, KC = 32, LS = KC + 2;
  __shared__ double As[T * LS], Bs[T * LS];          // As[row][k], Bs[col][k]
  const int r0 = blockIdx.y * T, c0 = blockIdx.x * T;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int fcol = lane & 15, frow = lane >> 4;      // C layout: col = lane & 15, row = (lane >> 4) + 4 reg
  const int fi = lane & 15, fk = lane >> 4;          // A / B operand: row (col) = lane & 15, k = lane >> 4
  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++) acc[m][n] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();
    for (int e = t; e < T * KC; e += 256) {
      const int rr = e / KC, q = e % KC;             // consecutive threads walk k: contiguous in A and in a transposed B
      const int k = k0 + q;
      double va = 0.0, vb = 0.0;
      if (r0 + rr < M && k < K) va = (double)A[(size_t)(r0 + rr) * lda + k] - (centre ? centre[k] : 0.0);
      if (b_transposed) { if (c0 + rr < N && k < K) vb = (double)B[(size_t)(c0 + rr) * ldb + k]; }
      As[rr * LS + q] = va;
      if (b_transposed) Bs[rr * LS + q] = vb;
    }
    if (!b_transposed) {
      for (int e = t; e < T * KC; e += 256) {
        const int q = e / T, cc = e % T;             // consecutive threads walk the columns of B's row k
        const int k = k0 + q;
        Bs[cc * LS + q] = (c0 + cc < N && k < K) ? (double)B[(size_t)k * ldb + c0 + cc] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int k4 = 0; k4 < KC; k4 += 4) {
      double a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        a[m] = As[(32 * wr + 16 * m + fi) * LS + k4 + fk];
        b[m] = Bs[(32 * wc + 16 * m + fi) * LS + k4 + fk];
      }
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
  }
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int r = r0 + 32 * wr + 16 * m + frow + 4 * g;
        const int c = c0 + 32 * wc + 16 * n + fcol;
        if (r < M && c < N) out[(size_t)r * N + c] = acc[m][n][g] + (bias ? bias[c] : 0.0);
      }
}

}  // namespace mslam

using namespace mslam;

extern "C" int mslam_asmk_aggregate(const float* des, const float* centroids, const int64_t* assign,
                                    const int64_t* uniq_words, uint32_t* sig_out, int n_des, int m_assign, int dim,
                                    int n_uniq, int n_centroids, void* stream) {
  MSLAM_REQUIRE(n_des >= 0 && m_assign >= 1 && n_uniq >= 0 && n_centroids >= 1, "asmk_aggregate: bad sizes");
  MSLAM_REQUIRE(dim >= 32 && (dim & 31) == 0, "asmk_aggregate: descriptor dimension %d is not a multiple of 32", dim);
  MSLAM_REQUIRE(n_des <= 60000, "asmk_aggregate: %d descriptors per image exceed the membership table in LDS", n_des);
  if (n_uniq == 0) return MSLAM_OK;
  MSLAM_REQUIRE(des && centroids && assign && uniq_words && sig_out, "asmk_aggregate: null pointer");
  hipLaunchKernelGGL(asmk_aggregate_kernel, dim3(n_uniq), dim3(256), (size_t)((n_des + 15) & ~15), (hipStream_t)stream,
                     des, centroids, assign, uniq_words, sig_out, n_des, m_assign, dim, n_centroids);
  MSLAM_LAUNCH_CHECK("asmk_aggregate");
  return MSLAM_OK;
}

extern "C" int mslam_asmk_search(const int32_t* entry_word, const uint32_t* entry_sig, const int32_t* img_start,
                                 int n_images, const int32_t* q_words, const uint32_t* q_sig, int n_q, int sig_words,
                                 float similarity_threshold, float alpha, double* scores, void* stream) {
  MSLAM_REQUIRE(n_images >= 0 && n_q >= 1 && sig_words >= 1, "asmk_search: bad sizes");
  if (n_images == 0) return MSLAM_OK;
  MSLAM_REQUIRE(entry_word && entry_sig && img_start && q_words && q_sig && scores, "asmk_search: null pointer");
  MSLAM_REQUIRE((sig_words & 3) != 0 || (((uintptr_t)entry_sig | (uintptr_t)q_sig) & 15) == 0,
                "asmk_search: signatures must be 16-byte aligned");
  hipLaunchKernelGGL(asmk_search_kernel, dim3(n_images), dim3(256), 0, (hipStream_t)stream, entry_word, entry_sig,
                     img_start, q_words, q_sig, n_q, sig_words, similarity_threshold, alpha, scores);
  MSLAM_LAUNCH_CHECK("asmk_search");
  return MSLAM_OK;
}

extern "C" int mslam_gemm_f64(const void* A, int a_is_f32, const void* B, int b_is_f32, int b_transposed,
                              const double* centre, const double* bias, double* out, int M, int N, int K, void* stream) {
  MSLAM_REQUIRE(A && B && out && M > 0 && N > 0 && K > 0, "gemm_f64: bad arguments");
  const dim3 grid((N + 63) / 64, (M + 63) / 64);
  const int lda = K, ldb = b_transposed ? K : N;
  hipStream_t s = (hipStream_t)stream;
  if (a_is_f32 && b_is_f32)
    hipLaunchKernelGGL((gemm_f64_kernel<float, float>), grid, dim3(256), 0, s, (const float*)A, (const float*)B, centre, bias,
                       out, M, N, K, lda, ldb, b_transposed);
  else if (a_is_f32)
    hipLaunchKernelGGL((gemm_f64_kernel<float, double>), grid, dim3(256), 0, s, (const float*)A, (const double*)B, centre,
                       bias, out, M, N, K, lda, ldb, b_transposed);
  else if (b_is_f32)
    hipLaunchKernelGGL((gemm_f64_kernel<double, float>), grid, dim3(256), 0, s, (const double*)A, (const float*)B, centre,
                       bias, out, M, N, K, lda, ldb, b_transposed);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<double, double>), grid, dim3(256), 0, s, (const double*)A, (const double*)B, centre,
                       bias, out, M, N, K, lda, ldb, b_transposed);
  MSLAM_LAUNCH_CHECK("gemm_f64");
  return MSLAM_OK;
}
