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
