// Patch statistics of the quality service (mast3r_slam/quality_core.py): per-patch medians of the tracking
// residual and of the confidence-derived uncertainty, robust z-scores and the three-way patch classification.
// The reference runs these as torch ops on small reshaped copies (nanmedian over a (gh, gw, ps*ps) view); here
// one block sorts one patch in LDS, and one block classifies the whole patch grid.
//
// Bit-level conventions followed (torch CPU semantics): median of n values = the (n-1)//2-th smallest
// (torch.median / nanmedian return the LOWER middle); nanmedian ignores NaN and masked pixels, an empty patch
// gives NaN -> nan_to_num -> 0; every formula is evaluated in float32 in the reference's operation order.
#include "common.h"

namespace mslam {

constexpr int kQMax = 1024;   // ps * ps <= 1024 and patch-grid size <= 4096 (classify)

// ascending bitonic sort of n (power of two) floats in LDS by all threads of the block; NaN never enters (callers
// map NaN / masked values to +inf)
__device__ __forceinline__ void bitonic_sort(float* s, int n) {
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const float a = s[i], b = s[p];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { s[i] = b; s[p] = a; }
        }
      }
      __syncthreads();
    }
}

// mode 0: nanmedian of x over valid pixels (valid may be null), NaN -> 0 when a mask is given (reduce_grid)
// mode 1: mean over the patch (valid null) or nanmean over valid pixels, NaN -> 0
// mode 2: median of U = 1 - sqrt(clamp(clamp(C/(C_thr+1e-8),0,1) * clamp(Q/(Q_thr+1e-8),0,1), 0, 1))  (x = C, y = Q)
__global__ __launch_bounds__(kQMax) void quality_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const uint8_t* __restrict__ valid, int h, int w, int ps,
                                                               int mode, float c_div, float q_div, float* __restrict__ out) {
  __shared__ float s[kQMax];
  __shared__ int cnt;
  __shared__ float fsum;
  const int gw = w / ps, n = ps * ps;
  const int py = blockIdx.x / gw, px = blockIdx.x % gw;
  const int t = threadIdx.x;
  if (t == 0) { cnt = 0; fsum = 0.0f; }
  __syncthreads();
  float v = INFINITY;
  bool ok = false;
  if (t < n) {
    const size_t pix = (size_t)(py * ps + t / ps) * w + (size_t)(px * ps + t % ps);
    if (mode == 2) {
      const float cn = fminf(fmaxf(x[pix] / c_div, 0.0f), 1.0f), qn = fminf(fmaxf(y[pix] / q_div, 0.0f), 1.0f);
      v = 1.0f - sqrtf(fminf(fmaxf(cn * qn, 0.0f), 1.0f));
      ok = true;
    } else {
      v = x[pix];
      ok = (valid == nullptr || valid[pix] != 0) && !(v != v);
    }
  }
  if (mode == 1) {   // mean: fixed-order tree sum (the reference's vectorised order is not reproduced; tolerance 1e-6)
    s[t] = ok ? v : 0.0f;
    if (ok) atomicAdd(&cnt, 1);
    __syncthreads();
    for (int off = kQMax / 2; off > 0; off >>= 1) {
      if (t < off && t + off < (int)blockDim.x) s[t] += s[t + off];
      __syncthreads();
    }
    if (t == 0) {
      const int denom = valid ? cnt : n;
      out[blockIdx.x] = denom > 0 ? s[0] / (float)denom : 0.0f;
    }
    return;
  }
  s[t] = ok ? v : INFINITY;
  if (ok) atomicAdd(&cnt, 1);
  __syncthreads();
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = n + t; i < np2; i += blockDim.x) s[i] = INFINITY;
  __syncthreads();
  bitonic_sort(s, np2);
  if (t == 0) out[blockIdx.x] = cnt > 0 ? s[(cnt - 1) / 2] : 0.0f;   // empty patch: nanmedian = NaN -> nan_to_num
  (void)fsum;
}

// lower median of n <= 4096 floats already in s[0..n) (padded to np2 with +inf); result broadcast to all threads
__device__ __forceinline__ float block_median(float* s, int n, int np2) {
  bitonic_sort(s, np2);
  const float m = s[(n - 1) / 2];
  __syncthreads();
  return m;
}

// classify (quality_core.py:63-117) on the flattened patch grid; one block
__global__ __launch_bounds__(1024) void quality_classify_kernel(const float* __restrict__ dc, const float* __restrict__ r,
                                                                const float* __restrict__ u, int n, float thr_zr,
                                                                float thr_zu, float thr_dc, float eps,
                                                                int64_t* __restrict__ cls, float* __restrict__ pri) {
  __shared__ float s[4096];
  __shared__ float pmax_s[32];
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  const int t = threadIdx.x;
  float stat[2][2];   // [r|u][median, mad]
  for (int which = 0; which < 2; which++) {
    const float* src = which ? u : r;
    for (int i = t; i < np2; i += blockDim.x) s[i] = i < n ? src[i] : INFINITY;
    __syncthreads();
    const float m = block_median(s, n, np2);
    for (int i = t; i < np2; i += blockDim.x) s[i] = i < n ? fabsf(src[i] - m) : INFINITY;
    __syncthreads();
    const float mad = block_median(s, n, np2) + eps;
    stat[which][0] = m; stat[which][1] = mad;
  }
  // classes and unnormalised priorities
  float pm = 0.0f;   // p >= 0 everywhere, so max over the grid >= 0 (torch: p.max() of a zero-initialised tensor)
  for (int i = t; i < n; i += blockDim.x) {
    const float zr = (r[i] - stat[0][0]) / stat[0][1], zu = (u[i] - stat[1][0]) / stat[1][1];
    const float d = dc[i];
    const bool c1 = (d < thr_dc) && (zu > thr_zu);
    const bool c2 = (d >= thr_dc) && (zr > thr_zr) && (zu > thr_zu);
    const bool c3 = (zr > thr_zr) && (zu <= thr_zu);
    int c = 0;
    if (c1) c = 1;
    if (c2) c = 2;
    if (c3) c = 3;
    float p = 0.0f;
    if (c == 1) p = (1.0f - fminf(fmaxf(d, 0.0f), 1.0f)) + fmaxf(zu, 0.0f);
    else if (c == 2) p = fmaxf(zr, 0.0f) + fmaxf(zu, 0.0f);
    else if (c == 3) p = fmaxf(zr, 0.0f) + fmaxf(1.0f - u[i], 0.0f);
    cls[i] = c;
    s[i] = p;
    pm = fmaxf(pm, p);
  }
  for (int off = 32; off > 0; off >>= 1) pm = fmaxf(pm, __shfl_xor(pm, off, 64));
  if ((t & 63) == 0) pmax_s[t >> 6] = pm;
  __syncthreads();
  float gmax = 0.0f;
  for (int k = 0; k < (int)(blockDim.x >> 6); k++) gmax = fmaxf(gmax, pmax_s[k]);
  const float denom = gmax + 1e-6f;
  for (int i = t; i < n; i += blockDim.x) pri[i] = s[i] / denom;
}

}  // namespace mslam

using namespace mslam;

extern "C" int mslam_quality_reduce_grid(const float* x, const float* y, const uint8_t* valid, int h, int w, int ps,
                                         int mode, double c_thr, double q_thr, float* out, void* stream) {
  MSLAM_REQUIRE(x && out && h > 0 && w > 0 && ps > 0, "quality_reduce_grid: bad arguments");
  MSLAM_REQUIRE(ps * ps <= kQMax, "quality_reduce_grid: patch %d x %d exceeds %d pixels", ps, ps, kQMax);
  MSLAM_REQUIRE(mode >= 0 && mode <= 2 && (mode != 2 || y), "quality_reduce_grid: bad mode %d", mode);
  const int gh = h / ps, gw = w / ps;
  if (gh * gw == 0) return MSLAM_OK;
  int threads = 64;
  while (threads < ps * ps) threads <<= 1;
  hipLaunchKernelGGL(quality_reduce_kernel, dim3(gh * gw), dim3(threads), 0, (hipStream_t)stream, x, y, valid, h, w, ps, mode,
                     (float)(c_thr + 1e-8), (float)(q_thr + 1e-8), out);
  MSLAM_LAUNCH_CHECK("quality_reduce_grid");
  return MSLAM_OK;
}

extern "C" int mslam_quality_classify(const float* delta_cov, const float* r, const float* u, int n, float thr_zr,
                                      float thr_zu, float thr_dc, int64_t* cls, float* pri, void* stream) {
  MSLAM_REQUIRE(delta_cov && r && u && cls && pri, "quality_classify: null pointer");
  MSLAM_REQUIRE(n > 0 && n <= 4096, "quality_classify: grid of %d patches is outside 1..4096", n);
  hipLaunchKernelGGL(quality_classify_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, delta_cov, r, u, n, thr_zr, thr_zu,
                     thr_dc, 1e-6f, cls, pri);
  MSLAM_LAUNCH_CHECK("quality_classify");
  return MSLAM_OK;
}
