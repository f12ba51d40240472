// Matching kernels for gfx950: iterative projection (LM on a ray image), fp16 descriptor
// refinement, and the fused pre/post-processing around them.
//
// Reference behaviour being reproduced (never copied; re-derived for wave64 / HBM-first layout):
//   iter_proj_kernel       mast3r_slam/backend/src/matching_kernels.cu:119-275
//   refine_matches_kernel  mast3r_slam/backend/src/matching_kernels.cu:25-81
//   prep_for_iter_proj     mast3r_slam/matching.py:25-49 (+ mast3r_slam/image.py:5-38)
//   occlusion / lin index  mast3r_slam/matching.py:68-76, 13-15
//
// Arithmetic contract (shared with oracle/matching_ref.c, which is written independently):
//   * the TU is compiled with -ffp-contract=off; every fused multiply-add is an explicit fmaf
//   * C++ `double`-literal sub-expressions of the reference are evaluated in fp64
//   * 1.0/x on a float is the correctly rounded fp32 divide (== fp64 divide rounded once)
#include "common.h"

namespace mslam {

// ---------------------------------------------------------------------------------------------
// iter_proj
// ---------------------------------------------------------------------------------------------
struct Bilin {
  float w11, w12, w21, w22;
  const float *r11, *r12, *r21, *r22;
};

__device__ __forceinline__ Bilin bilin_setup(const float* __restrict__ img, int w, float u, float v) {
  Bilin s;
  const int u11 = (int)floorf(u);
  const int v11 = (int)floorf(v);
  const float du = u - (float)u11;
  const float dv = v - (float)v11;
  const double dud = (double)du, dvd = (double)dv;
  const double omu = 1.0 - dud, omv = 1.0 - dvd;
  s.w11 = du * dv;
  s.w12 = (float)(omu * dvd);
  s.w21 = (float)(dud * omv);
  s.w22 = (float)(omu * omv);
  const float* base = img + ((size_t)v11 * w + u11) * 9;
  s.r22 = base;               // (v11,   u11)
  s.r21 = base + 9;           // (v11,   u11+1)
  s.r12 = base + (size_t)w * 9;      // (v11+1, u11)
  s.r11 = base + (size_t)w * 9 + 9;  // (v11+1, u11+1)
  return s;
}

__device__ __forceinline__ float bilin_ch(const Bilin& s, int c) {
  float t = s.w11 * s.r11[c];
  t = fmaf(s.w12, s.r12[c], t);
  t = fmaf(s.w21, s.r21[c], t);
  t = fmaf(s.w22, s.r22[c], t);
  return t;
}

__device__ __forceinline__ float dot3(float a0, float a1, float a2, float b0, float b1, float b2) {
  float t = a0 * b0;
  t = fmaf(a1, b1, t);
  t = fmaf(a2, b2, t);
  return t;
}

__global__ __launch_bounds__(256) void iter_proj_kernel(
    const float* __restrict__ rays_img, const float* __restrict__ pts, const float* __restrict__ p_init,
    float* __restrict__ p_new, uint8_t* __restrict__ converged, int h, int w, int n, int max_iter,
    float lambda_init, float cost_thresh) {
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int i = tile * 256 + threadIdx.x;
  const int bi = blockIdx.y;
  if (i >= n) return;
  const float* img = rays_img + (size_t)bi * h * w * 9;
  const size_t o = (size_t)bi * n + i;

  const float t0 = pts[o * 3 + 0], t1 = pts[o * 3 + 1], t2 = pts[o * 3 + 2];
  const float2 pi = *reinterpret_cast<const float2*>(p_init + o * 2);
  const float umax = (float)(w - 2), vmax = (float)(h - 2);
  float u = fminf(fmaxf(pi.x, 1.0f), umax);
  float v = fminf(fmaxf(pi.y, 1.0f), vmax);

  float lambda = lambda_init;
  uint8_t conv = 0;
  // The reference samples the ray image twice per iteration: all 9 channels at (u, v), then the 3 ray channels at the
  // trial point.  (u, v) is always either the previous (u, v) or the previous trial point, so its sample is already
  // known if the trial sample takes the gradient channels along: ONE dependent gather per iteration instead of two
  // (the loop is a chain of memory round trips at 3 waves per SIMD), same operations on the same values.
  Bilin s = bilin_setup(img, w, u, v);
  float r0 = bilin_ch(s, 0), r1 = bilin_ch(s, 1), r2 = bilin_ch(s, 2);
  float gx0 = bilin_ch(s, 3), gx1 = bilin_ch(s, 4), gx2 = bilin_ch(s, 5);
  float gy0 = bilin_ch(s, 6), gy1 = bilin_ch(s, 7), gy2 = bilin_ch(s, 8);
  float rinv = 1.0f / sqrtf(dot3(r0, r1, r2, r0, r1, r2));
  r0 *= rinv; r1 *= rinv; r2 *= rinv;
  float e0 = r0 - t0, e1 = r1 - t1, e2 = r2 - t2;
  float cost = dot3(e0, e1, e2, e0, e1, e2);
  for (int it = 0; it < max_iter; it++) {
    float A00 = dot3(gx0, gx1, gx2, gx0, gx1, gx2);
    const float A01 = dot3(gx0, gx1, gx2, gy0, gy1, gy2);
    float A11 = dot3(gy0, gy1, gy2, gy0, gy1, gy2);
    const float b0 = -dot3(e0, e1, e2, gx0, gx1, gx2);
    const float b1 = -dot3(e0, e1, e2, gy0, gy1, gy2);
    A00 += lambda;
    A11 += lambda;

    const float det_inv = 1.0f / fmaf(A00, A11, -(A01 * A01));
    const float delta_u = det_inv * fmaf(A11, b0, -(A01 * b1));
    const float delta_v = det_inv * fmaf(-A01, b0, A00 * b1);

    const float u_new = fminf(fmaxf(u + delta_u, 1.0f), umax);
    const float v_new = fminf(fmaxf(v + delta_v, 1.0f), vmax);

    s = bilin_setup(img, w, u_new, v_new);
    float n0 = bilin_ch(s, 0), n1 = bilin_ch(s, 1), n2 = bilin_ch(s, 2);
    // (unconditionally, also in the last iteration: a branch here splits the 36-byte corner reads into dword loads)
    const float hx0 = bilin_ch(s, 3), hx1 = bilin_ch(s, 4), hx2 = bilin_ch(s, 5);
    const float hy0 = bilin_ch(s, 6), hy1 = bilin_ch(s, 7), hy2 = bilin_ch(s, 8);
    rinv = 1.0f / sqrtf(dot3(n0, n1, n2, n0, n1, n2));
    n0 *= rinv; n1 *= rinv; n2 *= rinv;
    const float f0 = n0 - t0, f1 = n1 - t1, f2 = n2 - t2;
    const float new_cost = dot3(f0, f1, f2, f0, f1, f2);

    if (new_cost < cost) {
      u = u_new;
      v = v_new;
      lambda = (float)((double)lambda * 0.1);
      conv = new_cost < cost_thresh;
      e0 = f0; e1 = f1; e2 = f2;
      cost = new_cost;
      gx0 = hx0; gx1 = hx1; gx2 = hx2;
      gy0 = hy0; gy1 = hy1; gy2 = hy2;
    } else {
      lambda = (float)((double)lambda * 10.0);
      conv = cost < cost_thresh;
    }
  }
  *reinterpret_cast<float2*>(p_new + o * 2) = make_float2(u, v);
  converged[o] = conv;
}

// ---------------------------------------------------------------------------------------------
// refine_matches: IEEE half accumulate, sequential over k, strict '>' arg-max, dilation 5..1
//
// What bounds it (PMC, profiles/r02_pmc_refine_matches.json, 158 us at 384x512): 44 M VALU wave-instructions (59 per
// candidate: the sequential half chain the reference's results require) keep the VALUs 46 % busy; every 16-byte gather
// goes to L2 (19.8 M L2 read requests = the whole 2.3 GB of candidate descriptors per launch, L2 hit rate 98.8 %, HBM
// 27 MB): the working set of a CU's 32 waves between two uses of a line is ~100 KB against 32 KB of L1.  Tried in round
// 2 and dropped, all bit-identical: walking the candidates row-major with a tie-break on the reference order (lines
// shared by a wave's 2r+1 row neighbours; 152-160 us, the simultaneous misses are not merged) and staging each wave's
// strip of D11 in LDS (190-211 us: after the first dilation level every lane has moved up to +-15 px on its own and the
// strip no longer fits).
// ---------------------------------------------------------------------------------------------
typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int FDIM>
__device__ __forceinline__ h16 half_dot_seq(const h16* __restrict__ a_regs, const h16* __restrict__ d11) {
  // 16-byte vector loads of the candidate descriptor, products rounded individually,
  // then a strictly sequential half add chain k = 0..FDIM-1.
  h16 score = (h16)0.0f;
  static_assert(FDIM % 8 == 0, "FDIM must be a multiple of 8");
#pragma unroll
  for (int c = 0; c < FDIM / 8; c++) {
    const h16x8 d = *reinterpret_cast<const h16x8*>(d11 + c * 8);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const h16 prod = a_regs[c * 8 + k] * d[k];
      score = score + prod;
    }
  }
  return score;
}

template <int FDIM>
__global__ __launch_bounds__(256) void refine_matches_kernel(
    const h16* __restrict__ D11, const h16* __restrict__ D21, const int64_t* __restrict__ p1,
    int64_t* __restrict__ p1_new, int h, int w, int n, int fdim_rt, int radius, int dilation_max) {
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int i = tile * 256 + threadIdx.x;
  const int bi = blockIdx.y;
  if (i >= n) return;
  const int fdim = FDIM > 0 ? FDIM : fdim_rt;
  const h16* img = D11 + (size_t)bi * h * w * fdim;
  const size_t o = (size_t)bi * n + i;
  const h16* d21 = D21 + o * fdim;

  h16 a[FDIM > 0 ? FDIM : 1];
  if constexpr (FDIM > 0) {
#pragma unroll
    for (int c = 0; c < FDIM / 8; c++) {
      const h16x8 t = *reinterpret_cast<const h16x8*>(d21 + c * 8);
#pragma unroll
      for (int k = 0; k < 8; k++) a[c * 8 + k] = t[k];
    }
  }

  const longlong2 pp = *reinterpret_cast<const longlong2*>(p1 + o * 2);
  long long u0 = pp.x, v0 = pp.y;
  long long u_new = u0, v_new = v0;
  float max_score = 6.103515625e-05f;  // numeric_limits<half>::min() = 2^-14

  for (int d = dilation_max; d > 0; d--) {
    const int rd = radius * d;
    const int diam = 2 * rd + 1;
    for (int ii = 0; ii < diam; ii += d) {
      const long long u = u0 - rd + ii;
      const int ui = (int)u;
      if (ui < 0 || ui >= w) continue;
      for (int jj = 0; jj < diam; jj += d) {
        const long long v = v0 - rd + jj;
        const int vi = (int)v;
        if (vi < 0 || vi >= h) continue;
        // in range here, so 32-bit element arithmetic (the image is < 2^31 halves); 64-bit products cost 4x the VALU
        const h16* d11 = img + ((unsigned)vi * (unsigned)w + (unsigned)ui) * (unsigned)fdim;
        h16 score;
        if constexpr (FDIM > 0) {
          score = half_dot_seq<FDIM>(a, d11);
        } else {
          score = (h16)0.0f;
          for (int k = 0; k < fdim; k++) {
            const h16 prod = d21[k] * d11[k];
            score = score + prod;
          }
        }
        const float sf = (float)score;
        if (sf > max_score) {
          max_score = sf;
          u_new = u;
          v_new = v;
        }
      }
    }
    u0 = u_new;
    v0 = v_new;
  }
  longlong2 out;
  out.x = u_new;
  out.y = v_new;
  *reinterpret_cast<longlong2*>(p1_new + o * 2) = out;
}

// ---------------------------------------------------------------------------------------------
// prep_for_iter_proj fused: normalise rays, 3x3 gradients (reflect pad), concat; normalise X21;
// p_init from a linear index.  One 16x16 output tile per block, 18x18 halo tile in LDS.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int n) {
  // torch 'reflect' padding of width 1: -1 -> 1, n -> n-2
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}

__device__ __forceinline__ void normalize3(float x, float y, float z, float& ox, float& oy, float& oz) {
  // F.normalize: v / max(||v||_2, 1e-12)
  const float nrm = fmaxf(sqrtf(fmaf(z, z, fmaf(y, y, x * x))), 1e-12f);
  ox = x / nrm;
  oy = y / nrm;
  oz = z / nrm;
}

constexpr int kPrepTile = 16;

__global__ __launch_bounds__(256) void prep_iter_proj_kernel(
    const float* __restrict__ X11, const float* __restrict__ X21, const int64_t* __restrict__ idx_init,
    float* __restrict__ rays_out, float* __restrict__ pts_out, float* __restrict__ p_init, int h, int w) {
  __shared__ float tile[kPrepTile + 2][kPrepTile + 2][3];
  const int bi = blockIdx.z;
  const int x0 = blockIdx.x * kPrepTile, y0 = blockIdx.y * kPrepTile;
  const float* X = X11 + (size_t)bi * h * w * 3;

  for (int t = threadIdx.x; t < (kPrepTile + 2) * (kPrepTile + 2); t += 256) {
    const int ty = t / (kPrepTile + 2), tx = t % (kPrepTile + 2);
    const int yy = reflect_idx(min(y0 + ty - 1, h), h);
    const int xx = reflect_idx(min(x0 + tx - 1, w), w);
    const float* p = X + ((size_t)min(yy, h - 1) * w + min(xx, w - 1)) * 3;
    float a, b, c;
    normalize3(p[0], p[1], p[2], a, b, c);
    tile[ty][tx][0] = a;
    tile[ty][tx][1] = b;
    tile[ty][tx][2] = c;
  }
  __syncthreads();

  const int lx = threadIdx.x % kPrepTile, ly = threadIdx.x / kPrepTile;
  const int x = x0 + lx, y = y0 + ly;
  if (x >= w || y >= h) return;
  const size_t pix = (size_t)y * w + x;
  const size_t o = (size_t)bi * h * w + pix;
  float* out = rays_out + o * 9;
  const float k3 = 3.0f / 32.0f, k10 = 10.0f / 32.0f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float a00 = tile[ly][lx][c], a01 = tile[ly][lx + 1][c], a02 = tile[ly][lx + 2][c];
    const float a10 = tile[ly + 1][lx][c], a11 = tile[ly + 1][lx + 1][c], a12 = tile[ly + 1][lx + 2][c];
    const float a20 = tile[ly + 2][lx][c], a21 = tile[ly + 2][lx + 1][c], a22 = tile[ly + 2][lx + 2][c];
    // cross-correlation, taps visited row-major (zero taps skipped)
    float gx = -k3 * a00;
    gx = fmaf(k3, a02, gx);
    gx = fmaf(-k10, a10, gx);
    gx = fmaf(k10, a12, gx);
    gx = fmaf(-k3, a20, gx);
    gx = fmaf(k3, a22, gx);
    float gy = -k3 * a00;
    gy = fmaf(-k10, a01, gy);
    gy = fmaf(-k3, a02, gy);
    gy = fmaf(k3, a20, gy);
    gy = fmaf(k10, a21, gy);
    gy = fmaf(k3, a22, gy);
    out[c] = a11;
    out[3 + c] = gx;
    out[6 + c] = gy;
  }
  const float* q = X21 + o * 3;
  float a, b, c;
  normalize3(q[0], q[1], q[2], a, b, c);
  pts_out[o * 3 + 0] = a;
  pts_out[o * 3 + 1] = b;
  pts_out[o * 3 + 2] = c;
  const long long lin = idx_init ? idx_init[o] : (long long)pix;
  p_init[o * 2 + 0] = (float)(lin % w);
  p_init[o * 2 + 1] = (float)(lin / w);
}

// p1 = trunc(p); valid &= ||X11[p1] - X21|| < dist_thresh
__global__ __launch_bounds__(256) void match_occlusion_kernel(
    const float* __restrict__ X11, const float* __restrict__ X21, const float* __restrict__ p,
    int64_t* __restrict__ p1, uint8_t* __restrict__ valid, int h, int w, float dist_thresh) {
  const int n = h * w;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int bi = blockIdx.y;
  if (i >= n) return;
  const size_t o = (size_t)bi * n + i;
  const float2 pp = *reinterpret_cast<const float2*>(p + o * 2);
  const long long u = (long long)pp.x, v = (long long)pp.y;  // .long(): truncation toward zero
  const float* a = X11 + ((size_t)bi * n + (size_t)v * w + (size_t)u) * 3;
  const float* q = X21 + o * 3;
  const float d0 = a[0] - q[0], d1 = a[1] - q[1], d2 = a[2] - q[2];
  const float dist = sqrtf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
  longlong2 out;
  out.x = u;
  out.y = v;
  *reinterpret_cast<longlong2*>(p1 + o * 2) = out;
  valid[o] = (valid[o] != 0) && (dist < dist_thresh);
}

__global__ __launch_bounds__(256) void pixel_to_lin_kernel(const int64_t* __restrict__ p1,
                                                           int64_t* __restrict__ idx, size_t total, int w) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const longlong2 pp = *reinterpret_cast<const longlong2*>(p1 + i * 2);
  idx[i] = pp.x + (long long)w * pp.y;
}

}  // namespace mslam

using namespace mslam;

extern "C" int mslam_iter_proj(const float* rays_img_with_grad, const float* pts_3d_norm,
                               const float* p_init, float* p_new, uint8_t* converged, int b, int h,
                               int w, int n, int max_iter, float lambda_init, float cost_thresh,
                               void* stream) {
  MSLAM_REQUIRE(b >= 0 && n >= 0 && max_iter >= 0, "iter_proj: negative size");
  if (b == 0 || n == 0) return MSLAM_OK;
  MSLAM_REQUIRE(rays_img_with_grad && pts_3d_norm && p_init && p_new && converged,
                "iter_proj: null pointer");
  MSLAM_REQUIRE(h >= 3 && w >= 3, "iter_proj: ray image must be at least 3x3 (got %dx%d)", h, w);
  MSLAM_REQUIRE(b <= 65535, "iter_proj: batch %d exceeds grid limit", b);
  dim3 grid((n + 255) / 256, b);
  hipLaunchKernelGGL(iter_proj_kernel, grid, dim3(256), 0, (hipStream_t)stream, rays_img_with_grad,
                     pts_3d_norm, p_init, p_new, converged, h, w, n, max_iter, lambda_init, cost_thresh);
  MSLAM_LAUNCH_CHECK("iter_proj");
  return MSLAM_OK;
}

extern "C" int mslam_refine_matches(const uint16_t* D11, const uint16_t* D21, const int64_t* p1,
                                    int64_t* p1_new, int b, int h, int w, int n, int fdim, int radius,
                                    int dilation_max, void* stream) {
  MSLAM_REQUIRE(b >= 0 && n >= 0 && fdim >= 0, "refine_matches: negative size");
  if (b == 0 || n == 0) return MSLAM_OK;
  MSLAM_REQUIRE(D11 && D21 && p1 && p1_new, "refine_matches: null pointer");
  MSLAM_REQUIRE(b <= 65535, "refine_matches: batch %d exceeds grid limit", b);
  dim3 grid((n + 255) / 256, b);
  const h16* d11 = reinterpret_cast<const h16*>(D11);
  const h16* d21 = reinterpret_cast<const h16*>(D21);
  if (fdim == 24) {
    hipLaunchKernelGGL(refine_matches_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, d11, d21, p1,
                       p1_new, h, w, n, fdim, radius, dilation_max);
  } else if (fdim == 16) {
    hipLaunchKernelGGL(refine_matches_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, d11, d21, p1,
                       p1_new, h, w, n, fdim, radius, dilation_max);
  } else {
    hipLaunchKernelGGL(refine_matches_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, d11, d21, p1,
                       p1_new, h, w, n, fdim, radius, dilation_max);
  }
  MSLAM_LAUNCH_CHECK("refine_matches");
  return MSLAM_OK;
}

extern "C" int mslam_prep_iter_proj(const float* X11, const float* X21, const int64_t* idx_init,
                                    float* rays_img_with_grad, float* pts_3d_norm, float* p_init, int b,
                                    int h, int w, void* stream) {
  MSLAM_REQUIRE(b >= 0 && h >= 0 && w >= 0, "prep_iter_proj: negative size");
  if (b == 0 || h == 0 || w == 0) return MSLAM_OK;
  MSLAM_REQUIRE(X11 && X21 && rays_img_with_grad && pts_3d_norm && p_init, "prep_iter_proj: null pointer");
  MSLAM_REQUIRE(h >= 2 && w >= 2, "prep_iter_proj: reflect padding needs h,w >= 2");
  MSLAM_REQUIRE(b <= 65535, "prep_iter_proj: batch %d exceeds grid limit", b);
  dim3 grid((w + kPrepTile - 1) / kPrepTile, (h + kPrepTile - 1) / kPrepTile, b);
  hipLaunchKernelGGL(prep_iter_proj_kernel, grid, dim3(256), 0, (hipStream_t)stream, X11, X21, idx_init,
                     rays_img_with_grad, pts_3d_norm, p_init, h, w);
  MSLAM_LAUNCH_CHECK("prep_iter_proj");
  return MSLAM_OK;
}

extern "C" int mslam_match_occlusion(const float* X11, const float* X21, const float* p, int64_t* p1,
                                     uint8_t* valid, int b, int h, int w, float dist_thresh,
                                     void* stream) {
  MSLAM_REQUIRE(b >= 0 && h >= 0 && w >= 0, "match_occlusion: negative size");
  if (b == 0 || h == 0 || w == 0) return MSLAM_OK;
  MSLAM_REQUIRE(X11 && X21 && p && p1 && valid, "match_occlusion: null pointer");
  dim3 grid((h * w + 255) / 256, b);
  hipLaunchKernelGGL(match_occlusion_kernel, grid, dim3(256), 0, (hipStream_t)stream, X11, X21, p, p1,
                     valid, h, w, dist_thresh);
  MSLAM_LAUNCH_CHECK("match_occlusion");
  return MSLAM_OK;
}

extern "C" int mslam_pixel_to_lin(const int64_t* p1, int64_t* idx, int b, int n, int w, void* stream) {
  MSLAM_REQUIRE(b >= 0 && n >= 0, "pixel_to_lin: negative size");
  const size_t total = (size_t)b * n;
  if (total == 0) return MSLAM_OK;
  MSLAM_REQUIRE(p1 && idx, "pixel_to_lin: null pointer");
  hipLaunchKernelGGL(pixel_to_lin_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, p1, idx, total, w);
  MSLAM_LAUNCH_CHECK("pixel_to_lin");
  return MSLAM_OK;
}
