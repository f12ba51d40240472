/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Scalar CPU restatement of the reference's two matching kernels:
 *   iter_proj_kernel       /root/reference/mast3r_slam/backend/src/matching_kernels.cu:119-275
 *   refine_matches_kernel  /root/reference/mast3r_slam/backend/src/matching_kernels.cu:25-81
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Parity status: the CUDA kernels cannot be compiled or run in the build container (no nvcc,
 * no NVIDIA GPU) and the reference ships no test vectors for them, so this restatement is
 * "parity unpinned" against the CUDA binary.  It follows the source line by line with these
 * stated conventions for what the CUDA compiler leaves open:
 *   - `a*b + c*d + ...` chains are evaluated left to right with the products after the first
 *     contracted into fmaf (nvcc -fmad=true default): t=a*b; t=fmaf(c,d,t); ...
 *   - expressions that mix a `double` literal with floats (1.0-du, 1.0/x, lambda*=0.1) are
 *     evaluated in double and rounded to float once, as C++ promotion rules require
 *     (matching_kernels.cu:162-164,187,213,262,266).
 *   - refine_matches accumulates in IEEE half with separate round-to-nearest-even mul and add
 *     (`score += a*b` on __half operators, matching_kernels.cu:60-63); max_score starts at
 *     numeric_limits<half>::min() = 2^-14 (matching_kernels.cu:47).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#if defined(__F16C__)
#include <immintrin.h>
#endif

/* ---------- IEEE binary16 <-> binary32, software, round-to-nearest-even ---------- */
static inline float half_to_float_sw(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t f;
  if (exp == 0) {
    if (man == 0) {
      f = sign;
    } else { /* subnormal */
      int e = -1;
      do { e++; man <<= 1; } while ((man & 0x400u) == 0);
      man &= 0x3FFu;
      f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    f = sign | 0x7F800000u | (man << 13);
  } else {
    f = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float out;
  memcpy(&out, &f, 4);
  return out;
}

static inline uint16_t float_to_half_sw(float x) {
  uint32_t f;
  memcpy(&f, &x, 4);
  uint32_t sign = (f >> 16) & 0x8000u;
  uint32_t fexp = (f >> 23) & 0xFFu;
  uint32_t man = f & 0x7FFFFFu;
  if (fexp == 255) { /* inf / nan */
    return (uint16_t)(sign | 0x7C00u | (man ? 0x200u | (man >> 13) : 0));
  }
  int e = (int)fexp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7C00u); /* overflow -> inf */
  if (e <= 0) {                                  /* subnormal or zero */
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    int shift = 14 - e; /* 14..24 */
    uint32_t hm = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = man >> 13;
  uint32_t rem = man & 0x1FFFu;
  uint16_t out = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) out++; /* carries into exponent correctly */
  return out;
}

#if defined(__F16C__)
/* F16C conversions are IEEE round-to-nearest-even, bit-identical to the software routines above
 * (tests/test_matching_oracle.py checks the software path exhaustively against numpy). */
static inline float half_to_float(uint16_t h) { return _cvtsh_ss(h); }
static inline uint16_t float_to_half(float x) { return _cvtss_sh(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
#else
static inline float half_to_float(uint16_t h) { return half_to_float_sw(h); }
static inline uint16_t float_to_half(float x) { return float_to_half_sw(x); }
#endif
uint16_t oracle_float_to_half_sw(float x) { return float_to_half_sw(x); }
float oracle_half_to_float_sw(uint16_t h) { return half_to_float_sw(h); }
uint16_t oracle_float_to_half(float x) { return float_to_half(x); }

/* half*half is exact in float (22-bit product), half+half in float then RNE to half is
 * correctly rounded (24 >= 2*11+2), so float arithmetic + one rounding == IEEE half ops. */
static inline uint16_t hmul(uint16_t a, uint16_t b) {
  return float_to_half(half_to_float(a) * half_to_float(b));
}
static inline uint16_t hadd(uint16_t a, uint16_t b) {
  return float_to_half(half_to_float(a) + half_to_float(b));
}

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* bilinear fetch of channels [c0,c0+3) ; matching_kernels.cu:155-183 */
static inline void bilinear3(const float* img, int w, float u, float v, int c0, float* out) {
  int u11 = (int)floorf(u);
  int v11 = (int)floorf(v);
  float du = u - (float)u11;
  float dv = v - (float)v11;
  float w11 = du * dv;
  float w12 = (float)((1.0 - (double)du) * (double)dv);
  float w21 = (float)((double)du * (1.0 - (double)dv));
  float w22 = (float)((1.0 - (double)du) * (1.0 - (double)dv));
  const float* r11 = img + ((size_t)(v11 + 1) * w + (u11 + 1)) * 9;
  const float* r12 = img + ((size_t)(v11 + 1) * w + u11) * 9;
  const float* r21 = img + ((size_t)v11 * w + (u11 + 1)) * 9;
  const float* r22 = img + ((size_t)v11 * w + u11) * 9;
  for (int j = 0; j < 3; j++) {
    float t = w11 * r11[c0 + j];
    t = fmaf(w12, r12[c0 + j], t);
    t = fmaf(w21, r21[c0 + j], t);
    t = fmaf(w22, r22[c0 + j], t);
    out[j] = t;
  }
}

static inline float dot3f(const float* a, const float* b) {
  float t = a[0] * b[0];
  t = fmaf(a[1], b[1], t);
  t = fmaf(a[2], b[2], t);
  return t;
}

/* matching_kernels.cu:119-275.  rays_img (b,h,w,9) f32, pts (b,n,3), p_init (b,n,2)
 * -> p_new (b,n,2) f32, converged (b,n) u8. */
void oracle_iter_proj(const float* rays_img, const float* pts_3d_norm, const float* p_init,
                      float* p_new, uint8_t* converged, int b, int h, int w, int n, int max_iter,
                      float lambda_init, float cost_thresh) {
  for (int bi = 0; bi < b; bi++) {
    const float* img = rays_img + (size_t)bi * h * w * 9;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
      size_t o = (size_t)bi * n + i;
      const float* tgt = pts_3d_norm + o * 3;
      float u = p_init[o * 2 + 0];
      float v = p_init[o * 2 + 1];
      u = clampf(u, 1.0f, (float)(w - 2));
      v = clampf(v, 1.0f, (float)(h - 2));
      float r[3], gx[3], gy[3], err[3];
      float lambda = lambda_init;
      uint8_t conv = 0; /* torch::zeros init, matching_kernels.cu:300 */
      for (int it = 0; it < max_iter; it++) {
        bilinear3(img, w, u, v, 0, r);
        bilinear3(img, w, u, v, 3, gx);
        bilinear3(img, w, u, v, 6, gy);
        float r_norm = sqrtf(dot3f(r, r));
        float r_norm_inv = (float)(1.0 / (double)r_norm);
        for (int j = 0; j < 3; j++) r[j] *= r_norm_inv;
        for (int j = 0; j < 3; j++) err[j] = r[j] - tgt[j];
        float cost = dot3f(err, err);

        float A00 = dot3f(gx, gx);
        float A01 = dot3f(gx, gy);
        float A11 = dot3f(gy, gy);
        float b0 = -dot3f(err, gx);
        float b1 = -dot3f(err, gy);
        A00 += lambda;
        A11 += lambda;

        float det = fmaf(A00, A11, -(A01 * A01));
        float det_inv = (float)(1.0 / (double)det);
        float delta_u = det_inv * fmaf(A11, b0, -(A01 * b1));
        float delta_v = det_inv * fmaf(-A01, b0, A00 * b1);

        float u_new = clampf(u + delta_u, 1.0f, (float)(w - 2));
        float v_new = clampf(v + delta_v, 1.0f, (float)(h - 2));

        bilinear3(img, w, u_new, v_new, 0, r);
        r_norm = sqrtf(dot3f(r, r));
        r_norm_inv = (float)(1.0 / (double)r_norm);
        for (int j = 0; j < 3; j++) r[j] *= r_norm_inv;
        for (int j = 0; j < 3; j++) err[j] = r[j] - tgt[j];
        float new_cost = dot3f(err, err);

        if (new_cost < cost) {
          u = u_new;
          v = v_new;
          lambda = (float)((double)lambda * 0.1);
          conv = new_cost < cost_thresh;
        } else {
          lambda = (float)((double)lambda * 10.0);
          conv = cost < cost_thresh;
        }
      }
      p_new[o * 2 + 0] = u;
      p_new[o * 2 + 1] = v;
      converged[o] = conv;
    }
  }
}

/* matching_kernels.cu:25-81.  D11 (b,h,w,f) half bits, D21 (b,n,f) half bits, p1 (b,n,2) i64
 * -> p1_new (b,n,2) i64.  fused_fma != 0 selects the alternative contraction
 * (score = fma_half(a,b,score), single rounding) that ptxas is allowed to pick. */
void oracle_refine_matches(const uint16_t* D11, const uint16_t* D21, const int64_t* p1,
                           int64_t* p1_new, int b, int h, int w, int n, int fdim, int radius,
                           int dilation_max, int fused_fma) {
  for (int bi = 0; bi < b; bi++) {
    const uint16_t* img = D11 + (size_t)bi * h * w * fdim;
#pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
      size_t o = (size_t)bi * n + i;
      const uint16_t* d21 = D21 + o * fdim;
      int64_t u0 = p1[o * 2 + 0];
      int64_t v0 = p1[o * 2 + 1];
      uint16_t max_score = 0x0400; /* numeric_limits<half>::min() = 2^-14 */
      int64_t u_new = u0, v_new = v0;
      for (int d = dilation_max; d > 0; d--) {
        const int rd = radius * d;
        const int diam = 2 * rd + 1;
        for (int ii = 0; ii < diam; ii += d) {
          for (int jj = 0; jj < diam; jj += d) {
            const int64_t u = u0 - rd + ii;
            const int64_t v = v0 - rd + jj;
            /* inside_image takes int arguments (matching_kernels.cu:17): long -> int */
            const int ui = (int)u, vi = (int)v;
            if (vi >= 0 && vi < h && ui >= 0 && ui < w) {
              const uint16_t* d11 = img + ((size_t)v * w + (size_t)u) * fdim;
              uint16_t score = 0;
              if (!fused_fma) {
                for (int k = 0; k < fdim; k++) score = hadd(score, hmul(d21[k], d11[k]));
              } else {
                for (int k = 0; k < fdim; k++) {
                  /* exact product + half sum fits double exactly; one rounding to half */
                  double acc = (double)half_to_float(d21[k]) * (double)half_to_float(d11[k]) +
                               (double)half_to_float(score);
                  /* double -> half with a single rounding: go through float only if exact */
                  float af = (float)acc;
                  if ((double)af != acc) {
                    /* break the double-rounding tie by nudging toward the true value */
                    uint32_t bits;
                    memcpy(&bits, &af, 4);
                    if ((bits & 0x1FFFu) == 0x1000u) {
                      bits += ((double)af < acc) == (af > 0) ? 1u : (uint32_t)-1;
                      memcpy(&af, &bits, 4);
                    }
                  }
                  score = float_to_half(af);
                }
              }
              if (half_to_float(score) > half_to_float(max_score)) {
                max_score = score;
                u_new = u;
                v_new = v;
              }
            }
          }
        }
        u0 = u_new;
        v0 = v_new;
      }
      p1_new[o * 2 + 0] = u_new;
      p1_new[o * 2 + 1] = v_new;
    }
  }
}
