/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Scalar CPU restatement of the reference's Gauss-Newton backend:
 *   Sim3 helpers        /root/reference/mast3r_slam/backend/src/gn_kernels.cu:172-413
 *   pose_retr_kernel    gn_kernels.cu:415-453
 *   point_align_kernel  gn_kernels.cu:455-723
 *   ray_align_kernel    gn_kernels.cu:813-1138
 *   calib_proj_kernel   gn_kernels.cu:1231-1543
 *   blockReduce         gn_kernels.cu:36-55   (256-thread tree, emulated in the same order)
 *   host loop           gn_kernels.cu:1140-1228 (+ :725-811, :1546-1638), SparseBlock :57-159,
 *                       get_unique_kf_idx / create_inds :161-170
 *
 * Parity status: "parity unpinned" against the CUDA binary (cannot be built here: no nvcc, no
 * Eigen).  Independent pins: the residual/Jacobian model is cross-checked in tests against the
 * reference's pure-torch tracker formulae (tracker.py:225-318 + geometry.py) and the solve against
 * scipy's Cholesky.  Conventions: each virtual thread t of a 256-thread block accumulates points
 * k = t, t+256, ... in order; the block reduction follows the reference tree exactly; no FMA
 * contraction (the TU is built with -ffp-contract=off); `double` literals in float expressions
 * promote as C++ requires.  The fp64 solve is a dense Cholesky (Eigen::SimplicialLLT is a sparse
 * LL^T of the same matrix; the two differ only by fp64 rounding/ordering).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define THREADS 256
#define EPSF 1e-6

static inline float huber_w(float r) { /* gn_kernels.cu:172-175 */
  const float r_abs = fabsf(r);
  return (double)r_abs < 1.345 ? 1.0f : (float)(1.345 / (double)r_abs);
}

static void quat_comp(const float* qi, const float* qj, float* out) {
  out[0] = qi[3] * qj[0] + qi[0] * qj[3] + qi[1] * qj[2] - qi[2] * qj[1];
  out[1] = qi[3] * qj[1] - qi[0] * qj[2] + qi[1] * qj[3] + qi[2] * qj[0];
  out[2] = qi[3] * qj[2] + qi[0] * qj[1] - qi[1] * qj[0] + qi[2] * qj[3];
  out[3] = qi[3] * qj[3] - qi[0] * qj[0] - qi[1] * qj[1] - qi[2] * qj[2];
}

static void actSO3(const float* q, const float* X, float* Y) {
  float uv[3];
  uv[0] = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  uv[1] = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  uv[2] = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  float y0 = X[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  float y1 = X[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  float y2 = X[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
  Y[0] = y0; Y[1] = y1; Y[2] = y2; /* temporaries: the reference calls this with Y aliasing X (:270) */
}

static void actSim3(const float* t, const float* q, const float* s, const float* X, float* Y) {
  actSO3(q, X, Y);
  Y[0] *= s[0]; Y[1] *= s[0]; Y[2] *= s[0];
  Y[0] += t[0]; Y[1] += t[1]; Y[2] += t[2];
}

static void relSim3(const float* ti, const float* qi, const float* si, const float* tj,
                    const float* qj, const float* sj, float* tij, float* qij, float* sij) {
  float si_inv = 1.0f / si[0];
  sij[0] = si_inv * sj[0];
  float qi_inv[4] = {-qi[0], -qi[1], -qi[2], qi[3]};
  quat_comp(qi_inv, qj, qij);
  tij[0] = tj[0] - ti[0];
  tij[1] = tj[1] - ti[1];
  tij[2] = tj[2] - ti[2];
  /* NOTE gn_kernels.cu:270 calls actSO3(qi_inv, tij, tij) IN PLACE: actSO3 writes Y[0] before it
   * reads X[0] for Y[1]... the reference computes uv[] from X first and each Y[k] reads only X[k]
   * and uv[], so aliasing is harmless there; kept harmless here via temporaries. */
  actSO3(qi_inv, tij, tij);
  tij[0] *= si_inv; tij[1] *= si_inv; tij[2] *= si_inv;
}

static inline float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static void apply_Sim3_adj_inv(const float* t, const float* q, const float* s, const float* X, float* Y) {
  const float s_inv = 1.0f / s[0];
  float Ra[3];
  actSO3(q, &X[0], Ra);
  Y[0] = s_inv * Ra[0];
  Y[1] = s_inv * Ra[1];
  Y[2] = s_inv * Ra[2];
  actSO3(q, &X[3], &Y[3]);
  Y[3] += s_inv * (t[1] * Ra[2] - t[2] * Ra[1]);
  Y[4] += s_inv * (t[2] * Ra[0] - t[0] * Ra[2]);
  Y[5] += s_inv * (t[0] * Ra[1] - t[1] * Ra[0]);
  Y[6] = X[6] + (s_inv * dot3(t, Ra));
}

static void expSO3(const float* phi, float* q) {
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float imag, real;
  if ((double)theta_sq < EPSF) {
    float theta_p4 = theta_sq * theta_sq;
    imag = (float)(0.5 - (1.0 / 48.0) * (double)theta_sq + (1.0 / 3840.0) * (double)theta_p4);
    real = (float)(1.0 - (1.0 / 8.0) * (double)theta_sq + (1.0 / 384.0) * (double)theta_p4);
  } else {
    float theta = sqrtf(theta_sq);
    imag = sinf((float)(0.5 * (double)theta)) / theta;
    real = cosf((float)(0.5 * (double)theta));
  }
  q[0] = imag * phi[0];
  q[1] = imag * phi[1];
  q[2] = imag * phi[2];
  q[3] = real;
}

static void crossInplace(const float* a, float* b) {
  float x[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
  b[0] = x[0]; b[1] = x[1]; b[2] = x[2];
}

static void expSim3(const float* xi, float* t, float* q, float* s) { /* gn_kernels.cu:323-390 */
  float tau[3] = {xi[0], xi[1], xi[2]};
  float phi[3] = {xi[3], xi[4], xi[5]};
  float sigma = xi[6];
  float scale = expf(sigma);
  expSO3(phi, q);
  s[0] = scale;
  float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float theta = sqrtf(theta_sq);
  float A, B, C;
  const float one = 1.0f, half = 0.5f;
  if ((double)fabsf(sigma) < EPSF) {
    C = one;
    if ((double)fabsf(theta) < EPSF) {
      A = half;
      B = (float)(1.0 / 6.0);
    } else {
      A = (one - cosf(theta)) / theta_sq;
      B = (theta - sinf(theta)) / (theta_sq * theta);
    }
  } else {
    C = (scale - one) / sigma;
    if ((double)fabsf(theta) < EPSF) {
      float sigma_sq = sigma * sigma;
      A = ((sigma - one) * scale + one) / sigma_sq;
      B = (scale * half * sigma_sq + scale - one - sigma * scale) / (sigma_sq * sigma);
    } else {
      float a = scale * sinf(theta);
      float b = scale * cosf(theta);
      float c = theta_sq + sigma * sigma;
      A = (a * sigma + (one - b) * theta) / (theta * c);
      B = (C - ((b - one) * sigma + a * theta) / (c)) / (theta_sq);
    }
  }
  t[0] = C * tau[0]; t[1] = C * tau[1]; t[2] = C * tau[2];
  crossInplace(phi, tau);
  t[0] += A * tau[0]; t[1] += A * tau[1]; t[2] += A * tau[2];
  crossInplace(phi, tau);
  t[0] += B * tau[0]; t[1] += B * tau[1]; t[2] += B * tau[2];
}

static void retrSim3(const float* xi, const float* t, const float* q, const float* s, float* t1,
                     float* q1, float* s1) {
  float dt[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1}, ds[1] = {0};
  expSim3(xi, dt, dq, ds);
  quat_comp(dq, q, q1);
  actSO3(dq, t, t1);
  t1[0] *= ds[0]; t1[1] *= ds[0]; t1[2] *= ds[0];
  t1[0] += dt[0]; t1[1] += dt[1]; t1[2] += dt[2];
  s1[0] = ds[0] * s[0];
}

/* exported single-op entry points so tests can pin the Sim3 algebra on its own */
void oracle_sim3_exp(const float* xi, float* out8) { expSim3(xi, out8, out8 + 3, out8 + 7); }
void oracle_sim3_retr(const float* xi, const float* T, float* out8) {
  retrSim3(xi, T, T + 3, T + 7, out8, out8 + 3, out8 + 7);
}
void oracle_sim3_rel(const float* Ti, const float* Tj, float* out8) {
  relSim3(Ti, Ti + 3, Ti + 7, Tj, Tj + 3, Tj + 7, out8, out8 + 3, out8 + 7);
}
void oracle_sim3_act(const float* T, const float* X, float* Y, int n) {
  for (int i = 0; i < n; i++) actSim3(T, T + 3, T + 7, X + 3 * i, Y + 3 * i);
}
void oracle_sim3_adj_inv(const float* T, const float* X7, float* Y7) {
  apply_Sim3_adj_inv(T, T + 3, T + 7, X7, Y7);
}

void oracle_pose_retr(float* poses, const float* dx, int num_poses, int num_fix) { /* :415-453 */
  for (int k = num_fix; k < num_poses; k++) {
    float t1[3], q1[4], s1[1];
    float* P = poses + 8 * k;
    retrSim3(dx + 7 * (k - num_fix), P, P + 3, P + 7, t1, q1, s1);
    P[0] = t1[0]; P[1] = t1[1]; P[2] = t1[2];
    P[3] = q1[0]; P[4] = q1[1]; P[5] = q1[2]; P[6] = q1[3];
    P[7] = s1[0];
  }
}

/* blockReduce (gn_kernels.cu:36-55): returns sdata[0] after the 256 -> 1 tree. */
static float block_reduce(float* sdata) {
  for (int t = 0; t < 128; t++) sdata[t] += sdata[t + 128];
  for (int t = 0; t < 64; t++) sdata[t] += sdata[t + 64];
  /* warpReduce: 32 lanes in lockstep, each step reads before any lane writes */
  static const int steps[6] = {32, 16, 8, 4, 2, 1};
  for (int s = 0; s < 6; s++) {
    float tmp[32];
    for (int t = 0; t < 32; t++) tmp[t] = sdata[t] + sdata[t + steps[s]];
    for (int t = 0; t < 32; t++) sdata[t] = tmp[t];
  }
  return sdata[0];
}

typedef struct {
  int kind; /* 0 = rays, 1 = calib, 2 = points */
  float sigma_a, sigma_b; /* rays: ray,dist ; calib: pixel,depth ; points: point,- */
  float C_thresh, Q_thresh;
  int height, width, pixel_border;
  float z_eps;
  float fx, fy, cx, cy;
} gn_params;

#define H_DIM 105

/* accumulate one residual row: gn_kernels.cu:1002-1013 pattern */
static inline void accum_row(float* hij, float* vi, float* vj, float* Jx, const float* ti,
                             const float* qi, const float* si, float w, float err) {
  float* Ji = Jx;
  float* Jj = Jx + 7;
  apply_Sim3_adj_inv(ti, qi, si, Ji, Jj);
  for (int n = 0; n < 7; n++) Ji[n] = -Jj[n];
  int l = 0;
  for (int n = 0; n < 14; n++)
    for (int m = 0; m <= n; m++) {
      hij[l] += w * Jx[n] * Jx[m];
      l++;
    }
  for (int n = 0; n < 7; n++) {
    vi[n] += w * err * Ji[n];
    vj[n] += w * err * Jj[n];
  }
}

static void edge_kernel(const gn_params* P, const float* Twc, const float* Xs, const float* Cs,
                        int64_t ix, int64_t jx, const int64_t* idx, const uint8_t* valid_match,
                        const float* Q, int num_points, float* Hs /*4 x [7][7] strided*/,
                        size_t h_stride, float* gs, size_t g_stride) {
  float ti[3], tj[3], tij[3], qi[4], qj[4], qij[4], si[1], sj[1], sij[1];
  for (int k = 0; k < 3; k++) { ti[k] = Twc[ix * 8 + k]; tj[k] = Twc[jx * 8 + k]; }
  for (int k = 0; k < 4; k++) { qi[k] = Twc[ix * 8 + 3 + k]; qj[k] = Twc[jx * 8 + 3 + k]; }
  si[0] = Twc[ix * 8 + 7];
  sj[0] = Twc[jx * 8 + 7];
  relSim3(ti, qi, si, tj, qj, sj, tij, qij, sij);

  float* hij_all = (float*)calloc((size_t)THREADS * H_DIM, sizeof(float));
  float* vi_all = (float*)calloc((size_t)THREADS * 7, sizeof(float));
  float* vj_all = (float*)calloc((size_t)THREADS * 7, sizeof(float));
  const float* Xi_base = Xs + (size_t)ix * num_points * 3;
  const float* Xj_base = Xs + (size_t)jx * num_points * 3;
  const float* Ci_base = Cs + (size_t)ix * num_points;
  const float* Cj_base = Cs + (size_t)jx * num_points;
  const float sa_inv = 1.0f / P->sigma_a;
  const float sb_inv = P->kind == 2 ? 0.0f : 1.0f / P->sigma_b;

  for (int tid = 0; tid < THREADS; tid++) {
    float* hij = hij_all + (size_t)tid * H_DIM;
    float* vi = vi_all + tid * 7;
    float* vj = vj_all + tid * 7;
    float Jx[14];
    float* Ji = Jx;
    for (int k = tid; k < num_points; k += THREADS) {
      const int vm = valid_match[k] != 0;
      const int64_t ind_Xi = vm ? idx[k] : 0;
      const float* Xi = Xi_base + ind_Xi * 3;
      const float* Xj = Xj_base + (size_t)k * 3;
      float Xj_Ci[3];
      actSim3(tij, qij, sij, Xj, Xj_Ci);
      const float q = Q[k];
      const float ci = Ci_base[ind_Xi];
      const float cj = Cj_base[k];
      int valid = vm & (q > P->Q_thresh) & (ci > P->C_thresh) & (cj > P->C_thresh);

      if (P->kind == 0) { /* ---- ray_align_kernel :924-1089 ---- */
        const float norm2_i = Xi[0] * Xi[0] + Xi[1] * Xi[1] + Xi[2] * Xi[2];
        const float norm1_i = sqrtf(norm2_i);
        const float norm1_i_inv = 1.0f / norm1_i;
        float ri[3];
        for (int i = 0; i < 3; i++) ri[i] = norm1_i_inv * Xi[i];
        const float norm2_j = Xj_Ci[0] * Xj_Ci[0] + Xj_Ci[1] * Xj_Ci[1] + Xj_Ci[2] * Xj_Ci[2];
        const float norm1_j = sqrtf(norm2_j);
        const float norm1_j_inv = 1.0f / norm1_j;
        float rj[3];
        for (int i = 0; i < 3; i++) rj[i] = norm1_j_inv * Xj_Ci[i];
        float err[4] = {rj[0] - ri[0], rj[1] - ri[1], rj[2] - ri[2], norm1_j - norm1_i};
        const float sq = sqrtf(q);
        const float sqrt_w_ray = valid ? sa_inv * sq : 0.0f;
        const float sqrt_w_dist = valid ? sb_inv * sq : 0.0f;
        float w[4];
        w[0] = huber_w(sqrt_w_ray * err[0]);
        w[1] = huber_w(sqrt_w_ray * err[1]);
        w[2] = huber_w(sqrt_w_ray * err[2]);
        w[3] = huber_w(sqrt_w_dist * err[3]);
        const float wr = sqrt_w_ray * sqrt_w_ray, wd = sqrt_w_dist * sqrt_w_dist;
        w[0] *= wr; w[1] *= wr; w[2] *= wr; w[3] *= wd;
        const float norm3_j_inv = norm1_j_inv / norm2_j;
        const float drx_dPx = norm1_j_inv - Xj_Ci[0] * Xj_Ci[0] * norm3_j_inv;
        const float dry_dPy = norm1_j_inv - Xj_Ci[1] * Xj_Ci[1] * norm3_j_inv;
        const float drz_dPz = norm1_j_inv - Xj_Ci[2] * Xj_Ci[2] * norm3_j_inv;
        const float drx_dPy = -Xj_Ci[0] * Xj_Ci[1] * norm3_j_inv;
        const float drx_dPz = -Xj_Ci[0] * Xj_Ci[2] * norm3_j_inv;
        const float dry_dPz = -Xj_Ci[1] * Xj_Ci[2] * norm3_j_inv;
        Ji[0] = drx_dPx; Ji[1] = drx_dPy; Ji[2] = drx_dPz; Ji[3] = 0.0f; Ji[4] = rj[2]; Ji[5] = -rj[1]; Ji[6] = 0.0f;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[0], err[0]);
        Ji[0] = drx_dPy; Ji[1] = dry_dPy; Ji[2] = dry_dPz; Ji[3] = -rj[2]; Ji[4] = 0.0f; Ji[5] = rj[0]; Ji[6] = 0.0f;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[1], err[1]);
        Ji[0] = drx_dPz; Ji[1] = dry_dPz; Ji[2] = drz_dPz; Ji[3] = rj[1]; Ji[4] = -rj[0]; Ji[5] = 0.0f; Ji[6] = 0.0f;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[2], err[2]);
        Ji[0] = rj[0]; Ji[1] = rj[1]; Ji[2] = rj[2]; Ji[3] = 0.0f; Ji[4] = 0.0f; Ji[5] = 0.0f; Ji[6] = norm1_j;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[3], err[3]);
      } else if (P->kind == 1) { /* ---- calib_proj_kernel :1360-1495 ---- */
        const int u_target = (int)(ind_Xi % P->width);
        const int v_target = (int)(ind_Xi / P->width);
        const int valid_z = (Xj_Ci[2] > P->z_eps) && (Xi[2] > P->z_eps);
        const float zj_inv = valid_z ? 1.0f / Xj_Ci[2] : 0.0f;
        const float zj_log = valid_z ? logf(Xj_Ci[2]) : 0.0f;
        const float zi_log = valid_z ? logf(Xi[2]) : 0.0f;
        const float x_div_z = Xj_Ci[0] * zj_inv;
        const float y_div_z = Xj_Ci[1] * zj_inv;
        const float u = P->fx * x_div_z + P->cx;
        const float v = P->fy * y_div_z + P->cy;
        const int valid_u = (u > (float)P->pixel_border) && (u < (float)(P->width - 1 - P->pixel_border));
        const int valid_v = (v > (float)P->pixel_border) && (v < (float)(P->height - 1 - P->pixel_border));
        float err[3] = {u - (float)u_target, v - (float)v_target, zj_log - zi_log};
        valid = valid & valid_u & valid_v & valid_z;
        const float sq = sqrtf(q);
        const float sqrt_w_pixel = valid ? sa_inv * sq : 0.0f;
        const float sqrt_w_depth = valid ? sb_inv * sq : 0.0f;
        float w[3];
        w[0] = huber_w(sqrt_w_pixel * err[0]);
        w[1] = huber_w(sqrt_w_pixel * err[1]);
        w[2] = huber_w(sqrt_w_depth * err[2]);
        const float wp = sqrt_w_pixel * sqrt_w_pixel, wd = sqrt_w_depth * sqrt_w_depth;
        w[0] *= wp; w[1] *= wp; w[2] *= wd;
        const float fx = P->fx, fy = P->fy;
        Ji[0] = fx * zj_inv; Ji[1] = 0.0f; Ji[2] = -fx * x_div_z * zj_inv; Ji[3] = -fx * x_div_z * y_div_z;
        Ji[4] = fx * (1 + x_div_z * x_div_z); Ji[5] = -fx * y_div_z; Ji[6] = 0.0f;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[0], err[0]);
        Ji[0] = 0.0f; Ji[1] = fy * zj_inv; Ji[2] = -fy * y_div_z * zj_inv; Ji[3] = -fy * (1 + y_div_z * y_div_z);
        Ji[4] = fy * x_div_z * y_div_z; Ji[5] = fy * x_div_z; Ji[6] = 0.0f;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[1], err[1]);
        Ji[0] = 0.0f; Ji[1] = 0.0f; Ji[2] = zj_inv; Ji[3] = y_div_z; Ji[4] = -x_div_z; Ji[5] = 0.0f; Ji[6] = 1.0f;
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[2], err[2]);
      } else { /* ---- point_align_kernel :552-674 ---- */
        float err[3] = {Xj_Ci[0] - Xi[0], Xj_Ci[1] - Xi[1], Xj_Ci[2] - Xi[2]};
        const float sqrt_w_point = valid ? sa_inv * sqrtf(q) : 0.0f;
        float w[3];
        w[0] = huber_w(sqrt_w_point * err[0]);
        w[1] = huber_w(sqrt_w_point * err[1]);
        w[2] = huber_w(sqrt_w_point * err[2]);
        const float wc = sqrt_w_point * sqrt_w_point;
        w[0] *= wc; w[1] *= wc; w[2] *= wc;
        Ji[0] = 1.0f; Ji[1] = 0.0f; Ji[2] = 0.0f; Ji[3] = 0.0f; Ji[4] = Xj_Ci[2]; Ji[5] = -Xj_Ci[1]; Ji[6] = Xj_Ci[0];
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[0], err[0]);
        Ji[0] = 0.0f; Ji[1] = 1.0f; Ji[2] = 0.0f; Ji[3] = -Xj_Ci[2]; Ji[4] = 0.0f; Ji[5] = Xj_Ci[0]; Ji[6] = Xj_Ci[1];
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[1], err[1]);
        Ji[0] = 0.0f; Ji[1] = 0.0f; Ji[2] = 1.0f; Ji[3] = Xj_Ci[1]; Ji[4] = -Xj_Ci[0]; Ji[5] = 0.0f; Ji[6] = Xj_Ci[2];
        accum_row(hij, vi, vj, Jx, ti, qi, si, w[2], err[2]);
      }
    }
  }

  float sdata[THREADS];
  for (int n = 0; n < 7; n++) {
    for (int t = 0; t < THREADS; t++) sdata[t] = vi_all[t * 7 + n];
    gs[0 * g_stride + n] = block_reduce(sdata);
    for (int t = 0; t < THREADS; t++) sdata[t] = vj_all[t * 7 + n];
    gs[1 * g_stride + n] = block_reduce(sdata);
  }
  int l = 0;
  for (int n = 0; n < 14; n++)
    for (int m = 0; m <= n; m++) {
      for (int t = 0; t < THREADS; t++) sdata[t] = hij_all[(size_t)t * H_DIM + l];
      const float v = block_reduce(sdata);
      if (n < 7 && m < 7) {
        Hs[0 * h_stride + n * 7 + m] = v;
        Hs[0 * h_stride + m * 7 + n] = v;
      } else if (n >= 7 && m < 7) {
        Hs[1 * h_stride + m * 7 + (n - 7)] = v;
        Hs[2 * h_stride + (n - 7) * 7 + m] = v;
      } else {
        Hs[3 * h_stride + (n - 7) * 7 + (m - 7)] = v;
        Hs[3 * h_stride + (m - 7) * 7 + (n - 7)] = v;
      }
      l++;
    }
  free(hij_all);
  free(vi_all);
  free(vj_all);
}

/* One launch of {ray_align,calib_proj,point_align}_kernel over all edges.
 * ii_edge/jj_edge are ROW indices into Twc/Xs (already searchsorted). Hs (4,E,7,7), gs (2,E,7). */
void oracle_gn_edges(int kind, const float* Twc, const float* Xs, const float* Cs, const float* K,
                     const int64_t* ii_edge, const int64_t* jj_edge, const int64_t* idx_ii2jj,
                     const uint8_t* valid_match, const float* Q, int num_points, int E,
                     float sigma_a, float sigma_b, float C_thresh, float Q_thresh, int height,
                     int width, int pixel_border, float z_eps, float* Hs, float* gs) {
  gn_params P;
  memset(&P, 0, sizeof(P));
  P.kind = kind; P.sigma_a = sigma_a; P.sigma_b = sigma_b; P.C_thresh = C_thresh; P.Q_thresh = Q_thresh;
  P.height = height; P.width = width; P.pixel_border = pixel_border; P.z_eps = z_eps;
  if (K) { P.fx = K[0]; P.fy = K[4]; P.cx = K[2]; P.cy = K[5]; }
#pragma omp parallel for schedule(dynamic)
  for (int e = 0; e < E; e++) {
    edge_kernel(&P, Twc, Xs, Cs, ii_edge[e], jj_edge[e], idx_ii2jj + (size_t)e * num_points,
                valid_match + (size_t)e * num_points, Q + (size_t)e * num_points, num_points,
                Hs + (size_t)e * 49, (size_t)E * 49, gs + (size_t)e * 7, (size_t)E * 7);
  }
}

static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

/* dense fp64 LL^T, in place on the lower triangle; returns 0 on success */
static int cholesky_solve(double* A, double* b, int n) {
  for (int j = 0; j < n; j++) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; k++) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; k++) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= A[(size_t)i * n + k] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = b[i];
    for (int k = i + 1; k < n; k++) s -= A[(size_t)k * n + i] * b[k];
    b[i] = s / A[(size_t)i * n + i];
  }
  return 0;
}

/* Assemble (SparseBlock::update_lhs/update_rhs, :71-113) and solve (:132-153) in fp64.
 * Hs (4,E,7,7), gs (2,E,7), ii_opt/jj_opt (E) pose indices minus num_fix (negative = pinned).
 * dx (N,7) float = -x ; returns 0 on success, 1 when LLT fails (dx = 0, :147-150). */
int oracle_gn_solve(const float* Hs, const float* gs, const int64_t* ii_opt, const int64_t* jj_opt,
                    int E, int N, float* dx) {
  const int n = N * 7;
  double* A = (double*)calloc((size_t)n * n, sizeof(double));
  double* b = (double*)calloc((size_t)n, sizeof(double));
  const int64_t* rows[4] = {ii_opt, ii_opt, jj_opt, jj_opt};
  const int64_t* cols[4] = {ii_opt, jj_opt, ii_opt, jj_opt};
  for (int blk = 0; blk < 4; blk++)
    for (int e = 0; e < E; e++) {
      const int64_t i = rows[blk][e], j = cols[blk][e];
      if (i >= 0 && j >= 0)
        for (int k = 0; k < 7; k++)
          for (int l = 0; l < 7; l++)
            A[(size_t)(7 * i + k) * n + (7 * j + l)] += (double)Hs[((size_t)blk * E + e) * 49 + k * 7 + l];
    }
  const int64_t* gi[2] = {ii_opt, jj_opt};
  for (int blk = 0; blk < 2; blk++)
    for (int e = 0; e < E; e++) {
      const int64_t i = gi[blk][e];
      if (i >= 0)
        for (int k = 0; k < 7; k++) b[7 * i + k] += (double)gs[((size_t)blk * E + e) * 7 + k];
    }
  int fail = cholesky_solve(A, b, n);
  for (int k = 0; k < n; k++) dx[k] = fail ? 0.0f : -(float)b[k];
  free(A);
  free(b);
  return fail;
}

/* Full host loop gauss_newton_{rays,calib,points}_cuda.  Twc (P,8) is updated IN PLACE.
 * ii/jj hold global keyframe ids; dx_out (P-1,7).  Returns iterations run. */
int oracle_gauss_newton(int kind, float* Twc, const float* Xs, const float* Cs, const float* K,
                        const int64_t* ii, const int64_t* jj, const int64_t* idx_ii2jj,
                        const uint8_t* valid_match, const float* Q, int num_poses, int num_points,
                        int E, float sigma_a, float sigma_b, float C_thresh, float Q_thresh,
                        int height, int width, int pixel_border, float z_eps, int max_iter,
                        float delta_thresh, float* dx_out) {
  const int num_fix = 1;
  int64_t* all = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)(E > 0 ? E : 1));
  for (int e = 0; e < E; e++) { all[e] = ii[e]; all[E + e] = jj[e]; }
  qsort(all, 2 * (size_t)E, sizeof(int64_t), cmp_i64);
  int nu = 0;
  for (int k = 0; k < 2 * E; k++)
    if (k == 0 || all[k] != all[k - 1]) all[nu++] = all[k];
  int64_t* ie = (int64_t*)malloc(sizeof(int64_t) * 4 * (size_t)(E > 0 ? E : 1));
  int64_t *je = ie + E, *io = ie + 2 * E, *jo = ie + 3 * E;
  for (int e = 0; e < E; e++) { /* searchsorted (left) */
    int a = 0, c = 0;
    while (a < nu && all[a] < ii[e]) a++;
    while (c < nu && all[c] < jj[e]) c++;
    ie[e] = a; je[e] = c; io[e] = a - num_fix; jo[e] = c - num_fix;
  }
  float* Hs = (float*)calloc((size_t)4 * E * 49 + 1, sizeof(float));
  float* gs = (float*)calloc((size_t)2 * E * 7 + 1, sizeof(float));
  const int N = num_poses - num_fix;
  int it = 0;
  for (it = 0; it < max_iter; it++) {
    oracle_gn_edges(kind, Twc, Xs, Cs, K, ie, je, idx_ii2jj, valid_match, Q, num_points, E, sigma_a,
                    sigma_b, C_thresh, Q_thresh, height, width, pixel_border, z_eps, Hs, gs);
    oracle_gn_solve(Hs, gs, io, jo, E, N, dx_out);
    oracle_pose_retr(Twc, dx_out, num_poses, num_fix);
    float ss = 0.0f;
    for (int k = 0; k < N * 7; k++) ss += dx_out[k] * dx_out[k];
    if (sqrtf(ss) < delta_thresh) { it++; break; }
  }
  free(all); free(ie); free(Hs); free(gs);
  return it;
}
