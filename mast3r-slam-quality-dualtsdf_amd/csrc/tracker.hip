// Frame-to-keyframe Sim3 tracking on gfx950: the whole Gauss-Newton loop of the reference's
// FrameTracker (<= 50 iterations of ~30 small torch kernels + one .item() sync each) as two kernels per
// iteration and NO host synchronisation inside the loop.
//
// Reference behaviour reproduced:
//   opt_pose_ray_dist_sim3 / opt_pose_calib_sim3 / solve      mast3r_slam/tracker.py:208-318
//   act_Sim3, point_to_ray_dist, project_calib                mast3r_slam/geometry.py:17-104
//   huber, check_convergence                                  mast3r_slam/nonlinear_optimizer.py:5-33
//
// Per iteration:  accumulate  H = sum w x x^T (28), g = sum w e x (7), cost = 1/2 sum w e^2 over all
// pixels (residual e = h(T Xf) - z, rows x = d h / d(left perturbation of T); w = info * huber) with the
// same 35(+1)-accumulator wave64 reduction as the backend kernels, then ONE small workgroup: combine
// partials in fixed order, 7x7 Cholesky solve (fp64), T <- exp(tau) T, convergence test
// (|cost decrease|/old < rel_thresh or |tau| < delta_thresh) -> device-side `done` flag.
#include "common.h"
#include "sim3.h"

namespace mslam {

constexpr int kTAcc = 36;  // 28 + 7 + cost

struct TrackState {
  int done;        // converged (or failed): later launches are no-ops
  int iters;
  int chol_fail;   // torch.linalg.cholesky would have raised (tracker.py:72-93 -> frame skipped)
  float old_cost;  // +inf before the first iteration
  float last_cost;
  float last_delta_norm;
  float pad0, pad1;
};

struct TrackParams {
  float inv_sigma_a, inv_sigma_b;  // ray/dist or pixel/depth
  float huber_k;
  const float* K;  // device f32[3,3] (calib only)
  float border_lo, border_hi_u, border_hi_v, z_eps;
  int width;
};

__device__ __forceinline__ float huber_k(float r, float k) {
  const float a = fabsf(r);
  return a < k ? 1.0f : k / a;
}

template <unsigned NZ>
__device__ __forceinline__ void taccum_row(float (&acc)[kTAcc], const float (&x)[7], float w, float err) {
  const float we = w * err;
  int l = 0;
#pragma unroll
  for (int n = 0; n < 7; n++) {
    const float wx = w * x[n];
#pragma unroll
    for (int m = 0; m <= n; m++) {
      if (((NZ >> n) & 1u) && ((NZ >> m) & 1u)) acc[l] = fmaf(wx, x[m], acc[l]);
      l++;
    }
    if ((NZ >> n) & 1u) acc[28 + n] = fmaf(we, x[n], acc[28 + n]);
  }
  acc[35] = fmaf(we, err, acc[35]);
}

// CALIB = 0: ray + distance residual (tracker.py:225-266); 1: pixel + log-depth (tracker.py:268-318)
template <int CALIB>
__global__ __launch_bounds__(256) void track_accum_kernel(const TrackState* __restrict__ st,
                                                          const float* __restrict__ T,  // relative pose T_CkCf (8)
                                                          const float* __restrict__ Xf, const float* __restrict__ Xk,
                                                          const int64_t* __restrict__ idx, const float* __restrict__ Qk,
                                                          const uint8_t* __restrict__ valid, int n, TrackParams P,
                                                          float* __restrict__ partial) {
  if (st->done) return;
  const Sim3f Tr = sim3_load(T);
  float R[9];
  quat_to_mat(Tr.q, R);
  float acc[kTAcc];
#pragma unroll
  for (int l = 0; l < kTAcc; l++) acc[l] = 0.0f;
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    const long long j = idx[k];
    const float xf0 = Xf[j * 3], xf1 = Xf[j * 3 + 1], xf2 = Xf[j * 3 + 2];
    const float xk0 = Xk[(size_t)k * 3], xk1 = Xk[(size_t)k * 3 + 1], xk2 = Xk[(size_t)k * 3 + 2];
    const float p0 = Tr.s * (R[0] * xf0 + R[1] * xf1 + R[2] * xf2) + Tr.t[0];
    const float p1 = Tr.s * (R[3] * xf0 + R[4] * xf1 + R[5] * xf2) + Tr.t[1];
    const float p2 = Tr.s * (R[6] * xf0 + R[7] * xf1 + R[8] * xf2) + Tr.t[2];
    const float vq = valid[k] ? sqrtf(Qk[k]) : 0.0f;
    if constexpr (CALIB == 0) {
      const float nk = sqrtf(xk0 * xk0 + xk1 * xk1 + xk2 * xk2), nk_inv = 1.0f / nk;
      const float n2 = p0 * p0 + p1 * p1 + p2 * p2;
      const float np = sqrtf(n2), np_inv = 1.0f / np;
      const float r0 = p0 * np_inv, r1 = p1 * np_inv, r2 = p2 * np_inv;
      const float e0 = r0 - xk0 * nk_inv, e1 = r1 - xk1 * nk_inv, e2 = r2 - xk2 * nk_inv, e3 = np - nk;
      const float sa = P.inv_sigma_a * vq, sb = P.inv_sigma_b * vq;
      const float w0 = huber_k(sa * e0, P.huber_k) * sa * sa, w1 = huber_k(sa * e1, P.huber_k) * sa * sa;
      const float w2 = huber_k(sa * e2, P.huber_k) * sa * sa, w3 = huber_k(sb * e3, P.huber_k) * sb * sb;
      const float n3 = np_inv / n2;
      const float dxx = np_inv - p0 * p0 * n3, dyy = np_inv - p1 * p1 * n3, dzz = np_inv - p2 * p2 * n3;
      const float dxy = -p0 * p1 * n3, dxz = -p0 * p2 * n3, dyz = -p1 * p2 * n3;
      { const float x[7] = {dxx, dxy, dxz, 0.0f, r2, -r1, 0.0f}; taccum_row<0b0110111>(acc, x, w0, e0); }
      { const float x[7] = {dxy, dyy, dyz, -r2, 0.0f, r0, 0.0f}; taccum_row<0b0101111>(acc, x, w1, e1); }
      { const float x[7] = {dxz, dyz, dzz, r1, -r0, 0.0f, 0.0f}; taccum_row<0b0011111>(acc, x, w2, e2); }
      { const float x[7] = {r0, r1, r2, 0.0f, 0.0f, 0.0f, np}; taccum_row<0b1000111>(acc, x, w3, e3); }
    } else {
      const float Pfx = P.K[0], Pfy = P.K[4], Pcx = P.K[2], Pcy = P.K[5];
      // measurement of pixel k: (u, v, log z_k) with validity z_k > eps (tracker.py:197-203)
      const float uk = (float)(k % P.width), vk = (float)(k / P.width);
      const bool valid_meas = xk2 > P.z_eps;
      const float lzk = valid_meas ? logf(xk2) : 0.0f;
      const float zinv = 1.0f / p2;
      const float u = Pfx * p0 * zinv + Pcx, v = Pfy * p1 * zinv + Pcy;
      const bool valid_z = p2 > P.z_eps;
      const bool valid_p = (u > P.border_lo) && (u < P.border_hi_u) && (v > P.border_lo) && (v < P.border_hi_v) && valid_z;
      const float lz = valid_z ? logf(p2) : 0.0f;
      const float gate = (valid_p && valid_meas) ? vq : 0.0f;
      const float e0 = u - (valid_meas ? uk : 0.0f), e1 = v - (valid_meas ? vk : 0.0f), e2 = lz - lzk;
      const float sa = P.inv_sigma_a * gate, sb = P.inv_sigma_b * gate;
      const float w0 = huber_k(sa * e0, P.huber_k) * sa * sa, w1 = huber_k(sa * e1, P.huber_k) * sa * sa;
      const float w2 = huber_k(sb * e2, P.huber_k) * sb * sb;
      const float xz = p0 * zinv, yz = p1 * zinv;
      // rows of d(u,v,log z)/dP [I, -[P]x, P]  (project_calib Jacobian geometry.py:92-102 x act_Sim3 :45-52)
      if (gate > 0.0f) {
        { const float x[7] = {Pfx * zinv, 0.0f, -Pfx * xz * zinv, -Pfx * xz * yz, Pfx * (1.0f + xz * xz), -Pfx * yz, 0.0f};
          taccum_row<0b0111101>(acc, x, w0, e0); }
        { const float x[7] = {0.0f, Pfy * zinv, -Pfy * yz * zinv, -Pfy * (1.0f + yz * yz), Pfy * xz * yz, Pfy * xz, 0.0f};
          taccum_row<0b0111110>(acc, x, w1, e1); }
        { const float x[7] = {0.0f, 0.0f, zinv, yz, -xz, 0.0f, 1.0f}; taccum_row<0b1011100>(acc, x, w2, e2); }
      }
    }
  }
  __shared__ float red[4][kTAcc];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < kTAcc; l++) {
    const float s = wave_sum(acc[l]);
    if (lane == 0) red[wid][l] = s;
  }
  __syncthreads();
  if (threadIdx.x < kTAcc)
    partial[(size_t)blockIdx.x * kTAcc + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void track_solve_kernel(TrackState* __restrict__ st, const float* __restrict__ partial,
                                                         int nblk, float* __restrict__ T, float rel_thresh,
                                                         float delta_thresh) {
  if (st->done) return;
  __shared__ double s[kTAcc];
  const int t = threadIdx.x;
  if (t < kTAcc) {
    double a = 0.0;
    for (int k = 0; k < nblk; k++) a += (double)partial[(size_t)k * kTAcc + t];
    s[t] = a;
  }
  __syncthreads();
  if (t != 0) return;
  double H[7][7], g[7];
  int l = 0;
  for (int n = 0; n < 7; n++)
    for (int m = 0; m <= n; m++) { H[n][m] = s[l]; H[m][n] = s[l]; l++; }
  for (int n = 0; n < 7; n++) g[n] = -s[28 + n];  // g = -A^T b
  const float cost = (float)(0.5 * s[35]);
  // Cholesky H = L L^T ; failure <=> torch.linalg.cholesky raises
  double L[7][7];
  bool fail = false;
  for (int j = 0; j < 7 && !fail; j++) {
    double d = H[j][j];
    for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
    if (!(d > 0.0)) { fail = true; break; }
    L[j][j] = sqrt(d);
    for (int i = j + 1; i < 7; i++) {
      double v = H[i][j];
      for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
      L[i][j] = v / L[j][j];
    }
  }
  st->iters += 1;
  st->last_cost = cost;
  if (fail) { st->chol_fail = 1; st->done = 1; return; }
  double y[7], x[7];
  for (int i = 0; i < 7; i++) { double v = g[i]; for (int k = 0; k < i; k++) v -= L[i][k] * y[k]; y[i] = v / L[i][i]; }
  for (int i = 6; i >= 0; i--) { double v = y[i]; for (int k = i + 1; k < 7; k++) v -= L[k][i] * x[k]; x[i] = v / L[i][i]; }
  float tau[7];
  double nn = 0.0;
  for (int i = 0; i < 7; i++) { tau[i] = (float)x[i]; nn += (double)tau[i] * tau[i]; }
  sim3_store(T, sim3_unit(sim3_retr(tau, sim3_load(T))));   // T_CkCf.retr(tau) is a lietorch call (tracker.py:255)
  const float delta_norm = (float)sqrt(nn);
  st->last_delta_norm = delta_norm;
  // check_convergence (nonlinear_optimizer.py:5-25): first iteration old = inf -> rel_dec = nan -> false
  const float old = st->old_cost;
  const float rel_dec = fabsf((old - cost) / old);
  if (rel_dec < rel_thresh || delta_norm < delta_thresh) st->done = 1;
  st->old_cost = cost;
}

__global__ void track_init_kernel(TrackState* st) {
  st->done = 0; st->iters = 0; st->chol_fail = 0; st->old_cost = INFINITY; st->last_cost = 0.0f;
  st->last_delta_norm = 0.0f; st->pad0 = st->pad1 = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// The tensor glue around the pose loop as three launches (FrameTracker.track: tracker.py:44-75 masks and
// fractions, :147-177 the keyframe rule's inputs and the fusion of the keyframe's pointmap).  Every value is
// computed with the single IEEE operations the frontend's tensor expressions use (no contraction in this TU),
// in the same order, so the fused form is bit-identical to the op-by-op form it replaces.
// ---------------------------------------------------------------------------------------------
// Workspace of mslam_track_prepare: [256 B reserved][int2 per block: #valid_opt, #valid_kf][n bytes: 1 = some valid
// match points at this frame pixel].  No atomics: device-scope atomics with a return value serialise at the memory
// side (the first form of this kernel spent 70-100 us on 196 608 atomicOr); flags are plain byte stores of the same
// value, counted by the verdict kernel.
__device__ __host__ inline size_t prep_partials_bytes(int n) { return (((size_t)((n + 255) / 256) * 8) + 255) & ~(size_t)255; }

__global__ __launch_bounds__(256) void track_prep_kernel(
    const int64_t* __restrict__ idx, const uint8_t* __restrict__ vmatch, const float* __restrict__ Qff,
    const float* __restrict__ Qkf, const float* __restrict__ Cf_sum, float inv_nf, const float* __restrict__ Ck_sum,
    float inv_nk, float C_conf, float Q_conf, int n, const float* __restrict__ T_WCk, const float* __restrict__ T_WCf,
    float* __restrict__ Qk, float* __restrict__ Ck_avg, uint8_t* __restrict__ valid_opt, uint8_t* __restrict__ valid_kf,
    float* __restrict__ T_rel, int2* __restrict__ partial, uint8_t* __restrict__ hit) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k == 0) {   // T_CkCf = T_WCk^-1 * T_WCf, each factor a lietorch object (unit quaternion)
    const Sim3f Ti = sim3_unit(sim3_inv(sim3_load(T_WCk)));
    sim3_store(T_rel, sim3_unit(sim3_mul(Ti, sim3_load(T_WCf))));
  }
  bool opt = false, kf = false;
  if (k < n) {
    const long long j = idx[k];
    const bool vm = vmatch[k] != 0;
    const float q = sqrtf(Qff[j] * Qkf[k]);
    const float cf = Cf_sum[j] * inv_nf;      // tensor / python int on the device = multiply by the fp32 reciprocal
    const float ck = Ck_sum[k] * inv_nk;
    const bool vq = q > Q_conf;
    opt = vm && (cf > C_conf) && (ck > C_conf) && vq;
    kf = vm && vq;
    Qk[k] = q;
    Ck_avg[k] = ck;
    valid_opt[k] = opt ? 1 : 0;
    valid_kf[k] = kf ? 1 : 0;
    if (vm) hit[j] = 1;
  }
  __shared__ int red[4][2];
  const int c_opt = __popcll(__ballot(opt)), c_kf = __popcll(__ballot(kf));
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = c_opt; red[threadIdx.x >> 6][1] = c_kf; }
  __syncthreads();
  if (threadIdx.x == 0)
    partial[blockIdx.x] = make_int2(red[0][0] + red[1][0] + red[2][0] + red[3][0], red[0][1] + red[1][1] + red[2][1] + red[3][1]);
}

// {match_frac, iterations, chol_fail, kf_frac, unique_frac, done}: a mean of 0/1 values is count * (1 / n) in fp32
__global__ __launch_bounds__(1024) void track_verdict_kernel(const int2* __restrict__ partial, const uint8_t* __restrict__ hit,
                                                             const int* __restrict__ status, int n, float* __restrict__ out) {
  const int t = threadIdx.x;
  int c[3] = {0, 0, 0};
  const int nblk = (n + 255) / 256;
  for (int b = t; b < nblk; b += 1024) { const int2 p = partial[b]; c[0] += p.x; c[1] += p.y; }
  const int n16 = n >> 4;            // hit is 256-byte aligned: 16 flags per load, the tail byte by byte
  for (int v = t; v < n16; v += 1024) {
    const uint4 w = reinterpret_cast<const uint4*>(hit)[v];
    c[2] += __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);   // flags are 0 / 1 bytes
  }
  for (int v = (n16 << 4) + t; v < n; v += 1024) c[2] += hit[v];
  __shared__ int red[16][3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    int s = c[q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((t & 63) == 0) red[t >> 6][q] = s;
  }
  __syncthreads();
  if (t != 0) return;
  int tot[3] = {0, 0, 0};
  for (int w = 0; w < 16; w++) { tot[0] += red[w][0]; tot[1] += red[w][1]; tot[2] += red[w][2]; }
  const float inv_n = 1.0f / (float)n;
  out[0] = (float)tot[0] * inv_n;
  out[1] = (float)status[1];
  out[2] = (float)status[2];
  out[3] = (float)tot[1] * inv_n;
  out[4] = (float)tot[2] * inv_n;
  out[5] = (float)status[0];
}

// The effects of a tracked frame (tracker.py:147-168, frame.py:41-105 'weighted_pointmap'): T_WCf = T_WCk * T_CkCf,
// the keyframe's pointmap fused with the frame's view of it: X' = ((C X) + (Ckf Xkk)) / (C + Ckf), C' = C + Ckf with
// Xkk = T_CkCf . Xkf
__global__ __launch_bounds__(256) void track_fuse_kernel(const float* __restrict__ T_WCk, const float* __restrict__ T_rel,
                                                         const float* __restrict__ Xkf, const float* __restrict__ Ckf,
                                                         const float* __restrict__ Xc, const float* __restrict__ Cc, int n,
                                                         float* __restrict__ T_WCf, float* __restrict__ Xn,
                                                         float* __restrict__ Cn) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const Sim3f Tr = sim3_load(T_rel);
  if (k == 0) sim3_store(T_WCf, sim3_unit(sim3_mul(sim3_load(T_WCk), Tr)));
  if (k >= n) return;
  const float x[3] = {Xkf[(size_t)k * 3], Xkf[(size_t)k * 3 + 1], Xkf[(size_t)k * 3 + 2]};
  float y[3];
  sim3_act(Tr, x, y);
  const float c = Cc[k], cn = Ckf[k], den = c + cn;
#pragma unroll
  for (int d = 0; d < 3; d++) Xn[(size_t)k * 3 + d] = ((c * Xc[(size_t)k * 3 + d]) + (cn * y[d])) / den;
  Cn[k] = den;
}

}  // namespace mslam

using namespace mslam;

extern "C" size_t mslam_track_workspace_bytes(int n_points) {
  (void)n_points;
  return 256 + sizeof(float) * kTAcc * 1024;
}

// T_rel f32[8] (device, updated in place) ; status_out (device, 8 x 4 bytes, may be NULL) receives the
// TrackState {done, iters, chol_fail, old_cost, last_cost, last_delta_norm, -, -}.
extern "C" int mslam_track_pose(int use_calib, float* T_rel, const float* Xf, const float* Xk, const int64_t* idx_f2k,
                                const float* Qk, const uint8_t* valid, int n_points, const float* K, int width,
                                int height, float sigma_a, float sigma_b, float huber, int pixel_border, float z_eps,
                                int first_iter, int max_iters, float rel_error, float delta_norm, void* status_out,
                                void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(n_points > 0 && max_iters >= 0 && first_iter >= 0, "track_pose: bad sizes");
  MSLAM_REQUIRE(T_rel && Xf && Xk && idx_f2k && Qk && valid && workspace, "track_pose: null pointer");
  MSLAM_REQUIRE(!use_calib || (K && width > 0 && height > 0), "track_pose: calib needs K, width, height");
  MSLAM_REQUIRE(workspace_bytes >= mslam_track_workspace_bytes(n_points), "track_pose: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  TrackState* st = (TrackState*)workspace;
  float* partial = (float*)((char*)workspace + 256);
  int nblk = (n_points + 256 * 8 - 1) / (256 * 8);  // >= 8 points per thread
  if (nblk > 1024) nblk = 1024;
  if (nblk < 1) nblk = 1;
  TrackParams P = {};
  P.inv_sigma_a = 1.0f / sigma_a; P.inv_sigma_b = 1.0f / sigma_b; P.huber_k = huber;
  P.width = width > 0 ? width : 1;
  if (use_calib) {
    P.K = K;
    P.border_lo = (float)pixel_border; P.border_hi_u = (float)(width - 1 - pixel_border);
    P.border_hi_v = (float)(height - 1 - pixel_border); P.z_eps = z_eps;
  }
  if (first_iter == 0) hipLaunchKernelGGL(track_init_kernel, dim3(1), dim3(1), 0, s, st);
  for (int it = first_iter; it < max_iters; it++) {
    if (use_calib)
      hipLaunchKernelGGL(track_accum_kernel<1>, dim3(nblk), dim3(256), 0, s, st, T_rel, Xf, Xk, idx_f2k, Qk, valid,
                         n_points, P, partial);
    else
      hipLaunchKernelGGL(track_accum_kernel<0>, dim3(nblk), dim3(256), 0, s, st, T_rel, Xf, Xk, idx_f2k, Qk, valid,
                         n_points, P, partial);
    hipLaunchKernelGGL(track_solve_kernel, dim3(1), dim3(64), 0, s, st, partial, nblk, T_rel, rel_error, delta_norm);
  }
  MSLAM_LAUNCH_CHECK("track_pose");
  if (status_out)
    return check_hip(hipMemcpyAsync(status_out, st, sizeof(TrackState), hipMemcpyDeviceToDevice, s), "track status");
  return MSLAM_OK;
}

extern "C" size_t mslam_track_prepare_workspace_bytes(int n_points) {
  return 256 + prep_partials_bytes(n_points) + (((size_t)n_points + 255) & ~(size_t)255);
}

extern "C" int mslam_track_prepare(const int64_t* idx_f2k, const uint8_t* valid_match, const float* Qff,
                                   const float* Qkf, const float* Cf_sum, float inv_nf, const float* Ck_sum,
                                   float inv_nk, float C_conf, float Q_conf, int n_points, const float* T_WCk,
                                   const float* T_WCf, float* Qk, float* Ck_avg, uint8_t* valid_opt, uint8_t* valid_kf,
                                   float* T_rel, void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(n_points > 0, "track_prepare: bad sizes");
  MSLAM_REQUIRE(idx_f2k && valid_match && Qff && Qkf && Cf_sum && Ck_sum && T_WCk && T_WCf && Qk && Ck_avg && valid_opt &&
                    valid_kf && T_rel && workspace, "track_prepare: null pointer");
  MSLAM_REQUIRE(workspace_bytes >= mslam_track_prepare_workspace_bytes(n_points), "track_prepare: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  MSLAM_REQUIRE(((uintptr_t)workspace & 15) == 0, "track_prepare: workspace must be 16-byte aligned");
  uint8_t* hit = (uint8_t*)workspace + 256 + prep_partials_bytes(n_points);
  int rc = check_hip(hipMemsetAsync(hit, 0, (size_t)n_points, s), "track_prepare: memset");
  if (rc) return rc;
  hipLaunchKernelGGL(track_prep_kernel, dim3((n_points + 255) / 256), dim3(256), 0, s, idx_f2k, valid_match, Qff, Qkf,
                     Cf_sum, inv_nf, Ck_sum, inv_nk, C_conf, Q_conf, n_points, T_WCk, T_WCf, Qk, Ck_avg, valid_opt,
                     valid_kf, T_rel, (int2*)((char*)workspace + 256), hit);
  MSLAM_LAUNCH_CHECK("track_prepare");
  return MSLAM_OK;
}

extern "C" int mslam_track_verdict(const void* prepare_workspace, const void* status, int n_points, float* verdict6,
                                   void* stream) {
  MSLAM_REQUIRE(prepare_workspace && status && verdict6 && n_points > 0, "track_verdict: bad arguments");
  const char* ws = (const char*)prepare_workspace;
  hipLaunchKernelGGL(track_verdict_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const int2*)(ws + 256),
                     (const uint8_t*)(ws + 256 + prep_partials_bytes(n_points)), (const int*)status, n_points, verdict6);
  MSLAM_LAUNCH_CHECK("track_verdict");
  return MSLAM_OK;
}

extern "C" int mslam_track_fuse(const float* T_WCk, const float* T_rel, const float* Xkf, const float* Ckf,
                                const float* X_canon, const float* C, int n_points, float* T_WCf, float* X_new,
                                float* C_new, void* stream) {
  MSLAM_REQUIRE(n_points > 0, "track_fuse: bad sizes");
  MSLAM_REQUIRE(T_WCk && T_rel && Xkf && Ckf && X_canon && C && T_WCf && X_new && C_new, "track_fuse: null pointer");
  hipLaunchKernelGGL(track_fuse_kernel, dim3((n_points + 255) / 256), dim3(256), 0, (hipStream_t)stream, T_WCk, T_rel, Xkf,
                     Ckf, X_canon, C, n_points, T_WCf, X_new, C_new);
  MSLAM_LAUNCH_CHECK("track_fuse");
  return MSLAM_OK;
}
