// Gauss-Newton backend for gfx950: per-edge residual/JtJ accumulation (rays | calib | points),
// fp64 normal-equation assembly, blocked dense fp64 Cholesky, Sim3 retraction — all on the
// device, no host round trip and no host synchronisation inside the GN loop.
//
// Reference behaviour reproduced (re-designed, not translated):
//   ray_align_kernel / calib_proj_kernel / point_align_kernel
//        mast3r_slam/backend/src/gn_kernels.cu:813-1138, 1231-1543, 455-723
//   host loop + SparseBlock (Eigen SimplicialLLT, fp64)   gn_kernels.cu:57-159, 1140-1228
//   get_unique_kf_idx / create_inds                        gn_kernels.cu:161-170
//   pose_retr_kernel                                        gn_kernels.cu:415-453
//
// MI355X-first design decisions:
//   * apply_Sim3_adj_inv is linear: Jj = M_i x with a per-edge 7x7 matrix M_i, and Ji = -Jj.
//     So the 14x14 block Hessian is [[A,-A],[-A,A]] with A = M (sum w x x^T) M^T.  The streaming
//     kernel accumulates only B = sum w x x^T (28 values) and u = sum w e x (7 values) of the RAW
//     (pre-adjoint) Jacobian rows - 35 accumulators per lane instead of 119, and no quaternion
//     algebra per residual row.  M is applied once per edge, in fp64, in the reduce kernel.
//   * every edge is split over several workgroups (>= 1k workgroups per launch instead of one
//     256-thread block per edge), wave64 shuffle reduction + one LDS pass, partials combined in
//     a fixed order (deterministic, no atomics).
//   * the fp64 solve never leaves the GPU: assemble -> right-looking blocked LL^T on the
//     augmented matrix [H | b] (forward substitution comes for free) -> blocked back substitution
//     -> dx = -x -> retraction -> ||dx|| test sets a device-side `done` flag that later
//     iterations' kernels read and exit on (replaces delta_norm.item(), gn_kernels.cu:1219-1222).
#include "common.h"
#include "sim3.h"

namespace mslam {

constexpr int kAcc = 35;      // 28 (lower triangle of 7x7) + 7
constexpr int kEdgeConst = 16;  // sR_ij (9) + t_ij (3) + pad
constexpr int kNB = 32;       // Cholesky panel width

struct GnState {
  int done;        // ||dx|| < delta_thresh reached (or LLT failure): later launches are no-ops
  int iters;       // GN iterations actually executed
  int chol_fail;   // current factorisation hit a non-positive pivot (Eigen: info() != Success)
  float last_norm;
};

// ---------------------------------------------------------------------------------------------
// index preparation: unique(sorted) + searchsorted, O(E^2) in one workgroup (E <= a few thousand)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_index_kernel(const int64_t* __restrict__ ii,
                                                       const int64_t* __restrict__ jj, int E,
                                                       int num_fix, int* __restrict__ ii_edge,
                                                       int* __restrict__ jj_edge, int* __restrict__ ii_opt,
                                                       int* __restrict__ jj_opt, int* __restrict__ first,
                                                       int* __restrict__ num_unique) {
  const int n2 = 2 * E;
  auto val = [&](int k) -> int64_t { return k < E ? ii[k] : jj[k - E]; };
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  // phase 1: first-occurrence flags
  int local_unique = 0;
  for (int k = threadIdx.x; k < n2; k += 256) {
    const int64_t v = val(k);
    int f = 1;
    for (int m = 0; m < k; m++)
      if (val(m) == v) { f = 0; break; }
    first[k] = f;
    local_unique += f;
  }
  __syncthreads();
  atomicAdd(&cnt, local_unique);
  // phase 2: rank = number of DISTINCT values smaller than v  (== searchsorted into unique())
  for (int k = threadIdx.x; k < n2; k += 256) {
    const int64_t v = val(k);
    int rank = 0;
    for (int m = 0; m < n2; m++) rank += (first[m] && val(m) < v) ? 1 : 0;
    if (k < E) { ii_edge[k] = rank; ii_opt[k] = rank - num_fix; }
    else { jj_edge[k - E] = rank; jj_opt[k - E] = rank - num_fix; }
  }
  __syncthreads();
  if (threadIdx.x == 0) *num_unique = cnt;
}

// per-edge constants: T_ij = Ti^-1 * Tj as scaled rotation + translation (thread 0's relSim3 in the
// reference, gn_kernels.cu:866-868)
__global__ void gn_edge_setup_kernel(const GnState* __restrict__ st, const float* __restrict__ Twc,
                                     const int* __restrict__ ii_edge, const int* __restrict__ jj_edge,
                                     int E, float* __restrict__ econst) {
  if (st->done) return;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const Sim3f Ti = sim3_load(Twc + 8 * ii_edge[e]);
  const Sim3f Tj = sim3_load(Twc + 8 * jj_edge[e]);
  const Sim3f Tij = sim3_rel(Ti, Tj);
  float R[9];
  quat_to_mat(Tij.q, R);
  float* c = econst + (size_t)e * kEdgeConst;
#pragma unroll
  for (int k = 0; k < 9; k++) c[k] = Tij.s * R[k];
  c[9] = Tij.t[0]; c[10] = Tij.t[1]; c[11] = Tij.t[2];
  c[12] = c[13] = c[14] = c[15] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// streaming accumulation
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float huber_w(float r) {
  const float a = fabsf(r);
  return a < 1.345f ? 1.0f : 1.345f / a;
}

// acc[0..27] += w * x x^T (lower triangle, row-major n>=m), acc[28..34] += w*err*x.
// NZ is a compile-time mask of the structurally non-zero entries of x.
template <unsigned NZ>
__device__ __forceinline__ void accum_row(float (&acc)[kAcc], const float (&x)[7], float w, float err) {
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
}

struct GnParams {
  float sa_inv, sb_inv;  // 1/sigma_a, 1/sigma_b
  float C_thresh, Q_thresh;
  int height, width;
  float border_lo, border_hi_u, border_hi_v;  // pixel_border, width-1-border, height-1-border
  float z_eps;
  const float* K;  // device f32[3,3] (calib only)
};

template <int KIND>  // 0 rays, 1 calib, 2 points
__global__ __launch_bounds__(256) void gn_accum_kernel(
    const GnState* __restrict__ st, const float* __restrict__ econst, const float* __restrict__ Xs,
    const float* __restrict__ Cs, const int* __restrict__ ii_edge, const int* __restrict__ jj_edge,
    const int64_t* __restrict__ idx_ii2jj, const uint8_t* __restrict__ valid_match,
    const float* __restrict__ Q, int num_points, int chunk_len, GnParams P, float* __restrict__ partial) {
  if (st->done) return;
  const int e = blockIdx.y;
  const int chunk = blockIdx.x;
  const int S = gridDim.x;
  // wave-uniform per-edge constants -> scalar loads
  const float* c = econst + (size_t)e * kEdgeConst;
  const float r00 = c[0], r01 = c[1], r02 = c[2], r10 = c[3], r11 = c[4], r12 = c[5], r20 = c[6],
              r21 = c[7], r22 = c[8], t0 = c[9], t1 = c[10], t2 = c[11];
  const int ix = ii_edge[e], jx = jj_edge[e];
  const float* __restrict__ Xi_base = Xs + (size_t)ix * num_points * 3;
  const float* __restrict__ Xj_base = Xs + (size_t)jx * num_points * 3;
  const float* __restrict__ Ci_base = Cs + (size_t)ix * num_points;
  const float* __restrict__ Cj_base = Cs + (size_t)jx * num_points;
  const int64_t* __restrict__ idx_e = idx_ii2jj + (size_t)e * num_points;
  const uint8_t* __restrict__ vm_e = valid_match + (size_t)e * num_points;
  const float* __restrict__ Q_e = Q + (size_t)e * num_points;

  float acc[kAcc];
#pragma unroll
  for (int l = 0; l < kAcc; l++) acc[l] = 0.0f;

  const int k_begin = chunk * chunk_len;
  const int k_end = min(k_begin + chunk_len, num_points);
  for (int k = k_begin + (int)threadIdx.x; k < k_end; k += 256) {
    const bool vm = vm_e[k] != 0;
    const long long ind = vm ? idx_e[k] : 0;  // invalid matches read index 0 (gn_kernels.cu:914)
    const float xj0 = Xj_base[(size_t)k * 3 + 0], xj1 = Xj_base[(size_t)k * 3 + 1], xj2 = Xj_base[(size_t)k * 3 + 2];
    const float xi0 = Xi_base[ind * 3 + 0], xi1 = Xi_base[ind * 3 + 1], xi2 = Xi_base[ind * 3 + 2];
    const float q = Q_e[k];
    const float ci = Ci_base[ind];
    const float cj = Cj_base[k];
    bool valid = vm & (q > P.Q_thresh) & (ci > P.C_thresh) & (cj > P.C_thresh);

    // X_j in frame i
    const float p0 = fmaf(r00, xj0, fmaf(r01, xj1, fmaf(r02, xj2, t0)));
    const float p1 = fmaf(r10, xj0, fmaf(r11, xj1, fmaf(r12, xj2, t1)));
    const float p2 = fmaf(r20, xj0, fmaf(r21, xj1, fmaf(r22, xj2, t2)));

    if constexpr (KIND == 0) {
      const float n2i = fmaf(xi2, xi2, fmaf(xi1, xi1, xi0 * xi0));
      const float n1i = sqrtf(n2i);
      const float n1i_inv = 1.0f / n1i;
      const float n2j = fmaf(p2, p2, fmaf(p1, p1, p0 * p0));
      const float n1j = sqrtf(n2j);
      const float n1j_inv = 1.0f / n1j;
      const float rj0 = p0 * n1j_inv, rj1 = p1 * n1j_inv, rj2 = p2 * n1j_inv;
      const float e0 = rj0 - xi0 * n1i_inv, e1 = rj1 - xi1 * n1i_inv, e2 = rj2 - xi2 * n1i_inv;
      const float e3 = n1j - n1i;
      const float sq = sqrtf(q);
      const float swr = valid ? P.sa_inv * sq : 0.0f;
      const float swd = valid ? P.sb_inv * sq : 0.0f;
      const float wr = swr * swr, wd = swd * swd;
      const float w0 = huber_w(swr * e0) * wr, w1 = huber_w(swr * e1) * wr, w2 = huber_w(swr * e2) * wr;
      const float w3 = huber_w(swd * e3) * wd;
      const float n3 = n1j_inv / n2j;
      const float dxx = n1j_inv - p0 * p0 * n3, dyy = n1j_inv - p1 * p1 * n3, dzz = n1j_inv - p2 * p2 * n3;
      const float dxy = -p0 * p1 * n3, dxz = -p0 * p2 * n3, dyz = -p1 * p2 * n3;
      {
        const float x[7] = {dxx, dxy, dxz, 0.0f, rj2, -rj1, 0.0f};
        accum_row<0b0110111>(acc, x, w0, e0);
      }
      {
        const float x[7] = {dxy, dyy, dyz, -rj2, 0.0f, rj0, 0.0f};
        accum_row<0b0101111>(acc, x, w1, e1);
      }
      {
        const float x[7] = {dxz, dyz, dzz, rj1, -rj0, 0.0f, 0.0f};
        accum_row<0b0011111>(acc, x, w2, e2);
      }
      {
        const float x[7] = {rj0, rj1, rj2, 0.0f, 0.0f, 0.0f, n1j};
        accum_row<0b1000111>(acc, x, w3, e3);
      }
    } else if constexpr (KIND == 1) {
      const float Pfx = P.K[0], Pfy = P.K[4], Pcx = P.K[2], Pcy = P.K[5];
      const int u_t = (int)(ind % P.width), v_t = (int)(ind / P.width);
      const bool valid_z = (p2 > P.z_eps) && (xi2 > P.z_eps);
      const float zinv = valid_z ? 1.0f / p2 : 0.0f;
      const float zj_log = valid_z ? logf(p2) : 0.0f;
      const float zi_log = valid_z ? logf(xi2) : 0.0f;
      const float xz = p0 * zinv, yz = p1 * zinv;
      const float u = fmaf(Pfx, xz, Pcx), v = fmaf(Pfy, yz, Pcy);
      const bool valid_u = (u > P.border_lo) && (u < P.border_hi_u);
      const bool valid_v = (v > P.border_lo) && (v < P.border_hi_v);
      valid = valid & valid_u & valid_v & valid_z;
      const float e0 = u - (float)u_t, e1 = v - (float)v_t, e2 = zj_log - zi_log;
      const float sq = sqrtf(q);
      const float swp = valid ? P.sa_inv * sq : 0.0f;
      const float swd = valid ? P.sb_inv * sq : 0.0f;
      const float wp = swp * swp, wd = swd * swd;
      const float w0 = huber_w(swp * e0) * wp, w1 = huber_w(swp * e1) * wp, w2 = huber_w(swd * e2) * wd;
      {
        const float x[7] = {Pfx * zinv, 0.0f, -Pfx * xz * zinv, -Pfx * xz * yz, Pfx * (1.0f + xz * xz),
                            -Pfx * yz, 0.0f};
        accum_row<0b0111101>(acc, x, w0, e0);
      }
      {
        const float x[7] = {0.0f, Pfy * zinv, -Pfy * yz * zinv, -Pfy * (1.0f + yz * yz), Pfy * xz * yz,
                            Pfy * xz, 0.0f};
        accum_row<0b0111110>(acc, x, w1, e1);
      }
      {
        const float x[7] = {0.0f, 0.0f, zinv, yz, -xz, 0.0f, 1.0f};
        accum_row<0b1011100>(acc, x, w2, e2);
      }
    } else {
      const float e0 = p0 - xi0, e1 = p1 - xi1, e2 = p2 - xi2;
      const float swp = valid ? P.sa_inv * sqrtf(q) : 0.0f;
      const float wc = swp * swp;
      const float w0 = huber_w(swp * e0) * wc, w1 = huber_w(swp * e1) * wc, w2 = huber_w(swp * e2) * wc;
      {
        const float x[7] = {1.0f, 0.0f, 0.0f, 0.0f, p2, -p1, p0};
        accum_row<0b1110001>(acc, x, w0, e0);
      }
      {
        const float x[7] = {0.0f, 1.0f, 0.0f, -p2, 0.0f, p0, p1};
        accum_row<0b1101010>(acc, x, w1, e1);
      }
      {
        const float x[7] = {0.0f, 0.0f, 1.0f, p1, -p0, 0.0f, p2};
        accum_row<0b1011100>(acc, x, w2, e2);
      }
    }
  }

  // wave64 shuffle reduction, then one LDS pass over the 4 waves
  __shared__ float red[4][kAcc];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < kAcc; l++) {
    const float s = wave_sum(acc[l]);
    if (lane == 0) red[wid][l] = s;
  }
  __syncthreads();
  if (threadIdx.x < kAcc) {
    const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    partial[((size_t)e * S + chunk) * kAcc + threadIdx.x] = s;
  }
}

// combine the S partials of each edge in fixed order (fp64), apply the adjoint matrix M_i, and emit
// the reference's block layout: Hs[4,E,7,7] = [ii, ij, ji, jj], gs[2,E,7] = [i, j].
__global__ __launch_bounds__(64) void gn_reduce_kernel(const GnState* __restrict__ st,
                                                       const float* __restrict__ partial, int S,
                                                       const float* __restrict__ Twc,
                                                       const int* __restrict__ ii_edge, int e0, int E,
                                                       float* __restrict__ Hs, float* __restrict__ gs) {
  if (st->done) return;
  const int e = blockIdx.x;  // local edge; global slot e0 + e
  __shared__ double B[7][7];
  __shared__ double u[7];
  __shared__ double M[7][7];
  __shared__ double MB[7][7];
  const int t = threadIdx.x;
  if (t < kAcc) {
    double s = 0.0;
    for (int c = 0; c < S; c++) s += (double)partial[((size_t)e * S + c) * kAcc + t];
    if (t < 28) {
      int n = 0, rem = t;  // t = n(n+1)/2 + m
      while (rem > n) { rem -= n + 1; n++; }
      B[n][rem] = s;
      B[rem][n] = s;
    } else {
      u[t - 28] = s;
    }
  }
  if (t == 63) {
    // M = [[s^-1 R, 0, 0], [s^-1 [t]x R, R, 0], [s^-1 t^T R, 0, 1]]   (apply_Sim3_adj_inv, :277-297)
    const float* p = Twc + 8 * ii_edge[e];
    const double tx = p[0], ty = p[1], tz = p[2], x = p[3], y = p[4], z = p[5], w = p[6];
    const double sinv = 1.0 / (double)p[7];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
                            {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
                            {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    for (int a = 0; a < 7; a++)
      for (int b = 0; b < 7; b++) M[a][b] = 0.0;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        M[a][b] = sinv * R[a][b];
        M[3 + a][3 + b] = R[a][b];
      }
    for (int b = 0; b < 3; b++) {
      M[3][b] = sinv * (ty * R[2][b] - tz * R[1][b]);
      M[4][b] = sinv * (tz * R[0][b] - tx * R[2][b]);
      M[5][b] = sinv * (tx * R[1][b] - ty * R[0][b]);
      M[6][b] = sinv * (tx * R[0][b] + ty * R[1][b] + tz * R[2][b]);
    }
    M[6][6] = 1.0;
  }
  __syncthreads();
  if (t < 49) {
    const int a = t / 7, b = t % 7;
    double s = 0.0;
    for (int k = 0; k < 7; k++) s += M[a][k] * B[k][b];
    MB[a][b] = s;
  }
  __syncthreads();
  if (t < 49) {
    const int a = t / 7, b = t % 7;
    double s = 0.0;
    for (int k = 0; k < 7; k++) s += MB[a][k] * M[b][k];
    const float A = (float)s;
    const size_t o = (size_t)(e0 + e) * 49 + t;
    const size_t blk = (size_t)E * 49;
    Hs[o] = A;
    Hs[blk + o] = -A;
    Hs[2 * blk + o] = -A;
    Hs[3 * blk + o] = A;
  } else if (t < 56) {
    const int a = t - 49;
    double s = 0.0;
    for (int k = 0; k < 7; k++) s += M[a][k] * u[k];
    gs[(size_t)(e0 + e) * 7 + a] = -(float)s;
    gs[(size_t)E * 7 + (size_t)(e0 + e) * 7 + a] = (float)s;
  }
}

// ---------------------------------------------------------------------------------------------
// fp64 normal equations: Haug is (np+1) x ld row-major; rows 0..np-1 = H (padded with identity),
// row np = b^T.  One workgroup per pose block-row; edges visited in the reference's triplet order.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_assemble_kernel(const GnState* __restrict__ st,
                                                          const float* __restrict__ Hs,
                                                          const float* __restrict__ gs,
                                                          const int* __restrict__ ii_opt,
                                                          const int* __restrict__ jj_opt, int E, int N,
                                                          int np, int ld, double* __restrict__ Haug) {
  if (st->done) return;
  const int br = blockIdx.x;  // block row in [0, N]  (N = the b row, N+1.. = padding rows)
  const int n = N * 7;
  if (br == N + 1) {  // identity padding rows n..np-1
    for (int r = n; r < np; r++)
      for (int cidx = threadIdx.x; cidx < ld; cidx += 256) Haug[(size_t)r * ld + cidx] = (cidx == r) ? 1.0 : 0.0;
    return;
  }
  extern __shared__ double rowbuf[];  // 7 x ld
  const int rows = (br == N) ? 1 : 7;
  for (int k = threadIdx.x; k < rows * ld; k += 256) rowbuf[k] = 0.0;
  __syncthreads();
  if (br < N) {
    // lhs: blocks [ii,ii], [ii,jj], [jj,ii], [jj,jj]  (update_lhs, gn_kernels.cu:1201-1203)
    for (int blk = 0; blk < 4; blk++) {
      const int* rix = (blk < 2) ? ii_opt : jj_opt;
      const int* cix = (blk & 1) ? jj_opt : ii_opt;
      for (int e = 0; e < E; e++) {
        if (rix[e] != br) continue;  // wave-uniform
        const int cj = cix[e];
        if (cj < 0) continue;
        if (threadIdx.x < 49) {
          const int k = threadIdx.x / 7, l = threadIdx.x % 7;
          rowbuf[k * ld + 7 * cj + l] += (double)Hs[((size_t)blk * E + e) * 49 + threadIdx.x];
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 7 * ld; k += 256) Haug[(size_t)(7 * br + k / ld) * ld + (k % ld)] = rowbuf[k];
  } else {
    // rhs row (update_rhs, gn_kernels.cu:1205-1206): b[i] += gs[0][e] (i = ii), gs[1][e] (i = jj)
    if (threadIdx.x < 7) {
      for (int blk = 0; blk < 2; blk++) {
        const int* rix = blk ? jj_opt : ii_opt;
        for (int e = 0; e < E; e++) {
          const int i = rix[e];
          if (i >= 0) rowbuf[7 * i + threadIdx.x] += (double)gs[((size_t)blk * E + e) * 7 + threadIdx.x];
        }
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < ld; k += 256) Haug[(size_t)np * ld + k] = rowbuf[k];
  }
}

// panel step: factor the kNB x kNB diagonal block at j0 in LDS, then solve the rows below it
// (including the augmented b row): L21 = A21 L11^-T.
__global__ __launch_bounds__(256) void chol_panel_kernel(GnState* __restrict__ st, double* __restrict__ A,
                                                         int np, int ld, int j0) {
  if (st->done) return;
  __shared__ double L[kNB][kNB + 1];
  __shared__ int fail;
  const int t = threadIdx.x;
  if (t == 0) fail = 0;
  for (int k = t; k < kNB * kNB; k += 256) {
    const int r = k / kNB, c = k % kNB;
    L[r][c] = (c <= r) ? A[(size_t)(j0 + r) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < kNB; j++) {
    if (t == 0) {
      const double d = L[j][j];
      if (!(d > 0.0)) { fail = 1; L[j][j] = 1.0; }
      else L[j][j] = sqrt(d);
    }
    __syncthreads();
    const double dj = L[j][j];
    if (t > j && t < kNB) L[t][j] /= dj;
    __syncthreads();
    // rank-1 update of the trailing lower triangle
    for (int k = t; k < (kNB - j - 1) * (kNB - j - 1); k += 256) {
      const int r = j + 1 + k / (kNB - j - 1), c = j + 1 + k % (kNB - j - 1);
      if (c <= r) L[r][c] -= L[r][j] * L[c][j];
    }
    __syncthreads();
  }
  if (fail && t == 0) st->chol_fail = 1;
  for (int k = t; k < kNB * kNB; k += 256) {
    const int r = k / kNB, c = k % kNB;
    if (c <= r) A[(size_t)(j0 + r) * ld + j0 + c] = L[r][c];
  }
  // rows below: x L11^T = a  -> forward substitution over columns, one row per thread
  for (int r = j0 + kNB + t; r <= np; r += 256) {
    double x[kNB];
    double* row = A + (size_t)r * ld + j0;
#pragma unroll
    for (int c = 0; c < kNB; c++) x[c] = row[c];
#pragma unroll
    for (int c = 0; c < kNB; c++) {
      double s = x[c];
#pragma unroll
      for (int k = 0; k < c; k++) s -= x[k] * L[c][k];
      x[c] = s / L[c][c];
    }
#pragma unroll
    for (int c = 0; c < kNB; c++) row[c] = x[c];
  }
}

// trailing update: C[i][j] -= sum_k L[i][j0+k] L[j][j0+k] on 64x64 tiles of the lower triangle
// (rows up to and including the b row np).
__global__ __launch_bounds__(256) void chol_update_kernel(const GnState* __restrict__ st,
                                                          double* __restrict__ A, int np, int ld, int j0) {
  if (st->done) return;
  const int base = j0 + kNB;
  const int ti = blockIdx.y, tj = blockIdx.x;
  if (tj > ti) return;
  const int r0 = base + ti * 64, c0 = base + tj * 64;
  if (r0 > np || c0 >= np) return;
  __shared__ double Li[kNB][65];
  __shared__ double Lj[kNB][65];
  const int t = threadIdx.x;
  for (int k = t; k < 64 * kNB; k += 256) {
    const int r = k / kNB, c = k % kNB;
    Li[c][r] = (r0 + r <= np) ? A[(size_t)(r0 + r) * ld + j0 + c] : 0.0;
    Lj[c][r] = (c0 + r < np) ? A[(size_t)(c0 + r) * ld + j0 + c] : 0.0;
  }
  __syncthreads();
  const int tx = t & 15, ty = t >> 4;
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
#pragma unroll 8
  for (int k = 0; k < kNB; k++) {
    double li[4], lj[4];
#pragma unroll
    for (int a = 0; a < 4; a++) { li[a] = Li[k][ty * 4 + a]; lj[a] = Lj[k][tx * 4 + a]; }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) acc[a][b] = fma(li[a], lj[b], acc[a][b]);
  }
#pragma unroll
  for (int a = 0; a < 4; a++) {
    const int r = r0 + ty * 4 + a;
    if (r > np) continue;
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const int c = c0 + tx * 4 + b;
      if (c < np && c <= r) A[(size_t)r * ld + c] -= acc[a][b];
    }
  }
}

// back substitution L^T x = y (y = row np), dx = -x, retraction, convergence flag.
__global__ __launch_bounds__(256) void chol_backsolve_retract_kernel(GnState* __restrict__ st,
                                                                     double* __restrict__ A, int np,
                                                                     int ld, int N, int num_fix,
                                                                     float* __restrict__ Twc,
                                                                     float* __restrict__ dx,
                                                                     float delta_thresh) {
  if (st->done) return;
  const int t = threadIdx.x;
  const int n = N * 7;
  double* y = A + (size_t)np * ld;
  __shared__ double xb[kNB];
  __shared__ double Lb[kNB][kNB + 1];
  __shared__ float red[256];
  const bool fail = st->chol_fail != 0;
  if (!fail) {
    for (int j0 = np - kNB; j0 >= 0; j0 -= kNB) {
      for (int k = t; k < kNB * kNB; k += 256) Lb[k / kNB][k % kNB] = A[(size_t)(j0 + k / kNB) * ld + j0 + k % kNB];
      __syncthreads();
      if (t == 0) {
        for (int c = kNB - 1; c >= 0; c--) {
          double s = y[j0 + c];
          for (int k = c + 1; k < kNB; k++) s -= Lb[k][c] * xb[k];
          xb[c] = s / Lb[c][c];
        }
        for (int c = 0; c < kNB; c++) y[j0 + c] = xb[c];
      }
      __syncthreads();
      // y[i] -= sum_k L[j0+k][i] x[k]  for i < j0   (rows of L are contiguous in i: coalesced)
      for (int i = t; i < j0; i += 256) {
        double s = y[i];
#pragma unroll 8
        for (int k = 0; k < kNB; k++) s -= A[(size_t)(j0 + k) * ld + i] * xb[k];
        y[i] = s;
      }
      __syncthreads();
    }
  }
  float ss = 0.0f;
  for (int k = t; k < n; k += 256) {
    const float d = fail ? 0.0f : -(float)y[k];  // "NOTE: Accounting for negative here!" :1208-1209
    dx[k] = d;
    ss = fmaf(d, d, ss);
  }
  red[t] = ss;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) red[t] += red[t + s];
    __syncthreads();
  }
  if (!fail) {
    for (int k = num_fix + t; k < N + num_fix; k += 256) {
      const Sim3f T = sim3_load(Twc + 8 * k);
      sim3_store(Twc + 8 * k, sim3_retr(dx + 7 * (k - num_fix), T));
    }
  }
  if (t == 0) {
    const float nrm = sqrtf(red[0]);
    st->last_norm = nrm;
    st->iters += 1;
    st->chol_fail = 0;
    if (nrm < delta_thresh) st->done = 1;
  }
}

__global__ void gn_state_init_kernel(GnState* st) {
  st->done = 0;
  st->iters = 0;
  st->chol_fail = 0;
  st->last_norm = 0.0f;
}

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct GnWorkspace {
  GnState* st;
  int *ii_edge, *jj_edge, *ii_opt, *jj_opt, *first, *num_unique;
  float* econst;
  float* partial;
  float* Hs;
  float* gs;
  double* Haug;
  int S, chunk_len, np, ld;
  size_t bytes;
};

static GnWorkspace gn_carve(void* base, int P, int E, int HW) {
  GnWorkspace w;
  const int N = P > 1 ? P - 1 : 0;
  const int n = N * 7;
  w.np = (int)align_up((size_t)(n > 0 ? n : 1), kNB);
  w.ld = w.np;
  // split every edge over S workgroups: >= ~1k workgroups per launch, >= 16 points per thread
  int S = E > 0 ? (1024 + E - 1) / E : 1;
  const int s_cap = HW / (256 * 16) > 0 ? HW / (256 * 16) : 1;
  if (S > s_cap) S = s_cap;
  if (S < 1) S = 1;
  w.S = S;
  w.chunk_len = (int)align_up((size_t)((HW + S - 1) / S), 256);
  if (w.chunk_len < 256) w.chunk_len = 256;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (char*)base + o; };
  w.st = (GnState*)take(sizeof(GnState));
  w.ii_edge = (int*)take(sizeof(int) * (size_t)E);
  w.jj_edge = (int*)take(sizeof(int) * (size_t)E);
  w.ii_opt = (int*)take(sizeof(int) * (size_t)E);
  w.jj_opt = (int*)take(sizeof(int) * (size_t)E);
  w.first = (int*)take(sizeof(int) * 2 * (size_t)E);
  w.num_unique = (int*)take(sizeof(int));
  w.econst = (float*)take(sizeof(float) * (size_t)E * kEdgeConst);
  w.partial = (float*)take(sizeof(float) * (size_t)E * S * kAcc);
  w.Hs = (float*)take(sizeof(float) * (size_t)4 * E * 49);
  w.gs = (float*)take(sizeof(float) * (size_t)2 * E * 7);
  w.Haug = (double*)take(sizeof(double) * (size_t)(w.np + 1) * w.ld);
  w.bytes = off;
  return w;
}

static void fill_params(GnParams& P, int kind, const float* K_dev, float sigma_a, float sigma_b,
                        float C_thresh, float Q_thresh, int height, int width, int pixel_border,
                        float z_eps) {
  P.sa_inv = 1.0f / sigma_a;
  P.sb_inv = kind == 2 ? 0.0f : 1.0f / sigma_b;
  P.C_thresh = C_thresh;
  P.Q_thresh = Q_thresh;
  P.height = height;
  P.width = width > 0 ? width : 1;
  P.border_lo = (float)pixel_border;
  P.border_hi_u = (float)(width - 1 - pixel_border);
  P.border_hi_v = (float)(height - 1 - pixel_border);
  P.z_eps = z_eps;
  P.K = K_dev;
}

// edges [e0, e0+cnt) of the global edge list; idx/vm/Q are LOCAL arrays of cnt rows; Hs/gs are the
// global [4,E,7,7] / [2,E,7] buffers.
static int launch_accumulate(int kind, const GnWorkspace& w, const float* Twc, const float* Xs,
                             const float* Cs, const int64_t* idx, const uint8_t* vm, const float* Q, int HW,
                             int E, int e0, int cnt, const GnParams& P, float* Hs, float* gs, hipStream_t s) {
  if (cnt <= 0) return MSLAM_OK;
  hipLaunchKernelGGL(gn_edge_setup_kernel, dim3((cnt + 63) / 64), dim3(64), 0, s, w.st, Twc, w.ii_edge + e0,
                     w.jj_edge + e0, cnt, w.econst);
  dim3 grid(w.S, cnt);
  if (kind == 0)
    hipLaunchKernelGGL(gn_accum_kernel<0>, grid, dim3(256), 0, s, w.st, w.econst, Xs, Cs, w.ii_edge + e0,
                       w.jj_edge + e0, idx, vm, Q, HW, w.chunk_len, P, w.partial);
  else if (kind == 1)
    hipLaunchKernelGGL(gn_accum_kernel<1>, grid, dim3(256), 0, s, w.st, w.econst, Xs, Cs, w.ii_edge + e0,
                       w.jj_edge + e0, idx, vm, Q, HW, w.chunk_len, P, w.partial);
  else
    hipLaunchKernelGGL(gn_accum_kernel<2>, grid, dim3(256), 0, s, w.st, w.econst, Xs, Cs, w.ii_edge + e0,
                       w.jj_edge + e0, idx, vm, Q, HW, w.chunk_len, P, w.partial);
  hipLaunchKernelGGL(gn_reduce_kernel, dim3(cnt), dim3(64), 0, s, w.st, w.partial, w.S, Twc, w.ii_edge + e0, e0, E,
                     Hs, gs);
  return check_hip(hipGetLastError(), "gn accumulate launch");
}

static int launch_solve(const GnWorkspace& w, const float* Hs, const float* gs, int E, int P, float* Twc,
                        float* dx, float delta_thresh, hipStream_t s) {
  const int N = P - 1;
  const size_t shmem = sizeof(double) * 7 * (size_t)w.ld;
  MSLAM_REQUIRE(shmem <= 160 * 1024, "gauss_newton: %d poses exceed the assemble kernel's LDS row buffer", P);
  if (shmem > 64 * 1024) {
    int rc = check_hip(hipFuncSetAttribute((const void*)gn_assemble_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                       "hipFuncSetAttribute");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(gn_assemble_kernel, dim3(N + 2), dim3(256), shmem, s, w.st, Hs, gs, w.ii_opt, w.jj_opt, E, N,
                     w.np, w.ld, w.Haug);
  for (int j0 = 0; j0 < w.np; j0 += kNB) {
    hipLaunchKernelGGL(chol_panel_kernel, dim3(1), dim3(256), 0, s, w.st, w.Haug, w.np, w.ld, j0);
    const int rem = w.np + 1 - (j0 + kNB);  // rows below the panel, including the b row
    if (rem > 0) {
      const int tiles = (rem + 63) / 64;
      hipLaunchKernelGGL(chol_update_kernel, dim3(tiles, tiles), dim3(256), 0, s, w.st, w.Haug, w.np, w.ld, j0);
    }
  }
  hipLaunchKernelGGL(chol_backsolve_retract_kernel, dim3(1), dim3(256), 0, s, w.st, w.Haug, w.np, w.ld, N, 1, Twc,
                     dx, delta_thresh);
  return check_hip(hipGetLastError(), "gn solve launch");
}

static int gn_check_ws(const GnWorkspace& w, void* workspace, size_t workspace_bytes, const char* who) {
  if (!workspace || w.bytes > workspace_bytes) {
    set_error("%s: workspace too small (%zu < %zu)", who, workspace ? workspace_bytes : (size_t)0, w.bytes);
    return MSLAM_ENOMEM;
  }
  return MSLAM_OK;
}

}  // namespace mslam

using namespace mslam;

extern "C" size_t mslam_gn_workspace_bytes(int num_poses, int num_edges, int num_points) {
  if (num_poses < 0 || num_edges < 0 || num_points < 0) return 0;
  return gn_carve(nullptr, num_poses, num_edges, num_points).bytes;
}

extern "C" int mslam_gn_begin(const int64_t* ii, const int64_t* jj, int num_poses, int num_edges,
                              int num_points, void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(num_poses >= 2 && num_edges >= 1, "gn_begin: need >= 2 poses and >= 1 edge");
  MSLAM_REQUIRE(ii && jj, "gn_begin: null pointer");
  MSLAM_REQUIRE(num_edges <= 65535, "gn_begin: %d edges exceed the grid limit", num_edges);
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points);
  int rc = gn_check_ws(w, workspace, workspace_bytes, "gn_begin");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_state_init_kernel, dim3(1), dim3(1), 0, s, w.st);
  hipLaunchKernelGGL(gn_index_kernel, dim3(1), dim3(256), 0, s, ii, jj, num_edges, 1, w.ii_edge, w.jj_edge,
                     w.ii_opt, w.jj_opt, w.first, w.num_unique);
  return check_hip(hipGetLastError(), "gn_begin launch");
}

extern "C" int mslam_gn_accumulate(int kind, const float* Twc, const float* Xs, const float* Cs, const float* K,
                                   const int64_t* idx_ii2jj, const uint8_t* valid_match, const float* Q,
                                   int num_poses, int num_points, int num_edges, int edge_begin,
                                   int edge_count, float sigma_a, float sigma_b, float C_thresh,
                                   float Q_thresh, int height, int width, int pixel_border, float z_eps,
                                   float* Hs, float* gs, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  MSLAM_REQUIRE(kind >= 0 && kind <= 2, "gn_accumulate: kind must be 0 (rays), 1 (calib) or 2 (points)");
  MSLAM_REQUIRE(num_poses >= 2 && num_points >= 1 && num_edges >= 1, "gn_accumulate: bad sizes");
  MSLAM_REQUIRE(edge_begin >= 0 && edge_count >= 0 && edge_begin + edge_count <= num_edges,
                "gn_accumulate: edge range [%d,%d) outside [0,%d)", edge_begin, edge_begin + edge_count, num_edges);
  if (edge_count == 0) return MSLAM_OK;
  MSLAM_REQUIRE(Twc && Xs && Cs && idx_ii2jj && valid_match && Q && Hs && gs, "gn_accumulate: null pointer");
  MSLAM_REQUIRE(kind != 1 || (K && width > 0 && height > 0), "gn_accumulate: calib needs K, height, width");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points);
  int rc = gn_check_ws(w, workspace, workspace_bytes, "gn_accumulate");
  if (rc) return rc;
  GnParams P;
  fill_params(P, kind, K, sigma_a, sigma_b, C_thresh, Q_thresh, height, width, pixel_border, z_eps);
  return launch_accumulate(kind, w, Twc, Xs, Cs, idx_ii2jj, valid_match, Q, num_points, num_edges, edge_begin,
                           edge_count, P, Hs, gs, (hipStream_t)stream);
}

extern "C" int mslam_gn_solve_retract(const float* Hs, const float* gs, int num_poses, int num_edges,
                                      int num_points, float* Twc, float* dx, float delta_thresh,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(num_poses >= 2 && num_edges >= 1, "gn_solve_retract: need >= 2 poses and >= 1 edge");
  MSLAM_REQUIRE(Hs && gs && Twc && dx, "gn_solve_retract: null pointer");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points);
  int rc = gn_check_ws(w, workspace, workspace_bytes, "gn_solve_retract");
  if (rc) return rc;
  return launch_solve(w, Hs, gs, num_edges, num_poses, Twc, dx, delta_thresh, (hipStream_t)stream);
}

extern "C" int mslam_gn_status(int* status4, int num_poses, int num_edges, int num_points, void* workspace,
                               size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(status4, "gn_status: null pointer");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points);
  int rc = gn_check_ws(w, workspace, workspace_bytes, "gn_status");
  if (rc) return rc;
  return check_hip(hipMemcpyAsync(status4, w.st, sizeof(GnState), hipMemcpyDeviceToDevice, (hipStream_t)stream),
                   "gn_status copy");
}

static int gauss_newton_impl(int kind, float* Twc, const float* Xs, const float* Cs, const float* K,
                             const int64_t* ii, const int64_t* jj, const int64_t* idx_ii2jj,
                             const uint8_t* valid_match, const float* Q, int num_poses, int num_points,
                             int num_edges, float sigma_a, float sigma_b, float C_thresh, float Q_thresh,
                             int height, int width, int pixel_border, float z_eps, int max_iter,
                             float delta_thresh, float* dx, void* workspace, size_t workspace_bytes,
                             void* stream) {
  MSLAM_REQUIRE(num_poses >= 2, "gauss_newton: need at least 2 poses (got %d)", num_poses);
  MSLAM_REQUIRE(num_edges >= 1 && num_points >= 1, "gauss_newton: need at least one edge and one point");
  MSLAM_REQUIRE(Twc && Xs && Cs && ii && jj && idx_ii2jj && valid_match && Q && dx, "gauss_newton: null pointer");
  MSLAM_REQUIRE(kind != 1 || (K && width > 0 && height > 0), "gauss_newton_calib needs K, height, width");
  int rc = mslam_gn_begin(ii, jj, num_poses, num_edges, num_points, workspace, workspace_bytes, stream);
  if (rc) return rc;
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points);
  hipStream_t s = (hipStream_t)stream;
  GnParams P;
  fill_params(P, kind, K, sigma_a, sigma_b, C_thresh, Q_thresh, height, width, pixel_border, z_eps);
  rc = check_hip(hipMemsetAsync(dx, 0, sizeof(float) * 7 * (size_t)(num_poses - 1), s), "dx memset");
  if (rc) return rc;
  for (int it = 0; it < max_iter; it++) {
    rc = launch_accumulate(kind, w, Twc, Xs, Cs, idx_ii2jj, valid_match, Q, num_points, num_edges, 0, num_edges, P,
                           w.Hs, w.gs, s);
    if (rc) return rc;
    rc = launch_solve(w, w.Hs, w.gs, num_edges, num_poses, Twc, dx, delta_thresh, s);
    if (rc) return rc;
  }
  return MSLAM_OK;
}

extern "C" int mslam_gauss_newton_rays(float* Twc, const float* Xs, const float* Cs, const int64_t* ii,
                                       const int64_t* jj, const int64_t* idx_ii2jj, const uint8_t* valid_match,
                                       const float* Q, int num_poses, int num_points, int num_edges,
                                       float sigma_ray, float sigma_dist, float C_thresh, float Q_thresh,
                                       int max_iter, float delta_thresh, float* dx, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return gauss_newton_impl(0, Twc, Xs, Cs, nullptr, ii, jj, idx_ii2jj, valid_match, Q, num_poses, num_points,
                           num_edges, sigma_ray, sigma_dist, C_thresh, Q_thresh, 0, 0, 0, 0.0f, max_iter,
                           delta_thresh, dx, workspace, workspace_bytes, stream);
}

extern "C" int mslam_gauss_newton_calib(float* Twc, const float* Xs, const float* Cs, const float* K,
                                        const int64_t* ii, const int64_t* jj, const int64_t* idx_ii2jj,
                                        const uint8_t* valid_match, const float* Q, int num_poses,
                                        int num_points, int num_edges, int height, int width, int pixel_border,
                                        float z_eps, float sigma_pixel, float sigma_depth, float C_thresh,
                                        float Q_thresh, int max_iter, float delta_thresh, float* dx,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  return gauss_newton_impl(1, Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, num_poses, num_points,
                           num_edges, sigma_pixel, sigma_depth, C_thresh, Q_thresh, height, width, pixel_border,
                           z_eps, max_iter, delta_thresh, dx, workspace, workspace_bytes, stream);
}

extern "C" int mslam_gauss_newton_points(float* Twc, const float* Xs, const float* Cs, const int64_t* ii,
                                         const int64_t* jj, const int64_t* idx_ii2jj,
                                         const uint8_t* valid_match, const float* Q, int num_poses,
                                         int num_points, int num_edges, float sigma_point, float C_thresh,
                                         float Q_thresh, int max_iter, float delta_thresh, float* dx,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return gauss_newton_impl(2, Twc, Xs, Cs, nullptr, ii, jj, idx_ii2jj, valid_match, Q, num_poses, num_points,
                           num_edges, sigma_point, 1.0f, C_thresh, Q_thresh, 0, 0, 0, 0.0f, max_iter,
                           delta_thresh, dx, workspace, workspace_bytes, stream);
}

// ---- lietorch-surface Sim3 ops on arrays of poses (forward only) ------------------------------
namespace mslam {
__global__ void sim3_act_kernel(const float* __restrict__ T, const float* __restrict__ X, float* __restrict__ Y,
                                long long n_pts_per_pose, long long total, int broadcast_pose) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long pi = broadcast_pose ? 0 : i / n_pts_per_pose;
  const Sim3f P = sim3_load(T + 8 * pi);
  const float x[3] = {X[i * 3], X[i * 3 + 1], X[i * 3 + 2]};
  float y[3];
  sim3_act(P, x, y);
  Y[i * 3] = y[0]; Y[i * 3 + 1] = y[1]; Y[i * 3 + 2] = y[2];
}

__global__ void sim3_unary_kernel(int op, const float* __restrict__ A, const float* __restrict__ B,
                                  float* __restrict__ O, int n, int bcast_a, int bcast_b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Sim3f R;
  if (op == 0) {  // inv
    R = sim3_inv(sim3_load(A + 8 * (bcast_a ? 0 : i)));
  } else if (op == 1) {  // mul
    R = sim3_mul(sim3_load(A + 8 * (bcast_a ? 0 : i)), sim3_load(B + 8 * (bcast_b ? 0 : i)));
  } else if (op == 2) {  // exp (A is xi[n,7])
    R = sim3_exp(A + 7 * (bcast_a ? 0 : i));
  } else {  // retr: exp(A=xi) * B
    R = sim3_retr(A + 7 * (bcast_a ? 0 : i), sim3_load(B + 8 * (bcast_b ? 0 : i)));
  }
  sim3_store(O + 8 * i, R);
}
}  // namespace mslam

extern "C" int mslam_sim3_act(const float* T, const float* X, float* Y, int num_poses, long long pts_per_pose,
                              int broadcast_pose, void* stream) {
  const long long total = (long long)num_poses * pts_per_pose;
  if (total <= 0) return MSLAM_OK;
  MSLAM_REQUIRE(T && X && Y, "sim3_act: null pointer");
  hipLaunchKernelGGL(sim3_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T,
                     X, Y, pts_per_pose, total, broadcast_pose);
  MSLAM_LAUNCH_CHECK("sim3_act");
  return MSLAM_OK;
}

extern "C" int mslam_sim3_op(int op, const float* A, const float* B, float* out, int n, int bcast_a, int bcast_b,
                             void* stream) {
  MSLAM_REQUIRE(op >= 0 && op <= 3, "sim3_op: op must be 0 inv, 1 mul, 2 exp, 3 retr");
  if (n <= 0) return MSLAM_OK;
  MSLAM_REQUIRE(A && out && (op == 0 || op == 2 || B), "sim3_op: null pointer");
  hipLaunchKernelGGL(sim3_unary_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, op, A, B, out, n,
                     bcast_a, bcast_b);
  MSLAM_LAUNCH_CHECK("sim3_op");
  return MSLAM_OK;
}
