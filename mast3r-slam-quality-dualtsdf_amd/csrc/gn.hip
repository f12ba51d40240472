// Gauss-Newton backend for gfx950: per-edge residual/JtJ accumulation (rays | calib | points),
// fp64 normal-equation assembly, blocked dense fp64 Cholesky on the f64 matrix cores, Sim3
// retraction - all on the device, no host round trip and no host synchronisation inside the loop.
//
// Reference behaviour reproduced (re-designed, not translated):
//   ray_align_kernel / calib_proj_kernel / point_align_kernel
//        mast3r_slam/backend/src/gn_kernels.cu:813-1138, 1231-1543, 455-723
//   host loop + SparseBlock (Eigen SimplicialLLT, fp64)   gn_kernels.cu:57-159, 1140-1228
//   get_unique_kf_idx / create_inds                        gn_kernels.cu:161-170
//   pose_retr_kernel                                        gn_kernels.cu:415-453
//
// MI355X-first design decisions (DESIGN.md "GN"):
//   * apply_Sim3_adj_inv is linear: Jj = M_i x with a per-edge 7x7 matrix M_i, and Ji = -Jj.
//     So the 14x14 block Hessian is [[A,-A],[-A,A]] with A = M (sum w x x^T) M^T.  The streaming
//     kernel accumulates only B = sum w x x^T (28 values) and u = sum w e x (7 values) of the RAW
//     (pre-adjoint) Jacobian rows - 35 accumulators per lane instead of 119, and no quaternion
//     algebra per residual row.  M is applied once per edge, in fp64, in the reduce kernel.
//   * the match data of an edge (idx, valid, Q) and the pointmaps / confidences do not change
//     during the <= 10 iterations of one call - only the poses do.  gn_compact therefore resolves
//     the gather Xi[idx], applies the pose-independent gates (valid_match, Q > Q_thresh,
//     C > C_thresh) ONCE and writes the surviving points as a dense struct-of-arrays stream
//     (28 B per point for rays/points, 32 B for calib); every iteration then streams that with
//     16-B loads and no gather instead of re-reading 45 B per point through an index
//     (HBM is plentiful on this part: 288 GB buys the stream for thousands of edges).
//   * every edge is split over several workgroups (>= 1k workgroups per launch), wave64 shuffle
//     reduction + one LDS pass, partials combined in a fixed order (deterministic, no atomics).
//   * the fp64 solve never leaves the GPU and has no size cap: assemble (one wave per pose
//     block-row) -> right-looking blocked LL^T on the augmented matrix [H | b] (forward
//     substitution comes for free), ONE launch per panel: the trailing update runs on
//     v_mfma_f64_16x16x4_f64 in 64x64 tiles and the workgroups that own the next panel's columns
//     factor its diagonal block (redundantly, in LDS) and solve their rows of it in the same
//     launch -> multi-workgroup blocked back substitution -> dx = -x -> retraction -> ||dx|| test
//     sets a device-side `done` flag that later iterations' kernels read and exit on (replaces
//     delta_norm.item(), gn_kernels.cu:1219-1222).
#include "common.h"
#include "sim3.h"

namespace mslam {

constexpr int kAcc = 35;        // 28 (lower triangle of 7x7) + 7
constexpr int kEdgeConst = 16;  // sR_ij (9) + t_ij (3) + pad
constexpr int kPlanes = 8;      // compact stream: xi(3) xj(3) sqrt(q) ind
constexpr int kTile = 64;       // Cholesky trailing-update tile

struct GnState {
  int done;        // ||dx|| < delta_thresh reached: later launches are no-ops
  int iters;       // GN iterations actually executed
  int chol_fail;   // current factorisation hit a non-positive pivot (Eigen: info() != Success)
  float last_norm;
};

// ---------------------------------------------------------------------------------------------
// index preparation: unique(sorted) + searchsorted by brute force, one thread per entry
// (2E entries, each scanned against all 2E: L2-resident, two short launches per GN call)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t ij_val(const int64_t* ii, const int64_t* jj, int E, int k) {
  return k < E ? ii[k] : jj[k - E];
}

__global__ __launch_bounds__(256) void gn_first_kernel(const int64_t* __restrict__ ii,
                                                       const int64_t* __restrict__ jj, int E,
                                                       int* __restrict__ first) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= 2 * E) return;
  const int64_t v = ij_val(ii, jj, E, k);
  int f = 1;
  for (int m = 0; m < k; m++)
    if (ij_val(ii, jj, E, m) == v) { f = 0; break; }
  first[k] = f;
}

__global__ __launch_bounds__(256) void gn_rank_kernel(const int64_t* __restrict__ ii,
                                                      const int64_t* __restrict__ jj, int E, int num_fix,
                                                      const int* __restrict__ first,
                                                      int* __restrict__ ii_edge, int* __restrict__ jj_edge,
                                                      int* __restrict__ ii_opt, int* __restrict__ jj_opt) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= 2 * E) return;
  const int64_t v = ij_val(ii, jj, E, k);
  int rank = 0;  // number of DISTINCT values smaller than v  (== searchsorted into unique())
  for (int m = 0; m < 2 * E; m++) rank += (first[m] && ij_val(ii, jj, E, m) < v) ? 1 : 0;
  if (k < E) { ii_edge[k] = rank; ii_opt[k] = rank - num_fix; }
  else { jj_edge[k - E] = rank; jj_opt[k - E] = rank - num_fix; }
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
// compaction: one pass over the call's constant inputs
// ---------------------------------------------------------------------------------------------
// Workgroup (chunk, local edge) walks its chunk of the edge's points in order, 256 at a time, and
// appends the points that pass the pose-independent gates to its slot of the stream (order
// preserved: deterministic).  Slot layout: kPlanes planes of chunk_len floats.
__global__ __launch_bounds__(256) void gn_compact_kernel(
    const float* __restrict__ Xs, const float* __restrict__ Cs, const int* __restrict__ ii_edge,
    const int* __restrict__ jj_edge, const int64_t* __restrict__ idx_ii2jj,
    const uint8_t* __restrict__ valid_match, const float* __restrict__ Q, int num_points, int chunk_len,
    float C_thresh, float Q_thresh, float* __restrict__ stream, int* __restrict__ counts, int S, int slot_begin) {
  // one-dimensional grid of S x edges workgroups (a y dimension would cap the edge count at 65 535)
  const int e = blockIdx.x / S, chunk = blockIdx.x - e * S;
  const int ix = ii_edge[e], jx = jj_edge[e];
  const float* __restrict__ Xi_base = Xs + (size_t)ix * num_points * 3;
  const float* __restrict__ Xj_base = Xs + (size_t)jx * num_points * 3;
  const float* __restrict__ Ci_base = Cs + (size_t)ix * num_points;
  const float* __restrict__ Cj_base = Cs + (size_t)jx * num_points;
  const int64_t* __restrict__ idx_e = idx_ii2jj + (size_t)e * num_points;
  const uint8_t* __restrict__ vm_e = valid_match + (size_t)e * num_points;
  const float* __restrict__ Q_e = Q + (size_t)e * num_points;
  // the edge's slot of the stream: edges of one accumulate range may be compacted by several calls (slot_begin)
  float* __restrict__ slot = stream + ((size_t)(slot_begin + e) * S + chunk) * (size_t)chunk_len * kPlanes;
  __shared__ int wave_cnt[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int k_begin = chunk * chunk_len;
  const int k_end = min(k_begin + chunk_len, num_points);
  int base = 0;  // points emitted so far (uniform)
  for (int k0 = k_begin; k0 < k_end; k0 += 256) {
    const int k = k0 + (int)threadIdx.x;
    bool valid = false;
    long long ind = 0;
    float q = 0.0f;
    if (k < k_end) {
      const bool vm = vm_e[k] != 0;
      ind = vm ? idx_e[k] : 0;  // invalid matches read index 0 (gn_kernels.cu:914)
      q = Q_e[k];
      const float ci = Ci_base[ind], cj = Cj_base[k];
      valid = vm & (q > Q_thresh) & (ci > C_thresh) & (cj > C_thresh);
    }
    const unsigned long long m = __ballot(valid);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wid; w++) off += wave_cnt[w];
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    if (valid) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
      slot[0 * (size_t)chunk_len + pos] = Xi_base[ind * 3 + 0];
      slot[1 * (size_t)chunk_len + pos] = Xi_base[ind * 3 + 1];
      slot[2 * (size_t)chunk_len + pos] = Xi_base[ind * 3 + 2];
      slot[3 * (size_t)chunk_len + pos] = Xj_base[(size_t)k * 3 + 0];
      slot[4 * (size_t)chunk_len + pos] = Xj_base[(size_t)k * 3 + 1];
      slot[5 * (size_t)chunk_len + pos] = Xj_base[(size_t)k * 3 + 2];
      slot[6 * (size_t)chunk_len + pos] = sqrtf(q);
      slot[7 * (size_t)chunk_len + pos] = __int_as_float((int)ind);
    }
    base += total;
    __syncthreads();  // wave_cnt is rewritten by the next round
  }
  if (threadIdx.x == 0) counts[(slot_begin + e) * S + chunk] = base;
}

// ---------------------------------------------------------------------------------------------
// streaming accumulation
// ---------------------------------------------------------------------------------------------
// v_rcp_f32 / v_rsq_f32 (1 ulp) instead of the correctly rounded division / square root the translation
// unit is otherwise built with: the blocks are compared at rel-L2 1e-5, and an IEEE division costs ten
// instructions in a loop whose arithmetic is as long as its memory time.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

__device__ __forceinline__ float huber_w(float r) {
  return fminf(1.0f, 1.345f * fast_rcp(fabsf(r)));  // == (a < 1.345 ? 1 : 1.345 / a), branch-free
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

struct EdgeConst {
  float r00, r01, r02, r10, r11, r12, r20, r21, r22, t0, t1, t2;
};

// one surviving point: the reference's residual rows and raw Jacobian rows (gn_kernels.cu:905-1010
// rays, :1320-1420 calib, :540-600 points), weights = huber(sqrt_info * r) * info
template <int KIND>
__device__ __forceinline__ void gn_point(float (&acc)[kAcc], const EdgeConst& c, const GnParams& P, float xi0,
                                         float xi1, float xi2, float xj0, float xj1, float xj2, float sq,
                                         int ind) {
  // X_j in frame i
  const float p0 = fmaf(c.r00, xj0, fmaf(c.r01, xj1, fmaf(c.r02, xj2, c.t0)));
  const float p1 = fmaf(c.r10, xj0, fmaf(c.r11, xj1, fmaf(c.r12, xj2, c.t1)));
  const float p2 = fmaf(c.r20, xj0, fmaf(c.r21, xj1, fmaf(c.r22, xj2, c.t2)));
  if constexpr (KIND == 0) {
    const float n2i = fmaf(xi2, xi2, fmaf(xi1, xi1, xi0 * xi0));
    const float n1i_inv = fast_rsq(n2i);
    const float n1i = n2i * n1i_inv;
    const float n2j = fmaf(p2, p2, fmaf(p1, p1, p0 * p0));
    const float n1j_inv = fast_rsq(n2j);
    const float n1j = n2j * n1j_inv;
    const float rj0 = p0 * n1j_inv, rj1 = p1 * n1j_inv, rj2 = p2 * n1j_inv;
    const float e0 = rj0 - xi0 * n1i_inv, e1 = rj1 - xi1 * n1i_inv, e2 = rj2 - xi2 * n1i_inv;
    const float e3 = n1j - n1i;
    const float swr = P.sa_inv * sq;
    const float swd = P.sb_inv * sq;
    const float wr = swr * swr, wd = swd * swd;
    const float w0 = huber_w(swr * e0) * wr, w1 = huber_w(swr * e1) * wr, w2 = huber_w(swr * e2) * wr;
    const float w3 = huber_w(swd * e3) * wd;
    const float n3 = n1j_inv * n1j_inv * n1j_inv;
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
    const int u_t = ind % P.width, v_t = ind / P.width;
    const bool valid_z = (p2 > P.z_eps) && (xi2 > P.z_eps);
    const float zinv = valid_z ? fast_rcp(p2) : 0.0f;
    const float zj_log = valid_z ? logf(p2) : 0.0f;
    const float zi_log = valid_z ? logf(xi2) : 0.0f;
    const float xz = p0 * zinv, yz = p1 * zinv;
    const float u = fmaf(Pfx, xz, Pcx), v = fmaf(Pfy, yz, Pcy);
    const bool valid_u = (u > P.border_lo) && (u < P.border_hi_u);
    const bool valid_v = (v > P.border_lo) && (v < P.border_hi_v);
    const bool valid = valid_u & valid_v & valid_z;  // the pose-dependent gates stay per iteration
    const float e0 = u - (float)u_t, e1 = v - (float)v_t, e2 = zj_log - zi_log;
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
    const float swp = P.sa_inv * sq;
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

template <int KIND>  // 0 rays, 1 calib, 2 points
__global__ __launch_bounds__(256) void gn_accum_kernel(const GnState* __restrict__ st,
                                                       const float* __restrict__ econst,
                                                       const float* __restrict__ stream,
                                                       const int* __restrict__ counts, int chunk_len,
                                                       GnParams P, float* __restrict__ partial, int S) {
  if (st->done) return;
  const int e = blockIdx.x / S, chunk = blockIdx.x - e * S;     // one-dimensional grid: no 65 535-edge cap
  // wave-uniform per-edge constants -> scalar loads
  const float* cc = econst + (size_t)e * kEdgeConst;
  const EdgeConst c = {cc[0], cc[1], cc[2], cc[3], cc[4], cc[5], cc[6], cc[7], cc[8], cc[9], cc[10], cc[11]};
  const float* __restrict__ slot = stream + ((size_t)e * S + chunk) * (size_t)chunk_len * kPlanes;
  const int n = counts[e * S + chunk];

  float acc[kAcc];
#pragma unroll
  for (int l = 0; l < kAcc; l++) acc[l] = 0.0f;

  // four consecutive points per lane and round: one 16-B load per plane, no per-point predicate in the
  // main loop (a predicate makes the compiler sink the loads into four branches and split them)
  const int n4 = n & ~3;
  for (int k = (int)threadIdx.x * 4; k < n4; k += 1024) {
    float f[kPlanes][4];
#pragma unroll
    for (int p = 0; p < kPlanes; p++) {
      if (p < (KIND == 1 ? 8 : 7)) {
        const float4 v = *reinterpret_cast<const float4*>(slot + (size_t)p * chunk_len + k);
        f[p][0] = v.x; f[p][1] = v.y; f[p][2] = v.z; f[p][3] = v.w;
      } else {
        f[p][0] = f[p][1] = f[p][2] = f[p][3] = 0.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
      gn_point<KIND>(acc, c, P, f[0][j], f[1][j], f[2][j], f[3][j], f[4][j], f[5][j], f[6][j],
                     KIND == 1 ? __float_as_int(f[7][j]) : 0);
  }
  if ((int)threadIdx.x < n - n4) {  // the <= 3 points behind the last full group
    const int k = n4 + (int)threadIdx.x;
    const float* q = slot + k;
    gn_point<KIND>(acc, c, P, q[0], q[(size_t)chunk_len], q[2 * (size_t)chunk_len], q[3 * (size_t)chunk_len],
                   q[4 * (size_t)chunk_len], q[5 * (size_t)chunk_len], q[6 * (size_t)chunk_len],
                   KIND == 1 ? __float_as_int(q[7 * (size_t)chunk_len]) : 0);
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
// row np = b^T.  Haug is zeroed by the caller; ONE WAVE per pose block-row visits the edges in the
// reference's triplet order (update_lhs / update_rhs, gn_kernels.cu:71-113, 1201-1206): 64 edges are
// tested per step with a ballot, the matching 7x7 blocks are summed per column block in an LDS table
// (first kTab distinct columns of the row; further ones fall back to read-modify-write of the row,
// which this wave owns), the rhs block in registers.
// ---------------------------------------------------------------------------------------------
constexpr int kTab = 32;

__global__ __launch_bounds__(64) void gn_assemble_kernel(const GnState* __restrict__ st,
                                                         const float* __restrict__ Hs,
                                                         const float* __restrict__ gs,
                                                         const int* __restrict__ ii_opt,
                                                         const int* __restrict__ jj_opt, int E, int N,
                                                         int np, int ld, double* __restrict__ Haug,
                                                         int* __restrict__ tmin32) {
  if (st->done) return;
  const int br = blockIdx.x;  // block row in [0, N); N = identity padding rows
  const int lane = threadIdx.x;
  const int n = N * 7;
  if (br == N) {
    for (int r = n + lane; r < np; r += 64) {
      Haug[(size_t)r * ld + r] = 1.0;
      atomicMin(&tmin32[r >> 5], r);          // a padding row holds its diagonal only
    }
    if (lane == 0) tmin32[np >> 5] = 0;        // the rhs row is dense
    return;
  }
  int cmin = br;                               // smallest unpinned column block of this block row (wave-uniform)
  __shared__ double tab[kTab][49];
  __shared__ int tabcol[kTab];
  int ntab = 0;
  double bacc = 0.0;  // lanes 0..6: rhs block of this pose
  const int k7 = lane / 7, l7 = lane % 7;
  for (int blk = 0; blk < 4; blk++) {
    const int* rix = (blk < 2) ? ii_opt : jj_opt;
    const int* cix = (blk & 1) ? jj_opt : ii_opt;
    for (int e0 = 0; e0 < E; e0 += 64) {
      const int e = e0 + lane;
      const bool row_match = e < E && rix[e] == br;
      const int cj_l = row_match ? cix[e] : -1;
      unsigned long long m = __ballot(row_match);
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int ee = e0 + b;
        if ((blk == 0 || blk == 2) && lane < 7)  // rhs: gs[0] rows of ii == br, then gs[1] rows of jj == br
          bacc += (double)gs[((size_t)(blk >> 1) * E + ee) * 7 + lane];
        const int cj = __shfl(cj_l, b, 64);
        if (cj < 0) continue;  // pinned column
        cmin = min(cmin, cj);
        const double v = lane < 49 ? (double)Hs[((size_t)blk * E + ee) * 49 + lane] : 0.0;
        const unsigned long long hit = __ballot(lane < ntab && tabcol[lane] == cj);
        int slot = hit ? __ffsll((long long)hit) - 1 : -1;
        if (slot < 0 && ntab < kTab) {
          slot = ntab++;
          if (lane == 0) tabcol[slot] = cj;
          if (lane < 49) tab[slot][lane] = 0.0;
          __builtin_amdgcn_wave_barrier();
        }
        if (lane < 49) {
          if (slot >= 0) tab[slot][lane] += v;
          else Haug[(size_t)(7 * br + k7) * ld + 7 * cj + l7] += v;
        }
      }
    }
  }
  __syncthreads();
  for (int s = 0; s < ntab; s++) {
    const int cj = tabcol[s];
    if (lane < 49) Haug[(size_t)(7 * br + k7) * ld + 7 * cj + l7] += tab[s][lane];
  }
  if (lane < 7) Haug[(size_t)np * ld + 7 * br + lane] = bacc;
  if (lane < 7) atomicMin(&tmin32[(7 * br + lane) >> 5], 7 * cmin);
}

// ---------------------------------------------------------------------------------------------
// blocked right-looking LL^T, one launch per panel
// ---------------------------------------------------------------------------------------------
// chol_step<NB>(j0): panel j0 (columns j0..j0+NB, rows below its diagonal block) is final.  Every 64x64
// tile of the lower triangle of the trailing matrix (rows j1..np INCLUDING the rhs row np, columns
// j1..np-1, j1 = j0 + NB) gets the rank-NB update A -= L_i L_j^T on the f64 matrix cores (the tile is
// loaded straight into the accumulators, the A operand is negated).  The tiles of the first tile
// column additionally finish the NEXT panel (columns j1..j1+NB): each of them computes the updated
// diagonal block D = A11 - Lp Lp^T itself (same code in every workgroup: identical bits), factors it
// in LDS and solves its own rows X L11^T = A21'.  Nobody writes A11 in this launch (the other
// workgroups read it): the factor of the diagonal block goes to the side array Ldiag[panel].
// update == 0: no trailing update, only the panel at j1 = j0 + NB is finished (the prologue, j0 = -NB, and the first
// panel of every outer block of the two-level schedule below).  clim: columns >= clim are left alone (they receive
// this panel's update later, as part of their outer block's rank-W update, chol_outer_kernel).
// Envelope: with the keyframes in temporal order H is a band (consecutive + recent-neighbour edges) plus a few far
// rows (loop closures), and LL^T fill stays inside each row's envelope [first non-zero column, diagonal].  tmin32[b] =
// the smallest first column over the rows [32 b, 32 b + 32) (written by gn_assemble).  A tile whose row block or
// column block starts its envelope behind the panel has a zero panel operand: its workgroup leaves at once (the
// reference hands the same structure to a sparse LL^T, gn_kernels.cu:57-159).
// Every global load of a workgroup is issued in ONE phase at its start (the launch is latency-bound:
// 28 dependent launches per factorisation at 125 keyframes).
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <int NB>
__global__ __launch_bounds__(256) void chol_step_kernel(GnState* __restrict__ st, double* __restrict__ A,
                                                        double* __restrict__ Ldiag, int np, int ld, int j0,
                                                        const int* __restrict__ tmin32, int update, int clim) {
  if (st->done) return;
  constexpr int LS = NB + 2;   // LDS row stride of the panel rows: conflict-free f64 MFMA operand reads
  constexpr int TS = kTile + 1;
  constexpr int DS = NB + 1;
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  const int j1 = j0 + NB;
  const int r0 = j1 + ti * kTile, c0 = j1 + tj * kTile;
  if (r0 > np || c0 >= np || c0 >= clim) return;
  const bool has_update = update != 0;
  {
    const int last = np >> 5;                  // r0, c0 are multiples of 32; a tile covers two 32-row blocks
    const int fr = min(tmin32[min(r0 >> 5, last)], tmin32[min((r0 >> 5) + 1, last)]);
    const int fc = min(tmin32[min(c0 >> 5, last)], tmin32[min((c0 >> 5) + 1, last)]);
    if (tj != 0 && (fr >= j1 || fc >= j1)) return;   // L_i or L_j of this panel is zero: nothing to subtract
    if (tj == 0 && fr >= j1 + NB) return;            // ... and nothing in the next panel's columns of these rows either
  }
  extern __shared__ double sh[];
  double* Li = sh;                        // [64][LS] rows r0.. of panel j0
  double* Lj = sh + kTile * LS;           // [64][LS] rows c0.. of panel j0 (first tile column: rows j1.. = Lp)
  double* T = sh;                         // [64][TS] aliases Li/Lj after the MFMA loop (first tile column)
  double* D = sh + 2 * kTile * LS;        // [NB][DS]
  double* dsave = D + NB * DS;            // [NB] sqrt of the pivots
  double* rdiag = dsave + NB;             // [NB] their reciprocals
  __shared__ int fail;
  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int fcol = lane & 15, frow = lane >> 4;

  // ---- one load phase: the tile into the accumulators, panel rows (and A11) into LDS ----
  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int r = r0 + 32 * wr + 16 * m + frow + 4 * g;
        const int c = c0 + 32 * wc + 16 * n + fcol;
        const bool in = r <= np && c < np && c <= r && c < clim;
        const double v = A[in ? (size_t)r * ld + c : (size_t)0];  // always-valid address + select: no branch
        acc[m][n][g] = in ? v : 0.0;
      }
  if (has_update) {
    for (int k = t; k < kTile * NB; k += 256) {
      const int r = k / NB, q = k % NB;
      const double vi = A[(size_t)min(r0 + r, np) * ld + j0 + q];
      const double vj = A[(size_t)min(c0 + r, np) * ld + j0 + q];
      Li[r * LS + q] = (r0 + r <= np) ? vi : 0.0;
      Lj[r * LS + q] = (c0 + r < np) ? vj : 0.0;
    }
  }
  if (tj == 0) {
    for (int k = t; k < NB * NB; k += 256) {
      const int a = k / NB, b = k % NB;
      D[a * DS + b] = A[(size_t)(j1 + a) * ld + j1 + b];
    }
    if (t == 0) fail = 0;
  }
  __syncthreads();

  if (has_update) {
    // first tile column: D = A11 - Lp Lp^T with Lp = rows j1..j1+NB of the panel = the first NB rows of Lj
    if (tj == 0) {
      for (int k = t; k < NB * NB; k += 256) {
        const int a = k / NB, b = k % NB;
        if (b <= a) {
          double s = 0.0;
#pragma unroll 8
          for (int q = 0; q < NB; q++) s = fma(Lj[a * LS + q], Lj[b * LS + q], s);
          D[a * DS + b] -= s;
        }
      }
    }
    // rank-NB update of the tile on the matrix cores: acc += (-Li) Lj^T
    const int fi = lane & 15, fk = lane >> 4;
#pragma unroll 4
    for (int k4 = 0; k4 < NB; k4 += 4) {
      double a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        a[m] = -Li[(32 * wr + 16 * m + fi) * LS + k4 + fk];
        b[m] = Lj[(32 * wc + 16 * m + fi) * LS + k4 + fk];
      }
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
  }

  if (tj != 0) {
    if (has_update) {
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
          for (int g = 0; g < 4; g++) {
            const int r = r0 + 32 * wr + 16 * m + frow + 4 * g;
            const int c = c0 + 32 * wc + 16 * n + fcol;
            if (r <= np && c < np && c <= r && c < clim) A[(size_t)r * ld + c] = acc[m][n][g];
          }
    }
    return;
  }

  // ---- first tile column: panel columns into T, columns beyond the panel straight back ----
  __syncthreads();  // all fragment reads of Li / Lj and the D update are done: T may alias them
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int rr = 32 * wr + 16 * m + frow + 4 * g;
        const int cc = 32 * wc + 16 * n + fcol;
        const int r = r0 + rr, c = c0 + cc;
        if (cc < NB) T[rr * TS + cc] = acc[m][n][g];
        else if (has_update && r <= np && c < np && c <= r && c < clim) A[(size_t)r * ld + c] = acc[m][n][g];
      }

  // ---- factor D in LDS (every workgroup of the column: same bits), two barriers per pivot ----
  __syncthreads();
  for (int j = 0; j < NB; j++) {
    const double djj = D[j * DS + j];
    const bool ok = djj > 0.0;
    double d, rinv;
    if (djj > 1e-30 && djj < 1e30) {
      // 1/sqrt by Newton from the f32 seed (three steps: 24 -> 48 -> 53+ bits); the library sqrt + division
      // are ~400 cycles of dependent latency per pivot on the critical path of every launch
      double r = (double)__builtin_amdgcn_rsqf((float)djj);
      const double h = 0.5 * djj;
      r = r * (1.5 - h * r * r);
      r = r * (1.5 - h * r * r);
      r = r * (1.5 - h * r * r);
      rinv = r;
      d = djj * r;
    } else {
      d = ok ? sqrt(djj) : 1.0;
      rinv = 1.0 / d;
    }
    if (t > j && t < NB) D[t * DS + j] *= rinv;
    if (t == j) {
      dsave[j] = d;
      rdiag[j] = rinv;
      if (!ok) fail = 1;
    }
    __syncthreads();
    // rank-1 update of the trailing lower triangle; (row, column) from the element index by shifts
#pragma unroll
    for (int i = 0; i < NB * NB / 256; i++) {
      const int k = t + 256 * i;
      const int r = k / NB, c = k % NB;
      if (c > j && c <= r) D[r * DS + c] -= D[r * DS + j] * D[c * DS + j];
    }
    __syncthreads();
  }
  if (ti == 0) {
    if (fail && t == 0) st->chol_fail = 1;
    double* Ld = Ldiag + (size_t)(j1 / NB) * NB * NB;
    for (int k = t; k < NB * NB; k += 256) {
      const int a = k / NB, b = k % NB;
      Ld[k] = (b < a) ? D[a * DS + b] : (b == a ? dsave[a] : 0.0);
    }
  }

  // ---- rows of the panel below its diagonal block: x L11^T = t, one row per lane of wave 0; the result
  // goes back through T so that the global stores are whole 256-B row segments ----
  if (t < kTile) {
    const int r = r0 + t;
    const bool in_diag = (ti == 0) && (t < NB);
    if (r <= np && !in_diag) {
      double x[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) x[c] = T[t * TS + c];
#pragma unroll
      for (int c = 0; c < NB; c++) {
        double s = x[c];
#pragma unroll
        for (int k = 0; k < c; k++) s -= x[k] * D[c * DS + k];
        x[c] = s * rdiag[c];
      }
#pragma unroll
      for (int c = 0; c < NB; c++) T[t * TS + c] = x[c];
    }
  }
  __syncthreads();
  for (int k = t; k < kTile * NB; k += 256) {
    const int rr = k / NB, c = k % NB;
    const int r = r0 + rr;
    const bool in_diag = (ti == 0) && (rr < NB);
    if (r <= np && !in_diag) A[(size_t)r * ld + j1 + c] = T[rr * TS + c];
  }
}

// Two-level schedule for large systems (config 5: 8 743 unknowns).  The one-level loop above streams the whole trailing
// matrix once per 32-wide panel: n^3 / (6 NB) x 16 B = 56 GB per factorisation at n = 8 743, i.e. HBM-bound.  Panels are
// therefore grouped into outer blocks of W columns: inside a block the panel steps update only the block's own columns
// (clim), and ONE launch then applies the block's rank-W update to everything behind it - a quarter of the traffic at
// W = 128, and f64-MFMA work of useful depth (W / 4 k-steps per tile instead of 8).
// chol_outer(J0, W): A[r, c] -= L[r, J0:J0+W] L[c, J0:J0+W]^T for every 64x64 tile of the lower triangle with c >= J0 + W
// (rows up to and including the rhs row np).  Envelope skip as in chol_step.
__global__ __launch_bounds__(256) void chol_outer_kernel(const GnState* __restrict__ st, double* __restrict__ A, int np,
                                                         int ld, int J0, int W, const int* __restrict__ tmin32) {
  if (st->done) return;
  constexpr int KC = 32, LS = KC + 2;
  const int tj = blockIdx.x, ti = blockIdx.y;
  if (tj > ti) return;
  const int C0 = J0 + W;
  const int r0 = C0 + ti * kTile, c0 = C0 + tj * kTile;
  if (r0 > np || c0 >= np) return;
  {
    const int last = np >> 5;
    const int fr = min(tmin32[min(r0 >> 5, last)], tmin32[min((r0 >> 5) + 1, last)]);
    const int fc = min(tmin32[min(c0 >> 5, last)], tmin32[min((c0 >> 5) + 1, last)]);
    if (fr >= C0 || fc >= C0) return;          // these rows have nothing in the block's columns
  }
  __shared__ double Li[kTile * LS], Lj[kTile * LS];
  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int fcol = lane & 15, frow = lane >> 4;
  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; m++)
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int r = r0 + 32 * wr + 16 * m + frow + 4 * g;
        const int c = c0 + 32 * wc + 16 * n + fcol;
        const bool in = r <= np && c < np && c <= r;
        const double v = A[in ? (size_t)r * ld + c : (size_t)0];
        acc[m][n][g] = in ? v : 0.0;
      }
  const int fi = lane & 15, fk = lane >> 4;
  for (int q0 = 0; q0 < W; q0 += KC) {
    __syncthreads();                           // the previous chunk's fragment reads are done
    for (int k = t; k < kTile * KC; k += 256) {
      const int r = k / KC, q = k % KC;
      const double vi = A[(size_t)min(r0 + r, np) * ld + J0 + q0 + q];
      const double vj = A[(size_t)min(c0 + r, np) * ld + J0 + q0 + q];
      Li[r * LS + q] = (r0 + r <= np) ? vi : 0.0;
      Lj[r * LS + q] = (c0 + r < np) ? vj : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int k4 = 0; k4 < KC; k4 += 4) {
      double a[2], b[2];
#pragma unroll
      for (int m = 0; m < 2; m++) {
        a[m] = -Li[(32 * wr + 16 * m + fi) * LS + k4 + fk];
        b[m] = Lj[(32 * wc + 16 * m + fi) * LS + k4 + fk];
      }
#pragma unroll
      for (int m = 0; m < 2; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
          acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
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
        if (r <= np && c < np && c <= r) A[(size_t)r * ld + c] = acc[m][n][g];
      }
}

// back substitution L^T x = y (y = row np), one launch per panel from the last to the first: every
// workgroup solves the panel's NB unknowns itself (column-oriented, one wave) and then removes them
// from its 256 entries of y above the panel.  x goes to xs (np doubles).  Used above kBackSmall unknowns.
template <int NB>
__global__ __launch_bounds__(256) void chol_back_kernel(const GnState* __restrict__ st,
                                                        double* __restrict__ A,
                                                        const double* __restrict__ Ldiag,
                                                        double* __restrict__ xs, int np, int ld, int jb) {
  if (st->done) return;
  __shared__ double Lb[NB][NB + 1];
  __shared__ double xb[NB];
  const int t = threadIdx.x;
  double* y = A + (size_t)np * ld;
  const double* Ld = Ldiag + (size_t)(jb / NB) * NB * NB;
  for (int k = t; k < NB * NB; k += 256) Lb[k / NB][k % NB] = Ld[k];
  __syncthreads();
  if (t < 64) {
    double s = t < NB ? y[jb + t] : 0.0;
    const double rd = t < NB ? 1.0 / Lb[t][t] : 0.0;
    for (int c = NB - 1; c >= 0; c--) {
      const double xc = __shfl(s * rd, c, 64);
      if (t == c) xb[c] = xc;
      if (t < c) s -= Lb[c][t] * xc;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < NB) xs[jb + t] = xb[t];
  const int i = blockIdx.x * 256 + t;
  if (i < jb) {
    double s = y[i];
#pragma unroll 8
    for (int k = 0; k < NB; k++) s -= A[(size_t)(jb + k) * ld + i] * xb[k];  // rows of L: contiguous in i
    y[i] = s;
  }
}

// the same back substitution as ONE workgroup and ONE launch while y fits in LDS (np <= kBackSmall):
// panel after panel, y never leaves LDS.
constexpr int kBackSmall = 2048;
constexpr int kOuterW = 128;        // outer block width of the two-level LL^T ...
constexpr int kOuterFrom = 1536;    // ... used above this many unknowns

template <int NB>
__global__ __launch_bounds__(256) void chol_back_small_kernel(const GnState* __restrict__ st,
                                                              const double* __restrict__ A,
                                                              const double* __restrict__ Ldiag,
                                                              double* __restrict__ xs, int np, int ld) {
  if (st->done) return;
  __shared__ double ys[kBackSmall];
  __shared__ double Lb[NB][NB + 1];
  __shared__ double xb[NB];
  const int t = threadIdx.x;
  for (int i = t; i < np; i += 256) ys[i] = A[(size_t)np * ld + i];
  for (int jb = np - NB; jb >= 0; jb -= NB) {
    const double* Ld = Ldiag + (size_t)(jb / NB) * NB * NB;
    for (int k = t; k < NB * NB; k += 256) Lb[k / NB][k % NB] = Ld[k];
    __syncthreads();  // Lb loaded; ys of this panel final
    if (t < 64) {
      double s = t < NB ? ys[jb + t] : 0.0;
      const double rd = t < NB ? 1.0 / Lb[t][t] : 0.0;
      for (int c = NB - 1; c >= 0; c--) {
        const double xc = __shfl(s * rd, c, 64);
        if (t == c) xb[c] = xc;
        if (t < c) s -= Lb[c][t] * xc;
      }
    }
    __syncthreads();
    if (t < NB) xs[jb + t] = xb[t];
    for (int i = t; i < jb; i += 256) {
      double s = ys[i];
#pragma unroll 8
      for (int k = 0; k < NB; k++) s -= A[(size_t)(jb + k) * ld + i] * xb[k];
      ys[i] = s;
    }
    // the next round's first barrier orders these writes (and the reads of Lb / xb) before their reuse
    __syncthreads();
  }
}

// dx = -x, retraction, convergence flag
__global__ __launch_bounds__(256) void gn_finish_kernel(GnState* __restrict__ st,
                                                        const double* __restrict__ xs, int N, int num_fix,
                                                        float* __restrict__ Twc, float* __restrict__ dx,
                                                        float delta_thresh) {
  if (st->done) return;
  const int t = threadIdx.x;
  const int n = N * 7;
  __shared__ float red[256];
  const bool fail = st->chol_fail != 0;
  float ss = 0.0f;
  for (int k = t; k < n; k += 256) {
    const float d = fail ? 0.0f : -(float)xs[k];  // "NOTE: Accounting for negative here!" :1208-1209
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
  int *ii_edge, *jj_edge, *ii_opt, *jj_opt, *first;
  float* econst;
  float* Hs;
  float* gs;
  double* Haug;
  double* Ldiag;
  double* xs;
  int* tmin32;   // envelope of H: first non-zero column of the rows [32 b, 32 b + 32), see chol_step_kernel
  int* counts;
  float* partial;
  float* stream;
  int S, chunk_len, np, ld, nb;
  size_t bytes_fixed;  // everything the index / solve entry points touch
  size_t bytes;        // + compact stream, counts and partials of `local_edges` edges
};

// Layout: arrays whose size depends only on (P, E) first, the per-local-edge arrays last.
static GnWorkspace gn_carve(void* base, int P, int E, int HW, int local_edges) {
  GnWorkspace w;
  const int N = P > 1 ? P - 1 : 0;
  const int n = N * 7;
  w.np = (int)align_up((size_t)(n > 0 ? n : 1), kTile);
  w.ld = w.np;
  // 32-wide panels at every size: a panel step is latency-bound (~32 us against 105-147 us for a 64-wide one), and with
  // the envelope skip the extra passes over the trailing matrix cost less than the longer steps (measured at 300-1 250
  // keyframes, profiles/r02_gn_panel_width.log: 46 vs 66 ms per iteration at 8 743 unknowns)
  w.nb = 32;
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
  w.econst = (float*)take(sizeof(float) * (size_t)E * kEdgeConst);
  w.Hs = (float*)take(sizeof(float) * (size_t)4 * E * 49);
  w.gs = (float*)take(sizeof(float) * (size_t)2 * E * 7);
  w.Haug = (double*)take(sizeof(double) * (size_t)(w.np + 1) * w.ld);
  w.Ldiag = (double*)take(sizeof(double) * (size_t)w.np * w.nb);
  w.xs = (double*)take(sizeof(double) * (size_t)w.np);
  w.tmin32 = (int*)take(sizeof(int) * (size_t)(w.np / 32 + 2));
  w.bytes_fixed = off;
  const size_t L = local_edges > 0 ? (size_t)local_edges : 0;
  w.counts = (int*)take(sizeof(int) * L * S);
  w.partial = (float*)take(sizeof(float) * L * S * kAcc);
  w.stream = (float*)take(sizeof(float) * L * S * (size_t)w.chunk_len * kPlanes);
  w.bytes = off;
  return w;
}

static void fill_params(GnParams& P, int kind, const float* K_dev, float sigma_a, float sigma_b, int height,
                        int width, int pixel_border, float z_eps) {
  P.sa_inv = 1.0f / sigma_a;
  P.sb_inv = kind == 2 ? 0.0f : 1.0f / sigma_b;
  P.C_thresh = 0.0f;
  P.Q_thresh = 0.0f;
  P.height = height;
  P.width = width > 0 ? width : 1;
  P.border_lo = (float)pixel_border;
  P.border_hi_u = (float)(width - 1 - pixel_border);
  P.border_hi_v = (float)(height - 1 - pixel_border);
  P.z_eps = z_eps;
  P.K = K_dev;
}

// edges [e0, e0+cnt) of the global edge list; the stream / counts / partials are indexed by the LOCAL
// edge (0..cnt-1); Hs/gs are the global [4,E,7,7] / [2,E,7] buffers.
static int launch_accumulate(int kind, const GnWorkspace& w, const float* Twc, int E, int e0, int cnt,
                             const GnParams& P, float* Hs, float* gs, hipStream_t s) {
  if (cnt <= 0) return MSLAM_OK;
  hipLaunchKernelGGL(gn_edge_setup_kernel, dim3((cnt + 63) / 64), dim3(64), 0, s, w.st, Twc, w.ii_edge + e0,
                     w.jj_edge + e0, cnt, w.econst);
  dim3 grid((unsigned)w.S * (unsigned)cnt);
  if (kind == 0)
    hipLaunchKernelGGL(gn_accum_kernel<0>, grid, dim3(256), 0, s, w.st, w.econst, w.stream, w.counts, w.chunk_len,
                       P, w.partial, w.S);
  else if (kind == 1)
    hipLaunchKernelGGL(gn_accum_kernel<1>, grid, dim3(256), 0, s, w.st, w.econst, w.stream, w.counts, w.chunk_len,
                       P, w.partial, w.S);
  else
    hipLaunchKernelGGL(gn_accum_kernel<2>, grid, dim3(256), 0, s, w.st, w.econst, w.stream, w.counts, w.chunk_len,
                       P, w.partial, w.S);
  hipLaunchKernelGGL(gn_reduce_kernel, dim3(cnt), dim3(64), 0, s, w.st, w.partial, w.S, Twc, w.ii_edge + e0, e0, E,
                     Hs, gs);
  return check_hip(hipGetLastError(), "gn accumulate launch");
}

template <int NB>
static int launch_cholesky(const GnWorkspace& w, hipStream_t s) {
  constexpr size_t shmem = sizeof(double) * (2 * kTile * (NB + 2) + NB * (NB + 1) + 2 * NB);
  static bool attr_set = false;
  if (shmem > 48 * 1024 && !attr_set) {
    int rc = check_hip(hipFuncSetAttribute((const void*)chol_step_kernel<NB>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                       "hipFuncSetAttribute(chol_step)");
    if (rc) return rc;
    attr_set = true;
  }
  auto step = [&](int j0, int update, int clim) {
    const int j1 = j0 + NB;
    const int tiles_r = (w.np - j1 + 1 + kTile - 1) / kTile;                           // rows j1..np (np = the rhs row)
    const int cend = clim < w.np ? clim : w.np;
    const int tiles_c = update ? (cend - j1 + kTile - 1) / kTile : 1;                 // no update: only the panel column
    hipLaunchKernelGGL(chol_step_kernel<NB>, dim3(tiles_c > 0 ? tiles_c : 1, tiles_r), dim3(256), shmem, s, w.st, w.Haug,
                       w.Ldiag, w.np, w.ld, j0, w.tmin32, update, clim);
  };
  // one level (every panel step updates the whole trailing matrix) while that matrix is small: the steps are latency-
  // bound there; outer blocks of kOuterW columns above (see chol_outer_kernel)
  const int W = w.np > kOuterFrom ? kOuterW : w.np;
  step(-NB, 0, w.np);                                        // the first panel
  for (int J0 = 0; J0 < w.np; J0 += W) {
    const int jend = J0 + W < w.np ? J0 + W : w.np;
    for (int j0 = J0; j0 + NB < jend; j0 += NB) step(j0, 1, jend);     // panels inside the block
    if (jend < w.np) {
      const int tiles = (w.np - jend + 1 + kTile - 1) / kTile;
      hipLaunchKernelGGL(chol_outer_kernel, dim3((w.np - jend + kTile - 1) / kTile, tiles), dim3(256), 0, s, w.st, w.Haug,
                         w.np, w.ld, J0, jend - J0, w.tmin32);
      step(jend - NB, 0, w.np);                              // the next block's first panel: its columns are up to date
    }
  }
  if (w.np <= kBackSmall) {
    hipLaunchKernelGGL(chol_back_small_kernel<NB>, dim3(1), dim3(256), 0, s, w.st, w.Haug, w.Ldiag, w.xs, w.np, w.ld);
  } else {
    for (int jb = w.np - NB; jb >= 0; jb -= NB) {
      const int blocks = jb > 0 ? (jb + 255) / 256 : 1;
      hipLaunchKernelGGL(chol_back_kernel<NB>, dim3(blocks), dim3(256), 0, s, w.st, w.Haug, w.Ldiag, w.xs, w.np,
                         w.ld, jb);
    }
  }
  return MSLAM_OK;
}

static int launch_solve(const GnWorkspace& w, const float* Hs, const float* gs, int E, int P, float* Twc,
                        float* dx, float delta_thresh, hipStream_t s) {
  const int N = P - 1;
  int rc = check_hip(hipMemsetAsync(w.Haug, 0, sizeof(double) * (size_t)(w.np + 1) * w.ld, s), "Haug memset");
  if (rc) return rc;
  rc = check_hip(hipMemsetAsync(w.tmin32, 0x7f, sizeof(int) * (size_t)(w.np / 32 + 2), s), "tmin32 memset");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_assemble_kernel, dim3(N + 1), dim3(64), 0, s, w.st, Hs, gs, w.ii_opt, w.jj_opt, E, N, w.np,
                     w.ld, w.Haug, w.tmin32);
  rc = launch_cholesky<32>(w, s);
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finish_kernel, dim3(1), dim3(256), 0, s, w.st, w.xs, N, 1, Twc, dx, delta_thresh);
  return check_hip(hipGetLastError(), "gn solve launch");
}

static int gn_check_ws(size_t need, void* workspace, size_t workspace_bytes, const char* who) {
  if (!workspace || need > workspace_bytes) {
    set_error("%s: workspace too small (%zu < %zu)", who, workspace ? workspace_bytes : (size_t)0, need);
    return MSLAM_ENOMEM;
  }
  return MSLAM_OK;
}

}  // namespace mslam

using namespace mslam;

extern "C" size_t mslam_gn_workspace_bytes(int num_poses, int num_edges, int num_points, int local_edges) {
  if (num_poses < 0 || num_edges < 0 || num_points < 0 || local_edges < 0) return 0;
  return gn_carve(nullptr, num_poses, num_edges, num_points, local_edges).bytes;
}

extern "C" int mslam_gn_begin(const int64_t* ii, const int64_t* jj, int num_poses, int num_edges,
                              int num_points, void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(num_poses >= 2 && num_edges >= 1, "gn_begin: need >= 2 poses and >= 1 edge");
  MSLAM_REQUIRE(ii && jj, "gn_begin: null pointer");
  MSLAM_REQUIRE(num_edges <= (1 << 22), "gn_begin: %d edges (the index preparation is quadratic in the edge count)", num_edges);
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points, 0);
  int rc = gn_check_ws(w.bytes_fixed, workspace, workspace_bytes, "gn_begin");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (2 * num_edges + 255) / 256;
  hipLaunchKernelGGL(gn_state_init_kernel, dim3(1), dim3(1), 0, s, w.st);
  hipLaunchKernelGGL(gn_first_kernel, dim3(blocks), dim3(256), 0, s, ii, jj, num_edges, w.first);
  hipLaunchKernelGGL(gn_rank_kernel, dim3(blocks), dim3(256), 0, s, ii, jj, num_edges, 1, w.first, w.ii_edge,
                     w.jj_edge, w.ii_opt, w.jj_opt);
  return check_hip(hipGetLastError(), "gn_begin launch");
}

extern "C" int mslam_gn_compact_at(const float* Xs, const float* Cs, const int64_t* idx_ii2jj,
                                   const uint8_t* valid_match, const float* Q, int num_poses, int num_points,
                                   int num_edges, int edge_begin, int edge_count, int slot_begin, int range_count,
                                   float C_thresh, float Q_thresh, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  MSLAM_REQUIRE(num_poses >= 2 && num_points >= 1 && num_edges >= 1, "gn_compact: bad sizes");
  MSLAM_REQUIRE(edge_begin >= 0 && edge_count >= 0 && edge_begin + edge_count <= num_edges,
                "gn_compact: edge range [%d,%d) outside [0,%d)", edge_begin, edge_begin + edge_count, num_edges);
  MSLAM_REQUIRE(slot_begin >= 0 && slot_begin + edge_count <= range_count,
                "gn_compact: slots [%d,%d) outside the accumulate range of %d edges", slot_begin,
                slot_begin + edge_count, range_count);
  if (edge_count == 0) return MSLAM_OK;
  MSLAM_REQUIRE(Xs && Cs && idx_ii2jj && valid_match && Q, "gn_compact: null pointer");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points, range_count);
  int rc = gn_check_ws(w.bytes, workspace, workspace_bytes, "gn_compact");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_compact_kernel, dim3((unsigned)w.S * (unsigned)edge_count), dim3(256), 0, (hipStream_t)stream, Xs,
                     Cs, w.ii_edge + edge_begin, w.jj_edge + edge_begin, idx_ii2jj, valid_match, Q, num_points,
                     w.chunk_len, C_thresh, Q_thresh, w.stream, w.counts, w.S, slot_begin);
  return check_hip(hipGetLastError(), "gn_compact launch");
}

extern "C" int mslam_gn_compact(const float* Xs, const float* Cs, const int64_t* idx_ii2jj,
                                const uint8_t* valid_match, const float* Q, int num_poses, int num_points,
                                int num_edges, int edge_begin, int edge_count, float C_thresh, float Q_thresh,
                                void* workspace, size_t workspace_bytes, void* stream) {
  return mslam_gn_compact_at(Xs, Cs, idx_ii2jj, valid_match, Q, num_poses, num_points, num_edges, edge_begin,
                             edge_count, 0, edge_count, C_thresh, Q_thresh, workspace, workspace_bytes, stream);
}

extern "C" int mslam_gn_accumulate(int kind, const float* Twc, const float* K, int num_poses, int num_points,
                                   int num_edges, int edge_begin, int edge_count, float sigma_a, float sigma_b,
                                   int height, int width, int pixel_border, float z_eps, float* Hs, float* gs,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(kind >= 0 && kind <= 2, "gn_accumulate: kind must be 0 (rays), 1 (calib) or 2 (points)");
  MSLAM_REQUIRE(num_poses >= 2 && num_points >= 1 && num_edges >= 1, "gn_accumulate: bad sizes");
  MSLAM_REQUIRE(edge_begin >= 0 && edge_count >= 0 && edge_begin + edge_count <= num_edges,
                "gn_accumulate: edge range [%d,%d) outside [0,%d)", edge_begin, edge_begin + edge_count, num_edges);
  if (edge_count == 0) return MSLAM_OK;
  MSLAM_REQUIRE(Twc && Hs && gs, "gn_accumulate: null pointer");
  MSLAM_REQUIRE(kind != 1 || (K && width > 0 && height > 0), "gn_accumulate: calib needs K, height, width");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points, edge_count);
  int rc = gn_check_ws(w.bytes, workspace, workspace_bytes, "gn_accumulate");
  if (rc) return rc;
  GnParams P;
  fill_params(P, kind, K, sigma_a, sigma_b, height, width, pixel_border, z_eps);
  return launch_accumulate(kind, w, Twc, num_edges, edge_begin, edge_count, P, Hs, gs, (hipStream_t)stream);
}

extern "C" int mslam_gn_solve_retract(const float* Hs, const float* gs, int num_poses, int num_edges,
                                      int num_points, float* Twc, float* dx, float delta_thresh,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(num_poses >= 2 && num_edges >= 1, "gn_solve_retract: need >= 2 poses and >= 1 edge");
  MSLAM_REQUIRE(Hs && gs && Twc && dx, "gn_solve_retract: null pointer");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points, 0);
  int rc = gn_check_ws(w.bytes_fixed, workspace, workspace_bytes, "gn_solve_retract");
  if (rc) return rc;
  return launch_solve(w, Hs, gs, num_edges, num_poses, Twc, dx, delta_thresh, (hipStream_t)stream);
}

extern "C" int mslam_gn_status(int* status4, int num_poses, int num_edges, int num_points, void* workspace,
                               size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(status4, "gn_status: null pointer");
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points, 0);
  int rc = gn_check_ws(w.bytes_fixed, workspace, workspace_bytes, "gn_status");
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
  rc = mslam_gn_compact(Xs, Cs, idx_ii2jj, valid_match, Q, num_poses, num_points, num_edges, 0, num_edges, C_thresh,
                        Q_thresh, workspace, workspace_bytes, stream);
  if (rc) return rc;
  GnWorkspace w = gn_carve(workspace, num_poses, num_edges, num_points, num_edges);
  hipStream_t s = (hipStream_t)stream;
  GnParams P;
  fill_params(P, kind, K, sigma_a, sigma_b, height, width, pixel_border, z_eps);
  rc = check_hip(hipMemsetAsync(dx, 0, sizeof(float) * 7 * (size_t)(num_poses - 1), s), "dx memset");
  if (rc) return rc;
  for (int it = 0; it < max_iter; it++) {
    rc = launch_accumulate(kind, w, Twc, num_edges, 0, num_edges, P, w.Hs, w.gs, s);
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
  sim3_store(O + 8 * i, sim3_unit(R));   // lietorch normalises the quaternion of every group element it constructs
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
