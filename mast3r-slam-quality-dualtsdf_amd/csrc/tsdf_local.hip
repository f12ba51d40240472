// Local dense-block TSDF of the dual-TSDF refiner on gfx950: per-block volume build (<= 64^3 voxels)
// and ray-cast zero-crossing extraction.
//
// Reference behaviour reproduced (python double loops with a .item() per voxel there, seconds per block):
//   TSDFRefiner._build_tsdf_robust      mast3r_slam/tsdf_refine.py:837-940
//   TSDFRefiner._extract_surface_safe   tsdf_refine.py:942-1021
//   TSDFRefiner._sample_tsdf_trilinear  tsdf_refine.py:1023-1064
//
// The build keeps the reference's SEQUENTIAL semantics (float32 grid re-rounded after every update,
// python-double arithmetic in between) by replaying each voxel's samples in (point, sample) order:
// emit (thread per point) -> alloc (thread per touched voxel) -> scatter (thread per record) ->
// replay (thread per touched voxel).  torch.linspace is evaluated with torch's per-element device
// formula (start + step*i | end - step*(n-1-i)), which is what the reference executes on its GPU.
// TU built with -ffp-contract=off: float32 steps that feed integer grid indices are not fused.
#include "common.h"

namespace mslam {

struct LocalHdr {
  uint32_t n_valid, n_records, n_touched, seg_cursor;
  uint32_t pad[4];
};

struct LocalWs {
  LocalHdr* hdr;
  uint32_t *cnt, *off, *fill;      // per voxel (64^3 max)
  uint32_t *touched;               // [64^3]
  uint32_t *rec_seq, *rec_vox;     // [max_rec]
  double *rec_sdf, *rec_w;
  uint32_t* seg_seq;
  double *seg_sdf, *seg_w;
  size_t bytes;
};

constexpr uint32_t kMaxVox = 64 * 64 * 64;
constexpr int kLocalSamples = 32;

static LocalWs local_carve(void* base, size_t max_rec) {
  LocalWs w;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += (b + 255) / 256 * 256; return p + r; };
  w.hdr = (LocalHdr*)take(sizeof(LocalHdr));
  w.cnt = (uint32_t*)take(kMaxVox * 4);
  w.off = (uint32_t*)take(kMaxVox * 4);
  w.fill = (uint32_t*)take(kMaxVox * 4);
  w.touched = (uint32_t*)take(kMaxVox * 4);
  w.rec_seq = (uint32_t*)take(max_rec * 4);
  w.rec_vox = (uint32_t*)take(max_rec * 4);
  w.rec_sdf = (double*)take(max_rec * 8);
  w.rec_w = (double*)take(max_rec * 8);
  w.seg_seq = (uint32_t*)take(max_rec * 4);
  w.seg_sdf = (double*)take(max_rec * 8);
  w.seg_w = (double*)take(max_rec * 8);
  w.bytes = o;
  return w;
}

struct LocalGrid {
  float min[3], max[3], actual[3], origin[3];
  int nx, ny, nz;
  float min_conf;
  double voxel_size, trunc;
};

__device__ __forceinline__ bool point_valid(const LocalGrid& G, const float* p, float c) {
  return isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]) && p[0] >= G.min[0] && p[0] <= G.max[0] &&
         p[1] >= G.min[1] && p[1] <= G.max[1] && p[2] >= G.min[2] && p[2] <= G.max[2] && c > G.min_conf;
}

__device__ __forceinline__ LocalGrid load_grid(const float* xyz_min, const float* xyz_max, const float* origin, int nx,
                                               int ny, int nz, float min_conf, double voxel_size, double trunc) {
  LocalGrid G;
  const int n[3] = {nx, ny, nz};
  for (int a = 0; a < 3; a++) {
    G.min[a] = xyz_min[a]; G.max[a] = xyz_max[a];
    G.actual[a] = (xyz_max[a] - xyz_min[a]) / (float)n[a];   // roi_size / tensor([nx,ny,nz]) in float32
    G.origin[a] = origin ? origin[a] : 0.0f;
  }
  G.nx = nx; G.ny = ny; G.nz = nz; G.min_conf = min_conf; G.voxel_size = voxel_size; G.trunc = trunc;
  return G;
}

__device__ __forceinline__ float linspace_at(float s, float e, float step, int n, int i) {
  if (n == 1) return s;
  return (i < n / 2) ? s + step * (float)i : e - step * (float)(n - i - 1);
}

__global__ __launch_bounds__(256) void local_init_kernel(LocalWs W, float* __restrict__ tsdf, float* __restrict__ weights,
                                                         uint32_t nvox) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { W.hdr->n_valid = 0; W.hdr->n_records = 0; W.hdr->n_touched = 0; W.hdr->seg_cursor = 0; }
  if (i < nvox) { tsdf[i] = 1.0f; weights[i] = 0.0f; W.cnt[i] = 0; W.fill[i] = 0; }
}

__global__ __launch_bounds__(256) void local_count_kernel(LocalWs W, const float* __restrict__ Xw,
                                                          const float* __restrict__ C, int n, const float* xyz_min,
                                                          const float* xyz_max, float min_conf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const LocalGrid G = load_grid(xyz_min, xyz_max, nullptr, 1, 1, 1, min_conf, 0.0, 0.0);
  const bool v = i < n && point_valid(G, Xw + 3 * (size_t)i, C[i]);
  const unsigned long long b = __ballot(v);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&W.hdr->n_valid, (uint32_t)__popcll(b));
}

__global__ __launch_bounds__(256) void local_emit_kernel(LocalWs W, const float* __restrict__ Xw,
                                                         const float* __restrict__ C, int n, const float* xyz_min,
                                                         const float* xyz_max, const float* origin, int nx, int ny,
                                                         int nz, float min_conf, double voxel_size, double trunc,
                                                         uint32_t max_rec) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n || W.hdr->n_valid < 5) return;   // `if len(valid_points_world) < 5: return` (:877-878)
  const LocalGrid G = load_grid(xyz_min, xyz_max, origin, nx, ny, nz, min_conf, voxel_size, trunc);
  const float* p = Xw + 3 * (size_t)i;
  if (!point_valid(G, p, C[i])) return;
  const double conf = (double)C[i];
  const float r0 = p[0] - G.origin[0], r1 = p[1] - G.origin[1], r2 = p[2] - G.origin[2];
  const float lenf = sqrtf((r0 * r0 + r1 * r1) + r2 * r2);
  const double ray_length = (double)lenf;
  const float den = (float)(ray_length + 1e-8);
  const float d0 = r0 / den, d1 = r1 / den, d2 = r2 / den;
  if (ray_length < 0.05) return;
  const double t_start = fmax(0.05, ray_length - trunc * 2.0), t_end = ray_length + trunc * 2.0;
  int ns = (int)((t_end - t_start) / voxel_size) + 1;
  if (ns > kLocalSamples) ns = kLocalSamples;
  const float s = (float)t_start, e = (float)t_end;
  const float step = ns > 1 ? (e - s) / (float)(ns - 1) : 0.0f;
  for (int k = 0; k < ns; k++) {
    const float t = linspace_at(s, e, step, ns, k);
    const float s0 = G.origin[0] + d0 * t, s1 = G.origin[1] + d1 * t, s2 = G.origin[2] + d2 * t;
    if (s0 < G.min[0] || s1 < G.min[1] || s2 < G.min[2] || s0 > G.max[0] || s1 > G.max[1] || s2 > G.max[2]) continue;
    const float g0 = (s0 - G.min[0]) / G.actual[0], g1 = (s1 - G.min[1]) / G.actual[1], g2 = (s2 - G.min[2]) / G.actual[2];
    const int gx = (int)fminf(fmaxf(g0, 0.0f), (float)(nx - 1));
    const int gy = (int)fminf(fmaxf(g1, 0.0f), (float)(ny - 1));
    const int gz = (int)fminf(fmaxf(g2, 0.0f), (float)(nz - 1));
    double sdf = (ray_length - (double)t) / trunc;
    sdf = fmax(-1.0, fmin(1.0, sdf));
    const double weight = conf * fmax(0.0, 1.0 - fabs(sdf));
    const uint32_t vox = ((uint32_t)gz * ny + gy) * nx + gx;
    const uint32_t r = atomicAdd(&W.hdr->n_records, 1u);
    if (r >= max_rec) continue;
    W.rec_seq[r] = (uint32_t)i * kLocalSamples + k;
    W.rec_vox[r] = vox;
    W.rec_sdf[r] = sdf;
    W.rec_w[r] = weight;
    if (atomicAdd(&W.cnt[vox], 1u) == 0u) W.touched[atomicAdd(&W.hdr->n_touched, 1u)] = vox;
  }
}

__global__ __launch_bounds__(256) void local_alloc_kernel(LocalWs W) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W.hdr->n_touched) return;
  const uint32_t v = W.touched[i];
  W.off[v] = atomicAdd(&W.hdr->seg_cursor, W.cnt[v]);
}

__global__ __launch_bounds__(256) void local_scatter_kernel(LocalWs W, uint32_t max_rec) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= min(W.hdr->n_records, max_rec)) return;
  const uint32_t v = W.rec_vox[i];
  const uint32_t pos = W.off[v] + atomicAdd(&W.fill[v], 1u);
  W.seg_seq[pos] = W.rec_seq[i];
  W.seg_sdf[pos] = W.rec_sdf[i];
  W.seg_w[pos] = W.rec_w[i];
}

__global__ __launch_bounds__(256) void local_replay_kernel(LocalWs W, float* __restrict__ tsdf,
                                                           float* __restrict__ weights) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W.hdr->n_touched) return;
  const uint32_t v = W.touched[i];
  const uint32_t L = W.cnt[v], o = W.off[v];
  float tv = tsdf[v], wv = weights[v];
  long long last = -1;
  for (uint32_t r = 0; r < L; r++) {
    uint32_t best = 0xFFFFFFFFu, bj = 0;
    for (uint32_t j = 0; j < L; j++) {
      const uint32_t sq = W.seg_seq[o + j];
      if ((long long)sq > last && sq < best) { best = sq; bj = j; }
    }
    last = best;
    const double sdf = W.seg_sdf[o + bj], weight = W.seg_w[o + bj];
    const double old_w = (double)wv, new_w = old_w + weight;
    if (new_w > 1e-6) {
      tv = (float)(((double)tv * old_w + sdf * weight) / new_w);
      wv = (float)new_w;
    }
  }
  tsdf[v] = tv;
  weights[v] = wv;
}

// ---- ray cast: one thread per selected pixel, 64 samples, trilinear TSDF in double ---------------
__device__ __forceinline__ double trilinear(const float* __restrict__ vol, int nx, int ny, int nz, float x, float y,
                                            float z) {
  x = fminf(fmaxf(x, 0.0f), (float)(nx - 1));
  y = fminf(fmaxf(y, 0.0f), (float)(ny - 1));
  z = fminf(fmaxf(z, 0.0f), (float)(nz - 1));
  const int x0 = (int)floorf(x), y0 = (int)floorf(y), z0 = (int)floorf(z);
  const int x1 = min(x0 + 1, nx - 1), y1 = min(y0 + 1, ny - 1), z1 = min(z0 + 1, nz - 1);
  const double xd = (double)(x - (float)x0), yd = (double)(y - (float)y0), zd = (double)(z - (float)z0);
  auto c = [&](int zz, int yy, int xx) { return (double)vol[((size_t)zz * ny + yy) * nx + xx]; };
  const double c00 = c(z0, y0, x0) * (1 - xd) + c(z0, y0, x1) * xd;
  const double c01 = c(z0, y1, x0) * (1 - xd) + c(z0, y1, x1) * xd;
  const double c10 = c(z1, y0, x0) * (1 - xd) + c(z1, y0, x1) * xd;
  const double c11 = c(z1, y1, x0) * (1 - xd) + c(z1, y1, x1) * xd;
  const double c0 = c00 * (1 - yd) + c01 * yd;
  const double c1 = c10 * (1 - yd) + c11 * yd;
  return c0 * (1 - zd) + c1 * zd;
}

__global__ void local_raycast_kernel(const float* __restrict__ vol, int nx, int ny, int nz, const float* xyz_min,
                                     const float* xyz_max, const float* __restrict__ X, const int64_t* __restrict__ sel_pix,
                                     int n_sel, int n_samples, float max_disp, float* __restrict__ surf,
                                     uint8_t* __restrict__ hit) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sel) return;
  hit[k] = 0;
  const LocalGrid G = load_grid(xyz_min, xyz_max, nullptr, nx, ny, nz, 0.0f, 0.0, 0.0);
  const float* p = X + 3 * sel_pix[k];
  const float p0 = p[0], p1 = p[1], p2 = p[2];
  surf[3 * k] = p0; surf[3 * k + 1] = p1; surf[3 * k + 2] = p2;
  const double depth = (double)p2;
  if (depth < 0.05) return;
  const float s = (float)fmax(0.05, depth - 0.1), e = (float)(depth + 0.1);
  const float step = n_samples > 1 ? (e - s) / (float)(n_samples - 1) : 0.0f;
  const float depthf = (float)depth;
  bool have_prev = false;
  double prev_sdf = 0.0;
  float prev_t = 0.0f;
  for (int j = 0; j < n_samples; j++) {
    const float t = linspace_at(s, e, step, n_samples, j);
    const float sc = t / depthf;
    const float q0 = p0 * sc, q1 = p1 * sc, q2 = p2 * sc;   // depth >= 0.05 > 0.01 always here
    if (q0 < G.min[0] || q1 < G.min[1] || q2 < G.min[2] || q0 > G.max[0] || q1 > G.max[1] || q2 > G.max[2]) {
      have_prev = false;
      continue;
    }
    const double sdf = trilinear(vol, nx, ny, nz, (q0 - G.min[0]) / G.actual[0], (q1 - G.min[1]) / G.actual[1],
                                 (q2 - G.min[2]) / G.actual[2]);
    if (have_prev && prev_sdf * sdf < 0.0) {
      const double alpha = fabs(prev_sdf) / (fabs(prev_sdf) + fabs(sdf) + 1e-8);
      const float t_surf = prev_t + (float)alpha * (t - prev_t);
      const float ss = t_surf / depthf;
      const float u0 = p0 * ss, u1 = p1 * ss, u2 = p2 * ss;
      const float e0 = u0 - p0, e1 = u1 - p1, e2 = u2 - p2;
      const float disp = sqrtf((e0 * e0 + e1 * e1) + e2 * e2);
      if ((double)disp <= (double)max_disp) {
        surf[3 * k] = u0; surf[3 * k + 1] = u1; surf[3 * k + 2] = u2;
        hit[k] = 1;
      }
      return;
    }
    prev_sdf = sdf; prev_t = t; have_prev = true;
  }
}

}  // namespace mslam

using namespace mslam;

extern "C" size_t mslam_tsdf_local_workspace_bytes(int n_points) {
  if (n_points <= 0) return 0;
  return local_carve(nullptr, (size_t)n_points * kLocalSamples).bytes;
}

extern "C" int mslam_tsdf_local_build(const float* X_world, const float* C, const float* origin, const float* xyz_min,
                                      const float* xyz_max, int n_points, int nx, int ny, int nz, double voxel_size,
                                      double trunc, float min_confidence, float* tsdf, float* weights, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(n_points > 0 && nx >= 1 && ny >= 1 && nz >= 1 && nx <= 64 && ny <= 64 && nz <= 64,
                "tsdf_local_build: grid %dx%dx%d must be within 1..64 per axis", nx, ny, nz);
  MSLAM_REQUIRE(X_world && C && origin && xyz_min && xyz_max && tsdf && weights && workspace, "tsdf_local_build: null pointer");
  const size_t max_rec = (size_t)n_points * kLocalSamples;
  MSLAM_REQUIRE(max_rec < 0xFFFFFFFFull, "tsdf_local_build: too many points");
  LocalWs W = local_carve(workspace, max_rec);
  if (W.bytes > workspace_bytes) {
    set_error("tsdf_local_build: workspace too small (%zu < %zu)", workspace_bytes, W.bytes);
    return MSLAM_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  const uint32_t nvox = (uint32_t)nx * ny * nz;
  const unsigned pb = (n_points + 255) / 256, rb = (unsigned)((max_rec + 255) / 256);
  hipLaunchKernelGGL(local_init_kernel, dim3((nvox + 255) / 256), dim3(256), 0, s, W, tsdf, weights, nvox);
  hipLaunchKernelGGL(local_count_kernel, dim3(pb), dim3(256), 0, s, W, X_world, C, n_points, xyz_min, xyz_max, min_confidence);
  hipLaunchKernelGGL(local_emit_kernel, dim3(pb), dim3(256), 0, s, W, X_world, C, n_points, xyz_min, xyz_max, origin, nx,
                     ny, nz, min_confidence, voxel_size, trunc, (uint32_t)max_rec);
  hipLaunchKernelGGL(local_alloc_kernel, dim3((nvox + 255) / 256), dim3(256), 0, s, W);
  hipLaunchKernelGGL(local_scatter_kernel, dim3(rb), dim3(256), 0, s, W, (uint32_t)max_rec);
  hipLaunchKernelGGL(local_replay_kernel, dim3((nvox + 255) / 256), dim3(256), 0, s, W, tsdf, weights);
  MSLAM_LAUNCH_CHECK("tsdf_local_build");
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_local_raycast(const float* tsdf, int nx, int ny, int nz, const float* xyz_min,
                                        const float* xyz_max, const float* X_original, const int64_t* sel_pix, int n_sel,
                                        int n_samples, float max_displacement, float* surf, uint8_t* hit, void* stream) {
  MSLAM_REQUIRE(n_sel >= 0 && n_samples >= 1, "tsdf_local_raycast: bad sizes");
  if (n_sel == 0) return MSLAM_OK;
  MSLAM_REQUIRE(tsdf && xyz_min && xyz_max && X_original && sel_pix && surf && hit, "tsdf_local_raycast: null pointer");
  hipLaunchKernelGGL(local_raycast_kernel, dim3((n_sel + 63) / 64), dim3(64), 0, (hipStream_t)stream, tsdf, nx, ny, nz,
                     xyz_min, xyz_max, X_original, sel_pix, n_sel, n_samples, max_displacement, surf, hit);
  MSLAM_LAUNCH_CHECK("tsdf_local_raycast");
  return MSLAM_OK;
}
