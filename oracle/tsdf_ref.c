/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement of the reference's global sparse TSDF ("dual TSDF", world-space half):
 *   TSDFVolume.integrate       /root/reference/mast3r_slam/tsdf/global_volume.py:35-72
 *   TSDFVolume._update_voxel   global_volume.py:74-88
 *   TSDFVolume._world_to_voxel global_volume.py:133-134
 *   TSDFVolume.query           global_volume.py:93-105
 *   TSDFVolume._estimate_gradient global_volume.py:107-128
 *   TSDFPoseOptimizer._build_linear_system / _accumulate_system / _sim3_jacobian
 *                              /root/reference/mast3r_slam/tsdf/tsdf_optimizer.py:94-124
 *
 * Pinned by tests/golden/tsdf_global.npz, produced by RUNNING the reference's global_volume.py /
 * tsdf_optimizer.py in the build container (NumPy 2.2.6).  The arithmetic below therefore follows
 * NumPy-2 (NEP 50) promotion, under which every per-sample quantity of integrate() is float32:
 *   ray, ray_length (= sqrtf of OpenBLAS sdot: float products accumulated in double), direction,
 *   max_distance, linspace (f32: k*step, last = stop), sample, sdf, tsdf_value; weight is float64
 *   (conf is float64, math.exp works in double).  Voxel state: tsdf float64 after the 2nd touch,
 *   the float32 value itself after the first; weight float64.
 * (The reference pins numpy==1.26.4, whose legacy promotion makes `distances` float64; keys can then
 *  differ when a sample falls within one ulp of a voxel face.  Recorded in DESIGN.md.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int64_t kx, ky, kz;
  double tsdf, weight;
  uint8_t used;
  uint8_t touched_once; /* tsdf is still the np.float32 of the first touch (matters in the gradient) */
} Voxel;

typedef struct {
  double voxel_size, trunc, max_weight, min_weight;
  size_t cap, count;
  Voxel* tab;
} Vol;

static uint64_t mix(int64_t x, int64_t y, int64_t z) {
  uint64_t h = (uint64_t)x * 0x9E3779B97F4A7C15ull ^ (uint64_t)y * 0xC2B2AE3D27D4EB4Full ^ (uint64_t)z * 0x165667B19E3779F9ull;
  h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  return h;
}

static Voxel* find(Vol* v, int64_t x, int64_t y, int64_t z, int create) {
  size_t i = mix(x, y, z) & (v->cap - 1);
  for (;;) {
    Voxel* s = &v->tab[i];
    if (!s->used) {
      if (!create) return NULL;
      s->used = 1; s->kx = x; s->ky = y; s->kz = z; s->tsdf = 1.0; s->weight = 0.0;
      v->count++;
      return s;
    }
    if (s->kx == x && s->ky == y && s->kz == z) return s;
    i = (i + 1) & (v->cap - 1);
  }
}

static void grow(Vol* v) {
  Vol n = *v;
  n.cap = v->cap * 2; n.count = 0;
  n.tab = (Voxel*)calloc(n.cap, sizeof(Voxel));
  for (size_t i = 0; i < v->cap; i++)
    if (v->tab[i].used) {
      Voxel* d = find(&n, v->tab[i].kx, v->tab[i].ky, v->tab[i].kz, 1);
      *d = v->tab[i];
    }
  free(v->tab);
  *v = n;
}

void* oracle_tsdf_create(double voxel_size, double trunc, double max_weight, double min_weight) {
  Vol* v = (Vol*)calloc(1, sizeof(Vol));
  v->voxel_size = voxel_size; v->trunc = trunc; v->max_weight = max_weight; v->min_weight = min_weight;
  v->cap = 1 << 16;
  v->tab = (Voxel*)calloc(v->cap, sizeof(Voxel));
  return v;
}

void oracle_tsdf_free(void* p) {
  Vol* v = (Vol*)p;
  free(v->tab);
  free(v);
}

static inline void world_to_voxel(const Vol* v, const float* p, int64_t* k) { /* :133-134 */
  const float vs = (float)v->voxel_size; /* float32 array / python float -> float32 divide */
  for (int a = 0; a < 3; a++) k[a] = (int64_t)floorf(p[a] / vs);
}

static void update_voxel(Vol* v, const float* sample, float tsdf, double weight) { /* :74-88 */
  if (weight <= 0.0) return;
  int64_t k[3];
  world_to_voxel(v, sample, k);
  if (v->count * 2 >= v->cap) grow(v);
  size_t before = v->count;
  Voxel* s = find(v, k[0], k[1], k[2], 1);
  if (v->count != before) { /* first touch: stored as given, NOT clamped to max_weight */
    s->tsdf = (double)tsdf; s->weight = weight; s->touched_once = 1;
    return;
  }
  double total = s->weight + weight;
  if (total > v->max_weight) total = v->max_weight;
  s->tsdf = (s->tsdf * s->weight + (double)tsdf * weight) / (total > 1.0e-9 ? total : 1.0e-9);
  s->weight = total;
  s->touched_once = 0;
}

/* points (n,3) f32, conf (n) f64, origin (3) f32.  Returns the number of fused points. */
int oracle_tsdf_integrate(void* pv, const float* points, const double* conf, const float* origin, int n,
                          double step_scale) {
  Vol* v = (Vol*)pv;
  double step = v->voxel_size * step_scale;
  if (step < 1.0e-4) step = 1.0e-4;
  const float stepf = (float)step, truncf = (float)v->trunc;
  int fused = 0;
  for (int i = 0; i < n; i++) {
    const float* p = points + 3 * i;
    float ray[3] = {p[0] - origin[0], p[1] - origin[1], p[2] - origin[2]};
    /* np.linalg.norm -> sqrt(x.dot(x)); OpenBLAS sdot: float products, double accumulation */
    const float sq = (float)((double)(ray[0] * ray[0]) + (double)(ray[1] * ray[1]) + (double)(ray[2] * ray[2]));
    const float ray_length = sqrtf(sq);
    if (!isfinite(ray_length) || ray_length < 1.0e-4f) continue;
    const float dir[3] = {ray[0] / ray_length, ray[1] / ray_length, ray[2] / ray_length};
    const float max_distance = ray_length + truncf;
    int num = (int)(max_distance / stepf);
    if (num < 1) num = 1;
    /* np.linspace(0.0, max_distance, num) in float32 */
    const float lstep = num > 1 ? max_distance / (float)(num - 1) : 0.0f;
    for (int k = 0; k < num; k++) {
      float dist;
      if (num == 1) dist = 0.0f;              /* y = arange(1)*delta + 0.0 */
      else if (k == num - 1) dist = max_distance;
      else dist = (float)k * lstep;
      const float sample[3] = {origin[0] + dist * dir[0], origin[1] + dist * dir[1], origin[2] + dist * dir[2]};
      const float sdf = ray_length - dist;
      if (fabsf(sdf) > truncf) continue;
      float tv = sdf / truncf;
      tv = tv < -1.0f ? -1.0f : (tv > 1.0f ? 1.0f : tv);
      const float e = -fabsf(sdf) / truncf;
      const double weight = conf[i] * exp((double)e);
      update_voxel(v, sample, tv, weight);
    }
    fused++;
  }
  return fused;
}

size_t oracle_tsdf_size(void* pv) { return ((Vol*)pv)->count; }

void oracle_tsdf_dump(void* pv, int64_t* keys, double* tsdf, double* weight) {
  Vol* v = (Vol*)pv;
  size_t o = 0;
  for (size_t i = 0; i < v->cap; i++)
    if (v->tab[i].used) {
      keys[3 * o] = v->tab[i].kx; keys[3 * o + 1] = v->tab[i].ky; keys[3 * o + 2] = v->tab[i].kz;
      tsdf[o] = v->tab[i].tsdf; weight[o] = v->tab[i].weight;
      o++;
    }
}

/* query (:93-128) for n float32 points: status 0 = None/None, 1 = value only, 2 = value + gradient */
void oracle_tsdf_query(void* pv, const float* points, int n, double* value, double* grad, uint8_t* status) {
  Vol* v = (Vol*)pv;
  for (int i = 0; i < n; i++) {
    int64_t k[3];
    world_to_voxel(v, points + 3 * i, k);
    status[i] = 0; value[i] = 0.0; grad[3 * i] = grad[3 * i + 1] = grad[3 * i + 2] = 0.0;
    Voxel* c = find(v, k[0], k[1], k[2], 0);
    if (!c || c->weight < v->min_weight) continue;
    value[i] = c->tsdf;
    status[i] = 1;
    double g[3] = {0, 0, 0};
    double denom = 0.0;
    for (int a = 0; a < 3; a++) {
      int64_t kp[3] = {k[0], k[1], k[2]}, kn[3] = {k[0], k[1], k[2]};
      kp[a] += 1; kn[a] -= 1;
      Voxel* vp = find(v, kp[0], kp[1], kp[2], 0);
      Voxel* vn = find(v, kn[0], kn[1], kn[2], 0);
      if (!vp || !vn) continue;
      if (vp->weight < v->min_weight || vn->weight < v->min_weight) continue;
      if (vp->touched_once && vn->touched_once) {
        /* both still np.float32: the difference and the division happen in float32 */
        const float d = (float)vp->tsdf - (float)vn->tsdf;
        g[a] = (double)(d / (float)(2.0 * v->voxel_size));
      } else {
        g[a] = (vp->tsdf - vn->tsdf) / (2.0 * v->voxel_size);
      }
      denom += 1.0;
    }
    if (denom == 0.0) continue;
    const double norm = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    if (norm < 1.0e-9) continue;
    grad[3 * i] = g[0] / norm; grad[3 * i + 1] = g[1] / norm; grad[3 * i + 2] = g[2] / norm;
    status[i] = 2;
  }
}

/* tsdf_optimizer.py:94-124: residual = tsdf, J = [g, -(p x g), p.g], w = lambda*conf (float32),
 * Jw = sqrt(max(w,1e-6)) J ; H += Jw Jw^T ; b += Jw r sqrt(max(w,1e-6)).  Returns #residuals used. */
int oracle_tsdf_pose_system(void* pv, const float* points, const float* conf, int n, double lambda,
                            double* H, double* b) {
  double* value = (double*)malloc(sizeof(double) * (size_t)n);
  double* grad = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  uint8_t* st = (uint8_t*)malloc((size_t)n);
  oracle_tsdf_query(pv, points, n, value, grad, st);
  memset(H, 0, sizeof(double) * 49);
  memset(b, 0, sizeof(double) * 7);
  int used = 0;
  for (int i = 0; i < n; i++) {
    if (st[i] != 2) continue;
    const double p[3] = {points[3 * i], points[3 * i + 1], points[3 * i + 2]};
    const double* g = grad + 3 * i;
    double J[7];
    J[0] = g[0]; J[1] = g[1]; J[2] = g[2];
    J[3] = -(p[1] * g[2] - p[2] * g[1]);
    J[4] = -(p[2] * g[0] - p[0] * g[2]);
    J[5] = -(p[0] * g[1] - p[1] * g[0]);
    J[6] = p[0] * g[0] + p[1] * g[1] + p[2] * g[2];
    const float wf = (float)lambda * conf[i]; /* python float * np.float32 -> np.float32 */
    const double sw = sqrt(wf > 1.0e-6f ? (double)wf : 1.0e-6);
    const double r = value[i];
    if (!isfinite(r)) continue;
    double Jw[7];
    for (int k = 0; k < 7; k++) Jw[k] = sw * J[k];
    for (int k = 0; k < 7; k++) {
      for (int l = 0; l < 7; l++) H[k * 7 + l] += Jw[k] * Jw[l];
      b[k] += Jw[k] * r * sw;
    }
    used++;
  }
  free(value); free(grad); free(st);
  return used;
}
