// Global sparse TSDF for gfx950: GPU voxel hash (open addressing, 64-bit packed integer keys),
// ray-sample integration with EXACT sequential-update semantics, nearest-voxel query with
// 6-neighbour gradient, and the TSDF pose normal equations.
//
// Reference behaviour reproduced (python dict + per-voxel .item() loops there):
//   TSDFVolume.integrate / _update_voxel / _world_to_voxel   mast3r_slam/tsdf/global_volume.py:35-88,133-134
//   TSDFVolume.query / _estimate_gradient                    global_volume.py:93-128
//   TSDFPoseOptimizer._build_linear_system/_accumulate_system/_sim3_jacobian
//                                                            mast3r_slam/tsdf/tsdf_optimizer.py:94-124
//   solve + pose update                                      tsdf_optimizer.py:77-86
//
// Why not atomics on (tsdf*w, w): the reference's running update is order dependent once a voxel's
// weight hits max_weight (numerator keeps the unclamped sum), and the very first
// touch stores the sample un-averaged.  To return the same numbers we REPLAY each voxel's samples in
// the reference's order (point index, then sample index) - parallel across voxels, sequential inside
// one voxel:
//   emit    one thread per point walks its in-band samples, inserts keys (atomicCAS), appends records
//   alloc   one thread per touched voxel reserves a contiguous segment (atomicAdd cursor)
//   scatter one thread per record moves it into its voxel's segment (arrival order is arbitrary)
//   replay  one thread per touched voxel applies its records in increasing sequence number
// All fp32 steps that feed the integer key are the reference's float32 operations in the same order
// (TU built with -ffp-contract=off), so keys are bit-exact.
#include "common.h"
#include "sim3.h"

namespace mslam {

constexpr uint64_t kEmptyKey = ~0ull;
constexpr int kKeyBias = 1 << 20;  // coordinates in [-2^20, 2^20)

struct TsdfHeader {
  uint64_t capacity;     // slots (power of two)
  uint32_t count;        // occupied slots
  uint32_t overflow;     // != 0: table full or coordinate out of range (samples were dropped)
  uint32_t n_records;    // scratch: records emitted by the running integrate
  uint32_t n_touched;    // scratch: voxels touched by the running integrate
  uint32_t seg_cursor;   // scratch
  uint32_t fused;        // points fused by the last integrate (return value of the reference)
  uint32_t dump_cursor;
  uint32_t n_big;        // scratch: voxels with more samples than a thread sorts itself (replayed by a wave each)
};

struct TsdfTable {
  TsdfHeader* hdr;
  uint64_t* keys;
  double* tsdf;
  double* weight;
  uint32_t* cnt;
  uint32_t* off;
  uint32_t* fill;
  uint8_t* state;  // 0 absent, 1 touched once (value is still the float32 sample), 2 averaged
  uint64_t cap;
};

static inline size_t al(size_t x) { return (x + 255) / 256 * 256; }

__host__ __device__ inline TsdfTable table_carve(void* base, uint64_t cap) {
  TsdfTable t;
  char* p = (char*)base;
  size_t o = 0;
  t.hdr = (TsdfHeader*)(p + o); o += 256;
  t.keys = (uint64_t*)(p + o); o += (cap * 8 + 255) / 256 * 256;
  t.tsdf = (double*)(p + o); o += (cap * 8 + 255) / 256 * 256;
  t.weight = (double*)(p + o); o += (cap * 8 + 255) / 256 * 256;
  t.cnt = (uint32_t*)(p + o); o += (cap * 4 + 255) / 256 * 256;
  t.off = (uint32_t*)(p + o); o += (cap * 4 + 255) / 256 * 256;
  t.fill = (uint32_t*)(p + o); o += (cap * 4 + 255) / 256 * 256;
  t.state = (uint8_t*)(p + o); o += (cap + 255) / 256 * 256;
  t.cap = cap;
  return t;
}

static size_t table_bytes(uint64_t cap) {
  return 256 + 3 * al(cap * 8) + 3 * al(cap * 4) + al(cap);
}

__device__ __forceinline__ uint64_t mix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
  return k;
}

__device__ __forceinline__ bool pack_key(long long x, long long y, long long z, uint64_t& key) {
  const long long bx = x + kKeyBias, by = y + kKeyBias, bz = z + kKeyBias;
  if ((unsigned long long)bx >= 2ull * kKeyBias || (unsigned long long)by >= 2ull * kKeyBias ||
      (unsigned long long)bz >= 2ull * kKeyBias)
    return false;
  key = ((uint64_t)bx << 42) | ((uint64_t)by << 21) | (uint64_t)bz;
  return true;
}

__device__ __forceinline__ void unpack_key(uint64_t key, long long& x, long long& y, long long& z) {
  x = (long long)((key >> 42) & 0x1FFFFF) - kKeyBias;
  y = (long long)((key >> 21) & 0x1FFFFF) - kKeyBias;
  z = (long long)(key & 0x1FFFFF) - kKeyBias;
}

// floor(float32 / float32(voxel_size)) per axis  (global_volume.py:133-134 under NumPy-2 promotion)
__device__ __forceinline__ bool world_to_key(float px, float py, float pz, float vs, uint64_t& key) {
  return pack_key((long long)floorf(px / vs), (long long)floorf(py / vs), (long long)floorf(pz / vs), key);
}

// Probe sequences are bounded: the host keeps the load factor <= 1/2 (TSDFVolume.maintain grows and rehashes), where a
// linear-probe cluster of kMaxProbe slots does not occur; an insert that would need more reports overflow instead of
// scanning a multi-million-slot table.
constexpr uint64_t kMaxProbe = 1024;

__device__ __forceinline__ int64_t table_find(const TsdfTable& t, uint64_t key) {
  uint64_t s = mix64(key) & (t.cap - 1);
  const uint64_t limit = t.cap < kMaxProbe ? t.cap : kMaxProbe;
  for (uint64_t probe = 0; probe < limit; probe++) {
    const uint64_t k = t.keys[s];
    if (k == key) return (int64_t)s;
    if (k == kEmptyKey) return -1;
    s = (s + 1) & (t.cap - 1);
  }
  return -1;
}

__device__ __forceinline__ int64_t table_insert(const TsdfTable& t, uint64_t key) {
  uint64_t s = mix64(key) & (t.cap - 1);
  const uint64_t limit = t.cap < kMaxProbe ? t.cap : kMaxProbe;
  for (uint64_t probe = 0; probe < limit; probe++) {
    const uint64_t prev = atomicCAS((unsigned long long*)&t.keys[s], (unsigned long long)kEmptyKey,
                                    (unsigned long long)key);
    if (prev == kEmptyKey) { atomicAdd(&t.hdr->count, 1u); return (int64_t)s; }
    if (prev == key) return (int64_t)s;
    s = (s + 1) & (t.cap - 1);
  }
  return -1;
}

__global__ void tsdf_init_kernel(void* base, uint64_t cap) {
  TsdfTable t = table_carve(base, cap);
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    TsdfHeader h;
    h.capacity = cap; h.count = 0; h.overflow = 0; h.n_records = 0; h.n_touched = 0; h.seg_cursor = 0;
    h.fused = 0; h.dump_cursor = 0; h.n_big = 0;
    *t.hdr = h;
  }
  if (i < cap) {
    t.keys[i] = kEmptyKey; t.tsdf[i] = 1.0; t.weight[i] = 0.0;
    t.cnt[i] = 0; t.off[i] = 0; t.fill[i] = 0; t.state[i] = 0;
  }
}

// growth: every occupied slot of the old table moves to the (initialised, larger) new one with its value, weight and
// first-touch state; between two integrate calls the scratch columns (cnt / off / fill) are zero.
__global__ __launch_bounds__(256) void tsdf_rehash_kernel(void* old_base, uint64_t old_cap, void* new_base,
                                                          uint64_t new_cap) {
  const TsdfTable a = table_carve(old_base, old_cap), b = table_carve(new_base, new_cap);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= old_cap) return;
  const uint64_t key = a.keys[i];
  if (key == kEmptyKey) return;
  const int64_t slot = table_insert(b, key);
  if (slot < 0) { b.hdr->overflow = 1; return; }
  b.tsdf[slot] = a.tsdf[i]; b.weight[slot] = a.weight[i]; b.state[slot] = a.state[i];
}

struct IntegrateScratch {
  uint32_t* rec_seq;    // [max_rec]
  uint32_t* rec_slot;
  float* rec_tsdf;
  double* rec_w;
  uint32_t* touched;    // [max_rec]
  uint32_t* seg_seq;    // [max_rec] scattered by voxel segment
  float* seg_tsdf;
  double* seg_w;
  size_t bytes;
};

static IntegrateScratch scratch_carve(void* base, size_t max_rec) {
  IntegrateScratch s;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t b) { size_t r = o; o += al(b); return p + r; };
  s.rec_seq = (uint32_t*)take(max_rec * 4);
  s.rec_slot = (uint32_t*)take(max_rec * 4);
  s.rec_tsdf = (float*)take(max_rec * 4);
  s.rec_w = (double*)take(max_rec * 8);
  s.touched = (uint32_t*)take(max_rec * 4);
  s.seg_seq = (uint32_t*)take(max_rec * 4);
  s.seg_tsdf = (float*)take(max_rec * 4);
  s.seg_w = (double*)take(max_rec * 8);
  s.bytes = o;
  return s;
}

__global__ void tsdf_begin_kernel(TsdfHeader* h) {
  h->n_records = 0; h->n_touched = 0; h->seg_cursor = 0; h->fused = 0; h->n_big = 0;
}

// one thread per point: global_volume.py:51-71
__global__ __launch_bounds__(256) void tsdf_emit_kernel(void* base, uint64_t cap, const float* __restrict__ points,
                                                        const double* __restrict__ conf,
                                                        const float* __restrict__ origin, int n, float vs,
                                                        float stepf, float truncf, int max_band, int shard_id,
                                                        int num_shards, IntegrateScratch S, uint32_t max_rec) {
  TsdfTable t = table_carve(base, cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float o0 = origin[0], o1 = origin[1], o2 = origin[2];
  const float r0 = points[3 * i] - o0, r1 = points[3 * i + 1] - o1, r2 = points[3 * i + 2] - o2;
  // np.linalg.norm: sqrt(sdot) with float products accumulated in double (OpenBLAS), rounded once
  const float sq = (float)(((double)(r0 * r0) + (double)(r1 * r1)) + (double)(r2 * r2));
  const float L = sqrtf(sq);
  if (!isfinite(L) || L < 1.0e-4f) return;
  if (shard_id == 0) atomicAdd(&t.hdr->fused, 1u);
  const float d0 = r0 / L, d1 = r1 / L, d2 = r2 / L;
  const float maxd = L + truncf;
  // A ray of more than 2^20 samples (15 km at the default 1.5 cm step) only occurs when a pose has diverged; the
  // reference would spend minutes in np.linspace there.  The point is dropped and reported (overflow code 3).
  if (!(maxd / stepf < 1048576.0f)) { t.hdr->overflow = 3; return; }
  int num = (int)(maxd / stepf);
  if (num < 1) num = 1;
  const float lstep = num > 1 ? maxd / (float)(num - 1) : 0.0f;
  const double cf = conf[i];
  // the in-band samples (|L - dist| <= trunc) are a contiguous index range; start a little early
  int k0 = num > 1 ? (int)((L - truncf) / lstep) - 2 : 0;
  if (k0 < 0) k0 = 0;
  int emitted = 0;
  for (int k = k0; k < num; k++) {
    float dist;
    if (num == 1) dist = 0.0f;
    else if (k == num - 1) dist = maxd;
    else dist = (float)k * lstep;
    const float sdf = L - dist;
    if (fabsf(sdf) > truncf) {
      if (sdf < 0.0f && k < num - 1) { k = num - 2; }  // past the band: only the exact end point is left
      continue;
    }
    const float s0 = o0 + dist * d0, s1 = o1 + dist * d1, s2 = o2 + dist * d2;
    float tv = sdf / truncf;
    tv = fminf(fmaxf(tv, -1.0f), 1.0f);
    const float e = -fabsf(sdf) / truncf;
    const double w = cf * exp((double)e);
    if (!(w > 0.0)) continue;  // `if weight <= 0.0: return` (NaN weights fall through in python; not emitted here)
    uint64_t key;
    if (!world_to_key(s0, s1, s2, vs, key)) { t.hdr->overflow = 1; continue; }
    if (num_shards > 1 && (int)((mix64(key) >> 40) % (uint64_t)num_shards) != shard_id) continue;
    const int64_t slot = table_insert(t, key);
    if (slot < 0) { t.hdr->overflow = 1; continue; }
    const uint32_t r = atomicAdd(&t.hdr->n_records, 1u);
    if (r >= max_rec || emitted >= max_band) { t.hdr->overflow = 2; continue; }
    S.rec_seq[r] = (uint32_t)i * (uint32_t)max_band + (uint32_t)emitted;
    S.rec_slot[r] = (uint32_t)slot;
    S.rec_tsdf[r] = tv;
    S.rec_w[r] = w;
    emitted++;
    if (atomicAdd(&t.cnt[slot], 1u) == 0u) S.touched[atomicAdd(&t.hdr->n_touched, 1u)] = (uint32_t)slot;
  }
}

__global__ __launch_bounds__(256) void tsdf_alloc_kernel(void* base, uint64_t cap, IntegrateScratch S) {
  TsdfTable t = table_carve(base, cap);
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= t.hdr->n_touched) return;
  const uint32_t slot = S.touched[i];
  t.off[slot] = atomicAdd(&t.hdr->seg_cursor, t.cnt[slot]);
}

__global__ __launch_bounds__(256) void tsdf_scatter_kernel(void* base, uint64_t cap, IntegrateScratch S,
                                                           uint32_t max_rec) {
  TsdfTable t = table_carve(base, cap);
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  const uint32_t nrec = min(t.hdr->n_records, max_rec);
  if (i >= nrec) return;
  const uint32_t slot = S.rec_slot[i];
  const uint32_t pos = t.off[slot] + atomicAdd(&t.fill[slot], 1u);
  S.seg_seq[pos] = S.rec_seq[i];
  S.seg_tsdf[pos] = S.rec_tsdf[i];
  S.seg_w[pos] = S.rec_w[i];
}

// one thread per touched voxel: apply its samples in the reference's order (global_volume.py:74-88)
constexpr uint32_t kSortMax = 48;

__global__ __launch_bounds__(256) void tsdf_replay_kernel(void* base, uint64_t cap, IntegrateScratch S,
                                                          double max_weight) {
  __shared__ uint32_t s_seq[256][kSortMax + 1];   // odd row stride: the rows of a wave fall on different banks
  __shared__ uint8_t s_pos[256][kSortMax + 4];
  TsdfTable t = table_carve(base, cap);
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= t.hdr->n_touched) return;
  const uint32_t slot = S.touched[i];
  const uint32_t L = t.cnt[slot], o = t.off[slot];
  double tsdf = t.tsdf[slot], weight = t.weight[slot];
  uint8_t state = t.state[slot];
  auto apply = [&](uint32_t j) {
    const double tv = (double)S.seg_tsdf[o + j];
    const double w = S.seg_w[o + j];
    if (state == 0) {  // first touch: stored as given, not averaged, not clamped
      tsdf = tv; weight = w; state = 1;
    } else {
      double total = weight + w;
      if (total > max_weight) total = max_weight;
      tsdf = (tsdf * weight + tv * w) / (total > 1.0e-9 ? total : 1.0e-9);
      weight = total;
      state = 2;
    }
  };
  if (L <= kSortMax) {
    // the usual case (a voxel sees ~10 samples of a keyframe): sequence numbers are read ONCE, insertion-sorted in
    // this thread's LDS row, then the records are applied in that order
    uint32_t* sq = &s_seq[threadIdx.x][0];
    uint8_t* sp = &s_pos[threadIdx.x][0];
    for (uint32_t r = 0; r < L; r++) {
      const uint32_t v = S.seg_seq[o + r];
      uint32_t k = r;
      while (k > 0 && sq[k - 1] > v) { sq[k] = sq[k - 1]; sp[k] = sp[k - 1]; k--; }
      sq[k] = v; sp[k] = (uint8_t)r;
    }
    for (uint32_t r = 0; r < L; r++) apply(sp[r]);
  } else {
    // a voxel close to the camera collects hundreds of samples of one keyframe: handed to tsdf_replay_big_kernel (one
    // wave per voxel)
    S.rec_slot[atomicAdd(&t.hdr->n_big, 1u)] = slot;   // rec_slot is dead after the scatter pass
    return;
  }
  t.tsdf[slot] = tsdf; t.weight[slot] = weight; t.state[slot] = state;
  t.cnt[slot] = 0; t.fill[slot] = 0;
}

// one WAVE per voxel with more than kSortMax samples: the lanes rank the sequence numbers (all distinct) against each
// other in LDS - O(L^2 / 64) compares per lane instead of O(L^2) global re-reads by one thread - and stage the samples in
// that order; lane 0 then applies the running update (sequential by definition).  Segments longer than kBigMax fall
// back to the selection scan.
constexpr uint32_t kBigMax = 2048;

__global__ __launch_bounds__(64) void tsdf_replay_big_kernel(void* base, uint64_t cap, IntegrateScratch S,
                                                             double max_weight) {
  __shared__ uint32_t b_seq[kBigMax];
  __shared__ float b_tsdf[kBigMax];
  __shared__ double b_w[kBigMax];
  TsdfTable t = table_carve(base, cap);
  const uint32_t n_big = t.hdr->n_big;
  const int lane = threadIdx.x;
  for (uint32_t v = blockIdx.x; v < n_big; v += gridDim.x) {
    const uint32_t slot = S.rec_slot[v];
    const uint32_t L = t.cnt[slot], o = t.off[slot];
    double tsdf = t.tsdf[slot], weight = t.weight[slot];
    uint8_t state = t.state[slot];
    auto apply = [&](double tv, double w) {
      if (state == 0) {
        tsdf = tv; weight = w; state = 1;
      } else {
        double total = weight + w;
        if (total > max_weight) total = max_weight;
        tsdf = (tsdf * weight + tv * w) / (total > 1.0e-9 ? total : 1.0e-9);
        weight = total;
        state = 2;
      }
    };
    if (L <= kBigMax) {
      for (uint32_t r = lane; r < L; r += 64) b_seq[r] = S.seg_seq[o + r];
      __syncthreads();
      for (uint32_t r = lane; r < L; r += 64) {
        const uint32_t mine = b_seq[r];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < L; j++) rank += b_seq[j] < mine ? 1u : 0u;
        b_tsdf[rank] = S.seg_tsdf[o + r];
        b_w[rank] = S.seg_w[o + r];
      }
      __syncthreads();
      if (lane == 0)
        for (uint32_t r = 0; r < L; r++) apply((double)b_tsdf[r], b_w[r]);
    } else if (lane == 0) {
      long long last = -1;
      for (uint32_t r = 0; r < L; r++) {
        uint32_t best = 0xFFFFFFFFu, bj = 0;
        for (uint32_t j = 0; j < L; j++) {
          const uint32_t sv = S.seg_seq[o + j];
          if ((long long)sv > last && sv < best) { best = sv; bj = j; }
        }
        last = best;
        apply((double)S.seg_tsdf[o + bj], S.seg_w[o + bj]);
      }
    }
    if (lane == 0) {
      t.tsdf[slot] = tsdf; t.weight[slot] = weight; t.state[slot] = state;
      t.cnt[slot] = 0; t.fill[slot] = 0;
    }
    __syncthreads();   // the LDS arrays are reused by the next voxel of this wave
  }
}

// ---------------------------------------------------------------------------------------------
// query + gradient (global_volume.py:93-128)
// ---------------------------------------------------------------------------------------------
// One voxel as the query sees it: state 0 = absent (not in the table, key out of range, or never fused).
struct VoxelVal { int state; double w, v; };

__device__ __forceinline__ void voxel_keys7(float px, float py, float pz, float vs, uint64_t (&keys)[7], bool (&ok)[7]) {
  const long long k[3] = {(long long)floorf(px / vs), (long long)floorf(py / vs), (long long)floorf(pz / vs)};
  ok[0] = pack_key(k[0], k[1], k[2], keys[0]);
#pragma unroll
  for (int a = 0; a < 3; a++) {
    long long kp[3] = {k[0], k[1], k[2]}, kn[3] = {k[0], k[1], k[2]};
    kp[a] += 1; kn[a] -= 1;
    ok[1 + 2 * a] = pack_key(kp[0], kp[1], kp[2], keys[1 + 2 * a]);
    ok[2 + 2 * a] = pack_key(kn[0], kn[1], kn[2], keys[2 + 2 * a]);
  }
}

__device__ __forceinline__ VoxelVal table_fetch(const TsdfTable& t, uint64_t key, bool ok) {
  VoxelVal r = {0, 0.0, 0.0};
  if (!ok) return r;
  const int64_t c = table_find(t, key);
  if (c < 0) return r;
  r.state = t.state[c]; r.w = t.weight[c]; r.v = t.tsdf[c];
  return r;
}

// global_volume.py:93-128 on the seven voxels a query reads (centre, +x, -x, +y, -y, +z, -z), however they were fetched
// (the local table, or - voxel-sharded volumes - the per-voxel sum over the owners' tables).
__device__ __forceinline__ int tsdf_query_core(const VoxelVal (&vx)[7], double voxel_size, double min_weight,
                                               double& value, double* g) {
  value = 0.0; g[0] = g[1] = g[2] = 0.0;
  if (vx[0].state == 0 || vx[0].w < min_weight) return 0;
  value = vx[0].v;
  double denom = 0.0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const VoxelVal &sp = vx[1 + 2 * a], &sn = vx[2 + 2 * a];
    if (sp.state == 0 || sn.state == 0) continue;
    if (sp.w < min_weight || sn.w < min_weight) continue;
    if (sp.state == 1 && sn.state == 1) {
      // both values are still np.float32 in the reference: float32 subtract and divide
      const float d = (float)sp.v - (float)sn.v;
      g[a] = (double)(d / (float)(2.0 * voxel_size));
    } else {
      g[a] = (sp.v - sn.v) / (2.0 * voxel_size);
    }
    denom += 1.0;
  }
  if (denom == 0.0) { g[0] = g[1] = g[2] = 0.0; return 1; }
  const double nrm = sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
  if (nrm < 1.0e-9) { g[0] = g[1] = g[2] = 0.0; return 1; }
  g[0] /= nrm; g[1] /= nrm; g[2] /= nrm;
  return 2;
}

__device__ __forceinline__ int tsdf_query_one(const TsdfTable& t, float px, float py, float pz, float vs,
                                              double voxel_size, double min_weight, double& value, double* g) {
  uint64_t keys[7];
  bool ok[7];
  voxel_keys7(px, py, pz, vs, keys, ok);
  VoxelVal vx[7];
  vx[0] = table_fetch(t, keys[0], ok[0]);
  if (vx[0].state == 0 || vx[0].w < min_weight) { value = 0.0; g[0] = g[1] = g[2] = 0.0; return 0; }   // no neighbour probes
#pragma unroll
  for (int k = 1; k < 7; k++) vx[k] = table_fetch(t, keys[k], ok[k]);
  return tsdf_query_core(vx, voxel_size, min_weight, value, g);
}

// Voxel-sharded volumes (owner-computes query): every rank looks the seven voxels of every point up in ITS
// table and writes (state, weight, tsdf) - zeros for voxels it does not hold; a voxel lives on exactly one rank, so the
// all-reduce(sum) of these arrays is, bit for bit, what one table holding everything would return.
__global__ __launch_bounds__(256) void tsdf_lookup7_kernel(void* base, uint64_t cap, const float* __restrict__ pts,
                                                           int n, const float* __restrict__ pose, float vs,
                                                           double* __restrict__ out) {
  TsdfTable t = table_carve(base, cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  if (pose) { const Sim3f T = sim3_load(pose); sim3_act(T, p, p); }
  uint64_t keys[7];
  bool ok[7];
  voxel_keys7(p[0], p[1], p[2], vs, keys, ok);
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const VoxelVal v = table_fetch(t, keys[k], ok[k]);
    double* o = out + ((size_t)i * 7 + k) * 3;
    o[0] = (double)v.state; o[1] = v.state ? v.w : 0.0; o[2] = v.state ? v.v : 0.0;
  }
}

__device__ __forceinline__ void lookup_load(const double* __restrict__ lk, int i, VoxelVal (&vx)[7]) {
#pragma unroll
  for (int k = 0; k < 7; k++) {
    const double* o = lk + ((size_t)i * 7 + k) * 3;
    vx[k].state = (int)o[0]; vx[k].w = o[1]; vx[k].v = o[2];
  }
}

__global__ __launch_bounds__(256) void tsdf_query_lookup_kernel(const double* __restrict__ lk, int n, double voxel_size,
                                                                double min_weight, double* __restrict__ value,
                                                                double* __restrict__ grad, uint8_t* __restrict__ status) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  VoxelVal vx[7];
  lookup_load(lk, i, vx);
  double v, g[3];
  const int st = tsdf_query_core(vx, voxel_size, min_weight, v, g);
  value[i] = v; grad[3 * i] = g[0]; grad[3 * i + 1] = g[1]; grad[3 * i + 2] = g[2];
  status[i] = (uint8_t)st;
}

__global__ __launch_bounds__(256) void tsdf_query_kernel(void* base, uint64_t cap, const float* __restrict__ pts,
                                                         int n, float vs, double voxel_size, double min_weight,
                                                         double* __restrict__ value, double* __restrict__ grad,
                                                         uint8_t* __restrict__ status) {
  TsdfTable t = table_carve(base, cap);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double v, g[3];
  const int st = tsdf_query_one(t, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], vs, voxel_size, min_weight, v, g);
  value[i] = v; grad[3 * i] = g[0]; grad[3 * i + 1] = g[1]; grad[3 * i + 2] = g[2];
  status[i] = (uint8_t)st;
}

// TSDF pose normal equations, fp64, deterministic two-stage reduction (tsdf_optimizer.py:94-116).
// pts are CAMERA-frame points when pose != nullptr (world = pose.act(p), computed in fp32 like
// lietorch's act), else already world points.  partial: [gridDim.x][36] = 28 (H lower) + 7 (b) + 1 (count)
// `lookup` != nullptr: the seven voxels of point i come from lookup[i] (all-reduced tsdf_lookup7 output of a voxel-sharded
// volume) instead of the table.
__global__ __launch_bounds__(256) void tsdf_pose_accum_kernel(void* base, uint64_t cap, const float* __restrict__ pts,
                                                              const float* __restrict__ conf, int n,
                                                              const float* __restrict__ pose, float vs,
                                                              double voxel_size, double min_weight, float lambda,
                                                              const double* __restrict__ lookup,
                                                              double* __restrict__ partial) {
  TsdfTable t = table_carve(base, cap);
  double acc[36];
#pragma unroll
  for (int l = 0; l < 36; l++) acc[l] = 0.0;
  Sim3f T;
  if (pose) T = sim3_load(pose);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    if (pose) sim3_act(T, p, p);
    double v, g[3];
    int st;
    if (lookup) {
      VoxelVal vx[7];
      lookup_load(lookup, i, vx);
      st = tsdf_query_core(vx, voxel_size, min_weight, v, g);
    } else {
      st = tsdf_query_one(t, p[0], p[1], p[2], vs, voxel_size, min_weight, v, g);
    }
    if (st != 2) continue;
    if (!isfinite(v)) continue;
    const double P[3] = {p[0], p[1], p[2]};
    double J[7];
    J[0] = g[0]; J[1] = g[1]; J[2] = g[2];
    J[3] = -(P[1] * g[2] - P[2] * g[1]);
    J[4] = -(P[2] * g[0] - P[0] * g[2]);
    J[5] = -(P[0] * g[1] - P[1] * g[0]);
    J[6] = P[0] * g[0] + P[1] * g[1] + P[2] * g[2];
    const float wf = lambda * conf[i];
    const double sw = sqrt(wf > 1.0e-6f ? (double)wf : 1.0e-6);
    double Jw[7];
#pragma unroll
    for (int k = 0; k < 7; k++) Jw[k] = sw * J[k];
    int l = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) {
#pragma unroll
      for (int m = 0; m <= k; m++) acc[l++] += Jw[k] * Jw[m];
    }
#pragma unroll
    for (int k = 0; k < 7; k++) acc[28 + k] += Jw[k] * v * sw;
    acc[35] += 1.0;
  }
  __shared__ double red[256];
  for (int l = 0; l < 36; l++) {
    red[threadIdx.x] = acc[l];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 36 + l] = red[0];
    __syncthreads();
  }
}

// combine partials; optionally solve (H + damping I) delta = -b (Gaussian elimination with partial
// pivoting, np.linalg.solve) and left-multiply pose by exp(delta) (tsdf_optimizer.py:80-86).
__global__ void tsdf_pose_finish_kernel(const double* __restrict__ partial, int nblk, double damping,
                                        double* __restrict__ H_out, double* __restrict__ b_out,
                                        int* __restrict__ used_out, float* __restrict__ pose, int do_update) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[36];
  for (int l = 0; l < 36; l++) {
    double a = 0.0;
    for (int k = 0; k < nblk; k++) a += partial[(size_t)k * 36 + l];
    s[l] = a;
  }
  double H[7][7], b[7];
  int l = 0;
  for (int k = 0; k < 7; k++)
    for (int m = 0; m <= k; m++) { H[k][m] = s[l]; H[m][k] = s[l]; l++; }
  for (int k = 0; k < 7; k++) b[k] = s[28 + k];
  const int used = (int)s[35];
  if (H_out) for (int k = 0; k < 49; k++) H_out[k] = H[k / 7][k % 7];
  if (b_out) for (int k = 0; k < 7; k++) b_out[k] = b[k];
  if (used_out) *used_out = used;
  if (!do_update || used < 6) return;  // `if len(residuals) < 6: break`
  double A[7][8];
  for (int k = 0; k < 7; k++) {
    for (int m = 0; m < 7; m++) A[k][m] = H[k][m] + (k == m ? damping : 0.0);
    A[k][7] = -b[k];
  }
  for (int c = 0; c < 7; c++) {
    int piv = c;
    for (int r = c + 1; r < 7; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (A[piv][c] == 0.0) return;  // singular: LinAlgError -> break
    if (piv != c) for (int m = 0; m < 8; m++) { const double tmp = A[c][m]; A[c][m] = A[piv][m]; A[piv][m] = tmp; }
    for (int r = c + 1; r < 7; r++) {
      const double f = A[r][c] / A[c][c];
      for (int m = c; m < 8; m++) A[r][m] -= f * A[c][m];
    }
  }
  float delta[7];
  // back substitution in fp64; delta is rounded to fp32 once per component, as
  // torch.from_numpy(delta).to(dtype=float32) does (tsdf_optimizer.py:85).
  double xd[7];
  for (int r = 6; r >= 0; r--) {
    double x = A[r][7];
    for (int m = r + 1; m < 7; m++) x -= A[r][m] * xd[m];
    xd[r] = x / A[r][r];
  }
  for (int r = 0; r < 7; r++) delta[r] = (float)xd[r];
  sim3_store(pose, sim3_unit(sim3_retr(delta, sim3_load(pose))));   // lietorch.Sim3.exp(delta) * pose (tsdf_optimizer.py:84-86)
}

__global__ __launch_bounds__(256) void tsdf_dump_kernel(void* base, uint64_t cap, int64_t* __restrict__ keys,
                                                        double* __restrict__ tsdf, double* __restrict__ weight,
                                                        uint32_t max_out) {
  TsdfTable t = table_carve(base, cap);
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  if (t.keys[i] == kEmptyKey || t.state[i] == 0) return;
  const uint32_t o = atomicAdd(&t.hdr->dump_cursor, 1u);
  if (o >= max_out) return;
  long long x, y, z;
  unpack_key(t.keys[i], x, y, z);
  keys[3 * (size_t)o] = x; keys[3 * (size_t)o + 1] = y; keys[3 * (size_t)o + 2] = z;
  tsdf[o] = t.tsdf[i]; weight[o] = t.weight[i];
}

__global__ void tsdf_dump_begin_kernel(TsdfHeader* h) { h->dump_cursor = 0; }

static int max_band_for(double voxel_size, double trunc, double step_scale) {
  double step = voxel_size * step_scale;
  if (step < 1.0e-4) step = 1.0e-4;
  return (int)(2.0 * trunc / step) + 4;
}

}  // namespace mslam

using namespace mslam;

extern "C" size_t mslam_tsdf_table_bytes(uint64_t capacity) {
  if (capacity == 0 || (capacity & (capacity - 1)) != 0) return 0;
  return table_bytes(capacity);
}

extern "C" int mslam_tsdf_table_init(void* table, size_t table_bytes_, uint64_t capacity, void* stream) {
  MSLAM_REQUIRE(table, "tsdf_table_init: null table");
  MSLAM_REQUIRE(capacity >= 1024 && (capacity & (capacity - 1)) == 0 && capacity <= (1ull << 32),
                "tsdf_table_init: capacity must be a power of two in [2^10, 2^32]");
  MSLAM_REQUIRE(table_bytes_ >= table_bytes(capacity), "tsdf_table_init: buffer too small for %llu slots",
                (unsigned long long)capacity);
  hipLaunchKernelGGL(tsdf_init_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     table, capacity);
  MSLAM_LAUNCH_CHECK("tsdf_table_init");
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_rehash(void* old_table, uint64_t old_capacity, void* new_table, uint64_t new_capacity,
                                 void* stream) {
  MSLAM_REQUIRE(old_table && new_table && old_table != new_table, "tsdf_rehash: bad tables");
  MSLAM_REQUIRE(new_capacity >= old_capacity && (new_capacity & (new_capacity - 1)) == 0 &&
                    (old_capacity & (old_capacity - 1)) == 0,
                "tsdf_rehash: capacities must be powers of two, new >= old");
  hipLaunchKernelGGL(tsdf_rehash_kernel, dim3((unsigned)((old_capacity + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, old_table, old_capacity, new_table, new_capacity);
  MSLAM_LAUNCH_CHECK("tsdf_rehash");
  return MSLAM_OK;
}

extern "C" size_t mslam_tsdf_integrate_workspace_bytes(int n_points, double voxel_size, double trunc,
                                                       double step_scale) {
  if (n_points <= 0) return 256;
  const size_t max_rec = (size_t)n_points * (size_t)max_band_for(voxel_size, trunc, step_scale);
  return scratch_carve(nullptr, max_rec).bytes;
}

extern "C" int mslam_tsdf_integrate(void* table, uint64_t capacity, const float* points_world, const double* conf,
                                    const float* cam_origin, int n_points, double voxel_size, double trunc,
                                    double max_weight, double step_scale, int shard_id, int num_shards,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(n_points >= 0, "tsdf_integrate: negative point count");
  if (n_points == 0) return MSLAM_OK;
  MSLAM_REQUIRE(table && points_world && conf && cam_origin && workspace, "tsdf_integrate: null pointer");
  MSLAM_REQUIRE(num_shards >= 1 && shard_id >= 0 && shard_id < num_shards, "tsdf_integrate: bad shard %d/%d",
                shard_id, num_shards);
  MSLAM_REQUIRE(voxel_size > 0 && trunc > 0, "tsdf_integrate: voxel_size and trunc must be positive");
  const int max_band = max_band_for(voxel_size, trunc, step_scale);
  const size_t max_rec = (size_t)n_points * (size_t)max_band;
  MSLAM_REQUIRE(max_rec < 0xFFFFFFFFull, "tsdf_integrate: %zu sample records exceed 2^32", max_rec);
  IntegrateScratch S = scratch_carve(workspace, max_rec);
  if (S.bytes > workspace_bytes) {
    set_error("tsdf_integrate: workspace too small (%zu < %zu)", workspace_bytes, S.bytes);
    return MSLAM_ENOMEM;
  }
  hipStream_t s = (hipStream_t)stream;
  double step = voxel_size * step_scale;
  if (step < 1.0e-4) step = 1.0e-4;
  TsdfTable t = table_carve(table, capacity);
  hipLaunchKernelGGL(tsdf_begin_kernel, dim3(1), dim3(1), 0, s, t.hdr);
  hipLaunchKernelGGL(tsdf_emit_kernel, dim3((n_points + 255) / 256), dim3(256), 0, s, table, capacity, points_world,
                     conf, cam_origin, n_points, (float)voxel_size, (float)step, (float)trunc, max_band, shard_id,
                     num_shards, S, (uint32_t)max_rec);
  const unsigned rec_blocks = (unsigned)((max_rec + 255) / 256);
  hipLaunchKernelGGL(tsdf_alloc_kernel, dim3(rec_blocks), dim3(256), 0, s, table, capacity, S);
  hipLaunchKernelGGL(tsdf_scatter_kernel, dim3(rec_blocks), dim3(256), 0, s, table, capacity, S, (uint32_t)max_rec);
  hipLaunchKernelGGL(tsdf_replay_kernel, dim3(rec_blocks), dim3(256), 0, s, table, capacity, S, max_weight);
  hipLaunchKernelGGL(tsdf_replay_big_kernel, dim3(2048), dim3(64), 0, s, table, capacity, S, max_weight);
  MSLAM_LAUNCH_CHECK("tsdf_integrate");
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_header(void* table, uint64_t capacity, uint32_t* out8_host, void* stream) {
  MSLAM_REQUIRE(table && out8_host, "tsdf_header: null pointer");
  TsdfTable t = table_carve(table, capacity);
  TsdfHeader h;
  int rc = check_hip(hipMemcpyAsync(&h, t.hdr, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream), "tsdf_header");
  if (rc) return rc;
  rc = check_hip(hipStreamSynchronize((hipStream_t)stream), "tsdf_header sync");
  if (rc) return rc;
  out8_host[0] = h.count; out8_host[1] = h.overflow; out8_host[2] = h.n_records; out8_host[3] = h.n_touched;
  out8_host[4] = h.fused; out8_host[5] = h.dump_cursor; out8_host[6] = (uint32_t)(h.capacity & 0xFFFFFFFFu);
  out8_host[7] = (uint32_t)(h.capacity >> 32);
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_dump(void* table, uint64_t capacity, int64_t* keys, double* tsdf, double* weight,
                               uint32_t max_out, void* stream) {
  MSLAM_REQUIRE(table && keys && tsdf && weight, "tsdf_dump: null pointer");
  TsdfTable t = table_carve(table, capacity);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tsdf_dump_begin_kernel, dim3(1), dim3(1), 0, s, t.hdr);
  hipLaunchKernelGGL(tsdf_dump_kernel, dim3((unsigned)((capacity + 255) / 256)), dim3(256), 0, s, table, capacity,
                     keys, tsdf, weight, max_out);
  MSLAM_LAUNCH_CHECK("tsdf_dump");
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_query(void* table, uint64_t capacity, const float* points, int n, double voxel_size,
                                double min_weight, double* value, double* grad, uint8_t* status, void* stream) {
  MSLAM_REQUIRE(n >= 0, "tsdf_query: negative count");
  if (n == 0) return MSLAM_OK;
  MSLAM_REQUIRE(table && points && value && grad && status, "tsdf_query: null pointer");
  hipLaunchKernelGGL(tsdf_query_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, capacity,
                     points, n, (float)voxel_size, voxel_size, min_weight, value, grad, status);
  MSLAM_LAUNCH_CHECK("tsdf_query");
  return MSLAM_OK;
}

static int pose_step_impl(void* table, uint64_t capacity, const double* lookup, const float* points, const float* conf,
                          int n, float* pose, int points_in_camera_frame, double voxel_size, double min_weight,
                          double lambda, double damping, int update_pose, double* H_out, double* b_out, int* used_out,
                          void* workspace, size_t workspace_bytes, void* stream, const char* what) {
  MSLAM_REQUIRE(n >= 0, "%s: negative count", what);
  MSLAM_REQUIRE((table || lookup) && workspace && (n == 0 || (points && conf)), "%s: null pointer", what);
  MSLAM_REQUIRE(!(update_pose || points_in_camera_frame) || pose, "%s: pose required", what);
  int nblk = (n + 255) / 256;
  if (nblk > 64) nblk = 64;
  if (nblk < 1) nblk = 1;
  MSLAM_REQUIRE(workspace_bytes >= sizeof(double) * 36 * 64, "%s: workspace needs %zu bytes", what,
                sizeof(double) * 36 * 64);
  double* partial = (double*)workspace;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tsdf_pose_accum_kernel, dim3(nblk), dim3(256), 0, s, table, capacity, points, conf, n,
                     points_in_camera_frame ? pose : (const float*)nullptr, (float)voxel_size, voxel_size, min_weight,
                     (float)lambda, lookup, partial);
  hipLaunchKernelGGL(tsdf_pose_finish_kernel, dim3(1), dim3(64), 0, s, partial, nblk, damping, H_out, b_out, used_out,
                     pose, update_pose);
  return check_hip(hipGetLastError(), what);
}

extern "C" int mslam_tsdf_pose_step(void* table, uint64_t capacity, const float* points, const float* conf, int n,
                                    float* pose, int points_in_camera_frame, double voxel_size, double min_weight,
                                    double lambda, double damping, int update_pose, double* H_out, double* b_out,
                                    int* used_out, void* workspace, size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(table, "tsdf_pose_step: null table");
  return pose_step_impl(table, capacity, nullptr, points, conf, n, pose, points_in_camera_frame, voxel_size, min_weight,
                        lambda, damping, update_pose, H_out, b_out, used_out, workspace, workspace_bytes, stream,
                        "tsdf_pose_step");
}

extern "C" int mslam_tsdf_lookup7(void* table, uint64_t capacity, const float* points, int n, const float* pose,
                                  double voxel_size, double* out, void* stream) {
  MSLAM_REQUIRE(n >= 0, "tsdf_lookup7: negative count");
  if (n == 0) return MSLAM_OK;
  MSLAM_REQUIRE(table && points && out, "tsdf_lookup7: null pointer");
  hipLaunchKernelGGL(tsdf_lookup7_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, capacity,
                     points, n, pose, (float)voxel_size, out);
  MSLAM_LAUNCH_CHECK("tsdf_lookup7");
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_query_lookup(const double* lookup, int n, double voxel_size, double min_weight, double* value,
                                       double* grad, uint8_t* status, void* stream) {
  MSLAM_REQUIRE(n >= 0, "tsdf_query_lookup: negative count");
  if (n == 0) return MSLAM_OK;
  MSLAM_REQUIRE(lookup && value && grad && status, "tsdf_query_lookup: null pointer");
  hipLaunchKernelGGL(tsdf_query_lookup_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, lookup, n,
                     voxel_size, min_weight, value, grad, status);
  MSLAM_LAUNCH_CHECK("tsdf_query_lookup");
  return MSLAM_OK;
}

extern "C" int mslam_tsdf_pose_step_lookup(const double* lookup, const float* points, const float* conf, int n,
                                           float* pose, int points_in_camera_frame, double voxel_size,
                                           double min_weight, double lambda, double damping, int update_pose,
                                           double* H_out, double* b_out, int* used_out, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  MSLAM_REQUIRE(lookup, "tsdf_pose_step_lookup: null lookup");
  return pose_step_impl(nullptr, 1024, lookup, points, conf, n, pose, points_in_camera_frame, voxel_size, min_weight,
                        lambda, damping, update_pose, H_out, b_out, used_out, workspace, workspace_bytes, stream,
                        "tsdf_pose_step_lookup");
}
