// Host-side dispatch of the bf16 MFMA GEMM (kernel: gemm_kernel.h, instantiations: gemm_t*.hip).
#include <stdlib.h>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "common.h"
#include "gemm.h"

namespace mslam {

int launch_gemm_t64(const GemmArgs& a, int stages, hipStream_t s);
int launch_gemm_t128(const GemmArgs& a, int waves, int stages, hipStream_t s);
int launch_gemm_t128x64(const GemmArgs& a, int stages, hipStream_t s);
int launch_gemm_t256(const GemmArgs& a, int bn, hipStream_t s);
int launch_gemm_t192(const GemmArgs& a, hipStream_t s);
int launch_gemm_t256w8(const GemmArgs& a, hipStream_t s);
int launch_gemm_8p(const GemmArgs& a, hipStream_t s);      // gemm8p.hip: 256x256, phase-interleaved K loop
bool gemm8p_supports(const GemmArgs& a);

// Per-shape tile choices for plain (non-conv, ungrouped) problems: the measured table below, editable at run time
// through mslam_gemm_tile_override (tools/insitu_tune.py finds the entries by timing whole network stages).
static std::mutex g_tile_mu;
static inline uint64_t tile_key(int M, int N, int K, bool conv = false) {   // M < 2^22, N < 2^20, K < 2^20
  return (uint64_t)M | ((uint64_t)N << 22) | ((uint64_t)K << 42) | ((uint64_t)conv << 63);
}
// measured in situ (tools/insitu_tune.py, profiles/r01_insitu_tune.log): shapes where the rule below is not the best
static std::unordered_map<uint64_t, int> g_tile_override = {
    {tile_key(3072, 3072, 1024), 1262},   // encoder qkv at a frame group of 4: 128x64 tiles instead of 144 tiles of 256x256
    {tile_key(3072, 768, 3072), 643},     // decoder fc2 at a frame group of 4: ring of 3
    {tile_key(6144, 3072, 768), 1282},    // decoder fc1 of the batch-8 backend call: 288 tiles of 256x256 leave a half-empty round
    // round 2 (tools/gemm_cfg_ab.py, interleaved A/B in one process, profiles/r02_gemm_cfg_ab.log)
    {tile_key(3072, 4096, 1024), 2192},   // encoder fc1 at a frame group of 4: 16 x 16 tiles of 192x256 = one per CU (256x256: 192 tiles)
    {tile_key(6144, 4096, 1024), 1282},   // the same at a group of 8: 384 tiles of 256x256 are one and a half rounds
    // round 3, encoder batches of 12 / 6 frames (bench default; tools/gemm_cfg_ab.py, gpurun_out -> profiles/r03_gemm_cfg_ab_group12.log)
    {tile_key(9216, 4096, 1024), 1282},   // fc1 + GELU: 96 us against 110 us for 576 tiles of 256x256 (2.25 rounds; the epilogue of a
                                          // 128x128 tile drains beside its neighbours' K loops)
    {tile_key(9216, 1024, 4096), 8256},   // fc2: phase-interleaved kernel, 78 us against 86 us
    {tile_key(4608, 4096, 1024), 1282},   // fc1 + GELU at 6 frames: 54 us against 70 us
};

int gemm_tile_override(int M, int N, int K, int cfg, bool conv) {
  std::lock_guard<std::mutex> lk(g_tile_mu);
  if (cfg > 0) g_tile_override[tile_key(M, N, K, conv)] = cfg;
  else g_tile_override.erase(tile_key(M, N, K, conv));
  return 0;
}

// Live per-launch timing of ONE shape (bench.py's `roofline` object): while enabled, every launch of the chosen plain
// (M, N, K) problem is bracketed by two HIP events on the stream it is launched on; read back after the timed region.
static std::mutex g_prof_mu;
static struct {
  bool on = false;
  int M = 0, N = 0, K = 0, cap = 0, n = 0;
  std::vector<hipEvent_t> ev;   // 2 * cap
} g_prof;

int gemm_profile_begin(int M, int N, int K, int max_samples) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  MSLAM_REQUIRE(!g_prof.on, "gemm_profile_begin: already profiling");
  MSLAM_REQUIRE(max_samples > 0 && max_samples <= 65536, "gemm_profile_begin: max_samples out of range");
  g_prof.ev.resize(2 * (size_t)max_samples);
  for (auto& e : g_prof.ev) {
    int rc = check_hip(hipEventCreate(&e), "hipEventCreate");
    if (rc) return rc;
  }
  g_prof.M = M; g_prof.N = N; g_prof.K = K; g_prof.cap = max_samples; g_prof.n = 0;
  g_prof.on = true;
  return MSLAM_OK;
}

int gemm_profile_end(double* avg_us, double* min_us, int* samples) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  MSLAM_REQUIRE(g_prof.on, "gemm_profile_end: not profiling");
  g_prof.on = false;
  double sum = 0.0, mn = 1e30;
  int rc = MSLAM_OK;
  for (int i = 0; i < g_prof.n && !rc; i++) {
    float ms = 0.0f;
    rc = check_hip(hipEventSynchronize(g_prof.ev[2 * i + 1]), "hipEventSynchronize");
    if (!rc) rc = check_hip(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]), "hipEventElapsedTime");
    sum += 1e3 * ms;
    if (1e3 * ms < mn) mn = 1e3 * ms;
  }
  if (samples) *samples = g_prof.n;
  if (avg_us) *avg_us = g_prof.n ? sum / g_prof.n : 0.0;
  if (min_us) *min_us = g_prof.n ? mn : 0.0;
  for (auto& e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  return rc;
}

static int launch_gemm_impl(const GemmArgs& a, hipStream_t stream);

int launch_gemm(const GemmArgs& a, hipStream_t stream) {
  int slot = -1;
  if (g_prof.on) {   // racy read is fine: begin/end are called with the GPU idle
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof.on && !a.a_conv && a.groups <= 1 && a.M == g_prof.M && a.N == g_prof.N && a.K == g_prof.K &&
        g_prof.n < g_prof.cap) {
      slot = g_prof.n++;
      (void)hipEventRecord(g_prof.ev[2 * slot], stream);
    }
  }
  const int rc = launch_gemm_impl(a, stream);
  if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], stream);
  return rc;
}

static int launch_gemm_impl(const GemmArgs& a, hipStream_t stream) {
  MSLAM_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem %dx%dx%d", a.M, a.N, a.K);
  MSLAM_REQUIRE(a.K % 8 == 0, "gemm: K=%d must be a multiple of 8", a.K);
  MSLAM_REQUIRE(!a.a_conv || a.cC % 8 == 0, "gemm: conv channels %d must be a multiple of 8", a.cC);
  MSLAM_REQUIRE(a.a_conv || a.lda % 8 == 0, "gemm: lda=%d must be a multiple of 8", a.lda);
  MSLAM_REQUIRE(!a.a_relu || a.a_conv, "gemm: ReLU-on-load is only built for the implicit-conv view");
  MSLAM_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.W & 15) == 0, "gemm: A and W must be 16-byte aligned");
  const size_t a_bytes = a.a_conv ? (size_t)a.cB * a.cH * a.cW * a.cC * 2 : ((size_t)(a.M - 1) * a.lda + a.K) * 2;
  MSLAM_REQUIRE(a_bytes < (1ull << 31) && (size_t)a.N * a.K * 2 < (1ull << 31), "gemm: operand larger than 2 GiB");
  MSLAM_REQUIRE(a.epi != EPI_ATTN || (a.sec_dim % 64 == 0 && a.ntok % 4 == 0 && a.kv_ntok % 4 == 0),
                "gemm: attention epilogue needs 64-wide heads and token counts divisible by 4");
  // Tile selection (measured on MI355X, tools/gemm_tune.py).  What a CU can pull into LDS grows with the
  // number of waves issuing DMA, not with the ring depth, so every configuration keeps several blocks
  // (or many waves) per CU: shallow rings, 32-64 KiB of LDS per 4 waves.  Larger tiles halve the L2->LDS
  // traffic and win as soon as they still cover the chip.
  // MSLAM_GEMM="<cfg>" forces one configuration for experiments:
  //   642/643/644: 64x64 ring 2/3/4; 1262/1263: 128x64 ring 2/3; 1242: 128x128 4 waves; 1282/1283: 128x128 8 waves ring 2/3;
  //   8256: 256x256 8 waves, phase-interleaved K loop (gemm8p.hip; plain epilogue only, else 2256);
  //   2128: 256x128 8 waves; 2256: 256x256 16 waves; 2192: 192x256 8 waves; 2258: 256x256 8 waves (A/B only: slower than
  //   2256 and 2192 on every shape of the table, profiles/r02_gemm_cfg_ab.log)
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("MSLAM_GEMM");
    forced = e ? atoi(e) : -1;
  }
  MSLAM_REQUIRE(a.groups <= 2 && (a.groups < 2 || (a.W1 && !a.a_conv)), "gemm: bad group description");
  const long ngrp = a.groups > 1 ? 2 : 1;
  auto blocks = [&](int bm, int bn) { return ngrp * ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  int cfg = forced;
  if (cfg < 0 && a.groups <= 1 && a.M < (1 << 22) && a.N < (1 << 20) && a.K < (1 << 20)) {
    static const bool log_shapes = getenv("MSLAM_GEMM_LOG") != nullptr;   // one stderr line per new shape (tuning)
    static std::unordered_map<uint64_t, int> seen;
    std::lock_guard<std::mutex> lk(g_tile_mu);
    const uint64_t key = tile_key(a.M, a.N, a.K, a.a_conv != 0);
    auto it = g_tile_override.find(key);
    if (it != g_tile_override.end()) cfg = it->second;
    if (log_shapes && seen.emplace(key, 1).second) fprintf(stderr, "mslam_gemm_shape conv=%d %d %d %d\n", (int)(a.a_conv != 0), a.M, a.N, a.K);
  }
  if (cfg < 0) {
    const bool narrow = a.N <= 128 || (a.N > 256 && a.N <= 384);   // a 256-wide tile would be >= 25 % padding
    if (narrow && blocks(256, 128) >= 128) cfg = 2128;
    else if (blocks(256, 256) >= 128) cfg = 2256;
    else if (blocks(128, 128) >= 256) cfg = 1282;   // one block per CU at least (same-box A/B in situ: tools/batch_time.py)
    else if (blocks(64, 64) >= 512) cfg = 642;
    else if (a.K >= 2048) cfg = 644;
    else cfg = 643;
  }
  switch (cfg) {
    case 642: return launch_gemm_t64(a, 2, stream);
    case 643: return launch_gemm_t64(a, 3, stream);
    case 644: return launch_gemm_t64(a, 4, stream);
    case 1262: return launch_gemm_t128x64(a, 2, stream);
    case 1263: return launch_gemm_t128x64(a, 3, stream);
    case 1242: return launch_gemm_t128(a, 4, 2, stream);
    case 1282: return launch_gemm_t128(a, 8, 2, stream);
    case 1283: return launch_gemm_t128(a, 8, 3, stream);
    case 2128: return launch_gemm_t256(a, 128, stream);
    case 2256: return launch_gemm_t256(a, 256, stream);
    case 2192: return launch_gemm_t192(a, stream);
    case 2258: return launch_gemm_t256w8(a, stream);
    case 8256: return gemm8p_supports(a) ? launch_gemm_8p(a, stream) : launch_gemm_t256(a, 256, stream);
    default: MSLAM_REQUIRE(false, "gemm: unknown configuration %d", cfg);
  }
}

}  // namespace mslam
