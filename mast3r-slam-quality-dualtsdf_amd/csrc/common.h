// Shared helpers for the libmslam_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/mslam_hip.h"

namespace mslam {

// Thread-local last-error text returned by mslam_last_error().
void set_error(const char* fmt, ...);

inline int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return MSLAM_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return MSLAM_EHIP;
}

#define MSLAM_REQUIRE(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      mslam::set_error(__VA_ARGS__);  \
      return MSLAM_EINVAL;            \
    }                                 \
  } while (0)

#define MSLAM_LAUNCH_CHECK(name)                                         \
  do {                                                                   \
    int _rc = mslam::check_hip(hipGetLastError(), name " launch");       \
    if (_rc) return _rc;                                                 \
  } while (0)

constexpr int kWave = 64;
constexpr int kNumXcd = 8;

// XCD-aware block remap (bijective for any grid size): blocks that the dispatcher deals
// round-robin to one XCD get a contiguous chunk of logical tile ids, so neighbouring tiles
// share that XCD's L2.  Speed only; correctness never depends on it.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk / kNumXcd, r = nblk % kNumXcd;
  const unsigned xcd = bid % kNumXcd;
  const unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / kNumXcd;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

}  // namespace mslam
