// 256-row block tiles for the large-M shapes (backend edge batches, DPT convolutions): 256x128 with
// 8 waves and 256x256 with 16 waves, every wave a 64x64 sub-tile.
#include "gemm_kernel.h"
namespace mslam {
// 192x256 with 8 waves (every wave 96x64): 3072 x 4096 (the encoder's fc1 at a frame group of 4) is 16 x 16 = 256 of
// these - one per CU - where 256x256 tiles leave a quarter of the chip idle (12 x 16 = 192).
int launch_gemm_t192(const GemmArgs& a, hipStream_t s) { return launch_cfg<2, 4, 3, 2, 2>(a, s); }

// 256x256 with 8 waves, every wave 128x64: 25 % fewer LDS fragment bytes per MFMA than the 16-wave form (6 fragments
// per 8 MFMAs instead of 4 per 4)
int launch_gemm_t256w8(const GemmArgs& a, hipStream_t s) { return launch_cfg<2, 4, 4, 2, 2>(a, s); }

int launch_gemm_t256(const GemmArgs& a, int bn, hipStream_t s) {
  if (bn == 256) return launch_cfg<4, 4, 2, 2, 2>(a, s);
  return launch_cfg<4, 2, 2, 2, 2>(a, s);
}
}  // namespace mslam
