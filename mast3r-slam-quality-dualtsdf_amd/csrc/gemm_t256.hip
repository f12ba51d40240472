// 256-row block tiles for the large-M shapes (backend edge batches, DPT convolutions): 256x128 with
// 8 waves and 256x256 with 16 waves, every wave a 64x64 sub-tile.
#include "gemm_kernel.h"
namespace mslam {
int launch_gemm_t256(const GemmArgs& a, int bn, hipStream_t s) {
  if (bn == 256) return launch_cfg<4, 4, 2, 2, 2>(a, s);
  return launch_cfg<4, 2, 2, 2, 2>(a, s);
}
}  // namespace mslam
