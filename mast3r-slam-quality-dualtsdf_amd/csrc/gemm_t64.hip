// 64x64 block tiles (4 waves, one 32x32 MFMA tile each): the small-M shapes of the tracked frame.
#include "gemm_kernel.h"
namespace mslam {
int launch_gemm_t64(const GemmArgs& a, int stages, hipStream_t s) {
  switch (stages) {
    case 2: return launch_cfg<2, 2, 1, 1, 2>(a, s);
    case 3: return launch_cfg<2, 2, 1, 1, 3>(a, s);
    default: return launch_cfg<2, 2, 1, 1, 4>(a, s);
  }
}
}  // namespace mslam
