// 128x128 block tiles: 4 waves of 64x64 (fewest LDS reads per MFMA) or 8 waves of 64x32 (more DMA
// streams per CU).
#include "gemm_kernel.h"
namespace mslam {
// 128x64: 4 waves of 64x32; 25 % less L2->LDS traffic per flop than 64x64 and still 3 blocks per CU
int launch_gemm_t128x64(const GemmArgs& a, int stages, hipStream_t s) {
  return stages >= 3 ? launch_cfg<2, 2, 2, 1, 3>(a, s) : launch_cfg<2, 2, 2, 1, 2>(a, s);
}
int launch_gemm_t128(const GemmArgs& a, int waves, int stages, hipStream_t s) {
  if (waves == 8) return stages >= 3 ? launch_cfg<2, 4, 2, 1, 3>(a, s) : launch_cfg<2, 4, 2, 1, 2>(a, s);
  return launch_cfg<2, 2, 2, 2, 2>(a, s);
}
}  // namespace mslam
