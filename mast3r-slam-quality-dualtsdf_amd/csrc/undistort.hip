// Lens undistortion of the dataset readers: cv2.remap(img, mapx, mapy, cv2.INTER_LINEAR) of
// mast3r_slam/dataloader.py:495-496 (Intrinsics.remap) as a HIP kernel.  OpenCV is not available here, so this follows
// OpenCV's PUBLISHED fixed-point algorithm (modules/imgproc/src/imgwarp.cpp, remapBilinear with CV_32FC1 maps, 8-bit
// images): the float map is rounded to 1/32 pixel (cvRound = round half to even of map * 32), the four bilinear weights
// of a sub-pixel position are the integers (32 - fx)(32 - fy) 32, fx (32 - fy) 32, (32 - fx) fy 32, fx fy 32 (they sum to
// 2^15 exactly, so initInterTab2D's sum correction never fires), the result is (sum + 2^14) >> 15, and taps outside the
// image read the constant border 0 (BORDER_CONSTANT).  Parity with the library is UNPINNED (nothing to compare with);
// tests/test_undistort*.py check the properties: identity maps reproduce the image bit for bit, the kernel equals a
// NumPy restatement of the same integer arithmetic.  HBM-bound: 2 x 4 B map + 4 taps x C bytes per output pixel.
#include "common.h"

namespace mslam {

__global__ __launch_bounds__(256) void remap_bilinear_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int ch,
                                                                const float* __restrict__ mapx,
                                                                const float* __restrict__ mapy,
                                                                uint8_t* __restrict__ dst, int dh, int dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= dh * dw) return;
  // cvRound(v * INTER_TAB_SIZE): round half to even
  const int sx = __float2int_rn(mapx[i] * 32.0f), sy = __float2int_rn(mapy[i] * 32.0f);
  const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const bool x0 = ix >= 0 && ix < sw, x1 = ix + 1 >= 0 && ix + 1 < sw;
  const bool y0 = iy >= 0 && iy < sh, y1 = iy + 1 >= 0 && iy + 1 < sh;
  for (int c = 0; c < ch; c++) {
    const int v00 = (x0 && y0) ? src[((size_t)iy * sw + ix) * ch + c] : 0;
    const int v01 = (x1 && y0) ? src[((size_t)iy * sw + ix + 1) * ch + c] : 0;
    const int v10 = (x0 && y1) ? src[((size_t)(iy + 1) * sw + ix) * ch + c] : 0;
    const int v11 = (x1 && y1) ? src[((size_t)(iy + 1) * sw + ix + 1) * ch + c] : 0;
    const int v = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
    dst[(size_t)i * ch + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

}  // namespace mslam

using namespace mslam;

extern "C" int mslam_remap_bilinear_u8(const uint8_t* src, int src_h, int src_w, int channels, const float* mapx,
                                       const float* mapy, uint8_t* dst, int dst_h, int dst_w, void* stream) {
  MSLAM_REQUIRE(src && mapx && mapy && dst, "remap_bilinear_u8: null pointer");
  MSLAM_REQUIRE(src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && channels > 0 && channels <= 4,
                "remap_bilinear_u8: bad sizes %dx%dx%d -> %dx%d", src_h, src_w, channels, dst_h, dst_w);
  const int n = dst_h * dst_w;
  hipLaunchKernelGGL(remap_bilinear_u8_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, src_h,
                     src_w, channels, mapx, mapy, dst, dst_h, dst_w);
  MSLAM_LAUNCH_CHECK("remap_bilinear_u8");
  return MSLAM_OK;
}
