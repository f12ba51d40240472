/*
 * mslam_hip.h — C ABI of libmslam_hip.so, the MI355X (gfx950) implementation of the per-frame
 * SLAM compute path of MASt3R-SLAM (dual-TSDF fork).
 *
 * Every entry point takes plain device pointers + sizes (no torch types) and a HIP stream
 * (`void* stream` is a hipStream_t; NULL = the null stream).  All pointers are DEVICE pointers
 * unless the parameter name ends in `_host`.  Every function returns 0 on success or a negative
 * MSLAM_E* code; mslam_last_error() returns a human-readable message for the calling thread.
 * Nothing here allocates device memory or synchronises the stream unless documented, so every
 * call can be captured into a hipGraph.
 *
 * Each declaration cites the reference interface it replaces (paths relative to the reference
 * repository root).  The Python binding a maintainer adds is shown in INTEGRATION.md.
 */
#ifndef MSLAM_HIP_H
#define MSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSLAM_OK 0
#define MSLAM_EINVAL (-1)    /* bad argument (shape / null pointer / unsupported size) */
#define MSLAM_EHIP (-2)      /* HIP runtime error (message in mslam_last_error) */
#define MSLAM_ENOMEM (-3)    /* workspace / table too small */
#define MSLAM_ENODEV (-4)    /* no gfx950 device */

const char* mslam_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int mslam_abi_version(void);
/* Returns 0 when a gfx950 device is present and usable, MSLAM_ENODEV otherwise. */
int mslam_device_check(void);

/* ------------------------------------------------------------------------------------------
 * Matching  (pybind `mast3r_slam_backends`, mast3r_slam/backend/src/gn.cpp:116-123)
 * ------------------------------------------------------------------------------------------ */

/* Replaces iter_proj(rays_img_with_grad, pts_3d_norm, p_init, max_iter, lambda_init, cost_thresh)
 *   binding   mast3r_slam/backend/src/gn.cpp:84-99
 *   launcher  mast3r_slam/backend/src/matching_kernels.cu:279-316
 *   kernel    mast3r_slam/backend/src/matching_kernels.cu:119-275
 * rays_img_with_grad f32[b,h,w,9], pts_3d_norm f32[b,n,3], p_init f32[b,n,2]
 *   -> p_new f32[b,n,2], converged u8[b,n] (torch.bool storage). */
int mslam_iter_proj(const float* rays_img_with_grad, const float* pts_3d_norm,
                    const float* p_init, float* p_new, uint8_t* converged, int b, int h, int w,
                    int n, int max_iter, float lambda_init, float cost_thresh, void* stream);

/* Replaces refine_matches(D11, D21, p1, radius, dilation_max)
 *   binding   mast3r_slam/backend/src/gn.cpp:101-114
 *   launcher  mast3r_slam/backend/src/matching_kernels.cu:84-116
 *   kernel    mast3r_slam/backend/src/matching_kernels.cu:25-81
 * D11 f16[b,h,w,fdim], D21 f16[b,n,fdim] (IEEE binary16 bits), p1 i64[b,n,2] -> p1_new i64[b,n,2]. */
int mslam_refine_matches(const uint16_t* D11, const uint16_t* D21, const int64_t* p1,
                         int64_t* p1_new, int b, int h, int w, int n, int fdim, int radius,
                         int dilation_max, void* stream);

/* Replaces prep_for_iter_proj + img_gradient (mast3r_slam/matching.py:25-49,
 * mast3r_slam/image.py:5-38) as ONE kernel: rays = normalize(X11), Scharr-like x/y gradients with
 * reflect padding, channel concat; pts_3d_norm = normalize(X21); p_init from idx_init (i64[b,n],
 * may be NULL = identity mapping).  X11,X21 f32[b,h,w,3]. */
int mslam_prep_iter_proj(const float* X11, const float* X21, const int64_t* idx_init,
                         float* rays_img_with_grad, float* pts_3d_norm, float* p_init, int b, int h,
                         int w, void* stream);

/* Replaces the occlusion test + pixel_to_lin of match_iterative_proj
 * (mast3r_slam/matching.py:68-76,87-90): p1 = trunc(p) ; valid &= |X11[p1]-X21| < dist_thresh.
 * p f32[b,n,2] -> p1 i64[b,n,2]; valid u8[b,n] in/out (converged flags in, valid_proj2 out). */
int mslam_match_occlusion(const float* X11, const float* X21, const float* p, int64_t* p1,
                          uint8_t* valid, int b, int h, int w, float dist_thresh, void* stream);

/* idx = u + w*v  (mast3r_slam/matching.py:13-15). p1 i64[b,n,2] -> idx i64[b,n]. */
int mslam_pixel_to_lin(const int64_t* p1, int64_t* idx, int b, int n, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSLAM_HIP_H */
