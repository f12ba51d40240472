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

/* ------------------------------------------------------------------------------------------
 * Gauss-Newton backend  (pybind gauss_newton_{rays,calib,points}, gn.cpp:3-82;
 *                        host loops gn_kernels.cu:725-811, 1140-1228, 1546-1638)
 *
 * Shapes: Twc f32[P,8] ([t,q(xyzw),s], UPDATED IN PLACE for rows >= 1), Xs f32[P,HW,3],
 * Cs f32[P,HW,1], ii/jj i64[E] (global keyframe ids; the callee maps them to rows with
 * unique+searchsorted and pins the first unique id, gn_kernels.cu:161-170,1157),
 * idx_ii2jj i64[E,HW], valid_match u8[E,HW,1], Q f32[E,HW,1], K f32[3,3] (device), dx f32[P-1,7].
 * `workspace` is a caller-owned device buffer of >= mslam_gn_workspace_bytes(P,E,HW,E_local) bytes
 * (E_local = the edges this caller compacts and accumulates: E for the fused entry points, the size of
 * the rank's edge slice in the sharded loop).  No cap on P: the fp64 normal equations are a dense
 * (7(P-1))^2 matrix in the workspace, factored by a blocked LL^T on the f64 matrix cores.
 * The whole GN loop is enqueued on `stream` with NO host synchronisation: convergence
 * (||dx|| < delta_thresh, gn_kernels.cu:1219-1222) is a device-side flag that turns the remaining
 * iterations' kernels into no-ops.  LLT failure => dx = 0 (gn_kernels.cu:147-150).
 * ------------------------------------------------------------------------------------------ */
size_t mslam_gn_workspace_bytes(int num_poses, int num_edges, int num_points, int local_edges);

/* Replaces gauss_newton_rays (gn.cpp:28-52; ray_align_kernel gn_kernels.cu:813-1138). */
int mslam_gauss_newton_rays(float* Twc, const float* Xs, const float* Cs, const int64_t* ii,
                            const int64_t* jj, const int64_t* idx_ii2jj, const uint8_t* valid_match,
                            const float* Q, int num_poses, int num_points, int num_edges,
                            float sigma_ray, float sigma_dist, float C_thresh, float Q_thresh,
                            int max_iter, float delta_thresh, float* dx, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Replaces gauss_newton_calib (gn.cpp:54-82; calib_proj_kernel gn_kernels.cu:1231-1543). */
int mslam_gauss_newton_calib(float* Twc, const float* Xs, const float* Cs, const float* K,
                             const int64_t* ii, const int64_t* jj, const int64_t* idx_ii2jj,
                             const uint8_t* valid_match, const float* Q, int num_poses, int num_points,
                             int num_edges, int height, int width, int pixel_border, float z_eps,
                             float sigma_pixel, float sigma_depth, float C_thresh, float Q_thresh,
                             int max_iter, float delta_thresh, float* dx, void* workspace,
                             size_t workspace_bytes, void* stream);

/* Replaces gauss_newton_points (gn.cpp:3-26; point_align_kernel gn_kernels.cu:455-723). */
int mslam_gauss_newton_points(float* Twc, const float* Xs, const float* Cs, const int64_t* ii,
                              const int64_t* jj, const int64_t* idx_ii2jj, const uint8_t* valid_match,
                              const float* Q, int num_poses, int num_points, int num_edges,
                              float sigma_point, float C_thresh, float Q_thresh, int max_iter,
                              float delta_thresh, float* dx, void* workspace, size_t workspace_bytes,
                              void* stream);

/* The same loop opened up for the multi-GPU factor graph (global_opt.py:123-223): every rank calls
 * begin() with the FULL edge list, then compact() ONCE for ITS edge range (per-edge inputs are local
 * arrays of edge_count rows): the pose-independent part of the reference's edge kernels - the gather
 * Xi[idx] and the gates valid_match, Q > Q_thresh, C > C_thresh (gn_kernels.cu:905-925) - is resolved
 * into a dense stream in the workspace.  Per iteration: accumulate() streams it at the current poses
 * into the global, reference-layout buffers Hs f32[4,E,7,7] / gs f32[2,E,7] ([ii,ij,ji,jj] / [i,j],
 * gn_kernels.cu:1120-1133; slots of other ranks' edges stay zero), the caller all-reduces Hs and gs
 * (RCCL sum), then solve_retract(), which is bit-identical on every rank.
 * kind: 0 rays (sigma_a=ray, sigma_b=dist), 1 calib (pixel, depth), 2 points (point, -). */
int mslam_gn_begin(const int64_t* ii, const int64_t* jj, int num_poses, int num_edges, int num_points,
                   void* workspace, size_t workspace_bytes, void* stream);
int mslam_gn_compact(const float* Xs, const float* Cs, const int64_t* idx_ii2jj, const uint8_t* valid_match,
                     const float* Q, int num_poses, int num_points, int num_edges, int edge_begin,
                     int edge_count, float C_thresh, float Q_thresh, void* workspace, size_t workspace_bytes,
                     void* stream);
/* compact() for edge ranges whose per-edge inputs do not lie in ONE array: the factor graph keeps the forward and the
 * backward direction of its edges in two row-appendable buffers (global_opt.py:106-112 concatenates them for every
 * solve: O(edges) copies of [E, HW] arrays per keyframe); a rank's accumulate range of `range_count` edges is compacted
 * by one call per source array, each filling the slots [slot_begin, slot_begin + edge_count) of that range. */
int mslam_gn_compact_at(const float* Xs, const float* Cs, const int64_t* idx_ii2jj, const uint8_t* valid_match,
                        const float* Q, int num_poses, int num_points, int num_edges, int edge_begin, int edge_count,
                        int slot_begin, int range_count, float C_thresh, float Q_thresh, void* workspace,
                        size_t workspace_bytes, void* stream);
int mslam_gn_accumulate(int kind, const float* Twc, const float* K, int num_poses, int num_points,
                        int num_edges, int edge_begin, int edge_count, float sigma_a, float sigma_b,
                        int height, int width, int pixel_border, float z_eps, float* Hs, float* gs,
                        void* workspace, size_t workspace_bytes, void* stream);
int mslam_gn_solve_retract(const float* Hs, const float* gs, int num_poses, int num_edges,
                           int num_points, float* Twc, float* dx, float delta_thresh, void* workspace,
                           size_t workspace_bytes, void* stream);
/* status4 (device i32[4]) <- {done, iterations run, chol_fail, bits of last ||dx|| (f32)}. */
int mslam_gn_status(int* status4, int num_poses, int num_edges, int num_points, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sim3 group ops (lietorch.Sim3 surface used by the hot path: act / inv / mul / exp / retr;
 * call sites tracker.py:150,232,247,264, geometry.py:45-52, global_manager.py:88-105;
 * maths restated in gn_kernels.cu:177-413).  Pose layout f32[n,8] = [t, q(xyzw), s]; xi f32[n,7]
 * = [tau, phi, sigma].  Forward only.
 * ------------------------------------------------------------------------------------------ */
/* Y[i] = T[pose(i)] . X[i]; pose(i) = i / pts_per_pose, or 0 when broadcast_pose. */
int mslam_sim3_act(const float* T, const float* X, float* Y, int num_poses, long long pts_per_pose,
                   int broadcast_pose, void* stream);
/* op 0: out = A^-1 ; 1: out = A*B ; 2: out = exp(A as xi) ; 3: out = exp(A as xi) * B (retr). */
int mslam_sim3_op(int op, const float* A, const float* B, float* out, int n, int bcast_a, int bcast_b,
                  void* stream);

/* ------------------------------------------------------------------------------------------
 * Global sparse TSDF (world-space half of the "dual TSDF"): GPU voxel hash replacing the python
 * dict of mast3r_slam/tsdf/global_volume.py.  `table` is a caller-owned device buffer of
 * mslam_tsdf_table_bytes(capacity) bytes (capacity = power of two slots); voxel keys are the
 * reference's integer triples floor(p / voxel_size) (global_volume.py:133-134), bit-exact.
 * ------------------------------------------------------------------------------------------ */
size_t mslam_tsdf_table_bytes(uint64_t capacity);
int mslam_tsdf_table_init(void* table, size_t table_bytes, uint64_t capacity, void* stream);

/* Replaces TSDFVolume.integrate (global_volume.py:35-72) + _update_voxel (:74-88).
 * points_world f32[n,3], conf f64[n], cam_origin f32[3] (all device).  Samples are replayed per
 * voxel in the reference's order, so results equal the sequential python loop, including the
 * first-touch and max_weight-saturation behaviour.  shard_id/num_shards: only voxels whose key
 * hashes to this shard are touched (multi-GPU: one table per rank, points replicated). */
size_t mslam_tsdf_integrate_workspace_bytes(int n_points, double voxel_size, double trunc, double step_scale);
int mslam_tsdf_integrate(void* table, uint64_t capacity, const float* points_world, const double* conf,
                         const float* cam_origin, int n_points, double voxel_size, double trunc,
                         double max_weight, double step_scale, int shard_id, int num_shards,
                         void* workspace, size_t workspace_bytes, void* stream);

/* out8_host <- {voxels, overflow flag, records of last integrate, voxels touched, points fused
 * (integrate's return value), dump cursor, capacity lo, capacity hi}.  SYNCHRONISES the stream. */
/* Growth of the voxel hash (the reference's dict is unbounded, global_volume.py:27): moves every voxel of `old_table`
 * into `new_table` (initialised by mslam_tsdf_table_init, capacity >= the old one) with value, weight and first-touch
 * state.  Call between integrate calls; the host side (TSDFVolume.maintain) does so when the load exceeds 1/2. */
int mslam_tsdf_rehash(void* old_table, uint64_t old_capacity, void* new_table, uint64_t new_capacity, void* stream);
int mslam_tsdf_header(void* table, uint64_t capacity, uint32_t* out8_host, void* stream);

/* Dict contents (TSDFVolume._voxels): keys i64[max_out,3], tsdf f64, weight f64, unordered. */
int mslam_tsdf_dump(void* table, uint64_t capacity, int64_t* keys, double* tsdf, double* weight,
                    uint32_t max_out, void* stream);

/* Replaces TSDFVolume.query + _estimate_gradient (global_volume.py:93-128) for n points:
 * status 0 = (None, None), 1 = (value, None), 2 = (value, unit gradient). */
int mslam_tsdf_query(void* table, uint64_t capacity, const float* points, int n, double voxel_size,
                     double min_weight, double* value, double* grad, uint8_t* status, void* stream);

/* Replaces one iteration of TSDFPoseOptimizer._optimize_single (tsdf_optimizer.py:77-86):
 * world = pose.act(points) (if points_in_camera_frame), residual/Jacobian/weights
 * (_build_linear_system :94-105, _sim3_jacobian :118-124), H,b (_accumulate_system :107-116, fp64),
 * and if update_pose: delta = solve(H + damping I, -b); pose <- exp(delta) * pose.
 * H_out f64[7,7], b_out f64[7], used_out i32 may be NULL.  workspace >= 64*36*8 bytes. */
int mslam_tsdf_pose_step(void* table, uint64_t capacity, const float* points, const float* conf, int n,
                         float* pose, int points_in_camera_frame, double voxel_size, double min_weight,
                         double lambda, double damping, int update_pose, double* H_out, double* b_out,
                         int* used_out, void* workspace, size_t workspace_bytes, void* stream);

/* Voxel-sharded volume (north_star: "TSDF voxel blocks shard across the 8 GPUs"; the reference's TSDFVolume is one
 * python dict, global_volume.py:15-31): a voxel lives in the table of exactly one rank, so a query / pose iteration is
 * owner-computes: every rank writes what ITS table holds for the seven voxels a query reads (centre, +x, -x, +y, -y,
 * +z, -z) of every point - out f64[n,7,3] = (state, weight, tsdf), zeros for voxels it does not own - the caller
 * all-reduces (sum) that array over the ranks, and the *_lookup entry points evaluate global_volume.py:93-128 /
 * tsdf_optimizer.py:77-124 on it: bit-identical to mslam_tsdf_query / mslam_tsdf_pose_step on one table holding
 * every voxel.  `pose` != NULL: points are camera-frame, moved with pose first (as mslam_tsdf_pose_step does). */
int mslam_tsdf_lookup7(void* table, uint64_t capacity, const float* points, int n, const float* pose,
                       double voxel_size, double* out, void* stream);
int mslam_tsdf_query_lookup(const double* lookup, int n, double voxel_size, double min_weight, double* value,
                            double* grad, uint8_t* status, void* stream);
int mslam_tsdf_pose_step_lookup(const double* lookup, const float* points, const float* conf, int n, float* pose,
                                int points_in_camera_frame, double voxel_size, double min_weight, double lambda,
                                double damping, int update_pose, double* H_out, double* b_out, int* used_out,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * MASt3R two-view forward.  Replaces the three model methods the SLAM front/back-end call
 * (mast3r_slam/mast3r_utils.py:34-40,57-64,74): model._encode_image, model._decoder,
 * model._downstream_head (thirdparty/mast3r/dust3r/dust3r/model.py:127-139,171-196;
 * mast3r/catmlp_dpt_head.py:71-96).  bf16 MFMA operands, fp32 accumulate / residual / softmax.
 *
 * create: cfg9 = {enc_dim, enc_depth, enc_heads, dec_dim, dec_depth, dec_heads, patch, desc_dim,
 * dpt_feature_dim}; weight_ptrs = DEVICE pointers in the canonical order documented in
 * mast3r_slam/mast3r_model.py::canonical_weights (matrices bf16 [out,in] with conv kernels
 * re-laid-out tap-major, biases / LayerNorm / last 1x1 conv fp32), weight_numels their element
 * counts (validated).  The model keeps the pointers; the caller keeps the tensors alive.
 * SYNCHRONISES the stream once (RoPE table upload).
 * ------------------------------------------------------------------------------------------ */
int mslam_mast3r_create(void** handle_out, const int* cfg9_host, void* const* weight_ptrs_host,
                        const long long* weight_numels_host, int n_weights, void* stream);
int mslam_mast3r_destroy(void* handle);
size_t mslam_mast3r_workspace_bytes(void* handle, int batch, int H, int W);

/* _encode_image: img f32[B,3,H,W] (ImgNorm range) -> feat f32[B, (H/16)(W/16), enc_dim]
 * (enc_norm output).  pos is implicit: token n = (y = n / (W/16), x = n % (W/16)). */
int mslam_mast3r_encode(void* handle, const float* img, int batch, int H, int W, float* feat_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* _decoder + both _downstream_head calls of mast3r_utils.decoder (mast3r_utils.py:34-40):
 * feat1, feat2 f32[B,N,enc_dim] -> for side s in {1,2}: X f32[B,H,W,3] (pts3d), C f32[B,H,W]
 * (conf), D f32[B,H,W,desc_dim] (desc), Q f32[B,H,W] (desc_conf).  dec_last1/2 (may be NULL):
 * f32[B,N,dec_dim] = dec_norm'ed last decoder tokens (for tests). */
int mslam_mast3r_decode(void* handle, const float* feat1, const float* feat2, int batch, int H, int W,
                        float* X1, float* C1, float* D1, float* Q1, float* X2, float* C2, float* D2,
                        float* Q2, float* dec_last1, float* dec_last2, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frame-to-keyframe tracking GN.  Replaces FrameTracker.opt_pose_ray_dist_sim3 /
 * opt_pose_calib_sim3 + solve (mast3r_slam/tracker.py:208-318; geometry.py:17-104;
 * nonlinear_optimizer.py:5-33): <= max_iters Gauss-Newton iterations on the relative Sim3
 * T_CkCf (f32[8] device, updated IN PLACE), residuals between keyframe points Xk f32[n,3] and
 * T.Xf[idx_f2k] (Xf f32[n,3], idx i64[n]), weights sqrt(Qk) (f32[n]) gated by valid (u8[n]).
 * use_calib: pixel + log-depth residual with K f32[3,3] (device).  No host sync: convergence
 * (rel. cost decrease < rel_error or |tau| < delta_norm) is a device flag.  status_out (device,
 * 32 bytes, may be NULL) <- {i32 done, i32 iters, i32 chol_fail, f32 old_cost, f32 last_cost,
 * f32 last_delta_norm, -, -}; chol_fail mirrors the exception path of tracker.py:72-93.
 * The loop state lives in `workspace`: iterations [first_iter, max_iters) are enqueued, first_iter == 0 resets the
 * state, so a caller may enqueue a first chunk, read the status and continue only when `done` is still 0.
 * ------------------------------------------------------------------------------------------ */
size_t mslam_track_workspace_bytes(int n_points);
int mslam_track_pose(int use_calib, float* T_rel, const float* Xf, const float* Xk, const int64_t* idx_f2k,
                     const float* Qk, const uint8_t* valid, int n_points, const float* K, int width,
                     int height, float sigma_a, float sigma_b, float huber, int pixel_border, float z_eps,
                     int first_iter, int max_iters, float rel_error, float delta_norm, void* status_out, void* workspace,
                     size_t workspace_bytes, void* stream);

/* The tensor expressions of FrameTracker.track around the pose loop (mast3r_slam/tracker.py:44-75: Qk, the validity
 * masks, match_frac; :147-177: the keyframe rule's two fractions and the keyframe's pointmap fused with the frame's view
 * of it, frame.py:41-105 'weighted_pointmap'; lietorch inv / mul / act at tracker.py:150,232) as three launches.  Each
 * value is produced by the same single IEEE operations, in the same order, as the op-by-op tensor form.
 * mslam_track_prepare: for keyframe pixel k with j = idx_f2k[k]:
 *   Qk[k] = sqrt(Qff[j] * Qkf[k]);  Cf = Cf_sum[j] * inv_nf;  Ck_avg[k] = Ck_sum[k] * inv_nk   (C / N of a Frame)
 *   valid_opt[k] = valid_match[k] & Cf > C_conf & Ck > C_conf & Qk > Q_conf;  valid_kf[k] = valid_match[k] & Qk > Q_conf
 *   T_rel = T_WCk^-1 * T_WCf (f32[8] each, unit quaternions as lietorch keeps them)
 *   workspace (16-byte aligned) <- per-block counts of valid_opt / valid_kf + one byte flag per frame pixel that some
 *   valid match points at (the verdict kernel counts them: the number of distinct idx_f2k[k] with valid_match[k]).
 * mslam_track_verdict: verdict6 <- {#valid_opt / n, iterations, chol_fail, #valid_kf / n, #distinct / n, done} with the
 *   solver status of mslam_track_pose (status_out) - the six scalars FrameTracker reads per frame.
 * mslam_track_fuse: T_WCf = T_WCk * T_rel;  X_new = ((C X_canon) + (Ckf (T_rel . Xkf))) / (C + Ckf);  C_new = C + Ckf. */
size_t mslam_track_prepare_workspace_bytes(int n_points);
int mslam_track_prepare(const int64_t* idx_f2k, const uint8_t* valid_match, const float* Qff, const float* Qkf,
                        const float* Cf_sum, float inv_nf, const float* Ck_sum, float inv_nk, float C_conf, float Q_conf,
                        int n_points, const float* T_WCk, const float* T_WCf, float* Qk, float* Ck_avg,
                        uint8_t* valid_opt, uint8_t* valid_kf, float* T_rel, void* workspace, size_t workspace_bytes,
                        void* stream);
int mslam_track_verdict(const void* prepare_workspace, const void* status, int n_points, float* verdict6, void* stream);
int mslam_track_fuse(const float* T_WCk, const float* T_rel, const float* Xkf, const float* Ckf, const float* X_canon,
                     const float* C, int n_points, float* T_WCf, float* X_new, float* C_new, void* stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks of the MASt3R forward, exported for kernel-level parity tests and roofline
 * measurement (they have no counterpart in the reference's API: it calls cuBLAS/cuDNN through
 * torch, blocks.py:88-109, dpt_block.py:33-68).
 * ------------------------------------------------------------------------------------------ */
/* out[M,N] = act(A[M,K] . W[N,K]^T + bias) (+ residual f32[M,N]); A, W bf16; act 0 none, 1 GELU(erf),
 * 2 ReLU; out f32 or bf16.  K % 8 == 0. */
int mslam_gemm_bf16(const void* A, const void* W, const float* bias, const void* residual_f32, void* out,
                    int M, int N, int K, int act, int out_is_bf16, void* stream);
/* Tuning hook: force the tile configuration (codes in csrc/gemm.hip: 642 ... 2256; 0 = back to the built-in choice)
 * for every plain GEMM of exactly this shape (M < 0: the implicit-conv GEMM of shape |M| x N x K), process-wide.  tools/insitu_tune.py uses it to time whole network
 * stages under alternative tilings; results do not depend on the tiling (same K order per output element). */
int mslam_gemm_tile_override(int M, int N, int K, int cfg);
/* Measurement hook (bench.py `roofline`): between begin and end every launch of the plain GEMM of exactly this
 * shape - from any entry point, on any stream - is bracketed by two HIP events recorded on the stream it is
 * launched on (at most max_samples launches).  end() waits for the recorded events and returns the average and
 * minimum launch duration in microseconds (host doubles / int). */
int mslam_gemm_profile_begin(int M, int N, int K, int max_samples);
int mslam_gemm_profile_end(double* avg_us, double* min_us, int* samples);
/* NHWC bf16 conv (ks 1|3, stride 1|2, pad ks/2), W bf16 [Cout, ks*ks*Cin] tap-major; optional ReLU on
 * the input, act on the output, bf16 residual added after act. */
int mslam_conv2d_nhwc_bf16(const void* in, const void* W, const float* bias, const void* residual_bf16,
                           void* out_bf16, int B, int H, int Wd, int Cin, int Cout, int ks, int stride,
                           int relu_in, int act, void* stream);
/* O[B,Nq,H*64] = softmax(Q K^T) V ; Q,K bf16 [B,H,N,64] (Q pre-scaled by 1/8), VT bf16 [B,H,64,Nk]. */
int mslam_attention_bf16(const void* Q, const void* K, const void* VT, void* O, int batch, int heads,
                         int nq, int nk, void* stream);
/* torch.nn.LayerNorm over the last dim (D <= 2048): x f32[rows,D] -> bf16 and/or f32 outputs. */
int mslam_layernorm_f32(const float* x, const float* w, const float* b, void* out_bf16, float* out_f32,
                        int rows, int D, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * Local dense-block TSDF (camera-side half of the "dual TSDF"): replaces the python loops of
 * TSDFRefiner._build_tsdf_robust (mast3r_slam/tsdf_refine.py:837-940) and
 * _extract_surface_safe + _sample_tsdf_trilinear (:942-1064).  Grid layout [nz,ny,nx] f32, dims
 * (<= 64 per axis) computed by the caller as clamp(ceil(roi/voxel_size), max=max_grid_dim) (:844-846).
 * build: X_world f32[n,3] = T_WC.act(X_canon), C f32[n], origin/xyz_min/xyz_max f32[3] (device);
 * sequential float32 running-average semantics are preserved by per-voxel ordered replay.
 * raycast: sel_pix i64[n_sel] = pixel indices to march (the reference draws <= 100 with randperm);
 * surf f32[n_sel,3] <- surface point (or the original point), hit u8[n_sel].
 * ------------------------------------------------------------------------------------------ */
size_t mslam_tsdf_local_workspace_bytes(int n_points);
int mslam_tsdf_local_build(const float* X_world, const float* C, const float* origin, const float* xyz_min,
                           const float* xyz_max, int n_points, int nx, int ny, int nz, double voxel_size,
                           double trunc, float min_confidence, float* tsdf, float* weights, void* workspace,
                           size_t workspace_bytes, void* stream);
int mslam_tsdf_local_raycast(const float* tsdf, int nx, int ny, int nz, const float* xyz_min,
                             const float* xyz_max, const float* X_original, const int64_t* sel_pix, int n_sel,
                             int n_samples, float max_displacement, float* surf, uint8_t* hit, void* stream);

/* ------------------------------------------------------------------------------------------
 * Quality service patch statistics: replaces the torch reshape + nanmedian pipeline of
 * mast3r_slam/quality_core.py.  reduce_grid (:15-29) / u_from_CQ (:45-52) / r_from_scalar (:54-55) /
 * valid_grid (:57-59): one value per ps x ps patch of an h x w map, out f32[(h/ps)*(w/ps)].
 *   mode 0: nanmedian of x over valid pixels (valid u8[h*w] or NULL), empty patch -> 0
 *   mode 1: mean (valid NULL) / nanmean over valid pixels
 *   mode 2: median of U = 1 - sqrt(clamp(C/(c_thr+1e-8)) * clamp(Q/(q_thr+1e-8))), x = C, y = Q
 * classify (:66-117): robust z-scores of r and u over the n-patch grid, class ids int64[n], normalised priority f32[n].
 * ------------------------------------------------------------------------------------------ */
int mslam_quality_reduce_grid(const float* x, const float* y, const uint8_t* valid, int h, int w, int ps, int mode,
                              double c_thr, double q_thr, float* out, void* stream);
int mslam_quality_classify(const float* delta_cov, const float* r, const float* u, int n, float thr_zr, float thr_zu,
                           float thr_dc, int64_t* cls, float* pri, void* stream);

/* ------------------------------------------------------------------------------------------
 * Synthetic-data source (NOT a reference interface): the procedural box room of mast3r_slam/synthetic.py rendered
 * for `batch` view pairs in one kernel, in the layout of the MASt3R heads' outputs (X f32[b,h,w,3], C f32[b,h,w],
 * D f32[b,h,w,24], Q f32[b,h,w]; side 1 = view i in frame i, side 2 = view j in frame i).  ki / kj: device f32[batch]
 * camera-path indices; Wm (3x24) and phase (24): HOST doubles of the descriptor field.  Stands in for a trained
 * network's output in runs without a checkpoint (mast3r_slam/synthetic_gpu.py, bench.py).
 * ------------------------------------------------------------------------------------------ */
int mslam_room_pair(const float* ki, const float* kj, int batch, int h, int w, int n_frames, double fx, double fy,
                    double cx, double cy, double noise, const double* Wm_3x24, const double* phase_24, float* X1,
                    float* C1, float* D1, float* Q1, float* X2, float* C2, float* D2, float* Q2, void* stream);

/* cv2.remap(img, mapx, mapy, cv2.INTER_LINEAR) of the calibrated dataset readers (mast3r_slam/dataloader.py:495-496,
 * Intrinsics.remap) for 8-bit images: src u8[src_h, src_w, channels], maps f32[dst_h, dst_w] (position in src of every
 * dst pixel), dst u8[dst_h, dst_w, channels]; OpenCV's published fixed-point bilinear (1/32-pixel positions, 2^15
 * weights, constant border 0).  Parity with the library unpinned (OpenCV is absent): see csrc/undistort.hip. */
int mslam_remap_bilinear_u8(const uint8_t* src, int src_h, int src_w, int channels, const float* mapx,
                            const float* mapy, uint8_t* dst, int dst_h, int dst_w, void* stream);

/* fp64 GEMM of the retrieval head (Whitener.forward, thirdparty/mast3r/mast3r/retrieval/model.py:62-77; the projector's
 * Linear layers, model.py:108-151) on the f64 matrix cores:  out f64[M,N] = (A[M,K] - centre[K]) . B + bias[N].
 * A is f32 or f64 row-major [M,K]; B is f32 or f64, [K,N] row-major (b_transposed = 0) or [N,K] row-major
 * (b_transposed = 1: an nn.Linear weight); centre f64[K] and bias f64[N] may be NULL. */
int mslam_gemm_f64(const void* A, int a_is_f32, const void* B, int b_is_f32, int b_transposed, const double* centre,
                   const double* bias, double* out, int M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * Retrieval database: ASMK with binarised residuals.  Replaces, for one image at a time,
 *   ASMKKernel.aggregate_image + hamming.binarize_and_pack_2D   (thirdparty/mast3r/asmk/asmk/kernel.py:28-42,
 *                                                                asmk/cython/hamming.pyx:93-127)
 *   IVF.search + ASMKKernel.similarity + functional.asmk_kernel (asmk/inverted_file.py:90-114, kernel.py:59-71,
 *                                                                functional.py:10-15; use_idf False, processor.py:85)
 * as mast3r_slam/retrieval_database.py:107-166 calls them.
 *
 * mslam_asmk_aggregate: des f32[n_des, dim]; centroids f32[n_centroids, dim]; assign i64[n_des, m_assign] (the output
 *   of quantize_custom); uniq_words i64[n_uniq] = sorted unique values of `assign`; sig_out u32[n_uniq, dim/32]:
 *   bit (31 - d%32) of word d/32 = (sum of residuals of dimension d > 0).  dim must be a multiple of 32.
 * mslam_asmk_search: the inverted file as flat arrays in insertion order - entry_word i32[n_entries], entry_sig
 *   u32[n_entries, sig_words], img_start i32[n_images + 1] (entries of image i are [img_start[i], img_start[i+1]), words
 *   ascending) - queried with q_words i32[n_q] (sorted, unique) / q_sig u32[n_q, sig_words];
 *   scores f64[n_images] = sum over shared words of sim^alpha [sim >= threshold] / sqrt(entries of the image), over
 *   sqrt(n_q); sim = 1 - 2 hamming / (32 sig_words).  Signatures 16-byte aligned when sig_words % 4 == 0.
 * ------------------------------------------------------------------------------------------ */
int mslam_asmk_aggregate(const float* des, const float* centroids, const int64_t* assign, const int64_t* uniq_words,
                         uint32_t* sig_out, int n_des, int m_assign, int dim, int n_uniq, int n_centroids, void* stream);
int mslam_asmk_search(const int32_t* entry_word, const uint32_t* entry_sig, const int32_t* img_start, int n_images,
                      const int32_t* q_words, const uint32_t* q_sig, int n_q, int sig_words, float similarity_threshold,
                      float alpha, double* scores, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSLAM_HIP_H */
