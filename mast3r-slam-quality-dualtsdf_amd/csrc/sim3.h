// Sim3 algebra shared by the GN, tracker and TSDF kernels (device + host inline).
// Pose storage is lietorch's: [tx,ty,tz, qx,qy,qz,qw, s]  (frame.py:24, lietorch Sim3.data).
// Formulae follow the reference's in-tree restatement of lietorch
// (mast3r_slam/backend/src/gn_kernels.cu:177-413); written from the maths, fmaf where natural.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace mslam {

#define MSLAM_HD __host__ __device__ __forceinline__

struct Sim3f {
  float t[3];
  float q[4];  // x,y,z,w
  float s;
};

MSLAM_HD Sim3f sim3_load(const float* p) {
  Sim3f T;
  T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
  T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
  T.s = p[7];
  return T;
}

MSLAM_HD void sim3_store(float* p, const Sim3f& T) {
  p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2];
  p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
  p[7] = T.s;
}

// qi * qj  (gn_kernels.cu:178-184)
MSLAM_HD void quat_mul(const float* a, const float* b, float* o) {
  const float o0 = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const float o1 = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  const float o2 = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  const float o3 = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}

// rotate X by unit quaternion q (gn_kernels.cu:195-205): X + w*uv + q x uv, uv = 2 q x X
MSLAM_HD void quat_rot(const float* q, const float* X, float* Y) {
  const float u0 = 2.0f * (q[1] * X[2] - q[2] * X[1]);
  const float u1 = 2.0f * (q[2] * X[0] - q[0] * X[2]);
  const float u2 = 2.0f * (q[0] * X[1] - q[1] * X[0]);
  const float y0 = X[0] + q[3] * u0 + (q[1] * u2 - q[2] * u1);
  const float y1 = X[1] + q[3] * u1 + (q[2] * u0 - q[0] * u2);
  const float y2 = X[2] + q[3] * u2 + (q[0] * u1 - q[1] * u0);
  Y[0] = y0; Y[1] = y1; Y[2] = y2;
}

// s*R*X + t
MSLAM_HD void sim3_act(const Sim3f& T, const float* X, float* Y) {
  quat_rot(T.q, X, Y);
  Y[0] = Y[0] * T.s + T.t[0];
  Y[1] = Y[1] * T.s + T.t[1];
  Y[2] = Y[2] * T.s + T.t[2];
}

// lietorch keeps the rotation a UNIT quaternion: every SO3 / RxSO3 object it constructs - the result of inv, *, exp,
// retr - normalises its quaternion (lietorch/include/so3.h, rxso3.h constructors).  Without it the frontend's cycle
// T_CkCf = T_WCk^-1 * T_WCf ; T_WCf = T_WCk * T_CkCf squares the keyframe's norm error into every frame and the poses
// stop being similarities after a few keyframe generations.  The reference's in-kernel restatement
// (gn_kernels.cu:177-413, used by its GN kernels and here by theirs) does NOT normalise; the lietorch-surface ops
// (mslam_sim3_op) and the loops that stand for Python-level lietorch calls (tracking retraction, TSDF pose update) do.
MSLAM_HD Sim3f sim3_unit(Sim3f T) {
  const float n2 = T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3];
  const float inv = 1.0f / sqrtf(n2);
  T.q[0] *= inv; T.q[1] *= inv; T.q[2] *= inv; T.q[3] *= inv;
  return T;
}

MSLAM_HD Sim3f sim3_inv(const Sim3f& T) {
  Sim3f I;
  I.q[0] = -T.q[0]; I.q[1] = -T.q[1]; I.q[2] = -T.q[2]; I.q[3] = T.q[3];
  I.s = 1.0f / T.s;
  float r[3];
  quat_rot(I.q, T.t, r);
  I.t[0] = -I.s * r[0]; I.t[1] = -I.s * r[1]; I.t[2] = -I.s * r[2];
  return I;
}

// A * B  (composition: X -> A(B(X)))
MSLAM_HD Sim3f sim3_mul(const Sim3f& A, const Sim3f& B) {
  Sim3f C;
  quat_mul(A.q, B.q, C.q);
  float r[3];
  quat_rot(A.q, B.t, r);
  C.t[0] = A.s * r[0] + A.t[0];
  C.t[1] = A.s * r[1] + A.t[1];
  C.t[2] = A.s * r[2] + A.t[2];
  C.s = A.s * B.s;
  return C;
}

// Ti^-1 * Tj  as the reference computes it (relSim3, gn_kernels.cu:252-272)
MSLAM_HD Sim3f sim3_rel(const Sim3f& Ti, const Sim3f& Tj) {
  Sim3f R;
  const float si_inv = 1.0f / Ti.s;
  R.s = si_inv * Tj.s;
  const float qi_inv[4] = {-Ti.q[0], -Ti.q[1], -Ti.q[2], Ti.q[3]};
  quat_mul(qi_inv, Tj.q, R.q);
  float d[3] = {Tj.t[0] - Ti.t[0], Tj.t[1] - Ti.t[1], Tj.t[2] - Ti.t[2]};
  quat_rot(qi_inv, d, d);
  R.t[0] = d[0] * si_inv; R.t[1] = d[1] * si_inv; R.t[2] = d[2] * si_inv;
  return R;
}

// row-major 3x3 rotation matrix of a unit quaternion
MSLAM_HD void quat_to_mat(const float* q, float* R) {
  const float x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1.0f - 2.0f * (y * y + z * z); R[1] = 2.0f * (x * y - z * w);        R[2] = 2.0f * (x * z + y * w);
  R[3] = 2.0f * (x * y + z * w);        R[4] = 1.0f - 2.0f * (x * x + z * z); R[5] = 2.0f * (y * z - x * w);
  R[6] = 2.0f * (x * z - y * w);        R[7] = 2.0f * (y * z + x * w);        R[8] = 1.0f - 2.0f * (x * x + y * y);
}

// SO3 exponential as a quaternion (gn_kernels.cu:299-321)
MSLAM_HD void so3_exp(const float* phi, float* q) {
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  float imag, real;
  if (theta_sq < 1e-6f) {
    const float theta_p4 = theta_sq * theta_sq;
    imag = 0.5f - (1.0f / 48.0f) * theta_sq + (1.0f / 3840.0f) * theta_p4;
    real = 1.0f - (1.0f / 8.0f) * theta_sq + (1.0f / 384.0f) * theta_p4;
  } else {
    const float theta = sqrtf(theta_sq);
    imag = sinf(0.5f * theta) / theta;
    real = cosf(0.5f * theta);
  }
  q[0] = imag * phi[0]; q[1] = imag * phi[1]; q[2] = imag * phi[2]; q[3] = real;
}

MSLAM_HD void cross3(const float* a, const float* b, float* o) {
  const float x = a[1] * b[2] - a[2] * b[1];
  const float y = a[2] * b[0] - a[0] * b[2];
  const float z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}

// Sim3 exponential of xi = [tau(3), phi(3), sigma]  (gn_kernels.cu:323-390; lietorch rxso3.h W matrix)
MSLAM_HD Sim3f sim3_exp(const float* xi) {
  Sim3f T;
  float tau[3] = {xi[0], xi[1], xi[2]};
  const float phi[3] = {xi[3], xi[4], xi[5]};
  const float sigma = xi[6];
  const float scale = expf(sigma);
  so3_exp(phi, T.q);
  T.s = scale;
  const float theta_sq = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float theta = sqrtf(theta_sq);
  float A, B, C;
  const float eps = 1e-6f;
  if (fabsf(sigma) < eps) {
    C = 1.0f;
    if (fabsf(theta) < eps) {
      A = 0.5f;
      B = 1.0f / 6.0f;
    } else {
      A = (1.0f - cosf(theta)) / theta_sq;
      B = (theta - sinf(theta)) / (theta_sq * theta);
    }
  } else {
    C = (scale - 1.0f) / sigma;
    if (fabsf(theta) < eps) {
      const float sigma_sq = sigma * sigma;
      A = ((sigma - 1.0f) * scale + 1.0f) / sigma_sq;
      B = (scale * 0.5f * sigma_sq + scale - 1.0f - sigma * scale) / (sigma_sq * sigma);
    } else {
      const float a = scale * sinf(theta);
      const float b = scale * cosf(theta);
      const float c = theta_sq + sigma * sigma;
      A = (a * sigma + (1.0f - b) * theta) / (theta * c);
      B = (C - ((b - 1.0f) * sigma + a * theta) / c) / theta_sq;
    }
  }
  T.t[0] = C * tau[0]; T.t[1] = C * tau[1]; T.t[2] = C * tau[2];
  cross3(phi, tau, tau);
  T.t[0] += A * tau[0]; T.t[1] += A * tau[1]; T.t[2] += A * tau[2];
  cross3(phi, tau, tau);
  T.t[0] += B * tau[0]; T.t[1] += B * tau[1]; T.t[2] += B * tau[2];
  return T;
}

// left retraction exp(xi) * T  (retrSim3, gn_kernels.cu:392-413)
MSLAM_HD Sim3f sim3_retr(const float* xi, const Sim3f& T) { return sim3_mul(sim3_exp(xi), T); }

}  // namespace mslam
