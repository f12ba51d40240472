// Synthetic-data source for runs without trained weights (BASELINE config 3): the procedural box room of
// mast3r_slam/synthetic.py rendered for a batch of view pairs in ONE kernel - pointmaps of both views in the first
// view's frame, confidences, 24-d descriptors - in the MASt3R heads' output layout.  Not part of the reference's
// interface: it stands in for what a trained network would output (mast3r_slam/synthetic_gpu.py::RoomGeometryModel),
// so that the rest of the loop works on meaningful geometry.  Same formulas as RoomRenderer.pair (torch), fused.
#include "common.h"

namespace mslam {

constexpr int kRoomDesc = 24;

struct RoomParams {
  int h, w, n_frames;
  double fx, fy, cx, cy;
  double noise;
  double Wm[3][kRoomDesc];
  double ph[kRoomDesc];
};

struct Pose { double t[3], q[4]; };   // scale 1

__device__ __forceinline__ Pose room_pose(double k, int n_frames) {
  const double a = 2.0 * 3.14159265358979323846 * k / (double)(n_frames > 1 ? n_frames : 1) * 3.0;
  Pose P;
  P.t[0] = 1.2 * cos(a); P.t[1] = 0.6 * sin(0.7 * a); P.t[2] = 0.4 * sin(a);
  const double r0 = 0.15 * sin(0.9 * a), r1 = a * 0.35, r2 = 0.1 * cos(1.3 * a);
  const double th = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
  if (th < 1e-12) { P.q[0] = P.q[1] = P.q[2] = 0.0; P.q[3] = 1.0; }
  else {
    const double s = sin(0.5 * th) / th;
    P.q[0] = r0 * s; P.q[1] = r1 * s; P.q[2] = r2 * s; P.q[3] = cos(0.5 * th);
  }
  return P;
}

__device__ __forceinline__ void qrot(const double* q, const double* X, double* Y) {
  const double u0 = 2.0 * (q[1] * X[2] - q[2] * X[1]);
  const double u1 = 2.0 * (q[2] * X[0] - q[0] * X[2]);
  const double u2 = 2.0 * (q[0] * X[1] - q[1] * X[0]);
  Y[0] = X[0] + q[3] * u0 + (q[1] * u2 - q[2] * u1);
  Y[1] = X[1] + q[3] * u1 + (q[2] * u0 - q[0] * u2);
  Y[2] = X[2] + q[3] * u2 + (q[0] * u1 - q[1] * u0);
}

// camera-frame point of pixel (u, v) of the view at pose P: ray (x, y, 1) scaled to the inside of the 6 x 4 x 3 m box
__device__ __forceinline__ void room_point(const Pose& P, double x, double y, double* Xc, double* Xw) {
  const double d[3] = {x, y, 1.0};
  double dw[3];
  qrot(P.q, d, dw);
  const double half[3] = {3.0, 2.0, 1.5};
  double t = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    double ta;
    if (fabs(dw[a]) < 1e-12) ta = INFINITY;
    else ta = ((dw[a] > 0 ? half[a] : -half[a]) - P.t[a]) / dw[a];
    t = fmin(t, ta);
  }
  Xc[0] = x * t; Xc[1] = y * t; Xc[2] = t;
  Xw[0] = dw[0] * t + P.t[0]; Xw[1] = dw[1] * t + P.t[1]; Xw[2] = dw[2] * t + P.t[2];
}

__device__ __forceinline__ double hash01(double pix, double ch, double ki, double kj, double salt) {
  const double x = pix * 0.618033988749895 + ch * 0.754877666246693 + (ki * 12.9898 + kj * 78.233 + salt * 37.719);
  const double v = sin(x * 12.9898) * 43758.5453;
  return v - floor(v);
}

__global__ __launch_bounds__(256) void room_pair_kernel(const float* __restrict__ ki_, const float* __restrict__ kj_, int B,
                                                        RoomParams R, float* __restrict__ X1, float* __restrict__ C1,
                                                        float* __restrict__ D1, float* __restrict__ Q1,
                                                        float* __restrict__ X2, float* __restrict__ C2,
                                                        float* __restrict__ D2, float* __restrict__ Q2) {
  const int hw = R.h * R.w;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)B * hw) return;
  const int b = (int)(gid / hw), pix = (int)(gid - (long long)b * hw);
  const double ki = (double)ki_[b], kj = (double)kj_[b];
  const Pose Pi = room_pose(ki, R.n_frames), Pj = room_pose(kj, R.n_frames);
  const double x = ((double)(pix % R.w) - R.cx) / R.fx, y = ((double)(pix / R.w) - R.cy) / R.fy;
  double Xi_i[3], Pw_i[3], Xj_j[3], Pw_j[3];
  room_point(Pi, x, y, Xi_i, Pw_i);
  room_point(Pj, x, y, Xj_j, Pw_j);
  // view j's point in frame i: R_i^T (Pw_j - t_i)
  const double qi_inv[4] = {-Pi.q[0], -Pi.q[1], -Pi.q[2], Pi.q[3]};
  const double dlt[3] = {Pw_j[0] - Pi.t[0], Pw_j[1] - Pi.t[1], Pw_j[2] - Pi.t[2]};
  double Xj_i[3];
  qrot(qi_inv, dlt, Xj_i);
  const double* Xs[2] = {Xi_i, Xj_i};
  const double* Pws[2] = {Pw_i, Pw_j};
  float* Xo[2] = {X1, X2};
  float* Co[2] = {C1, C2};
  float* Do[2] = {D1, D2};
  float* Qo[2] = {Q1, Q2};
#pragma unroll
  for (int side = 0; side < 2; side++) {
    const size_t o = (size_t)b * hw + pix;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double u1 = fmax(hash01(pix, c, ki, kj, 4 * side + 0), 1e-12), u2 = hash01(pix, c, ki, kj, 4 * side + 1);
      const double g = sqrt(-2.0 * log(u1)) * cos(2.0 * 3.14159265358979323846 * u2);
      Xo[side][o * 3 + c] = (float)(Xs[side][c] + R.noise * g);
    }
    Co[side][o] = (float)(1.0 + 2.0 * hash01(pix, 0, ki, kj, 4 * side + 2));
    Qo[side][o] = (float)(1.5 + 2.5 * hash01(pix, 0, ki, kj, 4 * side + 3));
    double d[kRoomDesc], n2 = 0.0;
#pragma unroll
    for (int f = 0; f < kRoomDesc; f++) {
      d[f] = sin(Pws[side][0] * R.Wm[0][f] + Pws[side][1] * R.Wm[1][f] + Pws[side][2] * R.Wm[2][f] + R.ph[f]);
      n2 += d[f] * d[f];
    }
    const double inv = 1.0 / sqrt(n2);
#pragma unroll
    for (int f = 0; f < kRoomDesc; f++) Do[side][o * kRoomDesc + f] = (float)(d[f] * inv);
  }
}

}  // namespace mslam

using namespace mslam;

extern "C" int mslam_room_pair(const float* ki, const float* kj, int batch, int h, int w, int n_frames, double fx,
                               double fy, double cx, double cy, double noise, const double* Wm_3x24,
                               const double* phase_24, float* X1, float* C1, float* D1, float* Q1, float* X2,
                               float* C2, float* D2, float* Q2, void* stream) {
  MSLAM_REQUIRE(batch > 0 && h > 0 && w > 0, "room_pair: bad sizes");
  MSLAM_REQUIRE(ki && kj && Wm_3x24 && phase_24 && X1 && C1 && D1 && Q1 && X2 && C2 && D2 && Q2, "room_pair: null pointer");
  RoomParams R;
  R.h = h; R.w = w; R.n_frames = n_frames; R.fx = fx; R.fy = fy; R.cx = cx; R.cy = cy; R.noise = noise;
  for (int a = 0; a < 3; a++)
    for (int f = 0; f < kRoomDesc; f++) R.Wm[a][f] = Wm_3x24[a * kRoomDesc + f];   // HOST arrays
  for (int f = 0; f < kRoomDesc; f++) R.ph[f] = phase_24[f];
  const long long total = (long long)batch * h * w;
  hipLaunchKernelGGL(room_pair_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ki, kj,
                     batch, R, X1, C1, D1, Q1, X2, C2, D2, Q2);
  MSLAM_LAUNCH_CHECK("room_pair");
  return MSLAM_OK;
}
