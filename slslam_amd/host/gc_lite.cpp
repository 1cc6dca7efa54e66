// slslam_amd/host/gc_lite.cpp — see gc_lite.h.
#include "gc_lite.h"

#include <cmath>

namespace {
inline void mat_vec(const double R[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
inline void cross(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
inline double norm3(const double a[3]) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
}  // namespace

extern "C" {

// ceres::AngleAxisToRotationMatrix semantics (reference src/gc.cpp:24-36)
void slslam_gc_rodrigues_to_R(const double w[3], double R[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > 0.0) {
    const double th = std::sqrt(th2), wx = w[0] / th, wy = w[1] / th, wz = w[2] / th;
    const double c = std::cos(th), s = std::sin(th), o = 1.0 - c;
    R[0] = c + wx * wx * o;       R[1] = wx * wy * o - wz * s;  R[2] = wx * wz * o + wy * s;
    R[3] = wy * wx * o + wz * s;  R[4] = c + wy * wy * o;       R[5] = wy * wz * o - wx * s;
    R[6] = wz * wx * o - wy * s;  R[7] = wz * wy * o + wx * s;  R[8] = c + wz * wz * o;
  } else {
    R[0] = 1; R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2]; R[4] = 1; R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
  }
}

// ceres::RotationMatrixToAngleAxis via the quaternion (reference src/gc.cpp:38-49)
void slslam_gc_R_to_rodrigues(const double R[9], double w[3]) {
  double q[4];
  const double tr = R[0] + R[4] + R[8];
  if (tr >= 0.0) {
    double t = std::sqrt(tr + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (R[7] - R[5]) * t; q[2] = (R[2] - R[6]) * t; q[3] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[4 * i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
    q[i + 1] = 0.5 * t; t = 0.5 / t;
    q[0] = (R[3 * k + j] - R[3 * j + k]) * t;
    q[j + 1] = (R[3 * j + i] + R[3 * i + j]) * t;
    q[k + 1] = (R[3 * k + i] + R[3 * i + k]) * t;
  }
  const double s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    const double two_theta = 2.0 * ((q[0] < 0.0) ? std::atan2(-s, -q[0]) : std::atan2(s, q[0]));
    const double kk = two_theta / s;
    w[0] = q[1] * kk; w[1] = q[2] * kk; w[2] = q[3] * kk;
  } else {
    w[0] = 2.0 * q[1]; w[1] = 2.0 * q[2]; w[2] = 2.0 * q[3];
  }
}

void slslam_gc_wt_to_Rt(const double wt[6], slslam_pose* T) {
  slslam_gc_rodrigues_to_R(wt, T->R);
  T->t[0] = wt[3]; T->t[1] = wt[4]; T->t[2] = wt[5];
}

void slslam_gc_Rt_to_wt(const slslam_pose* T, double wt[6]) {
  slslam_gc_R_to_rodrigues(T->R, wt);
  wt[3] = T->t[0]; wt[4] = T->t[1]; wt[5] = T->t[2];
}

void slslam_gc_T_inv(const slslam_pose* T, slslam_pose* Ti) {     // (R^T, -R^T t)
  slslam_pose o;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.R[3 * i + j] = T->R[3 * j + i];
  double v[3];
  mat_vec(o.R, T->t, v);
  o.t[0] = -v[0]; o.t[1] = -v[1]; o.t[2] = -v[2];
  *Ti = o;
}

void slslam_gc_T_20(const slslam_pose* T21, const slslam_pose* T10, slslam_pose* T20) {   // (R21 R10, R21 t10 + t21)
  slslam_pose o;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    double s = 0; for (int k = 0; k < 3; ++k) s += T21->R[3 * i + k] * T10->R[3 * k + j];
    o.R[3 * i + j] = s;
  }
  mat_vec(T21->R, T10->t, o.t);
  for (int i = 0; i < 3; ++i) o.t[i] += T21->t[i];
  *T20 = o;
}

void slslam_gc_T_21(const slslam_pose* T20, const slslam_pose* T10, slslam_pose* T21) {
  slslam_pose inv;
  slslam_gc_T_inv(T10, &inv);
  slslam_gc_T_20(T20, &inv, T21);
}

void slslam_gc_line_to_pose(const double line_w[6], const slslam_pose* T, double line_c[6]) {
  double cp[3], dv[3];
  mat_vec(T->R, line_w, cp);
  mat_vec(T->R, line_w + 3, dv);
  for (int i = 0; i < 3; ++i) { line_c[i] = cp[i] + T->t[i]; line_c[3 + i] = dv[i]; }
}

void slslam_gc_line_from_pose(const double line_c[6], const slslam_pose* T, double line_w[6]) {
  slslam_pose inv;
  slslam_gc_T_inv(T, &inv);
  slslam_gc_line_to_pose(line_c, &inv, line_w);
}

void slslam_gc_av_to_orth(const double av[6], double orth[4]) {
  double n[3];
  cross(av, av + 3, n);
  const double nn = norm3(n), vn = norm3(av + 3);
  const double x[3] = { n[0] / nn, n[1] / nn, n[2] / nn }, y[3] = { av[3] / vn, av[4] / vn, av[5] / vn };
  double z[3];
  cross(x, y, z);
  orth[0] = std::atan2(y[2], z[2]);
  orth[1] = std::asin(-x[2]);
  orth[2] = std::atan2(x[1], x[0]);
  orth[3] = std::asin(vn / std::sqrt(nn * nn + vn * vn));
}

void slslam_gc_orth_to_av(const double orth[4], double av[6]) {
  const double s1 = std::sin(orth[0]), c1 = std::cos(orth[0]), s2 = std::sin(orth[1]), c2 = std::cos(orth[1]);
  const double s3 = std::sin(orth[2]), c3 = std::cos(orth[2]), d = std::cos(orth[3]) / std::sin(orth[3]);
  av[0] = -(c1 * s2 * c3 + s1 * s3) * d;
  av[1] = -(c1 * s2 * s3 - s1 * c3) * d;
  av[2] = -(c1 * c2) * d;
  av[3] = s1 * s2 * c3 - c1 * s3;
  av[4] = s1 * s2 * s3 + c1 * c3;
  av[5] = s1 * c2;
}

}  // extern "C"
