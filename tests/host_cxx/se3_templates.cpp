// tests/host_cxx/se3_templates.cpp — the SE(3) templates of the mirrored header (slslam_amd/host/po_problem.h: gc_T_inv, gc_w_20,
// gc_T_20, reference src/po_problem.h:27-64) instantiated for double and compared with the matrix forms of gc_lite.h
// (slslam_gc_T_inv / slslam_gc_T_20): prints the largest difference over random poses, including rotations near 0 and near pi.
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include "po_problem.h"
#include "gc_lite.h"

static double urand() { return (double)std::rand() / RAND_MAX * 2.0 - 1.0; }

int main() {
  std::srand(7);
  const double scales[] = { 0.0, 1e-9, 1e-3, 0.7, 2.0, 3.1, 3.14159 };
  double worst_inv = 0.0, worst_comp = 0.0, worst_quat = 0.0;
  for (int it = 0; it < 4000; ++it) {
    double a[6], b[6];
    for (int i = 0; i < 6; ++i) { a[i] = urand(); b[i] = urand(); }
    const double sa = scales[it % 7], sb = scales[(it / 7) % 7];
    const double na = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]), nb = std::sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    for (int i = 0; i < 3; ++i) { a[i] *= sa / na; b[i] *= sb / nb; }
    // inverse
    double ai[6], back[6];
    gc_T_inv(a, ai);
    slslam_pose Ta, Tai, Tref;
    slslam_gc_wt_to_Rt(a, &Ta); slslam_gc_wt_to_Rt(ai, &Tai); slslam_gc_T_inv(&Ta, &Tref);
    for (int i = 0; i < 9; ++i) worst_inv = std::fmax(worst_inv, std::fabs(Tai.R[i] - Tref.R[i]));
    for (int i = 0; i < 3; ++i) worst_inv = std::fmax(worst_inv, std::fabs(Tai.t[i] - Tref.t[i]));
    gc_T_inv(ai, back);
    for (int i = 0; i < 6; ++i) worst_inv = std::fmax(worst_inv, std::fabs(back[i] - a[i]) * (sa > 3.0 ? 0.0 : 1.0));   // (near pi the angle-axis of a rotation is not unique to round-off)
    // composition: matrices of T20 = T21 T10
    double c[6];
    gc_T_20(a, b, c);
    slslam_pose Tb, Tc, Tcref;
    slslam_gc_wt_to_Rt(b, &Tb); slslam_gc_wt_to_Rt(c, &Tc); slslam_gc_T_20(&Ta, &Tb, &Tcref);
    for (int i = 0; i < 9; ++i) worst_comp = std::fmax(worst_comp, std::fabs(Tc.R[i] - Tcref.R[i]));
    for (int i = 0; i < 3; ++i) worst_comp = std::fmax(worst_comp, std::fabs(Tc.t[i] - Tcref.t[i]));
    // quaternion round trip
    double q[4], w2[3];
    ceres::AngleAxisToQuaternion(a, q);
    worst_quat = std::fmax(worst_quat, std::fabs(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] - 1.0));
    ceres::QuaternionToAngleAxis(q, w2);
    for (int i = 0; i < 3; ++i) worst_quat = std::fmax(worst_quat, std::fabs(w2[i] - a[i]));
  }
  std::printf("%.3e %.3e %.3e\n", worst_inv, worst_comp, worst_quat);
  return 0;
}
