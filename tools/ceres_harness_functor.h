// tools/ceres_harness_functor.h — the residual functor of tools/ceres_harness.cpp, in a header of its own so that the CPU test-suite
// can instantiate it for double (tests/test_host_cxx.py::test_ceres_harness_functor_known_answer: the survey's known-answer vector) on the
// rotation helpers of slslam_amd/host/ceres/rotation.h - the rest of the harness needs a Ceres installation.  Needs "ceres/rotation.h"
// (ceres::AngleAxisRotatePoint) included before it.
//
// Restated from the mathematics of the reference functor (src/lba_problem.h:46-118): a line in orthonormal parameters (a, b, g, t) ->
// closest point cp = -cot(t) col2(R_l), direction col1(R_l), R_l = Rz(g) Ry(b) Rx(a); rotated and translated into the keyframe; for both cameras
// of the stereo pair (the second 0.12 m along +x) the signed distances of the two observed end points to the projected line.
#ifndef SLSLAM_CERES_HARNESS_FUNCTOR_H_
#define SLSLAM_CERES_HARNESS_FUNCTOR_H_
#include <cmath>

struct StereoLineDistances {
  StereoLineDistances(const double* ob, double baseline) : b_(baseline) { for (int i = 0; i < 8; ++i) ob_[i] = ob[i]; }
  template <typename T>
  bool operator()(const T* const camera, const T* const line, T* residuals) const {
    const T sa = sin(line[0]), ca = cos(line[0]), sb = sin(line[1]), cb = cos(line[1]), sg = sin(line[2]), cg = cos(line[2]);
    const T cot = cos(line[3]) / sin(line[3]);
    // second and third column of R_l = Rz(g) Ry(b) Rx(a)
    const T col1[3] = { sa * sb * cg - ca * sg, sa * sb * sg + ca * cg, sa * cb };
    const T col2[3] = { ca * sb * cg + sa * sg, ca * sb * sg - sa * cg, ca * cb };
    const T cp[3] = { -cot * col2[0], -cot * col2[1], -cot * col2[2] };
    T p[3], d[3];
    ceres::AngleAxisRotatePoint(camera, cp, p);
    ceres::AngleAxisRotatePoint(camera, col1, d);
    p[0] += camera[3]; p[1] += camera[4]; p[2] += camera[5];
    for (int k = 0; k < 2; ++k) {
      const T px = p[0] - T(k * b_);
      const T n0 = p[1] * d[2] - p[2] * d[1], n1 = p[2] * d[0] - px * d[2], n2 = px * d[1] - p[1] * d[0];
      const T inv = T(1.0) / sqrt(n0 * n0 + n1 * n1);
      for (int e = 0; e < 2; ++e)
        residuals[2 * k + e] = -(T(ob_[4 * k + 2 * e]) * n0 + T(ob_[4 * k + 2 * e + 1]) * n1 + n2) * inv;
    }
    return true;
  }
  double ob_[8], b_;
};

#endif
