// tests/host_cxx/ceres_functor_kat.cpp — the residual functor of the optional true-Ceres harness (tools/ceres_harness_functor.h) instantiated for
// double on the rotation helpers of slslam_amd/host/ceres/rotation.h, against the survey's known-answer vector and at the
// identity keyframe.  Prints the largest difference.
#include <cstdio>
#include <cmath>
#include "ceres/rotation.h"
using std::sin; using std::cos; using std::sqrt;
#include "ceres_harness_functor.h"

int main() {
  const double line[4] = { 0.3, -0.4, 0.5, 0.6 };
  const double ob[8] = { 0.10, 0.05, -0.20, 0.15, 0.08, 0.05, -0.22, 0.15 };
  const double cam[6] = { 0.01, -0.02, 0.03, 0.10, -0.20, 0.30 }, cam0[6] = { 0, 0, 0, 0, 0, 0 };
  const double want[4] = { -0.6769197527221315, -0.4611016357636927, -0.5592511525774213, -0.350041545453428 };
  const double want0[4] = { -0.5345227518574093, -0.31919053715071777, -0.44312099656355386, -0.23270941889772545 };
  StereoLineDistances f(ob, 0.12);
  double r[4], r0[4], worst = 0.0;
  f(cam, line, r);
  f(cam0, line, r0);
  for (int i = 0; i < 4; ++i) worst = std::fmax(worst, std::fmax(std::fabs(r[i] - want[i]), std::fabs(r0[i] - want0[i])));
  std::printf("%.3e\n", worst);
  return 0;
}
