// tests/host_math/host_math.cpp — compiles the product's device math header for the HOST so the
// CPU test-suite can compare the analytic Jacobians with the oracle's dual numbers without a GPU.
// Test infrastructure only: nothing in the product links this.
#include "../../slslam_amd/csrc/lba_math.h"

extern "C" {
void hm_cam_prepare(const double* w, double* R, double* JL) { slslam::cam_prepare<double>(w, R, JL); }
void hm_obs_linearise(const double* cam, const double* line, const double* obs, double baseline,
                      double* r, double* jc, double* jl) {
  double R[9], JL[9], trig[7], cp[3], dv[3], dcp[12], ddv[9];
  slslam::cam_prepare<double>(cam, R, JL);
  slslam::line_trig<double>(line, trig);
  slslam::line_points_jac<double>(trig, cp, dv, dcp, ddv);
  slslam::obs_linearise<double>(R, JL, cam + 3, cp, dv, dcp, ddv, obs, baseline, r, jc, jl);
}
void hm_obs_residual(const double* cam, const double* line, const double* obs, double baseline, double* r) {
  double R[9], trig[7], cp[3], dv[3];
  slslam::cam_rotation<double>(cam, R);
  slslam::line_trig<double>(line, trig);
  slslam::line_points<double>(trig, cp, dv);
  slslam::obs_residual<double>(R, cam + 3, cp, dv, obs, baseline, r);
}
double hm_huber(double s, double a, double* cost) { return slslam::huber_scale<double>(s, a, cost); }
}
