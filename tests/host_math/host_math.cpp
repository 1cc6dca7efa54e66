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
// w = J_l^T (J_c y) of the back-substitution, contracted without forming the Jacobians (obs_backsub_w): y = (yw, yt)
void hm_obs_backsub_w(const double* cam, const double* line, const double* obs, double baseline, const double* y,
                      double* r, double* w) {
  double R[9], JL[9], trig[7], cp[3], dv[3], dcp[12], ddv[9], vw[3];
  slslam::cam_prepare<double>(cam, R, JL);
  for (int i = 0; i < 3; ++i) vw[i] = JL[3 * i] * y[0] + JL[3 * i + 1] * y[1] + JL[3 * i + 2] * y[2];
  slslam::line_trig<double>(line, trig);
  slslam::line_points_jac<double>(trig, cp, dv, dcp, ddv);
  slslam::obs_backsub_w<double>(R, cam + 3, vw, y + 3, cp, dv, dcp, ddv, obs, baseline, r, w);
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

// ---- pack inspection (host-only code of the product, lba_pack.cpp)
#include "../../slslam_amd/csrc/lba_pack.h"
#include <cstring>
extern "C" {
// Packs one window and copies the layout into caller buffers (sized generously by the test).
int hm_pack(int C, int L, int M, const int* cam, const int* line, const int* fixed, const double* obs, double* params,
            int* out_counts /*Cf, ntiles, nitems, nfree_params, nkept*/, int* line_order, int* line_ptr, int* ob_orig,
            int* ob_cam, int* tiles /*4 ints per tile: line_begin,nlines,flags,nitems*/, unsigned char* items, int* cam_cf,
            int max_tiles, int max_items, unsigned short* lane_map /*64 per tile*/) {
  slslam_lba_window w{C, L, M, cam, line, fixed, obs, params};
  slslam::PackedWindow P;
  const int rc = slslam::pack_window(&w, &P);
  if (rc) return rc;
  if ((int)P.tiles.size() > max_tiles || (int)P.items.size() / 2 > max_items) return -1;
  out_counts[0] = P.Cf; out_counts[1] = (int)P.tiles.size(); out_counts[2] = (int)P.items.size() / 2;
  out_counts[3] = P.nfree_params; out_counts[4] = P.nkept;
  std::memcpy(line_order, P.line_order.data(), sizeof(int) * L);
  std::memcpy(line_ptr, P.line_ptr.data(), sizeof(int) * (L + 1));
  std::memcpy(ob_orig, P.ob_orig.data(), sizeof(int) * M);
  std::memcpy(ob_cam, P.ob_cam.data(), sizeof(int) * M);
  std::memcpy(cam_cf, P.cam_cf.data(), sizeof(int) * C);
  for (size_t t = 0; t < P.tiles.size(); ++t) {
    tiles[4 * t] = P.tiles[t].line_begin; tiles[4 * t + 1] = P.tiles[t].nlines;
    tiles[4 * t + 2] = P.tiles[t].flags; tiles[4 * t + 3] = P.tiles[t].nitems;
  }
  std::memcpy(items, P.items.data(), P.items.size());
  std::memcpy(lane_map, P.lane_map.data(), P.lane_map.size() * sizeof(unsigned short));
  return 0;
}
}
