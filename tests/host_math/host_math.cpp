// tests/host_math/host_math.cpp — compiles the product's device math header for the HOST so the
// CPU test-suite can compare the analytic Jacobians with the oracle's dual numbers without a GPU.
// Test infrastructure only: nothing in the product links this.
#include "../../slslam_amd/csrc/lba_math.h"
#include "../../slslam_amd/csrc/lba_gram.h"

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
// The Gram formulation of the elimination sweep (lba_gram.h): blocks of one observation in RAW camera coordinates
// (J_c' = [tau | gP], i.e. without the SO(3) left Jacobian and without any scaling), line columns scaled by sl.
// out: r[4], D[21] = J_c'^T J_c', gc[6] = J_c'^T r, G[24] = J_c'^T J_l (row-major 6x4), H[10] = J_l^T J_l, gl[4] = J_l^T r.
void hm_obs_gram_blocks(const double* cam, const double* line, const double* obs, double baseline, const double* sl,
                        double* r, double* D, double* gc, double* G, double* H, double* gl) {
  double R[9], JL[9], trig[7], dc[3], e2[3], P[3], W[21], w[6], Ml[24], Y[24];
  slslam::cam_prepare<double>(cam, R, JL);
  slslam::line_trig<double>(line, trig);
  slslam::obs_gram<double>(R, cam + 3, trig, obs, baseline, dc, e2, P, r, W, w);
  const double Q[3] = { -trig[6] * e2[0], -trig[6] * e2[1], -trig[6] * e2[2] };
  const double zc[3] = { R[2], R[5], R[8] };
  slslam::gram_camera_block<double>(W, Q, dc, D);
  slslam::apply_mc<double>(Q, dc, w, gc);
  slslam::line_rows<double>(dc, e2, zc, trig[6], trig[1], trig[0], sl, Ml);
  slslam::gram_line<double>(W, w, Ml, Y, H, gl);
  for (int j = 0; j < 4; ++j) {
    double col[6];
    slslam::apply_mc<double>(Q, dc, Y + 6 * j, col);
    for (int a = 0; a < 6; ++a) G[4 * a + j] = col[a];
  }
}
double hm_huber(double s, double a, double* cost) { return slslam::huber_scale<double>(s, a, cost); }
// The grouped sweep's linearisation (obs_linearise_raw): robustified residual, J_c' = [tau | gP] sqrt(rho') (raw camera
// coordinates), J_l sl sqrt(rho'), block cost
void hm_obs_linearise_raw(const double* cam, const double* line, const double* obs, double baseline, double huber_delta,
                          const double* sl, double* rs, double* jc, double* jl, double* cost) {
  double R[9], trig[7];
  slslam::cam_rotation<double>(cam, R);
  slslam::line_trig<double>(line, trig);
  slslam::obs_linearise_raw<double>(R, cam + 3, trig, sl, obs, baseline, huber_delta, rs, jl, cost,
                                    [&](int row, const double (&j)[6]) { for (int a = 0; a < 6; ++a) jc[6 * row + a] = j[a]; });
}
// The mixed-precision form of the same (lba_precision = 1): residuals, cost and J_l in double, J_c' in float (returned widened)
void hm_obs_linearise_raw_mixed(const double* cam, const double* line, const double* obs, double baseline, double huber_delta,
                                const double* sl, double* rs, double* jc, double* jl, double* cost) {
  double R[9], trig[7];
  slslam::cam_rotation<double>(cam, R);
  slslam::line_trig<double>(line, trig);
  slslam::obs_linearise_raw_mixed(R, cam + 3, trig, sl, obs, baseline, huber_delta, rs, jl, cost,
                                  [&](int row, const float (&j)[6]) { for (int a = 0; a < 6; ++a) jc[6 * row + a] = (double)j[a]; });
}
}

// ---- pack inspection (host-only code of the product, lba_pack.cpp)
#include "../../slslam_amd/csrc/lba_pack.h"
#include <cstring>
// (memcpy with a null source is undefined even for zero bytes: an empty vector's data() may be null)
static void copy_bytes(void* dst, const void* src, size_t n) { if (n) std::memcpy(dst, src, n); }
extern "C" {
int hm_pack_g(int C, int L, int M, const int* cam, const int* line, const int* fixed, const double* obs, double* params,
              int* out_counts, int* line_order, int* line_ptr, int* ob_orig, int* ob_cam, int* tiles, unsigned char* items, int* cam_cf,
              int max_tiles, int max_items, unsigned short* lane_map, int grouping, unsigned* line_desc);
// Packs one window and copies the layout into caller buffers (sized generously by the test).
int hm_pack(int C, int L, int M, const int* cam, const int* line, const int* fixed, const double* obs, double* params,
            int* out_counts /*Cf, ntiles, nitems, nfree_params, nkept*/, int* line_order, int* line_ptr, int* ob_orig,
            int* ob_cam, int* tiles /*4 ints per tile: line_begin,nlines,flags,nitems*/, unsigned char* items, int* cam_cf,
            int max_tiles, int max_items, unsigned short* lane_map /*64 per tile*/) {
  return hm_pack_g(C, L, M, cam, line, fixed, obs, params, out_counts, line_order, line_ptr, ob_orig, ob_cam, tiles, items, cam_cf,
                   max_tiles, max_items, lane_map, 0, nullptr);
}
// ... with the packer's grouping mode (1: lines by first free camera, the grouped matrix-core sweep; the window is packed the
// default way first and packed again from the packed arrays, as the library does) and the line descriptors
int hm_pack_g(int C, int L, int M, const int* cam, const int* line, const int* fixed, const double* obs, double* params,
              int* out_counts, int* line_order, int* line_ptr, int* ob_orig, int* ob_cam, int* tiles, unsigned char* items, int* cam_cf,
              int max_tiles, int max_items, unsigned short* lane_map, int grouping, unsigned* line_desc) {
  slslam_lba_window w{C, L, M, cam, line, fixed, obs, params};
  slslam::PackedWindow P;
  int rc = slslam::pack_window(&w, &P);
  if (rc) return rc;
  if (grouping) {
    slslam::PackedWindow Q;
    rc = slslam::repack_window(P, grouping, &Q);
    if (rc) return rc;
    P = Q;
  }
  if (line_desc) copy_bytes(line_desc, P.line_desc.data(), sizeof(unsigned) * L);
  if ((int)P.tiles.size() > max_tiles || (int)P.items.size() / 2 > max_items) return -1;
  out_counts[0] = P.Cf; out_counts[1] = (int)P.tiles.size(); out_counts[2] = (int)P.items.size() / 2;
  out_counts[3] = P.nfree_params; out_counts[4] = P.nkept;
  copy_bytes(line_order, P.line_order.data(), sizeof(int) * L);
  copy_bytes(line_ptr, P.line_ptr.data(), sizeof(int) * (L + 1));
  copy_bytes(ob_orig, P.ob_orig.data(), sizeof(int) * M);
  copy_bytes(ob_cam, P.ob_cam.data(), sizeof(int) * M);
  copy_bytes(cam_cf, P.cam_cf.data(), sizeof(int) * C);
  for (size_t t = 0; t < P.tiles.size(); ++t) {
    tiles[4 * t] = P.tiles[t].line_begin; tiles[4 * t + 1] = P.tiles[t].nlines;
    tiles[4 * t + 2] = P.tiles[t].flags; tiles[4 * t + 3] = P.tiles[t].nitems;
  }
  copy_bytes(items, P.items.data(), P.items.size());
  copy_bytes(lane_map, P.lane_map.data(), P.lane_map.size() * sizeof(unsigned short));
  return 0;
}
}

// The three destinations of a pack's observations (lba_pack.h): the packed window itself (planes in P.ob), caller planes (a refill with the gather on
// the host: planes_out[8 M]), or - a refill whose batch permutes on the device - the RAW copy in the caller's order (raw_out[8 M]; the planes stay
// untouched).  Returns the packer's status; ob_orig / ob_cam / nkept of the pack come back in every mode.
extern "C" int hm_pack_dest(int C, int L, int M, const int* cam, const int* line, const int* fixed, const double* obs, double* params, int grouping,
                            int mode /*0 own, 1 planes, 2 raw*/, double* planes_out, double* raw_out, int* ob_orig, int* ob_cam, int* nkept, double* own_out) {
  slslam_lba_window w{C, L, M, cam, line, fixed, obs, params};
  slslam::PackedWindow P;
  slslam::ObPlanes d;
  for (int q = 0; q < 4; ++q) d.plane[q] = planes_out + (size_t)2 * q * M;
  if (mode == 2) d.raw = raw_out;
  const int rc = slslam::pack_window(&w, &P, grouping, mode == 0 ? nullptr : &d);
  if (rc) return rc;
  copy_bytes(ob_orig, P.ob_orig.data(), sizeof(int) * M);
  copy_bytes(ob_cam, P.ob_cam.data(), sizeof(int) * M);
  *nkept = P.nkept;
  if (mode == 0) copy_bytes(own_out, P.ob.data(), sizeof(double) * 8 * M);
  return (mode != 0 && !P.ob.empty()) ? -2 : 0;
}

// ---- the matrix-core elimination's index maps (lba_eliminate_mfma.h), exported so that the CPU suite can replay one tile
#include "../../slslam_amd/csrc/lba_eliminate_mfma_maps.h"
extern "C" {
// line descriptors of the packed window (free-camera mask | first lane << 16) and the duplicate-observation flag
int hm_line_desc(int C, int L, int M, const int* cam, const int* line, const int* fixed, const double* obs, double* params,
                 unsigned* line_desc, int* dup) {
  slslam_lba_window w{C, L, M, cam, line, fixed, obs, params};
  slslam::PackedWindow P;
  const int rc = slslam::pack_window(&w, &P);
  if (rc) return rc;
  std::memcpy(line_desc, P.line_desc.data(), sizeof(unsigned) * L);
  *dup = P.dup_free_obs ? 1 : 0;
  return 0;
}
// where lane `lane` stores entry (a, k) of its F block / where it fetches its operand of block I for a line (mask, first lane);
// returns -1 when the lane's camera does not see the line
int hm_panel_store(int lane, int a, int k) { return slslam::panel_store_index(lane, a, k); }
int hm_panel_fetch(int lane, int I, unsigned mask, int first_lane) { return slslam::panel_fetch_index(lane, I, mask, first_lane); }
unsigned hm_block_cam_mask(int I) { return slslam::block_cam_mask(I); }
int hm_ptile_of(int nw, int w, int e) { return slslam::ptile_of(nw, w, e); }
// accumulator entry (tile t, register q, lane) -> row / column of the stacked system
void hm_acc_rc(int t, int q, int lane, int* row, int* col) { slslam::acc_row_col(t, q, lane, row, col); }
}

// ---- the grouped matrix-core elimination's index maps (lba_eliminate_grouped.h)
#include "../../slslam_amd/csrc/lba_eliminate_grouped_maps.h"
extern "C" {
int hm_gp_store(int lane, int a, int k) { return slslam::gp_store_index(lane, a, k); }
int hm_gp_fetch(int lane, int r, unsigned d) { return slslam::gp_fetch_index(lane, r, d); }
int hm_gp_flush(int r, int c, int a, int q, int lane, int n) { return slslam::gp_flush_index(r, c, a, q, lane, n); }
int hm_gp_panel_doubles() { return slslam::kGpPanel; }
}

// ---- chunk boundaries (lba_pack.cpp): equal and graded cuts of a window's tiles
extern "C" {
int hm_chunk_boundaries(int ntiles, int tiles_per_chunk, int* out, int cap) {
  const std::vector<int> b = slslam::chunk_boundaries(ntiles, tiles_per_chunk);
  for (size_t i = 0; i < b.size() && (int)i < cap; ++i) out[i] = b[i];
  return (int)b.size();
}
int hm_chunk_boundaries_graded(int ntiles, int nchunks, const int* weights, int* out, int cap) {
  const std::vector<int> b = slslam::chunk_boundaries_graded(ntiles, nchunks, weights);
  for (size_t i = 0; i < b.size() && (int)i < cap; ++i) out[i] = b[i];
  return (int)b.size();
}
}
