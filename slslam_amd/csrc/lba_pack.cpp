// slslam_amd/csrc/lba_pack.cpp — see lba_pack.h.
#include "lba_pack.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace slslam {

int pack_window(const slslam_lba_window* w, PackedWindow* out) {
  if (!w || !out) return SLSLAM_ERR_INVALID_ARGUMENT;
  const int C = w->num_cameras, L = w->num_lines, M = w->num_observations;
  if (C < 0 || L < 0 || M < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (M > 0 && (!w->camera_index || !w->line_index || !w->fixed_index || !w->observations))
    return SLSLAM_ERR_INVALID_ARGUMENT;
  if ((C > 0 || L > 0) && !w->parameters) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (C > kMaxCams) return SLSLAM_ERR_UNSUPPORTED;
  PackedWindow& P = *out;
  P = PackedWindow();
  P.C = C; P.L = L; P.M = M;
  P.params0.assign(w->parameters, w->parameters + (size_t)6 * C + (size_t)4 * L);
  for (size_t i = 0; i < P.params0.size(); ++i) if (!std::isfinite(P.params0[i])) return SLSLAM_ERR_INVALID_ARGUMENT;

  // block constness: one flagged observation makes the block constant (lba_problem.cpp:88-91)
  std::vector<char> cam_const(C, 0), cam_used(C, 0), line_const(L, 0);
  std::vector<int> line_cnt(L, 0);
  for (int i = 0; i < M; ++i) {
    const int c = w->camera_index[i], l = w->line_index[i];
    if (c < 0 || c >= C || l < 0 || l >= L) return SLSLAM_ERR_INVALID_ARGUMENT;
    cam_used[c] = 1; line_cnt[l]++;
    if (w->fixed_index[2 * i]) cam_const[c] = 1;
    if (w->fixed_index[2 * i + 1]) line_const[l] = 1;
    for (int q = 0; q < 8; ++q) if (!std::isfinite(w->observations[8 * (size_t)i + q])) return SLSLAM_ERR_INVALID_ARGUMENT;
  }
  P.cam_cf.assign(C, -1);
  for (int c = 0; c < C; ++c) if (cam_used[c] && !cam_const[c]) P.cam_cf[c] = P.Cf++;
  if (P.Cf > kMaxFreeCams) return SLSLAM_ERR_UNSUPPORTED;
  P.cam_x.assign(w->parameters, w->parameters + (size_t)6 * C);

  // group width class per line
  std::vector<int> glog2(L);
  for (int l = 0; l < L; ++l) {
    if (line_cnt[l] > 64) return SLSLAM_ERR_UNSUPPORTED;
    int g = 1;
    while ((1 << g) < line_cnt[l]) ++g;
    glog2[l] = g;
  }
  P.line_order.resize(L);
  std::iota(P.line_order.begin(), P.line_order.end(), 0);
  std::stable_sort(P.line_order.begin(), P.line_order.end(), [&](int a, int b) { return glog2[a] < glog2[b]; });
  std::vector<int> line_pos(L);
  for (int s = 0; s < L; ++s) line_pos[P.line_order[s]] = s;

  P.line_ptr.assign(L + 1, 0);
  for (int s = 0; s < L; ++s) P.line_ptr[s + 1] = P.line_ptr[s] + line_cnt[P.line_order[s]];
  P.line_flags.resize(L);
  P.line_u.resize((size_t)4 * L);
  int free_lines = 0;
  for (int s = 0; s < L; ++s) {
    const int l = P.line_order[s];
    P.line_flags[s] = line_const[l] ? 1 : 0;
    for (int a = 0; a < 4; ++a) P.line_u[4 * (size_t)s + a] = w->parameters[(size_t)6 * C + 4 * (size_t)l + a];
    if (!line_const[l] && line_cnt[l] > 0) ++free_lines;
  }
  P.nfree_params = 6 * P.Cf + 4 * free_lines;

  // observations grouped by line; inside a line: free cameras first (ascending free index)
  P.ob_orig.assign(M, 0);
  {
    std::vector<int> fill(L, 0);
    for (int i = 0; i < M; ++i) {
      const int s = line_pos[w->line_index[i]];
      P.ob_orig[P.line_ptr[s] + fill[s]++] = i;
    }
    for (int s = 0; s < L; ++s) {
      auto key = [&](int i) { const int c = w->camera_index[i]; return P.cam_cf[c] >= 0 ? P.cam_cf[c] : P.Cf + c; };
      std::stable_sort(P.ob_orig.begin() + P.line_ptr[s], P.ob_orig.begin() + P.line_ptr[s + 1],
                       [&](int a, int b) { return key(a) < key(b); });
    }
  }
  P.ob_cam.resize(M);
  P.ob.resize((size_t)8 * M);
  P.nkept = 0;
  for (int o = 0; o < M; ++o) {
    const int i = P.ob_orig[o];
    P.ob_cam[o] = w->camera_index[i];
    for (int q = 0; q < 8; ++q) P.ob[(size_t)q * M + o] = w->observations[8 * (size_t)i + q];
    if (!(cam_const[w->camera_index[i]] && line_const[w->line_index[i]])) ++P.nkept;
  }

  // tiles and their off-diagonal camera-pair work items
  for (int s = 0; s < L;) {
    const int g = glog2[P.line_order[s]];
    const int per_tile = 64 >> g;
    int e = s;
    while (e < L && e - s < per_tile && glog2[P.line_order[e]] == g) ++e;
    Tile t;
    t.line_begin = s; t.nlines = (int16_t)(e - s); t.glog2 = (int16_t)g;
    t.item_off = (int)(P.items.size() / 2);
    for (int q = s; q < e; ++q) {
      if (P.line_flags[q] & 1) continue;                   // constant line: nothing to eliminate
      const int base_lane = (q - s) << g;
      const int k = P.line_ptr[q + 1] - P.line_ptr[q];
      int kf = 0;                                          // free-camera observations come first
      while (kf < k && P.cam_cf[P.ob_cam[P.line_ptr[q] + kf]] >= 0) ++kf;
      for (int i = 0; i < kf; ++i)
        for (int j = i + 1; j < kf; ++j) { P.items.push_back((uint8_t)(base_lane + i)); P.items.push_back((uint8_t)(base_lane + j)); }
    }
    t.nitems = (int)(P.items.size() / 2) - t.item_off;
    P.tiles.push_back(t);
    s = e;
  }
  return SLSLAM_OK;
}

std::vector<int> chunk_boundaries(int ntiles, int tiles_per_chunk) {
  std::vector<int> b;
  if (tiles_per_chunk < 1) tiles_per_chunk = 1;
  const int nchunks = ntiles > 0 ? (ntiles + tiles_per_chunk - 1) / tiles_per_chunk : 0;
  b.push_back(0);
  for (int c = 1; c <= nchunks; ++c) b.push_back((int)(((long long)ntiles * c) / nchunks));
  return b;
}

}  // namespace slslam
