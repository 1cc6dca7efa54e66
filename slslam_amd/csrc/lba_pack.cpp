// slslam_amd/csrc/lba_pack.cpp — see lba_pack.h.
#include "lba_pack.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <set>

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

  // lane runs: a line takes max(k, 1) consecutive lanes.  Runs are bin-packed (first fit, decreasing) into
  // 16-lane rows, rows into 4-row tiles; the sorted line order is the order in which lines appear in the tiles.
  //   * lines with more than 16 observations take whole rows of one tile,
  //   * lines with 4..16 observations share rows; the rows are dealt to the tiles so that every tile gets about
  //     the same number of off-diagonal pair items (the pair passes of a tile cost ceil(items / 64)),
  //   * lines with fewer than 4 observations (which need more than one sin/cos round per lane in the
  //     back-substitution) are kept in tiles of their own.
  std::vector<int> kfree(L, 0);
  for (int i = 0; i < M; ++i) if (P.cam_cf[w->camera_index[i]] >= 0) kfree[w->line_index[i]]++;
  auto lanes_of = [&](int l) { return std::max(line_cnt[l], 1); };
  auto items_of = [&](int l) { return line_const[l] ? 0 : (kfree[l] * (kfree[l] - 1)) / 2; };
  struct Row { int used = 0, items = 0; std::vector<int> lines; };
  auto pack_rows = [&](std::vector<int> ls) {
    std::stable_sort(ls.begin(), ls.end(), [&](int x, int y) { return lanes_of(x) > lanes_of(y); });
    std::vector<Row> rows;
    std::set<int> open_by_room[17];             // rows indexed by remaining room
    for (int l : ls) {
      const int need = lanes_of(l);
      int best = -1;                            // first fit: the oldest row with enough room
      for (int room = need; room <= 16; ++room)
        if (!open_by_room[room].empty() && (best < 0 || *open_by_room[room].begin() < best)) best = *open_by_room[room].begin();
      if (best < 0) { best = (int)rows.size(); rows.push_back(Row()); }
      else open_by_room[16 - rows[best].used].erase(best);
      rows[best].used += need; rows[best].items += items_of(l); rows[best].lines.push_back(l);
      if (rows[best].used < 16) open_by_room[16 - rows[best].used].insert(best);
    }
    return rows;
  };
  std::vector<int> big, mid, small;
  for (int l = 0; l < L; ++l) {
    if (line_cnt[l] > 64) return SLSLAM_ERR_UNSUPPORTED;
    (line_cnt[l] > 16 ? big : line_cnt[l] >= 4 ? mid : small).push_back(l);
  }
  struct TileRows { std::vector<std::vector<int>> rows; };   // a multi-row line is one entry of `rows` holding one line
  std::vector<TileRows> tplan;
  {
    int used_rows = 4;
    std::stable_sort(big.begin(), big.end(), [&](int x, int y) { return line_cnt[x] > line_cnt[y]; });
    for (int l : big) {
      const int nr = (line_cnt[l] + 15) / 16;
      if (used_rows + nr > 4) { tplan.push_back(TileRows()); used_rows = 0; }
      tplan.back().rows.push_back(std::vector<int>(1, l));
      used_rows += nr;
    }
  }
  {
    std::vector<Row> rows = pack_rows(mid);
    std::stable_sort(rows.begin(), rows.end(), [](const Row& x, const Row& y) { return x.items > y.items; });
    const int R = (int)rows.size(), T = (R + 3) / 4, base = (int)tplan.size();
    tplan.resize(base + T);
    for (int r = 0; r < R; ++r) {                 // boustrophedon deal: heavy rows meet light rows
      const int pass = r / T, pos = r % T;
      tplan[base + ((pass & 1) ? T - 1 - pos : pos)].rows.push_back(rows[r].lines);
    }
  }
  {
    const std::vector<Row> rows = pack_rows(small);
    for (size_t r = 0; r < rows.size(); ++r) {
      if (r % 4 == 0) tplan.push_back(TileRows());
      tplan.back().rows.push_back(rows[r].lines);
    }
  }
  P.line_order.clear();
  for (const TileRows& tr : tplan) for (const auto& row : tr.rows) for (int l : row) P.line_order.push_back(l);
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

  // tiles, their lane maps and their off-diagonal camera-pair work items
  {
    int s = 0;
    for (const TileRows& tr : tplan) {
      Tile t;
      t.line_begin = s; t.item_off = (int)(P.items.size() / 2);
      std::vector<uint16_t> map(64, (uint16_t)0x00FF);
      int lane = 0, nl = 0, min_lanes = 64, max_run = 1, multi = 0;
      for (const auto& row : tr.rows) {
        lane = (lane + 15) & ~15;                            // every entry of the plan starts a row
        for (size_t q = 0; q < row.size(); ++q, ++s, ++nl) {
          const int k = P.line_ptr[s + 1] - P.line_ptr[s], run = std::max(k, 1);
          for (int j = 0; j < run; ++j) map[lane + j] = (uint16_t)(nl | (j << 8));
          min_lanes = std::min(min_lanes, run);
          max_run = std::max(max_run, std::min(run, 16));
          if (run > 16) multi = 1;
          if (!(P.line_flags[s] & 1)) {
            int kf = 0;                                      // free-camera observations come first
            while (kf < k && P.cam_cf[P.ob_cam[P.line_ptr[s] + kf]] >= 0) ++kf;
            for (int i = 0; i < kf; ++i)
              for (int j = i + 1; j < kf; ++j) { P.items.push_back((uint8_t)(lane + i)); P.items.push_back((uint8_t)(lane + j)); }
          }
          lane += run;
        }
      }
      const int rounds_log2 = min_lanes >= 4 ? 0 : min_lanes >= 2 ? 1 : 2;
      t.nlines = (int16_t)nl;
      t.flags = (int16_t)(multi | (rounds_log2 << 1) | (max_run << 3));
      t.nitems = (int)(P.items.size() / 2) - t.item_off;
      P.tiles.push_back(t);
      P.lane_map.insert(P.lane_map.end(), map.begin(), map.end());
    }
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
