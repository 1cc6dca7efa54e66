// slslam_amd/csrc/lba_pack.cpp — see lba_pack.h.
#include "lba_pack.h"
#include "lba_eliminate_grouped_maps.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace slslam {

namespace {
// true iff every value is finite: exponent field all ones <=> NaN / Inf.  Branch-free integer form (the compiler vectorises it).
bool all_finite(const double* v, size_t n) {
  uint64_t acc = 0;
  for (size_t i = 0; i < n; ++i) {
    uint64_t x;
    std::memcpy(&x, v + i, 8);
    acc |= ((x & 0x7ff0000000000000ull) + 0x0010000000000000ull) & 0x8000000000000000ull;
  }
  return acc == 0;
}
}  // namespace

int pack_window(const slslam_lba_window* w, PackedWindow* out, int grouping, const ObPlanes* ob_dest) {
  if (!w || !out) return SLSLAM_ERR_INVALID_ARGUMENT;
  const int C = w->num_cameras, L = w->num_lines, M = w->num_observations;
  if (C < 0 || L < 0 || M < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (M > 0 && (!w->camera_index || !w->line_index || !w->fixed_index || !w->observations))
    return SLSLAM_ERR_INVALID_ARGUMENT;
  if ((C > 0 || L > 0) && !w->parameters) return SLSLAM_ERR_INVALID_ARGUMENT;
  PackedWindow& P = *out;
  // (the vectors of `out` keep their capacity: a batch that is refilled packs into the windows of the refill before last)
  P.Cf = 0; P.nfree_params = 0; P.nkept = 0; P.big = false; P.dup_free_obs = false;
  P.tiles.clear(); P.lane_map.clear(); P.items.clear(); P.ob.clear();
  P.C = C; P.L = L; P.M = M; P.grouping = grouping ? 1 : 0;
  P.params0.assign(w->parameters, w->parameters + (size_t)6 * C + (size_t)4 * L);
  // (the observations are tested where they are read anyway, in the gather below: one pass over them instead of two)
  if (!all_finite(P.params0.data(), P.params0.size())) return SLSLAM_ERR_INVALID_ARGUMENT;

  // block constness: one flagged observation makes the block constant (lba_problem.cpp:88-91)
  std::vector<char> cam_const(C, 0), cam_used(C, 0), line_const(L, 0);
  std::vector<int> line_cnt(L, 0);
  for (int i = 0; i < M; ++i) {
    const int c = w->camera_index[i], l = w->line_index[i];
    if (c < 0 || c >= C || l < 0 || l >= L) return SLSLAM_ERR_INVALID_ARGUMENT;
    cam_used[c] = 1; line_cnt[l]++;
    if (w->fixed_index[2 * i]) cam_const[c] = 1;
    if (w->fixed_index[2 * i + 1]) line_const[l] = 1;
  }
  P.cam_cf.assign(C, -1);
  for (int c = 0; c < C; ++c) if (cam_used[c] && !cam_const[c]) P.cam_cf[c] = P.Cf++;
  // windows beyond what the tiled sweeps hold on chip take the global-memory path (lba_big.h): no tiles are built for them
  P.big = C > kMaxCams || P.Cf > kMaxFreeCams;
  for (int l = 0; l < L && !P.big; ++l) if (line_cnt[l] > 64) P.big = true;
  P.cam_x.assign(w->parameters, w->parameters + (size_t)6 * C);

  // lane runs: a line takes max(k, 1) consecutive lanes.  Runs are bin-packed (best fit, decreasing) into
  // 16-lane rows, rows into 4-row tiles; the sorted line order is the order in which lines appear in the tiles.
  //   * lines with more than 16 observations take whole rows of one tile,
  //   * lines with 4..16 observations share rows; the rows are dealt to the tiles so that every tile gets about
  //     the same number of off-diagonal pair items (the pair passes of a tile cost ceil(items / 64)),
  //   * lines with fewer than 4 observations (which need more than one sin/cos round per lane in the
  //     back-substitution) are kept in tiles of their own.
  std::vector<int> kfree(L, 0);
  std::vector<unsigned> fmask(L, 0u);            // free cameras that see the line
  for (int i = 0; i < M && !P.big; ++i)        // (oversize windows have up to 64 free cameras and no tiles: no masks, no work items)
    if (P.cam_cf[w->camera_index[i]] >= 0) {
      const unsigned bit = 1u << P.cam_cf[w->camera_index[i]];
      if (fmask[w->line_index[i]] & bit) P.dup_free_obs = true;
      kfree[w->line_index[i]]++; fmask[w->line_index[i]] |= bit;
    }
  auto lanes_of = [&](int l) { return std::max(line_cnt[l], 1); };
  auto items_of = [&](int l) { return line_const[l] ? 0 : (kfree[l] * (kfree[l] - 1)) / 2; };
  struct Row { int used, items, head, tail; unsigned mask; };   // the lines of a row are chained through `next`; mask: free cameras
  std::vector<Row> rows;
  std::vector<int> next(L, -1);
  rows.reserve((size_t)L / 2 + 8);
  auto append = [&](int r, int l) {
    if (rows[r].head < 0) rows[r].head = l; else next[rows[r].tail] = l;
    rows[r].tail = l; rows[r].used += lanes_of(l); rows[r].items += items_of(l); rows[r].mask |= fmask[l];
  };
  // best fit over lines of decreasing length (counting sort by length, original order inside a length): a line goes to
  // the fullest open row that still holds it, else it opens a row.  Returns the range of row ids created.
  std::vector<int> tile_rows;                      // row ids, tile after tile
  std::vector<int> tile_ptr(1, 0);
  if (!P.big) {
    std::vector<int> by_len[17];
    bool carry_open = false;                         // grouping: the open rows survive from one call to the next
    struct OpenRow { int row; unsigned mask; };      // an open row with the free cameras of its lines (beside the id: the search below reads nothing else)
    std::vector<OpenRow> open_keep[17];
    static const bool no_disjoint = std::getenv("SLSLAM_PACK_NO_DISJOINT") != nullptr;       // (experiment: the camera-disjoint preference off for the grouped packing)
    const int first_pass = (grouping && no_disjoint) ? 1 : 0;
    auto pack_rows = [&](int len_lo, int len_hi) {
      const int first = (int)rows.size();
      std::vector<OpenRow> open_local[17];
      std::vector<OpenRow>* open_by_room = carry_open ? open_keep : open_local;
      for (int len = len_hi; len >= len_lo; --len)
        for (int l : by_len[len]) {
          // the fullest open row that holds the line - and, among the rows of that fill, preferably one none of whose
          // lines shares a free camera with it: the lanes of one 16-lane row that add to the same camera record
          // serialise in the LDS (tools/micro/lds_atomic_bench.hip)
          int r = -1;
          const unsigned fm = fmask[l];
          for (int pass = first_pass; pass < 2 && r < 0; ++pass)          // pass 0: rows without a common free camera only
            for (int room = len; room <= 16 && r < 0; ++room) {
              std::vector<OpenRow>& cand = open_by_room[room];
              for (size_t c = cand.size(); c-- > 0 && cand.size() - c <= 32;)
                if (pass == 1 || !(cand[c].mask & fm)) { r = cand[c].row; cand.erase(cand.begin() + c); break; }
            }
          if (r < 0) { r = (int)rows.size(); rows.push_back(Row{0, 0, -1, -1, 0u}); }
          append(r, l);
          if (rows[r].used < 16) open_by_room[16 - rows[r].used].push_back(OpenRow{ r, rows[r].mask });
        }
      return std::make_pair(first, (int)rows.size());
    };
    std::vector<int> big;
    for (int l = 0; l < L; ++l) {
      if (line_cnt[l] > 16) big.push_back(l); else by_len[lanes_of(l)].push_back(l);
    }
    {
      int used_rows = 4;
      std::stable_sort(big.begin(), big.end(), [&](int x, int y) { return line_cnt[x] > line_cnt[y]; });
      for (int l : big) {                            // a long line is a row entry of its own that spans (k + 15) / 16 rows
        const int nr = (line_cnt[l] + 15) / 16;
        if (used_rows + nr > 4) { if (!tile_rows.empty()) tile_ptr.push_back((int)tile_rows.size()); used_rows = 0; }
        rows.push_back(Row{0, 0, -1, -1, 0u});
        append((int)rows.size() - 1, l);
        tile_rows.push_back((int)rows.size() - 1);
        used_rows += nr;
      }
      if (!tile_rows.empty()) tile_ptr.push_back((int)tile_rows.size());
    }
    // grouping: key of a line = first free camera that sees it (x 2, + 1 unless its camera range spans more than 8 cameras, i.e. a
    // fourth 16-row block of the reduced system counted from that camera); lines without elimination work (constant, or seen by
    // no free camera) come last
    auto group_key = [&](int l) -> int {
      const unsigned m = line_const[l] ? 0u : (fmask[l] & 0xfffffu);
      if (!m) return 1000;
      const int a = __builtin_ctz(m), hi = 31 - __builtin_clz(m);
      return 2 * a + ((hi - a + 1) > 8 ? 0 : 1);       // (the wide lines first: the lanes they leave free in their rows go to lines of their own group)
    };
    if (grouping) {
      // the two length classes of the default packing are kept (lines with fewer than 4 lanes in tiles of their own: the
      // back-substitution's sin/cos rounds), each of them group after group, the rows filling the tiles in that order
      std::vector<int> lens[2][17];
      for (int len = 1; len <= 16; ++len) { lens[len < 4 ? 1 : 0][len].swap(by_len[len]); }
      std::vector<int> key_of((size_t)L);
      for (int l = 0; l < L; ++l) key_of[(size_t)l] = group_key(l);
      for (int cls = 0; cls < 2; ++cls) {
        // keys are 2 a + {0, 1} with a < 20, or 1000 (no elimination work): 42 buckets, each with its lines by length in their original order
        enum { kBuckets = 42 };
        auto bucket_of = [](int key) { return key >= 1000 ? kBuckets - 1 : key; };
        std::vector<int> bucket[kBuckets][17];
        bool present[kBuckets] = {};
        for (int len = 1; len <= 16; ++len)
          for (int l : lens[cls][len]) { const int bq = bucket_of(key_of[(size_t)l]); bucket[bq][len].push_back(l); present[bq] = true; }
        const size_t first_row = tile_rows.size();
        const int row0 = (int)rows.size();
        carry_open = true;                             // a group's lines may finish the open rows of the group before it
        for (int q = 0; q <= 16; ++q) open_keep[q].clear();
        int prev_first = row0;                         // first row the previous group opened
        for (int bq = 0; bq < kBuckets; ++bq) {        // ascending keys
          if (!present[bq]) continue;
          for (int len = 1; len <= 16; ++len) by_len[len].swap(bucket[bq][len]);
          // (only the rows the previous group left open: an older row would put this group's line in the middle of another
          // group's tiles, and the sweep adds its accumulators to memory whenever the group changes)
          for (int q = 0; q <= 16; ++q) {
            std::vector<OpenRow>& v = open_keep[q];
            v.erase(std::remove_if(v.begin(), v.end(), [&](const OpenRow& r) { return r.row < prev_first; }), v.end());
          }
          prev_first = (int)rows.size();
          pack_rows(1, 16);
        }
        carry_open = false;
        {
          // rows in the order of the group of their FIRST line; a row that another group finished sits last among them, at the
          // seam between the two groups
          std::vector<int> order((int)rows.size() - row0);
          std::iota(order.begin(), order.end(), row0);
          auto mixed = [&](int r) { const int k0 = key_of[(size_t)rows[r].head]; for (int l = rows[r].head; l >= 0; l = next[l]) if (key_of[(size_t)l] != k0) return 1; return 0; };
          std::vector<int> rk(rows.size(), 0);
          for (int r : order) rk[r] = key_of[(size_t)rows[r].head] * 2 + mixed(r);
          std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return rk[x] < rk[y]; });
          for (int r : order) tile_rows.push_back(r);
        }
        for (size_t r = first_row; r < tile_rows.size(); ++r)
          if ((r - first_row) % 4 == 3 || r + 1 == tile_rows.size()) tile_ptr.push_back((int)r + 1);
      }
    } else {
    {
      const std::pair<int, int> rr = pack_rows(4, 16);
      std::vector<int> order(rr.second - rr.first);
      std::iota(order.begin(), order.end(), rr.first);
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return rows[x].items > rows[y].items; });
      const int R = (int)order.size(), T = (R + 3) / 4;
      std::vector<int> slot((size_t)4 * T, -1);
      for (int r = 0; r < R; ++r) {                  // boustrophedon deal: heavy rows meet light rows
        const int pass = r / T, pos = r % T;
        slot[(size_t)4 * ((pass & 1) ? T - 1 - pos : pos) + pass] = order[r];
      }
      for (int t = 0; t < T; ++t) {
        for (int q = 0; q < 4; ++q) if (slot[(size_t)4 * t + q] >= 0) tile_rows.push_back(slot[(size_t)4 * t + q]);
        tile_ptr.push_back((int)tile_rows.size());
      }
    }
    {
      const std::pair<int, int> rr = pack_rows(1, 3);
      for (int r = rr.first; r < rr.second; ++r) {
        tile_rows.push_back(r);
        if ((r - rr.first) % 4 == 3 || r + 1 == rr.second) tile_ptr.push_back((int)tile_rows.size());
      }
    }
    }
  }
  P.line_order.clear();
  P.line_order.reserve(L);
  if (P.big) { for (int l = 0; l < L; ++l) P.line_order.push_back(l); }
  else for (int r : tile_rows) for (int l = rows[r].head; l >= 0; l = next[l]) P.line_order.push_back(l);
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
    std::vector<int> cam_key((size_t)C);           // free cameras first (ascending free index), then the others by id
    for (int c = 0; c < C; ++c) cam_key[(size_t)c] = P.cam_cf[c] >= 0 ? P.cam_cf[c] : P.Cf + c;
    auto key = [&](int i) { return cam_key[(size_t)w->camera_index[i]]; };
    for (int s = 0; s < L; ++s) {                  // stable insertion sort of the (at most 64) observations of a line
      int* o = P.ob_orig.data() + P.line_ptr[s];
      const int k = P.line_ptr[s + 1] - P.line_ptr[s];
      for (int a = 1; a < k; ++a) {
        const int v = o[a], kv = key(v);
        int b = a;
        while (b > 0 && key(o[b - 1]) > kv) { o[b] = o[b - 1]; --b; }
        o[b] = v;
      }
    }
  }
  P.ob_cam.resize(M);
  if (!ob_dest) P.ob.resize((size_t)8 * M);
  P.nkept = 0;
  {
    double* pl[4] = { P.ob.data(), P.ob.data() + 2 * (size_t)M, P.ob.data() + 4 * (size_t)M, P.ob.data() + 6 * (size_t)M };
    if (ob_dest) for (int q = 0; q < 4; ++q) pl[q] = ob_dest->plane[q];
    uint64_t nonfinite = 0;
    for (int o = 0; o < M; ++o) {
      const int i = P.ob_orig[o];
      // (a stream of windows reads every observation from memory exactly once, here, in the order of the sorted lines: ask for the lines a
      // few observations ahead - 16 threads packing cold windows were waiting on these loads, tools/pack_bench2.cpp)
      if (o + 16 < M) { const double* nx = w->observations + 8 * (size_t)P.ob_orig[o + 16]; __builtin_prefetch(nx); __builtin_prefetch(nx + 7); }
      P.ob_cam[o] = w->camera_index[i];
      const double* src = w->observations + 8 * (size_t)i;
      for (int q = 0; q < 4; ++q) { pl[q][2 * (size_t)o] = src[2 * q]; pl[q][2 * (size_t)o + 1] = src[2 * q + 1]; }
      for (int q = 0; q < 8; ++q) {               // exponent field all ones <=> NaN / Inf (branch-free, as all_finite)
        uint64_t x;
        std::memcpy(&x, src + q, 8);
        nonfinite |= ((x & 0x7ff0000000000000ull) + 0x0010000000000000ull) & 0x8000000000000000ull;
      }
      if (!(cam_const[w->camera_index[i]] && line_const[w->line_index[i]])) ++P.nkept;
    }
    if (nonfinite) return SLSLAM_ERR_INVALID_ARGUMENT;
  }

  // tiles, their lane maps and their off-diagonal camera-pair work items
  P.line_desc.assign(L, 0u);
  if (!P.big) {
    int s = 0;
    P.tiles.reserve(tile_ptr.size());
    // (the grouped matrix-core sweep has no pair phase: a window packed for it carries no work items)
    size_t items_total = 0;
    for (int l = 0; l < L && !grouping; ++l) items_total += (size_t)items_of(l);
    P.items.resize(2 * items_total);
    uint8_t* item_w = P.items.data();                         // the items are written through a cursor (sized exactly above)
    P.lane_map.reserve(64 * tile_ptr.size());
    P.line_desc.assign(L, 0u);
    uint16_t map[64];
    for (size_t ti = 0; ti + 1 < tile_ptr.size(); ++ti) {
      Tile t;
      t.line_begin = s; t.item_off = (int)((item_w - P.items.data()) / 2);
      for (int q = 0; q < 64; ++q) map[q] = (uint16_t)0x00FF;
      unsigned char row_cam_seen[4][kMaxFreeCams] = {};
      int lane = 0, nl = 0, min_lanes = 64, max_run = 1, multi = 0;
      for (int ri = tile_ptr[ti]; ri < tile_ptr[ti + 1]; ++ri) {
        lane = (lane + 15) & ~15;                            // every row entry starts a row
        for (int l = rows[tile_rows[ri]].head; l >= 0; l = next[l], ++s, ++nl) {
          const int k = P.line_ptr[s + 1] - P.line_ptr[s], run = std::max(k, 1);
          for (int j = 0; j < run; ++j) map[lane + j] = (uint16_t)(nl | (j << 8));
          // skew flag (bit 15): every second lane of this 16-lane row that holds an observation of the same free camera adds its
          // camera-record entries one step late (lba_kernels.h, diagonal block) - no two of them meet on one LDS address
          for (int j = 0; j < k && j < 64; ++j) {
            const int cf = P.cam_cf[P.ob_cam[P.line_ptr[s] + j]];
            if (cf < 0) continue;
            const int row = (lane + j) >> 4;
            if (row_cam_seen[row][cf]++ & 1) map[lane + j] |= (uint16_t)0x8000;
          }
          min_lanes = std::min(min_lanes, run);
          max_run = std::max(max_run, std::min(run, 16));
          if (run > 16) multi = 1;
          if (!(P.line_flags[s] & 1) && !grouping) {
            int kf = 0;                                      // free-camera observations come first
            while (kf < k && P.cam_cf[P.ob_cam[P.line_ptr[s] + kf]] >= 0) ++kf;
            for (int i = 0; i < kf; ++i)
              for (int j = i + 1; j < kf; ++j) { *item_w++ = (uint8_t)(lane + i); *item_w++ = (uint8_t)(lane + j); }
          }
          {
            // what the matrix-core elimination needs to find the line's F blocks and to know which accumulator tiles the line
            // updates: free-camera mask (bits 0-9; the observations of these cameras are the first lanes of the run, ascending
            // free index) | first lane of the run << 10 | tile t = (I, J), J <= I, touched (both 16-row blocks hold a row of
            // one of the line's cameras; camera cf owns rows 6 cf .. 6 cf + 5) << (16 + t)
            const uint32_t m = (P.line_flags[s] & 1) ? 0u : (fmask[P.line_order[s]] & 0x3ffu);
            bool blk[4] = { false, false, false, false };
            for (int cf = 0; cf < 10; ++cf)
              if ((m >> cf) & 1u) { blk[(6 * cf) / 16] = true; blk[(6 * cf + 5) / 16] = true; }
            uint32_t tiles_touched = 0;
            for (int I = 0, t = 0; I < 4; ++I)
              for (int J = 0; J <= I; ++J, ++t) if (blk[I] && blk[J]) tiles_touched |= 1u << t;
            P.line_desc[s] = m | ((uint32_t)lane << 10) | (tiles_touched << 16);
            if (grouping) {
              // grouped sweep: the rows of the line's cameras counted from its first one, a = lowest bit of m: 6 (hi - a + 1) rows
              // = that many 16-row blocks
              // ... | bit 23: the range has holes (some camera between the first and the last does not see the line: the
              // observation of camera a + i is then NOT lane first + i) | cameras in the range << 24
              uint32_t a = 0, nblk = 0, wdt = 0, holes = 0;
              if (m) {
                a = (uint32_t)__builtin_ctz(m);
                wdt = (uint32_t)(31 - __builtin_clz(m)) - a + 1u;
                nblk = (6u * wdt + 15u) / 16u;
                holes = (uint32_t)__builtin_popcount(m) != wdt ? 1u : 0u;
              }
              P.line_desc[s] = gp_desc(m, (uint32_t)lane, a, nblk, holes, wdt);
            }
          }
          lane += run;
        }
      }
      if (grouping) {
        // the grouped sweep walks a tile's descriptors, not its lanes (a descriptor names its first lane): group after group,
        // inside a group by the number of blocks, lines without elimination work last (the walk ends at the first of them)
        auto key = [](uint32_t d) { return gp_mask(d) ? (int)(gp_group(d) * 8u + gp_blocks(d)) : 1 << 20; };
        std::stable_sort(P.line_desc.begin() + t.line_begin, P.line_desc.begin() + t.line_begin + nl,
                         [&](uint32_t x, uint32_t y) { return key(x) < key(y); });
      }
      const int rounds_log2 = min_lanes >= 4 ? 0 : min_lanes >= 2 ? 1 : 2;
      t.nlines = (int16_t)nl;
      t.flags = (int16_t)(multi | (rounds_log2 << 1) | (max_run << 3));
      t.nitems = (int)((item_w - P.items.data()) / 2) - t.item_off;
      P.tiles.push_back(t);
      P.lane_map.insert(P.lane_map.end(), map, map + 64);
    }
    P.items.resize((size_t)(item_w - P.items.data()));
  }
  return SLSLAM_OK;
}

int repack_window(const PackedWindow& P, int grouping, PackedWindow* out) {
  if (!out) return SLSLAM_ERR_INVALID_ARGUMENT;
  const size_t M = (size_t)P.M;
  std::vector<int> cam(M), line(M), fixed(2 * M);
  std::vector<double> obs(8 * M);
  for (int s = 0; s < P.L; ++s)
    for (int o = P.line_ptr[s]; o < P.line_ptr[s + 1]; ++o) {
      const size_t i = (size_t)P.ob_orig[o];
      cam[i] = P.ob_cam[o]; line[i] = P.line_order[s];
      fixed[2 * i] = P.cam_cf[P.ob_cam[o]] < 0 ? 1 : 0;       // (a camera with observations and no free index is a constant one)
      fixed[2 * i + 1] = P.line_flags[s] & 1;
      for (int q = 0; q < 4; ++q) {
        obs[8 * i + 2 * q] = P.ob[((size_t)q * M + (size_t)o) * 2];
        obs[8 * i + 2 * q + 1] = P.ob[((size_t)q * M + (size_t)o) * 2 + 1];
      }
    }
  slslam_lba_window w{};
  w.num_cameras = P.C; w.num_lines = P.L; w.num_observations = P.M;
  w.camera_index = cam.data(); w.line_index = line.data(); w.fixed_index = fixed.data();
  w.observations = obs.data(); w.parameters = const_cast<double*>(P.params0.data());     // (read only by the packer)
  return pack_window(&w, out, grouping);
}

std::vector<int> chunk_boundaries(int ntiles, int tiles_per_chunk) {
  std::vector<int> b;
  if (tiles_per_chunk < 1) tiles_per_chunk = 1;
  const int nchunks = ntiles > 0 ? (ntiles + tiles_per_chunk - 1) / tiles_per_chunk : 0;
  b.push_back(0);
  for (int c = 1; c <= nchunks; ++c) b.push_back((int)(((long long)ntiles * c) / nchunks));
  return b;
}

std::vector<int> chunk_boundaries_graded(int ntiles, int nchunks, const int* weights) {
  std::vector<int> b(1, 0);
  if (ntiles <= 0 || nchunks <= 0) return b;
  long long total = 0;
  for (int c = 0; c < nchunks; ++c) total += weights[c] > 0 ? weights[c] : 1;
  if (ntiles < 2 * nchunks) return chunk_boundaries(ntiles, (ntiles + nchunks - 1) / nchunks);
  // boundary c = round(ntiles * (w_0 + ... + w_{c-1}) / total), every chunk at least one tile
  long long acc = 0;
  for (int c = 0; c < nchunks; ++c) {
    acc += weights[c] > 0 ? weights[c] : 1;
    int e = (int)((ntiles * acc + total / 2) / total);
    const int lo = b.back() + 1, hi = ntiles - (nchunks - 1 - c);
    if (e < lo) e = lo;
    if (e > hi) e = hi;
    b.push_back(c == nchunks - 1 ? ntiles : e);
  }
  return b;
}

}  // namespace slslam
