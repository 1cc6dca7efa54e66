// slslam_amd/csrc/lba_pack.cpp — see lba_pack.h.
#include "lba_pack.h"
#include "lba_eliminate_grouped_maps.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

// host-side phase clock of pack_window (tools/pack_bench.cpp -DSLS_PACK_TIMING): where a window's pack time goes
#if defined(SLS_PACK_TIMING)
#include <chrono>
namespace slslam { double g_pack_phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0}; }
#define SLS_PACK_T0 auto pk_t_ = std::chrono::steady_clock::now()
#define SLS_PACK_MARK(i) do { const auto n_ = std::chrono::steady_clock::now(); slslam::g_pack_phase_ms[i] += std::chrono::duration<double, std::milli>(n_ - pk_t_).count(); pk_t_ = n_; } while (0)
#else
#define SLS_PACK_T0 do {} while (0)
#define SLS_PACK_MARK(i) do {} while (0)
#endif

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
  SLS_PACK_T0;
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
  // (at most 64 cameras - every window of the tiled sweeps: the cameras that see a line as a 64-bit mask, gathered in this pass, give the
  // free-camera masks and counts below without a second pass over the observations; `twice`: some camera sees some line more than once)
  const bool small_c = C <= 64;
  std::vector<uint64_t> cmask(small_c ? (size_t)L : 0, 0ull);
  uint64_t twice = 0;
  if (small_c) {
    for (int i = 0; i < M; ++i) {
      const int c = w->camera_index[i], l = w->line_index[i];
      if (c < 0 || c >= C || l < 0 || l >= L) return SLSLAM_ERR_INVALID_ARGUMENT;
      const uint64_t bit = 1ull << c;
      twice |= cmask[(size_t)l] & bit;
      cmask[(size_t)l] |= bit;
      cam_used[c] = 1; line_cnt[l]++;
      if (w->fixed_index[2 * i]) cam_const[c] = 1;
      if (w->fixed_index[2 * i + 1]) line_const[l] = 1;
    }
  } else {
    for (int i = 0; i < M; ++i) {
      const int c = w->camera_index[i], l = w->line_index[i];
      if (c < 0 || c >= C || l < 0 || l >= L) return SLSLAM_ERR_INVALID_ARGUMENT;
      cam_used[c] = 1; line_cnt[l]++;
      if (w->fixed_index[2 * i]) cam_const[c] = 1;
      if (w->fixed_index[2 * i + 1]) line_const[l] = 1;
    }
  }
  SLS_PACK_MARK(0);
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
  if (!P.big && small_c && !twice) {
    // no camera sees a line twice: a line's free-camera mask is its camera mask with every free camera's bit moved to its free index, its
    // free observations are as many as the mask has bits
    // (byte by byte through a table: tab[b][v] = free-index bits of the cameras 8 b + {bits of v})
    const int nbytes = (C + 7) / 8;
    std::vector<unsigned> tab((size_t)nbytes * 256, 0u);
    for (int b = 0; b < nbytes; ++b)
      for (int v = 1; v < 256; ++v) {
        const int c = 8 * b + __builtin_ctz((unsigned)v), cf = c < C ? P.cam_cf[c] : -1;
        tab[(size_t)b * 256 + v] = tab[(size_t)b * 256 + (v & (v - 1))] | (cf >= 0 ? 1u << cf : 0u);
      }
    for (int l = 0; l < L; ++l) {
      unsigned fm = 0u;
      const uint64_t m = cmask[(size_t)l];
      for (int b = 0; b < nbytes; ++b) fm |= tab[(size_t)b * 256 + ((m >> (8 * b)) & 255u)];
      fmask[l] = fm; kfree[l] = __builtin_popcount(fm);
    }
  } else
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
  std::vector<int> tile_rows;                      // row ids, tile after tile
  std::vector<int> tile_ptr(1, 0);
  if (!P.big) {
    // open rows by the lanes they have left, with the free cameras of their lines (beside the id: the search below reads nothing else).
    // and_mask[room]: a subset of the cameras EVERY row of that list holds (the AND of the masks pushed since the list was last empty -
    // rows leave, so the true AND can only have more bits): a line that shares one of them shares a camera with every row of the list,
    // and the camera-disjoint pass skips the list without reading it.  (The lines of one group all see the group's first camera: for
    // the grouped packing that pass found nothing in list after list, 2 x 16 x 32 mask tests per line.)
    struct OpenRow { int row; unsigned mask; };
    struct OpenLists {
      std::vector<OpenRow> by_room[17];
      unsigned and_mask[17];
      void clear() { for (int q = 0; q <= 16; ++q) { by_room[q].clear(); and_mask[q] = ~0u; } }
      void push(int room, OpenRow r) { std::vector<OpenRow>& v = by_room[room]; and_mask[room] = v.empty() ? r.mask : (and_mask[room] & r.mask); v.push_back(r); }
    } open;
    // best fit over lines of decreasing length (the caller hands them over in that order, original order inside a length): a line goes to
    // the fullest open row that still holds it, else it opens a row
    auto pack_rows = [&](const int* lb, const int* le) {
      for (; lb != le; ++lb) {
        const int l = *lb, len = lanes_of(l);
        // the fullest open row that holds the line - and, among the rows of that fill, preferably one none of whose
        // lines shares a free camera with it: the lanes of one 16-lane row that add to the same camera record
        // serialise in the LDS (tools/micro/lds_atomic_bench.hip)
        int r = -1;
        const unsigned fm = fmask[l];
        for (int pass = 0; pass < 2 && r < 0; ++pass)                   // pass 0: rows without a common free camera only
          for (int room = len; room <= 16 && r < 0; ++room) {
            std::vector<OpenRow>& cand = open.by_room[room];
            if (cand.empty() || (pass == 0 && (open.and_mask[room] & fm))) continue;
            for (size_t c = cand.size(); c-- > 0 && cand.size() - c <= 32;)
              if (pass == 1 || !(cand[c].mask & fm)) { r = cand[c].row; cand.erase(cand.begin() + c); break; }
          }
        if (r < 0) { r = (int)rows.size(); rows.push_back(Row{0, 0, -1, -1, 0u}); }
        append(r, l);
        if (rows[r].used < 16) open.push(16 - rows[r].used, OpenRow{ r, rows[r].mask });
      }
    };
    std::vector<int> big;
    for (int l = 0; l < L; ++l) if (line_cnt[l] > 16) big.push_back(l);
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
    // The lines of at most 16 lanes in the order the rows are built from them - one counting sort (stable: original order inside a
    // cell): class (0: at least 4 lanes; 1: fewer, which need more than one sin/cos round per lane in the back-substitution and are kept in
    // tiles of their own), then - grouped packing - the group bucket (keys 2 a + {0, 1} with a < 20, or the last one: no elimination work),
    // then the length, longest first.
    enum { kBuckets = 42 };
    const int nb = grouping ? kBuckets : 1;
    std::vector<int> key_of;
    if (grouping) { key_of.resize((size_t)L); for (int l = 0; l < L; ++l) key_of[(size_t)l] = group_key(l); }
    auto cell_of = [&](int l) {
      const int len = lanes_of(l), cls = len < 4 ? 1 : 0;
      const int bq = grouping ? (key_of[(size_t)l] >= 1000 ? kBuckets - 1 : key_of[(size_t)l]) : 0;
      return (cls * nb + bq) * 16 + (16 - len);
    };
    std::vector<int> cell_ptr((size_t)2 * nb * 16 + 1, 0), seq;
    for (int l = 0; l < L; ++l) if (line_cnt[l] <= 16) ++cell_ptr[(size_t)cell_of(l) + 1];
    for (size_t q = 1; q < cell_ptr.size(); ++q) cell_ptr[q] += cell_ptr[q - 1];
    seq.resize((size_t)cell_ptr.back());
    {
      std::vector<int> cur(cell_ptr.begin(), cell_ptr.end() - 1);
      for (int l = 0; l < L; ++l) if (line_cnt[l] <= 16) seq[(size_t)cur[(size_t)cell_of(l)]++] = l;
    }
    auto bucket_range = [&](int cls, int bq) { return std::make_pair(seq.data() + cell_ptr[(size_t)(cls * nb + bq) * 16], seq.data() + cell_ptr[(size_t)(cls * nb + bq + 1) * 16]); };
    if (grouping) {
      // the two length classes of the default packing are kept, each of them group after group, the rows filling the tiles in that order
      for (int cls = 0; cls < 2; ++cls) {
        const size_t first_row = tile_rows.size();
        const int row0 = (int)rows.size();
        open.clear();                                  // (a group's lines may finish the open rows of the group before it: the lists live through the class)
        int prev_first = row0;                         // first row the previous group opened
        for (int bq = 0; bq < kBuckets; ++bq) {        // ascending keys
          const std::pair<const int*, const int*> rg = bucket_range(cls, bq);
          if (rg.first == rg.second) continue;
          // (only the rows the previous group left open: an older row would put this group's line in the middle of another
          // group's tiles, and the sweep adds its accumulators to memory whenever the group changes)
          for (int q = 0; q <= 16; ++q) {
            std::vector<OpenRow>& v = open.by_room[q];
            v.erase(std::remove_if(v.begin(), v.end(), [&](const OpenRow& r) { return r.row < prev_first; }), v.end());
          }
          prev_first = (int)rows.size();
          pack_rows(rg.first, rg.second);
        }
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
      const int r_first = (int)rows.size();
      open.clear();
      const std::pair<const int*, const int*> rg = bucket_range(0, 0);
      pack_rows(rg.first, rg.second);
      const std::pair<int, int> rr(r_first, (int)rows.size());
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
      const int r_first = (int)rows.size();
      open.clear();
      const std::pair<const int*, const int*> rg = bucket_range(1, 0);
      pack_rows(rg.first, rg.second);
      const std::pair<int, int> rr(r_first, (int)rows.size());
      for (int r = rr.first; r < rr.second; ++r) {
        tile_rows.push_back(r);
        if ((r - rr.first) % 4 == 3 || r + 1 == rr.second) tile_ptr.push_back((int)tile_rows.size());
      }
    }
    }
  }
  SLS_PACK_MARK(1);
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

  SLS_PACK_MARK(2);
  // observations grouped by line; inside a line: free cameras first (ascending free index)
  P.ob_orig.assign(M, 0);
  {
    // one 64-bit key per observation, camera key << 32 | original index, dropped into its line's range in the order of the caller's
    // arrays (ascending index): sorting the keys of a line IS the stable sort of its observations by camera - and a line whose
    // observations arrive in camera order (the reference's packer emits them so, src/slam.cpp:848-882) costs one comparison each
    std::vector<int> fill(P.line_ptr.begin(), P.line_ptr.end() - 1);
    std::vector<int> cam_key((size_t)C);           // free cameras first (ascending free index), then the others by id
    for (int c = 0; c < C; ++c) cam_key[(size_t)c] = P.cam_cf[c] >= 0 ? P.cam_cf[c] : P.Cf + c;
    std::vector<uint64_t> okey((size_t)M);
    for (int i = 0; i < M; ++i) {
      const int s = line_pos[w->line_index[i]];
      okey[(size_t)fill[(size_t)s]++] = (uint64_t)(uint32_t)cam_key[(size_t)w->camera_index[i]] << 32 | (uint32_t)i;
    }
    for (int s = 0; s < L; ++s) {                  // insertion sort of the (at most 64, or - oversize windows - any number of) observations of a line
      uint64_t* o = okey.data() + P.line_ptr[s];
      const int k = P.line_ptr[s + 1] - P.line_ptr[s];
      for (int a = 1; a < k; ++a) {
        const uint64_t v = o[a];
        int b = a;
        while (b > 0 && o[b - 1] > v) { o[b] = o[b - 1]; --b; }
        o[b] = v;
      }
    }
    for (int o = 0; o < M; ++o) P.ob_orig[(size_t)o] = (int)(uint32_t)okey[(size_t)o];
  }
  SLS_PACK_MARK(3);
  P.ob_cam.resize(M);
  if (!ob_dest) P.ob.resize((size_t)8 * M);
  P.nkept = 0;
  {
    double* pl[4] = { P.ob.data(), P.ob.data() + 2 * (size_t)M, P.ob.data() + 4 * (size_t)M, P.ob.data() + 6 * (size_t)M };
    if (ob_dest) for (int q = 0; q < 4; ++q) pl[q] = ob_dest->plane[q];
    // a refill packs straight into the pinned host image the copy engine reads next: nobody on the host reads these planes again, so they are
    // written past the caches (when the four planes are 16-byte aligned, which the image's are)
    bool stream_out = ob_dest != nullptr;
    for (int q = 0; q < 4; ++q) if (reinterpret_cast<uintptr_t>(pl[q]) & 15u) stream_out = false;
    static const bool no_stream = std::getenv("SLSLAM_PACK_NO_STREAMING_STORES") != nullptr;      // (measurement switch)
    if (no_stream) stream_out = false;
    uint64_t nonfinite = 0;
    if (ob_dest && ob_dest->raw) {
      // the device gathers (refill): here only the linear copy of the caller's array, tested for NaN / Inf on the way, and the sorted camera ids
      const double* src = w->observations;
      double* dst = ob_dest->raw;
      const size_t n8 = (size_t)8 * (size_t)M;
#if defined(__SSE2__)
      if (!(reinterpret_cast<uintptr_t>(dst) & 15u) && !no_stream) {
        // (non-temporal 16-byte stores, the exponent test of all_finite on the same registers)
        const __m128i expo = _mm_set1_epi64x((long long)0x7ff0000000000000ull), one = _mm_set1_epi64x((long long)0x0010000000000000ull);
        __m128i acc = _mm_setzero_si128();
        for (size_t q = 0; q < n8; q += 2) {
          const __m128d v = _mm_loadu_pd(src + q);
          _mm_stream_pd(dst + q, v);
          acc = _mm_or_si128(acc, _mm_add_epi64(_mm_and_si128(_mm_castpd_si128(v), expo), one));
        }
        _mm_sfence();
        uint64_t lanes[2];
        _mm_storeu_si128(reinterpret_cast<__m128i*>(lanes), acc);
        nonfinite |= (lanes[0] | lanes[1]) & 0x8000000000000000ull;
      } else
#endif
      {
        std::memcpy(dst, src, n8 * sizeof(double));
        for (size_t q = 0; q < n8; ++q) {
          uint64_t x;
          std::memcpy(&x, src + q, 8);
          nonfinite |= ((x & 0x7ff0000000000000ull) + 0x0010000000000000ull) & 0x8000000000000000ull;
        }
      }
      for (int o = 0; o < M; ++o) {
        const int i = P.ob_orig[o];
        P.ob_cam[o] = w->camera_index[i];
        if (!(cam_const[w->camera_index[i]] && line_const[w->line_index[i]])) ++P.nkept;
      }
    } else
    for (int o = 0; o < M; ++o) {
      const int i = P.ob_orig[o];
      // (a stream of windows reads every observation from memory exactly once, here, in the order of the sorted lines: ask for the lines a
      // few observations ahead - 16 threads packing cold windows were waiting on these loads, tools/pack_bench2.cpp)
      if (o + 16 < M) { const double* nx = w->observations + 8 * (size_t)P.ob_orig[o + 16]; __builtin_prefetch(nx); __builtin_prefetch(nx + 7); }
      P.ob_cam[o] = w->camera_index[i];
      const double* src = w->observations + 8 * (size_t)i;
#if defined(__SSE2__)
      if (stream_out) {                             // (16-byte non-temporal stores: four write-combining streams, no line is read in order to be overwritten)
        for (int q = 0; q < 4; ++q) _mm_stream_pd(pl[q] + 2 * (size_t)o, _mm_loadu_pd(src + 2 * q));
      } else
#endif
      for (int q = 0; q < 4; ++q) { pl[q][2 * (size_t)o] = src[2 * q]; pl[q][2 * (size_t)o + 1] = src[2 * q + 1]; }
      for (int q = 0; q < 8; ++q) {               // exponent field all ones <=> NaN / Inf (branch-free, as all_finite)
        uint64_t x;
        std::memcpy(&x, src + q, 8);
        nonfinite |= ((x & 0x7ff0000000000000ull) + 0x0010000000000000ull) & 0x8000000000000000ull;
      }
      if (!(cam_const[w->camera_index[i]] && line_const[w->line_index[i]])) ++P.nkept;
    }
#if defined(__SSE2__)
    if (stream_out) _mm_sfence();
#endif
    if (nonfinite) return SLSLAM_ERR_INVALID_ARGUMENT;
  }

  SLS_PACK_MARK(4);
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
            if (!grouping) {
              bool blk[4] = { false, false, false, false };
              for (int cf = 0; cf < 10; ++cf)
                if ((m >> cf) & 1u) { blk[(6 * cf) / 16] = true; blk[(6 * cf + 5) / 16] = true; }
              uint32_t tiles_touched = 0;
              for (int I = 0, t = 0; I < 4; ++I)
                for (int J = 0; J <= I; ++J, ++t) if (blk[I] && blk[J]) tiles_touched |= 1u << t;
              P.line_desc[s] = m | ((uint32_t)lane << 10) | (tiles_touched << 16);
            }
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
        uint32_t* d = P.line_desc.data() + t.line_begin;       // (a stable insertion sort: at most 64 descriptors, mostly in order already)
        for (int a = 1; a < nl; ++a) {
          const uint32_t v = d[a];
          const int kv = key(v);
          int b = a;
          while (b > 0 && key(d[b - 1]) > kv) { d[b] = d[b - 1]; --b; }
          d[b] = v;
        }
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
  SLS_PACK_MARK(5);
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
