// tests/host_cxx/dist_demo.cpp — the C-level fan-out (include/slslam_dist.h) as a C++ host would drive it: one process per GPU.
//   dist_demo <rank> <world> <id_file> <windows.bin> <out.bin> [device] [inject]
// inject = 1: before the real solve the rank drives the two error paths of slslam_dist_solve - a shard reported as failed
// (slslam_dist_debug_fail_next_shard) and a shard with a malformed window - and checks that both calls RETURN (the all-reduce was entered,
// the all-gather was not), report an error and leave the communicator usable for the real solve that follows.
// rank 0 writes the communicator id to <id_file>, the other ranks wait for it (any launcher-side channel would do); every rank reads the
// job's window list, solves its contiguous shard (slslam_dist_shard_range) in place and takes part in the one all-reduce + one
// all-gather; rank 0 writes [sums(3) | slot | counts(world) | gathered(world * slot)] to <out.bin>.
// What is fanned out is the reference's per-window call LBAProblem::build + ceres::Solve (src/slam.cpp:924-944); the sums are the ones
// its caller accumulates at :949-952.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "slslam_dist.h"

struct Win { std::vector<int> cam, line, fixed; std::vector<double> obs, par; slslam_lba_window c; };

static void rd(FILE* f, void* p, size_t n) { if (n && fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }

int main(int argc, char** argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: dist_demo rank world id_file windows.bin out.bin [device]\n"); return 2; }
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
  const int device = argc > 6 ? std::atoi(argv[6]) : rank;
  unsigned char id[SLSLAM_DIST_ID_BYTES];
  if (rank == 0) {
    int rc = slslam_dist_unique_id(id);
    if (rc != SLSLAM_OK) { std::fprintf(stderr, "unique id: %s\n", slslam_status_string(rc)); return rc; }
    std::string tmp = std::string(argv[3]) + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) return 2;
    std::fclose(f);
    std::rename(tmp.c_str(), argv[3]);                      // atomically: a waiting rank never reads half an id
  } else {
    FILE* f = nullptr;
    for (int tries = 0; tries < 600 && !(f = std::fopen(argv[3], "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (!f) { std::fprintf(stderr, "rank %d: no communicator id\n", rank); return 2; }
    rd(f, id, sizeof(id));
    std::fclose(f);
  }
  // the job's windows (every rank reads the list; only its shard is touched)
  FILE* f = std::fopen(argv[4], "rb");
  if (!f) return 2;
  int K = 0;
  rd(f, &K, sizeof(K));
  std::vector<Win> wins((size_t)K);
  long long max_params = 0;
  for (Win& w : wins) {
    int h[3];
    rd(f, h, sizeof(h));
    const size_t M = (size_t)h[2], np = (size_t)6 * h[0] + (size_t)4 * h[1];
    w.cam.resize(M); w.line.resize(M); w.fixed.resize(2 * M); w.obs.resize(8 * M); w.par.resize(np);
    rd(f, w.cam.data(), 4 * M); rd(f, w.line.data(), 4 * M); rd(f, w.fixed.data(), 8 * M); rd(f, w.obs.data(), 64 * M); rd(f, w.par.data(), 8 * np);
    w.c.num_cameras = h[0]; w.c.num_lines = h[1]; w.c.num_observations = h[2];
    w.c.camera_index = w.cam.data(); w.c.line_index = w.line.data(); w.c.fixed_index = w.fixed.data(); w.c.observations = w.obs.data(); w.c.parameters = w.par.data();
  }
  std::fclose(f);
  slslam_dist* d = nullptr;
  int rc = slslam_dist_create(rank, world, device, id, &d);
  if (rc != SLSLAM_OK) { std::fprintf(stderr, "rank %d: create: %s\n", rank, slslam_status_string(rc)); return rc; }
  long long lo = 0, hi = 0, slot = 1;
  for (int r = 0; r < world; ++r) {                          // the bound every rank passes: the largest shard's parameter count
    long long a, b, c = 0;
    slslam_dist_shard_range(K, r, world, &a, &b);
    for (long long i = a; i < b; ++i) c += (long long)wins[(size_t)i].par.size();
    if (c > slot) slot = c;
    (void)max_params;
  }
  slslam_dist_shard_range(K, rank, world, &lo, &hi);
  std::vector<slslam_lba_window> mine;
  for (long long i = lo; i < hi; ++i) mine.push_back(wins[(size_t)i].c);
  slslam_solver_options opt;
  slslam_default_options(&opt);
  double sums[3] = { 0, 0, 0 };
  std::vector<double> gathered((size_t)slot * (size_t)world, 0.0);
  std::vector<long long> counts((size_t)world, 0);
  std::vector<std::vector<double>> keep;
  for (const slslam_lba_window& m : mine) keep.emplace_back(m.parameters, m.parameters + 6 * m.num_cameras + 4 * m.num_lines);
  auto restore = [&] { for (size_t i = 0; i < mine.size(); ++i) std::memcpy(mine[i].parameters, keep[i].data(), keep[i].size() * sizeof(double)); };
  const bool inject = argc > 7 && std::atoi(argv[7]) == 1 && !mine.empty();
  if (inject) {
    slslam_dist_debug_fail_next_shard(d);
    const int r1 = slslam_dist_solve(d, mine.data(), (int)mine.size(), &opt, sums, gathered.data(), slot, counts.data());
    restore();
    std::vector<int> bad_cam(mine[0].camera_index, mine[0].camera_index + mine[0].num_observations);
    if (!bad_cam.empty()) bad_cam[0] = mine[0].num_cameras + 3;                 // out of range: the build refuses the window
    std::vector<slslam_lba_window> bad = mine;
    bad[0].camera_index = bad_cam.data();
    const int r2 = slslam_dist_solve(d, bad.data(), (int)bad.size(), &opt, sums, gathered.data(), slot, counts.data());
    restore();
    if (r1 == SLSLAM_OK || r2 != SLSLAM_ERR_INVALID_ARGUMENT || sums[0] != 0.0) { std::fprintf(stderr, "rank %d: error paths: %d %d sums %.0f\n", rank, r1, r2, sums[0]); return 3; }
    std::printf("rank %d: error paths ok (failed shard -> %s, malformed window -> %s; collectives completed)\n", rank, slslam_status_string(r1), slslam_status_string(r2));
  }
  rc = slslam_dist_solve(d, mine.data(), (int)mine.size(), &opt, sums, gathered.data(), slot, counts.data());
  if (rc != SLSLAM_OK) std::fprintf(stderr, "rank %d: solve: %s\n", rank, slslam_status_string(rc));
  std::printf("rank %d of %d: windows [%lld, %lld), job sums: %.0f LM iterations, cost %.9e -> %.9e\n", rank, world, lo, hi, sums[0], sums[1], sums[2]);
  if (rank == 0 && rc == SLSLAM_OK) {
    FILE* o = std::fopen(argv[5], "wb");
    if (!o) return 2;
    const double s = (double)slot;
    fwrite(sums, 8, 3, o); fwrite(&s, 8, 1, o);
    for (long long c : counts) { const double v = (double)c; fwrite(&v, 8, 1, o); }
    fwrite(gathered.data(), 8, gathered.size(), o);
    std::fclose(o);
  }
  if (inject && rc == SLSLAM_OK) {
    // the STREAMED form (slslam_dist_stream_*): the same shard three times through a depth-2 stream, the arrays in page-locked memory so that the
    // second and third set are read by the GPU in place and built on the device; every set's all-reduced sums and solved parameters must be
    // those of the batch call above
    std::vector<std::vector<double>> solved;
    for (const slslam_lba_window& m : mine) solved.emplace_back(m.parameters, m.parameters + 6 * m.num_cameras + 4 * m.num_lines);
    size_t need = 4096;
    for (const slslam_lba_window& m : mine) need += 80 * (size_t)m.num_observations + 8 * (6 * (size_t)m.num_cameras + 4 * (size_t)m.num_lines) + 5 * 64;
    char* arena = nullptr;
    int prc = slslam_pinned_alloc(3 * need, (void**)&arena);
    if (prc != SLSLAM_OK) { std::fprintf(stderr, "rank %d: pinned alloc: %s\n", rank, slslam_status_string(prc)); return prc; }
    size_t off = 0;
    auto take = [&](const void* src, size_t bytes) { char* p = arena + off; std::memcpy(p, src, bytes); off += (bytes + 63) & ~(size_t)63; return (void*)p; };
    std::vector<std::vector<slslam_lba_window>> sets(3);
    for (int k = 0; k < 3; ++k)
      for (size_t i = 0; i < mine.size(); ++i) {
        const slslam_lba_window& m = mine[(i + (size_t)k) % mine.size()];          // another order in every set
        const size_t M = (size_t)m.num_observations, np = 6 * (size_t)m.num_cameras + 4 * (size_t)m.num_lines;
        slslam_lba_window c = m;
        c.camera_index = (const int*)take(m.camera_index, 4 * M); c.line_index = (const int*)take(m.line_index, 4 * M);
        c.fixed_index = (const int*)take(m.fixed_index, 8 * M); c.observations = (const double*)take(m.observations, 64 * M);
        c.parameters = (double*)take(keep[(i + (size_t)k) % mine.size()].data(), 8 * np);
        sets[(size_t)k].push_back(c);
      }
    slslam_dist_stream* ds = nullptr;
    int src = slslam_dist_stream_create(d, &opt, 2, &ds);
    int tk[3] = { -1, -1, -1 };
    double ssum[3][3];
    for (int k = 0; k < 3 && src == SLSLAM_OK; ++k) {
      if (k >= 2) src = slslam_dist_stream_collect(ds, tk[k - 2], ssum[k - 2]);
      if (src == SLSLAM_OK) src = slslam_dist_stream_submit(ds, sets[(size_t)k].data(), (int)sets[(size_t)k].size(), &tk[k]);
    }
    for (int k = 1; k < 3 && src == SLSLAM_OK; ++k) src = slslam_dist_stream_collect(ds, tk[k], ssum[k]);
    if (src != SLSLAM_OK) { std::fprintf(stderr, "rank %d: dist stream: %s\n", rank, slslam_status_string(src)); return src; }
    long long dev_builds = 0, zero_copy = 0, fallbacks = 0;
    bool same = true;
    for (int k = 0; k < 3; ++k) {
      same = same && ssum[k][0] == sums[0] && std::fabs(ssum[k][1] - sums[1]) <= 1e-12 * sums[1] && std::fabs(ssum[k][2] - sums[2]) <= 1e-12 * sums[2];
      for (size_t i = 0; i < mine.size(); ++i) {
        const std::vector<double>& wnt = solved[(i + (size_t)k) % mine.size()];
        same = same && std::memcmp(sets[(size_t)k][i].parameters, wnt.data(), wnt.size() * sizeof(double)) == 0;
      }
    }
    slslam_dist_stream_destroy(ds);
    (void)slslam_pinned_free(arena);
    (void)dev_builds; (void)zero_copy; (void)fallbacks;
    if (!same) { std::fprintf(stderr, "rank %d: streamed sets differ from the batch call\n", rank); return 4; }
    std::printf("rank %d: streamed fan-out ok (3 sets through slslam_dist_stream, sums and parameters equal to the batch call)\n", rank);
  }
  slslam_dist_destroy(d);
  return rc;
}
