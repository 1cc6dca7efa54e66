// tests/host_cxx/stream_demo.cpp — a STREAM of windows from a C++ host through the C ABI alone (include/slslam_hip.h: slslam_lba_stream_*):
// what a maintainer of the reference would write around SLAM::bundle_adjustment's arrays (src/slam.cpp:899-921) to keep the GPU busy with the
// windows of many sequences - batches of windows submitted while earlier ones are being solved, results collected in ticket order, the
// solved parameters written back in place (src/slam.cpp:957-972 reads them from there).  No Python, no torch.
//   stream_demo <windows.bin> <out.bin> <windows per batch> <batches> <depth> <host threads>
// windows.bin: [count | per window: C, L, M | camera_index[M] | line_index[M] | fixed_index[2M] | observations[8M] | parameters[6C+4L]];
// batch k takes the windows k * per, ..., k * per + per - 1 (mod count), each batch working on its OWN copies of the arrays;
// out.bin: [LM steps total | per batch, per window: parameters].
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "slslam_hip.h"

struct Win { std::vector<int> cam, line, fixed; std::vector<double> obs, par; };

static void rd(FILE* f, void* p, size_t n) { if (n && fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: stream_demo windows.bin out.bin per_batch batches depth host_threads\n"); return 2; }
  const int per = std::atoi(argv[3]), nb = std::atoi(argv[4]), depth = std::atoi(argv[5]), threads = std::atoi(argv[6]);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int count = 0;
  rd(f, &count, 4);
  std::vector<Win> src((size_t)count);
  std::vector<int> C((size_t)count), L((size_t)count), M((size_t)count);
  for (int i = 0; i < count; ++i) {
    int h[3]; rd(f, h, 12);
    C[i] = h[0]; L[i] = h[1]; M[i] = h[2];
    Win& w = src[(size_t)i];
    w.cam.resize(h[2]); w.line.resize(h[2]); w.fixed.resize(2 * (size_t)h[2]); w.obs.resize(8 * (size_t)h[2]); w.par.resize(6 * (size_t)h[0] + 4 * (size_t)h[1]);
    rd(f, w.cam.data(), 4 * w.cam.size()); rd(f, w.line.data(), 4 * w.line.size()); rd(f, w.fixed.data(), 4 * w.fixed.size());
    rd(f, w.obs.data(), 8 * w.obs.size()); rd(f, w.par.data(), 8 * w.par.size());
  }
  std::fclose(f);
  // every batch owns its parameter arrays (they are solved in place while later batches are being packed)
  std::vector<std::vector<std::vector<double> > > params((size_t)nb);
  std::vector<std::vector<slslam_lba_window> > batch((size_t)nb);
  for (int k = 0; k < nb; ++k) {
    params[k].resize((size_t)per); batch[k].resize((size_t)per);
    for (int j = 0; j < per; ++j) {
      const int i = (k * per + j) % count;
      params[k][j] = src[(size_t)i].par;
      slslam_lba_window& w = batch[k][j];
      w.num_cameras = C[i]; w.num_lines = L[i]; w.num_observations = M[i];
      w.camera_index = src[(size_t)i].cam.data(); w.line_index = src[(size_t)i].line.data(); w.fixed_index = src[(size_t)i].fixed.data();
      w.observations = src[(size_t)i].obs.data(); w.parameters = params[k][j].data();
    }
  }
  slslam_solver_options opt;
  slslam_default_options(&opt);
  opt.host_threads = threads;
  opt.reproducible = 1;                                  // a window's bytes do not depend on the batch it travels in
  slslam_lba_stream* st = nullptr;
  int rc = slslam_lba_stream_create(-1, &opt, depth, &st);
  if (rc != SLSLAM_OK) { std::fprintf(stderr, "stream create: %s\n", slslam_status_string(rc)); return rc; }
  std::vector<int> ticket((size_t)nb, -1);
  std::vector<slslam_summary> sm((size_t)per);
  double steps = 0.0;
  for (int k = 0; k < nb && rc == SLSLAM_OK; ++k) {
    if (k >= depth) {                                     // the slot of batch k is that of batch k - depth: its results first
      rc = slslam_lba_stream_collect(st, ticket[k - depth], sm.data());
      for (int j = 0; j < per && rc == SLSLAM_OK; ++j) steps += sm[j].num_successful_steps + sm[j].num_unsuccessful_steps;     // reference src/slam.cpp:949-950
    }
    if (rc == SLSLAM_OK) rc = slslam_lba_stream_submit(st, batch[k].data(), per, &ticket[k]);
  }
  for (int k = nb > depth ? nb - depth : 0; k < nb && rc == SLSLAM_OK; ++k) {
    rc = slslam_lba_stream_collect(st, ticket[k], sm.data());
    for (int j = 0; j < per && rc == SLSLAM_OK; ++j) steps += sm[j].num_successful_steps + sm[j].num_unsuccessful_steps;
  }
  long long refills = 0, builds = 0, windows = 0, its = 0;
  int used_threads = 0;
  if (rc == SLSLAM_OK) rc = slslam_lba_stream_stats(st, nullptr, nullptr, nullptr, &refills, &builds, &windows, &its, &used_threads);
  slslam_lba_stream_destroy(st);
  if (rc != SLSLAM_OK) { std::fprintf(stderr, "stream: %s\n", slslam_status_string(rc)); return rc; }
  std::printf("stream_demo: %d batches x %d windows, %lld built, %lld refilled, %lld LM steps, %d host threads\n", nb, per, builds, refills, its, used_threads);
  if ((double)its != steps || windows != (long long)nb * per) { std::fprintf(stderr, "bookkeeping mismatch\n"); return 3; }
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 2;
  fwrite(&steps, 8, 1, o);
  for (int k = 0; k < nb; ++k) for (int j = 0; j < per; ++j) fwrite(params[k][j].data(), 8, params[k][j].size(), o);
  std::fclose(o);
  return 0;
}
