// tools/ceres_harness.cpp — OPTIONAL true-Ceres leg of the CPU baseline.
//
// The reference solves its line bundle adjustment with Ceres Solver (README:8 pins 1.7.0; src/lba_problem.cpp:54-132 wires the problem).
// Neither Ceres nor Eigen exists in the build image or on the GPU boxes seen so far, so this file has never been compiled against a real
// installation; bench.py builds and runs it ONLY where its probe finds one (bench.py::ceres_probe) and keeps the in-repo oracle
// (cpu_baseline.kind = "port") whenever the build or the run fails.  It is this repository's own code on the public Ceres API:
//   * the residual is restated from the mathematics of the reference functor (src/lba_problem.h:46-118: a line in orthonormal
//     parameters (a, b, g, t) -> closest point cp = -cot(t) col2(R_l), direction col1(R_l), R_l = Rz(g) Ry(b) Rx(a); rotated and translated
//     into the keyframe; for both cameras of the stereo pair the signed distances of the two observed end points to the projected line),
//     written through the line's frame rather than term by term;
//   * the wiring follows the published semantics of src/lba_problem.cpp: one AutoDiffCostFunction<., 4, 6, 4> per observation, HuberLoss(1 / 406.05)
//     when the robust flag is on, a block constant when any of its observations flags it, SPARSE_NORMAL_CHOLESKY, one thread, silent.
// Input: the binary window format of tests/host_cxx/drop_in_demo.cpp (int32 header {C, L, M, max_iter, robust}, camera_index[M], line_index[M],
// fixed_index[2M], observations[8M], parameters[6C + 4L]).  Output: one JSON line per window on stdout; the solved parameters to <out>.
//   ceres_harness <in.bin> <out.bin> [repetitions]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ceres/ceres.h"
#include "ceres/rotation.h"
#include "ceres_harness_functor.h"

namespace {

bool read_exact(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: ceres_harness <in.bin> <out.bin> [repetitions]\n"); return 2; }
  const int reps = argc > 3 ? std::atoi(argv[3]) : 1;
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 2;
  int hdr[5];
  if (!read_exact(f, hdr, sizeof(hdr))) return 2;
  const int C = hdr[0], L = hdr[1], M = hdr[2], max_iter = hdr[3], robust = hdr[4];
  std::vector<int> cam(M), line(M), fixed(2 * (size_t)M);
  std::vector<double> obs(8 * (size_t)M), x0(6 * (size_t)C + 4 * (size_t)L);
  if (!read_exact(f, cam.data(), sizeof(int) * M) || !read_exact(f, line.data(), sizeof(int) * M) || !read_exact(f, fixed.data(), sizeof(int) * 2 * M) ||
      !read_exact(f, obs.data(), sizeof(double) * 8 * M) || !read_exact(f, x0.data(), sizeof(double) * x0.size())) return 2;
  std::fclose(f);

  std::vector<double> x;
  double seconds = 0.0, initial_cost = 0.0, final_cost = 0.0;
  long iterations = 0;
  int ok_steps = 0, bad_steps = 0, termination = -1;
  for (int rep = 0; rep < (reps > 0 ? reps : 1); ++rep) {
    x = x0;
    ceres::Problem problem;
    for (int i = 0; i < M; ++i) {
      ceres::CostFunction* cost = new ceres::AutoDiffCostFunction<StereoLineDistances, 4, 6, 4>(new StereoLineDistances(&obs[8 * (size_t)i], 0.12));
      ceres::LossFunction* loss = robust ? new ceres::HuberLoss(1.0 / 406.05) : NULL;
      double* cb = &x[6 * (size_t)cam[i]];
      double* lb = &x[6 * (size_t)C + 4 * (size_t)line[i]];
      problem.AddResidualBlock(cost, loss, cb, lb);
    }
    for (int i = 0; i < M; ++i) {
      if (fixed[2 * (size_t)i]) problem.SetParameterBlockConstant(&x[6 * (size_t)cam[i]]);
      if (fixed[2 * (size_t)i + 1]) problem.SetParameterBlockConstant(&x[6 * (size_t)C + 4 * (size_t)line[i]]);
    }
    ceres::Solver::Options options;
    options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
    options.max_num_iterations = max_iter;
    options.num_threads = 1;
    options.minimizer_progress_to_stdout = false;
    options.logging_type = ceres::SILENT;
    ceres::Solver::Summary summary;
    const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ceres::Solve(options, &problem, &summary);
    seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ok_steps = summary.num_successful_steps; bad_steps = summary.num_unsuccessful_steps;
    iterations += ok_steps + bad_steps;
    initial_cost = summary.initial_cost; final_cost = summary.final_cost;
    termination = (int)summary.termination_type;
  }
  std::printf("{\"cameras\": %d, \"lines\": %d, \"observations\": %d, \"repetitions\": %d, \"lm_iterations\": %ld, \"seconds\": %.6f, "
              "\"num_successful_steps\": %d, \"num_unsuccessful_steps\": %d, \"initial_cost\": %.17g, \"final_cost\": %.17g, \"termination_type\": %d}\n",
              C, L, M, reps, iterations, seconds, ok_steps, bad_steps, initial_cost, final_cost, termination);
  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 2;
  std::fwrite(x.data(), sizeof(double), x.size(), o);
  std::fclose(o);
  return 0;
}
