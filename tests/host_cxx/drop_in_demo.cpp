// tests/host_cxx/drop_in_demo.cpp — the reference's own call protocol against the MI355X back-end.
// The three blocks below are written the way SLAM::bundle_adjustment / motion_only_ba
// (reference src/slam.cpp:899-972, :618-674) and SLAM::pose_optimization (:1262-1301) call the
// optimisation layer: new[] the arrays, hand them to the problem object (which takes ownership),
// build, set_options, ceres::Solve, read `parameters` back before the object dies.
//   drop_in_demo lba <in.bin> <out.bin>      drop_in_demo po <in.bin> <out.bin>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lba_problem.h"
#include "po_problem.h"

static void rd(FILE* f, void* p, size_t n) { if (fread(p, 1, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); } }

static int run_lba(const char* in, const char* out) {
  FILE* f = std::fopen(in, "rb");
  if (!f) return 2;
  int hdr[5];
  rd(f, hdr, sizeof(hdr));
  const int num_cameras = hdr[0], num_lines = hdr[1], num_observations = hdr[2], max_num_iter = hdr[3];
  slslam::flag_robust = hdr[4] != 0;                       // FLAGS_robust
  const int num_parameters = 6 * num_cameras + 4 * num_lines;

  int* line_index = new int[num_observations];
  int* camera_index = new int[num_observations];
  int* fixed_index = new int[2 * num_observations];
  double* observations = new double[8 * num_observations];
  double* parameters = new double[num_parameters];
  rd(f, camera_index, sizeof(int) * num_observations);
  rd(f, line_index, sizeof(int) * num_observations);
  rd(f, fixed_index, sizeof(int) * 2 * num_observations);
  rd(f, observations, sizeof(double) * 8 * num_observations);
  rd(f, parameters, sizeof(double) * num_parameters);
  std::fclose(f);

  ceres::lba_param_t param;
  param.num_cameras = num_cameras;
  param.num_lines = num_lines;
  param.num_observations = num_observations;
  param.num_iterations = max_num_iter;
  param.num_parameters = num_parameters;
  param.mode = MODE_SPARSE_SCHUR;

  ceres::LBAProblem ba_problem(param);
  ba_problem.set_line_index(line_index);
  ba_problem.set_camera_index(camera_index);
  ba_problem.set_fixed_index(fixed_index);
  ba_problem.set_observations(observations);
  ba_problem.set_parameters(parameters);

  ceres::Problem problem;
  ba_problem.build(&problem);
  ceres::Solver::Options options;
  ba_problem.set_options(&options);
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);

  double tail[5] = { (double)summary.num_successful_steps, (double)summary.num_unsuccessful_steps,
                     (double)summary.initial_cost, (double)summary.final_cost, (double)summary.backend_status };
  FILE* o = std::fopen(out, "wb");
  std::fwrite(parameters, sizeof(double), num_parameters, o);
  std::fwrite(tail, sizeof(double), 5, o);
  std::fclose(o);
  std::printf("%s", summary.FullReport().c_str());
  return summary.backend_status;
}

static int run_po(const char* in, const char* out) {
  FILE* f = std::fopen(in, "rb");
  if (!f) return 2;
  int hdr[2];
  rd(f, hdr, sizeof(hdr));
  const int kfs_size = hdr[0], edge_size = hdr[1];
  int* pose_index_1 = new int[edge_size];
  int* pose_index_2 = new int[edge_size];
  double* constraints = new double[edge_size * 6];
  double* parameters = new double[kfs_size * 6];
  rd(f, pose_index_1, sizeof(int) * edge_size);
  rd(f, pose_index_2, sizeof(int) * edge_size);
  rd(f, constraints, sizeof(double) * 6 * edge_size);
  rd(f, parameters, sizeof(double) * 6 * kfs_size);
  std::fclose(f);

  ceres::POProblem po_problem(edge_size, 10);
  po_problem.set_pose_index_1(pose_index_1);
  po_problem.set_pose_index_2(pose_index_2);
  po_problem.set_constraints(constraints);
  po_problem.set_parameters(parameters);
  ceres::Problem problem;
  po_problem.build(&problem);
  ceres::Solver::Options options;
  po_problem.set_options(&options);
  ceres::Solver::Summary summary;
  ceres::Solve(options, &problem, &summary);

  double tail[5] = { (double)summary.num_successful_steps, (double)summary.num_unsuccessful_steps,
                     (double)summary.initial_cost, (double)summary.final_cost, (double)summary.backend_status };
  FILE* o = std::fopen(out, "wb");
  std::fwrite(parameters, sizeof(double), 6 * kfs_size, o);
  std::fwrite(tail, sizeof(double), 5, o);
  std::fclose(o);
  return summary.backend_status;
}

int main(int argc, char** argv) {
  if (argc != 4) { std::fprintf(stderr, "usage: %s lba|po in.bin out.bin\n", argv[0]); return 64; }
  if (!std::strcmp(argv[1], "lba")) return run_lba(argv[2], argv[3]);
  if (!std::strcmp(argv[1], "po")) return run_po(argv[2], argv[3]);
  return 64;
}
