// slslam_amd/host/problems.cpp — LBAProblem / POProblem / ceres facade: thin C++ marshalling into
// the C ABI (include/slslam_hip.h).  Mirrors reference src/lba_problem.cpp and src/po_problem.cpp.
#include <cstdio>
#include <sstream>

#include "../../include/slslam_hip.h"
#include "lba_problem.h"
#include "po_problem.h"

namespace slslam {
bool flag_robust = true;   // FLAGS_robust default (reference src/main.cpp:27)
}

namespace ceres {

// ---------------------------------------------------------------- LBAProblem (src/lba_problem.cpp)
LBAProblem::LBAProblem(lba_param_t param)
    : mode_(param.mode), num_cameras_(param.num_cameras), num_lines_(param.num_lines),
      num_observations_(param.num_observations), num_parameters_(param.num_parameters),
      num_iterations_(param.num_iterations), num_threads(1), eta(1e-2),
      robustify(slslam::flag_robust), logging_type(false),
      line_index_(nullptr), camera_index_(nullptr), fixed_index_(nullptr), observations_(nullptr), parameters_(nullptr) {}

LBAProblem::~LBAProblem() {   // takes ownership like the reference (lba_problem.cpp:46-52)
  delete[] line_index_;
  delete[] camera_index_;
  delete[] fixed_index_;
  delete[] observations_;
  delete[] parameters_;
}

void LBAProblem::build(Problem* problem) {
  // reference: one AutoDiffCostFunction<LineReprojectionError,4,6,4> + HuberLoss per observation and
  // SetParameterBlockConstant per flagged block (lba_problem.cpp:62-92).  Here the array contract is
  // bound as a whole; the C ABI applies the same wiring rules on the device.
  problem->BindLBA(this);
}

void LBAProblem::set_options(Solver::Options* options) {
  // the reference's switch falls through (lba_problem.cpp:96-101): always SPARSE_NORMAL_CHOLESKY.
  // A direct solve of the normal equations and the Schur solve the kernels do give the same step.
  options->linear_solver_type = SPARSE_NORMAL_CHOLESKY;
  options->num_linear_solver_threads = num_threads;
  delete options->linear_solver_ordering;
  options->linear_solver_ordering = new ParameterBlockOrdering;
  for (int i = 0; i < num_lines_; ++i) options->linear_solver_ordering->AddElementToGroup(mutable_lines() + 4 * i, 0);
  for (int i = 0; i < num_cameras_; ++i) options->linear_solver_ordering->AddElementToGroup(mutable_cameras() + 6 * i, 0);
  options->max_num_iterations = num_iterations_;
  options->minimizer_progress_to_stdout = true;
  options->num_threads = num_threads;
  options->eta = eta;
  if (!logging_type) options->logging_type = SILENT;
}

// ---------------------------------------------------------------- POProblem (src/po_problem.cpp)
POProblem::POProblem(int a, int n)
    : num_iterations(n), num_threads(1), eta(1e-2), robustify(false), size_(a), num_poses_(-1),
      pose_index_1_(nullptr), pose_index_2_(nullptr), constraints_(nullptr), parameters_(nullptr) {}

POProblem::~POProblem() {
  delete[] pose_index_1_;
  delete[] pose_index_2_;
  delete[] constraints_;
  delete[] parameters_;
}

int POProblem::num_poses() const {
  if (num_poses_ >= 0) return num_poses_;
  int m = -1;
  for (int i = 0; i < size_; ++i) {
    if (pose_index_1_[i] > m) m = pose_index_1_[i];
    if (pose_index_2_[i] > m) m = pose_index_2_[i];
  }
  return m + 1;
}

void POProblem::build(Problem* problem) { problem->BindPO(this); }

void POProblem::set_options(Solver::Options* options) {
  options->linear_solver_type = SPARSE_NORMAL_CHOLESKY;
  options->num_linear_solver_threads = num_threads;
  options->max_num_iterations = num_iterations;
  options->minimizer_progress_to_stdout = true;
  options->num_threads = num_threads;
  options->eta = eta;
  options->logging_type = SILENT;
}

// ---------------------------------------------------------------- facade
Solver::Options::Options()
    : linear_solver_type(SPARSE_NORMAL_CHOLESKY), num_linear_solver_threads(1), linear_solver_ordering(nullptr),
      max_num_iterations(50), minimizer_progress_to_stdout(false), num_threads(1), eta(1e-1),
      logging_type(PER_MINIMIZER_ITERATION),
      initial_trust_region_radius(1e4), max_trust_region_radius(1e16), min_trust_region_radius(1e-32),
      min_relative_decrease(1e-3), min_lm_diagonal(1e-6), max_lm_diagonal(1e32),
      max_num_consecutive_invalid_steps(5), function_tolerance(1e-6), gradient_tolerance(1e-10),
      parameter_tolerance(1e-8), jacobi_scaling(true) {}

Solver::Options::~Options() { delete linear_solver_ordering; }

Solver::Summary::Summary()
    : termination_type(DID_NOT_RUN), initial_cost(-1.0), final_cost(-1.0), fixed_cost(-1.0),
      num_successful_steps(-1), num_unsuccessful_steps(-1), num_parameters_reduced(-1),
      num_residual_blocks_reduced(-1), backend_status(0) {}

static const char* termination_name(SolverTerminationType t) {
  switch (t) {
    case NO_CONVERGENCE: return "NO_CONVERGENCE";
    case FUNCTION_TOLERANCE: return "FUNCTION_TOLERANCE";
    case GRADIENT_TOLERANCE: return "GRADIENT_TOLERANCE";
    case PARAMETER_TOLERANCE: return "PARAMETER_TOLERANCE";
    case NUMERICAL_FAILURE: return "NUMERICAL_FAILURE";
    default: return "DID_NOT_RUN";
  }
}

std::string Solver::Summary::BriefReport() const {
  std::ostringstream o;
  o << "slslam_amd (MI355X) report: iterations: " << (num_successful_steps + num_unsuccessful_steps)
    << ", initial cost: " << initial_cost << ", final cost: " << final_cost << ", termination: " << termination_name(termination_type);
  return o.str();
}

std::string Solver::Summary::FullReport() const {
  std::ostringstream o;
  o << BriefReport() << "\n  reduced parameters: " << num_parameters_reduced << "  residual blocks: " << num_residual_blocks_reduced
    << "  fixed cost: " << fixed_cost << "  successful/unsuccessful steps: " << num_successful_steps << "/" << num_unsuccessful_steps
    << "  backend status: " << slslam_status_string(backend_status) << "\n";
  return o.str();
}

static void fill_options(const Solver::Options& in, slslam_solver_options* o) {
  slslam_default_options(o);
  o->max_num_iterations = in.max_num_iterations;
  o->initial_trust_region_radius = in.initial_trust_region_radius;
  o->max_trust_region_radius = in.max_trust_region_radius;
  o->min_trust_region_radius = in.min_trust_region_radius;
  o->min_relative_decrease = in.min_relative_decrease;
  o->min_lm_diagonal = in.min_lm_diagonal;
  o->max_lm_diagonal = in.max_lm_diagonal;
  o->max_num_consecutive_invalid_steps = in.max_num_consecutive_invalid_steps;
  o->function_tolerance = in.function_tolerance;
  o->gradient_tolerance = in.gradient_tolerance;
  o->parameter_tolerance = in.parameter_tolerance;
  o->jacobi_scaling = in.jacobi_scaling ? 1 : 0;
}

static SolverTerminationType map_termination(int t) {
  switch (t) {
    case SLSLAM_GRADIENT_TOLERANCE: return GRADIENT_TOLERANCE;
    case SLSLAM_FUNCTION_TOLERANCE: return FUNCTION_TOLERANCE;
    case SLSLAM_PARAMETER_TOLERANCE: case SLSLAM_MIN_RADIUS: return PARAMETER_TOLERANCE;
    case SLSLAM_NUMERICAL_FAILURE: return NUMERICAL_FAILURE;
    default: return NO_CONVERGENCE;
  }
}

void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary) {
  Solver::Summary local;
  Solver::Summary* s = summary ? summary : &local;
  *s = Solver::Summary();
  slslam_solver_options o;
  fill_options(options, &o);
  slslam_summary r;
  int rc = SLSLAM_ERR_INVALID_ARGUMENT;
  if (problem && problem->lba()) {
    LBAProblem* p = problem->lba();
    slslam_lba_window w;
    w.num_cameras = p->num_cameras(); w.num_lines = p->num_lines(); w.num_observations = p->num_observations();
    w.camera_index = p->camera_index(); w.line_index = p->line_index(); w.fixed_index = p->fixed_index();
    w.observations = p->observations(); w.parameters = p->mutable_cameras();
    if (!p->robust()) o.huber_delta = 0.0;      // robustify ? HuberLoss(1/406.05) : NULL  (lba_problem.cpp:78-80)
    rc = slslam_lba_solve(&w, &o, &r, nullptr, 0, nullptr);
  } else if (problem && problem->po()) {
    POProblem* p = problem->po();
    slslam_po_graph g;
    g.num_poses = p->num_poses(); g.num_edges = p->num_size();
    g.pose_index_1 = p->pose_index_1(); g.pose_index_2 = p->pose_index_2();
    g.constraints = p->constraints(); g.parameters = p->parameters();
    rc = slslam_po_solve(&g, &o, &r, nullptr, 0, nullptr);
  }
  s->backend_status = rc;
  if (rc != SLSLAM_OK) {
    // the reference ignores failures (nothing is returned by ceres::Solve, slam.cpp:663,944,1293);
    // make them loud instead of silently leaving the parameters unsolved.
    std::fprintf(stderr, "slslam_amd: ceres::Solve failed: %s\n", slslam_status_string(rc));
    s->termination_type = NUMERICAL_FAILURE;
    return;
  }
  s->termination_type = map_termination(r.termination_type);
  s->initial_cost = r.initial_cost; s->final_cost = r.final_cost; s->fixed_cost = r.fixed_cost;
  s->num_successful_steps = r.num_successful_steps; s->num_unsuccessful_steps = r.num_unsuccessful_steps;
  s->num_parameters_reduced = r.num_free_parameters; s->num_residual_blocks_reduced = r.num_residual_blocks;
  if (options.logging_type != SILENT && options.minimizer_progress_to_stdout) std::printf("%s\n", s->BriefReport().c_str());
}

}  // namespace ceres
