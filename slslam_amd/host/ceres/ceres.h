// slslam_amd/host/ceres/ceres.h — the sliver of the Ceres 1.7 API that leaks through the
// reference's LBAProblem / POProblem signatures and their three call sites
// (reference src/slam.cpp:643-663, :924-952, :1283-1293), re-declared so that those call sites
// compile UNCHANGED against the MI355X back-end.  Nothing here optimises anything on the CPU:
// ceres::Problem only records which problem object wired itself up, and ceres::Solve marshals
// into the C ABI (include/slslam_hip.h), which runs the hand-written HIP kernels.
//
// Names kept: ceres::Problem, ceres::Solver::{Options,Summary}, ceres::Solve,
// ceres::ParameterBlockOrdering::AddElementToGroup, enums SPARSE_SCHUR / SPARSE_NORMAL_CHOLESKY /
// SILENT, Options fields linear_solver_type, num_linear_solver_threads, linear_solver_ordering,
// max_num_iterations, minimizer_progress_to_stdout, num_threads, eta, logging_type; Summary fields
// num_successful_steps, num_unsuccessful_steps, initial_cost, final_cost, FullReport().
#ifndef SLSLAM_HOST_CERES_CERES_H_
#define SLSLAM_HOST_CERES_CERES_H_

#include <string>

namespace ceres {

enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR };
enum LoggingType { SILENT, PER_MINIMIZER_ITERATION };
enum SolverTerminationType {
  DID_NOT_RUN, NO_CONVERGENCE, FUNCTION_TOLERANCE, GRADIENT_TOLERANCE, PARAMETER_TOLERANCE, NUMERICAL_FAILURE, USER_ABORT, USER_SUCCESS
};

// The reference puts every block into group 0 (src/lba_problem.cpp:114-122): a direct solver's
// elimination ordering changes round-off, not the step, so the ordering is accepted and ignored.
class ParameterBlockOrdering {
 public:
  bool AddElementToGroup(const double*, int) { ++num_elements_; return true; }
  int NumElements() const { return num_elements_; }
 private:
  int num_elements_ = 0;
};

class LBAProblem;
class POProblem;

// Records what XProblem::build wired up.  The reference calls problem->AddResidualBlock(...) M
// times inside build(); here build() binds the whole array contract in one call.
class Problem {
 public:
  Problem() : lba_(nullptr), po_(nullptr) {}
  void BindLBA(LBAProblem* p) { lba_ = p; po_ = nullptr; }
  void BindPO(POProblem* p) { po_ = p; lba_ = nullptr; }
  LBAProblem* lba() const { return lba_; }
  POProblem* po() const { return po_; }
 private:
  LBAProblem* lba_;
  POProblem* po_;
};

class Solver {
 public:
  struct Options {
    Options();
    ~Options();
    LinearSolverType linear_solver_type;
    int num_linear_solver_threads;
    ParameterBlockOrdering* linear_solver_ordering;   // owned, as in Ceres 1.7
    int max_num_iterations;
    bool minimizer_progress_to_stdout;
    int num_threads;
    double eta;
    LoggingType logging_type;
    // trust-region policy (Ceres 1.7 defaults; the reference never touches them)
    double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
    double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
    int max_num_consecutive_invalid_steps;
    double function_tolerance, gradient_tolerance, parameter_tolerance;
    bool jacobi_scaling;
   private:
    Options(const Options&);
    Options& operator=(const Options&);
  };
  struct Summary {
    Summary();
    std::string BriefReport() const;
    std::string FullReport() const;
    SolverTerminationType termination_type;
    double initial_cost, final_cost, fixed_cost;
    int num_successful_steps, num_unsuccessful_steps;
    int num_parameters_reduced, num_residual_blocks_reduced;
    int backend_status;            // SLSLAM_* status of the C ABI call (0 = ok)
  };
};

// Runs the bound problem on the GPU through slslam_lba_solve / slslam_po_solve.
void Solve(const Solver::Options& options, Problem* problem, Solver::Summary* summary);

}  // namespace ceres
#endif
