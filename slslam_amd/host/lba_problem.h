// slslam_amd/host/lba_problem.h — host-side mirror of the reference's LBAProblem surface
// (reference src/lba_problem.h:32-33, :123-197), so that SLAM::bundle_adjustment and
// SLAM::motion_only_ba (src/slam.cpp:899-972, :618-674) keep working as written:
//
//   ceres::lba_param_t param; ...                    ceres::LBAProblem ba_problem(param);
//   ba_problem.set_line_index(line_index); ...       ceres::Problem problem;
//   ba_problem.build(&problem);                       ceres::Solver::Options options;
//   ba_problem.set_options(&options);                 ceres::Solver::Summary summary;
//   ceres::Solve(options, &problem, &summary);        -> parameters[] solved in place
//
// Same names, argument meaning and ownership (the destructor delete[]s the five arrays handed in,
// src/lba_problem.cpp:46-52).  build() binds the arrays instead of allocating M cost functions;
// the residual functor itself (src/lba_problem.h:46-118) lives in slslam_amd/csrc/lba_math.h and
// runs on the GPU.
//
// Attribution: the class surface declared here (names, signatures, accessor layout) mirrors the interface of
// SLSLAM's src/lba_problem.h — Copyright (C) 2015 Guoxuan Zhang, Jin Han Lee, Jongwoo Lim, Il Hong Suh, distributed under the
// GNU General Public License, version 2 or later — because the drop-in contract is that interface.  Only the
// declarations are mirrored; the implementation behind them is this repository's own.
#ifndef LBA_PROBLEM_H_
#define LBA_PROBLEM_H_

#include <string>
#include "ceres/ceres.h"
#include "ceres/rotation.h"

#define MODE_SPARSE_SCHUR            1
#define MODE_SPARSE_NORMAL_CHOLESKY  2

namespace slslam {
// Stands in for gflags' FLAGS_robust (reference src/main.cpp:27, read at src/lba_problem.cpp:35)
// when the caller does not link gflags.  Default true, as the reference's flag.
extern bool flag_robust;
}

namespace ceres {

// Observation of one line in a stereo pair: endpoints (x0,y0),(x1,y1) in the first camera and
// (x2,y2),(x3,y3) in the second (reference src/lba_problem.h:41-44,120).  Kept as the data holder
// it is in the reference; the templated operator() is evaluated by the HIP kernels.
struct LineReprojectionError {
  LineReprojectionError(double x0, double y0, double x1, double y1, double x2, double y2, double x3, double y3)
      : x0(x0), y0(y0), x1(x1), y1(y1), x2(x2), y2(y2), x3(x3), y3(y3) {}
  double x0, y0, x1, y1, x2, y2, x3, y3;
};

typedef struct {
  int num_cameras;
  int num_lines;
  int num_observations;
  int num_iterations;
  int num_parameters;
  int mode;
} lba_param_t;

class LBAProblem {
 public:
  explicit LBAProblem(lba_param_t param);
  ~LBAProblem();

  int camera_block_size()      const { return 6;                 }
  int line_block_size()        const { return 4;                 }
  int num_cameras()            const { return num_cameras_;      }
  int num_lines()              const { return num_lines_;        }
  int num_observations()       const { return num_observations_; }
  int num_parameters()         const { return num_parameters_;   }
  const int* line_index()      const { return line_index_;       }
  const int* camera_index()    const { return camera_index_;     }
  const int* fixed_index()     const { return fixed_index_;      }
  const double* observations() const { return observations_;     }
  const double* parameters()   const { return parameters_;       }
  double* mutable_cameras()          { return parameters_;       }
  double* mutable_lines()            { return parameters_ + camera_block_size() * num_cameras_; }

  inline void set_line_index(int* idx)    { line_index_ = idx;   }
  inline void set_camera_index(int* idx)  { camera_index_ = idx; }
  inline void set_fixed_index(int* idx)   { fixed_index_ = idx;  }
  inline void set_observations(double* d) { observations_ = d;   }
  inline void set_parameters(double* d)   { parameters_ = d;     }
  inline void set_logging_type(bool b)    { logging_type = b;    }

  void build(Problem* problem);
  void set_options(Solver::Options* options);

  // read by ceres::Solve when it marshals into the C ABI
  bool robust() const { return robustify; }
  int num_iterations() const { return num_iterations_; }
  int mode() const { return mode_; }

 private:
  int mode_;
  int num_cameras_;
  int num_lines_;
  int num_observations_;
  int num_parameters_;
  int num_iterations_;
  int num_threads;
  double eta;
  bool robustify;
  bool logging_type;

  int* line_index_;
  int* camera_index_;
  int* fixed_index_;
  double* observations_;
  // [camera_1, ..., camera_n, line_1, ..., line_m]
  double* parameters_;
};

}  // namespace ceres
#endif  // LBA_PROBLEM_H_
