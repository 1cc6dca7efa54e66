// slslam_amd/host/po_problem.h — host-side mirror of the reference's POProblem surface
// (reference src/po_problem.h:110-144) for SLAM::pose_optimization (src/slam.cpp:1262-1301).
// Same names, argument meaning and ownership (destructor delete[]s the four arrays,
// src/po_problem.cpp:33-38).  The SE(3) functor (src/po_problem.h:27-108) runs on the GPU
// (slslam_amd/csrc/po_kernels.h).
//
// Attribution: the class surface declared here (names, signatures, accessor layout) mirrors the interface of
// SLSLAM's src/po_problem.h — Copyright (C) 2015 Guoxuan Zhang, Jin Han Lee, Jongwoo Lim, Il Hong Suh, distributed under the
// GNU General Public License, version 2 or later — because the drop-in contract is that interface.  Only the
// declarations are mirrored; the implementation behind them is this repository's own.
#ifndef PO_PROBLEM_H_
#define PO_PROBLEM_H_

#include <string>
#include "ceres/ceres.h"
#include "ceres/rotation.h"

namespace ceres {

// Edge constraint C = T_{n2<-n1} as (angle-axis, translation) (reference src/po_problem.h:68-72,108)
struct PoseConstraintError {
  PoseConstraintError(double wo0, double wo1, double wo2, double to0, double to1, double to2)
      : wo0(wo0), wo1(wo1), wo2(wo2), to0(to0), to1(to1), to2(to2) {}
  double wo0, wo1, wo2, to0, to1, to2;
};

class POProblem {
 public:
  explicit POProblem(int s, int n);
  ~POProblem();

  int pose_block_size()        const { return 6;             }
  int num_size()               const { return size_;         }
  const int* pose_index_1()    const { return pose_index_1_; }
  const int* pose_index_2()    const { return pose_index_2_; }
  const double* constraints()  const { return constraints_;  }
  double* parameters()         const { return parameters_;   }

  inline void set_size(int s)             { size_ = s;           }
  inline void set_num_iterations(int s)   { num_iterations = s;  }
  inline void set_pose_index_1(int* idx)  { pose_index_1_ = idx; }
  inline void set_pose_index_2(int* idx)  { pose_index_2_ = idx; }
  inline void set_constraints(double* d)  { constraints_ = d;    }
  inline void set_parameters(double* d)   { parameters_ = d;     }
  // The reference sizes `parameters` by kfs.size() and never tells POProblem (slam.cpp:1265,1276-1280);
  // the back-end needs the count to size the dense system, so by default it is inferred as
  // 1 + max pose index over the edges; a caller with trailing unreferenced poses may set it.
  inline void set_num_poses(int n)        { num_poses_ = n;      }
  int num_poses() const;

  void build(Problem* problem);
  void set_options(Solver::Options* options);

 private:
  int num_iterations;
  int num_threads;
  double eta;
  bool robustify;
  int size_;
  int num_poses_;

  int* pose_index_1_;
  int* pose_index_2_;
  double* constraints_;
  double* parameters_;
};

}  // namespace ceres
#endif  // PO_PROBLEM_H_
