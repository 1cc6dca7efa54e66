// slslam_amd/host/po_problem.h — host-side mirror of the reference's POProblem surface
// (reference src/po_problem.h:110-144) for SLAM::pose_optimization (src/slam.cpp:1262-1301).
// Same names, argument meaning and ownership (destructor delete[]s the four arrays,
// src/po_problem.cpp:33-38).  The SE(3) functor (src/po_problem.h:27-108) runs on the GPU
// (slslam_amd/csrc/po_kernels.h); its three SE(3) templates are kept below for host callers.
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

// The SE(3) helpers of the reference header (src/po_problem.h:27-64), poses as [angle-axis (3) | translation (3)], kept so that
// code written against that header finds the same three templates.  On the GPU the same algebra runs on dual numbers
// (slslam_amd/csrc/po_kernels.h); these host forms are for callers and for the tests (tests/test_host_cxx.py compares them with
// the matrix forms of gc_lite.h).
// Pi = P^-1: rotation -w, translation R(-w)(-t)
template <typename T>
void gc_T_inv(T P[6], T Pi[6]) {
  T minus_t[3];
  for (int i = 0; i < 3; ++i) { Pi[i] = -P[i]; minus_t[i] = -P[3 + i]; }
  ceres::AngleAxisRotatePoint(Pi, minus_t, Pi + 3);
}
// R20 = R21 R10 on angle-axis vectors, through unit quaternions
template <typename T>
void gc_w_20(T w21[3], T w10[3], T w20[3]) {
  T qa[4], qb[4], qab[4];
  ceres::AngleAxisToQuaternion(w21, qa);
  ceres::AngleAxisToQuaternion(w10, qb);
  ceres::QuaternionProduct(qa, qb, qab);
  ceres::QuaternionToAngleAxis(qab, w20);
}
// T20 = T21 T10: w20 = w21 (+) w10, t20 = R(w21) t10 + t21
template <typename T>
void gc_T_20(T T21[6], T T10[6], T T20[6]) {
  gc_w_20(T21, T10, T20);
  ceres::AngleAxisRotatePoint(T21, T10 + 3, T20 + 3);
  for (int i = 3; i < 6; ++i) T20[i] += T21[i];
}

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
