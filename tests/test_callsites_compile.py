"""The three call sites of the reference compile UNCHANGED against the mirrored headers (slslam_amd/host).

Build-container-only check: the blocks are read from /root/reference/src/slam.cpp at test time (nothing of them is kept
in this repository), each is wrapped in a function that declares the locals the surrounding reference code provides
(std::vector's of indices and of fixed-size vectors with operator(), the summary accumulators, kfs[i]->T, ...), and the
translation unit is compiled with g++ against slslam_amd/host.  Skipped where the reference tree is absent (GPU box).

Blocks (SURVEY.md 8b "call protocol"): SLAM::motion_only_ba src/slam.cpp:618-663, SLAM::bundle_adjustment :899-952,
SLAM::pose_optimization :1262-1293."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/slam.cpp"
HOST = os.path.join(ROOT, "slslam_amd", "host")

PRELUDE = r"""
#include <iostream>
#include <vector>
#include <string>
#include "lba_problem.h"
#include "po_problem.h"
using namespace std;
// what the reference gets from Eigen / gflags / its own headers at these call sites: fixed-size vectors read with
// operator() and operator[], the iteration flag, pose_t with gc_Rt_to_wt
template <int N> struct VecN { double v[N]; double operator()(int i) const { return v[i]; } double& operator()(int i) { return v[i]; }
                               double operator[](int i) const { return v[i]; } double& operator[](int i) { return v[i]; } };
typedef VecN<4> Vector4d; typedef VecN<6> Vector6d; typedef VecN<8> Vector8d;
struct pose_t { double R[9], t[3]; };
static Vector6d gc_Rt_to_wt(const pose_t&) { return Vector6d(); }
static int FLAGS_max_num_iter = 10;
struct keyframe_t { pose_t T; };
"""


def _block(lines, a, b):
    return "".join(lines[a - 1:b])


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_reference_call_sites_compile_unchanged(tmp_path):
    lines = open(REF).readlines()
    moba = _block(lines, 618, 663)
    ba = _block(lines, 899, 952)
    po = _block(lines, 1262, 1293)
    for text, needle in ((moba, "ceres::LBAProblem ba_problem( param );"), (ba, "m_sum_num_iteration"), (po, "ceres::POProblem po_problem")):
        assert needle in text and "ceres::Solve(options, &problem, &summary);" in text      # the line ranges still hold the blocks
    src = PRELUDE + r"""
double* site_motion_only_ba(vector<int>& vec_line_index, vector<int>& vec_camera_index, vector<int>& vec_fixed_index,
                            vector<Vector8d>& vec_observations, vector<Vector6d>& vec_camera_param, vector<Vector4d>& vec_line_param) {
  int num_cameras = vec_camera_param.size(), num_lines = vec_line_param.size();
  int num_parameters = 6 * num_cameras + 4 * num_lines, num_observations = vec_observations.size();
""" + moba + r"""
  double out = parameters[0]; (void)out;
  return 0;
}
int m_sum_num_iteration; double m_sum_final_cost, m_sum_init_cost;
void site_bundle_adjustment(vector<int>& vec_line_index, vector<int>& vec_camera_index, vector<int>& vec_fixed_index,
                            vector<Vector8d>& vec_observations, vector<Vector6d>& vec_camera_param, vector<Vector4d>& vec_line_param) {
  int num_cameras = vec_camera_param.size(), num_lines = vec_line_param.size();
  int num_parameters = 6 * num_cameras + 4 * num_lines, num_observations = vec_observations.size();
  (void)num_cameras; (void)num_lines; (void)num_parameters; (void)num_observations;
  {
""" + ba + r"""
  }
}
void site_pose_optimization(vector<int>& pose_vec_1, vector<int>& pose_vec_2, vector<pose_t>& ctrs_vec, vector<keyframe_t*>& kfs, int edge_size) {
  int kfs_size = kfs.size();
""" + po + r"""
}
int main() { return 0; }
"""
    f = tmp_path / "callsites.cpp"
    f.write_text(src)
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-Wno-unused-variable", "-I", HOST, str(f)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # and they link against the host library (the marshalling into the C ABI)
    lib = os.path.join(ROOT, "slslam_amd", "_lib", "libslslam_host.so")
    if os.path.exists(lib):
        exe = tmp_path / "callsites"
        r = subprocess.run(["g++", "-std=c++11", "-Wno-unused-variable", "-I", HOST, str(f), "-o", str(exe), lib,
                            "-Wl,-rpath," + os.path.dirname(lib)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
