// slslam_amd/csrc/ransac_api.hip — RANSAC hypothesis scoring.
//
// Replaces the scoring loop of SLAM::ransac_motion (reference src/slam.cpp:396-413) whose body is
// SLAM::reprojection_error (src/slam.cpp:691-726).  One 64-lane wave scores one hypothesis against 64
// lines (lane <-> line): the pose is wave-uniform (scalar loads), observations and lines are read
// coalesced, the inlier set of the block is one __ballot() word and its popcount the block's score.
// The reference mixes float and double (float `sql = nc.head(2).norm()`, `float error`); the same
// conversions are applied in the same places so that scores and inlier sets are bit-identical.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../include/slslam_hip.h"

namespace {

// no FMA contraction: the reference evaluates these expressions with separate multiplies and adds,
// and the inlier test sits on a threshold
#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void k_ransac_score(int H, int K, int words, const double* __restrict__ poses,
                                                     const double* __restrict__ obs, const double* __restrict__ lines,
                                                     double baseline, double thr, int* scores, unsigned long long* bits) {
  const int h = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
  const int k = blk * 64 + lane;
  const double* T = poses + 12 * (long long)h;
  const double t0 = T[9], t1 = T[10], t2 = T[11];
  // `if ( motion[j].t.norm() > 1 ) continue;`  (slam.cpp:398-399)
  if (sqrt(t0 * t0 + t1 * t1 + t2 * t2) > 1.0) {
    if (blk == 0 && lane == 0) scores[h] = -1;
    if (bits && lane == 0) bits[(long long)h * words + blk] = 0ull;
    return;
  }
  bool inlier = false;
  if (k < K) {
    const double* ft = obs + 8 * (long long)k;
    const double* ln = lines + 6 * (long long)k;
    const double cp[3] = { ln[0], ln[1], ln[2] }, dv[3] = { ln[3], ln[4], ln[5] };
    double tt0 = t0;
    float error = 0.f;
    // dvc = T.R * dv is the same for both cameras
    const double d0 = T[0] * dv[0] + T[1] * dv[1] + T[2] * dv[2];
    const double d1 = T[3] * dv[0] + T[4] * dv[1] + T[5] * dv[2];
    const double d2 = T[6] * dv[0] + T[7] * dv[1] + T[8] * dv[2];
    for (int i = 0; i < 2; ++i) {
      if (i == 1) tt0 -= baseline;                                   // T.t(0) -= baseline
      const double c0 = T[0] * cp[0] + T[1] * cp[1] + T[2] * cp[2] + tt0;   // gc_point_to_pose
      const double c1 = T[3] * cp[0] + T[4] * cp[1] + T[5] * cp[2] + t1;
      const double c2 = T[6] * cp[0] + T[7] * cp[1] + T[8] * cp[2] + t2;
      double n0 = c1 * d2 - c2 * d1, n1 = c2 * d0 - c0 * d2, n2 = c0 * d1 - c1 * d0;   // cpc.cross(dvc)
      const float sql = (float)sqrt(n0 * n0 + n1 * n1);              // float sql = nc.head(2).norm()
      n0 /= (double)sql; n1 /= (double)sql; n2 /= (double)sql;       // nc /= sql
      const double e1 = fabs(n0 * ft[4 * i] + n1 * ft[4 * i + 1] + n2);          // nc.dot(p1), p1 = (x, y, 1)
      const double e2 = fabs(n0 * ft[4 * i + 2] + n1 * ft[4 * i + 3] + n2);
      error = (float)((double)error + e1);                           // float error += double
      error = (float)((double)error + e2);
    }
    const float ret = (float)((double)error / 4.0);                  // return error / 4.0  (float function)
    inlier = (double)ret < thr;                                      // error < error_thr (double)
  }
  const unsigned long long m = __ballot(inlier);
  if (lane == 0) {
    if (bits) bits[(long long)h * words + blk] = m;
    atomicAdd(&scores[h], __popcll(m));
  }
}

}  // namespace

extern "C" int slslam_ransac_score(const slslam_ransac_frame* f, double baseline, double thr, int* scores,
                                   unsigned long long* inlier_bits) {
  if (!f || !scores || f->num_hypotheses < 0 || f->num_lines < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  const int H = f->num_hypotheses, K = f->num_lines;
  if (H > 0 && !f->poses) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (K > 0 && (!f->observations || !f->lines)) return SLSLAM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  if (H == 0) return SLSLAM_OK;
  const int words = (K + 63) / 64;
  if (K == 0) {
    for (int h = 0; h < H; ++h) {
      const double* t = f->poses + 12 * (size_t)h + 9;
      scores[h] = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]) > 1.0 ? -1 : 0;
    }
    return SLSLAM_OK;
  }
  double *d_poses = nullptr, *d_obs = nullptr, *d_lines = nullptr;
  int* d_scores = nullptr;
  unsigned long long* d_bits = nullptr;
  int rc = SLSLAM_OK;
#define RS_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { std::fprintf(stderr, "slslam: %s failed: %s\n", #expr, hipGetErrorString(_e)); rc = SLSLAM_ERR_HIP; goto done; } } while (0)
  RS_TRY(hipMalloc((void**)&d_poses, sizeof(double) * 12 * H));
  RS_TRY(hipMalloc((void**)&d_obs, sizeof(double) * 8 * K));
  RS_TRY(hipMalloc((void**)&d_lines, sizeof(double) * 6 * K));
  RS_TRY(hipMalloc((void**)&d_scores, sizeof(int) * H));
  RS_TRY(hipMalloc((void**)&d_bits, sizeof(unsigned long long) * (size_t)H * words));
  RS_TRY(hipMemcpy(d_poses, f->poses, sizeof(double) * 12 * H, hipMemcpyHostToDevice));
  RS_TRY(hipMemcpy(d_obs, f->observations, sizeof(double) * 8 * K, hipMemcpyHostToDevice));
  RS_TRY(hipMemcpy(d_lines, f->lines, sizeof(double) * 6 * K, hipMemcpyHostToDevice));
  RS_TRY(hipMemset(d_scores, 0, sizeof(int) * H));
  hipLaunchKernelGGL(k_ransac_score, dim3((unsigned)words, (unsigned)H), dim3(64), 0, 0, H, K, words, d_poses, d_obs, d_lines,
                     baseline, thr, d_scores, d_bits);
  RS_TRY(hipGetLastError());
  RS_TRY(hipMemcpy(scores, d_scores, sizeof(int) * H, hipMemcpyDeviceToHost));
  if (inlier_bits) RS_TRY(hipMemcpy(inlier_bits, d_bits, sizeof(unsigned long long) * (size_t)H * words, hipMemcpyDeviceToHost));
#undef RS_TRY
done:
  (void)hipFree(d_poses); (void)hipFree(d_obs); (void)hipFree(d_lines); (void)hipFree(d_scores); (void)hipFree(d_bits);
  return rc;
}
