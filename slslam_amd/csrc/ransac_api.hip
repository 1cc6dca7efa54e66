// slslam_amd/csrc/ransac_api.hip — RANSAC hypothesis scoring.
//
// Replaces the scoring loop of SLAM::ransac_motion (reference src/slam.cpp:396-413) whose body is
// SLAM::reprojection_error (src/slam.cpp:691-726).  One 64-lane wave scores one hypothesis against 64
// lines (lane <-> line): the pose is wave-uniform (scalar loads), observations and lines are read
// coalesced, the inlier set of the block is one __ballot() word and its popcount the block's score.
// The reference mixes float and double (float `sql = nc.head(2).norm()`, `float error`); the same
// conversions are applied in the same places so that scores and inlier sets are bit-identical.
#include <hip/hip_runtime.h>

#include <algorithm>
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
                                                     double baseline, double thr, int* scores, unsigned long long* bits,
                                                     const int* __restrict__ valid) {
  const int h = blockIdx.y, blk = blockIdx.x, lane = threadIdx.x;
  const int k = blk * 64 + lane;
  const double* T = poses + 12 * (long long)h;
  const double t0 = T[9], t1 = T[10], t2 = T[11];
  // `if ( num_sol == 0 ) continue;` (slam.cpp:394) and `if ( motion[j].t.norm() > 1 ) continue;` (:398-399)
  if ((valid && !valid[h]) || sqrt(t0 * t0 + t1 * t1 + t2 * t2) > 1.0) {
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


// ---------------------------------------------------------------------------------------------
// Hypothesis generation: SLAM::vo_angle_axis_approx (reference src/slam.cpp:433-574), lane <-> trial.
// Each lane walks its s sampled correspondences twice (rotation rows K, then translation rows M),
// accumulating the 3x3 normal matrices in registers; (A^T A)^-1 by partially pivoted LU as Eigen's
// dynamic inverse() does.  No FMA contraction (pragma above): same operation order as the oracle.
__device__ inline void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ inline double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
__device__ inline void image_line(const double* ob, double l[3]) {
  const double p1[3] = { ob[0], ob[1], 1 }, p2[3] = { ob[2], ob[3], 1 };
  cross3(p1, p2, l);
}
__device__ inline void solve_normal3(const double N[9], const double v[3], double x[3]) {
  double a[3][6];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = N[3 * i + j]; a[i][3 + j] = i == j ? 1.0 : 0.0; }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int piv = c;
#pragma unroll
    for (int r = c + 1; r < 3; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
#pragma unroll
    for (int r = c + 1; r < 3; ++r)
      if (piv == r)
        for (int j = 0; j < 6; ++j) { const double t = a[c][j]; a[c][j] = a[r][j]; a[r][j] = t; }
#pragma unroll
    for (int r = c + 1; r < 3; ++r) {
      const double f = a[r][c] / a[c][c];
#pragma unroll
      for (int j = c; j < 6; ++j) a[r][j] -= f * a[c][j];
    }
  }
  double inv[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 2; i >= 0; --i) {
      double s = a[i][3 + j];
#pragma unroll
      for (int k = i + 1; k < 3; ++k) s -= a[i][k] * inv[k][j];
      inv[i][j] = s / a[i][i];
    }
#pragma unroll
  for (int i = 0; i < 3; ++i) x[i] = inv[i][0] * v[0] + inv[i][1] * v[1] + inv[i][2] * v[2];
}
__device__ inline void aa_to_matrix(const double w[3], double R[9]) {   // ceres::AngleAxisToRotationMatrix, row-major
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2), wx = w[0] / th, wy = w[1] / th, wz = w[2] / th;
    const double c = cos(th), s = sin(th);
    R[0] = c + wx * wx * (1 - c);       R[1] = wx * wy * (1 - c) - wz * s;  R[2] = wy * s + wx * wz * (1 - c);
    R[3] = wz * s + wx * wy * (1 - c);  R[4] = c + wy * wy * (1 - c);       R[5] = -wx * s + wy * wz * (1 - c);
    R[6] = -wy * s + wx * wz * (1 - c); R[7] = wx * s + wy * wz * (1 - c);  R[8] = c + wz * wz * (1 - c);
  } else {
    R[0] = 1; R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2]; R[4] = 1; R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
  }
}

__global__ __launch_bounds__(64) void k_ransac_generate(int H, int s, const int* __restrict__ samples,
                                                        const double* __restrict__ obs0, const double* __restrict__ obs1,
                                                        double baseline, double* poses, int* valid) {
  const int h = blockIdx.x * 64 + threadIdx.x;
  if (h >= H) return;
  const int* smp = samples + (long long)h * s;
  double N[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, v[3] = { 0, 0, 0 };
  bool ok = true;
  for (int i = 0; i < s; ++i) {                                          // slam.cpp:437-482
    const double* o0 = obs0 + 8 * (long long)smp[i];
    const double* o1 = obs1 + 8 * (long long)smp[i];
    double l1[3], l2[3], l3[3], l4[3], lx[3];
    image_line(o0, l1); image_line(o0 + 4, l2); image_line(o1, l3); image_line(o1 + 4, l4);
    cross3(l1, l2, lx);
    const double lxn = norm3(lx);
    if (lxn == 0) ok = false;
    lx[0] /= lxn; lx[1] /= lxn; lx[2] /= lxn;
    for (int j = 0; j < 2; ++j) {
      const double* tl = j == 0 ? l3 : l4;
      const double tln = norm3(tl);
      if (tln == 0) ok = false;
      const double ly[3] = { tl[0] / tln, tl[1] / tln, tl[2] / tln };
      const double row[4] = { lx[2] * ly[1] - lx[1] * ly[2], lx[0] * ly[2] - lx[2] * ly[0], lx[1] * ly[0] - lx[0] * ly[1],
                              lx[0] * ly[0] + lx[1] * ly[1] + lx[2] * ly[2] };
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) N[3 * a + b] += row[a] * row[b];
        v[a] += row[a] * (-row[3]);
      }
    }
  }
  double w[3], R[9];
  solve_normal3(N, v, w);                                                // :484-488
  w[0] = -w[0]; w[1] = -w[1]; w[2] = -w[2];
  aa_to_matrix(w, R);
  for (int a = 0; a < 9; ++a) N[a] = 0;
  v[0] = v[1] = v[2] = 0;
  for (int i = 0; i < s; ++i) {                                          // :495-559
    const double* o0 = obs0 + 8 * (long long)smp[i];
    const double* o1 = obs1 + 8 * (long long)smp[i];
    double l1[3], l2[3], lx[3];
    image_line(o0, l1);
    const double l1n = norm3(l1);
    if (l1n == 0) ok = false;
    l1[0] /= l1n; l1[1] /= l1n; l1[2] /= l1n;
    image_line(o0 + 4, l2);
    const double l2n = norm3(l2);
    if (l2n == 0) ok = false;
    l2[0] /= l2n; l2[1] /= l2n; l2[2] /= l2n;
    cross3(l1, l2, lx);
    if (norm3(lx) == 0) ok = false;
    for (int j = 0; j < 2; ++j) {
      double l3[3];
      image_line(o1 + 4 * j, l3);
      const double l3n = norm3(l3);
      if (l3n == 0) ok = false;
      l3[0] /= l3n; l3[1] /= l3n; l3[2] /= l3n;
      double c[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double rc[3] = { R[k], R[3 + k], R[6 + k] };
        double u[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) u[q] = -l2[0] * (baseline * rc[q]) + -l2[1] * (0.0 * rc[q]) + -l2[2] * (0.0 * rc[q]);
        c[k] = u[0] * l3[0] + u[1] * l3[1] + u[2] * l3[2];
        if (j == 1) c[k] += l2[k] * baseline * l3[0];
      }
      const double rows[3][4] = {
        { l1[1] * l2[2] * l3[0] - l1[2] * l2[1] * l3[0], l1[1] * l2[2] * l3[1] - l1[2] * l2[1] * l3[1],
          l1[1] * l2[2] * l3[2] - l1[2] * l2[1] * l3[2], l1[1] * c[2] - l1[2] * c[1] },
        { l1[2] * l2[0] * l3[0] - l1[0] * l2[2] * l3[0], l1[2] * l2[0] * l3[1] - l1[0] * l2[2] * l3[1],
          l1[2] * l2[0] * l3[2] - l1[0] * l2[2] * l3[2], l1[2] * c[0] - l1[0] * c[2] },
        { l1[0] * l2[1] * l3[0] - l1[1] * l2[0] * l3[0], l1[0] * l2[1] * l3[1] - l1[1] * l2[0] * l3[1],
          l1[0] * l2[1] * l3[2] - l1[1] * l2[0] * l3[2], l1[0] * c[1] - l1[1] * c[0] } };
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
          for (int b = 0; b < 3; ++b) N[3 * a + b] += rows[r][a] * rows[r][b];
          v[a] += rows[r][a] * (-rows[r][3]);
        }
    }
  }
  double t[3];
  solve_normal3(N, v, t);                                                // :561-565
  double* P = poses + 12 * (long long)h;
  for (int q = 0; q < 9; ++q) P[q] = R[q];
  P[9] = t[0]; P[10] = t[1]; P[11] = t[2];
  valid[h] = ok ? 1 : 0;
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
                     baseline, thr, d_scores, d_bits, (const int*)nullptr);
  RS_TRY(hipGetLastError());
  RS_TRY(hipMemcpy(scores, d_scores, sizeof(int) * H, hipMemcpyDeviceToHost));
  if (inlier_bits) RS_TRY(hipMemcpy(inlier_bits, d_bits, sizeof(unsigned long long) * (size_t)H * words, hipMemcpyDeviceToHost));
#undef RS_TRY
done:
  (void)hipFree(d_poses); (void)hipFree(d_obs); (void)hipFree(d_lines); (void)hipFree(d_scores); (void)hipFree(d_bits);
  return rc;
}


namespace {
template <typename T>
struct DevArr {
  T* p = nullptr;
  ~DevArr() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, sizeof(T) * (n ? n : 1)); }
};
}  // namespace

#define RM_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { std::fprintf(stderr, "slslam: %s failed: %s\n", #expr, hipGetErrorString(_e)); return SLSLAM_ERR_HIP; } } while (0)

extern "C" int slslam_ransac_generate(const slslam_ransac_trials* tr, double baseline, double* poses, int* valid) {
  if (!tr || !poses || !valid || tr->num_trials < 0 || tr->num_lines < 0 || tr->sample_size < 1 || tr->sample_size > 16)
    return SLSLAM_ERR_INVALID_ARGUMENT;
  const int H = tr->num_trials, K = tr->num_lines, s = tr->sample_size;
  if (H > 0 && (!tr->samples || !tr->observations0 || !tr->observations1 || K == 0)) return SLSLAM_ERR_INVALID_ARGUMENT;
  for (long long i = 0; i < (long long)H * s; ++i)
    if (tr->samples[i] < 0 || tr->samples[i] >= K) return SLSLAM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  if (H == 0) return SLSLAM_OK;
  DevArr<double> d_o0, d_o1, d_poses;
  DevArr<int> d_smp, d_valid;
  RM_TRY(d_o0.alloc(8 * (size_t)K)); RM_TRY(d_o1.alloc(8 * (size_t)K)); RM_TRY(d_poses.alloc(12 * (size_t)H));
  RM_TRY(d_smp.alloc((size_t)H * s)); RM_TRY(d_valid.alloc(H));
  RM_TRY(hipMemcpy(d_o0.p, tr->observations0, sizeof(double) * 8 * K, hipMemcpyHostToDevice));
  RM_TRY(hipMemcpy(d_o1.p, tr->observations1, sizeof(double) * 8 * K, hipMemcpyHostToDevice));
  RM_TRY(hipMemcpy(d_smp.p, tr->samples, sizeof(int) * (size_t)H * s, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_ransac_generate, dim3((unsigned)((H + 63) / 64)), dim3(64), 0, 0, H, s, d_smp.p, d_o0.p, d_o1.p, baseline,
                     d_poses.p, d_valid.p);
  RM_TRY(hipGetLastError());
  RM_TRY(hipMemcpy(poses, d_poses.p, sizeof(double) * 12 * H, hipMemcpyDeviceToHost));
  RM_TRY(hipMemcpy(valid, d_valid.p, sizeof(int) * H, hipMemcpyDeviceToHost));
  return SLSLAM_OK;
}

extern "C" int slslam_ransac_motion(const slslam_ransac_trials* tr, const double* lines, double baseline, double error_thr,
                                    double prob_free_outliers, int max_trials, int* best_score_io, int* trial_cnt,
                                    double* best_pose, unsigned long long* best_inlier_bits) {
  if (!tr || !best_score_io || !trial_cnt || !best_pose || tr->num_trials < 0 || tr->num_lines < 0 || tr->sample_size < 1 ||
      tr->sample_size > 16)
    return SLSLAM_ERR_INVALID_ARGUMENT;
  const int H = tr->num_trials, K = tr->num_lines, s = tr->sample_size;
  if (H > 0 && K > 0 && (!tr->samples || !tr->observations0 || !tr->observations1 || !lines)) return SLSLAM_ERR_INVALID_ARGUMENT;
  // comm_size == 0: the reference's trial loop never runs (ransac_trial = 0), whatever the sample array holds
  for (long long i = 0; K > 0 && i < (long long)H * s; ++i)
    if (tr->samples[i] < 0 || tr->samples[i] >= K) return SLSLAM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  *trial_cnt = 0;
  if (H == 0 || K == 0) return SLSLAM_OK;                     // ransac_trial = comm_size = 0: the loop body never runs
  const int words = (K + 63) / 64;
  DevArr<double> d_o0, d_o1, d_lines, d_poses;
  DevArr<int> d_smp, d_valid, d_scores;
  DevArr<unsigned long long> d_bits;
  RM_TRY(d_o0.alloc(8 * (size_t)K)); RM_TRY(d_o1.alloc(8 * (size_t)K)); RM_TRY(d_lines.alloc(6 * (size_t)K));
  RM_TRY(d_poses.alloc(12 * (size_t)H)); RM_TRY(d_smp.alloc((size_t)H * s)); RM_TRY(d_valid.alloc(H)); RM_TRY(d_scores.alloc(H));
  RM_TRY(d_bits.alloc((size_t)H * words));
  RM_TRY(hipMemcpy(d_o0.p, tr->observations0, sizeof(double) * 8 * K, hipMemcpyHostToDevice));
  RM_TRY(hipMemcpy(d_o1.p, tr->observations1, sizeof(double) * 8 * K, hipMemcpyHostToDevice));
  RM_TRY(hipMemcpy(d_lines.p, lines, sizeof(double) * 6 * K, hipMemcpyHostToDevice));
  RM_TRY(hipMemcpy(d_smp.p, tr->samples, sizeof(int) * (size_t)H * s, hipMemcpyHostToDevice));
  RM_TRY(hipMemset(d_scores.p, 0, sizeof(int) * H));
  // every pre-drawn trial at once: motion from its sample (the reference passes -baseline, slam.cpp:391-392) ...
  hipLaunchKernelGGL(k_ransac_generate, dim3((unsigned)((H + 63) / 64)), dim3(64), 0, 0, H, s, d_smp.p, d_o0.p, d_o1.p, -baseline,
                     d_poses.p, d_valid.p);
  // ... and its score against all common lines
  hipLaunchKernelGGL(k_ransac_score, dim3((unsigned)words, (unsigned)H), dim3(64), 0, 0, H, K, words, d_poses.p, d_o1.p, d_lines.p,
                     baseline, error_thr, d_scores.p, d_bits.p, (const int*)d_valid.p);
  RM_TRY(hipGetLastError());
  std::vector<int> scores(H);
  RM_TRY(hipMemcpy(scores.data(), d_scores.p, sizeof(int) * H, hipMemcpyDeviceToHost));
  // the adaptive trial loop of the reference (slam.cpp:363, :415-423), replayed in trial order over the scores
  int best = *best_score_io, best_h = -1, ransac_trial = K, t = 0;
  for (; t < ransac_trial && t <= max_trials && t < H; ++t) {
    if (scores[t] > best) {
      best = scores[t]; best_h = t;
      const double prob_s_outliers = 1 - std::pow(best / (double)K, s);
      ransac_trial = (int)(std::log(1 - prob_free_outliers) / std::log(std::min(1 - 1e-6, std::max(1e-6, prob_s_outliers))));
    }
  }
  *trial_cnt = t;
  *best_score_io = best;
  if (best_h >= 0) {
    RM_TRY(hipMemcpy(best_pose, d_poses.p + 12 * (size_t)best_h, sizeof(double) * 12, hipMemcpyDeviceToHost));
    if (best_inlier_bits)
      RM_TRY(hipMemcpy(best_inlier_bits, d_bits.p + (size_t)best_h * words, sizeof(unsigned long long) * words, hipMemcpyDeviceToHost));
  }
  return SLSLAM_OK;
}


// Many frames at once (replay of a sequence, several cameras): one device allocation, one upload, two launches
// per frame enqueued back to back without host synchronisation, one download of every frame's scores, then the
// per-frame trial loops on the host and one gather of the winners.  Per frame the results are those of
// slslam_ransac_motion.
extern "C" int slslam_ransac_motion_batch(int num_frames, const slslam_ransac_trials* frames, const double* const* lines,
                                          double baseline, double error_thr, double prob_free_outliers, int max_trials,
                                          int* best_score_io, int* trial_cnt, double* best_pose,
                                          unsigned long long* const* best_inlier_bits) {
  if (num_frames < 0 || (num_frames > 0 && (!frames || !lines || !best_score_io || !trial_cnt || !best_pose)))
    return SLSLAM_ERR_INVALID_ARGUMENT;
  struct Off { size_t o0, o1, ln, smp, poses, valid, scores, bits; int H, K, s, words; };
  std::vector<Off> off(num_frames);
  size_t nd = 0, ni = 0, nb = 0;       // doubles, ints, 64-bit words
  for (int f = 0; f < num_frames; ++f) {
    const slslam_ransac_trials& tr = frames[f];
    if (tr.num_trials < 0 || tr.num_lines < 0 || tr.sample_size < 1 || tr.sample_size > 16) return SLSLAM_ERR_INVALID_ARGUMENT;
    const int H = tr.num_trials, K = tr.num_lines, s = tr.sample_size;
    if (H > 0 && K > 0 && (!tr.samples || !tr.observations0 || !tr.observations1 || !lines[f])) return SLSLAM_ERR_INVALID_ARGUMENT;
    for (long long i = 0; K > 0 && i < (long long)H * s; ++i)      // a frame without common lines runs no trials
      if (tr.samples[i] < 0 || tr.samples[i] >= K) return SLSLAM_ERR_INVALID_ARGUMENT;
    Off& o = off[f];
    o.H = H; o.K = K; o.s = s; o.words = (K + 63) / 64;
    o.o0 = nd; nd += 8 * (size_t)K; o.o1 = nd; nd += 8 * (size_t)K; o.ln = nd; nd += 6 * (size_t)K; o.poses = nd; nd += 12 * (size_t)H;
    o.smp = ni; ni += (size_t)H * s; o.valid = ni; ni += H; o.scores = ni; ni += H;
    o.bits = nb; nb += (size_t)H * o.words;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  for (int f = 0; f < num_frames; ++f) trial_cnt[f] = 0;
  if (num_frames == 0) return SLSLAM_OK;
  std::vector<double> hd(nd ? nd : 1, 0.0);
  std::vector<int> hi(ni ? ni : 1, 0);
  for (int f = 0; f < num_frames; ++f) {
    const slslam_ransac_trials& tr = frames[f];
    const Off& o = off[f];
    if (o.H == 0 || o.K == 0) continue;
    std::copy(tr.observations0, tr.observations0 + 8 * (size_t)o.K, hd.begin() + o.o0);
    std::copy(tr.observations1, tr.observations1 + 8 * (size_t)o.K, hd.begin() + o.o1);
    std::copy(lines[f], lines[f] + 6 * (size_t)o.K, hd.begin() + o.ln);
    std::copy(tr.samples, tr.samples + (size_t)o.H * o.s, hi.begin() + o.smp);
  }
  DevArr<double> dd;
  DevArr<int> di;
  DevArr<unsigned long long> db;
  RM_TRY(dd.alloc(nd)); RM_TRY(di.alloc(ni)); RM_TRY(db.alloc(nb));
  RM_TRY(hipMemcpy(dd.p, hd.data(), sizeof(double) * hd.size(), hipMemcpyHostToDevice));
  RM_TRY(hipMemcpy(di.p, hi.data(), sizeof(int) * hi.size(), hipMemcpyHostToDevice));     // scores arrive zeroed
  for (int f = 0; f < num_frames; ++f) {
    const Off& o = off[f];
    if (o.H == 0 || o.K == 0) continue;
    hipLaunchKernelGGL(k_ransac_generate, dim3((unsigned)((o.H + 63) / 64)), dim3(64), 0, 0, o.H, o.s, di.p + o.smp, dd.p + o.o0,
                       dd.p + o.o1, -baseline, dd.p + o.poses, di.p + o.valid);
    hipLaunchKernelGGL(k_ransac_score, dim3((unsigned)o.words, (unsigned)o.H), dim3(64), 0, 0, o.H, o.K, o.words, dd.p + o.poses,
                       dd.p + o.o1, dd.p + o.ln, baseline, error_thr, di.p + o.scores, db.p + o.bits, (const int*)(di.p + o.valid));
  }
  RM_TRY(hipGetLastError());
  RM_TRY(hipMemcpy(hi.data(), di.p, sizeof(int) * hi.size(), hipMemcpyDeviceToHost));
  for (int f = 0; f < num_frames; ++f) {
    const Off& o = off[f];
    if (o.H == 0 || o.K == 0) continue;
    const int* scores = hi.data() + o.scores;
    int best = best_score_io[f], best_h = -1, ransac_trial = o.K, t = 0;
    for (; t < ransac_trial && t <= max_trials && t < o.H; ++t) {
      if (scores[t] > best) {
        best = scores[t]; best_h = t;
        const double prob_s_outliers = 1 - std::pow(best / (double)o.K, o.s);
        ransac_trial = (int)(std::log(1 - prob_free_outliers) / std::log(std::min(1 - 1e-6, std::max(1e-6, prob_s_outliers))));
      }
    }
    trial_cnt[f] = t;
    best_score_io[f] = best;
    if (best_h >= 0) {
      RM_TRY(hipMemcpy(best_pose + 12 * (size_t)f, dd.p + o.poses + 12 * (size_t)best_h, sizeof(double) * 12, hipMemcpyDeviceToHost));
      if (best_inlier_bits && best_inlier_bits[f])
        RM_TRY(hipMemcpy(best_inlier_bits[f], db.p + o.bits + (size_t)best_h * o.words, sizeof(unsigned long long) * o.words,
                         hipMemcpyDeviceToHost));
    }
  }
  return SLSLAM_OK;
}
