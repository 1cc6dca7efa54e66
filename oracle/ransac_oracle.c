/* oracle/ransac_oracle.c — CPU restatement of the RANSAC scoring loop, TEST INFRASTRUCTURE ONLY.
 *   SLAM::reprojection_error   reference src/slam.cpp:691-726
 *   scoring loop               reference src/slam.cpp:396-413 (skip |t| > 1, count error < error_thr)
 * The reference mixes precisions: `float sql`, `float error`, a float return value compared with the
 * double threshold; every conversion is kept where the reference has it. */
#include <math.h>
#include "slslam_oracle.h"

float oracle_reprojection_error(const double ft[8], const double R[9], const double t_in[3], const double line[6], double baseline) {
  float error = 0;                                                     /* :693 */
  const double* cp = line; const double* dv = line + 3;                /* :696-697 */
  double t[3] = { t_in[0], t_in[1], t_in[2] };
  for (int i = 0; i < 2; ++i) {                                        /* :703 */
    double p1[3], p2[3];
    if (i == 0) { p1[0] = ft[0]; p1[1] = ft[1]; p1[2] = 1; p2[0] = ft[2]; p2[1] = ft[3]; p2[2] = 1; }
    else { t[0] -= baseline; p1[0] = ft[4]; p1[1] = ft[5]; p1[2] = 1; p2[0] = ft[6]; p2[1] = ft[7]; p2[2] = 1; }   /* :708 */
    double cpc[3], dvc[3], nc[3];
    for (int r = 0; r < 3; ++r) {
      cpc[r] = R[3 * r] * cp[0] + R[3 * r + 1] * cp[1] + R[3 * r + 2] * cp[2] + t[r];    /* gc_point_to_pose, gc.cpp:55-57 */
      dvc[r] = R[3 * r] * dv[0] + R[3 * r + 1] * dv[1] + R[3 * r + 2] * dv[2];
    }
    nc[0] = cpc[1] * dvc[2] - cpc[2] * dvc[1];                         /* :716 */
    nc[1] = cpc[2] * dvc[0] - cpc[0] * dvc[2];
    nc[2] = cpc[0] * dvc[1] - cpc[1] * dvc[0];
    const float sql = (float)sqrt(nc[0] * nc[0] + nc[1] * nc[1]);       /* :718 */
    nc[0] /= sql; nc[1] /= sql; nc[2] /= sql;                          /* :719 */
    error += fabs(nc[0] * p1[0] + nc[1] * p1[1] + nc[2] * p1[2]);      /* :721 */
    error += fabs(nc[0] * p2[0] + nc[1] * p2[1] + nc[2] * p2[2]);      /* :722 */
  }
  return error / 4.0;                                                  /* :725 */
}

void oracle_ransac_score(int H, const double* poses, int K, const double* obs, const double* lines,
                         double baseline, double thr, int* scores, unsigned char* inliers) {
  for (int h = 0; h < H; ++h) {
    const double* T = poses + 12 * (long)h;
    if (sqrt(T[9] * T[9] + T[10] * T[10] + T[11] * T[11]) > 1.0) {     /* slam.cpp:398-399 */
      scores[h] = -1;
      if (inliers) for (int k = 0; k < K; ++k) inliers[(long)h * K + k] = 0;
      continue;
    }
    int score = 0;
    for (int k = 0; k < K; ++k) {
      const float error = oracle_reprojection_error(obs + 8 * (long)k, T, T + 9, lines + 6 * (long)k, baseline);
      const int in = error < thr;                                      /* :404 */
      score += in;
      if (inliers) inliers[(long)h * K + k] = (unsigned char)in;
    }
    scores[h] = score;
  }
}
