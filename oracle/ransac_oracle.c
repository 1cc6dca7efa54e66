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

/* ---------------------------------------------------------------------------------------------
 * Hypothesis generation and the adaptive trial loop.
 *   SLAM::vo_angle_axis_approx   reference src/slam.cpp:433-574
 *   SLAM::ransac_motion          reference src/slam.cpp:322-427
 * Eigen's dynamic-size (At*A).inverse() is a partially pivoted LU; restated as such. */

/* ceres::AngleAxisToRotationMatrix as reached through gc_Rodriguez (gc.cpp:24-35); row-major out */
static void aa_to_matrix(const double w[3], double R[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2), wx = w[0] / th, wy = w[1] / th, wz = w[2] / th;
    const double c = cos(th), s = sin(th);
    R[0] = c + wx * wx * (1 - c);      R[1] = wx * wy * (1 - c) - wz * s; R[2] = wy * s + wx * wz * (1 - c);
    R[3] = wz * s + wx * wy * (1 - c); R[4] = c + wy * wy * (1 - c);      R[5] = -wx * s + wy * wz * (1 - c);
    R[6] = -wy * s + wx * wz * (1 - c); R[7] = wx * s + wy * wz * (1 - c); R[8] = c + wz * wz * (1 - c);
  } else {
    R[0] = 1; R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2]; R[4] = 1; R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0]; R[8] = 1;
  }
}

static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void image_line(const double* ob, double l[3]) {          /* p1 x p2 with p = (x, y, 1) */
  const double p1[3] = { ob[0], ob[1], 1 }, p2[3] = { ob[2], ob[3], 1 };
  cross3(p1, p2, l);
}

/* x = (A^T A)^-1 A^T rhs for the 3 unknown columns accumulated as N = A^T A (3x3), v = A^T rhs:
 * inverse by LU with partial pivoting (what MatrixXd::inverse() does), then the product. */
static void solve_normal3(const double N[9], const double v[3], double x[3]) {
  double a[3][6];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { a[i][j] = N[3 * i + j]; a[i][3 + j] = i == j ? 1.0 : 0.0; }
  for (int c = 0; c < 3; ++c) {
    int piv = c;
    for (int r = c + 1; r < 3; ++r) if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
    if (piv != c) for (int j = 0; j < 6; ++j) { const double t = a[c][j]; a[c][j] = a[piv][j]; a[piv][j] = t; }
    for (int r = c + 1; r < 3; ++r) {
      const double f = a[r][c] / a[c][c];
      for (int j = c; j < 6; ++j) a[r][j] -= f * a[c][j];
    }
  }
  double inv[3][3];
  for (int j = 0; j < 3; ++j)
    for (int i = 2; i >= 0; --i) {
      double s = a[i][3 + j];
      for (int k = i + 1; k < 3; ++k) s -= a[i][k] * inv[k][j];
      inv[i][j] = s / a[i][i];
    }
  for (int i = 0; i < 3; ++i) x[i] = inv[i][0] * v[0] + inv[i][1] * v[1] + inv[i][2] * v[2];
}

int oracle_vo_angle_axis_approx(int nfeat, const double* obs0, const double* obs1, double baseline, double pose[12]) {
  double N[9] = { 0 }, v[3] = { 0 };
  /* rotation: rows of K, slam.cpp:437-482; w = -(A^T A)^-1 A^T b with b = -K.col(3) (:484-488) */
  for (int i = 0; i < nfeat; ++i) {
    double l1[3], l2[3], l3[3], l4[3], lx[3];
    image_line(obs0 + 8 * i, l1); image_line(obs0 + 8 * i + 4, l2);
    image_line(obs1 + 8 * i, l3); image_line(obs1 + 8 * i + 4, l4);
    cross3(l1, l2, lx);
    const double lxn = norm3(lx);
    if (lxn == 0) return 0;
    lx[0] /= lxn; lx[1] /= lxn; lx[2] /= lxn;
    for (int j = 0; j < 2; ++j) {
      const double* tl = j == 0 ? l3 : l4;
      const double tln = norm3(tl);
      if (tln == 0) return 0;
      const double ly[3] = { tl[0] / tln, tl[1] / tln, tl[2] / tln };
      const double row[4] = { lx[2] * ly[1] - lx[1] * ly[2], lx[0] * ly[2] - lx[2] * ly[0], lx[1] * ly[0] - lx[0] * ly[1],
                              lx[0] * ly[0] + lx[1] * ly[1] + lx[2] * ly[2] };
      for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) N[3 * a + b] += row[a] * row[b];
        v[a] += row[a] * (-row[3]);
      }
    }
  }
  double w[3], R[9];
  solve_normal3(N, v, w);
  w[0] = -w[0]; w[1] = -w[1]; w[2] = -w[2];
  aa_to_matrix(w, R);

  /* translation: rows of M, slam.cpp:490-559; t = (A^T A)^-1 A^T b with b = -M.col(3) (:561-565) */
  for (int a = 0; a < 9; ++a) N[a] = 0;
  v[0] = v[1] = v[2] = 0;
  for (int i = 0; i < nfeat; ++i) {
    double l1[3], l2[3], lx[3];
    image_line(obs0 + 8 * i, l1);
    const double l1n = norm3(l1);
    if (l1n == 0) return 0;
    l1[0] /= l1n; l1[1] /= l1n; l1[2] /= l1n;
    image_line(obs0 + 8 * i + 4, l2);
    const double l2n = norm3(l2);
    if (l2n == 0) return 0;
    l2[0] /= l2n; l2[1] /= l2n; l2[2] /= l2n;
    cross3(l1, l2, lx);
    if (norm3(lx) == 0) return 0;
    for (int j = 0; j < 2; ++j) {
      double l3[3];
      image_line(obs1 + 8 * i + 4 * j, l3);
      const double l3n = norm3(l3);
      if (l3n == 0) return 0;
      l3[0] /= l3n; l3[1] /= l3n; l3[2] /= l3n;
      /* c_k = -l2^T (a4 R.col(k)^T) l3 [+ l2(k) B l3(0) for the right image], a4 = (B, 0, 0) */
      double c[3];
      for (int k = 0; k < 3; ++k) {
        const double rc[3] = { R[k], R[3 + k], R[6 + k] };          /* R.col(k) */
        double u[3];
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
      for (int r = 0; r < 3; ++r)
        for (int a = 0; a < 3; ++a) {
          for (int b = 0; b < 3; ++b) N[3 * a + b] += rows[r][a] * rows[r][b];
          v[a] += rows[r][a] * (-rows[r][3]);
        }
    }
  }
  double t[3];
  solve_normal3(N, v, t);
  for (int q = 0; q < 9; ++q) pose[q] = R[q];       /* T[0] = gc_wt_to_Rt(wt): R = Rodrigues(w), :567-571 */
  pose[9] = t[0]; pose[10] = t[1]; pose[11] = t[2];
  return 1;
}

/* SLAM::ransac_motion (slam.cpp:322-427) for a given sequence of samples (the reference draws
 * them with rand.rand_sample, one per trial).  samples: [max_draws][s]; returns the number of trials
 * executed (trial_cnt), best score, best pose and the inlier flags of the best hypothesis. */
int oracle_ransac_motion(int K, const double* obs0, const double* obs1, const double* lines, int s, int max_draws,
                         const int* samples, double baseline, double thr, double prob_free_outliers, int max_trials,
                         int* best_score_io, double best_pose[12], unsigned char* best_inliers) {
  int best_score = *best_score_io, trial_cnt = 0, ransac_trial = K;        /* :330-332 */
  double o0[8 * 16], o1[8 * 16];
  for (; trial_cnt < ransac_trial && trial_cnt <= max_trials && trial_cnt < max_draws; ++trial_cnt) {   /* :363 */
    const int* smp = samples + (long)trial_cnt * s;
    for (int j = 0; j < s; ++j)
      for (int q = 0; q < 8; ++q) { o0[8 * j + q] = obs0[8 * (long)smp[j] + q]; o1[8 * j + q] = obs1[8 * (long)smp[j] + q]; }
    double T[12];
    if (!oracle_vo_angle_axis_approx(s, o0, o1, -baseline, T)) continue;   /* :391-394 */
    if (sqrt(T[9] * T[9] + T[10] * T[10] + T[11] * T[11]) > 1.0) continue;  /* :398-399 */
    int score = 0;
    for (int k = 0; k < K; ++k)
      score += oracle_reprojection_error(obs1 + 8 * (long)k, T, T + 9, lines + 6 * (long)k, baseline) < thr;
    if (score > best_score) {                                              /* :415-423 */
      best_score = score;
      for (int q = 0; q < 12; ++q) best_pose[q] = T[q];
      if (best_inliers)
        for (int k = 0; k < K; ++k)
          best_inliers[k] = oracle_reprojection_error(obs1 + 8 * (long)k, T, T + 9, lines + 6 * (long)k, baseline) < thr;
      const double prob_s_outliers = 1 - pow(score / (double)K, s);
      const double den = log(fmin(1 - 1e-6, fmax(1e-6, prob_s_outliers)));
      ransac_trial = (int)(log(1 - prob_free_outliers) / den);
    }
  }
  *best_score_io = best_score;
  return trial_cnt;
}
