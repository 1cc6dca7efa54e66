// slslam_amd/csrc/lba_gram.h — the normal-equation blocks of one stereo line observation WITHOUT forming its Jacobians.
//
// What it replaces: the same arithmetic as lba_math.h::obs_linearise + the block products of the elimination sweep
// (reference: LineReprojectionError through AutoDiffCostFunction<...,4,6,4>, src/lba_problem.h:46-118,
// src/lba_problem.cpp:65-74, and the J^T J Ceres forms from it), factored for registers.  Every row of both Jacobians is
// a fixed linear map of the row's six "gradient" numbers g = (gP, gD) = (dr/dP, dr/d dc) (P: the line's closest point, dc:
// its direction, camera frame):
//     J_cam row  = [ tau | gP ] T_c,     tau = Q x gP + dc x gD,  Q = R cp          (T_c: camera constants, below)
//     J_line row = Ml g,                 Ml 4x6 from the line's frame in camera coordinates
// so everything the sweep needs is a product with the 6x6 Gram matrix  W = sum_rows g g^T  and  w = sum_rows g r :
//     J_l^T J_l = Ml W Ml^T,  J_l^T r = Ml w,  J_c'^T J_l = Mc (W Ml^T),  J_c'^T J_c' = Mc W Mc^T,  J_c'^T r = Mc w
// with Mc v = (Q x vP + dc x vD, vP).  A lane keeps W (21) + w (6) + a 15-number frame instead of 44 Jacobian entries,
// the Huber weight multiplies 27 numbers instead of 44, and the camera-side constants T_c = diag(JL(w) diag(s_w), diag(s_t))
// (left Jacobian of SO(3) and the Jacobi column scale) are applied once per window by the reduced solve, as a
// congruence of the reduced system, instead of once per observation.
//
// Host-compilable (tests/host_math cross-checks it against obs_linearise and the oracle's dual numbers).
#ifndef SLSLAM_LBA_GRAM_H_
#define SLSLAM_LBA_GRAM_H_

#include "lba_math.h"

namespace slslam {

SLS_HD constexpr int gtri(int i, int j) { return (i * (i + 1)) / 2 + j; }   // i >= j

template <typename T>
SLS_HD void cross3(const T a[3], const T b[3], T o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// Residuals r[4] and the Gram data of the four rows: W (lower triangle of sum g g^T, g = (gP, gD)), w = sum g r.
// Also returns what the later stages need of the geometry: dc = R dv, e2 = R col2 (cp = -d col2), P = R cp + t.
// No Huber weight, no scaling: the caller multiplies W and w by rho'.
template <typename T>
SLS_HD void obs_gram(const T R[9], const T t[3], const T trig[7], const T ob[8], T baseline,
                     T dc[3], T e2[3], T P[3], T r[4], T W[21], T w[6]) {
  const T s1 = trig[0], c1 = trig[1], s2 = trig[2], c2 = trig[3], s3 = trig[4], c3 = trig[5], d = trig[6];
  const T col2[3] = { c1 * s2 * c3 + s1 * s3, c1 * s2 * s3 - s1 * c3, c1 * c2 };
  const T dv[3] = { s1 * s2 * c3 - c1 * s3, s1 * s2 * s3 + c1 * c3, s1 * c2 };
  for (int i = 0; i < 3; ++i) {
    e2[i] = R[3 * i] * col2[0] + R[3 * i + 1] * col2[1] + R[3 * i + 2] * col2[2];
    dc[i] = R[3 * i] * dv[0] + R[3 * i + 1] * dv[1] + R[3 * i + 2] * dv[2];
    P[i] = t[i] - d * e2[i];
  }
  for (int q = 0; q < 21; ++q) W[q] = T(0);
  for (int q = 0; q < 6; ++q) w[q] = T(0);
  T Pk[3] = { P[0], P[1], P[2] };
  for (int k = 0; k < 2; ++k) {
    if (k == 1) Pk[0] -= baseline;
    const T n0 = Pk[1] * dc[2] - Pk[2] * dc[1];
    const T n1 = Pk[2] * dc[0] - Pk[0] * dc[2];
    const T n2 = Pk[0] * dc[1] - Pk[1] * dc[0];
    const T is = inv_sqrt<T>(n0 * n0 + n1 * n1);
    const T m0 = n0 * is, m1 = n1 * is, m2 = n2 * is;
    for (int e = 0; e < 2; ++e) {
      const T x = ob[4 * k + 2 * e], y = ob[4 * k + 2 * e + 1];
      const T rho = x * m0 + y * m1 + m2;
      r[2 * k + e] = -rho;
      const T q[3] = { -(x - rho * m0) * is, -(y - rho * m1) * is, -is };
      T g[6];
      cross3<T>(dc, q, g);          // gP = dc x q
      cross3<T>(q, Pk, g + 3);      // gD = q x P_k
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j <= i; ++j) W[gtri(i, j)] += g[i] * g[j];
        w[i] -= g[i] * rho;
      }
    }
  }
}

// Mc v = (Q x vP + dc x vD, vP): one row of the camera Jacobian (before T_c) applied to a vector of g-space.
template <typename T>
SLS_HD void apply_mc(const T Q[3], const T dc[3], const T v[6], T out[6]) {
  T a[3], b[3];
  cross3<T>(Q, v, a);
  cross3<T>(dc, v + 3, b);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2];
  out[3] = v[0]; out[4] = v[1]; out[5] = v[2];
}

// D = Mc W Mc^T (lower triangle over (w0,w1,w2,t0,t1,t2), packed like tri_index: 21 values).
template <typename T>
SLS_HD void gram_camera_block(const T W[21], const T Q[3], const T dc[3], T D[21]) {
  // W = [[A, B], [B^T, C]]: A = sum gP gP^T, B = sum gP gD^T, C = sum gD gD^T
  T Dwt[9], E[9];                                   // Dwt = sum tau gP^T = [Q]x A + [dc]x B^T ; E = sum tau gD^T = [Q]x B + [dc]x C
#define SLS_WS(i, j) W[(i) >= (j) ? gtri((i), (j)) : gtri((j), (i))]
  for (int j = 0; j < 3; ++j) {
    const T Acol[3] = { SLS_WS(0, j), SLS_WS(1, j), SLS_WS(2, j) };
    const T Btcol[3] = { SLS_WS(3, j), SLS_WS(4, j), SLS_WS(5, j) };             // (sum gD gP_j): B[j][0..2]
    T a[3], b[3];
    cross3<T>(Q, Acol, a);
    cross3<T>(dc, Btcol, b);
    for (int i = 0; i < 3; ++i) Dwt[3 * i + j] = a[i] + b[i];
    const T Bcol[3] = { SLS_WS(3 + j, 0), SLS_WS(3 + j, 1), SLS_WS(3 + j, 2) };   // (sum gP gD_j): B[0..2][j]
    const T Ccol[3] = { SLS_WS(3, 3 + j), SLS_WS(4, 3 + j), SLS_WS(5, 3 + j) };
    cross3<T>(Q, Bcol, a);
    cross3<T>(dc, Ccol, b);
    for (int i = 0; i < 3; ++i) E[3 * i + j] = a[i] + b[i];
  }
#undef SLS_WS
  // Dww row i = Q x Dwt[i,:] + dc x E[i,:]   (sum tau tau^T = Dwt [Q]x^T + E [dc]x^T)
  for (int i = 0; i < 3; ++i) {
    T a[3], b[3];
    cross3<T>(Q, Dwt + 3 * i, a);
    cross3<T>(dc, E + 3 * i, b);
    for (int j = 0; j <= i; ++j) D[gtri(i, j)] = a[j] + b[j];
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) D[gtri(3 + i, j)] = Dwt[3 * j + i];       // D[t_i][w_j] = Dwt[j][i]
    for (int j = 0; j <= i; ++j) D[gtri(3 + i, 3 + j)] = W[gtri(i, j)];
  }
}

// The four rows of the line Jacobian as maps of g-space, Ml[6 j .. 6 j + 5] = (coefficients of gP | of gD), already
// multiplied by the Jacobi scale sl[j] of the line's column j.  zc = R e_z (third column of R).
//   a: d dc.gP + e2.gD          b: (s1 gD - d c1 gP).e0, e0 = dc x e2
//   g: -d (zc x e2).gP + (zc x dc).gD          t: (1 + d^2) e2.gP
template <typename T>
SLS_HD void line_rows(const T dc[3], const T e2[3], const T zc[3], T d, T c1, T s1, const T sl[4], T Ml[24]) {
  T e0[3], za[3], zb[3];
  cross3<T>(dc, e2, e0);
  cross3<T>(zc, e2, za);
  cross3<T>(zc, dc, zb);
  const T a0 = d * sl[0], b1 = -d * c1 * sl[1], b1d = s1 * sl[1], g2 = -d * sl[2], t3 = (T(1) + d * d) * sl[3];
  for (int i = 0; i < 3; ++i) {
    Ml[i] = a0 * dc[i];        Ml[3 + i] = sl[0] * e2[i];
    Ml[6 + i] = b1 * e0[i];    Ml[9 + i] = b1d * e0[i];
    Ml[12 + i] = g2 * za[i];   Ml[15 + i] = sl[2] * zb[i];
    Ml[18 + i] = t3 * e2[i];   Ml[21 + i] = T(0);
  }
}

// Y_j = W Ml_j^T (6 values each, Y[6 j + i]),  H = Ml W Ml^T (lower triangle, packed as chol4_inverse expects),  gl = Ml w.
template <typename T>
SLS_HD void gram_line(const T W[21], const T w[6], const T Ml[24], T Y[24], T H[10], T gl[4]) {
  for (int j = 0; j < 4; ++j) {
    const T* m = Ml + 6 * j;
    for (int i = 0; i < 6; ++i) {
      T s = T(0);
      for (int q = 0; q < (j == 3 ? 3 : 6); ++q) s += W[q <= i ? gtri(i, q) : gtri(q, i)] * m[q];
      Y[6 * j + i] = s;
    }
    for (int b = 0; b <= j; ++b) {
      const T* mb = Ml + 6 * b;
      T s = T(0);
      for (int q = 0; q < (b == 3 ? 3 : 6); ++q) s += Y[6 * j + q] * mb[q];
      H[gtri(j, b)] = s;
    }
    T s = T(0);
    for (int q = 0; q < (j == 3 ? 3 : 6); ++q) s += m[q] * w[q];
    gl[j] = s;
  }
}

}  // namespace slslam
#endif  // SLSLAM_LBA_GRAM_H_
