// slslam_amd/csrc/lba_math.h — per-observation arithmetic of the line bundle adjustment,
// shared by every LBA kernel (and compiled for the host by tests/ to cross-check it against the
// oracle's dual-number Jacobians without a GPU).
//
// What it replaces: the reference evaluates `LineReprojectionError::operator()<Jet<double,10>>`
// (reference src/lba_problem.h:46-118) through ceres::AutoDiffCostFunction<...,4,6,4>
// (src/lba_problem.cpp:65-74) once per observation.  Here the same residual is differentiated
// analytically and factored so that the expensive pieces are computed once per owner instead of
// once per observation:
//   per camera  : R(w) and the SO(3) left Jacobian JL(w) (cam_prepare)        — 1 sincos pair
//   per line    : sin/cos table of (a,b,g) and cot(t)   (line_trig)          — the only other trig
//   per obs     : 4 residuals, J_cam 4x6, J_line 4x4    (obs_linearise)      — FMAs, 2 rsqrt
//
// Derivation (SURVEY.md 8a "Analytic-Jacobian recipe"): with P_k = R cp + t - k B e_x,
// dc = R dv, n = P_k x dc, s = sqrt(n0^2+n1^2), m = n/s, r = -(x m0 + y m1 + m2):
//   q  = dr/dn  = -([x y 1] - rho [m0 m1 0]) / s,  rho = -r
//   gP = dr/dP  = dc x q,     gD = dr/ddc = q x P
//   dr/dt = gP;  dr/dw = ((R cp) x gP + dc x gD)^T JL(w)   [d(R p)/dw = -[R p]x JL];
//   dr/du_j = (R^T gP).dcp_j + (R^T gD).ddv_j
#ifndef SLSLAM_LBA_MATH_H_
#define SLSLAM_LBA_MATH_H_

#include <math.h>

#if defined(__HIPCC__)
#define SLS_HD __host__ __device__ __forceinline__
#else
#define SLS_HD inline
#endif

namespace slslam {

// 1/sqrt(x): one v_rsq_f64 + refinement on the device instead of a full-precision sqrt followed by a
// full-precision divide (~30 fp64 instructions).
template <typename T>
SLS_HD T inv_sqrt(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(x);
#else
  return T(1) / sqrt(x);
#endif
}

// Rotation matrix (row-major) of the angle-axis w and the left Jacobian of SO(3),
//   d(R(w) p)/dw = -[R p]x JL(w),   JL = (sin t/t) I + ((1-cos t)/t) [u]x + (1 - sin t/t) u u^T.
// Value follows ceres::AngleAxisRotatePoint as used at lba_problem.h:75-76:
// R = c I + s [u]x + (1-c) u u^T for theta > 0 and the first-order branch I + [w]x at theta == 0
// (hit by every solve: the newest keyframe is exactly identity, slam.cpp:1322), where JL = I.
template <typename T>
SLS_HD void cam_prepare(const T w[3], T R[9], T JL[9]) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > T(0)) {
    const T th = sqrt(th2);
    const T ith = T(1) / th;
    const T u[3] = { w[0] * ith, w[1] * ith, w[2] * ith };
    const T s = sin(th), c = cos(th);
    const T sh = sin(T(0.5) * th);
    const T omc = T(2) * sh * sh;                       // 1 - cos(theta) without cancellation
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[3 * i + j] = omc * u[i] * u[j] + (i == j ? c : T(0));
    R[1] += -s * u[2]; R[2] += s * u[1];
    R[3] += s * u[2];  R[5] += -s * u[0];
    R[6] += -s * u[1]; R[7] += s * u[0];
    const T sot = s * ith, oot = omc * ith, rem = T(1) - sot;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) JL[3 * i + j] = rem * u[i] * u[j] + (i == j ? sot : T(0));
    JL[1] += -oot * u[2]; JL[2] += oot * u[1];
    JL[3] += oot * u[2];  JL[5] += -oot * u[0];
    JL[6] += -oot * u[1]; JL[7] += oot * u[0];
  } else {
    R[0] = T(1); R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2]; R[4] = T(1); R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0]; R[8] = T(1);
    for (int k = 0; k < 9; ++k) JL[k] = T(0);
    JL[0] = JL[4] = JL[8] = T(1);
  }
}

// Rotation only (candidate-cost evaluation needs no derivative).
template <typename T>
SLS_HD void cam_rotation(const T w[3], T R[9]) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (th2 > T(0)) {
    const T th = sqrt(th2);
    const T ith = T(1) / th;
    const T u[3] = { w[0] * ith, w[1] * ith, w[2] * ith };
    const T s = sin(th), c = cos(th);
    const T sh = sin(T(0.5) * th);
    const T omc = T(2) * sh * sh;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[3 * i + j] = omc * u[i] * u[j] + (i == j ? c : T(0));
    R[1] += -s * u[2]; R[2] += s * u[1];
    R[3] += s * u[2];  R[5] += -s * u[0];
    R[6] += -s * u[1]; R[7] += s * u[0];
  } else {
    R[0] = T(1); R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2]; R[4] = T(1); R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0]; R[8] = T(1);
  }
}

// sin/cos table of the orthonormal line parameters (lba_problem.h:56-63):
// trig = {s1,c1,s2,c2,s3,c3,d} with d = cos(t)/sin(t).
template <typename T>
SLS_HD void line_trig(const T u[4], T trig[7]) {
  trig[0] = sin(u[0]); trig[1] = cos(u[0]);
  trig[2] = sin(u[1]); trig[3] = cos(u[1]);
  trig[4] = sin(u[2]); trig[5] = cos(u[2]);
  trig[6] = cos(u[3]) / sin(u[3]);
}

// Closest point cp = -d col2 and direction dv = col1 of R_l = Rz(g) Ry(b) Rx(a)
// (lba_problem.h:66-72), from the trig table.
template <typename T>
SLS_HD void line_points(const T trig[7], T cp[3], T dv[3]) {
  const T s1 = trig[0], c1 = trig[1], s2 = trig[2], c2 = trig[3], s3 = trig[4], c3 = trig[5], d = trig[6];
  cp[0] = -(c1 * s2 * c3 + s1 * s3) * d;
  cp[1] = -(c1 * s2 * s3 - s1 * c3) * d;
  cp[2] = -(c1 * c2) * d;
  dv[0] = s1 * s2 * c3 - c1 * s3;
  dv[1] = s1 * s2 * s3 + c1 * c3;
  dv[2] = s1 * c2;
}

// cp, dv and their partials w.r.t. (a,b,g,t): dcp[j] = d cp / d u_j, ddv[j] = d dv / d u_j
// (ddv[3] == 0 and is not stored).
template <typename T>
SLS_HD void line_points_jac(const T trig[7], T cp[3], T dv[3], T dcp[12], T ddv[9]) {
  const T s1 = trig[0], c1 = trig[1], s2 = trig[2], c2 = trig[3], s3 = trig[4], c3 = trig[5], d = trig[6];
  const T col0[3] = { c2 * c3, c2 * s3, -s2 };
  const T col2[3] = { c1 * s2 * c3 + s1 * s3, c1 * s2 * s3 - s1 * c3, c1 * c2 };
  dv[0] = s1 * s2 * c3 - c1 * s3;
  dv[1] = s1 * s2 * s3 + c1 * c3;
  dv[2] = s1 * c2;
  for (int i = 0; i < 3; ++i) cp[i] = -d * col2[i];
  // a: d col1 = col2, d col2 = -col1
  for (int i = 0; i < 3; ++i) { dcp[i] = d * dv[i]; ddv[i] = col2[i]; }
  // b: d col1 = s1 col0, d col2 = c1 col0
  for (int i = 0; i < 3; ++i) { dcp[3 + i] = -d * c1 * col0[i]; ddv[3 + i] = s1 * col0[i]; }
  // g: d col = e_z x col
  dcp[6] = d * col2[1]; dcp[7] = -d * col2[0]; dcp[8] = T(0);
  ddv[6] = -dv[1];      ddv[7] = dv[0];        ddv[8] = T(0);
  // t: d d/dt = -(1 + d^2)
  const T dd = T(1) + d * d;
  for (int i = 0; i < 3; ++i) dcp[9 + i] = dd * col2[i];
}

// 4 residuals of one stereo observation (lba_problem.h:75-115); no derivatives.
template <typename T>
SLS_HD void obs_residual(const T R[9], const T t[3], const T cp[3], const T dv[3],
                         const T ob[8], T baseline, T r[4]) {
  T P[3], dc[3];
  for (int i = 0; i < 3; ++i) {
    P[i] = R[3 * i] * cp[0] + R[3 * i + 1] * cp[1] + R[3 * i + 2] * cp[2] + t[i];
    dc[i] = R[3 * i] * dv[0] + R[3 * i + 1] * dv[1] + R[3 * i + 2] * dv[2];
  }
  for (int k = 0; k < 2; ++k) {
    if (k == 1) P[0] -= baseline;
    const T n0 = P[1] * dc[2] - P[2] * dc[1];
    const T n1 = P[2] * dc[0] - P[0] * dc[2];
    const T n2 = P[0] * dc[1] - P[1] * dc[0];
    const T is = inv_sqrt<T>(n0 * n0 + n1 * n1);
    const T m0 = n0 * is, m1 = n1 * is, m2 = n2 * is;
    r[2 * k] = -(ob[4 * k] * m0 + ob[4 * k + 1] * m1 + m2);
    r[2 * k + 1] = -(ob[4 * k + 2] * m0 + ob[4 * k + 3] * m1 + m2);
  }
}

// Residuals and row-major Jacobians Jc[4][6] (w, t) and Jl[4][4] (a, b, g, t).
template <typename T>
SLS_HD void obs_linearise(const T R[9], const T JL[9], const T t[3],
                          const T cp[3], const T dv[3], const T dcp[12], const T ddv[9],
                          const T ob[8], T baseline, T r[4], T Jc[24], T Jl[16]) {
  T Q[3], P[3], dc[3];
  for (int i = 0; i < 3; ++i) {
    Q[i] = R[3 * i] * cp[0] + R[3 * i + 1] * cp[1] + R[3 * i + 2] * cp[2];
    P[i] = Q[i] + t[i];
    dc[i] = R[3 * i] * dv[0] + R[3 * i + 1] * dv[1] + R[3 * i + 2] * dv[2];
  }
  for (int k = 0; k < 2; ++k) {
    if (k == 1) P[0] -= baseline;
    const T n0 = P[1] * dc[2] - P[2] * dc[1];
    const T n1 = P[2] * dc[0] - P[0] * dc[2];
    const T n2 = P[0] * dc[1] - P[1] * dc[0];
    const T is = inv_sqrt<T>(n0 * n0 + n1 * n1);
    const T m0 = n0 * is, m1 = n1 * is, m2 = n2 * is;
    for (int e = 0; e < 2; ++e) {
      const int row = 2 * k + e;
      const T x = ob[4 * k + 2 * e], y = ob[4 * k + 2 * e + 1];
      const T rho = x * m0 + y * m1 + m2;
      r[row] = -rho;
      const T q0 = -(x - rho * m0) * is, q1 = -(y - rho * m1) * is, q2 = -is;
      const T gP[3] = { dc[1] * q2 - dc[2] * q1, dc[2] * q0 - dc[0] * q2, dc[0] * q1 - dc[1] * q0 };
      const T gD[3] = { q1 * P[2] - q2 * P[1], q2 * P[0] - q0 * P[2], q0 * P[1] - q1 * P[0] };
      // tau = (R cp) x gP + dc x gD ;  dr/dw = tau^T JL
      const T tau[3] = { Q[1] * gP[2] - Q[2] * gP[1] + dc[1] * gD[2] - dc[2] * gD[1],
                         Q[2] * gP[0] - Q[0] * gP[2] + dc[2] * gD[0] - dc[0] * gD[2],
                         Q[0] * gP[1] - Q[1] * gP[0] + dc[0] * gD[1] - dc[1] * gD[0] };
      T* jc = Jc + 6 * row;
      for (int j = 0; j < 3; ++j) jc[j] = tau[0] * JL[j] + tau[1] * JL[3 + j] + tau[2] * JL[6 + j];
      jc[3] = gP[0]; jc[4] = gP[1]; jc[5] = gP[2];
      T hP[3], hD[3];                                    // R^T gP, R^T gD
      for (int i = 0; i < 3; ++i) {
        hP[i] = R[i] * gP[0] + R[3 + i] * gP[1] + R[6 + i] * gP[2];
        hD[i] = R[i] * gD[0] + R[3 + i] * gD[1] + R[6 + i] * gD[2];
      }
      T* jl = Jl + 4 * row;
      for (int j = 0; j < 3; ++j)
        jl[j] = hP[0] * dcp[3 * j] + hP[1] * dcp[3 * j + 1] + hP[2] * dcp[3 * j + 2]
              + hD[0] * ddv[3 * j] + hD[1] * ddv[3 * j + 1] + hD[2] * ddv[3 * j + 2];
      jl[3] = hP[0] * dcp[9] + hP[1] * dcp[10] + hP[2] * dcp[11];
    }
  }
}

template <typename T>
SLS_HD T huber_scale(T s, T a, T* cost);      // (defined at the end of this header)

// The same observation in RAW camera coordinates, robustified and with the line's Jacobi scale folded in:
//   J_c' rows = [tau | gP] sqrt(rho')  - the SO(3) left Jacobian (d r / d w = tau^T JL(w)) and the Jacobi column scale of the camera
//     are per-camera constants, which the reduced solve applies once per window as a congruence of the reduced system;
//   J_l  rows = (gP^T Mp_j + gD^T Md_j) sl_j sqrt(rho'),  [Mp_j | Md_j] = R [d cp / d u_j | d dv / d u_j]: with the line's frame
//     in camera coordinates E = R R_l = [e0 dc e2] (cp = -d col2, dv = col1) and r3 = R e_z:
//       a: Mp = d dc          Md = e2           b: Mp = -d c1 e0      Md = s1 e0
//       g: Mp = -d (r3 x e2)  Md = r3 x dc      t: Mp = (1 + d^2) e2  Md = 0
//     - seven 3-vectors per observation instead of R and the 21 partials of line_points_jac, and no R^T gP / R^T gD per row.
// The four rho come first, so that the Huber factor sqrt(rho') multiplies the two 1/s of the stereo pair (every entry of both
// Jacobians is linear in the row gradient q, which is linear in 1/s) instead of the 44 entries one by one.  The rows of J_c' are
// handed to `jc_row(row, jc[6])` as they are formed (the grouped elimination sweep parks them in LDS: lba_eliminate_grouped.h).
template <typename T, typename Sink>
SLS_HD void obs_linearise_raw(const T R[9], const T t[3], const T trig[7], const T sl[4], const T ob[8], T baseline, T huber_delta,
                              T rs[4], T Jl[16], T* cost, Sink&& jc_row) {
  const T s1 = trig[0], c1 = trig[1], s2 = trig[2], c2 = trig[3], s3 = trig[4], c3 = trig[5], d = trig[6];
  const T col0[3] = { c2 * c3, c2 * s3, -s2 };
  const T col1[3] = { s1 * s2 * c3 - c1 * s3, s1 * s2 * s3 + c1 * c3, s1 * c2 };
  const T col2[3] = { c1 * s2 * c3 + s1 * s3, c1 * s2 * s3 - s1 * c3, c1 * c2 };
  T e0[3], dc[3], e2[3], Q[3], P[3];
  for (int i = 0; i < 3; ++i) {
    e0[i] = R[3 * i] * col0[0] + R[3 * i + 1] * col0[1] + R[3 * i + 2] * col0[2];
    dc[i] = R[3 * i] * col1[0] + R[3 * i + 1] * col1[1] + R[3 * i + 2] * col1[2];
    e2[i] = R[3 * i] * col2[0] + R[3 * i + 1] * col2[1] + R[3 * i + 2] * col2[2];
    Q[i] = -d * e2[i];
    P[i] = Q[i] + t[i];
  }
  const T r3[3] = { R[2], R[5], R[8] };
  T Mp[4][3], Md[3][3];
  {
    const T k0 = d * sl[0], k1p = -d * c1 * sl[1], k1d = s1 * sl[1], k2p = -d * sl[2], k3 = (T(1) + d * d) * sl[3];
    const T x2[3] = { r3[1] * e2[2] - r3[2] * e2[1], r3[2] * e2[0] - r3[0] * e2[2], r3[0] * e2[1] - r3[1] * e2[0] };
    const T x1[3] = { r3[1] * dc[2] - r3[2] * dc[1], r3[2] * dc[0] - r3[0] * dc[2], r3[0] * dc[1] - r3[1] * dc[0] };
    for (int i = 0; i < 3; ++i) {
      Mp[0][i] = k0 * dc[i];  Md[0][i] = sl[0] * e2[i];
      Mp[1][i] = k1p * e0[i]; Md[1][i] = k1d * e0[i];
      Mp[2][i] = k2p * x2[i]; Md[2][i] = sl[2] * x1[i];
      Mp[3][i] = k3 * e2[i];
    }
  }
  T m[2][2], is[2], rho[4], px[2] = { P[0], P[0] - baseline };
  for (int k = 0; k < 2; ++k) {
    const T n0 = P[1] * dc[2] - P[2] * dc[1];
    const T n1 = P[2] * dc[0] - px[k] * dc[2];
    const T n2 = px[k] * dc[1] - P[1] * dc[0];
    is[k] = inv_sqrt<T>(n0 * n0 + n1 * n1);
    m[k][0] = n0 * is[k]; m[k][1] = n1 * is[k];
    const T m2 = n2 * is[k];
    for (int e = 0; e < 2; ++e) rho[2 * k + e] = ob[4 * k + 2 * e] * m[k][0] + ob[4 * k + 2 * e + 1] * m[k][1] + m2;
  }
  const T sr = huber_scale<T>(rho[0] * rho[0] + rho[1] * rho[1] + rho[2] * rho[2] + rho[3] * rho[3], huber_delta, cost);
  for (int k = 0; k < 2; ++k) {
    const T iss = is[k] * sr;
    for (int e = 0; e < 2; ++e) {
      const int row = 2 * k + e;
      const T x = ob[4 * k + 2 * e], y = ob[4 * k + 2 * e + 1];
      rs[row] = -rho[row] * sr;
      const T q0 = -(x - rho[row] * m[k][0]) * iss, q1 = -(y - rho[row] * m[k][1]) * iss, q2 = -iss;
      const T gP[3] = { dc[1] * q2 - dc[2] * q1, dc[2] * q0 - dc[0] * q2, dc[0] * q1 - dc[1] * q0 };
      const T gD[3] = { q1 * P[2] - q2 * P[1], q2 * px[k] - q0 * P[2], q0 * P[1] - q1 * px[k] };
      T jc[6];
      jc[0] = Q[1] * gP[2] - Q[2] * gP[1] + dc[1] * gD[2] - dc[2] * gD[1];
      jc[1] = Q[2] * gP[0] - Q[0] * gP[2] + dc[2] * gD[0] - dc[0] * gD[2];
      jc[2] = Q[0] * gP[1] - Q[1] * gP[0] + dc[0] * gD[1] - dc[1] * gD[0];
      jc[3] = gP[0]; jc[4] = gP[1]; jc[5] = gP[2];
      jc_row(row, jc);
      T* jl = Jl + 4 * row;
      for (int j = 0; j < 3; ++j)
        jl[j] = gP[0] * Mp[j][0] + gP[1] * Mp[j][1] + gP[2] * Mp[j][2] + gD[0] * Md[j][0] + gD[1] * Md[j][1] + gD[2] * Md[j][2];
      jl[3] = gP[0] * Mp[3][0] + gP[1] * Mp[3][1] + gP[2] * Mp[3][2];
    }
  }
}

// MIXED precision (slslam_solver_options.lba_precision = 1; the steady elimination sweeps only - the first sweep of a solve, which is also
// Ceres' initial evaluation, stays double): the same observation with
//   * the geometry, the residuals, the Huber factor, the row gradients q, gP, gD and the LINE Jacobian J_l in DOUBLE, exactly as above;
//   * the CAMERA Jacobian J_c' = [tau | gP] in FLOAT, two rows at a time - the two endpoints of one camera of the stereo pair share
//     everything but q - on 2-wide vectors, which gfx950 executes as packed fp32 (v_pk_fma_f32 / v_pk_mul_f32).
// Why only J_c': measured, not assumed (profiles/round5_mixed_precision_study.txt).  With BOTH Jacobians in float 8 of 28 bench-family
// windows change an accept / reject decision and final costs move by up to 2e-2: a window holds a few depth-degenerate lines (seen
// under a few degrees of parallax: an eigenvalue of their 4 x 4 block ~1e-8 of the others), and a line Jacobian that is off by 1e-6 of its
// row's largest entry moves their step by percents.  J_l in double and J_c' in float: 28 of 28 windows take the same decisions, final cost
// within 7e-5, poses within 2e-6 - camera blocks sum ~600 observations each and are well conditioned.  (The other way round is as bad
// as both; only the depth column of J_l in double does not help.)
// tau = Q x gP + dc x gD with gP = dc x q, gD = q x P_k, P_k = Q + t_k: for a far line (|Q| = |d| large, reference src/lba_problem.h:63)
// the two products cancel to the size of q - in float the rotation part of J_c' would lose its digits there.  By the Jacobi identity
// Q x (dc x q) + dc x (q x Q) = q x (dc x Q), so tau = q x W + dc x (q x t_k) with W = dc x Q (formed in double, once per
// observation): nothing cancels any more.
#if defined(__HIPCC__)
typedef float sls_f2 __attribute__((ext_vector_type(2)));
#else
typedef float sls_f2 __attribute__((vector_size(8)));
#endif
template <typename Sink>
SLS_HD void obs_linearise_raw_mixed(const double R[9], const double t[3], const double trig[7], const double sl[4], const double ob[8],
                                    double baseline, double huber_delta, double rs[4], double Jl[16], double* cost, Sink&& jc_row) {
  const double s1 = trig[0], c1 = trig[1], s2 = trig[2], c2 = trig[3], s3 = trig[4], c3 = trig[5], d = trig[6];
  const double col0[3] = { c2 * c3, c2 * s3, -s2 };
  const double col1[3] = { s1 * s2 * c3 - c1 * s3, s1 * s2 * s3 + c1 * c3, s1 * c2 };
  const double col2[3] = { c1 * s2 * c3 + s1 * s3, c1 * s2 * s3 - s1 * c3, c1 * c2 };
  double e0[3], dc[3], e2[3], Q[3], P[3];
  for (int i = 0; i < 3; ++i) {
    e0[i] = R[3 * i] * col0[0] + R[3 * i + 1] * col0[1] + R[3 * i + 2] * col0[2];
    dc[i] = R[3 * i] * col1[0] + R[3 * i + 1] * col1[1] + R[3 * i + 2] * col1[2];
    e2[i] = R[3 * i] * col2[0] + R[3 * i + 1] * col2[1] + R[3 * i + 2] * col2[2];
    Q[i] = -d * e2[i];
    P[i] = Q[i] + t[i];
  }
  const double r3[3] = { R[2], R[5], R[8] };
  double Mp[4][3], Md[3][3];
  {
    const double k0 = d * sl[0], k1p = -d * c1 * sl[1], k1d = s1 * sl[1], k2p = -d * sl[2], k3 = (1.0 + d * d) * sl[3];
    const double x2[3] = { r3[1] * e2[2] - r3[2] * e2[1], r3[2] * e2[0] - r3[0] * e2[2], r3[0] * e2[1] - r3[1] * e2[0] };
    const double x1[3] = { r3[1] * dc[2] - r3[2] * dc[1], r3[2] * dc[0] - r3[0] * dc[2], r3[0] * dc[1] - r3[1] * dc[0] };
    for (int i = 0; i < 3; ++i) {
      Mp[0][i] = k0 * dc[i];  Md[0][i] = sl[0] * e2[i];
      Mp[1][i] = k1p * e0[i]; Md[1][i] = k1d * e0[i];
      Mp[2][i] = k2p * x2[i]; Md[2][i] = sl[2] * x1[i];
      Mp[3][i] = k3 * e2[i];
    }
  }
  double m[2][2], is[2], rho[4], px[2] = { P[0], P[0] - baseline };
  for (int k = 0; k < 2; ++k) {
    const double n0 = P[1] * dc[2] - P[2] * dc[1];
    const double n1 = P[2] * dc[0] - px[k] * dc[2];
    const double n2 = px[k] * dc[1] - P[1] * dc[0];
    is[k] = inv_sqrt<double>(n0 * n0 + n1 * n1);
    m[k][0] = n0 * is[k]; m[k][1] = n1 * is[k];
    const double m2 = n2 * is[k];
    for (int e = 0; e < 2; ++e) rho[2 * k + e] = ob[4 * k + 2 * e] * m[k][0] + ob[4 * k + 2 * e + 1] * m[k][1] + m2;
  }
  const double sr = huber_scale<double>(rho[0] * rho[0] + rho[1] * rho[1] + rho[2] * rho[2] + rho[3] * rho[3], huber_delta, cost);
  // what the float part needs of the observation, narrowed once
  float dcf[3];
  for (int i = 0; i < 3; ++i) dcf[i] = (float)dc[i];
  const float Wf[3] = { (float)(dc[1] * Q[2] - dc[2] * Q[1]), (float)(dc[2] * Q[0] - dc[0] * Q[2]), (float)(dc[0] * Q[1] - dc[1] * Q[0]) };
  const float tkf[2][3] = { { (float)t[0], (float)t[1], (float)t[2] }, { (float)(t[0] - baseline), (float)t[1], (float)t[2] } };
  for (int k = 0; k < 2; ++k) {
    const double iss = is[k] * sr;
    double qd[2][3];
    for (int e = 0; e < 2; ++e) {
      const int row = 2 * k + e;
      const double x = ob[4 * k + 2 * e], y = ob[4 * k + 2 * e + 1];
      rs[row] = -rho[row] * sr;
      const double q0 = -(x - rho[row] * m[k][0]) * iss, q1 = -(y - rho[row] * m[k][1]) * iss, q2 = -iss;
      qd[e][0] = q0; qd[e][1] = q1; qd[e][2] = q2;
      const double gP[3] = { dc[1] * q2 - dc[2] * q1, dc[2] * q0 - dc[0] * q2, dc[0] * q1 - dc[1] * q0 };
      const double gD[3] = { q1 * P[2] - q2 * P[1], q2 * px[k] - q0 * P[2], q0 * P[1] - q1 * px[k] };
      double* jl = Jl + 4 * row;
      for (int j = 0; j < 3; ++j)
        jl[j] = gP[0] * Mp[j][0] + gP[1] * Mp[j][1] + gP[2] * Mp[j][2] + gD[0] * Md[j][0] + gD[1] * Md[j][1] + gD[2] * Md[j][2];
      jl[3] = gP[0] * Mp[3][0] + gP[1] * Mp[3][1] + gP[2] * Mp[3][2];
    }
    // ---- J_c' of the two rows of this camera, in float
    const sls_f2 q0 = { (float)qd[0][0], (float)qd[1][0] }, q1 = { (float)qd[0][1], (float)qd[1][1] };
    const float q2 = (float)qd[0][2];
    const sls_f2 gP[3] = { dcf[1] * q2 - q1 * dcf[2], q0 * dcf[2] - dcf[0] * q2, q1 * dcf[0] - q0 * dcf[1] };
    const sls_f2 u[3] = { q1 * tkf[k][2] - q2 * tkf[k][1], q2 * tkf[k][0] - q0 * tkf[k][2], q0 * tkf[k][1] - q1 * tkf[k][0] };
    sls_f2 jc[6];
    jc[0] = q1 * Wf[2] - q2 * Wf[1] + u[2] * dcf[1] - u[1] * dcf[2];
    jc[1] = q2 * Wf[0] - q0 * Wf[2] + u[0] * dcf[2] - u[2] * dcf[0];
    jc[2] = q0 * Wf[1] - q1 * Wf[0] + u[1] * dcf[0] - u[0] * dcf[1];
    jc[3] = gP[0]; jc[4] = gP[1]; jc[5] = gP[2];
    for (int e = 0; e < 2; ++e) {
      float row[6];
      for (int a = 0; a < 6; ++a) row[a] = jc[a][e];
      jc_row(2 * k + e, row);
    }
  }
}

// What the back-substitution needs of an observation, in one pass: the residual and  w = J_l^T (J_c y_c)  (4 values,
// unscaled: the caller applies the Huber factor and the Jacobi scale of the line's columns).  Neither Jacobian is
// formed: per residual row the gradients gP, gD w.r.t. the camera-frame point and direction give the row's
// (J_c y_c) = tau . (J_L y_w) + gP . y_t (tau = Q x gP + dc x gD: the rotation part through the camera's J_L), and because J_l's row is linear in (gP, gD),
//   sum_rows (J_c y)_row J_l[row] = (R^T sum_rows (J_c y)_row gP)^T dcp + (R^T sum_rows (J_c y)_row gD)^T ddv.
template <typename T>
SLS_HD void obs_backsub_w(const T R[9], const T t[3], const T vw[3], const T yt[3],
                          const T cp[3], const T dv[3], const T dcp[12], const T ddv[9],
                          const T ob[8], T baseline, T r[4], T w[4]) {
  T Q[3], P[3], dc[3];
  for (int i = 0; i < 3; ++i) {
    Q[i] = R[3 * i] * cp[0] + R[3 * i + 1] * cp[1] + R[3 * i + 2] * cp[2];
    P[i] = Q[i] + t[i];
    dc[i] = R[3 * i] * dv[0] + R[3 * i + 1] * dv[1] + R[3 * i + 2] * dv[2];
  }
  T GP[3] = { T(0), T(0), T(0) }, GD[3] = { T(0), T(0), T(0) };
  for (int k = 0; k < 2; ++k) {
    if (k == 1) P[0] -= baseline;
    const T n0 = P[1] * dc[2] - P[2] * dc[1];
    const T n1 = P[2] * dc[0] - P[0] * dc[2];
    const T n2 = P[0] * dc[1] - P[1] * dc[0];
    const T is = inv_sqrt<T>(n0 * n0 + n1 * n1);
    const T m0 = n0 * is, m1 = n1 * is, m2 = n2 * is;
    for (int e = 0; e < 2; ++e) {
      const int row = 2 * k + e;
      const T x = ob[4 * k + 2 * e], y = ob[4 * k + 2 * e + 1];
      const T rho = x * m0 + y * m1 + m2;
      r[row] = -rho;
      const T q0 = -(x - rho * m0) * is, q1 = -(y - rho * m1) * is, q2 = -is;
      const T gP[3] = { dc[1] * q2 - dc[2] * q1, dc[2] * q0 - dc[0] * q2, dc[0] * q1 - dc[1] * q0 };
      const T gD[3] = { q1 * P[2] - q2 * P[1], q2 * P[0] - q0 * P[2], q0 * P[1] - q1 * P[0] };
      const T tau[3] = { Q[1] * gP[2] - Q[2] * gP[1] + dc[1] * gD[2] - dc[2] * gD[1],
                         Q[2] * gP[0] - Q[0] * gP[2] + dc[2] * gD[0] - dc[0] * gD[2],
                         Q[0] * gP[1] - Q[1] * gP[0] + dc[0] * gD[1] - dc[1] * gD[0] };
      const T jy = tau[0] * vw[0] + tau[1] * vw[1] + tau[2] * vw[2] + gP[0] * yt[0] + gP[1] * yt[1] + gP[2] * yt[2];
      for (int i = 0; i < 3; ++i) { GP[i] += jy * gP[i]; GD[i] += jy * gD[i]; }
    }
  }
  T hP[3], hD[3];                                        // R^T GP, R^T GD
  for (int i = 0; i < 3; ++i) {
    hP[i] = R[i] * GP[0] + R[3 + i] * GP[1] + R[6 + i] * GP[2];
    hD[i] = R[i] * GD[0] + R[3 + i] * GD[1] + R[6 + i] * GD[2];
  }
  for (int j = 0; j < 3; ++j)
    w[j] = hP[0] * dcp[3 * j] + hP[1] * dcp[3 * j + 1] + hP[2] * dcp[3 * j + 2]
         + hD[0] * ddv[3 * j] + hD[1] * ddv[3 * j + 1] + hD[2] * ddv[3 * j + 2];
  w[3] = hP[0] * dcp[9] + hP[1] * dcp[10] + hP[2] * dcp[11];
}

// ceres::HuberLoss(a) + Corrector for rho'' <= 0 (lba_problem.cpp:78-80): returns the factor
// sqrt(rho') that scales both residual and Jacobian, and the block cost rho/2.
// a <= 0 disables the loss (FLAGS_robust = false).
template <typename T>
SLS_HD T huber_scale(T s, T a, T* cost) {
  if (a > T(0) && s > a * a) {
    const T t = inv_sqrt<T>(s);                 // 1 / ||r||
    *cost = T(0.5) * (T(2) * a * (s * t) - a * a);
    const T q = a * t;                          // rho' = a / ||r||
    return q * inv_sqrt<T>(q);                  // sqrt(rho')
  }
  *cost = T(0.5) * s;
  return T(1);
}

}  // namespace slslam
#endif  // SLSLAM_LBA_MATH_H_
