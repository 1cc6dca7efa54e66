// slslam_amd/host/ceres/rotation.h — the four rotation helpers the reference's SE(3) templates call (src/po_problem.h:27-64:
// ceres::AngleAxisRotatePoint, AngleAxisToQuaternion, QuaternionProduct, QuaternionToAngleAxis; also src/lba_problem.h:75-76),
// so that `#include "ceres/rotation.h"` gives a caller of the mirrored headers the same names on the host.
//
// Restated from the published definitions of Ceres Solver 1.7.0's rotation.h (the version the reference pins, README:8) - the
// same definitions the device functors (slslam_amd/csrc/po_kernels.h, lba_math.h) follow:
//   * angle-axis a, theta = |a|; quaternion q = [w, x, y, z];
//   * AngleAxisToQuaternion: q = [cos(theta / 2), a sin(theta / 2) / theta], and [1, a / 2] at theta == 0;
//   * QuaternionToAngleAxis: a = q_v * 2 atan2(|q_v|, w) / |q_v| with the angle taken in (-pi, pi] (both signs flipped when w < 0),
//     and a = 2 q_v when q_v == 0;
//   * QuaternionProduct: Hamilton product;
//   * AngleAxisRotatePoint: Rodrigues' formula p cos + (k x p) sin + k (k . p)(1 - cos), k = a / theta, and the first-order form
//     p + a x p at theta == 0.
// Templated on the scalar like the originals; instantiated for double by tests/host_cxx and by callers of gc_T_inv / gc_w_20 / gc_T_20
// (po_problem.h).  The GPU path does not use this header.
#ifndef SLSLAM_HOST_CERES_ROTATION_H_
#define SLSLAM_HOST_CERES_ROTATION_H_

#include <cmath>

namespace ceres {

template <typename T>
inline void AngleAxisToQuaternion(const T* angle_axis, T* quaternion) {
  using std::sqrt; using std::sin; using std::cos;
  const T &a0 = angle_axis[0], &a1 = angle_axis[1], &a2 = angle_axis[2];
  const T theta2 = a0 * a0 + a1 * a1 + a2 * a2;
  T k(0.5), w(1.0);
  if (theta2 > T(0.0)) {
    const T theta = sqrt(theta2), half = theta * T(0.5);
    k = sin(half) / theta;
    w = cos(half);
  }
  quaternion[0] = w; quaternion[1] = a0 * k; quaternion[2] = a1 * k; quaternion[3] = a2 * k;
}

template <typename T>
inline void QuaternionToAngleAxis(const T* quaternion, T* angle_axis) {
  using std::sqrt; using std::atan2;
  const T &q1 = quaternion[1], &q2 = quaternion[2], &q3 = quaternion[3];
  const T s2 = q1 * q1 + q2 * q2 + q3 * q3;
  T k(2.0);
  if (s2 > T(0.0)) {
    const T s = sqrt(s2), &c = quaternion[0];
    const T two_theta = T(2.0) * ((c < T(0.0)) ? atan2(-s, -c) : atan2(s, c));
    k = two_theta / s;
  }
  angle_axis[0] = q1 * k; angle_axis[1] = q2 * k; angle_axis[2] = q3 * k;
}

template <typename T>
inline void QuaternionProduct(const T z[4], const T w[4], T zw[4]) {
  zw[0] = z[0] * w[0] - z[1] * w[1] - z[2] * w[2] - z[3] * w[3];
  zw[1] = z[0] * w[1] + z[1] * w[0] + z[2] * w[3] - z[3] * w[2];
  zw[2] = z[0] * w[2] - z[1] * w[3] + z[2] * w[0] + z[3] * w[1];
  zw[3] = z[0] * w[3] + z[1] * w[2] - z[2] * w[1] + z[3] * w[0];
}

template <typename T>
inline void AngleAxisRotatePoint(const T angle_axis[3], const T pt[3], T result[3]) {
  using std::sqrt; using std::sin; using std::cos;
  const T theta2 = angle_axis[0] * angle_axis[0] + angle_axis[1] * angle_axis[1] + angle_axis[2] * angle_axis[2];
  const T p[3] = { pt[0], pt[1], pt[2] };                      // (result may alias pt)
  if (theta2 > T(0.0)) {
    const T theta = sqrt(theta2), c = cos(theta), s = sin(theta);
    const T k[3] = { angle_axis[0] / theta, angle_axis[1] / theta, angle_axis[2] / theta };
    const T kxp[3] = { k[1] * p[2] - k[2] * p[1], k[2] * p[0] - k[0] * p[2], k[0] * p[1] - k[1] * p[0] };
    const T t = (k[0] * p[0] + k[1] * p[1] + k[2] * p[2]) * (T(1.0) - c);
    for (int i = 0; i < 3; ++i) result[i] = p[i] * c + kxp[i] * s + k[i] * t;
  } else {
    const T* a = angle_axis;
    result[0] = p[0] + (a[1] * p[2] - a[2] * p[1]);
    result[1] = p[1] + (a[2] * p[0] - a[0] * p[2]);
    result[2] = p[2] + (a[0] * p[1] - a[1] * p[0]);
  }
}

}  // namespace ceres
#endif
