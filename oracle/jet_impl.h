/* oracle/jet_impl.h — forward-mode dual numbers ("jets") of compile-time width JN.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product library.
 *
 * The reference differentiates its residual functors with ceres::AutoDiffCostFunction
 * (reference src/lba_problem.cpp:65-74, src/po_problem.cpp:45-52), i.e. by evaluating the
 * templated functor on ceres::Jet<double,N>.  Ceres is a third-party dependency that is absent
 * from /root/reference (README:8 pins "Ceres Solver 1.7.0"), so this header restates the
 * published Jet algebra (value + N partials, chain rule per elementary op).
 *
 * Usage: #define JN <width> and JT <typename> and JF(name) <prefixing macro>, then include.
 */
#include <math.h>

typedef struct { double v; double d[JN]; } JT;

static inline JT JF(cst)(double c) { JT r; r.v = c; for (int i = 0; i < JN; ++i) r.d[i] = 0.0; return r; }
static inline JT JF(var)(double c, int k) { JT r = JF(cst)(c); r.d[k] = 1.0; return r; }
static inline JT JF(neg)(JT a) { JT r; r.v = -a.v; for (int i = 0; i < JN; ++i) r.d[i] = -a.d[i]; return r; }
static inline JT JF(add)(JT a, JT b) { JT r; r.v = a.v + b.v; for (int i = 0; i < JN; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
static inline JT JF(sub)(JT a, JT b) { JT r; r.v = a.v - b.v; for (int i = 0; i < JN; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
static inline JT JF(mul)(JT a, JT b) { JT r; r.v = a.v * b.v; for (int i = 0; i < JN; ++i) r.d[i] = a.v * b.d[i] + a.d[i] * b.v; return r; }
static inline JT JF(div)(JT a, JT b) {
  /* Ceres jet.h: h = f/g, dh = (df - h dg)/g */
  JT r; const double ginv = 1.0 / b.v; r.v = a.v * ginv;
  for (int i = 0; i < JN; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ginv;
  return r; }
static inline JT JF(adds)(JT a, double s) { JT r = a; r.v += s; return r; }
static inline JT JF(muls)(JT a, double s) { JT r; r.v = a.v * s; for (int i = 0; i < JN; ++i) r.d[i] = a.d[i] * s; return r; }
static inline JT JF(sin)(JT a) { JT r; const double c = cos(a.v); r.v = sin(a.v); for (int i = 0; i < JN; ++i) r.d[i] = c * a.d[i]; return r; }
static inline JT JF(cos)(JT a) { JT r; const double s = -sin(a.v); r.v = cos(a.v); for (int i = 0; i < JN; ++i) r.d[i] = s * a.d[i]; return r; }
static inline JT JF(sqrt)(JT a) { JT r; r.v = sqrt(a.v); const double k = 1.0 / (2.0 * r.v); for (int i = 0; i < JN; ++i) r.d[i] = k * a.d[i]; return r; }
static inline JT JF(atan2)(JT y, JT x) {
  /* d atan2(y,x) = (x dy - y dx)/(x^2+y^2) */
  JT r; r.v = atan2(y.v, x.v); const double k = 1.0 / (x.v * x.v + y.v * y.v);
  for (int i = 0; i < JN; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * k;
  return r; }

/* ---- ceres/rotation.h restatements (Ceres 1.7.0; published algorithm, header not in tree) ---- */

/* ceres::AngleAxisRotatePoint — Rodrigues away from zero, first-order branch p + w x p at zero.
 * Used by reference src/lba_problem.h:75-76 and src/po_problem.h:38,59. */
static inline void JF(aa_rotate_point)(const JT w[3], const JT p[3], JT out[3]) {
  const JT theta2 = JF(add)(JF(add)(JF(mul)(w[0], w[0]), JF(mul)(w[1], w[1])), JF(mul)(w[2], w[2]));
  if (theta2.v > 0.0) {
    const JT theta = JF(sqrt)(theta2);
    JT u[3] = { JF(div)(w[0], theta), JF(div)(w[1], theta), JF(div)(w[2], theta) };
    const JT ct = JF(cos)(theta), st = JF(sin)(theta);
    JT uxp[3] = { JF(sub)(JF(mul)(u[1], p[2]), JF(mul)(u[2], p[1])),
                  JF(sub)(JF(mul)(u[2], p[0]), JF(mul)(u[0], p[2])),
                  JF(sub)(JF(mul)(u[0], p[1]), JF(mul)(u[1], p[0])) };
    const JT udp = JF(add)(JF(add)(JF(mul)(u[0], p[0]), JF(mul)(u[1], p[1])), JF(mul)(u[2], p[2]));
    const JT omc = JF(sub)(JF(cst)(1.0), ct);
    for (int i = 0; i < 3; ++i)
      out[i] = JF(add)(JF(add)(JF(mul)(p[i], ct), JF(mul)(uxp[i], st)), JF(mul)(JF(mul)(u[i], omc), udp));
  } else {
    JT wxp[3] = { JF(sub)(JF(mul)(w[1], p[2]), JF(mul)(w[2], p[1])),
                  JF(sub)(JF(mul)(w[2], p[0]), JF(mul)(w[0], p[2])),
                  JF(sub)(JF(mul)(w[0], p[1]), JF(mul)(w[1], p[0])) };
    for (int i = 0; i < 3; ++i) out[i] = JF(add)(p[i], wxp[i]);
  }
}

/* ceres::AngleAxisToQuaternion ([w,x,y,z]); used by reference src/po_problem.h:47-48 */
static inline void JF(aa_to_quat)(const JT a[3], JT q[4]) {
  const JT t2 = JF(add)(JF(add)(JF(mul)(a[0], a[0]), JF(mul)(a[1], a[1])), JF(mul)(a[2], a[2]));
  if (t2.v > 0.0) {
    const JT theta = JF(sqrt)(t2);
    const JT half = JF(muls)(theta, 0.5);
    const JT k = JF(div)(JF(sin)(half), theta);
    q[0] = JF(cos)(half);
    q[1] = JF(mul)(a[0], k); q[2] = JF(mul)(a[1], k); q[3] = JF(mul)(a[2], k);
  } else {
    q[0] = JF(cst)(1.0);
    q[1] = JF(muls)(a[0], 0.5); q[2] = JF(muls)(a[1], 0.5); q[3] = JF(muls)(a[2], 0.5);
  }
}

/* ceres::QuaternionProduct (Hamilton, [w,x,y,z]); reference src/po_problem.h:50 */
static inline void JF(quat_product)(const JT z[4], const JT w[4], JT zw[4]) {
  zw[0] = JF(sub)(JF(sub)(JF(sub)(JF(mul)(z[0], w[0]), JF(mul)(z[1], w[1])), JF(mul)(z[2], w[2])), JF(mul)(z[3], w[3]));
  zw[1] = JF(sub)(JF(add)(JF(add)(JF(mul)(z[0], w[1]), JF(mul)(z[1], w[0])), JF(mul)(z[2], w[3])), JF(mul)(z[3], w[2]));
  zw[2] = JF(add)(JF(add)(JF(sub)(JF(mul)(z[0], w[2]), JF(mul)(z[1], w[3])), JF(mul)(z[2], w[0])), JF(mul)(z[3], w[1]));
  zw[3] = JF(add)(JF(sub)(JF(add)(JF(mul)(z[0], w[3]), JF(mul)(z[1], w[2])), JF(mul)(z[2], w[1])), JF(mul)(z[3], w[0]));
}

/* ceres::QuaternionToAngleAxis; reference src/po_problem.h:51 */
static inline void JF(quat_to_aa)(const JT q[4], JT a[3]) {
  const JT s2 = JF(add)(JF(add)(JF(mul)(q[1], q[1]), JF(mul)(q[2], q[2])), JF(mul)(q[3], q[3]));
  if (s2.v > 0.0) {
    const JT st = JF(sqrt)(s2);
    const JT ct = q[0];
    const JT two_theta = JF(muls)((ct.v < 0.0) ? JF(atan2)(JF(neg)(st), JF(neg)(ct)) : JF(atan2)(st, ct), 2.0);
    const JT k = JF(div)(two_theta, st);
    a[0] = JF(mul)(q[1], k); a[1] = JF(mul)(q[2], k); a[2] = JF(mul)(q[3], k);
  } else {
    a[0] = JF(muls)(q[1], 2.0); a[1] = JF(muls)(q[2], 2.0); a[2] = JF(muls)(q[3], 2.0);
  }
}
