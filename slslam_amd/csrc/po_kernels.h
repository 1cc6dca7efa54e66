// slslam_amd/csrc/po_kernels.h — hand-written CDNA4 kernels of the pose-graph optimisation.
//
// What this replaces: ceres::Solve on the problem POProblem::build wires up (reference
// src/po_problem.cpp:40-77, call site src/slam.cpp:1283-1293): one <6,6,6> residual block per edge,
// Te = T2^-1 * (C * T1) in (angle-axis, translation) form (src/po_problem.h:68-108), pose1 of
// edge 0 constant, no loss function, SPARSE_NORMAL_CHOLESKY.
//
//   k_po_linearise   lane <-> (edge, seed direction): the templated SE(3) functor is evaluated on
//                    a one-direction dual number per lane (12 lanes = the 12 columns of [J1|J2]),
//                    columns are exchanged by wave shuffles and J^T J / J^T r go to the dense
//                    normal matrix with fp64 global atomics (E ~ 300 edges: tiny)
//   k_po_prepare     gradient max-norm, Jacobi scale (first call), LM damping, rhs
//   k_po_potrf_diag / k_po_panel_update
//                    blocked right-looking Cholesky of the dense (6N)^2 matrix, 64x64 blocks;
//                    TRSM and the trailing SYRK/GEMM run on v_mfma_f64_16x16x4_f64 - the one
//                    real matrix contraction on this path
//   k_po_trisolve    forward / backward substitution
//   k_po_candidate   x+ = x - scale*y and the step statistics;  k_po_cost  cost at x+
//   k_po_update      trust-region bookkeeping (same policy as the LBA path)
#ifndef SLSLAM_PO_KERNELS_H_
#define SLSLAM_PO_KERNELS_H_

#include <hip/hip_runtime.h>
#include "lba_types.h"
#include "dense_tile.h"

namespace slslam {

// ------------------------------------------------------------------------------------------
// one-direction forward-mode dual number
struct Dual {
  double v, d;
};
__device__ __forceinline__ Dual mk(double v, double d = 0.0) { Dual r; r.v = v; r.d = d; return r; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return mk(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return mk(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return mk(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return mk(a.v * b.v, a.v * b.d + a.d * b.v); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) { const double q = a.v / b.v; return mk(q, (a.d - q * b.d) / b.v); }
__device__ __forceinline__ Dual dsin(Dual a) { return mk(sin(a.v), cos(a.v) * a.d); }
__device__ __forceinline__ Dual dcos(Dual a) { return mk(cos(a.v), -sin(a.v) * a.d); }
__device__ __forceinline__ Dual dsqrt(Dual a) { const double s = sqrt(a.v); return mk(s, a.d / (2.0 * s)); }
__device__ __forceinline__ Dual datan2(Dual y, Dual x) { return mk(atan2(y.v, x.v), (x.v * y.d - y.v * x.d) / (x.v * x.v + y.v * y.v)); }
__device__ __forceinline__ double val(Dual a) { return a.v; }

// plain-double overloads so that the functor below also instantiates for T = double
__device__ __forceinline__ double dsin(double a) { return sin(a); }
__device__ __forceinline__ double dcos(double a) { return cos(a); }
__device__ __forceinline__ double dsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ double datan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ double val(double a) { return a; }
template <typename T> __device__ __forceinline__ T cst(double v);
template <> __device__ __forceinline__ double cst<double>(double v) { return v; }
template <> __device__ __forceinline__ Dual cst<Dual>(double v) { return mk(v); }

// ceres::AngleAxisRotatePoint (Rodrigues; first-order branch at zero) — used by gc_T_inv / gc_T_20
template <typename T>
__device__ __forceinline__ void aa_rotate(const T w[3], const T p[3], T out[3]) {
  const T th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  if (val(th2) > 0.0) {
    const T th = dsqrt(th2);
    const T u0 = w[0] / th, u1 = w[1] / th, u2 = w[2] / th;
    const T c = dcos(th), s = dsin(th);
    const T x0 = u1 * p[2] - u2 * p[1], x1 = u2 * p[0] - u0 * p[2], x2 = u0 * p[1] - u1 * p[0];
    const T dot = u0 * p[0] + u1 * p[1] + u2 * p[2];
    const T omc = cst<T>(1.0) - c;
    out[0] = p[0] * c + x0 * s + u0 * omc * dot;
    out[1] = p[1] * c + x1 * s + u1 * omc * dot;
    out[2] = p[2] * c + x2 * s + u2 * omc * dot;
  } else {
    out[0] = p[0] + (w[1] * p[2] - w[2] * p[1]);
    out[1] = p[1] + (w[2] * p[0] - w[0] * p[2]);
    out[2] = p[2] + (w[0] * p[1] - w[1] * p[0]);
  }
}

// gc_w_20 (reference src/po_problem.h:42-52): w20 = log(exp(w21) exp(w10)) through quaternions
// (ceres::AngleAxisToQuaternion, QuaternionProduct, QuaternionToAngleAxis)
template <typename T>
__device__ __forceinline__ void aa_to_quat(const T a[3], T q[4]) {
  const T t2 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
  if (val(t2) > 0.0) {
    const T th = dsqrt(t2);
    const T half = th * cst<T>(0.5);
    const T k = dsin(half) / th;
    q[0] = dcos(half); q[1] = a[0] * k; q[2] = a[1] * k; q[3] = a[2] * k;
  } else {
    q[0] = cst<T>(1.0); q[1] = a[0] * cst<T>(0.5); q[2] = a[1] * cst<T>(0.5); q[3] = a[2] * cst<T>(0.5);
  }
}
template <typename T>
__device__ __forceinline__ void quat_to_aa(const T q[4], T a[3]) {
  const T s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (val(s2) > 0.0) {
    const T st = dsqrt(s2);
    const T two = cst<T>(2.0) * ((val(q[0]) < 0.0) ? datan2(-st, -q[0]) : datan2(st, q[0]));
    const T k = two / st;
    a[0] = q[1] * k; a[1] = q[2] * k; a[2] = q[3] * k;
  } else {
    a[0] = q[1] * cst<T>(2.0); a[1] = q[2] * cst<T>(2.0); a[2] = q[3] * cst<T>(2.0);
  }
}
template <typename T>
__device__ __forceinline__ void se3_compose(const T T21[6], const T T10[6], T T20[6]) {   // gc_T_20, :55-64
  T q21[4], q10[4], q[4];
  aa_to_quat<T>(T21, q21);
  aa_to_quat<T>(T10, q10);
  q[0] = q21[0] * q10[0] - q21[1] * q10[1] - q21[2] * q10[2] - q21[3] * q10[3];
  q[1] = q21[0] * q10[1] + q21[1] * q10[0] + q21[2] * q10[3] - q21[3] * q10[2];
  q[2] = q21[0] * q10[2] - q21[1] * q10[3] + q21[2] * q10[0] + q21[3] * q10[1];
  q[3] = q21[0] * q10[3] + q21[1] * q10[2] - q21[2] * q10[1] + q21[3] * q10[0];
  quat_to_aa<T>(q, T20);
  aa_rotate<T>(T21, T10 + 3, T20 + 3);
  T20[3] = T20[3] + T21[3]; T20[4] = T20[4] + T21[4]; T20[5] = T20[5] + T21[5];
}
template <typename T>
__device__ __forceinline__ void se3_inverse(const T P[6], T Pi[6]) {                       // gc_T_inv, :27-39
  Pi[0] = -P[0]; Pi[1] = -P[1]; Pi[2] = -P[2];
  const T v[3] = { -P[3], -P[4], -P[5] };
  aa_rotate<T>(Pi, v, Pi + 3);
}
// PoseConstraintError::operator() (reference src/po_problem.h:74-105): Te = T2^-1 * (C * T1)
template <typename T>
__device__ __forceinline__ void pose_constraint_error(const T T1[6], const T T2[6], const T C[6], T Te[6]) {
  T Tc[6], T2i[6];
  se3_compose<T>(C, T1, Tc);
  se3_inverse<T>(T2, T2i);
  se3_compose<T>(T2i, Tc, Te);
}

// ------------------------------------------------------------------------------------------
struct PoPtrs {
  const int* p1; const int* p2;      // [E]
  const double* cons;                // [6E]
  const int* slot;                   // [N] offset of the pose in the reduced vector or -1 (gauge / unused)
  double* x;                         // [2][6N] accepted / candidate poses
  double* scale;                     // [n]
  double* H;                         // [n x ld] normal matrix, lower triangle; overwritten by its factor
  double* g;                         // [n] scaled gradient J'^T r
  double* d2;                        // [n] LM damping
  double* y;                         // [n] rhs -> solution
  double* scal;                      // [8] 0 cost, 1 candidate cost, 2 model, 3 dn2, 4 xn2, 5 fixed cost
  int* flags;                        // [2] 0: factorisation failure
  LMState* st;
  IterRec* trace;
  int N, E, n, ld;
};
enum { kPoCost = 0, kPoCandCost = 1, kPoModel = 2, kPoDn2 = 3, kPoXn2 = 4, kPoFixed = 5 };

#ifndef SLSLAM_PO_FACTOR_ONLY    // (lba_api.hip includes this header for the blocked Cholesky kernels only: lba_big.h)
// lane <-> (edge, column of [J1|J2]); 5 edges per wave.
// mode 0: accumulate H, g, cost at the accepted point (scaled columns)
// mode 1: cost only at the candidate point
__global__ __launch_bounds__(64) void k_po_linearise(PoPtrs p, int mode) {
  const LMState* st = p.st;
  if (st->status != kRunning) return;
  const int lane = threadIdx.x;
  const int el = lane / 12, d = lane - 12 * el;
  const int e = blockIdx.x * 5 + el;
  const bool ok = el < 5 && e < p.E;
  const int buf = mode ? 1 - st->cur : st->cur;
  const double* X = p.x + (long long)buf * 6 * p.N;
  const int es = ok ? e : 0;
  const int a = p.p1[es], b = p.p2[es];
  double cost = 0.0;
  if (mode == 1) {
    if (ok && d == 0) {
      double T1[6], T2[6], C[6], Te[6];
      for (int i = 0; i < 6; ++i) { T1[i] = X[6 * a + i]; T2[i] = X[6 * b + i]; C[i] = p.cons[6 * es + i]; }
      pose_constraint_error<double>(T1, T2, C, Te);
      const bool kept = p.slot[a] >= 0 || p.slot[b] >= 0;
      if (kept) for (int i = 0; i < 6; ++i) cost += 0.5 * Te[i] * Te[i];
    }
    for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
    if (lane == 0) atomicAdd(&p.scal[kPoCandCost], cost);
    return;
  }
  Dual T1[6], T2[6], C[6], Te[6];
  for (int i = 0; i < 6; ++i) {
    T1[i] = mk(X[6 * a + i], d == i ? 1.0 : 0.0);
    T2[i] = mk(X[6 * b + i], d == 6 + i ? 1.0 : 0.0);
    C[i] = mk(p.cons[6 * es + i]);
  }
  pose_constraint_error<Dual>(T1, T2, C, Te);
  const int sa = p.slot[a], sb = p.slot[b];
  const int my = d < 6 ? (sa >= 0 ? sa + d : -1) : (sb >= 0 ? sb + d - 6 : -1);
  const bool kept = sa >= 0 || sb >= 0;
  const double sc = (my >= 0) ? p.scale[my] : 0.0;
  double col[6], gsum = 0.0;
  for (int q = 0; q < 6; ++q) { col[q] = Te[q].d * sc; gsum += col[q] * Te[q].v; }
  const int base = 12 * el;
  for (int dp = 0; dp < 12; ++dp) {
    double h = 0.0;
    for (int q = 0; q < 6; ++q) h += col[q] * __shfl(col[q], base + dp);
    const int other = __shfl(my, base + dp);
    if (ok && my >= 0 && other >= 0 && my >= other) atomicAdd(&p.H[(long long)my * p.ld + other], h);
  }
  if (ok && my >= 0) atomicAdd(&p.g[my], gsum);
  if (ok && d == 0) {
    double c = 0.0;
    for (int q = 0; q < 6; ++q) c += 0.5 * Te[q].v * Te[q].v;
    if (kept) cost = c; else atomicAdd(&p.scal[kPoFixed], c);
  }
  for (int o = 32; o > 0; o >>= 1) cost += __shfl_xor(cost, o);
  if (lane == 0) atomicAdd(&p.scal[kPoCost], cost);
}

// Structured factorisation only: zeroes what the linearisation is about to add into - the lower-triangle entries of every edge's two
// pose blocks and their coupling (the same index rule as k_po_linearise), the junction block (the dense factorisation reads all of it),
// the gradient and the cost - instead of a memset of the whole dense matrix (20 MB at 1554 unknowns, a third of a structured solve
// in fill kernels).  Everything else the structured path reads from H it has written itself (k_po_chain_eliminate: factor and fill).
__global__ __launch_bounds__(256) void k_po_zero_structured(PoPtrs p, int n_chain /* unknowns of the level-1 chains: the square behind them is zeroed */) {
  if (p.st->status != kRunning) return;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n_edge = (long long)p.E * 144;
  const int nj = p.n - n_chain;
  const long long n_junc = (long long)nj * nj;
  if (t < n_edge) {
    const int e = (int)(t / 144), q = (int)(t - 144LL * e), d = q / 12, dp = q - 12 * d;
    const int sa = p.slot[p.p1[e]], sb = p.slot[p.p2[e]];
    const int my = d < 6 ? (sa >= 0 ? sa + d : -1) : (sb >= 0 ? sb + d - 6 : -1);
    const int other = dp < 6 ? (sa >= 0 ? sa + dp : -1) : (sb >= 0 ? sb + dp - 6 : -1);
    if (my >= 0 && other >= 0 && my >= other) p.H[(long long)my * p.ld + other] = 0.0;
  } else if (t < n_edge + n_junc) {
    const long long q = t - n_edge;
    const int r = (int)(q / nj), c = (int)(q - (long long)r * nj);
    p.H[(long long)(n_chain + r) * p.ld + n_chain + c] = 0.0;
  } else if (t < n_edge + n_junc + p.n) {
    p.g[t - n_edge - n_junc] = 0.0;
  } else if (t == n_edge + n_junc + p.n) {
    p.scal[kPoCost] = 0.0;
  }
}

// one workgroup: gradient max-norm, Jacobi scale on the first call, damping, rhs.
// first = 1: H holds the UNSCALED J^T J (scale == 1); compute scale, rescale H and g in place.
__global__ __launch_bounds__(256) void k_po_prepare(PoPtrs p, Policy pol, int first) {
  LMState* st = p.st;
  if (st->status != kRunning) return;
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double gm = 0.0;
  for (int i0 = tid; i0 < p.n; i0 += 8 * 256) {        // eight entries of this thread at a time: their loads travel together
    double sv[8], hv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u, is = i < p.n ? i : 0;
      sv[u] = p.scale[is]; gv[u] = p.g[is];
      hv[u] = p.H[(long long)is * p.ld + is];             // (unconditionally: a load under `first` is merged with its default by copies, i.e. waited for on the spot)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      if (i >= p.n) continue;
      double s = sv[u];
      if (first) {
        s = pol.jacobi_scaling ? 1.0 / (1.0 + sqrt(hv[u])) : 1.0;
        gm = fmax(gm, fabs(gv[u]));
        p.scale[i] = s;
      } else {
        gm = fmax(gm, fabs(gv[u] / s));
      }
    }
  }
  red[tid] = gm;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmax(red[tid], red[tid + o]); __syncthreads(); }
  gm = red[0];
  __syncthreads();
  const int ngc = st->need_grad_check;
  __syncthreads();
  if (first) {
    // initial evaluation (Ceres: cost, gradient, column norms at x0).  The host re-linearises with
    // the scale computed here, so nothing is damped or solved in this call.
    // |x|^2 over the free poses: the workgroup's threads share the poses (one thread walking all of them was 56 us of dependent
    // loads, once per solve), partial sums added in a fixed tree
    {
      const double* X = p.x + (long long)st->cur * 6 * p.N;
      double part = 0.0;
      for (int k = tid; k < p.N; k += 256) {
        const int sl = p.slot[k];
        double v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = X[6 * k + i];
        if (sl >= 0) for (int i = 0; i < 6; ++i) part += v[i] * v[i];
      }
      red[tid] = part;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    }
    if (tid == 0) {
      const double xn2 = red[0];
      st->cost = p.scal[kPoCost]; st->fixed_cost = p.scal[kPoFixed];
      st->initial_cost = st->cost + st->fixed_cost; st->min_cost = st->initial_cost;
      st->x_norm = sqrt(xn2); st->grad_max = gm;
      st->abs_grad_tol = pol.gradient_tolerance * (gm > 1e-12 ? gm : 1e-12);
      IterRec rec;
      rec.iteration = 0; rec.step_is_valid = 0; rec.step_is_successful = 0; rec.pad = 0;
      rec.cost = st->initial_cost; rec.cost_change = 0; rec.gradient_max_norm = gm; rec.step_norm = 0;
      rec.relative_decrease = 0; rec.trust_region_radius = st->radius; rec.model_cost_change = 0;
      if (!isfinite(st->cost)) st->status = 4;
      else if (gm <= st->abs_grad_tol) st->status = 1;
      else { p.trace[st->ntrace++] = rec; if (pol.max_num_iterations <= 0) st->status = 0; }
    }
    return;
  } else if (ngc) {
    if (tid == 0) {
      st->grad_max = gm; st->need_grad_check = 0;
      if (st->ntrace > 0 && st->ntrace <= kMaxTrace) p.trace[st->ntrace - 1].gradient_max_norm = gm;
      if (gm <= st->abs_grad_tol) st->status = 1;
    }
    __syncthreads();
    if (st->status != kRunning) return;
  }
  const double radius = st->radius;
  for (int i0 = tid; i0 < p.n; i0 += 8 * 256) {
    double hv[8], gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u, is = i < p.n ? i : 0;
      hv[u] = p.H[(long long)is * p.ld + is]; gv[u] = p.g[is];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + 256 * u;
      if (i >= p.n) continue;
      const double d2 = fmin(fmax(hv[u], pol.min_lm_diagonal), pol.max_lm_diagonal) / radius;
      p.d2[i] = d2;
      p.H[(long long)i * p.ld + i] = hv[u] + d2;
      p.y[i] = gv[u];
    }
  }
  if (tid == 0) { p.flags[0] = 0; p.scal[kPoCandCost] = 0.0; }
}

#endif
// ---- blocked Cholesky, block size 64 --------------------------------------------------------
// Templated on the factorisation type: double (default; v_mfma_f64_16x16x4_f64) or float
// (v_mfma_f32_16x16x4_f32) for the fp32-vs-fp64 tolerance study of BASELINE config 5.  In the float
// variant only the factor is single precision: residuals, Jacobians, gradient, costs and the LM
// bookkeeping stay fp64, so a less accurate step is caught by the gain ratio like any other.
enum { kNB = 64, kLdT = 66 };   // LDS tile leading dimension (f64: 132 dwords == 4 mod 64, conflict-free b64 reads)

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<double> {
  typedef v4f64 acc_t;
  static __device__ __forceinline__ acc_t zero() { return (acc_t){ 0.0, 0.0, 0.0, 0.0 }; }
  static __device__ __forceinline__ acc_t mac(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int i) { return (lane >> 4) + 4 * i; }     // f64 C/D map
};
template <> struct Mfma<float> {
  typedef v4f32 acc_t;
  static __device__ __forceinline__ acc_t zero() { return (acc_t){ 0.f, 0.f, 0.f, 0.f }; }
  static __device__ __forceinline__ acc_t mac(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int i) { return (lane >> 4) * 4 + i; }     // standard C/D map
};

// fp64 normal matrix (lower triangle) -> float copy for the single-precision factorisation
#ifndef SLSLAM_PO_FACTOR_ONLY
__global__ __launch_bounds__(256) void k_po_to_f32(PoPtrs p, float* Hf) {
  if (p.st->status != kRunning) return;
  const long long total = (long long)p.n * p.ld;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long long)gridDim.x * blockDim.x)
    Hf[q] = (float)p.H[q];
}

#endif
// Factor the diagonal block A[k0:k0+nb, k0:k0+nb] in LDS, write L11 back and its inverse to linv.  One workgroup of four
// waves, the 64 x 64 block as 4 x 4 tiles of 16: per tile column the diagonal tile is factored in the registers of wave 0
// (dense_tile.h: row per lane, DPP broadcasts, the tile's inverse from the same sweep), the panel below it
// (L_ik = A_ik X_kk^T) and the trailing update run on the 16x16x4 MFMA of T, one tile per wave at a time.  The inverse of
// the whole block (TRSM of the panel kernel and the substitution use it as a matrix) is then assembled tile by tile,
// X_ij = -X_ii sum_k L_ik X_kj, wave <-> tile column, the partial sums staged in the block's unused upper triangle.
// The 64 x 64 block in LDS (Tt: lower triangle, identity below the matrix's last row; Li: zero) -> its Cholesky factor in Tt and
// the factor's inverse in Li.  256 threads; the caller synchronises before and after.
template <typename T>
__device__ __forceinline__ void po_potrf_lds(PoPtrs& p, T* Tt, T* Li, int tid) {
  const int wave = tid >> 6, lane = tid & 63;
  int fail = 0;
  const int am = lane & 15, ak = lane >> 4, col = lane & 15;      // MFMA operand / result coordinates of this lane
  for (int kb = 0; kb < 4; ++kb) {
    T* D = Tt + (16 * kb) * kLdT + 16 * kb;
    T* X = Li + (16 * kb) * kLdT + 16 * kb;
    if (wave == 0) diag_tile_factor<T>(D, kLdT, lane, fail, [&](int r, int c, T v) { X[r * kLdT + c] = v; });
    __syncthreads();
    for (int i = kb + 1 + wave; i < 4; i += 4) {                  // panel: L(i,kb) = A(i,kb) X_kk^T
      T* P = Tt + (16 * i) * kLdT + 16 * kb;
      typename Mfma<T>::acc_t acc = Mfma<T>::zero();
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc = Mfma<T>::mac(P[am * kLdT + 4 * s4 + ak], X[am * kLdT + 4 * s4 + ak], acc);   // B[k][n] = X[n][k]
#pragma unroll
      for (int q = 0; q < 4; ++q) P[Mfma<T>::row(lane, q) * kLdT + col] = acc[q];
    }
    __syncthreads();
    int tile = 0;                                                 // trailing update: A(i,j) -= L(i,kb) L(j,kb)^T, kb < j <= i
    for (int i = kb + 1; i < 4; ++i)
      for (int j = kb + 1; j <= i; ++j, ++tile) {
        if ((tile & 3) != wave) continue;
        T* C = Tt + (16 * i) * kLdT + 16 * j;
        const T* Pi = Tt + (16 * i) * kLdT + 16 * kb;
        const T* Pj = Tt + (16 * j) * kLdT + 16 * kb;
        typename Mfma<T>::acc_t acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = C[Mfma<T>::row(lane, q) * kLdT + col];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = Mfma<T>::mac(-Pi[am * kLdT + 4 * s4 + ak], Pj[am * kLdT + 4 * s4 + ak], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) C[Mfma<T>::row(lane, q) * kLdT + col] = acc[q];
      }
    __syncthreads();
  }
  if (__syncthreads_or(fail)) { if (tid == 0) p.flags[0] = 1; }
  // inverse of the block: wave j assembles tile column j below the diagonal, top down (X_ij needs X_kj, j <= k < i)
  if (wave < 3) {
    const int j = wave;
    for (int i = j + 1; i < 4; ++i) {
      typename Mfma<T>::acc_t acc = Mfma<T>::zero();
      for (int k = j; k < i; ++k) {
        const T* Lik = Tt + (16 * i) * kLdT + 16 * k;
        const T* Xkj = Li + (16 * k) * kLdT + 16 * j;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = Mfma<T>::mac(Lik[am * kLdT + 4 * s4 + ak], Xkj[(4 * s4 + ak) * kLdT + am], acc);   // B[k][n] = X_kj[k][n]
      }
      T* S = Tt + (16 * j) * kLdT + 16 * i;                       // scratch: tile (j, i) of the block's upper triangle
#pragma unroll
      for (int q = 0; q < 4; ++q) S[Mfma<T>::row(lane, q) * kLdT + col] = acc[q];
      const T* Xii = Li + (16 * i) * kLdT + 16 * i;
      typename Mfma<T>::acc_t acc2 = Mfma<T>::zero();
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) acc2 = Mfma<T>::mac(-Xii[am * kLdT + 4 * s4 + ak], S[(4 * s4 + ak) * kLdT + am], acc2);
      T* Xij = Li + (16 * i) * kLdT + 16 * j;
#pragma unroll
      for (int q = 0; q < 4; ++q) Xij[Mfma<T>::row(lane, q) * kLdT + col] = acc2[q];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_po_potrf_diag(PoPtrs p, T* A, T* linv, int k0, T* Aout = nullptr) {     // Aout: where the factor goes (default: in place)
  if (p.st->status != kRunning) return;
  if (!Aout) Aout = A;
  __shared__ T Tt[kNB * kLdT];
  __shared__ T Li[kNB * kLdT];
  const int tid = threadIdx.x;
  const int nb = min((int)kNB, p.n - k0);
  for (int q = tid; q < kNB * kNB; q += 256) {
    const int r = q / kNB, c = q - r * kNB;
    T v = (r == c) ? T(1) : T(0);
    if (r < nb && c <= r) v = A[(long long)(k0 + r) * p.ld + k0 + c];
    Tt[r * kLdT + c] = (c <= r) ? v : T(0);
    Li[r * kLdT + c] = T(0);
  }
  __syncthreads();
  po_potrf_lds<T>(p, Tt, Li, tid);
  __syncthreads();
  for (int q = tid; q < kNB * kNB; q += 256) {
    const int r = q / kNB, c = q - r * kNB;
    linv[q] = (c <= r) ? Li[r * kLdT + c] : T(0);
    if (r < nb && c <= r) Aout[(long long)(k0 + r) * p.ld + k0 + c] = Tt[r * kLdT + c];
  }
}

// C(64x64) = B(64x64) * M(64x64)^T on the 16x16x4 MFMA of T; 4 waves, wave w owns tile row w.
template <typename T>
__device__ __forceinline__ void tile_mul_bt(const T* Bs, const T* Ms, int wave, int lane, typename Mfma<T>::acc_t acc[4]) {
  for (int tc = 0; tc < 4; ++tc) acc[tc] = Mfma<T>::zero();
  const int rr = lane & 15, kk = lane >> 4;
  for (int k4 = 0; k4 < kNB / 4; ++k4) {
    const T a = Bs[(wave * 16 + rr) * kLdT + k4 * 4 + kk];
#pragma unroll
    for (int tc = 0; tc < 4; ++tc) {
      const T b = Ms[(tc * 16 + rr) * kLdT + k4 * 4 + kk];
      acc[tc] = Mfma<T>::mac(a, b, acc[tc]);
    }
  }
}

// mode 0 (TRSM):  A[rows_i, k0:k0+64] <- A[rows_i, k0:k0+64] * inv(L11)^T        grid.x = row blocks
// mode 1 (SYRK):  A[rows_i, rows_j]   -= L[rows_i, k0:] * L[rows_j, k0:]^T        grid.x = lower tile pairs
template <typename T>
__global__ __launch_bounds__(256) void k_po_panel_update(PoPtrs p, T* A, const T* linv, int k0, int mode) {
  if (p.st->status != kRunning) return;
  __shared__ T Bs[kNB * kLdT];
  __shared__ T Ms[kNB * kLdT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int t0 = k0 + kNB;                       // first trailing row
  int bi, bj = 0;
  if (mode == 0) bi = blockIdx.x;
  else {                                          // unpack lower-triangular pair index
    int t = blockIdx.x;
    bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi * (bi + 1)) / 2 > t) --bi;
    while (((bi + 1) * (bi + 2)) / 2 <= t) ++bi;
    bj = t - (bi * (bi + 1)) / 2;
  }
  const int ri = t0 + bi * kNB, rj = t0 + bj * kNB;
  for (int q = tid; q < kNB * kNB; q += 256) {
    const int r = q / kNB, c = q - r * kNB;
    const bool cin = k0 + c < p.n;
    Bs[r * kLdT + c] = (ri + r < p.n && cin) ? A[(long long)(ri + r) * p.ld + k0 + c] : T(0);
    if (mode == 0) Ms[r * kLdT + c] = linv[q];
    else Ms[r * kLdT + c] = (rj + r < p.n && cin) ? A[(long long)(rj + r) * p.ld + k0 + c] : T(0);
  }
  __syncthreads();
  typename Mfma<T>::acc_t acc[4];
  tile_mul_bt<T>(Bs, Ms, wave, lane, acc);
  const int col = lane & 15;
  for (int tc = 0; tc < 4; ++tc)
    for (int i = 0; i < 4; ++i) {
      const int r = ri + wave * 16 + Mfma<T>::row(lane, i);
      if (r >= p.n) continue;
      if (mode == 0) {
        const int c = k0 + tc * 16 + col;
        if (c < p.n) A[(long long)r * p.ld + c] = acc[tc][i];
      } else {
        const int c = rj + tc * 16 + col;
        if (c < p.n && c <= r) A[(long long)r * p.ld + c] -= acc[tc][i];
      }
    }
}

// One block step of the right-looking factorisation in ONE launch (the launch chain above takes three: potrf_diag, TRSM, SYRK).
// Given the factored diagonal block k and its inverse (linv_all[bk]), workgroup (bi, bj), bj <= bi, of the trailing matrix
// (the factor goes to a matrix of its own, Lf: the raw panel blocks A(r, k) are read by every workgroup of block row / column r
// of this launch, so the one that holds their final value may not write it over them)
//   * solves the two panel blocks it needs itself, L_i = A(r_i, k) inv(L_kk)^T and L_j likewise (the TRSM of a 64 x 64 block costs one
//     tile product: recomputing it per tile triples the arithmetic of a step - 1554^3 flops in all, nothing next to two launches
//     and their gaps on the critical path of every step) - the diagonal workgroups (bi == bj) write L_i, the panel's final value;
//   * subtracts L_i L_j^T from its tile;
//   * and the workgroup of the NEXT diagonal block (bi == bj == 0) goes on to factor it (po_potrf_lds) and to leave its inverse in
//     linv_all[bk + 1]: the next launch starts from there.
// Dynamic LDS: three 64 x 66 tiles of T.
template <typename T>
__global__ __launch_bounds__(256) void k_po_step(PoPtrs p, T* A, T* Lf, T* linv_all, int bk) {
  if (p.st->status != kRunning) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char po_step_smem[];
  T* Bi = reinterpret_cast<T*>(po_step_smem);
  T* Bj = Bi + kNB * kLdT;
  T* Ms = Bj + kNB * kLdT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int k0 = bk * kNB, t0 = k0 + kNB;
  const T* linv = linv_all + (size_t)bk * kNB * kNB;
  int bi, bj;
  {
    const int t = blockIdx.x;
    bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((bi * (bi + 1)) / 2 > t) --bi;
    while (((bi + 1) * (bi + 2)) / 2 <= t) ++bi;
    bj = t - (bi * (bi + 1)) / 2;
  }
  const int ri = t0 + bi * kNB, rj = t0 + bj * kNB;
  const bool diag = bi == bj;
  for (int q = tid; q < kNB * kNB; q += 256) {
    const int r = q / kNB, c = q - r * kNB;
    const bool cin = k0 + c < p.n;
    Bi[r * kLdT + c] = (ri + r < p.n && cin) ? A[(long long)(ri + r) * p.ld + k0 + c] : T(0);
    if (!diag) Bj[r * kLdT + c] = (rj + r < p.n && cin) ? A[(long long)(rj + r) * p.ld + k0 + c] : T(0);
    Ms[r * kLdT + c] = linv[q];
  }
  __syncthreads();
  typename Mfma<T>::acc_t li[4], lj[4];
  tile_mul_bt<T>(Bi, Ms, wave, lane, li);
  if (!diag) tile_mul_bt<T>(Bj, Ms, wave, lane, lj);
  __syncthreads();
  const int col = lane & 15;
#pragma unroll
  for (int tc = 0; tc < 4; ++tc)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = wave * 16 + Mfma<T>::row(lane, i), cc = tc * 16 + col;
      Bi[rr * kLdT + cc] = li[tc][i];
      if (!diag) Bj[rr * kLdT + cc] = lj[tc][i];
      else if (ri + rr < p.n && k0 + cc < p.n) Lf[(long long)(ri + rr) * p.ld + k0 + cc] = li[tc][i];    // the panel block, final
    }
  __syncthreads();
  typename Mfma<T>::acc_t acc[4];
  tile_mul_bt<T>(Bi, diag ? Bi : Bj, wave, lane, acc);
  const bool next_diag = bi == 0 && bj == 0;                     // (wave-uniform, workgroup-uniform)
  T* Tt = Bj;                                                     // free from here on for the diagonal workgroups
  T* Li = Ms;
  if (next_diag) {
    for (int q = tid; q < kNB * kNB; q += 256) {
      const int r = q / kNB, c = q - r * kNB;
      Tt[r * kLdT + c] = (r == c) ? T(1) : T(0);
      Li[r * kLdT + c] = T(0);
    }
    __syncthreads();
  }
  for (int tc = 0; tc < 4; ++tc)
    for (int i = 0; i < 4; ++i) {
      const int r = ri + wave * 16 + Mfma<T>::row(lane, i), c = rj + tc * 16 + col;
      if (r >= p.n || c >= p.n || c > r) continue;
      const T v = A[(long long)r * p.ld + c] - acc[tc][i];
      if (next_diag) Tt[(r - ri) * kLdT + (c - rj)] = v;          // (written to memory after the factorisation)
      else A[(long long)r * p.ld + c] = v;
    }
  if (!next_diag) return;
  __syncthreads();
  po_potrf_lds<T>(p, Tt, Li, tid);
  __syncthreads();
  T* linv_next = linv_all + (size_t)(bk + 1) * kNB * kNB;
  const int nb = min((int)kNB, p.n - t0);
  for (int q = tid; q < kNB * kNB; q += 256) {
    const int r = q / kNB, c = q - r * kNB;
    linv_next[q] = (c <= r) ? Li[r * kLdT + c] : T(0);
    if (r < nb && c <= r) Lf[(long long)(t0 + r) * p.ld + t0 + c] = Tt[r * kLdT + c];
  }
}
enum { kPoStepLdsTiles = 3 };

// forward then backward substitution with the factor in A and the inverted diagonal blocks in
// linv_all (one 64x64 block per block column, kept by k_po_potrf_diag): every step is a parallel
// matrix-vector product - no serial pivot loop.  rhs / solution stay fp64.  One workgroup.
template <typename T>
__global__ __launch_bounds__(1024) void k_po_trisolve(PoPtrs p, const T* A, const T* linv_all) {
  if (p.st->status != kRunning) return;
  __shared__ double yb[kNB], yn[kNB];
  __shared__ T Ls[kNB * (kNB + 1)];
  const int tid = threadIdx.x, nthr = blockDim.x;     // 1024 threads: the matrix-vector updates are bound by loads in flight
  const int n = p.n;
  const int nblk = (n + kNB - 1) / kNB;
  // forward: y_k <- inv(L_kk) y_k ;  y_r -= L[r, k] y_k for the rows below
  for (int bk = 0; bk < nblk; ++bk) {
    const int k0 = bk * kNB, nb = min((int)kNB, n - k0);
    const T* Li = linv_all + (size_t)bk * kNB * kNB;
    for (int q = tid; q < kNB * kNB; q += nthr) Ls[(q / kNB) * (kNB + 1) + (q % kNB)] = Li[q];    // coalesced
    if (tid < kNB) yb[tid] = tid < nb ? p.y[k0 + tid] : 0.0;
    __syncthreads();
    // sixteen threads per row (a row of the factor is contiguous in memory: coalesced loads), partial dot products combined
    // over the sixteen with a fixed butterfly; 64 rows per pass of the 1024 threads
    const int rsub = tid >> 4, csub = tid & 15, rows_per_pass = nthr >> 4;
    for (int r = rsub; r < kNB; r += rows_per_pass) {
      double s = 0.0;
      if (r < nb) for (int j = csub; j <= r; j += 16) s += (double)Ls[r * (kNB + 1) + j] * yb[j];
      s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
      if (r < nb && csub == 0) { yn[r] = s; p.y[k0 + r] = s; }
    }
    __syncthreads();
    for (int r0 = k0 + nb; r0 < n; r0 += rows_per_pass) {
      const int r = r0 + rsub;
      double s = 0.0;
      if (r < n) for (int j = csub; j < nb; j += 16) s += (double)A[(long long)r * p.ld + k0 + j] * yn[j];
      s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
      if (r < n && csub == 0) p.y[r] -= s;
    }
    __syncthreads();
  }
  // backward: y_k <- inv(L_kk)^T y_k ;  y_r -= L[k, r]^T y_k for the rows above
  for (int bk = nblk - 1; bk >= 0; --bk) {
    const int k0 = bk * kNB, nb = min((int)kNB, n - k0);
    const T* Li = linv_all + (size_t)bk * kNB * kNB;
    __syncthreads();
    for (int q = tid; q < kNB * kNB; q += nthr) Ls[(q / kNB) * (kNB + 1) + (q % kNB)] = Li[q];
    if (tid < kNB) yb[tid] = tid < nb ? p.y[k0 + tid] : 0.0;
    __syncthreads();
    {
      const int rsub = tid >> 4, csub = tid & 15, rows_per_pass = nthr >> 4;
      for (int r = rsub; r < kNB; r += rows_per_pass) {               // row r of inv(L_kk)^T: column r of the stored inverse
        double s = 0.0;
        if (r < nb) for (int j = r + csub; j < nb; j += 16) s += (double)Ls[j * (kNB + 1) + r] * yb[j];
        s += __shfl_xor(s, 8, 16); s += __shfl_xor(s, 4, 16); s += __shfl_xor(s, 2, 16); s += __shfl_xor(s, 1, 16);
        if (r < nb && csub == 0) { yn[r] = s; p.y[k0 + r] = s; }
      }
    }
    __syncthreads();
    for (int r = tid; r < k0; r += nthr) {
      double s = 0.0;
      for (int j = 0; j < nb; ++j) s += (double)A[(long long)(k0 + j) * p.ld + r] * yn[j];
      p.y[r] -= s;
    }
    __syncthreads();
  }
}

// The same two substitutions spread over the chip: one workgroup per 64-row block, all resident (the host checks), forward
// then backward in ONE launch.  k_po_trisolve walks the whole factor (10 MB at 1554 unknowns) with a single workgroup - bound by
// what one CU can load, 0.8 ms, half of a dense iteration.  Here workgroup r owns block row r:
//   forward   y_r = inv(L_rr) (b_r - sum_{k < r} L[r, k] y_k)       reads block row r of the factor, left to right
//   backward  x_r = inv(L_rr)^T (y_r - sum_{k > r} L[k, r]^T x_k)   reads block column r, bottom up
// and waits, block by block, for the owner of block k to publish its part (flag[k] = epoch of this launch: device-scope release /
// acquire; parts are read with device-scope loads).  The next block of the factor is requested before the wait, so a hop of the
// dependent chain costs the flag's trip plus one 64 x 64 matrix-vector product.  Sums are taken in a fixed order: the solution does
// not depend on the timing.  rhs / solution stay fp64.
template <typename T>
__global__ __launch_bounds__(256) void k_po_trisolve_wide(PoPtrs p, const T* A, const T* linv_all, unsigned* flags, unsigned epoch) {
  if (p.st->status != kRunning) return;
  __shared__ double yk[kNB], part[4][kNB], acc_s[kNB];
  const int tid = threadIdx.x, n = p.n, nblk = (n + kNB - 1) / kNB;
  const int r = blockIdx.x, r0 = r * kNB, nbr = min((int)kNB, n - r0);
  const T* Li = linv_all + (size_t)r * kNB * kNB;
  auto wait_for = [&](int k, unsigned e) {
    if (tid == 0) while (__hip_atomic_load(flags + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < e) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
  };
  auto publish = [&](unsigned e) {
    __syncthreads();
    if (tid == 0) { __threadfence(); __hip_atomic_store(flags + r, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
  };
  auto load_part = [&](int k) {                       // y_k / x_k of block k into LDS (zeros past the matrix)
    if (tid < kNB) yk[tid] = (k * kNB + tid < n) ? __hip_atomic_load(p.y + k * kNB + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    __syncthreads();
  };
  // ---- forward: four threads per row of the block, sixteen columns each
  {
    const int row = tid >> 2, q4 = tid & 3;
    double acc = 0.0;
    T cur[16] = {}, nxt[16] = {};
    auto fetch = [&](int k, T (&v)[16]) {
      const bool ok = r0 + row < n && k < r;
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (ok && k * kNB + 16 * q4 + j < n) ? A[(long long)(r0 + row) * p.ld + k * kNB + 16 * q4 + j] : T(0);
    };
    if (r > 0) fetch(0, cur);
    for (int k = 0; k < r; ++k) {
      if (k + 1 < r) fetch(k + 1, nxt);
      wait_for(k, epoch);
      load_part(k);
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += (double)cur[j] * yk[16 * q4 + j];
      s += __shfl_xor(s, 1, 4); s += __shfl_xor(s, 2, 4);
      acc += s;                                        // (block after block, in order)
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[j] = nxt[j];
    }
    if (q4 == 0) acc_s[row] = (r0 + row < n) ? p.y[r0 + row] - acc : 0.0;
    __syncthreads();
    double s = 0.0;                                    // y_r = inv(L_rr) acc_s : row `row` of the inverse, columns <= row
    if (row < nbr) for (int j = 16 * q4; j < 16 * q4 + 16 && j <= row; ++j) s += (double)Li[row * kNB + j] * acc_s[j];
    s += __shfl_xor(s, 1, 4); s += __shfl_xor(s, 2, 4);
    __syncthreads();
    if (q4 == 0 && row < nbr) __hip_atomic_store(p.y + r0 + row, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    publish(epoch);
  }
  // ---- backward: thread (g, j) takes column j of the block and sixteen of its rows; the four groups meet in LDS
  {
    const int col = tid & 63, g = tid >> 6;
    double acc = 0.0;
    T cur[16] = {}, nxt[16] = {};
    auto fetch = [&](int k, T (&v)[16]) {
      const bool ok = r0 + col < n && k > r && k < nblk;
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = (ok && k * kNB + 16 * g + i < n) ? A[(long long)(k * kNB + 16 * g + i) * p.ld + r0 + col] : T(0);
    };
    if (r + 1 < nblk) fetch(nblk - 1, cur);
    for (int k = nblk - 1; k > r; --k) {
      if (k - 1 > r) fetch(k - 1, nxt);
      wait_for(k, epoch + 1u);
      load_part(k);
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) s += (double)cur[i] * yk[16 * g + i];
      part[g][col] = s;
      __syncthreads();
      if (g == 0) acc += (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
    }
    // every block has finished its forward part before any backward part is published (the last block starts the backward
    // chain only after its own forward part, which needed all the others)
    if (g == 0) acc_s[col] = (r0 + col < n) ? p.y[r0 + col] - acc : 0.0;
    __syncthreads();
    double s = 0.0;                                    // x_r = inv(L_rr)^T acc_s : column `col` of the inverse, rows >= col
    if (col < nbr) for (int i = max(col, 16 * g); i < 16 * g + 16 && i < nbr; ++i) s += (double)Li[i * kNB + col] * acc_s[i];
    part[g][col] = s;
    __syncthreads();
    if (g == 0 && col < nbr) __hip_atomic_store(p.y + r0 + col, (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    publish(epoch + 1u);
  }
}

#ifndef SLSLAM_PO_FACTOR_ONLY
// ------------------------------------------------------------------------------------------
// Structured factorisation (default): a pose graph is chains of odometry edges tied together by a few
// loop closures.  The host orders the unknowns chains first (each chain's poses consecutive, in path
// order), junction poses (degree >= 3) last; then
//   k_po_chain_eliminate   one wave per chain: block-tridiagonal elimination along the chain with the
//                          fill towards the chain's (at most two) junctions carried along; the Schur
//                          complement contributions go to the junction block of the same dense matrix
//   dense blocked Cholesky (the MFMA kernels above) of the junction block only (6 * #junctions unknowns)
//   k_po_chain_backsub     one wave per chain: back-substitution from the junction solution
// In place on the dense lower-triangular storage (room for the fill), so linearisation, damping and
// the LM bookkeeping are unchanged.  Chains are independent: a 260-pose graph with 8 loop closures is
// ~17 chains of ~15 poses eliminated concurrently + one ~100 x 100 dense system, instead of 25
// dependent 64-wide block steps over a 1554 x 1554 matrix.
struct PoChain { int start, len, jl, jr; };   // offsets in the reduced vector; jl / jr = -1 when the end is free

__device__ __forceinline__ bool chol6_and_inverse(const double* Ds /* LDS, 6x6 symmetric */, double L[21], double Li[21]) {
  // packing: (r, c) -> r (r + 1) / 2 + c, c <= r.  1/sqrt by v_rsq_f64 + refinement: no divide on the chain.
  bool ok = true;
  double invd[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double sacc = Ds[6 * r + c];
#pragma unroll
      for (int k = 0; k < c; ++k) sacc -= L[(r * (r + 1)) / 2 + k] * L[(c * (c + 1)) / 2 + k];
      if (c == r) {
        if (!(sacc > 0.0) || !isfinite(sacc)) { ok = false; sacc = 1.0; }
        invd[r] = rsqrt(sacc);
        L[(r * (r + 1)) / 2 + r] = sacc * invd[r];
      } else {
        L[(r * (r + 1)) / 2 + c] = sacc * invd[c];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {          // column c of the inverse
#pragma unroll
    for (int r = c; r < 6; ++r) {
      double sacc = (r == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = c; k < r; ++k) sacc -= L[(r * (r + 1)) / 2 + k] * Li[(k * (k + 1)) / 2 + c];
      Li[(r * (r + 1)) / 2 + c] = sacc * invd[r];
    }
  }
  return ok;
}

__global__ __launch_bounds__(64) void k_po_chain_eliminate(PoPtrs p, const PoChain* chains) {
  if (p.st->status != kRunning) return;
  __shared__ double Ds[36], Bs[36], Cs[36], Rs[36], Lis[36], Bt[36], Ct[36], Rt[36], gs[6], gt[6];
  const PoChain ch = chains[blockIdx.x];
  const int lane = threadIdx.x;
  const int r = lane / 6, c = lane - 6 * r;
  const bool el = lane < 36;
  const long long ld = p.ld;
  double* H = p.H;
  double accLL = 0.0, accRR = 0.0, accRL = 0.0, accgL = 0.0, accgR = 0.0;
  // state of the node being eliminated: D (symmetric), C (coupling to the left junction), B / R (coupling to
  // the next chain node / the right junction), g.  Everything the NEXT node needs from memory is requested
  // at the top of a step and consumed at its end.
  double dcur = 0.0, ccur = 0.0, gcur = 0.0, bcur = 0.0, rcur = 0.0;
  if (el) {
    const int s0 = ch.start;
    dcur = H[(s0 + max(r, c)) * ld + s0 + min(r, c)];
    ccur = ch.jl >= 0 ? H[(ch.jl + r) * ld + s0 + c] : 0.0;
    if (ch.len > 1) bcur = H[(s0 + 6 + r) * ld + s0 + c];
    else if (ch.jr >= 0) rcur = H[(ch.jr + r) * ld + s0 + c];
  }
  if (lane < 6) gcur = p.y[ch.start + lane];
  int fail = 0;
  for (int i = 0; i < ch.len; ++i) {
    const int si = ch.start + 6 * i, sn = si + 6;
    const bool last = i == ch.len - 1, next_last = i == ch.len - 2;
    double dnext = 0.0, gnext = 0.0, bnext = 0.0, rnext = 0.0;
    if (el && !last) {
      dnext = H[(sn + max(r, c)) * ld + sn + min(r, c)];
      if (!next_last) bnext = H[(sn + 6 + r) * ld + sn + c];
      else if (ch.jr >= 0) rnext = H[(ch.jr + r) * ld + sn + c];
    }
    if (lane < 6 && !last) gnext = p.y[sn + lane];
    if (el) { Ds[lane] = dcur; Bs[lane] = bcur; Cs[lane] = ccur; Rs[lane] = rcur; }
    if (lane < 6) gs[lane] = gcur;
    __syncthreads();
    double L[21], Li[21];
    if (!chol6_and_inverse(Ds, L, Li)) fail = 1;       // every lane, redundantly: no cross-lane traffic
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int q = 0; q < 6; ++q) Lis[6 * a + q] = q <= a ? Li[(a * (a + 1)) / 2 + q] : 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int q = 0; q <= a; ++q) H[(si + a) * ld + si + q] = L[(a * (a + 1)) / 2 + q];   // keep the factor
    }
    __syncthreads();
    // X~ = X L^-T :  X~[r][c] = sum_k X[r][k] Linv[c][k]
    if (el) {
      double bt = 0.0, ct = 0.0, rt = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { const double li = Lis[6 * c + k]; bt += Bs[6 * r + k] * li; ct += Cs[6 * r + k] * li; rt += Rs[6 * r + k] * li; }
      Bt[lane] = bt; Ct[lane] = ct; Rt[lane] = rt;
      if (!last) H[(sn + r) * ld + si + c] = bt;
      else if (ch.jr >= 0) H[(ch.jr + r) * ld + si + c] = rt;
      if (ch.jl >= 0) H[(ch.jl + r) * ld + si + c] = ct;
    }
    if (lane < 6) {
      double gg = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) gg += Lis[6 * lane + k] * gs[k];
      gt[lane] = gg;
      p.y[si + lane] = gg;
    }
    __syncthreads();
    if (el) {
      double dd = 0.0, cc = 0.0, ll = 0.0, r2 = 0.0, rl = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        dd += Bt[6 * r + k] * Bt[6 * c + k];
        cc += Ct[6 * r + k] * Bt[6 * c + k];
        ll += Ct[6 * r + k] * Ct[6 * c + k];
        r2 += Rt[6 * r + k] * Rt[6 * c + k];
        rl += Rt[6 * r + k] * Ct[6 * c + k];
      }
      dcur = dnext - dd;         // D_{i+1} - B~ B~^T
      ccur = -cc;                // fill: H(jl, v_{i+1}) = - C~ B~^T  (an interior chain node has no edge to a junction)
      bcur = bnext; rcur = rnext;
      accLL += ll; accRR += r2; accRL += rl;
    }
    if (lane < 6) {
      double gb = 0.0, gl = 0.0, gr = 0.0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { gb += Bt[6 * lane + k] * gt[k]; gl += Ct[6 * lane + k] * gt[k]; gr += Rt[6 * lane + k] * gt[k]; }
      gcur = gnext - gb;
      accgL += gl; accgR += gr;
    }
    __syncthreads();
  }
  // Schur complement contributions of this chain to the junction system (several chains meet at a junction: atomics)
  if (el) {
    if (ch.jl >= 0 && r >= c) atomicAdd(&H[(ch.jl + r) * ld + ch.jl + c], -accLL);
    if (ch.jr >= 0 && r >= c) atomicAdd(&H[(ch.jr + r) * ld + ch.jr + c], -accRR);
    if (ch.jl >= 0 && ch.jr >= 0) {
      if (ch.jr > ch.jl) atomicAdd(&H[(ch.jr + r) * ld + ch.jl + c], -accRL);
      else atomicAdd(&H[(ch.jl + c) * ld + ch.jr + r], -accRL);
    }
  }
  if (lane < 6) {
    if (ch.jl >= 0) atomicAdd(&p.y[ch.jl + lane], -accgL);
    if (ch.jr >= 0) atomicAdd(&p.y[ch.jr + lane], -accgR);
  }
  if (__any(fail) && lane == 0) p.flags[0] = 1;
}

// y_i = L_i^-T ( g~_i - B~_i^T y_{i+1} - C~_i^T y_jl - [last] R~^T y_jr ), from the end of the chain to its start.
// The three 6x6 blocks of the next step are requested while the current one is solved.
__global__ __launch_bounds__(64) void k_po_chain_backsub(PoPtrs p, const PoChain* chains) {
  if (p.st->status != kRunning) return;
  __shared__ double Ls[36], Xs[36], Cs[36], gsh[6], rhs[6], ysol[6], ylr[12], idiag[6];
  const PoChain ch = chains[blockIdx.x];
  const int lane = threadIdx.x;
  const int r = lane / 6, c = lane - 6 * r;
  const bool el = lane < 36;
  const long long ld = p.ld;
  const double* H = p.H;
  if (lane < 6) { ylr[lane] = ch.jl >= 0 ? p.y[ch.jl + lane] : 0.0; ylr[6 + lane] = ch.jr >= 0 ? p.y[ch.jr + lane] : 0.0; ysol[lane] = 0.0; }
  auto load = [&](int i, double& l, double& x, double& cc, double& g) {
    const int si = ch.start + 6 * i, sn = si + 6;
    const bool last = i == ch.len - 1;
    l = x = cc = g = 0.0;
    if (el) {
      l = r >= c ? H[(si + r) * ld + si + c] : 0.0;
      if (!last) x = H[(sn + r) * ld + si + c];                 // B~[r][c]
      else if (ch.jr >= 0) x = H[(ch.jr + r) * ld + si + c];    // R~[r][c]
      if (ch.jl >= 0) cc = H[(ch.jl + r) * ld + si + c];        // C~[r][c]
    }
    if (lane < 6) g = p.y[si + lane];
  };
  double l, x, cc, g;
  load(ch.len - 1, l, x, cc, g);
  for (int i = ch.len - 1; i >= 0; --i) {
    const int si = ch.start + 6 * i;
    const bool last = i == ch.len - 1;
    if (el) { Ls[lane] = l; Xs[lane] = x; Cs[lane] = cc; }
    if (lane < 6) gsh[lane] = g;
    if (i > 0) load(i - 1, l, x, cc, g);                        // in flight during this step
    __syncthreads();
    if (lane < 6) {
      // rhs[a] = g~[a] - sum_k ( X[k][a] yv[k] + C~[k][a] yl[k] ),  yv = y_{i+1} or y_jr
      double sacc = gsh[lane];
#pragma unroll
      for (int k = 0; k < 6; ++k) sacc -= Xs[6 * k + lane] * (last ? ylr[6 + k] : ysol[k]) + Cs[6 * k + lane] * ylr[k];
      rhs[lane] = sacc;
      idiag[lane] = 1.0 / Ls[7 * lane];
    }
    __syncthreads();
    if (lane == 0) {                              // L^T y = rhs, 6 unknowns
      double y[6];
#pragma unroll
      for (int a = 5; a >= 0; --a) {
        double t = rhs[a];
#pragma unroll
        for (int k = a + 1; k < 6; ++k) t -= Ls[6 * k + a] * y[k];
        y[a] = t * idiag[a];
      }
#pragma unroll
      for (int a = 0; a < 6; ++a) ysol[a] = y[a];
    }
    __syncthreads();
    if (lane < 6) p.y[si + lane] = ysol[lane];
  }
}

// candidate poses and step statistics; one workgroup.
__global__ __launch_bounds__(256) void k_po_candidate(PoPtrs p) {
  LMState* st = p.st;
  if (st->status != kRunning) return;
  __shared__ double red[3][256];
  const int tid = threadIdx.x;
  const double* X = p.x + (long long)st->cur * 6 * p.N;
  double* Xc = p.x + (long long)(1 - st->cur) * 6 * p.N;
  double model = 0.0, dn2 = 0.0, xn2 = 0.0;
  int bad = 0;
  for (int k = tid; k < p.N; k += 256) {
    // (all loads of a pose first, the conditional ones from a safe index: behind the stores of the candidate and under the
    // condition they would be one dependent round trip per component)
    const int s = p.slot[k], ss = s >= 0 ? s : 0;
    double xv[6], yv[6], gv[6], dv[6], sv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { xv[i] = X[6 * k + i]; yv[i] = p.y[ss + i]; gv[i] = p.g[ss + i]; dv[i] = p.d2[ss + i]; sv[i] = p.scale[ss + i]; }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      double v = xv[i];
      if (s >= 0) {
        const double y = yv[i];
        if (!isfinite(y)) bad = 1;
        model += 0.5 * y * (gv[i] + dv[i] * y);
        const double xn = v - y * sv[i];
        const double dd = v - xn;
        dn2 += dd * dd; xn2 += xn * xn;
        v = xn;
      }
      Xc[6 * k + i] = v;
    }
  }
  red[0][tid] = model; red[1][tid] = dn2; red[2][tid] = xn2;
  const int any_bad = __syncthreads_or(bad);
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) for (int q = 0; q < 3; ++q) red[q][tid] += red[q][tid + o];
    __syncthreads();
  }
  if (tid == 0) {
    p.scal[kPoModel] = red[0][0]; p.scal[kPoDn2] = red[1][0]; p.scal[kPoXn2] = red[2][0];
    st->solve_failed = (any_bad || p.flags[0]) ? 1 : 0;
  }
}

// trust-region bookkeeping (one thread); same policy as k_lm_update.
__global__ void k_po_update(PoPtrs p, Policy pol) {
  LMState* st = p.st;
  if (st->status != kRunning || threadIdx.x != 0 || blockIdx.x != 0) return;
  double new_cost = p.scal[kPoCandCost];
  const double model = p.scal[kPoModel], cost = st->cost;
  IterRec rec;
  rec.pad = 0; rec.iteration = st->iter + 1; rec.step_is_valid = 0; rec.step_is_successful = 0;
  rec.model_cost_change = model; rec.cost_change = 0; rec.step_norm = 0; rec.relative_decrease = 0;
  rec.gradient_max_norm = st->grad_max;
  const bool valid = !st->solve_failed && !(model < 0.0);
  if (!isfinite(new_cost)) new_cost = 1.7976931348623157e308;
  if (!valid) {
    if (++st->n_invalid >= pol.max_invalid) { st->status = 4; return; }
  } else {
    st->n_invalid = 0;
    rec.step_is_valid = 1;
    rec.step_norm = sqrt(p.scal[kPoDn2]);
    if (rec.step_norm <= pol.parameter_tolerance * (st->x_norm + pol.parameter_tolerance)) { st->status = 3; return; }
    rec.cost_change = cost - new_cost;
    if (fabs(rec.cost_change) < pol.function_tolerance * cost) { st->status = 2; return; }
    rec.relative_decrease = rec.cost_change / model;
    rec.step_is_successful = rec.relative_decrease > pol.min_relative_decrease;
  }
  if (rec.step_is_successful) {
    st->n_success++;
    const double q = 2.0 * rec.relative_decrease - 1.0;
    double f = 1.0 - q * q * q;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    st->radius = fmin(st->radius / f, pol.max_radius);
    st->decrease_factor = 2.0;
    st->cur = 1 - st->cur;
    st->cost = new_cost;
    st->x_norm = sqrt(p.scal[kPoXn2]);
    st->need_grad_check = 1;
  } else {
    st->n_unsuccess++;
    if (rec.step_is_valid) { st->radius = st->radius / st->decrease_factor; st->decrease_factor *= 2.0; }
    else st->radius *= 0.5;
  }
  rec.cost = st->cost + st->fixed_cost;
  rec.trust_region_radius = st->radius;
  if (rec.cost < st->min_cost) st->min_cost = rec.cost;
  if (st->ntrace < kMaxTrace) p.trace[st->ntrace] = rec;
  st->ntrace++;
  st->iter = rec.iteration;
  if (st->radius < pol.min_radius) { st->status = 5; return; }
  if (st->iter >= pol.max_num_iterations) { st->status = 0; return; }
}

#endif
}  // namespace slslam
#endif  // SLSLAM_PO_KERNELS_H_
