// slslam_amd/csrc/lba_big.h — the line bundle adjustment for windows beyond what the tiled sweeps hold on chip: more than
// 20 free or 64 cameras, or a line observed by more than 64 keyframes.  The reference makes the window size a flag
// (src/main.cpp:22 ba_window_size) and its study runs W = 40 (80 keyframes, 40 free: BASELINE.md section 1), where the
// reduced camera system (240^2) no longer fits one wave's LDS partial or one workgroup's Cholesky.
//
// Same algorithm, same LM bookkeeping (lm_step, k_lm_update), same per-observation arithmetic (lba_math.h); what changes is
// where the sums live: everything accumulates in HBM / L2 with fp64 global atomics, one thread per observation / line /
// camera pair, and the reduced system is factored by the pose-graph path's blocked MFMA Cholesky (po_kernels.h:
// k_po_potrf_diag / k_po_panel_update / k_po_trisolve on v_mfma_f64_16x16x4_f64).  Written for coverage of the
// reference's parameter range, not for speed: sums by atomics are not bitwise reproducible run to run.
//
// One LM iteration:  k_big_cameras(0) -> k_big_linearise -> k_big_line -> k_big_rescale -> k_big_schur -> k_big_prepare ->
// [potrf / panel updates / trisolve per window] -> k_big_finish -> k_big_cameras(1) -> k_big_backsub_obs ->
// k_big_backsub_line -> k_big_cost -> k_lm_update.
#ifndef SLSLAM_LBA_BIG_H_
#define SLSLAM_LBA_BIG_H_

#include "lba_kernels.h"

namespace slslam {

enum { kBigObs = 46 };         // doubles kept per observation: Jc[24] | Jl[16] | rs[4] | cost | pad
enum { kBigLine = 18 };        // per-line accumulators: H[10] | g[4] | w[4] (back-substitution)
enum { kBigCam = 21 };         // R[9] JL[9] t[3] per camera and buffer (accepted, candidate)
enum { kBgCost = 0, kBgFixed = 1, kBgGmaxLine = 2, kBgXn2Line = 3, kBgFail = 4, kBgWasFresh = 5, kBgScal = 8 };

struct BigPtrs {
  const int* ob_line;          // [nobs] sorted line of every sorted observation
  const int* cam_win;          // [ncam]
  double* J;                   // [nobs][kBigObs]
  double* camtab;              // [ncam][2][kBigCam]
  double* line_acc;            // [nline][kBigLine]
  double* sys;                 // per window: S [n x ld] | b [n] | g [n] | h [n] | y [n]
  const long long* sys_off;    // [nwin]
  const int* pair_i;           // [npairs] observations (sorted index) i <= j of one free line, both of free cameras
  const int* pair_j;
  double* scal;                // [nwin][kBgScal]
  int* flags;                  // [nwin][2] factorisation failure (PoPtrs.flags)
  long long npairs, nobs;
};
__host__ __device__ inline int big_ld(int n) { return ((n + 7) / 8) * 8 + 8; }
__host__ __device__ inline long long big_sys_doubles(int n) { return (long long)(n > 0 ? n : 1) * big_ld(n) + 4LL * (n > 0 ? n : 1); }

__device__ __forceinline__ void atomic_max_nonneg(double* p, double v) {       // v >= 0: the bit patterns order like the values
  atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}

// thread <-> camera: rotation, SO(3) left Jacobian and translation of the accepted (which = 0) or candidate (1) pose
__global__ __launch_bounds__(256) void k_big_cameras(BatchPtrs p, BigPtrs bg, int which) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.ncam) return;
  const LMState* st = p.state + bg.cam_win[i];
  if (st->status != kRunning) return;
  const int buf = which ? 1 - st->cur : st->cur;
  const double* x = p.cam_x + ((long long)i * 2 + buf) * kCamRec;
  double w[3] = { x[0], x[1], x[2] }, R[9], JL[9];
  cam_prepare<double>(w, R, JL);
  double* ct = bg.camtab + ((long long)i * 2 + which) * kBigCam;
  for (int q = 0; q < 9; ++q) { ct[q] = R[q]; ct[9 + q] = JL[q]; }
  ct[18] = x[3]; ct[19] = x[4]; ct[20] = x[5];
}

// thread <-> observation: residual, Jacobians, Huber, scaling; kept per observation; line block, camera block, gradients
// and costs by atomics.  At the first sweep of a solve (fresh) all scales are 1 (see k_big_line / k_big_prepare).
__global__ __launch_bounds__(128) void k_big_linearise(BatchPtrs p, BigPtrs bg, Policy pol) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  const WinDesc wd = p.wins[w];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const bool fresh = st->fresh != 0;
  const int cam = wd.cam_off + p.ob_cam[o], cf = p.cam_cf[cam];
  const bool line_free = !(p.line_flags[ls] & 1);
  const bool kept = !(cf < 0 && !line_free);
  const double* ct = bg.camtab + (long long)cam * 2 * kBigCam;
  double R[9], JL[9], t[3], trig[7], ob[8];
  for (int q = 0; q < 9; ++q) { R[q] = ct[q]; JL[q] = ct[9 + q]; }
  for (int q = 0; q < 3; ++q) t[q] = ct[18 + q];
  const double* lrec = p.line_x + line_rec(p, ls, cur);
  for (int q = 0; q < 7; ++q) trig[q] = lrec[4 + q];
  for (int q = 0; q < 4; ++q) {
    const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
    ob[2 * q] = e.x; ob[2 * q + 1] = e.y;
  }
  double cp[3], dv[3], dcp[12], ddv[9], r[4], Jc[24], Jl[16], cost;
  line_points_jac<double>(trig, cp, dv, dcp, ddv);
  obs_linearise<double>(R, JL, t, cp, dv, dcp, ddv, ob, pol.baseline, r, Jc, Jl);
  const double sr = huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &cost);
  double* J = bg.J + o * kBigObs;
  for (int q = 0; q < 4; ++q) {
    r[q] *= sr;
    for (int a = 0; a < 6; ++a) Jc[6 * q + a] *= sr * (fresh || cf < 0 ? 1.0 : p.cam_scale[(long long)cam * 6 + a]);
    for (int a = 0; a < 4; ++a) Jl[4 * q + a] *= sr * (fresh ? 1.0 : p.line_scale[(long long)ls * 4 + a]);
  }
  for (int q = 0; q < 24; ++q) J[q] = Jc[q];
  for (int q = 0; q < 16; ++q) J[24 + q] = Jl[q];
  for (int q = 0; q < 4; ++q) J[40 + q] = r[q];
  J[44] = cost;
  double* sc = bg.scal + (long long)w * kBgScal;
  if (kept) atomicAdd(&sc[kBgCost], cost);
  else if (fresh) atomicAdd(&sc[kBgFixed], cost);
  if (line_free) {
    double* la = bg.line_acc + (long long)ls * kBigLine;
    int q = 0;
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b <= a; ++b, ++q) {
        double h = 0.0;
        for (int k = 0; k < 4; ++k) h += Jl[4 * k + a] * Jl[4 * k + b];
        atomicAdd(&la[q], h);
      }
    for (int a = 0; a < 4; ++a) {
      double g = 0.0;
      for (int k = 0; k < 4; ++k) g += Jl[4 * k + a] * r[k];
      atomicAdd(&la[10 + a], g);
    }
  }
  if (cf >= 0) {
    const int n = wd.n, ld = big_ld(n);
    double* S = bg.sys + bg.sys_off[w];
    double* bvec = S + (long long)n * ld;
    double* gvec = bvec + n;
    double* hvec = gvec + n;
    for (int a = 0; a < 6; ++a) {
      double ga = 0.0;
      for (int k = 0; k < 4; ++k) ga += Jc[6 * k + a] * r[k];
      atomicAdd(&bvec[6 * cf + a], ga);
      atomicAdd(&gvec[6 * cf + a], ga);
      for (int b = 0; b <= a; ++b) {
        double v = 0.0;
        for (int k = 0; k < 4; ++k) v += Jc[6 * k + a] * Jc[6 * k + b];
        atomicAdd(&S[(long long)(6 * cf + a) * ld + 6 * cf + b], v);
        if (a == b) atomicAdd(&hvec[6 * cf + a], v);
      }
    }
  }
}

// thread <-> line: Jacobi scale (first sweep), LM damping, 4x4 Cholesky, K = chol^-1, u = K g; kept for the Schur products
// and the back-substitution (line_elim: K[10] u[4] D2[4] g[4])
__global__ __launch_bounds__(128) void k_big_line(BatchPtrs p, BigPtrs bg, Policy pol) {
  const int ls = blockIdx.x * blockDim.x + threadIdx.x;
  if (ls >= p.nline) return;
  const int w = p.line_win[ls];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const bool fresh = st->fresh != 0;
  const int cur = st->cur;
  const int k = p.line_ptr[ls + 1] - p.line_ptr[ls];
  const bool line_active = !(p.line_flags[ls] & 1) && k > 0;
  double* la = bg.line_acc + (long long)ls * kBigLine;
  double H[10], g[4], D2[4], K[10], u[4] = { 0, 0, 0, 0 };
  for (int q = 0; q < 10; ++q) H[q] = la[q];
  for (int q = 0; q < 4; ++q) g[q] = la[10 + q];
  for (int q = 0; q < 4; ++q) la[14 + q] = 0.0;                       // w of the coming back-substitution
  double* sc = bg.scal + (long long)w * kBgScal;
  double* lsc = p.line_scale + (long long)ls * 4;
  if (fresh) {
    const double d[4] = { H[0], H[2], H[5], H[9] };
    const double* ul = p.line_x + line_rec(p, ls, cur);
    double sl[4], gm = 0.0, xn2 = 0.0;
    for (int a = 0; a < 4; ++a) {
      sl[a] = (pol.jacobi_scaling && line_active) ? 1.0 / (1.0 + sqrt(d[a])) : 1.0;
      lsc[a] = sl[a];
      if (line_active) { gm = fmax(gm, fabs(g[a])); xn2 += ul[a] * ul[a]; }
    }
    if (line_active) { atomic_max_nonneg(&sc[kBgGmaxLine], gm); atomicAdd(&sc[kBgXn2Line], xn2); }
    int q = 0;
    for (int a = 0; a < 4; ++a) {
      for (int b = 0; b <= a; ++b, ++q) H[q] *= sl[a] * sl[b];
      g[a] *= sl[a];
    }
  }
  lm_diag4(H, pol, 1.0 / st->radius, D2);
  bool ok = true;
  if (line_active) ok = chol4_inverse(H, D2, K);
  else { for (int q = 0; q < 10; ++q) K[q] = 0.0; }
  if (!ok) atomic_max_nonneg(&sc[kBgFail], 1.0);
  if (line_active) {
    u[0] = K[0] * g[0];
    u[1] = K[1] * g[0] + K[2] * g[1];
    u[2] = K[3] * g[0] + K[4] * g[1] + K[5] * g[2];
    u[3] = K[6] * g[0] + K[7] * g[1] + K[8] * g[2] + K[9] * g[3];
    if (st->need_grad_check) {
      double gm = 0.0;
      for (int a = 0; a < 4; ++a) gm = fmax(gm, fabs(g[a] / lsc[a]));
      atomic_max_nonneg(&sc[kBgGmaxLine], gm);
    }
  }
  double* le = p.line_elim + (long long)ls * p.line_elim_stride;
  for (int q = 0; q < 10; ++q) le[q] = K[q];
  for (int q = 0; q < 4; ++q) { le[kLeU + q] = u[q]; le[kLeD2 + q] = D2[q]; le[kLeG + q] = g[q]; }
}

// first sweep only: the line Jacobians were kept unscaled (the scale comes out of k_big_line)
__global__ __launch_bounds__(256) void k_big_rescale(BatchPtrs p, BigPtrs bg) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o];
  const LMState* st = p.state + p.line_win[ls];
  if (st->status != kRunning || !st->fresh) return;
  double* Jl = bg.J + o * kBigObs + 24;
  const double* lsc = p.line_scale + (long long)ls * 4;
  for (int q = 0; q < 4; ++q)
    for (int a = 0; a < 4; ++a) Jl[4 * q + a] *= lsc[a];
}

__device__ __forceinline__ void big_F(const double* J, const double K[10], double F[24]) {
  for (int a = 0; a < 6; ++a) {
    double h[4];
    for (int b = 0; b < 4; ++b) {
      double s = 0.0;
      for (int r = 0; r < 4; ++r) s += J[6 * r + a] * J[24 + 4 * r + b];
      h[b] = s;
    }
    F[4 * a + 0] = h[0] * K[0];
    F[4 * a + 1] = h[0] * K[1] + h[1] * K[2];
    F[4 * a + 2] = h[0] * K[3] + h[1] * K[4] + h[2] * K[5];
    F[4 * a + 3] = h[0] * K[6] + h[1] * K[7] + h[2] * K[8] + h[3] * K[9];
  }
}

// thread <-> pair of observations (i <= j) of a free line by free cameras: - F_j F_i^T into the reduced system, i == j also
// - F u into b
__global__ __launch_bounds__(128) void k_big_schur(BatchPtrs p, BigPtrs bg) {
  const long long q0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q0 >= bg.npairs) return;
  const int oi = bg.pair_i[q0], oj = bg.pair_j[q0];
  const int ls = bg.ob_line[oi], w = p.line_win[ls];
  const WinDesc wd = p.wins[w];
  if (p.state[w].status != kRunning) return;
  const double* le = p.line_elim + (long long)ls * p.line_elim_stride;
  double K[10], Fi[24], Fj[24];
  for (int q = 0; q < 10; ++q) K[q] = le[q];
  big_F(bg.J + (long long)oi * kBigObs, K, Fi);
  const int ci = p.cam_cf[wd.cam_off + p.ob_cam[oi]], cj = p.cam_cf[wd.cam_off + p.ob_cam[oj]];
  const int n = wd.n, ld = big_ld(n);
  double* S = bg.sys + bg.sys_off[w];
  if (oi == oj) {
    double* bvec = S + (long long)n * ld;
    for (int a = 0; a < 6; ++a) {
      atomicAdd(&bvec[6 * ci + a], -(Fi[4 * a] * le[kLeU] + Fi[4 * a + 1] * le[kLeU + 1] + Fi[4 * a + 2] * le[kLeU + 2] + Fi[4 * a + 3] * le[kLeU + 3]));
      for (int b = 0; b <= a; ++b) {
        double v = 0.0;
        for (int m = 0; m < 4; ++m) v += Fi[4 * a + m] * Fi[4 * b + m];
        atomicAdd(&S[(long long)(6 * ci + a) * ld + 6 * ci + b], -v);
      }
    }
    return;
  }
  big_F(bg.J + (long long)oj * kBigObs, K, Fj);
  if (ci == cj) {                                   // one camera observes the line twice: symmetric part
    for (int a = 0; a < 6; ++a)
      for (int b = 0; b <= a; ++b) {
        double v = 0.0;
        for (int m = 0; m < 4; ++m) v += Fj[4 * a + m] * Fi[4 * b + m] + Fi[4 * a + m] * Fj[4 * b + m];
        atomicAdd(&S[(long long)(6 * ci + a) * ld + 6 * ci + b], -v);
      }
    return;
  }
  const double* Fr = cj > ci ? Fj : Fi;            // rows: the camera with the larger free index (lower triangle)
  const double* Fc = cj > ci ? Fi : Fj;
  const int cr = cj > ci ? cj : ci, cc = cj > ci ? ci : cj;
  for (int a = 0; a < 6; ++a)
    for (int b = 0; b < 6; ++b) {
      double v = 0.0;
      for (int m = 0; m < 4; ++m) v += Fr[4 * a + m] * Fc[4 * b + m];
      atomicAdd(&S[(long long)(6 * cr + a) * ld + 6 * cc + b], -v);
    }
}

// one workgroup per window: Ceres' initial bookkeeping on the first sweep (cost, gradient norm, |x|, Jacobi scale of the
// camera columns, trace record 0, the tests that end a solve before its first step) and the congruence to scaled
// coordinates; gradient test after an accepted step; LM damping; right-hand side.  (k_reduced_solve steps 1b - 3.)
__global__ __launch_bounds__(256) void k_big_prepare(BatchPtrs p, BigPtrs bg, Policy pol) {
  __shared__ double red[8];
  const int w = blockIdx.x, tid = threadIdx.x;
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int n = wd.n, ld = big_ld(n);
  double* S = bg.sys + bg.sys_off[w];
  double* bvec = S + (long long)n * ld;
  double* gvec = bvec + n;
  double* hvec = gvec + n;
  double* yvec = hvec + n;
  const double* sc = bg.scal + (long long)w * kBgScal;
  const int cur = st->cur;
  const int fresh = st->fresh, need_grad = st->need_grad_check;
  const double radius = st->radius, abs_tol = st->abs_grad_tol;
  __syncthreads();
  if (fresh) {
    if (tid == 0) {
      double cost = sc[kBgCost], fixed = sc[kBgFixed], xn2 = sc[kBgXn2Line], gmax = sc[kBgGmaxLine];
      for (int c = 0; c < wd.C; ++c) {
        const int cf = p.cam_cf[wd.cam_off + c];
        for (int a = 0; a < 6; ++a) {
          double s = 1.0;
          if (cf >= 0) {
            const double x = p.cam_x[((long long)(wd.cam_off + c) * 2 + cur) * kCamRec + a];
            gmax = fmax(gmax, fabs(gvec[6 * cf + a]));
            xn2 += x * x;
            if (pol.jacobi_scaling) s = 1.0 / (1.0 + sqrt(hvec[6 * cf + a]));
          }
          p.cam_scale[(long long)(wd.cam_off + c) * 6 + a] = s;
        }
      }
      st->cost = cost; st->fixed_cost = fixed; st->initial_cost = cost + fixed; st->min_cost = cost + fixed;
      st->x_norm = sqrt(xn2);
      st->grad_max = gmax;
      st->abs_grad_tol = pol.gradient_tolerance * (gmax > 1e-12 ? gmax : 1e-12);
      st->need_grad_check = 0;
      st->fresh = 0;
      bg.scal[(long long)w * kBgScal + kBgWasFresh] = 1.0;      // the kept camera Jacobians are still unscaled: k_big_rescale_cameras
      int status = kRunning;
      if (wd.nfree_params == 0) status = 2;
      else if (!isfinite(cost)) status = 4;
      else if (gmax <= st->abs_grad_tol) status = 1;
      if (status == kRunning) {
        IterRec rec;
        rec.pad = 0;
        rec.iteration = 0; rec.step_is_valid = 0; rec.step_is_successful = 0;
        rec.cost = cost + fixed; rec.cost_change = 0; rec.gradient_max_norm = gmax; rec.step_norm = 0;
        rec.relative_decrease = 0; rec.trust_region_radius = st->radius; rec.model_cost_change = 0;
        push_trace(p, w, st, rec);
        if (pol.max_num_iterations <= 0) status = 0;
      }
      st->status = status;
      red[1] = (double)status;
    }
    __syncthreads();
    if (red[1] != (double)kRunning) return;
    // congruence with the Jacobi scale of the camera columns (unknown q of free camera cf: scale of that camera's entry q % 6)
    for (int c = tid; c < wd.C; c += 256) {
      const int cf = p.cam_cf[wd.cam_off + c];
      if (cf >= 0)
        for (int a = 0; a < 6; ++a) yvec[6 * cf + a] = p.cam_scale[(long long)(wd.cam_off + c) * 6 + a];
    }
    __syncthreads();
    for (long long q = tid; q < (long long)n * n; q += 256) {
      const int r = (int)(q / n), c = (int)(q - (long long)r * n);
      if (c <= r) S[(long long)r * ld + c] *= yvec[r] * yvec[c];
    }
    for (int q = tid; q < n; q += 256) { const double s = yvec[q]; bvec[q] *= s; gvec[q] *= s; hvec[q] *= s * s; }
    __syncthreads();
  } else if (need_grad) {
    double gm = 0.0;
    for (int c = tid; c < wd.C; c += 256) {
      const int cf = p.cam_cf[wd.cam_off + c];
      if (cf < 0) continue;
      for (int a = 0; a < 6; ++a) gm = fmax(gm, fabs(gvec[6 * cf + a] / p.cam_scale[(long long)(wd.cam_off + c) * 6 + a]));
    }
    gm = wave_max(gm);
    if ((tid & 63) == 0) red[2 + (tid >> 6)] = gm;
    __syncthreads();
    if (tid == 0) {
      gm = fmax(fmax(fmax(red[2], red[3]), fmax(red[4], red[5])), sc[kBgGmaxLine]);
      st->grad_max = gm;
      st->need_grad_check = 0;
      if (st->ntrace > 0 && st->ntrace <= kMaxTrace) p.trace[(long long)w * kMaxTrace + st->ntrace - 1].gradient_max_norm = gm;
      if (gm <= abs_tol) st->status = 1;
      red[0] = gm;
    }
    __syncthreads();
    if (red[0] <= abs_tol) return;
  }
  for (int q = tid; q < n; q += 256) {
    const double d2 = fmin(fmax(hvec[q], pol.min_lm_diagonal), pol.max_lm_diagonal) / radius;
    hvec[q] = d2;
    S[(long long)q * ld + q] += d2;
    yvec[q] = bvec[q];
  }
}

// first sweep only, after k_big_prepare derived the Jacobi scale of the camera columns: the camera Jacobians kept per
// observation go to scaled coordinates (the back-substitution multiplies them with the step in scaled coordinates)
__global__ __launch_bounds__(256) void k_big_rescale_cameras(BatchPtrs p, BigPtrs bg) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  if (p.state[w].status != kRunning || bg.scal[(long long)w * kBgScal + kBgWasFresh] == 0.0) return;
  const WinDesc wd = p.wins[w];
  const int cam = wd.cam_off + p.ob_cam[o];
  if (p.cam_cf[cam] < 0) return;
  double* Jc = bg.J + o * kBigObs;
  const double* cs = p.cam_scale + (long long)cam * 6;
  for (int q = 0; q < 4; ++q)
    for (int a = 0; a < 6; ++a) Jc[6 * q + a] *= cs[a];
}

// after the factorisation and the triangular solves: step statistics of the camera block, candidate camera poses
__global__ __launch_bounds__(64) void k_big_finish(BatchPtrs p, BigPtrs bg) {
  const int w = blockIdx.x, lane = threadIdx.x;
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int n = wd.n, ld = big_ld(n), cur = st->cur;
  const double* S = bg.sys + bg.sys_off[w];
  const double* gvec = S + (long long)n * ld + n;
  const double* hvec = gvec + n;
  const double* yvec = hvec + n;
  double model = 0.0, dn2 = 0.0, xn2 = 0.0;
  int bad = 0;
  for (int q = lane; q < n; q += 64) {
    const double y = yvec[q];
    if (!isfinite(y)) bad = 1;
    model += 0.5 * y * (gvec[q] + hvec[q] * y);
    p.ysys[wd.sys_off + q] = y;
  }
  for (int c = lane; c < wd.C; c += 64) {
    const int cf = p.cam_cf[wd.cam_off + c];
    const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
    double* xc = p.cam_x + ((long long)(wd.cam_off + c) * 2 + (1 - cur)) * kCamRec;
    for (int a = 0; a < 6; ++a) {
      double v = x[a];
      if (cf >= 0) {
        const double xn = v - yvec[6 * cf + a] * p.cam_scale[(long long)(wd.cam_off + c) * 6 + a];
        const double dd = v - xn;
        dn2 += dd * dd; xn2 += xn * xn;
        v = xn;
      }
      xc[a] = v;
    }
  }
  model = wave_sum(model); dn2 = wave_sum(dn2); xn2 = wave_sum(xn2);
  const int any_bad = __any(bad);
  if (lane == 0) {
    st->cam_model = model; st->cam_dn2 = dn2; st->cam_xn2 = xn2;
    st->solve_failed = (any_bad || bg.flags[2 * w] != 0 || bg.scal[(long long)w * kBgScal + kBgFail] != 0.0) ? 1 : 0;
  }
}

// thread <-> observation coupling a free camera to a free line: its term of  sum_i F_i^T y_c,i  (the line's w)
__global__ __launch_bounds__(128) void k_big_backsub_obs(BatchPtrs p, BigPtrs bg) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  const WinDesc wd = p.wins[w];
  if (p.state[w].status != kRunning) return;
  const int cf = p.cam_cf[wd.cam_off + p.ob_cam[o]];
  if (cf < 0 || (p.line_flags[ls] & 1)) return;
  const double* le = p.line_elim + (long long)ls * p.line_elim_stride;
  double K[10], F[24];
  for (int q = 0; q < 10; ++q) K[q] = le[q];
  big_F(bg.J + o * kBigObs, K, F);
  const double* y = p.ysys + wd.sys_off + 6 * cf;
  double* la = bg.line_acc + (long long)ls * kBigLine + 14;
  for (int m = 0; m < 4; ++m) {
    double v = 0.0;
    for (int a = 0; a < 6; ++a) v += F[4 * a + m] * y[a];
    atomicAdd(&la[m], v);
  }
}

// thread <-> line: y_l = K^T (u - w), candidate parameters and their sin/cos table, the line part of the step statistics
__global__ __launch_bounds__(128) void k_big_backsub_line(BatchPtrs p, BigPtrs bg) {
  const int ls = blockIdx.x * blockDim.x + threadIdx.x;
  if (ls >= p.nline) return;
  const int w = p.line_win[ls];
  const WinDesc wd = p.wins[w];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const int k = p.line_ptr[ls + 1] - p.line_ptr[ls];
  const bool line_active = !(p.line_flags[ls] & 1) && k > 0;
  const double* xl = p.line_x + line_rec(p, ls, cur);
  double* xc = p.line_x + line_rec(p, ls, (1 - cur));
  double xn[4] = { xl[0], xl[1], xl[2], xl[3] };
  if (line_active) {
    const double* le = p.line_elim + (long long)ls * p.line_elim_stride;
    const double* wv = bg.line_acc + (long long)ls * kBigLine + 14;
    const double* lsc = p.line_scale + (long long)ls * 4;
    const double z0 = le[kLeU] - wv[0], z1 = le[kLeU + 1] - wv[1], z2 = le[kLeU + 2] - wv[2], z3 = le[kLeU + 3] - wv[3];
    double y[4];
    y[0] = le[0] * z0 + le[1] * z1 + le[3] * z2 + le[6] * z3;
    y[1] = le[2] * z1 + le[4] * z2 + le[7] * z3;
    y[2] = le[5] * z2 + le[8] * z3;
    y[3] = le[9] * z3;
    double model = 0.0, dn2 = 0.0, xn2 = 0.0;
    for (int a = 0; a < 4; ++a) {
      model += 0.5 * y[a] * (le[kLeG + a] + le[kLeD2 + a] * y[a]);
      const double v = xn[a] - y[a] * lsc[a];
      const double dd = xn[a] - v;
      dn2 += dd * dd; xn2 += v * v;
      xn[a] = v;
    }
    double* bp = p.bs_part + (long long)wd.chunk_off * kBsStride;
    atomicAdd(&bp[kBsModel], model); atomicAdd(&bp[kBsDn2], dn2); atomicAdd(&bp[kBsXn2], xn2);
  }
  double trig[7];
  line_trig<double>(xn, trig);
  for (int a = 0; a < 4; ++a) xc[a] = xn[a];
  for (int a = 0; a < 7; ++a) xc[4 + a] = trig[a];
}

// thread <-> observation: cost at the candidate point
__global__ __launch_bounds__(128) void k_big_cost(BatchPtrs p, BigPtrs bg, Policy pol) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  const WinDesc wd = p.wins[w];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int cam = wd.cam_off + p.ob_cam[o], cf = p.cam_cf[cam];
  if (cf < 0 && (p.line_flags[ls] & 1)) return;                         // not in the reduced program
  const double* ct = bg.camtab + ((long long)cam * 2 + 1) * kBigCam;
  const double* lrec = p.line_x + line_rec(p, ls, (1 - st->cur));
  double R[9], t[3] = { ct[18], ct[19], ct[20] }, trig[7], ob[8], cp[3], dv[3], r[4], c;
  for (int q = 0; q < 9; ++q) R[q] = ct[q];
  for (int q = 0; q < 7; ++q) trig[q] = lrec[4 + q];
  for (int q = 0; q < 4; ++q) {
    const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
    ob[2 * q] = e.x; ob[2 * q + 1] = e.y;
  }
  line_points<double>(trig, cp, dv);
  obs_residual<double>(R, t, cp, dv, ob, pol.baseline, r);
  huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &c);
  atomicAdd(&p.cost_part[wd.chunk_off], c);
}

}  // namespace slslam
#endif  // SLSLAM_LBA_BIG_H_
