// slslam_amd/csrc/lba_big.h — the line bundle adjustment for windows beyond what the tiled sweeps hold on chip: more than
// 20 free or 64 cameras, or a line observed by more than 64 keyframes.  The reference makes the window size a flag
// (src/main.cpp:22 ba_window_size) and its study runs W = 40 (80 keyframes, 40 free: BASELINE.md section 1), where the
// reduced camera system (240^2) no longer fits one wave's LDS partial or one workgroup's Cholesky.
//
// Same algorithm, same LM bookkeeping (lm_step, k_lm_update), same per-observation arithmetic (lba_math.h); what changes is
// where the sums live: Jacobians and F blocks are kept per observation in HBM / L2, and every sum - a line's 4x4 block, a
// camera's record, a camera pair's block, the costs - is formed by one thread group walking a host-built list in a fixed
// order (gather, not scatter: no atomics, results bitwise reproducible; round 2 scattered with fp64 global atomics and
// spent most of its time on their contention).  The reduced system is factored by the pose-graph path's blocked MFMA
// Cholesky (po_kernels.h: k_po_potrf_diag / k_po_panel_update / k_po_trisolve on v_mfma_f64_16x16x4_f64).
//
// One LM iteration:  k_big_cameras(0) -> k_big_linearise -> k_big_line -> k_big_rescale -> k_big_F -> k_big_cam -> k_big_pairs ->
// k_big_prepare -> [potrf / panel updates / trisolve per window] -> k_big_finish -> k_big_cameras(1) -> k_big_backsub_line ->
// k_big_cost -> k_big_reduce (with the trust-region bookkeeping).
#ifndef SLSLAM_LBA_BIG_H_
#define SLSLAM_LBA_BIG_H_

#include "lba_kernels.h"

namespace slslam {

enum { kBigObs = 46 };         // doubles kept per observation: Jc[24] | Jl[16] | rs[4] | cost | cost at the candidate point
enum { kBigF = 24 };           // F = (Jc^T Jl) K^T per observation coupling a free camera to a free line
enum { kBigLine = 8 };         // per-line outputs: |g|_inf, |x|^2, failure flag | model, |dx|^2, |x+|^2 of the step | pad
enum { kBlGmax = 0, kBlXn2 = 1, kBlFail = 2, kBlModel = 3, kBlDn2 = 4, kBlXn2New = 5 };
enum { kBigCam = 21 };         // R[9] JL[9] t[3] per camera and buffer (accepted, candidate)
enum { kBgFail = 4, kBgWasFresh = 5, kBgScal = 8 };

struct BigPtrs {
  const int* ob_line;          // [nobs] sorted line of every sorted observation
  const int* cam_win;          // [ncam]
  double* J;                   // [nobs][kBigObs]
  double* F;                   // [nobs][kBigF]
  double* cost;                // [2][nobs] block cost of every observation at the accepted / at the candidate point (dense: the
                               // per-window sums stream them)
  double* camtab;              // [ncam][2][kBigCam]
  double* line_acc;            // [nline][kBigLine]
  double* sys;                 // per window: S [n x ld] | b [n] | g [n] | h [n] | y [n] | Jacobi scale of the unknowns [n]
  const long long* sys_off;    // [nwin]
  // gather lists, built on the host: every sum of the path is formed by ONE thread group walking a list in a fixed order - no
  // atomics, results bitwise reproducible
  const int* cam_ptr;          // [ncam + 1] observations (sorted index, ascending) of every camera
  const int* cam_obs;
  const int* pair_ptr;         // [npairs + 1] per (window, camera pair r >= c): the line's two observations (row camera r, column camera c)
  const int* pair_row;
  const int* pair_col;
  const int* pair_desc;        // [npairs] window | r << 16 | c << 24
  double* scal;                // [nwin][kBgScal]
  int* flags;                  // [nwin][2] factorisation failure (PoPtrs.flags)
  long long npairs, nobs;
};
__host__ __device__ inline int big_ld(int n) { return ((n + 7) / 8) * 8 + 8; }
__host__ __device__ inline long long big_sys_doubles(int n) { return (long long)(n > 0 ? n : 1) * big_ld(n) + 5LL * (n > 0 ? n : 1); }

// Sum of one value per thread over a 256-thread workgroup in a fixed order (the same tree every run); red: 4 doubles of LDS.
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ double block_max_256(double v, double* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

// thread <-> camera: rotation, SO(3) left Jacobian and translation of the accepted (which = 0) or candidate (1) pose
__device__ __forceinline__ void big_camera_entry(const BatchPtrs& p, const BigPtrs& bg, const int i, const int buf, const int which) {
  const double* x = p.cam_x + ((long long)i * 2 + buf) * kCamRec;
  double w[3] = { x[0], x[1], x[2] }, R[9], JL[9];
  cam_prepare<double>(w, R, JL);
  double* ct = bg.camtab + ((long long)i * 2 + which) * kBigCam;
  for (int q = 0; q < 9; ++q) { ct[q] = R[q]; ct[9 + q] = JL[q]; }
  ct[18] = x[3]; ct[19] = x[4]; ct[20] = x[5];
}
__global__ __launch_bounds__(256) void k_big_cameras(BatchPtrs p, BigPtrs bg, int which) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.ncam) return;
  const LMState* st = p.state + bg.cam_win[i];
  if (st->status != kRunning) return;
  big_camera_entry(p, bg, i, which ? 1 - st->cur : st->cur, which);
}

// thread <-> observation: residual, Jacobians, Huber, scaling, kept per observation.  At the first sweep of a solve (fresh)
// all scales are 1 (see k_big_line / k_big_prepare).
__device__ __forceinline__ void big_linearise_obs(const BatchPtrs& p, const BigPtrs& bg, const Policy& pol, const long long o, const int ls,
                                                  const WinDesc& wd, const int cur, const bool fresh) {
  const int cam = wd.cam_off + p.ob_cam[o], cf = p.cam_cf[cam];
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
  bg.cost[o] = cost;
}
__global__ __launch_bounds__(128) void k_big_linearise(BatchPtrs p, BigPtrs bg, Policy pol) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  big_linearise_obs(p, bg, pol, o, ls, p.wins[w], st->cur, st->fresh != 0);
}

// the line Jacobian of one observation to scaled coordinates (first sweep: the scale comes out of the line's block)
__device__ __forceinline__ void big_rescale_obs(const BigPtrs& bg, const long long o, const double* lsc) {
  double* Jl = bg.J + o * kBigObs + 24;
  for (int q = 0; q < 4; ++q)
    for (int a = 0; a < 4; ++a) Jl[4 * q + a] *= lsc[a];
}
// F = (Jc^T Jl) K^T of one observation coupling a free camera to a free line (zero otherwise); K: the line's chol^-1 (10)
__device__ __forceinline__ void big_F_obs(const BatchPtrs& p, const BigPtrs& bg, const long long o, const int ls, const WinDesc& wd,
                                          const double* K) {
  double* F = bg.F + o * kBigF;
  if (p.cam_cf[wd.cam_off + p.ob_cam[o]] < 0 || (p.line_flags[ls] & 1)) { for (int q = 0; q < kBigF; ++q) F[q] = 0.0; return; }
  const double* J = bg.J + o * kBigObs;
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

// wave <-> line: the line's normal-equation block summed over its observations (lane <-> observation, 64 at a time in order,
// fixed butterfly); Jacobi scale (first sweep), LM damping, 4x4 Cholesky, K = chol^-1, u = K g; kept for the Schur products and
// the back-substitution (line_elim)
// ALL: the same wave also linearises the line's observations first (k_big_linearise) and forms their F blocks afterwards
// (k_big_rescale, k_big_F): one launch instead of four on a path whose kernels each run for a few microseconds.  A lane reads
// back only what it wrote itself (lane <-> observation, the same chunks of 64 in every pass).
template <bool ALL>
__global__ __launch_bounds__(256) void k_big_line(BatchPtrs p, BigPtrs bg, Policy pol) {
  const int ls = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ls >= p.nline) return;
  const int w = p.line_win[ls];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const bool fresh = st->fresh != 0;
  const int cur = st->cur;
  const int o0 = p.line_ptr[ls], k = p.line_ptr[ls + 1] - o0;
  const bool line_active = !(p.line_flags[ls] & 1) && k > 0;
  if (ALL) {
    const WinDesc wd = p.wins[w];
    for (int base = 0; base < k; base += 64)
      if (base + lane < k) big_linearise_obs(p, bg, pol, o0 + base + lane, ls, wd, cur, fresh);
  }
  double* la = bg.line_acc + (long long)ls * kBigLine;
  double H[10], g[4], D2[4], K[10], u[4] = { 0, 0, 0, 0 };
#pragma unroll
  for (int q = 0; q < 10; ++q) H[q] = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) g[q] = 0.0;
  if (line_active) {
    for (int base = 0; base < k; base += 64) {
      const bool has = base + lane < k;
      const double* J = bg.J + (long long)(o0 + (has ? base + lane : 0)) * kBigObs;
      double Jl[16], r[4];
#pragma unroll
      for (int q = 0; q < 16; ++q) Jl[q] = has ? J[24 + q] : 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = has ? J[40 + q] : 0.0;
      int q = 0;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b, ++q) {
          double h = 0.0;
#pragma unroll
          for (int m = 0; m < 4; ++m) h += Jl[4 * m + a] * Jl[4 * m + b];
          H[q] += wave_sum(h);
        }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        double ga = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) ga += Jl[4 * m + a] * r[m];
        g[a] += wave_sum(ga);
      }
    }
  }
  double* lsc = p.line_scale + (long long)ls * 4;
  double gm = 0.0, xn2 = 0.0, sl[4] = { 1.0, 1.0, 1.0, 1.0 };
  if (fresh) {
    const double d[4] = { H[0], H[2], H[5], H[9] };
    const double* ul = p.line_x + line_rec(p, ls, cur);
    for (int a = 0; a < 4; ++a) {
      sl[a] = (pol.jacobi_scaling && line_active) ? 1.0 / (1.0 + sqrt(d[a])) : 1.0;
      if (line_active) { gm = fmax(gm, fabs(g[a])); xn2 += ul[a] * ul[a]; }
    }
    int q = 0;
    for (int a = 0; a < 4; ++a) {
      for (int b = 0; b <= a; ++b, ++q) H[q] *= sl[a] * sl[b];
      g[a] *= sl[a];
    }
  } else {
    for (int a = 0; a < 4; ++a) sl[a] = lsc[a];
  }
  lm_diag4(H, pol, 1.0 / st->radius, D2);
  bool ok = true;
  if (line_active) ok = chol4_inverse(H, D2, K);
  else { for (int q = 0; q < 10; ++q) K[q] = 0.0; }
  if (line_active) {
    u[0] = K[0] * g[0];
    u[1] = K[1] * g[0] + K[2] * g[1];
    u[2] = K[3] * g[0] + K[4] * g[1] + K[5] * g[2];
    u[3] = K[6] * g[0] + K[7] * g[1] + K[8] * g[2] + K[9] * g[3];
    if (st->need_grad_check)
      for (int a = 0; a < 4; ++a) gm = fmax(gm, fabs(g[a] / sl[a]));
  }
  if (ALL) {
    const WinDesc wd = p.wins[w];
    for (int base = 0; base < k; base += 64) {
      if (base + lane >= k) continue;
      const long long o = o0 + base + lane;
      if (fresh) big_rescale_obs(bg, o, sl);
      big_F_obs(p, bg, o, ls, wd, K);
    }
  }
  if (lane != 0) return;                                  // every lane holds the same values; one writes
  if (fresh) for (int a = 0; a < 4; ++a) lsc[a] = sl[a];
  la[kBlGmax] = gm; la[kBlXn2] = xn2; la[kBlFail] = ok ? 0.0 : 1.0;
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
  big_rescale_obs(bg, o, p.line_scale + (long long)ls * 4);
}

// thread <-> observation coupling a free camera to a free line: F = (Jc^T Jl) K^T, kept for the camera blocks, the pair
// blocks and the back-substitution
__global__ __launch_bounds__(128) void k_big_F(BatchPtrs p, BigPtrs bg) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  if (p.state[w].status != kRunning) return;
  big_F_obs(p, bg, o, ls, p.wins[w], p.line_elim + (long long)ls * p.line_elim_stride);
}

// One workgroup (4 waves) per free camera: its record of the reduced system - diagonal block Jc^T Jc - F F^T (21), b = Jc^T r - F u
// (6), gradient (6), diag(Jc^T Jc) (6) - summed over the camera's observations.  Wave s walks the s-th quarter of the camera's
// list, lane <-> observation, 64 at a time in list order; every entry is reduced over the wave with a fixed butterfly and
// added to the wave's running sum, the four sums are added in wave order: a fixed order, no atomics.
__global__ __launch_bounds__(256) void k_big_cam(BatchPtrs p, BigPtrs bg) {
  __shared__ double part[4][40];
  const int cam = blockIdx.x, lane = threadIdx.x & 63, sgm = threadIdx.x >> 6;
  const int w = bg.cam_win[cam];
  if (p.state[w].status != kRunning) return;
  const int cf = p.cam_cf[cam];
  if (cf < 0) return;
  const WinDesc wd = p.wins[w];
  const int c0 = bg.cam_ptr[cam], call = bg.cam_ptr[cam + 1] - c0;
  const int b0 = c0 + (int)(((long long)call * sgm) / 4), cnt = c0 + (int)(((long long)call * (sgm + 1)) / 4) - b0;
  double acc[39];
#pragma unroll
  for (int q = 0; q < 39; ++q) acc[q] = 0.0;
  for (int base = 0; base < cnt; base += 64) {
    const bool has = base + lane < cnt;
    const long long o = bg.cam_obs[has ? b0 + base + lane : c0];
    const double* J = bg.J + o * kBigObs;
    const double* Fp = bg.F + o * kBigF;
    const double* up = p.line_elim + (long long)bg.ob_line[o] * p.line_elim_stride + kLeU;
    double Jc[24], r[4], F[24], u[4];
#pragma unroll
    for (int q = 0; q < 24; ++q) { Jc[q] = has ? J[q] : 0.0; F[q] = has ? Fp[q] : 0.0; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { r[q] = has ? J[40 + q] : 0.0; u[q] = has ? up[q] : 0.0; }
    int e = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b, ++e) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v += Jc[6 * k + a] * Jc[6 * k + b];
#pragma unroll
        for (int m = 0; m < 4; ++m) v -= F[4 * a + m] * F[4 * b + m];
        acc[e] += wave_sum(v);
      }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double ga = 0.0, ha = 0.0, fu = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) { ga += Jc[6 * k + a] * r[k]; ha += Jc[6 * k + a] * Jc[6 * k + a]; }
#pragma unroll
      for (int m = 0; m < 4; ++m) fu += F[4 * a + m] * u[m];            // F = 0 where the line is constant
      acc[21 + a] += wave_sum(ga - fu);
      acc[27 + a] += wave_sum(ga);
      acc[33 + a] += wave_sum(ha);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 39; ++q) part[sgm][q] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x < 39) {
    const int e = threadIdx.x;
    const double tot = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
    const int n = wd.n, ld = big_ld(n);
    double* S = bg.sys + bg.sys_off[w];
    double* bvec = S + (long long)n * ld;
    if (e < 21) {
      int a = 0;
      while (((a + 1) * (a + 2)) / 2 <= e) ++a;
      S[(long long)(6 * cf + a) * ld + 6 * cf + (e - (a * (a + 1)) / 2)] = tot;
    } else {
      bvec[((e - 21) / 6) * n + 6 * cf + (e - 21) % 6] = tot;           // b | g | h are consecutive
    }
  }
}

// One workgroup (4 waves) per (window, camera pair r >= c): the pair's block - sum_lines F_r F_c^T of the reduced system.  Wave s
// walks the s-th quarter of the pair's list (the two observations of a line both cameras see), lane <-> item, 64 at a time in
// list order, every entry reduced over the wave with a fixed butterfly; the four sums are added in wave order.  r == c: a
// camera that observes a line twice (never in the reference's maps) - the symmetric part goes on top of the diagonal block
// k_big_cam stored.
__global__ __launch_bounds__(256) void k_big_pairs(BatchPtrs p, BigPtrs bg) {
  __shared__ double part[4][36];
  const int pr = blockIdx.x, lane = threadIdx.x & 63, sgm = threadIdx.x >> 6;
  const int desc = bg.pair_desc[pr], w = desc & 0xffff, r = (desc >> 16) & 0xff, c = (desc >> 24) & 0xff;
  if (p.state[w].status != kRunning) return;
  const int c0 = bg.pair_ptr[pr], call = bg.pair_ptr[pr + 1] - c0;
  if (r == c && call == 0) return;
  const int b0 = c0 + (int)(((long long)call * sgm) / 4), cnt = c0 + (int)(((long long)call * (sgm + 1)) / 4) - b0;
  double acc[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) acc[q] = 0.0;
  for (int base = 0; base < cnt; base += 64) {
    const bool has = base + lane < cnt;
    const int i = has ? b0 + base + lane : 0;
    const double* Frp = bg.F + (long long)bg.pair_row[i] * kBigF;
    const double* Fcp = bg.F + (long long)bg.pair_col[i] * kBigF;
    double Fr[24], Fc[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) { Fr[q] = has ? Frp[q] : 0.0; Fc[q] = has ? Fcp[q] : 0.0; }
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < 4; ++m) v += Fr[4 * a + m] * Fc[4 * b + m];
        acc[6 * a + b] -= wave_sum(v);
      }
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 36; ++q) part[sgm][q] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x < 36) part[0][threadIdx.x] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
  __syncthreads();
  if (threadIdx.x < 36) {
    const WinDesc wd = p.wins[w];
    const int ld = big_ld(wd.n), a = threadIdx.x / 6, b = threadIdx.x - 6 * a;
    double* S = bg.sys + bg.sys_off[w];
    if (r != c) S[(long long)(6 * r + a) * ld + 6 * c + b] = part[0][6 * a + b];
    else if (b <= a) S[(long long)(6 * r + a) * ld + 6 * c + b] += part[0][6 * a + b] + part[0][6 * b + a];
  }
}

// one workgroup per window: Ceres' initial bookkeeping on the first sweep (cost, gradient norm, |x|, Jacobi scale of the
// camera columns, trace record 0, the tests that end a solve before its first step) and the congruence to scaled
// coordinates; gradient test after an accepted step; LM damping; right-hand side.  (k_reduced_solve steps 1b - 3.)
__global__ __launch_bounds__(256) void k_big_prepare(BatchPtrs p, BigPtrs bg, Policy pol, int solver_scales) {
  __shared__ double red[8], red4[4];
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
  const int cur = st->cur;
  const int fresh = st->fresh, need_grad = st->need_grad_check;
  const double radius = st->radius, abs_tol = st->abs_grad_tol;
  // what the lines of the window reported (k_big_line), reduced in a fixed order
  double gmax_line, xn2_line, cost_sum = 0.0, fixed_sum = 0.0;
  {
    double gm = 0.0, xs = 0.0, fl = 0.0;
    for (int l = tid; l < wd.L; l += 256) {
      const double* la = bg.line_acc + (long long)(wd.line_off + l) * kBigLine;
      gm = fmax(gm, la[kBlGmax]); xs += la[kBlXn2]; fl = fmax(fl, la[kBlFail]);
    }
    gmax_line = block_max_256(gm, red4);
    xn2_line = block_sum_256(xs, red4);
    const double fail = block_max_256(fl, red4);
    if (tid == 0) { bg.scal[(long long)w * kBgScal + kBgFail] = fail; if (!fresh) bg.scal[(long long)w * kBgScal + kBgWasFresh] = 0.0; }
  }
  if (fresh) {
    double cs = 0.0, fs = 0.0;
    for (int o = tid; o < wd.M; o += 256) {
      const long long og = (long long)wd.obs_off + o;
      const bool kept = !(p.cam_cf[wd.cam_off + p.ob_cam[og]] < 0 && (p.line_flags[bg.ob_line[og]] & 1));
      const double c = bg.cost[og];
      if (kept) cs += c; else fs += c;
    }
    cost_sum = block_sum_256(cs, red4);
    fixed_sum = block_sum_256(fs, red4);
  }
  __syncthreads();
  if (fresh) {
    // camera entries: thread <-> (camera, component), sums in the fixed order of the workgroup's tree
    double gm_c = 0.0, xs_c = 0.0;
    for (int q = tid; q < 6 * wd.C; q += 256) {
      const int c = q / 6, a = q - 6 * c, cf = p.cam_cf[wd.cam_off + c];
      double s = 1.0;
      if (cf >= 0) {
        const double x = p.cam_x[((long long)(wd.cam_off + c) * 2 + cur) * kCamRec + a];
        gm_c = fmax(gm_c, fabs(gvec[6 * cf + a]));
        xs_c += x * x;
        if (pol.jacobi_scaling) s = 1.0 / (1.0 + sqrt(hvec[6 * cf + a]));
      }
      p.cam_scale[(long long)(wd.cam_off + c) * 6 + a] = s;
    }
    gm_c = block_max_256(gm_c, red4);
    xs_c = block_sum_256(xs_c, red4);
    if (tid == 0) {
      const double cost = cost_sum, fixed = fixed_sum, xn2 = xn2_line + xs_c, gmax = fmax(gmax_line, gm_c);
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
    // (solver_scales: k_big_solve applies the congruence and the damping to the copy of S it factorises - the matrix in
    // memory is only ever written by the sweeps that build it)
    if (!solver_scales)
      for (long long q = tid; q < (long long)n * n; q += 256) {
        const int r = (int)(q / n), c = (int)(q - (long long)r * n);
        if (c <= r) S[(long long)r * ld + c] *= yvec[r] * yvec[c];
      }
    for (int q = tid; q < n; q += 256) { const double s = yvec[q]; bvec[q] *= s; gvec[q] *= s; hvec[q] *= s * s; yvec[n + q] = s; }
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
      gm = fmax(fmax(fmax(red[2], red[3]), fmax(red[4], red[5])), gmax_line);
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
    if (!solver_scales) S[(long long)q * ld + q] += d2;
    yvec[q] = bvec[q];
  }
}

// first sweep only, after k_big_prepare derived the Jacobi scale of the camera columns: the camera Jacobians and F blocks kept
// per observation go to scaled coordinates (the back-substitution multiplies them with the step in scaled coordinates)
__global__ __launch_bounds__(256) void k_big_rescale_cameras(BatchPtrs p, BigPtrs bg) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  if (p.state[w].status != kRunning || bg.scal[(long long)w * kBgScal + kBgWasFresh] == 0.0) return;
  const WinDesc wd = p.wins[w];
  const int cam = wd.cam_off + p.ob_cam[o];
  if (p.cam_cf[cam] < 0) return;
  double* Jc = bg.J + o * kBigObs;
  double* F = bg.F + o * kBigF;
  const double* cs = p.cam_scale + (long long)cam * 6;
  for (int q = 0; q < 4; ++q)
    for (int a = 0; a < 6; ++a) Jc[6 * q + a] *= cs[a];
  for (int a = 0; a < 6; ++a)                                  // F = (Jc^T Jl) K^T follows its camera rows
    for (int m = 0; m < 4; ++m) F[4 * a + m] *= cs[a];
}

// after the factorisation and the triangular solves: step statistics of the camera block, candidate camera poses.  One wave;
// factor_failed: the factorisation met a non-positive pivot.
__device__ __forceinline__ void big_finish(const BatchPtrs& p, const BigPtrs& bg, const int w, const int lane, const int factor_failed) {
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
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
    st->solve_failed = (any_bad || factor_failed || bg.scal[(long long)w * kBgScal + kBgFail] != 0.0) ? 1 : 0;
  }
}
__global__ __launch_bounds__(64) void k_big_finish(BatchPtrs p, BigPtrs bg) {
  const int w = blockIdx.x;
  if (p.state[w].status != kRunning) return;
  big_finish(p, bg, w, threadIdx.x, bg.flags[2 * w] != 0);
}

// cost of one observation at the candidate point (candidate camera table; trig: sin / cos table of the candidate line)
__device__ __forceinline__ void big_cost_obs(const BatchPtrs& p, const BigPtrs& bg, const Policy& pol, const long long o, const int ls,
                                             const WinDesc& wd, const double* trig) {
  const int cam = wd.cam_off + p.ob_cam[o], cf = p.cam_cf[cam];
  double c = 0.0;
  if (!(cf < 0 && (p.line_flags[ls] & 1))) {                            // in the reduced program
    const double* ct = bg.camtab + ((long long)cam * 2 + 1) * kBigCam;
    double R[9], t[3] = { ct[18], ct[19], ct[20] }, ob[8], cp[3], dv[3], r[4];
    for (int q = 0; q < 9; ++q) R[q] = ct[q];
    for (int q = 0; q < 4; ++q) {
      const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
      ob[2 * q] = e.x; ob[2 * q + 1] = e.y;
    }
    line_points<double>(trig, cp, dv);
    obs_residual<double>(R, t, cp, dv, ob, pol.baseline, r);
    huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &c);
  }
  bg.cost[bg.nobs + o] = c;
}
// thread <-> observation: cost at the candidate point, kept per observation
__global__ __launch_bounds__(128) void k_big_cost(BatchPtrs p, BigPtrs bg, Policy pol) {
  const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= bg.nobs) return;
  const int ls = bg.ob_line[o], w = p.line_win[ls];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const double* lrec = p.line_x + line_rec(p, ls, (1 - st->cur));
  double trig[7];
  for (int q = 0; q < 7; ++q) trig[q] = lrec[4 + q];
  big_cost_obs(p, bg, pol, o, ls, p.wins[w], trig);
}

// wave <-> line: w = sum_i F_i^T y_c,i over the line's observations (lane <-> observation, fixed butterfly), y_l = K^T (u - w),
// candidate parameters and their sin/cos table, the line's part of the step statistics (summed per window by k_big_reduce)
// COST: the same wave then evaluates the cost of the line's observations at the candidate point (k_big_cost; the candidate
// camera table is there since the reduced solve) - one launch less.
template <bool COST>
__global__ __launch_bounds__(256) void k_big_backsub_line(BatchPtrs p, BigPtrs bg, Policy pol) {
  const int ls = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ls >= p.nline) return;
  const int w = p.line_win[ls];
  const WinDesc wd = p.wins[w];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const int o0 = p.line_ptr[ls], k = p.line_ptr[ls + 1] - o0;
  const bool line_active = !(p.line_flags[ls] & 1) && k > 0;
  const double* xl = p.line_x + line_rec(p, ls, cur);
  double* xc = p.line_x + line_rec(p, ls, (1 - cur));
  double xn[4] = { xl[0], xl[1], xl[2], xl[3] };
  double model = 0.0, dn2 = 0.0, xn2 = 0.0;
  if (line_active) {
    double wv[4] = { 0, 0, 0, 0 };
    for (int base = 0; base < k; base += 64) {
      const int j = base + lane;
      const int cf = j < k ? p.cam_cf[wd.cam_off + p.ob_cam[o0 + j]] : -1;
      const double* F = bg.F + (long long)(o0 + (j < k ? j : 0)) * kBigF;
      const double* y = p.ysys + wd.sys_off + 6 * (cf >= 0 ? cf : 0);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) v += F[4 * a + m] * y[a];
        wv[m] += wave_sum(cf >= 0 ? v : 0.0);
      }
    }
    const double* le = p.line_elim + (long long)ls * p.line_elim_stride;
    const double* lsc = p.line_scale + (long long)ls * 4;
    const double z0 = le[kLeU] - wv[0], z1 = le[kLeU + 1] - wv[1], z2 = le[kLeU + 2] - wv[2], z3 = le[kLeU + 3] - wv[3];
    double y[4];
    y[0] = le[0] * z0 + le[1] * z1 + le[3] * z2 + le[6] * z3;
    y[1] = le[2] * z1 + le[4] * z2 + le[7] * z3;
    y[2] = le[5] * z2 + le[8] * z3;
    y[3] = le[9] * z3;
    for (int a = 0; a < 4; ++a) {
      model += 0.5 * y[a] * (le[kLeG + a] + le[kLeD2 + a] * y[a]);
      const double v = xn[a] - y[a] * lsc[a];
      const double dd = xn[a] - v;
      dn2 += dd * dd; xn2 += v * v;
      xn[a] = v;
    }
  }
  double trig[7];
  line_trig<double>(xn, trig);                            // (every lane holds the same candidate)
  if (COST)
    for (int base = 0; base < k; base += 64)
      if (base + lane < k) big_cost_obs(p, bg, pol, o0 + base + lane, ls, wd, trig);
  if (lane != 0) return;
  double* la = bg.line_acc + (long long)ls * kBigLine;
  la[kBlModel] = model; la[kBlDn2] = dn2; la[kBlXn2New] = xn2;
  for (int a = 0; a < 4; ++a) xc[a] = xn[a];
  for (int a = 0; a < 7; ++a) xc[4 + a] = trig[a];
}

// one workgroup per window: the candidate cost and the lines' step statistics summed in a fixed order, for k_lm_update
__global__ __launch_bounds__(256) void k_big_reduce(BatchPtrs p, BigPtrs bg, Policy pol, int refresh_cameras) {
  __shared__ double red4[4];
  __shared__ int cur_now;
  const int w = blockIdx.x, tid = threadIdx.x;
  const WinDesc wd = p.wins[w];
  if (p.state[w].status != kRunning) return;
  double cs = 0.0, m = 0.0, d = 0.0, x = 0.0;
  for (int o = tid; o < wd.M; o += 256) cs += bg.cost[bg.nobs + wd.obs_off + o];
  for (int l = tid; l < wd.L; l += 256) {
    const double* la = bg.line_acc + (long long)(wd.line_off + l) * kBigLine;
    m += la[kBlModel]; d += la[kBlDn2]; x += la[kBlXn2New];
  }
  cs = block_sum_256(cs, red4); m = block_sum_256(m, red4); d = block_sum_256(d, red4); x = block_sum_256(x, red4);
  // ... and the trust-region bookkeeping of the step (what k_lm_update does for the tiled path)
  if (tid == 0) {
    LMState* st = p.state + w;
    lm_step(p, pol, w, st, cs, st->cam_model + m, st->cam_dn2 + d, st->cam_xn2 + x);
    cur_now = st->cur;                                     // (through LDS: the other threads must not read a stale copy)
  }
  // rotation / Jacobian table of the cameras at the point the next sweep linearises at (what k_big_cameras(0) does at the
  // start of an iteration)
  if (refresh_cameras) {
    __syncthreads();
    const int cur = cur_now;
    for (int c = tid; c < wd.C; c += 256) big_camera_entry(p, bg, wd.cam_off + c, cur, 0);
  }
}

}  // namespace slslam
#endif  // SLSLAM_LBA_BIG_H_
