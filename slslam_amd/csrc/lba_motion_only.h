// slslam_amd/csrc/lba_motion_only.h — the whole LM solve of a motion-only window in ONE launch, one wave per window.
//
// SLAM::motion_only_ba (reference src/slam.cpp:578-675) is the degenerate shape of the LBA problem: one free camera, every
// line constant, so the reduced system is the 6 x 6 block of that camera and nothing has to be eliminated or
// back-substituted.  The general path still costs 4 dependent launches per iteration; here a wave keeps the system in
// registers and loops: sweep the observations (same lane <-> observation tiles, same linearisation routines as the
// elimination kernel), wave-reduce J^T J / J^T r, damp, 6 x 6 Cholesky in registers (every lane, redundantly), candidate
// pose, sweep for the candidate cost, lm_step().  Same trust-region policy, same trace records as the general path.
#ifndef SLSLAM_LBA_MOTION_ONLY_H_
#define SLSLAM_LBA_MOTION_ONLY_H_

#include "lba_kernels.h"

namespace slslam {

__host__ __device__ inline int lds_doubles_motion_only(int C) {
  return C * kCamTab + 6 + (int)((sizeof(LMState) + 7) / 8) + (C + 7) / 8;
}

// M = L L^T (lower triangle packed by tri_index), then M y = b.  False if M is not positive definite / not finite.
__device__ __forceinline__ bool chol6_solve(const double (&M)[21], const double (&b)[6], double (&y)[6]) {
  double Lc[21];
  bool ok = true;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double v = M[tri_index(r, c)];
#pragma unroll
      for (int k = 0; k < c; ++k) v -= Lc[tri_index(r, k)] * Lc[tri_index(c, k)];
      if (c == r) { ok = ok && v > 0.0; Lc[tri_index(r, r)] = sqrt(v); }
      else Lc[tri_index(r, c)] = v / Lc[tri_index(c, c)];
    }
  }
  double z[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double v = b[r];
#pragma unroll
    for (int k = 0; k < r; ++k) v -= Lc[tri_index(r, k)] * z[k];
    z[r] = v / Lc[tri_index(r, r)];
  }
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    double v = z[r];
#pragma unroll
    for (int k = r + 1; k < 6; ++k) v -= Lc[tri_index(k, r)] * y[k];
    y[r] = v / Lc[tri_index(r, r)];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) ok = ok && isfinite(y[r]);
  return ok;
}

__global__ __launch_bounds__(64) void k_motion_only(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x, w = blockIdx.x;
  const WinDesc wd = p.wins[w];
  LMState* gst = p.state + w;
  if (gst->status != kRunning) return;
  double* camtab = smem;
  double* camscale = camtab + wd.C * kCamTab;                 // the one free camera's Jacobi scale
  LMState* st = (LMState*)(camscale + 6);
  signed char* camcf = (signed char*)((double*)st + (sizeof(LMState) + 7) / 8);
  const int cur = gst->cur;
  if (lane == 0) *st = *gst;
  load_cam_table<true>(p, wd, cur, lane, camtab, camscale, camcf, true);
  __syncthreads();
  int fc = 0;
  for (int c = 0; c < wd.C; ++c) if (camcf[c] == 0) fc = c;
  double x[6], sc[6];
  {
    const double* xg = p.cam_x + ((long long)(wd.cam_off + fc) * 2 + cur) * kCamRec;
#pragma unroll
    for (int a = 0; a < 6; ++a) { x[a] = xg[a]; sc[a] = 1.0; }
  }
  const int t0 = wd.tile_off, t1 = wd.tile_off + wd.ntiles;

  // camera table entry of the free camera at pose xx (R, J_L, t): every lane computes it, lane 0 stores it
  auto set_free_camera = [&](const double (&xx)[6]) {
    double wv[3] = { xx[0], xx[1], xx[2] }, R[9], JL[9];
    cam_prepare<double>(wv, R, JL);
    __syncthreads();
    if (lane == 0) {
      double* ct = camtab + fc * kCamTab;
#pragma unroll
      for (int q = 0; q < 9; ++q) { ct[q] = R[q]; ct[9 + q] = JL[q]; }
      ct[18] = xx[3]; ct[19] = xx[4]; ct[20] = xx[5];
    }
    __syncthreads();
  };

  double S[21], g[6], h[6];          // J'^T J' (lower triangle), J'^T r, diag(J'^T J') of the free camera at the accepted point
  bool have_lin = false;
  for (;;) {
    if (!have_lin) {
      // ---- sweep 1: linearise every observation at the accepted point (camscale: 1 in the first sweep of a solve)
      double aS[21], ag[6], cost = 0.0, fixed = 0.0;
#pragma unroll
      for (int q = 0; q < 21; ++q) aS[q] = 0.0;
#pragma unroll
      for (int q = 0; q < 6; ++q) ag[q] = 0.0;
      for (int t = t0; t < t1; ++t) {
        const TileCtx tc = fetch_tile(p, t, t1, lane);
        ObsPref pf;
        prefetch_obs<false, false>(p, tc, cur, wd.obs_off, pf, lane);
        LaneLin L;
        double ob[8];
        lane_linearise<true>(p, pol, camtab, camscale, camcf, tc.ls, tc.j, tc.k, tc.o0, tc.line_ok, tc.lflags, cur, wd.obs_off, L, ob, &pf, true);
        if (L.kept) cost += L.cost;
        else if (L.valid) fixed += L.cost;
        if (L.valid && L.cf >= 0) {
#pragma unroll
          for (int a = 0; a < 6; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ag[a] += L.Jc[6 * r + a] * L.rs[r];
#pragma unroll
            for (int b = 0; b <= a; ++b) {
#pragma unroll
              for (int r = 0; r < 4; ++r) aS[tri_index(a, b)] += L.Jc[6 * r + a] * L.Jc[6 * r + b];
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 21; ++q) S[q] = wave_sum(aS[q]);
#pragma unroll
      for (int q = 0; q < 6; ++q) { g[q] = wave_sum(ag[q]); h[q] = S[tri_index(q, q)]; }
      cost = wave_sum(cost); fixed = wave_sum(fixed);
      have_lin = true;

      if (st->fresh) {
        // ---- Ceres' initial evaluation: cost, gradient max-norm, |x|, Jacobi scale from diag(J^T J) at x0, trace record 0
        double gmax = 0.0, xn2 = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          gmax = fmax(gmax, fabs(g[a]));
          xn2 += x[a] * x[a];
          sc[a] = pol.jacobi_scaling ? 1.0 / (1.0 + sqrt(h[a])) : 1.0;
        }
        __syncthreads();
        if (lane == 0) {
          st->cost = cost; st->fixed_cost = fixed; st->initial_cost = cost + fixed; st->min_cost = cost + fixed;
          st->x_norm = sqrt(xn2);
          st->grad_max = gmax;
          st->abs_grad_tol = pol.gradient_tolerance * (gmax > 1e-12 ? gmax : 1e-12);
          st->need_grad_check = 0;
          st->fresh = 0;
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
            if (st->iter >= pol.max_num_iterations) status = 0;
          }
          st->status = status;
#pragma unroll
          for (int a = 0; a < 6; ++a) camscale[a] = sc[a];
        }
        __syncthreads();
        // the system to scaled coordinates (a congruence with diag(scale)), as the general path does
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          g[a] *= sc[a]; h[a] *= sc[a] * sc[a];
#pragma unroll
          for (int b = 0; b <= a; ++b) S[tri_index(a, b)] *= sc[a] * sc[b];
        }
      } else if (st->need_grad_check) {
        // ---- gradient max-norm at the newly accepted point (g is the scaled gradient)
        double gm = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) gm = fmax(gm, fabs(g[a] / sc[a]));
        __syncthreads();
        if (lane == 0) {
          st->grad_max = gm;
          st->need_grad_check = 0;
          if (st->ntrace > 0 && st->ntrace <= kMaxTrace) p.trace[(long long)w * kMaxTrace + st->ntrace - 1].gradient_max_norm = gm;
          if (gm <= st->abs_grad_tol) st->status = 1;
        }
        __syncthreads();
      }
      if (st->status != kRunning) break;
    }

    // ---- damped normal equations, step, candidate pose
    const double radius = st->radius;
    double M[21], D2[6], y[6], xc[6];
#pragma unroll
    for (int q = 0; q < 21; ++q) M[q] = S[q];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      D2[a] = fmin(fmax(h[a], pol.min_lm_diagonal), pol.max_lm_diagonal) / radius;
      M[tri_index(a, a)] += D2[a];
    }
    const bool ok = chol6_solve(M, g, y);
    double model = 0.0, dn2 = 0.0, xn2 = 0.0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      model += 0.5 * y[a] * (g[a] + D2[a] * y[a]);
      const double d = -y[a] * sc[a];
      xc[a] = x[a] + d;
      const double dd = x[a] - xc[a];
      dn2 += dd * dd;
      xn2 += xc[a] * xc[a];
    }

    // ---- sweep 2: cost at the candidate pose (residuals only)
    set_free_camera(xc);
    double ccost = 0.0;
    for (int t = t0; t < t1; ++t) {
      const TileCtx tc = fetch_tile(p, t, t1, lane);
      const bool valid = tc.line_ok && tc.j < tc.k;
      const int o = valid ? tc.o0 + tc.j : wd.obs_off;
      double ob[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
        ob[2 * q] = e.x; ob[2 * q + 1] = e.y;
      }
      const int cam = p.ob_cam[o];
      const int lsafe = tc.line_ok ? tc.ls : 0;
      const double* lrec = p.line_x + line_rec(p, lsafe, cur);
      double trig[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) trig[q] = lrec[4 + q];
      const double* ct = camtab + cam * kCamTab;
      double R[9], tt[3], cp[3], dv[3], r[4], c;
#pragma unroll
      for (int q = 0; q < 9; ++q) R[q] = ct[q];
      tt[0] = ct[18]; tt[1] = ct[19]; tt[2] = ct[20];
      line_points<double>(trig, cp, dv);
      obs_residual<double>(R, tt, cp, dv, ob, pol.baseline, r);
      huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &c);
      const bool line_free = !(tc.lflags & 1);
      if (valid && !(camcf[cam] < 0 && !line_free)) ccost += c;
    }
    ccost = wave_sum(ccost);

    // ---- accept / reject, radius, trace, stopping rules
    __syncthreads();
    int n_success_before = 0;
    if (lane == 0) {
      n_success_before = st->n_success;
      st->solve_failed = ok ? 0 : 1;
      lm_step(p, pol, w, st, ccost, model, dn2, xn2);
      st->pad = st->n_success != n_success_before ? 1 : 0;
    }
    __syncthreads();
    const bool accepted = st->pad != 0;
    if (accepted) {
#pragma unroll
      for (int a = 0; a < 6; ++a) x[a] = xc[a];
      have_lin = false;                 // the free camera's table already holds the candidate = new accepted pose
    }
    if (st->status != kRunning) break;
  }

  // ---- results: the accepted pose into both buffers of the camera record (lines never moved, their buffer is `cur`)
  __syncthreads();
  if (lane == 0) {
    st->cur = cur;
    st->pad = 0;
    *gst = *st;
    double* xa = p.cam_x + (long long)(wd.cam_off + fc) * 2 * kCamRec;
#pragma unroll
    for (int a = 0; a < 6; ++a) { xa[a] = x[a]; xa[kCamRec + a] = x[a]; p.cam_scale[(long long)(wd.cam_off + fc) * 6 + a] = sc[a]; }
  }
}

}  // namespace slslam
#endif
