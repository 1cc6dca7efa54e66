/* oracle/lba_oracle.c — CPU fp64 restatement of the reference's line bundle adjustment:
 *   residual functor      reference src/lba_problem.h:46-118
 *   problem wiring        reference src/lba_problem.cpp:54-93   (LBAProblem::build)
 *   solver configuration  reference src/lba_problem.cpp:95-132  (LBAProblem::set_options)
 *   call protocol         reference src/slam.cpp:924-952
 * plus the parts the reference delegates to Ceres 1.7.0 (autodiff, Huber corrector, program
 * reduction of constant blocks, LM) restated from the published algorithm.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see slslam_oracle.h).
 */
#include "lm_core.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define JN 10
#define JT jet10
#define JF(name) j10_##name
#include "jet_impl.h"
#undef JN
#undef JT
#undef JF

/* ------------------------------------------------------------------------------------------ */
/* LineReprojectionError::operator()<double>  (src/lba_problem.h:46-118)                      */
void oracle_line_residual(const double camera[6], const double line[4], const double obs[8],
                          double baseline, double residuals[4]) {
  const double a = line[0], b = line[1], g = line[2], t = line[3];        /* :50-54 */
  const double s1 = sin(a), c1 = cos(a), s2 = sin(b), c2 = cos(b), s3 = sin(g), c3 = cos(g); /* :56-61 */
  const double d = cos(t) / sin(t);                                        /* :63 */
  double cp[3], dv[3];
  cp[0] = -(c1 * s2 * c3 + s1 * s3) * d;                                   /* :66-68 */
  cp[1] = -(c1 * s2 * s3 - s1 * c3) * d;
  cp[2] = -(c1 * c2) * d;
  dv[0] = s1 * s2 * c3 - c1 * s3;                                          /* :70-72 */
  dv[1] = s1 * s2 * s3 + c1 * c3;
  dv[2] = s1 * c2;
  double pc[3], dc[3];
  {                                                                        /* :75-76 AngleAxisRotatePoint */
    const double* w = camera;
    const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double* pts[2] = { cp, dv };
    double* outs[2] = { pc, dc };
    for (int k = 0; k < 2; ++k) {
      const double* p = pts[k]; double* o = outs[k];
      if (theta2 > 0.0) {
        const double theta = sqrt(theta2);
        const double u0 = w[0] / theta, u1 = w[1] / theta, u2 = w[2] / theta;
        const double ct = cos(theta), st = sin(theta);
        const double x0 = u1 * p[2] - u2 * p[1], x1 = u2 * p[0] - u0 * p[2], x2 = u0 * p[1] - u1 * p[0];
        const double udp = u0 * p[0] + u1 * p[1] + u2 * p[2];
        o[0] = p[0] * ct + x0 * st + u0 * (1.0 - ct) * udp;
        o[1] = p[1] * ct + x1 * st + u1 * (1.0 - ct) * udp;
        o[2] = p[2] * ct + x2 * st + u2 * (1.0 - ct) * udp;
      } else {
        o[0] = p[0] + (w[1] * p[2] - w[2] * p[1]);
        o[1] = p[1] + (w[2] * p[0] - w[0] * p[2]);
        o[2] = p[2] + (w[0] * p[1] - w[1] * p[0]);
      }
    }
  }
  pc[0] += camera[3]; pc[1] += camera[4]; pc[2] += camera[5];              /* :81-83 */
  double n[3], sql;
  n[0] = pc[1] * dc[2] - pc[2] * dc[1];                                    /* :86-88 */
  n[1] = pc[2] * dc[0] - pc[0] * dc[2];
  n[2] = pc[0] * dc[1] - pc[1] * dc[0];
  sql = sqrt(n[0] * n[0] + n[1] * n[1]);                                   /* :90-93 */
  n[0] /= sql; n[1] /= sql; n[2] /= sql;
  residuals[0] = -(obs[0] * n[0] + obs[1] * n[1] + n[2]);                  /* :95-96 */
  residuals[1] = -(obs[2] * n[0] + obs[3] * n[1] + n[2]);
  pc[0] -= baseline;                                                       /* :101-103 */
  n[0] = pc[1] * dc[2] - pc[2] * dc[1];                                    /* :105-107 */
  n[1] = pc[2] * dc[0] - pc[0] * dc[2];
  n[2] = pc[0] * dc[1] - pc[1] * dc[0];
  sql = sqrt(n[0] * n[0] + n[1] * n[1]);                                   /* :109-112 */
  n[0] /= sql; n[1] /= sql; n[2] /= sql;
  residuals[2] = -(obs[4] * n[0] + obs[5] * n[1] + n[2]);                  /* :114-115 */
  residuals[3] = -(obs[6] * n[0] + obs[7] * n[1] + n[2]);
}

/* The same functor evaluated on Jet<double,10> = AutoDiffCostFunction<...,4,6,4>
 * (src/lba_problem.cpp:65-74).  Partials 0..5 = camera, 6..9 = line. */
void oracle_line_residual_jet(const double camera[6], const double line[4], const double obs[8],
                              double baseline, double residuals[4], double* j_cam, double* j_line) {
  jet10 cam[6], ln[4];
  for (int i = 0; i < 6; ++i) cam[i] = j10_var(camera[i], i);
  for (int i = 0; i < 4; ++i) ln[i] = j10_var(line[i], 6 + i);
  const jet10 a = ln[0], b = ln[1], g = ln[2], t = ln[3];
  const jet10 s1 = j10_sin(a), c1 = j10_cos(a), s2 = j10_sin(b), c2 = j10_cos(b), s3 = j10_sin(g), c3 = j10_cos(g);
  const jet10 d = j10_div(j10_cos(t), j10_sin(t));
  jet10 cp[3], dv[3];
  cp[0] = j10_mul(j10_neg(j10_add(j10_mul(j10_mul(c1, s2), c3), j10_mul(s1, s3))), d);
  cp[1] = j10_mul(j10_neg(j10_sub(j10_mul(j10_mul(c1, s2), s3), j10_mul(s1, c3))), d);
  cp[2] = j10_mul(j10_neg(j10_mul(c1, c2)), d);
  dv[0] = j10_sub(j10_mul(j10_mul(s1, s2), c3), j10_mul(c1, s3));
  dv[1] = j10_add(j10_mul(j10_mul(s1, s2), s3), j10_mul(c1, c3));
  dv[2] = j10_mul(s1, c2);
  jet10 pc[3], dc[3];
  j10_aa_rotate_point(cam, cp, pc);
  j10_aa_rotate_point(cam, dv, dc);
  pc[0] = j10_add(pc[0], cam[3]); pc[1] = j10_add(pc[1], cam[4]); pc[2] = j10_add(pc[2], cam[5]);
  jet10 res[4];
  for (int k = 0; k < 2; ++k) {
    if (k == 1) pc[0] = j10_sub(pc[0], j10_cst(baseline));
    jet10 n[3];
    n[0] = j10_sub(j10_mul(pc[1], dc[2]), j10_mul(pc[2], dc[1]));
    n[1] = j10_sub(j10_mul(pc[2], dc[0]), j10_mul(pc[0], dc[2]));
    n[2] = j10_sub(j10_mul(pc[0], dc[1]), j10_mul(pc[1], dc[0]));
    const jet10 sql = j10_sqrt(j10_add(j10_mul(n[0], n[0]), j10_mul(n[1], n[1])));
    n[0] = j10_div(n[0], sql); n[1] = j10_div(n[1], sql); n[2] = j10_div(n[2], sql);
    for (int e = 0; e < 2; ++e) {
      const double x = obs[4 * k + 2 * e], y = obs[4 * k + 2 * e + 1];
      res[2 * k + e] = j10_neg(j10_add(j10_add(j10_muls(n[0], x), j10_muls(n[1], y)), n[2]));
    }
  }
  for (int r = 0; r < 4; ++r) {
    residuals[r] = res[r].v;
    if (j_cam) for (int c = 0; c < 6; ++c) j_cam[6 * r + c] = res[r].d[c];
    if (j_line) for (int c = 0; c < 4; ++c) j_line[4 * r + c] = res[r].d[6 + c];
  }
}

/* ceres::HuberLoss::Evaluate (Ceres 1.7.0 loss_function.cc); used at src/lba_problem.cpp:78-80 */
void oracle_huber(double s, double a, double rho[3]) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    rho[0] = 2.0 * a * r - b;
    rho[1] = a / r;
    rho[2] = -rho[1] / (2.0 * s);
  } else {
    rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

/* ------------------------------------------------------------------------------------------ */

typedef struct {
  const oracle_lba_problem* p;
  double* params;          /* full parameter vector scratch [6C+4L] (user layout) */
  int *cam_slot, *line_slot; /* offset into reduced x, or -1 (constant / unused) */
  int *kept;               /* [M] residual block kept in the reduced program */
  int n_cam_free, n_line_free, n;
  double *r, *jc, *jl;     /* robustified residuals / Jacobians of kept blocks, [4M],[24M],[16M] */
  int *line_ptr, *line_obs;/* CSR: observations of each line */
  int linear_solver;
  double* work;            /* dense H or Schur scratch */
  double* hbuf; int* hslot; /* per-line gather of H_cl / W blocks (Schur) */
} lba_ctx;

/* residual block evaluation incl. the Ceres Corrector for rho'' <= 0 (Huber): both r and J are
 * scaled by sqrt(rho') (corrector.cc), block cost = rho/2 (residual_block.cc) */
static double eval_block(const oracle_lba_problem* p, const double* params, int i, int want_jac,
                         double r[4], double* jc, double* jl) {
  const double* cam = params + 6 * p->camera_index[i];                                  /* lba_problem.cpp:83 */
  const double* line = params + 6 * p->num_cameras + 4 * p->line_index[i];              /* :84 */
  const double* obs = p->observations + 8 * i;
  if (want_jac) oracle_line_residual_jet(cam, line, obs, p->baseline, r, jc, jl);
  else oracle_line_residual(cam, line, obs, p->baseline, r);
  const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
  if (p->huber_delta > 0.0) {
    double rho[3];
    oracle_huber(s, p->huber_delta, rho);
    const double sr = sqrt(rho[1]);
    if (want_jac) { for (int k = 0; k < 24; ++k) jc[k] *= sr; for (int k = 0; k < 16; ++k) jl[k] *= sr; }
    for (int k = 0; k < 4; ++k) r[k] *= sr;
    return 0.5 * rho[0];
  }
  return 0.5 * s;
}

static void scatter_x(lba_ctx* c, const double* x) {
  const oracle_lba_problem* p = c->p;
  for (int k = 0; k < p->num_cameras; ++k) if (c->cam_slot[k] >= 0) memcpy(c->params + 6 * k, x + c->cam_slot[k], 6 * sizeof(double));
  for (int k = 0; k < p->num_lines; ++k) if (c->line_slot[k] >= 0) memcpy(c->params + 6 * p->num_cameras + 4 * k, x + c->line_slot[k], 4 * sizeof(double));
}

static int lba_evaluate(void* vc, const double* x, double* cost, int want_jac, double* gradient) {
  lba_ctx* c = (lba_ctx*)vc;
  const oracle_lba_problem* p = c->p;
  scatter_x(c, x);
  double total = 0.0;
  if (want_jac && gradient) memset(gradient, 0, sizeof(double) * (size_t)c->n);
  for (int i = 0; i < p->num_observations; ++i) {
    if (!c->kept[i]) continue;
    double rr[4];
    double* r = want_jac ? c->r + 4 * i : rr;
    total += eval_block(p, c->params, i, want_jac, r, c->jc + 24 * i, c->jl + 16 * i);
    if (want_jac && gradient) {
      const int cs = c->cam_slot[p->camera_index[i]], ls = c->line_slot[p->line_index[i]];
      if (cs >= 0) for (int k = 0; k < 6; ++k) for (int q = 0; q < 4; ++q) gradient[cs + k] += c->jc[24 * i + 6 * q + k] * r[q];
      if (ls >= 0) for (int k = 0; k < 4; ++k) for (int q = 0; q < 4; ++q) gradient[ls + k] += c->jl[16 * i + 4 * q + k] * r[q];
    }
  }
  *cost = total;
  return isfinite(total) ? 1 : 0;
}

static void lba_sq_col_norm(void* vc, double* out) {
  lba_ctx* c = (lba_ctx*)vc; const oracle_lba_problem* p = c->p;
  memset(out, 0, sizeof(double) * (size_t)c->n);
  for (int i = 0; i < p->num_observations; ++i) {
    if (!c->kept[i]) continue;
    const int cs = c->cam_slot[p->camera_index[i]], ls = c->line_slot[p->line_index[i]];
    if (cs >= 0) for (int k = 0; k < 6; ++k) for (int q = 0; q < 4; ++q) { const double v = c->jc[24 * i + 6 * q + k]; out[cs + k] += v * v; }
    if (ls >= 0) for (int k = 0; k < 4; ++k) for (int q = 0; q < 4; ++q) { const double v = c->jl[16 * i + 4 * q + k]; out[ls + k] += v * v; }
  }
}

static void lba_scale_cols(void* vc, const double* s) {
  lba_ctx* c = (lba_ctx*)vc; const oracle_lba_problem* p = c->p;
  for (int i = 0; i < p->num_observations; ++i) {
    if (!c->kept[i]) continue;
    const int cs = c->cam_slot[p->camera_index[i]], ls = c->line_slot[p->line_index[i]];
    if (cs >= 0) for (int q = 0; q < 4; ++q) for (int k = 0; k < 6; ++k) c->jc[24 * i + 6 * q + k] *= s[cs + k];
    if (ls >= 0) for (int q = 0; q < 4; ++q) for (int k = 0; k < 4; ++k) c->jl[16 * i + 4 * q + k] *= s[ls + k];
  }
}

static double lba_model_cost_change(void* vc, const double* step) {
  lba_ctx* c = (lba_ctx*)vc; const oracle_lba_problem* p = c->p;
  double acc = 0.0;
  for (int i = 0; i < p->num_observations; ++i) {
    if (!c->kept[i]) continue;
    const int cs = c->cam_slot[p->camera_index[i]], ls = c->line_slot[p->line_index[i]];
    for (int q = 0; q < 4; ++q) {
      double m = 0.0;
      if (cs >= 0) for (int k = 0; k < 6; ++k) m += c->jc[24 * i + 6 * q + k] * step[cs + k];
      if (ls >= 0) for (int k = 0; k < 4; ++k) m += c->jl[16 * i + 4 * q + k] * step[ls + k];
      acc += m * (c->r[4 * i + q] + 0.5 * m);
    }
  }
  return -acc;
}

/* (a) dense normal equations: what SPARSE_NORMAL_CHOLESKY computes, without the sparsity
 * (reference always ends up with SPARSE_NORMAL_CHOLESKY: the switch at lba_problem.cpp:96-101
 * falls through) */
static int lba_solve_dense(lba_ctx* c, const double* lm_diag, double* y) {
  const oracle_lba_problem* p = c->p; const int n = c->n;
  double* H = c->work;
  memset(H, 0, sizeof(double) * (size_t)n * n);
  memset(y, 0, sizeof(double) * (size_t)n);
  for (int i = 0; i < p->num_observations; ++i) {
    if (!c->kept[i]) continue;
    const int cs = c->cam_slot[p->camera_index[i]], ls = c->line_slot[p->line_index[i]];
    int idx[10]; double col[10][4]; int m = 0;
    if (cs >= 0) for (int k = 0; k < 6; ++k) { idx[m] = cs + k; for (int q = 0; q < 4; ++q) col[m][q] = c->jc[24 * i + 6 * q + k]; ++m; }
    if (ls >= 0) for (int k = 0; k < 4; ++k) { idx[m] = ls + k; for (int q = 0; q < 4; ++q) col[m][q] = c->jl[16 * i + 4 * q + k]; ++m; }
    for (int a = 0; a < m; ++a) {
      double ga = 0; for (int q = 0; q < 4; ++q) ga += col[a][q] * c->r[4 * i + q];
      y[idx[a]] += ga;
      for (int b = 0; b < m; ++b) {
        double h = 0; for (int q = 0; q < 4; ++q) h += col[a][q] * col[b][q];
        H[(size_t)idx[a] * n + idx[b]] += h;
      }
    }
  }
  for (int i = 0; i < n; ++i) H[(size_t)i * n + i] += lm_diag[i] * lm_diag[i];
  if (oracle_dense_cholesky(H, n)) return 1;
  oracle_dense_cholesky_solve(H, n, y);
  return 0;
}

/* (b) block Schur complement: eliminate every free line (4x4), solve the reduced camera system
 * densely, back-substitute.  Algebraically identical to (a); this is the structure the HIP path
 * uses, and the only one that is tractable at the 2000-line configs. */
static void chol4_inv(const double a[16], double inv[16], int* fail) {
  double l[16]; memcpy(l, a, sizeof(l));
  if (oracle_dense_cholesky(l, 4)) { *fail = 1; return; }
  for (int col = 0; col < 4; ++col) {
    double e[4] = { 0, 0, 0, 0 }; e[col] = 1.0;
    oracle_dense_cholesky_solve(l, 4, e);
    for (int r = 0; r < 4; ++r) inv[4 * r + col] = e[r];
  }
}

static int lba_solve_schur(lba_ctx* c, const double* lm_diag, double* y) {
  const oracle_lba_problem* p = c->p;
  const int nc = 6 * c->n_cam_free;
  double* S = c->work;                  /* nc x nc */
  double* bc = S + (size_t)nc * nc;     /* nc */
  memset(S, 0, sizeof(double) * ((size_t)nc * nc + nc));
  /* camera-camera blocks and camera gradient */
  for (int i = 0; i < p->num_observations; ++i) {
    if (!c->kept[i]) continue;
    const int cs = c->cam_slot[p->camera_index[i]];
    if (cs < 0) continue;
    const double* jc = c->jc + 24 * i; const double* r = c->r + 4 * i;
    for (int a = 0; a < 6; ++a) {
      for (int q = 0; q < 4; ++q) bc[cs + a] += jc[6 * q + a] * r[q];
      for (int b = 0; b < 6; ++b) { double h = 0; for (int q = 0; q < 4; ++q) h += jc[6 * q + a] * jc[6 * q + b]; S[(size_t)(cs + a) * nc + cs + b] += h; }
    }
  }
  for (int i = 0; i < nc; ++i) S[(size_t)i * nc + i] += lm_diag[i] * lm_diag[i];
  /* eliminate lines */
  int fail = 0;
  for (int l = 0; l < p->num_lines; ++l) {
    const int ls = c->line_slot[l];
    if (ls < 0) continue;
    double A[16] = { 0 }, gl[4] = { 0 }, Ainv[16];
    for (int e = c->line_ptr[l]; e < c->line_ptr[l + 1]; ++e) {
      const int i = c->line_obs[e]; if (!c->kept[i]) continue;
      const double* jl = c->jl + 16 * i; const double* r = c->r + 4 * i;
      for (int a = 0; a < 4; ++a) { for (int q = 0; q < 4; ++q) gl[a] += jl[4 * q + a] * r[q];
        for (int b = 0; b < 4; ++b) { double h = 0; for (int q = 0; q < 4; ++q) h += jl[4 * q + a] * jl[4 * q + b]; A[4 * a + b] += h; } }
    }
    for (int a = 0; a < 4; ++a) A[4 * a + a] += lm_diag[ls + a] * lm_diag[ls + a];
    chol4_inv(A, Ainv, &fail);
    if (fail) return 1;
    double Ag[4]; for (int a = 0; a < 4; ++a) { Ag[a] = 0; for (int b = 0; b < 4; ++b) Ag[a] += Ainv[4 * a + b] * gl[b]; }
    /* gather E_l = [H_cl,i] for the free cameras observing this line, W_i = H_cl,i Ainv */
    int k = 0;
    for (int e = c->line_ptr[l]; e < c->line_ptr[l + 1]; ++e) {
      const int i = c->line_obs[e]; if (!c->kept[i]) continue;
      const int ci = c->cam_slot[p->camera_index[i]]; if (ci < 0) continue;
      double* Hi = c->hbuf + 48 * (size_t)k; double* W = Hi + 24;
      for (int a = 0; a < 6; ++a) for (int b = 0; b < 4; ++b) { double h = 0; for (int q = 0; q < 4; ++q) h += c->jc[24 * i + 6 * q + a] * c->jl[16 * i + 4 * q + b]; Hi[4 * a + b] = h; }
      for (int a = 0; a < 6; ++a) for (int b = 0; b < 4; ++b) { double h = 0; for (int m = 0; m < 4; ++m) h += Hi[4 * a + m] * Ainv[4 * m + b]; W[4 * a + b] = h; }
      for (int a = 0; a < 6; ++a) { double h = 0; for (int b = 0; b < 4; ++b) h += Hi[4 * a + b] * Ag[b]; bc[ci + a] -= h; }
      c->hslot[k++] = ci;
    }
    for (int i = 0; i < k; ++i) for (int j = 0; j < k; ++j) {
      const double* W = c->hbuf + 48 * (size_t)i + 24; const double* Hj = c->hbuf + 48 * (size_t)j;
      double* Sb = S + (size_t)c->hslot[i] * nc + c->hslot[j];
      for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) {
        double h = 0; for (int m = 0; m < 4; ++m) h += W[4 * a + m] * Hj[4 * b + m];
        Sb[(size_t)a * nc + b] -= h;
      }
    }
  }
  if (nc > 0) {
    if (oracle_dense_cholesky(S, nc)) return 1;
    oracle_dense_cholesky_solve(S, nc, bc);
    memcpy(y, bc, sizeof(double) * (size_t)nc);
  }
  /* back-substitution */
  for (int l = 0; l < p->num_lines; ++l) {
    const int ls = c->line_slot[l];
    if (ls < 0) continue;
    double A[16] = { 0 }, rhs[4] = { 0 }, Ainv[16];
    for (int e = c->line_ptr[l]; e < c->line_ptr[l + 1]; ++e) {
      const int i = c->line_obs[e]; if (!c->kept[i]) continue;
      const double* jl = c->jl + 16 * i; const double* r = c->r + 4 * i;
      const int ci = c->cam_slot[p->camera_index[i]];
      for (int a = 0; a < 4; ++a) {
        for (int q = 0; q < 4; ++q) {
          double rq = r[q];
          if (ci >= 0) for (int k = 0; k < 6; ++k) rq -= c->jc[24 * i + 6 * q + k] * y[ci + k];
          rhs[a] += jl[4 * q + a] * rq;
        }
        for (int b = 0; b < 4; ++b) { double h = 0; for (int q = 0; q < 4; ++q) h += jl[4 * q + a] * jl[4 * q + b]; A[4 * a + b] += h; }
      }
    }
    for (int a = 0; a < 4; ++a) A[4 * a + a] += lm_diag[ls + a] * lm_diag[ls + a];
    chol4_inv(A, Ainv, &fail);
    if (fail) return 1;
    for (int a = 0; a < 4; ++a) { double h = 0; for (int b = 0; b < 4; ++b) h += Ainv[4 * a + b] * rhs[b]; y[ls + a] = h; }
  }
  return 0;
}

static int lba_solve(void* vc, const double* lm_diag, double* y) {
  lba_ctx* c = (lba_ctx*)vc;
  return c->linear_solver == 1 ? lba_solve_schur(c, lm_diag, y) : lba_solve_dense(c, lm_diag, y);
}

/* ------------------------------------------------------------------------------------------ */

static void block_constness(const oracle_lba_problem* p, int* cam_const, int* line_const, int* cam_used, int* line_used) {
  /* SetParameterBlockConstant is per block (lba_problem.cpp:88-91): one flagged observation
   * makes the block constant for every residual that touches it. */
  memset(cam_const, 0, sizeof(int) * (size_t)p->num_cameras);
  memset(line_const, 0, sizeof(int) * (size_t)p->num_lines);
  memset(cam_used, 0, sizeof(int) * (size_t)p->num_cameras);
  memset(line_used, 0, sizeof(int) * (size_t)p->num_lines);
  for (int i = 0; i < p->num_observations; ++i) {
    cam_used[p->camera_index[i]] = 1; line_used[p->line_index[i]] = 1;
    if (p->fixed_index[2 * i]) cam_const[p->camera_index[i]] = 1;
    if (p->fixed_index[2 * i + 1]) line_const[p->line_index[i]] = 1;
  }
}

double oracle_lba_cost(const oracle_lba_problem* p, const double* params,
                       double* residuals, double* j_cam, double* j_line) {
  double total = 0.0;
  for (int i = 0; i < p->num_observations; ++i) {
    double r[4], jc[24], jl[16];
    const int wj = (j_cam || j_line);
    total += eval_block(p, params, i, wj, r, jc, jl);
    if (residuals) memcpy(residuals + 4 * i, r, sizeof(r));
    if (j_cam) memcpy(j_cam + 24 * i, jc, sizeof(jc));
    if (j_line) memcpy(j_line + 16 * i, jl, sizeof(jl));
  }
  return total;
}

int oracle_lba_solve(const oracle_lba_problem* p, const oracle_lm_options* opt, double* params,
                     oracle_summary* summary, oracle_iteration* trace, int trace_cap, int* trace_len) {
  const int C = p->num_cameras, L = p->num_lines, M = p->num_observations;
  lba_ctx c; memset(&c, 0, sizeof(c));
  c.p = p;
  int* ibuf = (int*)calloc((size_t)(3 * C + 3 * L + 2 * M + L + 2), sizeof(int));
  int* cam_const = ibuf, *cam_used = ibuf + C; c.cam_slot = ibuf + 2 * C;
  int* line_const = ibuf + 3 * C, *line_used = line_const + L; c.line_slot = line_const + 2 * L;
  c.kept = c.line_slot + L; c.line_obs = c.kept + M; c.line_ptr = c.line_obs + M;
  block_constness(p, cam_const, line_const, cam_used, line_used);

  /* program reduction (Ceres SolverImpl::RemoveFixedBlocksFromProgram): residual blocks whose
   * parameter blocks are all constant leave the program; their cost becomes fixed_cost. */
  memset(summary, 0, sizeof(*summary));
  double fixed_cost = 0.0; int kept_blocks = 0;
  for (int i = 0; i < M; ++i) {
    const int cc = cam_const[p->camera_index[i]], lc = line_const[p->line_index[i]];
    if (cc && lc) { double r[4]; fixed_cost += eval_block(p, params, i, 0, r, NULL, NULL); c.kept[i] = 0; }
    else { c.kept[i] = 1; ++kept_blocks; }
  }
  int n = 0;
  for (int k = 0; k < C; ++k) { if (cam_used[k] && !cam_const[k]) { c.cam_slot[k] = n; n += 6; ++c.n_cam_free; } else c.cam_slot[k] = -1; }
  for (int k = 0; k < L; ++k) { if (line_used[k] && !line_const[k]) { c.line_slot[k] = n; n += 4; ++c.n_line_free; } else c.line_slot[k] = -1; }
  c.n = n;
  summary->fixed_cost = fixed_cost;
  summary->num_free_parameters = n;
  summary->num_residual_blocks = kept_blocks;
  if (n == 0) {   /* "No non-constant parameter blocks found" */
    summary->initial_cost = summary->final_cost = fixed_cost;
    summary->termination_type = ORACLE_FUNCTION_TOLERANCE;
    if (trace_len) *trace_len = 0;
    free(ibuf);
    return 0;
  }
  /* CSR of observations per line */
  for (int i = 0; i < M; ++i) c.line_ptr[p->line_index[i] + 1]++;
  for (int k = 0; k < L; ++k) c.line_ptr[k + 1] += c.line_ptr[k];
  { int* fill = (int*)calloc((size_t)L + 1, sizeof(int));
    for (int i = 0; i < M; ++i) { const int l = p->line_index[i]; c.line_obs[c.line_ptr[l] + fill[l]++] = i; }
    free(fill); }

  c.linear_solver = opt->linear_solver;
  const size_t nc = (size_t)6 * c.n_cam_free;
  const size_t work = c.linear_solver == 1 ? nc * nc + nc + 16 : (size_t)n * n + 16;
  c.work = (double*)malloc(sizeof(double) * work);
  c.params = (double*)malloc(sizeof(double) * (size_t)(6 * C + 4 * L));
  memcpy(c.params, params, sizeof(double) * (size_t)(6 * C + 4 * L));
  { int maxk = 1; for (int k = 0; k < L; ++k) { const int d = c.line_ptr[k + 1] - c.line_ptr[k]; if (d > maxk) maxk = d; }
    c.hbuf = (double*)malloc(sizeof(double) * 48 * (size_t)maxk); c.hslot = (int*)malloc(sizeof(int) * (size_t)maxk); }
  c.r = (double*)malloc(sizeof(double) * (size_t)44 * (M > 0 ? M : 1));
  c.jc = c.r + 4 * (size_t)M; c.jl = c.jc + 24 * (size_t)M;

  double* x = (double*)malloc(sizeof(double) * (size_t)n);
  for (int k = 0; k < C; ++k) if (c.cam_slot[k] >= 0) memcpy(x + c.cam_slot[k], params + 6 * k, 6 * sizeof(double));
  for (int k = 0; k < L; ++k) if (c.line_slot[k] >= 0) memcpy(x + c.line_slot[k], params + 6 * C + 4 * k, 4 * sizeof(double));

  oracle_nlls P = { n, &c, lba_evaluate, lba_sq_col_norm, lba_scale_cols, lba_solve, lba_model_cost_change };
  const int rc = oracle_lm_minimize(&P, opt, x, summary, trace, trace_cap, trace_len);
  /* Ceres leaves user state untouched on NUMERICAL_FAILURE (solver_impl.cc) */
  if (summary->termination_type != ORACLE_NUMERICAL_FAILURE) {
    for (int k = 0; k < C; ++k) if (c.cam_slot[k] >= 0) memcpy(params + 6 * k, x + c.cam_slot[k], 6 * sizeof(double));
    for (int k = 0; k < L; ++k) if (c.line_slot[k] >= 0) memcpy(params + 6 * C + 4 * k, x + c.line_slot[k], 4 * sizeof(double));
  }
  free(x); free(c.r); free(c.params); free(c.work); free(c.hbuf); free(c.hslot); free(ibuf);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* gc_av_to_orth / gc_orth_to_av  (reference src/gc.cpp:361-379, :419-442) */
void oracle_av_to_orth(const double av[6], double orth[4]) {
  const double* a = av; const double* v = av + 3;
  const double n[3] = { a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0] };
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const double x[3] = { n[0] / nn, n[1] / nn, n[2] / nn };
  const double y[3] = { v[0] / vn, v[1] / vn, v[2] / vn };
  const double z2 = x[0] * y[1] - x[1] * y[0];
  orth[0] = atan2(y[2], z2);
  orth[1] = asin(-x[2]);
  orth[2] = atan2(x[1], x[0]);
  const double wn = sqrt(nn * nn + vn * vn);
  orth[3] = asin(vn / wn);
}

void oracle_orth_to_av(const double orth[4], double av[6]) {
  const double a = orth[0], b = orth[1], g = orth[2], t = orth[3];
  const double s1 = sin(a), c1 = cos(a), s2 = sin(b), c2 = cos(b), s3 = sin(g), c3 = cos(g);
  const double d = cos(t) / sin(t);
  av[0] = -(c1 * s2 * c3 + s1 * s3) * d;
  av[1] = -(c1 * s2 * s3 - s1 * c3) * d;
  av[2] = -(c1 * c2) * d;
  av[3] = s1 * s2 * c3 - c1 * s3;
  av[4] = s1 * s2 * s3 + c1 * c3;
  av[5] = s1 * c2;
}

/* Fan-out of independent windows over host cores: each window goes through
 * oracle_lba_solve unchanged, one window per OpenMP task.  Used only by bench.py's cpu_baseline leg. */
int oracle_lba_solve_many(int count, const oracle_lba_problem* problems, const oracle_lm_options* opt,
                          double* const* params, oracle_summary* summaries, int num_threads) {
  int rc_any = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads) reduction(| : rc_any)
#endif
  for (int i = 0; i < count; ++i)
    rc_any |= oracle_lba_solve(&problems[i], opt, params[i], &summaries[i], 0, 0, 0);
  (void)num_threads;
  return rc_any;
}
