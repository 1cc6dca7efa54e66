/* oracle/po_oracle.c — CPU fp64 restatement of the reference's pose-graph optimisation:
 *   SE(3) helpers         reference src/po_problem.h:27-64  (gc_T_inv, gc_w_20, gc_T_20 templates)
 *   residual functor      reference src/po_problem.h:74-105 (PoseConstraintError::operator())
 *   problem wiring        reference src/po_problem.cpp:40-65 (POProblem::build; pose1 of edge 0 constant)
 *   solver configuration  reference src/po_problem.cpp:67-77 (SPARSE_NORMAL_CHOLESKY, silent)
 *   call protocol         reference src/slam.cpp:1283-1293   (10 iterations)
 * The Ceres 1.7.0 pieces (Jet autodiff, rotation.h, LM) are restated from the published source.
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see slslam_oracle.h).
 */
#include "lm_core.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define JN 12
#define JT jet12
#define JF(name) j12_##name
#include "jet_impl.h"
#undef JN
#undef JT
#undef JF

/* gc_T_inv<T>  (src/po_problem.h:27-39) */
static void j12_T_inv(const jet12 P[6], jet12 Pi[6]) {
  Pi[0] = j12_neg(P[0]); Pi[1] = j12_neg(P[1]); Pi[2] = j12_neg(P[2]);
  jet12 v[3] = { j12_neg(P[3]), j12_neg(P[4]), j12_neg(P[5]) };
  j12_aa_rotate_point(Pi, v, Pi + 3);
}
/* gc_w_20<T>  (src/po_problem.h:42-52): R20 = R21 R10 through quaternions */
static void j12_w_20(const jet12 w21[3], const jet12 w10[3], jet12 w20[3]) {
  jet12 q21[4], q10[4], q20[4];
  j12_aa_to_quat(w21, q21);
  j12_aa_to_quat(w10, q10);
  j12_quat_product(q21, q10, q20);
  j12_quat_to_aa(q20, w20);
}
/* gc_T_20<T>  (src/po_problem.h:55-64): T20 = T21 T10 */
static void j12_T_20(const jet12 T21[6], const jet12 T10[6], jet12 T20[6]) {
  j12_w_20(T21, T10, T20);
  j12_aa_rotate_point(T21, T10 + 3, T20 + 3);
  T20[3] = j12_add(T20[3], T21[3]); T20[4] = j12_add(T20[4], T21[4]); T20[5] = j12_add(T20[5], T21[5]);
}

/* PoseConstraintError::operator()<Jet<double,12>>  (src/po_problem.h:74-105)
 *   Tc = C * T1;  Te = T2^-1 * Tc;  residual = Te */
void oracle_pose_residual_jet(const double pose1[6], const double pose2[6], const double c[6],
                              double residuals[6], double* j1, double* j2) {
  jet12 T1[6], T2[6], C[6], Tc[6], Te[6], T2i[6];
  for (int i = 0; i < 6; ++i) { T1[i] = j12_var(pose1[i], i); T2[i] = j12_var(pose2[i], 6 + i); C[i] = j12_cst(c[i]); }
  j12_T_20(C, T1, Tc);      /* :93 */
  j12_T_inv(T2, T2i);       /* :94 */
  j12_T_20(T2i, Tc, Te);    /* :95 */
  for (int r = 0; r < 6; ++r) {
    residuals[r] = Te[r].v;
    if (j1) for (int k = 0; k < 6; ++k) j1[6 * r + k] = Te[r].d[k];
    if (j2) for (int k = 0; k < 6; ++k) j2[6 * r + k] = Te[r].d[6 + k];
  }
}

void oracle_pose_residual(const double pose1[6], const double pose2[6], const double c[6], double residuals[6]) {
  oracle_pose_residual_jet(pose1, pose2, c, residuals, NULL, NULL);
}

typedef struct {
  const oracle_po_problem* p;
  double* params;
  int* slot;       /* pose -> offset in reduced x or -1 */
  int* kept;       /* edge kept */
  int n;
  double *r, *j1, *j2;  /* [6E], [36E], [36E] */
  double* H;
  /* envelope (skyline) storage of the normal matrix, linear_solver == 2: row i keeps columns first[i] .. i */
  int* first;
  long long* rowptr;
  double* sky;
} po_ctx;

static int po_evaluate(void* vc, const double* x, double* cost, int want_jac, double* gradient) {
  po_ctx* c = (po_ctx*)vc; const oracle_po_problem* p = c->p;
  for (int k = 0; k < p->num_poses; ++k) if (c->slot[k] >= 0) memcpy(c->params + 6 * k, x + c->slot[k], 6 * sizeof(double));
  double total = 0.0;
  if (want_jac && gradient) memset(gradient, 0, sizeof(double) * (size_t)c->n);
  for (int e = 0; e < p->num_edges; ++e) {
    if (!c->kept[e]) continue;
    double rr[6];
    double* r = want_jac ? c->r + 6 * e : rr;
    const int a = p->pose_index_1[e], b = p->pose_index_2[e];
    oracle_pose_residual_jet(c->params + 6 * a, c->params + 6 * b, p->constraints + 6 * e, r,
                             want_jac ? c->j1 + 36 * e : NULL, want_jac ? c->j2 + 36 * e : NULL);
    double s = 0; for (int q = 0; q < 6; ++q) s += r[q] * r[q];
    total += 0.5 * s;   /* no loss function: robustify = false (po_problem.cpp:27,55) */
    if (want_jac && gradient) {
      const int sa = c->slot[a], sb = c->slot[b];
      if (sa >= 0) for (int k = 0; k < 6; ++k) for (int q = 0; q < 6; ++q) gradient[sa + k] += c->j1[36 * e + 6 * q + k] * r[q];
      if (sb >= 0) for (int k = 0; k < 6; ++k) for (int q = 0; q < 6; ++q) gradient[sb + k] += c->j2[36 * e + 6 * q + k] * r[q];
    }
  }
  *cost = total;
  return isfinite(total) ? 1 : 0;
}

static void po_sq_col_norm(void* vc, double* out) {
  po_ctx* c = (po_ctx*)vc; const oracle_po_problem* p = c->p;
  memset(out, 0, sizeof(double) * (size_t)c->n);
  for (int e = 0; e < p->num_edges; ++e) {
    if (!c->kept[e]) continue;
    const int sa = c->slot[p->pose_index_1[e]], sb = c->slot[p->pose_index_2[e]];
    if (sa >= 0) for (int k = 0; k < 6; ++k) for (int q = 0; q < 6; ++q) { const double v = c->j1[36 * e + 6 * q + k]; out[sa + k] += v * v; }
    if (sb >= 0) for (int k = 0; k < 6; ++k) for (int q = 0; q < 6; ++q) { const double v = c->j2[36 * e + 6 * q + k]; out[sb + k] += v * v; }
  }
}

static void po_scale_cols(void* vc, const double* s) {
  po_ctx* c = (po_ctx*)vc; const oracle_po_problem* p = c->p;
  for (int e = 0; e < p->num_edges; ++e) {
    if (!c->kept[e]) continue;
    const int sa = c->slot[p->pose_index_1[e]], sb = c->slot[p->pose_index_2[e]];
    if (sa >= 0) for (int q = 0; q < 6; ++q) for (int k = 0; k < 6; ++k) c->j1[36 * e + 6 * q + k] *= s[sa + k];
    if (sb >= 0) for (int q = 0; q < 6; ++q) for (int k = 0; k < 6; ++k) c->j2[36 * e + 6 * q + k] *= s[sb + k];
  }
}

static double po_model_cost_change(void* vc, const double* step) {
  po_ctx* c = (po_ctx*)vc; const oracle_po_problem* p = c->p;
  double acc = 0.0;
  for (int e = 0; e < p->num_edges; ++e) {
    if (!c->kept[e]) continue;
    const int sa = c->slot[p->pose_index_1[e]], sb = c->slot[p->pose_index_2[e]];
    for (int q = 0; q < 6; ++q) {
      double m = 0.0;
      if (sa >= 0) for (int k = 0; k < 6; ++k) m += c->j1[36 * e + 6 * q + k] * step[sa + k];
      if (sb >= 0) for (int k = 0; k < 6; ++k) m += c->j2[36 * e + 6 * q + k] * step[sb + k];
      acc += m * (c->r[6 * e + q] + 0.5 * m);
    }
  }
  return -acc;
}

static int po_solve(void* vc, const double* lm_diag, double* y) {
  po_ctx* c = (po_ctx*)vc; const oracle_po_problem* p = c->p; const int n = c->n;
  double* H = c->H;
  memset(H, 0, sizeof(double) * (size_t)n * n);
  memset(y, 0, sizeof(double) * (size_t)n);
  for (int e = 0; e < p->num_edges; ++e) {
    if (!c->kept[e]) continue;
    const int s[2] = { c->slot[p->pose_index_1[e]], c->slot[p->pose_index_2[e]] };
    const double* J[2] = { c->j1 + 36 * e, c->j2 + 36 * e };
    const double* r = c->r + 6 * e;
    for (int u = 0; u < 2; ++u) {
      if (s[u] < 0) continue;
      for (int a = 0; a < 6; ++a) {
        double g = 0; for (int q = 0; q < 6; ++q) g += J[u][6 * q + a] * r[q];
        y[s[u] + a] += g;
        for (int v = 0; v < 2; ++v) {
          if (s[v] < 0) continue;
          /* a self-loop edge (pose1 == pose2) would alias; Ceres would reject duplicate blocks */
          for (int b = 0; b < 6; ++b) {
            double h = 0; for (int q = 0; q < 6; ++q) h += J[u][6 * q + a] * J[v][6 * q + b];
            H[(size_t)(s[u] + a) * n + s[v] + b] += h;
          }
        }
      }
    }
  }
  for (int i = 0; i < n; ++i) H[(size_t)i * n + i] += lm_diag[i] * lm_diag[i];
  if (oracle_dense_cholesky(H, n)) return 1;
  oracle_dense_cholesky_solve(H, n, y);
  return 0;
}

/* Sparse linear solver (opt->linear_solver == 2): what SPARSE_NORMAL_CHOLESKY (reference src/po_problem.cpp:68) exploits on a
 * pose graph - the normal matrix of a chain of odometry edges is block tridiagonal and a loop closure (a, b) adds one block far
 * from the diagonal.  Envelope (skyline) Cholesky in the natural pose order: row i stores columns first[i] .. i, fill stays
 * inside the envelope, so only the 6 rows of a loop closure's later pose are long.  260 poses / 8 loop closures: ~30 MFLOP per
 * factorisation instead of the 1.25 GFLOP of the dense n^3 / 3.  Same normal equations, same LM loop: results equal the
 * dense solver's to round-off (tests/test_oracle.py). */
static int po_solve_skyline(void* vc, const double* lm_diag, double* y) {
  po_ctx* c = (po_ctx*)vc; const oracle_po_problem* p = c->p; const int n = c->n;
  const int* first = c->first; const long long* rp = c->rowptr; double* A = c->sky;
  memset(A, 0, sizeof(double) * (size_t)rp[n]);
  memset(y, 0, sizeof(double) * (size_t)n);
#define SKY(i, j) A[rp[i] + ((j) - first[i])]
  for (int e = 0; e < p->num_edges; ++e) {
    if (!c->kept[e]) continue;
    const int s[2] = { c->slot[p->pose_index_1[e]], c->slot[p->pose_index_2[e]] };
    const double* J[2] = { c->j1 + 36 * e, c->j2 + 36 * e };
    const double* r = c->r + 6 * e;
    for (int u = 0; u < 2; ++u) {
      if (s[u] < 0) continue;
      for (int a = 0; a < 6; ++a) {
        double g = 0; for (int q = 0; q < 6; ++q) g += J[u][6 * q + a] * r[q];
        y[s[u] + a] += g;
        for (int v = 0; v < 2; ++v) {
          if (s[v] < 0) continue;
          for (int b = 0; b < 6; ++b) {
            if (s[v] + b > s[u] + a) continue;                 /* lower triangle only */
            double h = 0; for (int q = 0; q < 6; ++q) h += J[u][6 * q + a] * J[v][6 * q + b];
            SKY(s[u] + a, s[v] + b) += h;
          }
        }
      }
    }
  }
  for (int i = 0; i < n; ++i) SKY(i, i) += lm_diag[i] * lm_diag[i];
  for (int i = 0; i < n; ++i) {
    for (int j = first[i]; j <= i; ++j) {
      double sum = SKY(i, j);
      const int k0 = first[i] > first[j] ? first[i] : first[j];
      for (int k = k0; k < j; ++k) sum -= SKY(i, k) * SKY(j, k);
      if (j < i) SKY(i, j) = sum / SKY(j, j);
      else { if (!(sum > 0.0) || !isfinite(sum)) return 1; SKY(i, i) = sqrt(sum); }
    }
  }
  for (int i = 0; i < n; ++i) {                                 /* L z = g */
    double sum = y[i];
    for (int k = first[i]; k < i; ++k) sum -= SKY(i, k) * y[k];
    y[i] = sum / SKY(i, i);
  }
  for (int i = n - 1; i >= 0; --i) {                            /* L^T x = z, column sweep over row i */
    y[i] /= SKY(i, i);
    for (int k = first[i]; k < i; ++k) y[k] -= SKY(i, k) * y[i];
  }
#undef SKY
  return 0;
}

double oracle_po_cost(const oracle_po_problem* p, const double* params) {
  double total = 0.0;
  for (int e = 0; e < p->num_edges; ++e) {
    double r[6];
    oracle_pose_residual(params + 6 * p->pose_index_1[e], params + 6 * p->pose_index_2[e], p->constraints + 6 * e, r);
    for (int q = 0; q < 6; ++q) total += 0.5 * r[q] * r[q];
  }
  return total;
}

int oracle_po_solve(const oracle_po_problem* p, const oracle_lm_options* opt, double* params,
                    oracle_summary* summary, oracle_iteration* trace, int trace_cap, int* trace_len) {
  const int N = p->num_poses, E = p->num_edges;
  po_ctx c; memset(&c, 0, sizeof(c));
  c.p = p;
  int* ibuf = (int*)calloc((size_t)(2 * N + E + 1), sizeof(int));
  int* used = ibuf; c.slot = ibuf + N; c.kept = ibuf + 2 * N;
  memset(summary, 0, sizeof(*summary));
  if (trace_len) *trace_len = 0;
  if (E == 0) { summary->termination_type = ORACLE_FUNCTION_TOLERANCE; free(ibuf); return 0; }
  const int gauge = p->pose_index_1[0];     /* po_problem.cpp:62-63 */
  for (int e = 0; e < E; ++e) { used[p->pose_index_1[e]] = 1; used[p->pose_index_2[e]] = 1; }
  int n = 0;
  for (int k = 0; k < N; ++k) { if (used[k] && k != gauge) { c.slot[k] = n; n += 6; } else c.slot[k] = -1; }
  c.n = n;
  double fixed_cost = 0.0; int kept = 0;
  for (int e = 0; e < E; ++e) {
    if (c.slot[p->pose_index_1[e]] < 0 && c.slot[p->pose_index_2[e]] < 0) {
      double r[6];
      oracle_pose_residual(params + 6 * p->pose_index_1[e], params + 6 * p->pose_index_2[e], p->constraints + 6 * e, r);
      for (int q = 0; q < 6; ++q) fixed_cost += 0.5 * r[q] * r[q];
      c.kept[e] = 0;
    } else { c.kept[e] = 1; ++kept; }
  }
  summary->fixed_cost = fixed_cost;
  summary->num_free_parameters = n;
  summary->num_residual_blocks = kept;
  if (n == 0) { summary->initial_cost = summary->final_cost = fixed_cost; summary->termination_type = ORACLE_FUNCTION_TOLERANCE; free(ibuf); return 0; }
  c.params = (double*)malloc(sizeof(double) * (size_t)6 * N);
  memcpy(c.params, params, sizeof(double) * (size_t)6 * N);
  c.r = (double*)malloc(sizeof(double) * (size_t)78 * E);
  c.j1 = c.r + 6 * (size_t)E; c.j2 = c.j1 + 36 * (size_t)E;
  const int sparse = opt && opt->linear_solver == 2;
  if (sparse) {
    c.first = (int*)malloc(sizeof(int) * (size_t)n);
    c.rowptr = (long long*)malloc(sizeof(long long) * ((size_t)n + 1));
    for (int i = 0; i < n; ++i) c.first[i] = (i / 6) * 6;         /* a pose's own block */
    for (int e = 0; e < E; ++e) {
      if (!c.kept[e]) continue;
      const int sa = c.slot[p->pose_index_1[e]], sb = c.slot[p->pose_index_2[e]];
      if (sa < 0 || sb < 0) continue;
      const int hi = sa > sb ? sa : sb, lo = sa > sb ? sb : sa;
      for (int a = 0; a < 6; ++a) if (c.first[hi + a] > lo) c.first[hi + a] = lo;
    }
    c.rowptr[0] = 0;
    for (int i = 0; i < n; ++i) c.rowptr[i + 1] = c.rowptr[i] + (i - c.first[i] + 1);
    c.sky = (double*)malloc(sizeof(double) * (size_t)c.rowptr[n]);
    c.H = NULL;
  } else {
    c.H = (double*)malloc(sizeof(double) * (size_t)n * n);
  }
  double* x = (double*)malloc(sizeof(double) * (size_t)n);
  for (int k = 0; k < N; ++k) if (c.slot[k] >= 0) memcpy(x + c.slot[k], params + 6 * k, 6 * sizeof(double));
  oracle_nlls P = { n, &c, po_evaluate, po_sq_col_norm, po_scale_cols, sparse ? po_solve_skyline : po_solve, po_model_cost_change };
  const int rc = oracle_lm_minimize(&P, opt, x, summary, trace, trace_cap, trace_len);
  if (summary->termination_type != ORACLE_NUMERICAL_FAILURE)
    for (int k = 0; k < N; ++k) if (c.slot[k] >= 0) memcpy(params + 6 * k, x + c.slot[k], 6 * sizeof(double));
  free(x); free(c.H); free(c.first); free(c.rowptr); free(c.sky); free(c.r); free(c.params); free(ibuf);
  return rc;
}
