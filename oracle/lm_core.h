/* oracle/lm_core.h — internal: problem-agnostic trust-region driver used by both oracles.
 * TEST INFRASTRUCTURE ONLY. */
#ifndef ORACLE_LM_CORE_H_
#define ORACLE_LM_CORE_H_
#include "slslam_oracle.h"

typedef struct {
  int n;        /* number of effective (free) parameters */
  void* ctx;
  /* Evaluate at reduced vector x. Always returns the (robustified) cost of the reduced program.
   * If want_jac: keeps residuals and the UNSCALED Jacobian inside ctx and writes gradient = J^T r. */
  int    (*evaluate)(void* ctx, const double* x, double* cost, int want_jac, double* gradient);
  void   (*sq_col_norm)(void* ctx, double* out);            /* of the Jacobian currently held   */
  void   (*scale_cols)(void* ctx, const double* scale);     /* J <- J diag(scale), in place      */
  int    (*solve)(void* ctx, const double* lm_diag, double* y); /* (J'J + diag(lm_diag)^2) y = J'r */
  double (*model_cost_change)(void* ctx, const double* step);/* -(J s)'(r + J s / 2)              */
} oracle_nlls;

int oracle_lm_minimize(oracle_nlls* P, const oracle_lm_options* opt, double* x,
                       oracle_summary* summary, oracle_iteration* trace, int trace_cap, int* trace_len);

/* dense SPD solve helpers (row-major n x n, lower triangle referenced); return 0 ok, 1 not SPD */
int oracle_dense_cholesky(double* a, int n);
void oracle_dense_cholesky_solve(const double* l, int n, double* b);
#endif
