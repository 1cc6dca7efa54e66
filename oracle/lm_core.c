/* oracle/lm_core.c — restatement of Ceres 1.7.0's TrustRegionMinimizer::Minimize with the
 * LevenbergMarquardtStrategy, as the reference configures it (TRUST_REGION + LEVENBERG_MARQUARDT
 * are the Ceres defaults; the reference only sets max_num_iterations, threads, eta and the linear
 * solver: reference src/lba_problem.cpp:95-132, src/po_problem.cpp:67-77).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (Ceres absent, see slslam_oracle.h).
 *
 * Restated policy (Ceres 1.7.0 internal/ceres/trust_region_minimizer.cc and
 * levenberg_marquardt_strategy.cc, from the published source):
 *   - initial evaluation; gradient tolerance is RELATIVE to max(|g0|_inf, 1e-12)
 *   - Jacobi scaling: scale_j = 1/(1+||J_j||) estimated ONCE at x0, re-applied after every
 *     Jacobian evaluation
 *   - LM diagonal D^2 = clamp(diag(J'J), min_lm_diagonal, max_lm_diagonal) / radius on the scaled
 *     Jacobian, recomputed only after an accepted step (reuse_diagonal_)
 *   - step = -y with (J'J + D^2) y = J'r;  model_cost_change = -(Js)'(r + Js/2); <0 => invalid
 *   - delta = step .* scale; candidate cost; parameter- then function-tolerance tests RETURN
 *     before the step is applied (1.7 behaviour)
 *   - rho = cost_change / model_cost_change; accept iff rho > min_relative_decrease;
 *     accept: radius /= max(1/3, 1-(2 rho-1)^3), capped; decrease_factor = 2
 *     reject: radius /= decrease_factor; decrease_factor *= 2;   invalid: radius *= 0.5
 *   - max_num_iterations counts successful + unsuccessful steps
 *   - final_cost = min over recorded iteration costs (SolverImpl::SetSummaryFinalCost)
 */
#include "lm_core.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

void oracle_lm_default_options(oracle_lm_options* o) {
  o->max_num_iterations = 10;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->jacobi_scaling = 1;
  o->linear_solver = 0;
  o->policy_variant = 0;
}

int oracle_dense_cholesky(double* a, int n) {
  for (int j = 0; j < n; ++j) {
    double* aj = a + (size_t)j * n;
    double d = aj[j];
    for (int k = 0; k < j; ++k) d -= aj[k] * aj[k];
    if (!(d > 0.0) || !isfinite(d)) return 1;
    d = sqrt(d);
    aj[j] = d;
    const double inv = 1.0 / d;
    for (int i = j + 1; i < n; ++i) {
      double* ai = a + (size_t)i * n;
      double s = ai[j];
      for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
      ai[j] = s * inv;
    }
  }
  return 0;
}

void oracle_dense_cholesky_solve(const double* l, int n, double* b) {
  for (int i = 0; i < n; ++i) {
    const double* li = l + (size_t)i * n;
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= li[k] * b[k];
    b[i] = s / li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= l[(size_t)k * n + i] * b[k];
    b[i] = s / l[(size_t)i * n + i];
  }
}

static double vec_norm(const double* a, int n) { double s = 0; for (int i = 0; i < n; ++i) s += a[i] * a[i]; return sqrt(s); }
static double vec_maxabs(const double* a, int n) { double m = 0; for (int i = 0; i < n; ++i) { double v = fabs(a[i]); if (v > m) m = v; } return m; }

static void push_trace(oracle_iteration* trace, int cap, int* len, const oracle_iteration* it) {
  if (trace && *len < cap) trace[*len] = *it;
  (*len)++;
}

int oracle_lm_minimize(oracle_nlls* P, const oracle_lm_options* opt, double* x_min,
                       oracle_summary* summary, oracle_iteration* trace, int trace_cap, int* trace_len) {
  const int n = P->n;
  int tl = 0;
  double* x = (double*)malloc(sizeof(double) * (size_t)(6 * n + 1));
  double* step = x + n, *delta = x + 2 * n, *x_plus = x + 3 * n, *gradient = x + 4 * n, *scale = x + 5 * n;
  double* diagonal = (double*)malloc(sizeof(double) * (size_t)(2 * n + 1));
  double* lm_diag = diagonal + n;
  memcpy(x, x_min, sizeof(double) * (size_t)n);
  double x_norm = vec_norm(x, n);

  summary->num_successful_steps = 0;
  summary->num_unsuccessful_steps = 0;
  summary->termination_type = ORACLE_NO_CONVERGENCE;

  /* LevenbergMarquardtStrategy state */
  double radius = opt->initial_trust_region_radius;
  double decrease_factor = 2.0;
  int reuse_diagonal = 0;

  double cost = 0.0;
  double minimum_cost = 0.0, min_recorded_cost = 0.0;
  int have_cost = 0;
  int rc = 0;
  oracle_iteration it;
  int last_iteration = 0;
  double last_gradient_max_norm = 0.0;
  double initial_gradient_max_norm = 0.0, absolute_gradient_tolerance = 0.0;
  int num_consecutive_invalid_steps = 0;
  int pending_termination = -1;            /* policy_variant bit 1 */
  if (!P->evaluate(P->ctx, x, &cost, 1, gradient)) {
    summary->termination_type = ORACLE_NUMERICAL_FAILURE; rc = 1; goto done;
  }
  minimum_cost = cost;
  summary->initial_cost = cost + summary->fixed_cost;
  min_recorded_cost = summary->initial_cost;
  have_cost = 1;

  memset(&it, 0, sizeof(it));
  it.iteration = 0;
  it.cost = cost + summary->fixed_cost;
  it.gradient_max_norm = vec_maxabs(gradient, n);
  it.trust_region_radius = radius;
  initial_gradient_max_norm = it.gradient_max_norm > 1e-12 ? it.gradient_max_norm : 1e-12;
  absolute_gradient_tolerance = (opt->policy_variant & 1) ? opt->gradient_tolerance : opt->gradient_tolerance * initial_gradient_max_norm;
  if (it.gradient_max_norm <= absolute_gradient_tolerance) {
    summary->termination_type = ORACLE_GRADIENT_TOLERANCE;
    goto done;
  }
  push_trace(trace, trace_cap, &tl, &it);
  last_iteration = 0;
  last_gradient_max_norm = it.gradient_max_norm;

  if (opt->jacobi_scaling) {
    P->sq_col_norm(P->ctx, scale);
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + sqrt(scale[i]));
    P->scale_cols(P->ctx, scale);
  } else {
    for (int i = 0; i < n; ++i) scale[i] = 1.0;
  }

  for (;;) {
    if (last_iteration >= opt->max_num_iterations) {
      summary->termination_type = ORACLE_NO_CONVERGENCE;
      break;
    }
    /* ---- LevenbergMarquardtStrategy::ComputeStep ---- */
    if (!reuse_diagonal) {
      P->sq_col_norm(P->ctx, diagonal);
      for (int i = 0; i < n; ++i) {
        double d = diagonal[i];
        if (d < opt->min_lm_diagonal) d = opt->min_lm_diagonal;
        if (d > opt->max_lm_diagonal) d = opt->max_lm_diagonal;
        diagonal[i] = d;
      }
    }
    for (int i = 0; i < n; ++i) lm_diag[i] = sqrt(diagonal[i] / radius);
    int solve_failed = P->solve(P->ctx, lm_diag, step);
    if (!solve_failed) for (int i = 0; i < n; ++i) if (!isfinite(step[i])) { solve_failed = 1; break; }
    if (!solve_failed) for (int i = 0; i < n; ++i) step[i] = -step[i];
    reuse_diagonal = 1;

    memset(&it, 0, sizeof(it));
    it.iteration = last_iteration + 1;

    double model_cost_change = 0.0;
    if (!solve_failed) {
      model_cost_change = P->model_cost_change(P->ctx, step);
      if (!(model_cost_change < 0.0)) it.step_is_valid = 1;
    }
    it.model_cost_change = model_cost_change;

    if (!it.step_is_valid) {
      if (++num_consecutive_invalid_steps >= opt->max_num_consecutive_invalid_steps) {
        summary->termination_type = ORACLE_NUMERICAL_FAILURE;
        rc = 1; goto done;
      }
      it.cost = cost + summary->fixed_cost;
      it.cost_change = 0.0;
      it.gradient_max_norm = last_gradient_max_norm;
      it.step_norm = 0.0;
      it.relative_decrease = 0.0;
    } else {
      num_consecutive_invalid_steps = 0;
      for (int i = 0; i < n; ++i) { delta[i] = step[i] * scale[i]; x_plus[i] = x[i] + delta[i]; }
      double new_cost = DBL_MAX;
      if (!P->evaluate(P->ctx, x_plus, &new_cost, 0, NULL) || !isfinite(new_cost)) new_cost = DBL_MAX;
      double sn = 0.0;
      for (int i = 0; i < n; ++i) { const double d = x[i] - x_plus[i]; sn += d * d; }
      it.step_norm = sqrt(sn);
      const double step_size_tolerance = opt->parameter_tolerance * (x_norm + opt->parameter_tolerance);
      const int late_tests = (opt->policy_variant & 2) != 0;      /* sweep variant: decide after the step has been applied */
      if (it.step_norm <= step_size_tolerance) {
        if (!late_tests) { summary->termination_type = ORACLE_PARAMETER_TOLERANCE; goto done; }
        pending_termination = ORACLE_PARAMETER_TOLERANCE;
      }
      it.cost_change = cost - new_cost;
      const double absolute_function_tolerance = opt->function_tolerance * cost;
      if (fabs(it.cost_change) < absolute_function_tolerance) {
        if (!late_tests) { summary->termination_type = ORACLE_FUNCTION_TOLERANCE; goto done; }
        if (pending_termination < 0) pending_termination = ORACLE_FUNCTION_TOLERANCE;
      }
      it.relative_decrease = it.cost_change / model_cost_change;
      it.step_is_successful = it.relative_decrease > opt->min_relative_decrease;
    }

    if (it.step_is_successful) {
      ++summary->num_successful_steps;
      /* StepAccepted */
      {
        const double q = 2.0 * it.relative_decrease - 1.0;
        double f = 1.0 - q * q * q;
        if (f < 1.0 / 3.0) f = 1.0 / 3.0;
        radius = radius / f;
        if (radius > opt->max_trust_region_radius) radius = opt->max_trust_region_radius;
        decrease_factor = 2.0;
        reuse_diagonal = 0;
      }
      memcpy(x, x_plus, sizeof(double) * (size_t)n);
      x_norm = vec_norm(x, n);
      if (!P->evaluate(P->ctx, x, &cost, 1, gradient)) {
        summary->termination_type = ORACLE_NUMERICAL_FAILURE; rc = 1; goto done;
      }
      it.gradient_max_norm = vec_maxabs(gradient, n);
      if (cost < minimum_cost) {  /* monotonic steps: always true for an accepted step */
        memcpy(x_min, x, sizeof(double) * (size_t)n);
        minimum_cost = cost;
      }
      if (it.gradient_max_norm <= absolute_gradient_tolerance) {
        summary->termination_type = ORACLE_GRADIENT_TOLERANCE;
        if (cost + summary->fixed_cost < min_recorded_cost) min_recorded_cost = cost + summary->fixed_cost;
        goto done;
      }
      if (opt->jacobi_scaling) P->scale_cols(P->ctx, scale);
    } else {
      ++summary->num_unsuccessful_steps;
      if (it.step_is_valid) { radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1; }
      else { radius *= 0.5; reuse_diagonal = 1; }
      it.gradient_max_norm = last_gradient_max_norm;
    }
    it.cost = cost + summary->fixed_cost;
    it.trust_region_radius = radius;
    if (it.cost < min_recorded_cost) min_recorded_cost = it.cost;
    if (radius < opt->min_trust_region_radius) {
      summary->termination_type = ORACLE_MIN_RADIUS;
      push_trace(trace, trace_cap, &tl, &it);
      goto done;
    }
    push_trace(trace, trace_cap, &tl, &it);
    last_iteration = it.iteration;
    last_gradient_max_norm = it.gradient_max_norm;
    if (pending_termination >= 0) {
      if (it.step_is_successful) { summary->termination_type = pending_termination; goto done; }
      pending_termination = -1;
    }
  }

done:
  /* SolverImpl::SetSummaryFinalCost: min over recorded iteration costs */
  if (have_cost) summary->final_cost = min_recorded_cost < summary->initial_cost ? min_recorded_cost : summary->initial_cost;
  else { summary->initial_cost = summary->fixed_cost; summary->final_cost = summary->fixed_cost; }
  if (trace_len) *trace_len = tl;
  free(x);
  free(diagonal);
  return rc;
}
