// slslam_amd/csrc/po_api.hip — C ABI of the pose-graph path: slslam_po_solve replaces
// POProblem::build + POProblem::set_options + ceres::Solve (reference src/slam.cpp:1283-1293).
// No CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/slslam_hip.h"
#include "po_kernels.h"

using namespace slslam;

#define PO_TRY(expr)                                                                    \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      std::fprintf(stderr, "slslam: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      rc = (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? SLSLAM_ERR_NO_DEVICE : SLSLAM_ERR_HIP; \
      goto done;                                                                        \
    }                                                                                   \
  } while (0)

extern "C" int slslam_po_solve(const slslam_po_graph* g, const slslam_solver_options* opt_in,
                               slslam_summary* summary, slslam_iteration* trace, int trace_cap, int* trace_len) {
  if (!g) return SLSLAM_ERR_INVALID_ARGUMENT;
  const int N = g->num_poses, E = g->num_edges;
  if (N < 0 || E < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (E > 0 && (!g->pose_index_1 || !g->pose_index_2 || !g->constraints)) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (N > 0 && !g->parameters) return SLSLAM_ERR_INVALID_ARGUMENT;
  slslam_solver_options opt;
  if (opt_in) opt = *opt_in; else slslam_default_options(&opt);
  if (opt.max_num_iterations < 0 || opt.max_num_iterations > 100000) return SLSLAM_ERR_INVALID_ARGUMENT;
  for (int e = 0; e < E; ++e) {
    const int a = g->pose_index_1[e], b = g->pose_index_2[e];
    if (a < 0 || a >= N || b < 0 || b >= N || a == b) return SLSLAM_ERR_INVALID_ARGUMENT;
    for (int q = 0; q < 6; ++q) if (!std::isfinite(g->constraints[6 * (size_t)e + q])) return SLSLAM_ERR_INVALID_ARGUMENT;
  }
  for (size_t i = 0; i < (size_t)6 * N; ++i) if (!std::isfinite(g->parameters[i])) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (trace_len) *trace_len = 0;
  if (summary) std::memset(summary, 0, sizeof(*summary));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  if (E == 0) { if (summary) summary->termination_type = SLSLAM_FUNCTION_TOLERANCE; return SLSLAM_OK; }

  // program reduction: pose1 of edge 0 is constant (po_problem.cpp:62-63); unreferenced poses are not in the problem
  std::vector<int> slot(N, -1), used(N, 0);
  for (int e = 0; e < E; ++e) { used[g->pose_index_1[e]] = 1; used[g->pose_index_2[e]] = 1; }
  const int gauge = g->pose_index_1[0];
  int n = 0, kept = 0;
  for (int k = 0; k < N; ++k) if (used[k] && k != gauge) { slot[k] = n; n += 6; }
  for (int e = 0; e < E; ++e) if (slot[g->pose_index_1[e]] >= 0 || slot[g->pose_index_2[e]] >= 0) ++kept;
  const int ld = ((n + 7) / 8) * 8 + 8;

  Policy pol;
  pol.huber_delta = 0.0; pol.baseline = 0.0;
  pol.initial_radius = opt.initial_trust_region_radius; pol.max_radius = opt.max_trust_region_radius;
  pol.min_radius = opt.min_trust_region_radius; pol.min_relative_decrease = opt.min_relative_decrease;
  pol.min_lm_diagonal = opt.min_lm_diagonal; pol.max_lm_diagonal = opt.max_lm_diagonal;
  pol.function_tolerance = opt.function_tolerance; pol.gradient_tolerance = opt.gradient_tolerance;
  pol.parameter_tolerance = opt.parameter_tolerance; pol.max_num_iterations = opt.max_num_iterations;
  pol.max_invalid = opt.max_num_consecutive_invalid_steps; pol.jacobi_scaling = opt.jacobi_scaling; pol.pad = 0;

  int rc = SLSLAM_OK;
  PoPtrs p;
  std::memset(&p, 0, sizeof(p));
  int *d_p1 = nullptr, *d_p2 = nullptr, *d_slot = nullptr;
  double *d_cons = nullptr, *d_linv = nullptr;
  float *d_Hf = nullptr, *d_linvf = nullptr;
  const bool f32 = opt.po_factor_fp32 != 0;
  LMState hst;
  std::vector<IterRec> htrace(kMaxTrace);
  std::vector<double> x2((size_t)12 * N), ones((size_t)(n > 0 ? n : 1), 1.0);
  const size_t hbytes = (size_t)(n > 0 ? n : 1) * ld * sizeof(double);
  const int nblk = (n + kNB - 1) / kNB;
  const dim3 g_edges((unsigned)((E + 4) / 5));

  PO_TRY(hipMalloc((void**)&d_p1, sizeof(int) * E));
  PO_TRY(hipMalloc((void**)&d_p2, sizeof(int) * E));
  PO_TRY(hipMalloc((void**)&d_slot, sizeof(int) * (N > 0 ? N : 1)));
  PO_TRY(hipMalloc((void**)&d_cons, sizeof(double) * 6 * E));
  PO_TRY(hipMalloc((void**)&p.x, sizeof(double) * 12 * (N > 0 ? N : 1)));
  PO_TRY(hipMalloc((void**)&p.scale, sizeof(double) * ones.size()));
  PO_TRY(hipMalloc((void**)&p.H, hbytes));
  PO_TRY(hipMalloc((void**)&p.g, sizeof(double) * ones.size()));
  PO_TRY(hipMalloc((void**)&p.d2, sizeof(double) * ones.size()));
  PO_TRY(hipMalloc((void**)&p.y, sizeof(double) * ones.size()));
  PO_TRY(hipMalloc((void**)&d_linv, sizeof(double) * kNB * kNB * (size_t)(nblk > 0 ? nblk : 1)));
  if (f32) {
    PO_TRY(hipMalloc((void**)&d_Hf, sizeof(float) * (size_t)(n > 0 ? n : 1) * ld));
    PO_TRY(hipMalloc((void**)&d_linvf, sizeof(float) * kNB * kNB * (size_t)(nblk > 0 ? nblk : 1)));
  }
  PO_TRY(hipMalloc((void**)&p.scal, sizeof(double) * 8));
  PO_TRY(hipMalloc((void**)&p.flags, sizeof(int) * 2));
  PO_TRY(hipMalloc((void**)&p.st, sizeof(LMState)));
  PO_TRY(hipMalloc((void**)&p.trace, sizeof(IterRec) * kMaxTrace));
  PO_TRY(hipMemcpy(d_p1, g->pose_index_1, sizeof(int) * E, hipMemcpyHostToDevice));
  PO_TRY(hipMemcpy(d_p2, g->pose_index_2, sizeof(int) * E, hipMemcpyHostToDevice));
  PO_TRY(hipMemcpy(d_slot, slot.data(), sizeof(int) * N, hipMemcpyHostToDevice));
  PO_TRY(hipMemcpy(d_cons, g->constraints, sizeof(double) * 6 * E, hipMemcpyHostToDevice));
  std::memcpy(x2.data(), g->parameters, sizeof(double) * 6 * N);
  std::memcpy(x2.data() + (size_t)6 * N, g->parameters, sizeof(double) * 6 * N);
  PO_TRY(hipMemcpy(p.x, x2.data(), sizeof(double) * 12 * N, hipMemcpyHostToDevice));
  PO_TRY(hipMemcpy(p.scale, ones.data(), sizeof(double) * ones.size(), hipMemcpyHostToDevice));
  std::memset(&hst, 0, sizeof(hst));
  hst.radius = pol.initial_radius; hst.decrease_factor = 2.0; hst.status = kRunning;
  PO_TRY(hipMemcpy(p.st, &hst, sizeof(hst), hipMemcpyHostToDevice));
  PO_TRY(hipMemset(p.trace, 0, sizeof(IterRec) * kMaxTrace));
  PO_TRY(hipMemset(p.scal, 0, sizeof(double) * 8));
  PO_TRY(hipMemset(p.flags, 0, sizeof(int) * 2));
  p.p1 = d_p1; p.p2 = d_p2; p.cons = d_cons; p.slot = d_slot;
  p.N = N; p.E = E; p.n = n; p.ld = ld;

  // ---- initial evaluation: cost, gradient, column norms -> Jacobi scale
  PO_TRY(hipMemsetAsync(p.H, 0, hbytes, 0));
  PO_TRY(hipMemsetAsync(p.g, 0, sizeof(double) * ones.size(), 0));
  hipLaunchKernelGGL(k_po_linearise, g_edges, dim3(64), 0, 0, p, 0);
  hipLaunchKernelGGL(k_po_prepare, dim3(1), dim3(256), 0, 0, p, pol, 1);
  // ---- LM iterations, enqueued without host synchronisation; finished solves early-out on device
  for (int it = 0; it < pol.max_num_iterations && n > 0; ++it) {
    if (it > 0 && (it % 8) == 0) {      // long solves: stop enqueueing once the device reports termination
      PO_TRY(hipMemcpyAsync(&hst, p.st, sizeof(hst), hipMemcpyDeviceToHost, 0));
      PO_TRY(hipStreamSynchronize(0));
      if (hst.status != kRunning) break;
    }
    PO_TRY(hipMemsetAsync(p.H, 0, hbytes, 0));
    PO_TRY(hipMemsetAsync(p.g, 0, sizeof(double) * ones.size(), 0));
    PO_TRY(hipMemsetAsync(p.scal, 0, sizeof(double), 0));            // kPoCost
    hipLaunchKernelGGL(k_po_linearise, g_edges, dim3(64), 0, 0, p, 0);
    hipLaunchKernelGGL(k_po_prepare, dim3(1), dim3(256), 0, 0, p, pol, 0);
    if (f32) hipLaunchKernelGGL(k_po_to_f32, dim3(256), dim3(256), 0, 0, p, d_Hf);
    for (int bk = 0; bk < nblk; ++bk) {
      const int k0 = bk * kNB;
      const int rem = n - (k0 + kNB);
      const int tb = rem > 0 ? (rem + kNB - 1) / kNB : 0;
      if (f32) {
        hipLaunchKernelGGL(k_po_potrf_diag<float>, dim3(1), dim3(256), 0, 0, p, d_Hf, d_linvf + (size_t)bk * kNB * kNB, k0);
        if (tb > 0) {
          hipLaunchKernelGGL(k_po_panel_update<float>, dim3((unsigned)tb), dim3(256), 0, 0, p, d_Hf, (const float*)(d_linvf + (size_t)bk * kNB * kNB), k0, 0);
          hipLaunchKernelGGL(k_po_panel_update<float>, dim3((unsigned)(tb * (tb + 1) / 2)), dim3(256), 0, 0, p, d_Hf, (const float*)(d_linvf + (size_t)bk * kNB * kNB), k0, 1);
        }
      } else {
        hipLaunchKernelGGL(k_po_potrf_diag<double>, dim3(1), dim3(256), 0, 0, p, p.H, d_linv + (size_t)bk * kNB * kNB, k0);
        if (tb > 0) {
          hipLaunchKernelGGL(k_po_panel_update<double>, dim3((unsigned)tb), dim3(256), 0, 0, p, p.H, (const double*)(d_linv + (size_t)bk * kNB * kNB), k0, 0);
          hipLaunchKernelGGL(k_po_panel_update<double>, dim3((unsigned)(tb * (tb + 1) / 2)), dim3(256), 0, 0, p, p.H, (const double*)(d_linv + (size_t)bk * kNB * kNB), k0, 1);
        }
      }
    }
    if (f32) hipLaunchKernelGGL(k_po_trisolve<float>, dim3(1), dim3(256), 0, 0, p, (const float*)d_Hf, (const float*)d_linvf);
    else hipLaunchKernelGGL(k_po_trisolve<double>, dim3(1), dim3(256), 0, 0, p, (const double*)p.H, (const double*)d_linv);
    hipLaunchKernelGGL(k_po_candidate, dim3(1), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_po_linearise, g_edges, dim3(64), 0, 0, p, 1);
    hipLaunchKernelGGL(k_po_update, dim3(1), dim3(64), 0, 0, p, pol);
  }
  PO_TRY(hipGetLastError());
  PO_TRY(hipDeviceSynchronize());
  PO_TRY(hipMemcpy(&hst, p.st, sizeof(hst), hipMemcpyDeviceToHost));
  PO_TRY(hipMemcpy(htrace.data(), p.trace, sizeof(IterRec) * kMaxTrace, hipMemcpyDeviceToHost));
  PO_TRY(hipMemcpy(x2.data(), p.x, sizeof(double) * 12 * N, hipMemcpyDeviceToHost));
  {
    int term = hst.status == kRunning ? SLSLAM_NO_CONVERGENCE : hst.status;
    if (n == 0) term = SLSLAM_FUNCTION_TOLERANCE;     // no non-constant parameter blocks
    if (term != SLSLAM_NUMERICAL_FAILURE)
      std::memcpy(g->parameters, x2.data() + (size_t)hst.cur * 6 * N, sizeof(double) * 6 * N);
    if (summary) {
      summary->num_successful_steps = hst.n_success; summary->num_unsuccessful_steps = hst.n_unsuccess;
      summary->initial_cost = hst.initial_cost;
      summary->final_cost = hst.min_cost < hst.initial_cost ? hst.min_cost : hst.initial_cost;
      summary->fixed_cost = hst.fixed_cost; summary->termination_type = term;
      summary->num_free_parameters = n; summary->num_residual_blocks = kept;
    }
    const int nt = hst.ntrace < kMaxTrace ? hst.ntrace : kMaxTrace;
    if (trace_len) *trace_len = nt;
    for (int i = 0; trace && i < nt && i < trace_cap; ++i) {
      const IterRec& r = htrace[i];
      slslam_iteration& o = trace[i];
      o.iteration = r.iteration; o.step_is_valid = r.step_is_valid; o.step_is_successful = r.step_is_successful;
      o.cost = r.cost; o.cost_change = r.cost_change; o.gradient_max_norm = r.gradient_max_norm;
      o.step_norm = r.step_norm; o.relative_decrease = r.relative_decrease;
      o.trust_region_radius = r.trust_region_radius; o.model_cost_change = r.model_cost_change;
    }
  }
done:
  (void)hipFree(d_p1); (void)hipFree(d_p2); (void)hipFree(d_slot); (void)hipFree(d_cons);
  (void)hipFree(p.x); (void)hipFree(p.scale); (void)hipFree(p.H); (void)hipFree(p.g); (void)hipFree(p.d2);
  (void)hipFree(p.y); (void)hipFree(d_linv); (void)hipFree(d_Hf); (void)hipFree(d_linvf); (void)hipFree(p.scal); (void)hipFree(p.flags);
  (void)hipFree(p.st); (void)hipFree(p.trace);
  return rc;
}
