// slslam_amd/csrc/po_api.hip — C ABI of the pose-graph path: slslam_po_solve replaces
// POProblem::build + POProblem::set_options + ceres::Solve (reference src/slam.cpp:1283-1293).
// No CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <type_traits>
#include <mutex>
#include <vector>

#include "../../include/slslam_hip.h"
#include "po_kernels.h"
#include "device_cache.h"

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


namespace {

enum { kMaxChain = 32 };

// Symbolic analysis of the structured factorisation: junction = free pose with >= 3 distinct free
// neighbours (plus one pose per junction-free cycle); every other free pose lies on a chain.
// Output: slot[] (offset of each pose in the reduced vector: chains first, pose by pose along the
// path; junctions last), the chain descriptors, n_chain = unknowns before the junction block.
void order_chains_first(int N, int E, const int* p1, const int* p2, const std::vector<int>& used, int gauge,
                        std::vector<int>& slot, std::vector<PoChain>& chains, int* n_chain, int* n_total, std::vector<int>* level_counts) {
  std::vector<std::vector<int>> adj(N);
  auto add = [&](int a, int b) { for (int v : adj[a]) if (v == b) return; adj[a].push_back(b); };
  for (int e = 0; e < E; ++e) {
    const int a = p1[e], b = p2[e];
    if (a == gauge || b == gauge) continue;           // the constant pose only contributes to diagonals
    add(a, b); add(b, a);
  }
  std::vector<char> isfree(N, 0), junction(N, 0), done(N, 0);
  for (int k = 0; k < N; ++k) isfree[k] = used[k] && k != gauge;
  for (int k = 0; k < N; ++k) if (isfree[k] && adj[k].size() >= 3) junction[k] = 1;
  std::vector<std::vector<int>> paths;
  auto walk = [&](int start) {                        // start: a chain pose with at most one chain neighbour
    std::vector<int> path;
    int prev = -1, cur = start;
    while (cur >= 0) {
      path.push_back(cur); done[cur] = 1;
      int nxt = -1;
      for (int v : adj[cur]) if (v != prev && !junction[v] && !done[v]) { nxt = v; break; }
      prev = cur; cur = nxt;
    }
    paths.push_back(path);
  };
  for (int pass = 0; pass < 2; ++pass) {
    for (int k = 0; k < N; ++k) {
      if (!isfree[k] || junction[k] || done[k]) continue;
      int chain_nb = 0;
      for (int v : adj[k]) if (!junction[v] && !done[v]) ++chain_nb;
      if (chain_nb <= 1) walk(k);
    }
    if (pass == 0)                                    // what is left are junction-free cycles: open each at one pose
      for (int k = 0; k < N; ++k)
        if (isfree[k] && !junction[k] && !done[k]) {
          bool alone = true;                          // still untouched after this loop's earlier promotions?
          for (int v : adj[k]) if (junction[v]) alone = false;
          if (alone) {                                // one pose per cycle: walk the rest of the cycle right away, so that no
            junction[k] = 1;                          // other pose of it is promoted (its poses are chain poses from here on)
            for (int v : adj[k]) if (isfree[v] && !junction[v] && !done[v]) { walk(v); break; }
          }
        }
  }
  // a chain whose two ends meet the same junction: move its last pose to the junctions
  for (auto& path : paths) {
    if (path.size() < 2) continue;
    int jl = -1, jr = -1;
    for (int v : adj[path.front()]) if (junction[v]) jl = v;
    for (int v : adj[path.back()]) if (junction[v]) jr = v;
    if (jl >= 0 && jl == jr) { junction[path.back()] = 1; path.pop_back(); }
  }
  // A chain is sequential (one 6 x 6 step after the other, ~2 us each), so long paths are cut - on several LEVELS since round 5: a path of L poses
  // becomes pieces of s poses separated by single CUT poses (level-1 chains: their ends are cut poses or the path's junctions), and the cut
  // poses of a path, in path order, are a chain of their own on the next level: eliminating the pieces leaves them coupled to each other and to
  // the path's two junctions exactly as the poses of a chain are (the pieces' Schur complements land on the blocks (c_i, c_i), (c_i+1, c_i),
  // (junction, c_1), (junction, c_m) - what k_po_chain_eliminate reads for a chain c_1 .. c_m) - so that chain is cut the same way, and so on.
  // The same two kernels run once per level, upwards for the elimination, downwards for the substitution; s ~ L^(1/levels) with up to three
  // levels, i.e. a sequential depth of ~3 L^(1/3) steps for a long path instead of min(L, 32), and the cut poses no longer enlarge the dense
  // junction block (until round 5 a path was cut every 32 poses and the cut poses joined the junctions: 26 junction poses instead of 10
  // for the 260-pose bench graph - a second 64-wide block step of the dense factorisation in every iteration).
  struct Piece { std::vector<int> poses; int left, right; };      // left / right: the POSE the piece ends at (-1: a free end)
  std::vector<std::vector<Piece>> levels;
  {
    static const char* env_sub = std::getenv("SLSLAM_PO_SUBCHAIN");        // (experiments: piece length of the first level)
    auto piece_len = [&](size_t L, int level) -> size_t {
      if (level == 0 && env_sub) return (size_t)std::min((int)kMaxChain, std::max(1, std::atoi(env_sub)));
      static const char* env_lv = std::getenv("SLSLAM_PO_LEVELS");       // (experiments: levels a long path is spread over)
      int m = L <= 8 ? 1 : (L <= 72 ? 2 : 3);                          // levels this chain is spread over
      if (env_lv && L > 8) m = std::max(1, std::atoi(env_lv) - level);
      if (m <= 1) return L;
      return (size_t)std::min((int)kMaxChain, std::max(2, (int)std::lround(std::pow((double)L, 1.0 / m))));
    };
    struct Job { std::vector<int> seq; int left, right, level; };
    std::vector<Job> jobs;
    for (auto& path : paths) {
      int jl_pose = -1, jr_pose = -1;
      for (int v : adj[path.front()]) if (junction[v]) jl_pose = v;
      for (int v : adj[path.back()]) if (junction[v] && (path.size() > 1 || v != jl_pose)) jr_pose = v;
      jobs.push_back(Job{ path, jl_pose, jr_pose, 0 });
    }
    for (size_t q = 0; q < jobs.size(); ++q) {                         // (jobs grows: a cut chain is the next level's job)
      const Job job = jobs[q];
      if (levels.size() <= (size_t)job.level) levels.resize((size_t)job.level + 1);
      const size_t sub = std::max<size_t>(1, piece_len(job.seq.size(), job.level));
      std::vector<int> cuts;
      size_t b = 0;
      int left = job.left;
      while (job.seq.size() - b > sub + 1 && job.level < 7) {            // (at least one pose is left behind the cut pose; eight levels at most)
        levels[(size_t)job.level].push_back(Piece{ std::vector<int>(job.seq.begin() + b, job.seq.begin() + b + sub), left, job.seq[b + sub] });
        left = job.seq[b + sub];
        cuts.push_back(left);
        b += sub + 1;
      }
      // (the rest of the chain; longer than kMaxChain only when the level cap above was reached - then it is simply a long chain)
      levels[(size_t)job.level].push_back(Piece{ std::vector<int>(job.seq.begin() + b, job.seq.end()), left, job.right });
      if (!cuts.empty()) jobs.push_back(Job{ cuts, job.left, job.right, job.level + 1 });
    }
  }
  int n = 0;
  for (const auto& lv : levels)
    for (const Piece& pc : lv) {
      PoChain c;
      c.start = n; c.len = (int)pc.poses.size(); c.jl = -1; c.jr = -1;
      for (int v : pc.poses) { slot[v] = n; n += 6; }
      chains.push_back(c);
    }
  if (level_counts) { level_counts->clear(); for (const auto& lv : levels) level_counts->push_back((int)lv.size()); }
  *n_chain = n;
  for (int k = 0; k < N; ++k) if (isfree[k] && junction[k]) { slot[k] = n; n += 6; }
  *n_total = n;
  {
    size_t q = 0;
    for (const auto& lv : levels)
      for (const Piece& pc : lv) {
        PoChain& c = chains[q++];
        c.jl = pc.left >= 0 ? slot[pc.left] : -1;
        c.jr = pc.right >= 0 ? slot[pc.right] : -1;
      }
  }
}

}  // namespace

namespace {
// Optional timing of slslam_po_solve (slslam_po_set_profiling): hipEvents on the solve's stream around the whole device
// part and around every factorisation + triangular solve; read back with slslam_po_last_timing.
struct PoTiming { bool enabled = false; double total_ms = 0, factor_ms = 0, factor_max_ms = 0; int factor_calls = 0, unknowns = 0, junction_unknowns = 0; };
thread_local PoTiming g_po_timing;
}  // namespace

extern "C" int slslam_po_set_profiling(int enable) { g_po_timing.enabled = enable != 0; return SLSLAM_OK; }
extern "C" int slslam_po_last_timing(double* total_ms, double* factor_ms, int* factor_calls, int* unknowns, int* junction_unknowns) {
  if (total_ms) *total_ms = g_po_timing.total_ms;
  if (factor_ms) *factor_ms = g_po_timing.factor_max_ms;
  if (factor_calls) *factor_calls = g_po_timing.factor_calls;
  if (unknowns) *unknowns = g_po_timing.unknowns;
  if (junction_unknowns) *junction_unknowns = g_po_timing.junction_unknowns;
  return SLSLAM_OK;
}

namespace {
// Blocked right-looking Cholesky of an nb-block matrix: the first diagonal block, then ONE launch per block step (k_po_step:
// panel solve + trailing update + the next diagonal block's factorisation; three launches per step until round 4).
template <typename T>
void po_factor_dense(PoPtrs& pp, T* A, T* Lf, T* linv, int nb) {
  if (nb <= 0) return;
  static const bool old_chain = std::getenv("SLSLAM_PO_LAUNCH_CHAIN") != nullptr;     // timing comparisons only: the round 1-3 chain
  if (old_chain) {
    for (int bk = 0; bk < nb; ++bk) {
      const int tb = nb - 1 - bk;
      hipLaunchKernelGGL(k_po_potrf_diag<T>, dim3(1), dim3(256), 0, 0, pp, A, linv + (size_t)bk * kNB * kNB, bk * kNB, (T*)nullptr);
      if (tb > 0) {
        hipLaunchKernelGGL(k_po_panel_update<T>, dim3((unsigned)tb), dim3(256), 0, 0, pp, A, (const T*)(linv + (size_t)bk * kNB * kNB), bk * kNB, 0);
        hipLaunchKernelGGL(k_po_panel_update<T>, dim3((unsigned)(tb * (tb + 1) / 2)), dim3(256), 0, 0, pp, A, (const T*)(linv + (size_t)bk * kNB * kNB), bk * kNB, 1);
      }
    }
    // (in place: the caller's substitution reads Lf)
    // (row by row: A and Lf may be the junction block of a larger matrix - pitch ld, width n - and n * ld elements would run past both)
    (void)hipMemcpy2DAsync(Lf, sizeof(T) * (size_t)pp.ld, A, sizeof(T) * (size_t)pp.ld, sizeof(T) * (size_t)pp.n, (size_t)pp.n, hipMemcpyDeviceToDevice, 0);
    return;
  }
  hipLaunchKernelGGL(k_po_potrf_diag<T>, dim3(1), dim3(256), 0, 0, pp, A, linv, 0, Lf);
  for (int bk = 0; bk + 1 < nb; ++bk) {
    const int tb = nb - 1 - bk;
    hipLaunchKernelGGL(k_po_step<T>, dim3((unsigned)(tb * (tb + 1) / 2)), dim3(256), kPoStepLdsTiles * kNB * kLdT * sizeof(T), 0, pp, A, Lf, linv, bk);
  }
}
// (k_po_step keeps three 64 x 66 tiles in dynamic LDS: 101 KB of doubles - above the 64 KB a kernel gets without asking)
// HIP function attributes are per DEVICE and slslam_po_solve runs on whatever device is current: the limit is raised once per device
// (a bit per device id under a mutex; a process that solves on device 0 and then on device 1 raises it on both).
hipError_t po_step_lds_attributes() {
  static std::mutex mu;
  static unsigned long long done_mask = 0;          // devices 0..63; beyond that the attribute is set on every call (it is cheap)
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < 64 && ((done_mask >> dev) & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute((const void*)k_po_step<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPoStepLdsTiles * kNB * kLdT * sizeof(double)));
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_po_step<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kPoStepLdsTiles * kNB * kLdT * sizeof(float)));
  if (e == hipSuccess && dev >= 0 && dev < 64) done_mask |= 1ull << dev;
  return e;
}
}  // namespace

namespace { thread_local int po_iter_hint = -1; }     // LM iterations the calling thread's previous pose-graph solve took

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
  const bool f32 = opt.po_factor_fp32 != 0;
  const bool structured = !f32 && !opt.po_dense_factor;
  std::vector<PoChain> chains;
  int n_chain = 0;                       // unknowns of the chain poses (ordered first: level-1 chains, then the chains of cut poses, level after level)
  std::vector<int> level_counts;         // chains per level (listed level after level)
  if (!structured) {
    for (int k = 0; k < N; ++k) if (used[k] && k != gauge) { slot[k] = n; n += 6; }
  } else {
    // The symbolic analysis depends on the TOPOLOGY alone, and the reference's graph only changes when a loop closure adds an edge
    // (src/slam.cpp:1248-1280): the calling thread keeps the analysis of its last graph and reuses it when the edge lists are the same.
    struct Symbolic { int N = -1, E = -1, n_chain = 0, n = 0; std::vector<int> p1, p2, slot, level_counts; std::vector<PoChain> chains; };
    static thread_local Symbolic* sym = nullptr;
    if (!sym) sym = new Symbolic();          // (never destroyed: no teardown order to get wrong at thread exit)
    const bool hit = sym->N == N && sym->E == E && std::memcmp(sym->p1.data(), g->pose_index_1, sizeof(int) * (size_t)E) == 0 &&
                     std::memcmp(sym->p2.data(), g->pose_index_2, sizeof(int) * (size_t)E) == 0 && !std::getenv("SLSLAM_PO_NO_SYMBOLIC_CACHE");
    if (!hit) {
      order_chains_first(N, E, g->pose_index_1, g->pose_index_2, used, gauge, slot, chains, &n_chain, &n, &level_counts);
      sym->N = N; sym->E = E; sym->n_chain = n_chain; sym->n = n;
      sym->p1.assign(g->pose_index_1, g->pose_index_1 + E); sym->p2.assign(g->pose_index_2, g->pose_index_2 + E);
      sym->slot = slot; sym->level_counts = level_counts; sym->chains = chains;
    } else { slot = sym->slot; chains = sym->chains; level_counts = sym->level_counts; n_chain = sym->n_chain; n = sym->n; }
  }
  for (int e = 0; e < E; ++e) if (slot[g->pose_index_1[e]] >= 0 || slot[g->pose_index_2[e]] >= 0) ++kept;
  const int ld = ((n + 7) / 8) * 8 + 8;

  Policy pol;
  pol.huber_delta = 0.0; pol.baseline = 0.0;
  pol.initial_radius = opt.initial_trust_region_radius; pol.max_radius = opt.max_trust_region_radius;
  pol.min_radius = opt.min_trust_region_radius; pol.min_relative_decrease = opt.min_relative_decrease;
  pol.min_lm_diagonal = opt.min_lm_diagonal; pol.max_lm_diagonal = opt.max_lm_diagonal;
  pol.function_tolerance = opt.function_tolerance; pol.gradient_tolerance = opt.gradient_tolerance;
  pol.parameter_tolerance = opt.parameter_tolerance; pol.max_num_iterations = opt.max_num_iterations;
  pol.max_invalid = opt.max_num_consecutive_invalid_steps; pol.jacobi_scaling = opt.jacobi_scaling; pol.keep_jacobian = 0;

  int rc = SLSLAM_OK;
  const bool timing = g_po_timing.enabled;
  std::vector<hipEvent_t> tev;            // [0] start, [1] end, then (start, stop) per factorisation
  auto stamp = [&]() { if (timing) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, 0); tev.push_back(e); } } };
  PoPtrs p, pj;
  std::memset(&p, 0, sizeof(p));
  PoChain* d_chains = nullptr;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  int arena_device = 0;
  const int nj = structured ? n - n_chain : 0;
  const int nblk_j = (nj + kNB - 1) / kNB;
  // what the level-1 eliminations ADD into - the blocks of the cut poses and of the junctions, among themselves - starts at zero: the square behind
  // the level-1 unknowns (the dense factorisation reads the junction block of it)
  const int n_level1 = level_counts.empty() ? 0 : level_counts[0];
  const int n_l1 = (structured && n_level1 > 0) ? chains[(size_t)n_level1 - 1].start + 6 * chains[(size_t)n_level1 - 1].len : 0;
  const int nz = structured ? n - n_l1 : 0;
  const long long zero_items = (long long)E * 144 + (long long)nz * nz + n + 1;       // k_po_zero_structured
  const dim3 g_zero((unsigned)((zero_items + 255) / 256));
  const bool zero_small = structured && n > 0 && !std::getenv("SLSLAM_PO_FULL_MEMSET");
  int *d_p1 = nullptr, *d_p2 = nullptr, *d_slot = nullptr;
  double *d_cons = nullptr, *d_linv = nullptr;
  float *d_Hf = nullptr, *d_linvf = nullptr, *d_Lff = nullptr;
  double* d_Lf = nullptr;             // the Cholesky factor (k_po_step keeps it apart from the matrix it updates)
  unsigned* d_tri_flags = nullptr;    // k_po_trisolve_wide: one progress word per 64-row block
  unsigned tri_epoch = 1;
  int num_cus = 0;
  int wide_resident = 0;             // workgroups of k_po_trisolve_wide the device keeps resident together (0: not known - the one-workgroup substitution runs)
  LMState hst;
  int next_check = 8;
  bool have_results = false;
  std::vector<IterRec> htrace(kMaxTrace);
  std::vector<double> x2((size_t)12 * N), ones((size_t)(n > 0 ? n : 1), 1.0);
  const size_t hbytes = (size_t)(n > 0 ? n : 1) * ld * sizeof(double);
  const int nblk = (n + kNB - 1) / kNB;
  const dim3 g_edges((unsigned)((E + 4) / 5));

  // one device allocation carved into the work arrays (17 hipMalloc / hipFree pairs cost more than a small solve); what the host fills comes
  // FIRST and contiguous - state | trace | poses (what comes back, in one copy) | indices, slots, constraints, scale, chains, the small zeroed
  // words - so that it goes up in ONE copy from a pinned image the calling thread keeps (a dozen synchronous hipMemcpy / hipMemset calls were a
  // quarter of a 260-pose solve's host clock)
  size_t up_bytes = 0, down_bytes = 0;
  char* stage = nullptr;
  {
    const size_t nn = ones.size(), nb2 = (size_t)kNB * kNB * (size_t)(nblk > 0 ? nblk : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t o_st = take(sizeof(LMState)), o_trace = take(sizeof(IterRec) * kMaxTrace), o_x = take(sizeof(double) * 12 * (N > 0 ? N : 1));
    down_bytes = off;
    const size_t o_p1 = take(sizeof(int) * E), o_p2 = take(sizeof(int) * E), o_slot = take(sizeof(int) * (N > 0 ? N : 1)),
                 o_cons = take(sizeof(double) * 6 * E), o_scale = take(sizeof(double) * nn), o_chains = take(sizeof(PoChain) * (chains.size() + 1)),
                 o_scal = take(sizeof(double) * 8), o_flags = take(sizeof(int) * 2), o_tri = take(sizeof(unsigned) * (size_t)(nblk + 1));
    up_bytes = off;
    const size_t o_H = take(hbytes), o_g = take(sizeof(double) * nn), o_d2 = take(sizeof(double) * nn), o_y = take(sizeof(double) * nn),
                 o_linv = take(sizeof(double) * nb2), o_Hf = take(f32 ? sizeof(float) * (size_t)(n > 0 ? n : 1) * ld : 0),
                 o_Lf = take(f32 ? 0 : hbytes), o_Lff = take(f32 ? sizeof(float) * (size_t)(n > 0 ? n : 1) * ld : 0),
                 o_linvf = take(f32 ? sizeof(float) * nb2 : 0);
    arena_bytes = off;
    (void)hipGetDevice(&arena_device);
    PO_TRY(DeviceBlockCache::acquire(off, arena_device, &arena));   // the block of the previous one-shot solve, if large enough
    d_p1 = (int*)(arena + o_p1); d_p2 = (int*)(arena + o_p2); d_slot = (int*)(arena + o_slot); d_cons = (double*)(arena + o_cons);
    p.x = (double*)(arena + o_x); p.scale = (double*)(arena + o_scale); p.H = (double*)(arena + o_H); p.g = (double*)(arena + o_g);
    p.d2 = (double*)(arena + o_d2); p.y = (double*)(arena + o_y); d_linv = (double*)(arena + o_linv);
    if (f32) { d_Hf = (float*)(arena + o_Hf); d_linvf = (float*)(arena + o_linvf); d_Lff = (float*)(arena + o_Lff); }
    else d_Lf = (double*)(arena + o_Lf);
    d_tri_flags = (unsigned*)(arena + o_tri);
    p.scal = (double*)(arena + o_scal); p.flags = (int*)(arena + o_flags); p.st = (LMState*)(arena + o_st);
    p.trace = (IterRec*)(arena + o_trace); d_chains = (PoChain*)(arena + o_chains);
    // the pinned image (kept per calling thread, grown on demand)
    struct HostStage { char* p = nullptr; size_t bytes = 0; };
    static thread_local HostStage* hs = nullptr;
    if (!hs) hs = new HostStage();
    if (hs->bytes < up_bytes) {
      if (hs->p) (void)hipHostFree(hs->p);
      hs->p = nullptr; hs->bytes = 0;
      PO_TRY(hipHostMalloc((void**)&hs->p, up_bytes + up_bytes / 4 + 4096, hipHostMallocDefault));
      hs->bytes = up_bytes + up_bytes / 4 + 4096;
    }
    stage = hs->p;
    std::memset(stage, 0, up_bytes);
    std::memset(&hst, 0, sizeof(hst));
    hst.radius = pol.initial_radius; hst.decrease_factor = 2.0; hst.status = kRunning;
    std::memcpy(stage + o_st, &hst, sizeof(hst));
    std::memcpy(stage + o_x, g->parameters, sizeof(double) * 6 * N);
    std::memcpy(stage + o_x + sizeof(double) * 6 * N, g->parameters, sizeof(double) * 6 * N);
    std::memcpy(stage + o_p1, g->pose_index_1, sizeof(int) * E); std::memcpy(stage + o_p2, g->pose_index_2, sizeof(int) * E);
    std::memcpy(stage + o_slot, slot.data(), sizeof(int) * N); std::memcpy(stage + o_cons, g->constraints, sizeof(double) * 6 * E);
    std::memcpy(stage + o_scale, ones.data(), sizeof(double) * nn);
    if (!chains.empty()) std::memcpy(stage + o_chains, chains.data(), sizeof(PoChain) * chains.size());
    PO_TRY(hipMemcpyAsync(arena, stage, up_bytes, hipMemcpyHostToDevice, 0));
  }
  (void)hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, arena_device);
  // k_po_trisolve_wide spin-waits across workgroups: all of its nblk workgroups have to be resident together.  Ask the runtime how
  // many fit (a CU mask or a compute partition shows up here); when it cannot tell, the one-workgroup substitution runs instead.
  {
    int per_cu = 0;
    const hipError_t eo = f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_po_trisolve_wide<float>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_po_trisolve_wide<double>, 256, 0);
    if (eo == hipSuccess && per_cu > 0 && num_cus > 0) wide_resident = per_cu >= 1 ? num_cus : 0;   // (one workgroup per CU is all that is counted on)
    else (void)hipGetLastError();
  }
  p.p1 = d_p1; p.p2 = d_p2; p.cons = d_cons; p.slot = d_slot;
  p.N = N; p.E = E; p.n = n; p.ld = ld;
  pj = p;                                  // the junction block as a matrix of its own (same leading dimension)
  pj.n = nj; pj.H = p.H + (size_t)n_chain * ld + n_chain; pj.y = p.y + n_chain;

  if (po_step_lds_attributes() != hipSuccess) { PO_TRY(hipErrorInvalidValue); }
  stamp(); stamp();                       // [1] is re-recorded at the end
  // ---- initial evaluation: cost, gradient, column norms -> Jacobi scale
  // (structured: only what the linearisation adds into and the junction block are zeroed, by one small launch: k_po_zero_structured)
  if (zero_small) {
    PO_TRY(hipMemsetAsync(p.g, 0, sizeof(double) * ones.size(), 0));      // (once: the padding behind the n gradient entries)
    hipLaunchKernelGGL(k_po_zero_structured, g_zero, dim3(256), 0, 0, p, n_l1);
  } else {
    PO_TRY(hipMemsetAsync(p.H, 0, hbytes, 0));
    PO_TRY(hipMemsetAsync(p.g, 0, sizeof(double) * ones.size(), 0));
  }
  hipLaunchKernelGGL(k_po_linearise, g_edges, dim3(64), 0, 0, p, 0);
  hipLaunchKernelGGL(k_po_prepare, dim3(1), dim3(256), 0, 0, p, pol, 1);
  // ---- LM iterations, enqueued without host synchronisation; finished solves early-out on device
  // An iteration enqueued behind a finished solve early-outs on the device, but its ~14 launches still cost ~45 us: the host asks the device
  // whether it is done after one iteration more than this thread's PREVIOUS solve took steps (the terminating test runs at the head of the next one; consecutive pose graphs of a session are alike;
  // 8 when there is no history), then every 4.  The answer replaces the wait at the end, it is not an extra one.
  next_check = po_iter_hint >= 0 ? std::min(po_iter_hint + 1, 8) : 8;
  for (int it = 0; it < pol.max_num_iterations && n > 0; ++it) {
    if (it == next_check) {             // stop enqueueing once the device reports termination
      // (state | trace | poses in one copy: when the solve has finished, this IS the download)
      PO_TRY(hipMemcpyAsync(stage, arena, down_bytes, hipMemcpyDeviceToHost, 0));
      PO_TRY(hipStreamSynchronize(0));
      std::memcpy(&hst, stage + ((char*)p.st - arena), sizeof(hst));
      if (hst.status != kRunning) { have_results = true; break; }
      next_check += 4;
    }
    if (zero_small) hipLaunchKernelGGL(k_po_zero_structured, g_zero, dim3(256), 0, 0, p, n_l1);
    else {
      PO_TRY(hipMemsetAsync(p.H, 0, hbytes, 0));
      PO_TRY(hipMemsetAsync(p.g, 0, sizeof(double) * ones.size(), 0));
      PO_TRY(hipMemsetAsync(p.scal, 0, sizeof(double), 0));            // kPoCost
    }
    hipLaunchKernelGGL(k_po_linearise, g_edges, dim3(64), 0, 0, p, 0);
    hipLaunchKernelGGL(k_po_prepare, dim3(1), dim3(256), 0, 0, p, pol, 0);
    if (f32) hipLaunchKernelGGL(k_po_to_f32, dim3(256), dim3(256), 0, 0, p, d_Hf);
    stamp();
    if (structured) {
      // chains eliminated concurrently, then the dense MFMA Cholesky of the junction block only
      {
        size_t off = 0;                                                // level after level: the pieces, then the chains of their cut poses, ...
        for (int cnt : level_counts) {
          if (cnt > 0) hipLaunchKernelGGL(k_po_chain_eliminate, dim3((unsigned)cnt), dim3(64), 0, 0, p, (const PoChain*)(d_chains + off));
          off += (size_t)cnt;
        }
      }
      double* Lf_j = d_Lf + (size_t)n_chain * ld + n_chain;          // the junction block's factor (same leading dimension)
      po_factor_dense<double>(pj, pj.H, Lf_j, d_linv, nblk_j);
      if (nj > 0) hipLaunchKernelGGL(k_po_trisolve<double>, dim3(1), dim3(1024), 0, 0, pj, (const double*)Lf_j, (const double*)d_linv);
      {
        size_t off = chains.size();                                    // ... and back down
        for (size_t lv = level_counts.size(); lv-- > 0;) {
          off -= (size_t)level_counts[lv];
          if (level_counts[lv] > 0) hipLaunchKernelGGL(k_po_chain_backsub, dim3((unsigned)level_counts[lv]), dim3(64), 0, 0, p, (const PoChain*)(d_chains + off));
        }
      }
    }
    if (!structured) {
      if (f32) po_factor_dense<float>(p, d_Hf, d_Lff, d_linvf, nblk);
      else po_factor_dense<double>(p, p.H, d_Lf, d_linv, nblk);
    }
    if (structured) { /* solved above */ }
    else if (nblk >= 4 && nblk <= wide_resident && !std::getenv("SLSLAM_PO_LAUNCH_CHAIN")) {
      // one workgroup per 64-row block, all resident: the substitutions spread over the chip (k_po_trisolve_wide)
      if (f32) hipLaunchKernelGGL(k_po_trisolve_wide<float>, dim3((unsigned)nblk), dim3(256), 0, 0, p, (const float*)d_Lff, (const float*)d_linvf, d_tri_flags, tri_epoch);
      else hipLaunchKernelGGL(k_po_trisolve_wide<double>, dim3((unsigned)nblk), dim3(256), 0, 0, p, (const double*)d_Lf, (const double*)d_linv, d_tri_flags, tri_epoch);
      tri_epoch += 2u;
    }
    else if (f32) hipLaunchKernelGGL(k_po_trisolve<float>, dim3(1), dim3(1024), 0, 0, p, (const float*)d_Lff, (const float*)d_linvf);
    else hipLaunchKernelGGL(k_po_trisolve<double>, dim3(1), dim3(1024), 0, 0, p, (const double*)d_Lf, (const double*)d_linv);
    stamp();
    hipLaunchKernelGGL(k_po_candidate, dim3(1), dim3(256), 0, 0, p);
    hipLaunchKernelGGL(k_po_linearise, g_edges, dim3(64), 0, 0, p, 1);
    hipLaunchKernelGGL(k_po_update, dim3(1), dim3(64), 0, 0, p, pol);
  }
  PO_TRY(hipGetLastError());
  if (timing && tev.size() >= 2) (void)hipEventRecord(tev[1], 0);
  if (!have_results || timing) PO_TRY(hipDeviceSynchronize());
  if (timing) {
    PoTiming& T = g_po_timing;
    T.total_ms = 0; T.factor_ms = 0; T.factor_max_ms = 0; T.factor_calls = 0; T.unknowns = n; T.junction_unknowns = nj;
    float ms = 0.f;
    if (tev.size() >= 2 && hipEventElapsedTime(&ms, tev[0], tev[1]) == hipSuccess) T.total_ms = ms;
    for (size_t i = 2; i + 1 < tev.size(); i += 2)
      if (hipEventElapsedTime(&ms, tev[i], tev[i + 1]) == hipSuccess) { T.factor_ms += ms; T.factor_calls++; if (ms > T.factor_max_ms) T.factor_max_ms = ms; }
    for (hipEvent_t e : tev) (void)hipEventDestroy(e);
    tev.clear();
  }
  // state | trace | poses come back in ONE copy (they are the first bytes of the block)
  if (!have_results) PO_TRY(hipMemcpy(stage, arena, down_bytes, hipMemcpyDeviceToHost));
  std::memcpy(&hst, stage + ((char*)p.st - arena), sizeof(hst));
  std::memcpy(htrace.data(), stage + ((char*)p.trace - arena), sizeof(IterRec) * kMaxTrace);
  std::memcpy(x2.data(), stage + ((char*)p.x - arena), sizeof(double) * 12 * N);
  {
    int term = hst.status == kRunning ? SLSLAM_NO_CONVERGENCE : hst.status;
    if (n == 0) term = SLSLAM_FUNCTION_TOLERANCE;     // no non-constant parameter blocks
    po_iter_hint = hst.n_success + hst.n_unsuccess;
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
  for (hipEvent_t e : tev) (void)hipEventDestroy(e);       // a failed solve leaves through here with its profiling events alive
  DeviceBlockCache::give_back(arena, arena_bytes, arena_device);
  return rc;
}


namespace { thread_local int g_last_level1 = 0; }
/* (inspection, beside slslam_po_structure: how many of the chains it listed - the first ones - are level-1 chains; the rest are the level-2
 * chains of cut poses) */
extern "C" int slslam_po_structure_level1(void) { return g_last_level1; }

extern "C" int slslam_po_structure(const slslam_po_graph* g, int* slot_out, int max_chains, int* num_chains, int* chain_start,
                                   int* chain_len, int* chain_left, int* chain_right, int* num_chain_unknowns, int* num_unknowns) {
  if (!g || !slot_out || !num_chains || g->num_poses < 0 || g->num_edges < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  const int N = g->num_poses, E = g->num_edges;
  if (E > 0 && (!g->pose_index_1 || !g->pose_index_2)) return SLSLAM_ERR_INVALID_ARGUMENT;
  for (int e = 0; e < E; ++e) {
    const int a = g->pose_index_1[e], b = g->pose_index_2[e];
    if (a < 0 || a >= N || b < 0 || b >= N || a == b) return SLSLAM_ERR_INVALID_ARGUMENT;
  }
  std::vector<int> slot(N, -1), used(N, 0);
  for (int e = 0; e < E; ++e) { used[g->pose_index_1[e]] = 1; used[g->pose_index_2[e]] = 1; }
  std::vector<PoChain> chains;
  int n_chain = 0, n = 0;
  std::vector<int> lc;
  if (E > 0) order_chains_first(N, E, g->pose_index_1, g->pose_index_2, used, g->pose_index_1[0], slot, chains, &n_chain, &n, &lc);
  g_last_level1 = lc.empty() ? 0 : lc[0];
  for (int k = 0; k < N; ++k) slot_out[k] = slot[k];
  *num_chains = (int)chains.size();
  if (num_chain_unknowns) *num_chain_unknowns = n_chain;
  if (num_unknowns) *num_unknowns = n;
  if ((int)chains.size() > max_chains) return SLSLAM_ERR_UNSUPPORTED;
  for (size_t c = 0; c < chains.size(); ++c) {
    if (chain_start) chain_start[c] = chains[c].start;
    if (chain_len) chain_len[c] = chains[c].len;
    if (chain_left) chain_left[c] = chains[c].jl;
    if (chain_right) chain_right[c] = chains[c].jr;
  }
  return SLSLAM_OK;
}
