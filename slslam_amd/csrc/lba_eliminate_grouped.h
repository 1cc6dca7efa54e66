// slslam_amd/csrc/lba_eliminate_grouped.h — the elimination sweep of the line bundle adjustment with the Schur outer products
// on the matrix cores and GROUP-LOCAL accumulators (windows with at most 10 free cameras, lines grouped by their first free
// camera: PackedWindow.grouping = 1).
//
// What it replaces: the first observation sweep of an LM iteration — the part of ceres::Solve that evaluates the residual
// blocks LBAProblem::build wires up (reference src/lba_problem.cpp:54-93) and forms the normal equations — i.e. the job of
// k_linearise_schur<false> (lba_kernels.h) and of k_eliminate_mfma (lba_eliminate_mfma.h), whose slab layout and raw camera
// coordinates it shares (the reduced solve assembles S = T^T (blockdiag(J_c'^T J_c') - P) T from it), with a different way of
// holding P = sum_lines X X^T:
//
//   * lane <-> observation, a line owns a run of lanes (unchanged): residual and Jacobians in raw camera coordinates
//     (lba_math.h::obs_linearise_raw), the line's 4x4 block by segmented DPP scans, K = chol(H_ll + D^2)^-1,
//     F_i = (J_c,i'^T J_l,i) K^T (6x4 per observation) — row by row straight into an LDS panel (one 24-double slab per
//     lane), never all in registers;
//   * a sliding window's line is seen by a run of consecutive keyframes.  Counted from its FIRST free camera a, the rows of the
//     reduced system it touches are 6 (hi - a + 1) <= 16 nb, nb = 1..4 blocks of 16 rows; the packer lets the lines of a
//     window follow each other by a.  The wave keeps the lower-triangular 16x16 tiles of the GROUP-LOCAL 48x48 sum (6 tiles =
//     48 registers, v_mfma_f64_16x16x4_f64 accumulators) for the whole run of lines that share a, and adds them into the chunk's
//     slab (global memory, private to the wave: returnless fp64 atomics in program order, so the result is reproducible) only
//     when a changes: a handful of times per chunk.  The fourth block row (lines seen by 9 or 10 free cameras, which the packer
//     keeps together) lives in four more tiles for the duration of one 64-lane tile of lines;
//   * per LINE: lane l fetches X[16 r + (l & 15)][l >> 4] for the nb blocks r (the line's free-camera mask locates the source
//     lane's slab; a camera that does not see the line reads the panel's zero slab) and the wave issues nb (nb + 1) / 2 MFMAs:
//     no operand shuffles, no LDS atomics, no camera-pair work items;
//   * the block-diagonal part J_c'^T J_c', the gradient and b' = g_c' - F K g_l (33 values per observation) go through LDS
//     atomics into one record per free camera, as in k_eliminate_mfma.
#ifndef SLSLAM_LBA_ELIMINATE_GROUPED_H_
#define SLSLAM_LBA_ELIMINATE_GROUPED_H_

#include <type_traits>
#include "lba_kernels.h"
#include "lba_eliminate_mfma.h"
#include "lba_eliminate_grouped_maps.h"

// timing experiments (results WRONG when set): 1 no matrix-core phase, 2 operands fetched but no products issued, 4 one product per line only
#if !defined(GP_ABLATE)
#define GP_ABLATE 0
#endif
// lines fetched together per block count (1: one by one)
#if !defined(GP_GB1)
#define GP_GB1 4
#endif
#if !defined(GP_GB2)
#define GP_GB2 4
#endif
#if !defined(GP_GB3)
#define GP_GB3 2
#endif
#if !defined(GP_SETPRIO)
#define GP_SETPRIO 1
#endif

namespace slslam {


__host__ __device__ inline int lds_bytes_eliminate_grouped(int C, int n) {
  return (kGpPanel + C * kCamTabG + (n / 6) * kDiagRec) * 8 + ((C + 15) / 16) * 16;
}

// One group-local accumulator tile (block row r, block column c <= r; the group's first camera is a) added into the chunk's slab
// (layout of lba_eliminate_mfma_maps.h: tile (I, J) of the window's 64x64 system, entry q * 64 + lane = row (lane >> 4) + 4 q,
// column lane & 15).  Lower triangle only; exact zeros (rows past the cameras of the group's lines) are skipped.
__device__ __forceinline__ void grouped_flush_tile(double* slab, const solve_acc_t& A, int r, int c, int a, int n, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = gp_flush_index(r, c, a, q, lane, n);
    const double v = A[q];
    if (idx >= 0 && v != 0.0) __hip_atomic_fetch_add(slab + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// FRESH: the first sweep of a solve (it is also Ceres' initial evaluation: every window of the launch has LMState.fresh set - the host
// knows which launch that is, as for k_linearise_schur<false, 1>: lm_step ends a window at max_num_iterations, so no window of a
// later launch is fresh); the compile-time form keeps the first sweep's extras out of the steady sweep's registers.
// KEEP (lba_keep_jacobian, off by default - measured slower): the linearising sweep leaves its J blocks in memory
// and the sweep after a rejected step replays them (second loop below); the plain instantiation carries none of it.
// MIXED (slslam_solver_options.lba_precision = 1, steady sweeps only): the camera Jacobian J_c' of an observation is FORMED in float
// (lba_math.h::obs_linearise_raw_mixed: packed fp32; geometry, residuals and the line Jacobian in double - why exactly this split is
// measured there), parked in LDS as floats and widened to double when it comes back; every product and every sum is the double path's.
// Products of two floats are exact in double, so the camera-side blocks the sweep builds are those of a least-squares problem whose
// camera Jacobian is the float one - a consistent system, i.e. a Gauss-Newton step with a J_c' that is off by ~1e-6.  (A first form
// that also took the line Jacobian and the four-term block products in float was measured and dropped: 11 of 28 windows changed an
// accept / reject decision, final costs moved by up to 2e-2, profiles/round5_mixed_precision_study.txt.)
template <bool FRESH, bool KEEP = false, bool MIXED = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_eliminate_grouped(BatchPtrs p, Policy pol) {
  static_assert(!(MIXED && (FRESH || KEEP)), "the mixed-precision form exists for the plain steady sweep only");
  using JT = double;                                                     // Jacobian entries once they are formed (widened in the MIXED form)
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  SLS_K1_STAMP_INIT;
  const Chunk ck = p.chunks[blockIdx.x];
  if (ck.win < 0) return;                              // an unused entry of a refillable batch's chunk array (lba_types.h)
  SLS_K1_WALL(30);
  const WinDesc wd = p.wins[ck.win];
  const LMState* st = p.state + ck.win;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const TileReq rq0 = request_tile(p, ck.tile_begin, ck.tile_end, lane);      // the first tile's context travels while the tables are built
  const double inv_radius = 1.0 / st->radius;
  const bool need_grad = st->need_grad_check != 0;
  const bool same_point = st->same_point != 0;     // J_c'^T J_c' and g_c' in the slab are still those of this point
  constexpr bool fresh = FRESH;                    // this sweep is also Ceres' initial evaluation (see k_linearise_schur)
  // keep_jacobian: a sweep that linearises leaves h = J_c'^T J_l of every observation and the line blocks in memory; the sweep after a
  // rejected step (same point, new radius) starts from them (second loop below)
  constexpr bool keep = !FRESH && KEEP;
  const bool replay = keep && same_point;
  const int n = wd.n, ncf = n / 6;
  double* panel = smem;                                        // [65][kGpSlab]
  double* camtab = panel + kGpPanel;                           // [C][kCamTabG]: R | t
  double* diag = camtab + wd.C * kCamTabG;                     // [ncf][kDiagRec]
  signed char* camcf = (signed char*)(diag + ncf * kDiagRec);
  double* slab = p.slab + ck.slab_off;
  // the first tile's observations are requested before the camera table is built (not after it and the zeroing below: the set-up of a
  // chunk is a chain of round trips to memory - chunk, window and state, cameras, first tile, its observations - and every link
  // that can overlap another one is a few microseconds per chunk)
  TileCtx nxt = resolve_tile(rq0);
  ObsPref pfn;
  if (!keep) prefetch_obs<FRESH, false>(p, nxt, cur, wd.obs_off, pfn, lane);      // (the KEEP form has no register for it here)
  for (int c = lane; c < wd.C; c += 64) {
    const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
    double w[3] = { x[0], x[1], x[2] }, R[9];
    cam_rotation<double>(w, R);
    double* ct = camtab + c * kCamTabG;
    for (int q = 0; q < 9; ++q) ct[q] = R[q];
    ct[9] = x[3]; ct[10] = x[4]; ct[11] = x[5];
    camcf[c] = (signed char)p.cam_cf[wd.cam_off + c];
  }
  for (int q = lane; q < ncf * kDiagRec; q += 64) diag[q] = 0.0;
  if (lane < kGpSlab) panel[64 * kGpSlab + lane] = 0.0;
  // the chunk's tiles of P start at zero: the groups add into them
  for (int q = lane; q < kPTiles * kPTileDoubles; q += 64) slab[q] = 0.0;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // ... and the zeros are in memory before the first add is issued

  SLS_PHASE("prologue");
  solve_acc_t acc[kGpPersist];
#pragma unroll
  for (int e = 0; e < kGpPersist; ++e) acc[e] = solve_acc_t{ 0.0, 0.0, 0.0, 0.0 };
  int cur_a = -1;                                               // first free camera of the group the accumulators belong to

  // ---- matrix-core phase: per line the rank-4 update X X^T of the group-local tiles.  The tile's descriptors come group
  // after group and, inside a group, by their number of blocks (the packer sorted them; a descriptor names the first lane of its
  // line's run), lines without elimination work last: the walk is a few branch-free loops.
  auto matrix_phase = [&](const unsigned descv) {
    SLS_K1_STAMP(5);
    SLS_PHASE("mfma_setup");
    int l2 = lane;
    asm volatile("" : "+v"(l2));                   // keeps the fetch constants out of the registers live across the tile
    // lane l wants X[16 r + (l & 15)][l >> 4]: row rho = 16 r + (l & 15) belongs to the camera `slot` = rho / 6 places after the
    // group's first one, entry rho % 6, column l >> 4 of that camera's F block: byte offset pre[r] from the slab of the line's first lane
    int pre[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pre[r] = gp_pre(l2, r) * 8;              // (bytes)
    const char* pbytes = reinterpret_cast<const char*>(panel);
    // per line (lane i <-> i-th descriptor): slab of its first lane | end of its cameras' slabs << 16 | range has holes << 31
    const unsigned d_first = (descv >> 10) & 63u, d_wdt = (descv >> 24) & 15u, d_nb = (descv >> 20) & 7u, d_group = (descv >> 16) & 15u;
    const unsigned d_bl = d_first * (unsigned)(kGpSlab * 8) | (d_wdt * (unsigned)(kGpSlab * 8)) << 16 | ((descv >> 23) & 1u) << 31;
    const bool d_active = (descv & 0x3ffu) != 0u;    // (lanes past the tile's lines hold 0)
    // operands of the s-th line, blocks r < NB (its number of blocks).  A line seen by every camera of its range (no holes) has the
    // observation of camera a + i in lane first + i: one add per block; the rows past its last camera (last block only: the others
    // are full) read the panel's zero slab.  With holes the free-camera mask locates the source lane.
    auto fetch = [&](int sidx, auto nbtag, double (&X)[4]) {
      constexpr int NB = decltype(nbtag)::value;
      const unsigned bl = (unsigned)__builtin_amdgcn_readlane((int)d_bl, sidx);
      if ((int)bl >= 0) {
        const int base = (int)(bl & 0xffffu), lim = (int)(bl >> 16);
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          const int ad = (r < NB - 1 || pre[r] < lim) ? pre[r] + base : 64 * (kGpSlab * 8);
          X[r] = *reinterpret_cast<const double*>(pbytes + ad);
        }
      } else {
        const unsigned d = (unsigned)__builtin_amdgcn_readlane((int)descv, sidx);
        const unsigned mask = d & 0x3ffu, a = (d >> 16) & 15u, first = (d >> 10) & 63u;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          const unsigned slot = (unsigned)pre[r] / (unsigned)(kGpSlab * 8), cfb = a + slot;
          const bool present = ((mask >> cfb) & 1u) != 0u;
          const unsigned src = present ? first + (unsigned)__popc(mask & ((1u << cfb) - 1u)) : 64u;
          X[r] = *reinterpret_cast<const double*>(pbytes + (src * (kGpSlab * 8) + (unsigned)pre[r] - slot * (kGpSlab * 8)));
        }
      }
    };
    // The fourth block row (rows 48..63 from the group's first camera: lines seen by 9 or 10 free cameras, which the packer keeps
    // together at the head of their group) has its four tiles in registers this phase has to spare (the linearisation's are dead);
    // they do not outlive the tile of lines.
    solve_acc_t row3[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      row3[c] = solve_acc_t{ 0.0, 0.0, 0.0, 0.0 };
      asm volatile("" : "+v"(row3[c]));              // (defined HERE: not a loop-invariant to carry through the front end)
    }
    bool row3_used = false;
    auto products = [&](auto nbtag, const double (&X)[4]) {
      constexpr int NB = decltype(nbtag)::value;
      if (GP_ABLATE & 2) { for (int r = 0; r < NB; ++r) keep_alive(X[r]); return; }
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0], X[0], acc[0], 0, 0, 0);
      if (GP_ABLATE & 4) { for (int r = 1; r < NB; ++r) keep_alive(X[r]); return; }
      if (NB >= 2) {
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1], X[0], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1], X[1], acc[2], 0, 0, 0);
      }
      if (NB >= 3) {
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[2], X[0], acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[2], X[1], acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[2], X[2], acc[5], 0, 0, 0);
      }
      if (NB >= 4) {
        row3[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[3], X[0], row3[0], 0, 0, 0);
        row3[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[3], X[1], row3[1], 0, 0, 0);
        row3[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[3], X[2], row3[2], 0, 0, 0);
        row3[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[3], X[3], row3[3], 0, 0, 0);
      }
    };
    // lines [s, e) of one block count, GB at a time: the operands of the next GB lines are requested before the products of these
    // are issued - the matrix pipe gets GB * NB (NB + 1) / 2 products back to back while the next round trip to the panel is
    // under way; the last (e - s) % GB lines one by one
    auto run = [&](int s, int e, auto nbtag, auto gbtag) {
      constexpr int GB = decltype(gbtag)::value;
      if (s >= e) return;
      if (GB > 1 && s + GB <= e) {
        double Xa[GB][4], Xb[GB][4];
#pragma unroll
        for (int i = 0; i < GB; ++i) fetch(s + i, nbtag, Xa[i]);
        for (; s + 2 * GB <= e; s += 2 * GB) {
#pragma unroll
          for (int i = 0; i < GB; ++i) fetch(s + GB + i, nbtag, Xb[i]);
#pragma unroll
          for (int i = 0; i < GB; ++i) products(nbtag, Xa[i]);
          if (s + 3 * GB <= e) {
#pragma unroll
            for (int i = 0; i < GB; ++i) fetch(s + 2 * GB + i, nbtag, Xa[i]);
          }
#pragma unroll
          for (int i = 0; i < GB; ++i) products(nbtag, Xb[i]);
        }
        if (s + GB <= e) {
#pragma unroll
          for (int i = 0; i < GB; ++i) products(nbtag, Xa[i]);
          s += GB;
        }
      }
      if (s >= e) return;
      double Xa[4], Xb[4];
      fetch(s, nbtag, Xa);
      for (; s + 1 < e; s += 2) {
        fetch(s + 1, nbtag, Xb);
        products(nbtag, Xa);
        if (s + 2 < e) fetch(s + 2, nbtag, Xa);
        products(nbtag, Xb);
      }
      if (s < e) products(nbtag, Xa);
    };
    auto flush_group = [&](int a) {
      grouped_flush_tile(slab, acc[0], 0, 0, a, n, l2); grouped_flush_tile(slab, acc[1], 1, 0, a, n, l2);
      grouped_flush_tile(slab, acc[2], 1, 1, a, n, l2); grouped_flush_tile(slab, acc[3], 2, 0, a, n, l2);
      grouped_flush_tile(slab, acc[4], 2, 1, a, n, l2); grouped_flush_tile(slab, acc[5], 2, 2, a, n, l2);
#pragma unroll
      for (int e = 0; e < kGpPersist; ++e) acc[e] = solve_acc_t{ 0.0, 0.0, 0.0, 0.0 };
    };
    auto flush_row3 = [&](int a) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { grouped_flush_tile(slab, row3[c], 3, c, a, n, l2); row3[c] = solve_acc_t{ 0.0, 0.0, 0.0, 0.0 }; }
      row3_used = false;
    };
    if (GP_SETPRIO) __builtin_amdgcn_s_setprio(1);   // the few VALU slots this phase needs come first: they feed the matrix pipe
    const int nact = (GP_ABLATE & 1) ? 0 : __popcll(__ballot(d_active));
    for (int sb = 0; sb < nact;) {                   // a segment: the lines of one group (a tile has one, at a seam two)
      const int a = (int)(((unsigned)__builtin_amdgcn_readlane((int)descv, sb) >> 16) & 15u);
      const bool in_seg = d_active && d_group == (unsigned)a;
      const int e1 = sb + __popcll(__ballot(in_seg && d_nb == 1u)), e2 = e1 + __popcll(__ballot(in_seg && d_nb == 2u));
      const int e3 = e2 + __popcll(__ballot(in_seg && d_nb == 3u)), e4 = e3 + __popcll(__ballot(in_seg && d_nb >= 4u));
      if (a != cur_a) {
        if (cur_a >= 0) { flush_group(cur_a); if (row3_used) flush_row3(cur_a); }
        cur_a = a;
      }
      run(sb, e1, std::integral_constant<int, 1>(), std::integral_constant<int, GP_GB1>());
      run(e1, e2, std::integral_constant<int, 2>(), std::integral_constant<int, GP_GB2>());
      run(e2, e3, std::integral_constant<int, 3>(), std::integral_constant<int, GP_GB3>());
      if (e4 > e3) { run(e3, e4, std::integral_constant<int, 4>(), std::integral_constant<int, 1>()); row3_used = true; }
      sb = e4;
    }
    SLS_K1_STAMP(6);
    SLS_PHASE("row3");
    if (row3_used) flush_row3(cur_a);
    if (GP_SETPRIO) __builtin_amdgcn_s_setprio(0);
    SLS_K1_STAMP(9);
  };

  double acc_cost = 0.0, acc_fixed = 0.0, acc_gmax = 0.0, acc_xn2 = 0.0;
  int fail = 0;
  if (!replay) {
    if (keep) prefetch_obs<FRESH, false>(p, nxt, cur, wd.obs_off, pfn, lane);
    SLS_K1_STAMP(0);
    for (int t = ck.tile_begin; t < ck.tile_end; ++t) {
      SLS_PHASE("tile_head");
      const TileCtx tc = nxt;
      const ObsPref pf = pfn;
      const unsigned descv = tc.desc;
      const SegCtx sg = make_seg(tc, lane);
      const int j = tc.j, ls = tc.ls, k = tc.k;
      const bool line_ok = tc.line_ok;
      const bool valid = line_ok && j < k;
      const bool line_free = line_ok && !(tc.lflags & 1);
      const int cf = camcf[pf.cam];
      const bool kept = valid && !(cf < 0 && !line_free);
      const bool line_active = line_free && k > 0;     // uniform over the line's run
      const bool cam_free = valid && cf >= 0;
      // (first sweep: |x|^2 of the line's parameters for Ceres' initial evaluation, from the prefetched record - a load where it is
      // needed would stand there for a round trip)
      const double un2 = fresh ? pf.u[0] * pf.u[0] + pf.u[1] * pf.u[1] + pf.u[2] * pf.u[2] + pf.u[3] * pf.u[3] : 0.0;

      // ---- residual and Jacobians in raw camera coordinates, robustified; the line's columns Jacobi-scaled.  The rows of J_c' are
      // parked in the lane's own slab of the F panel (free until this tile's F rows are written) and come back when the
      // linearisation's operands are dead: the register peak of the sweep is here, and the group's accumulators sit on top of it
      SLS_PHASE("linearise");
      double rs[4];
      JT Jl[16];
      double* slabF = panel + lane * kGpSlab;
      {
        const double* ct = camtab + pf.cam * kCamTabG;
        double R[9], tt[3], sl[4], cost;
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = ct[q];
        tt[0] = ct[9]; tt[1] = ct[10]; tt[2] = ct[11];
#pragma unroll
        for (int a = 0; a < 4; ++a) sl[a] = fresh ? 1.0 : pf.lsc[a];
        if constexpr (MIXED) {
          obs_linearise_raw_mixed(R, tt, pf.trig, sl, pf.ob, pol.baseline, pol.huber_delta, rs, Jl, &cost,
            [&](int row, const float (&jc)[6]) {          // parked as floats (half the LDS traffic), widened when they come back
              float2* dst = reinterpret_cast<float2*>(slabF) + 3 * row;
              dst[0] = make_float2(jc[0], jc[1]); dst[1] = make_float2(jc[2], jc[3]); dst[2] = make_float2(jc[4], jc[5]);
            });
        } else {
          obs_linearise_raw<double>(R, tt, pf.trig, sl, pf.ob, pol.baseline, pol.huber_delta, rs, Jl, &cost,
            [&](int row, const double (&jc)[6]) {
              double2* dst = reinterpret_cast<double2*>(slabF + 6 * row);
              dst[0] = make_double2(jc[0], jc[1]); dst[1] = make_double2(jc[2], jc[3]); dst[2] = make_double2(jc[4], jc[5]);
            });
        }
        if (kept) acc_cost += cost;
        if (fresh && valid && !kept) acc_fixed += cost;
      }
      SLS_K1_STAMP(1);
      SLS_PHASE("fetch_next_ctx");
      const TileReq rq = request_tile(p, t + 1, ck.tile_end, lane);        // in flight while the rest of this tile is processed

      // ---- the line's 4x4 block and gradient, summed over its run of lanes
      SLS_PHASE("line_block");
      double H[10], g[4];
      // the launch after an accepted step tests the gradient in unscaled coordinates: the line's Jacobi scale again (its registers went
      // to the linearisation), requested here so that the round trip overlaps the scans instead of standing before its use
      // (on every sweep: a load under the uniform condition would be merged with its default by copies, i.e. waited for on the spot)
      double lsc_again[4] = { 1.0, 1.0, 1.0, 1.0 };
      if (!fresh) {
        const double2* lp = reinterpret_cast<const double2*>(p.line_scale + (long long)(line_ok ? ls : 0) * 4);
        const double2 s01 = lp[0], s23 = lp[1];
        lsc_again[0] = s01.x; lsc_again[1] = s01.y; lsc_again[2] = s23.x; lsc_again[3] = s23.y;
      }
      {
        double v[14];
        int q = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            JT h = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) h += Jl[4 * r + a] * Jl[4 * r + b];
            v[q++] = (double)h;
          }
        JT rsj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rsj[r] = (JT)rs[r];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          JT ga = 0;
#pragma unroll
          for (int r = 0; r < 4; ++r) ga += Jl[4 * r + a] * rsj[r];
          v[10 + a] = (double)ga;
        }
        // (the run totals come back through ds_bpermute: this sweep's LDS pipe is lightly loaded - unlike the LDS-atomic sweep's, which
        // moves them on the VALU - and 28 of them are cheaper than four rounds of 28 selects; measured 1.235 -> 1.224 ms)
        seg_sum_n<14, false, false>(v, sg);
        if (!fresh) {
          // the scale has arrived by now; taking it HERE, on every path, keeps its wait ahead of the stores of the line's factor below (a
          // wait behind them - the registers are re-used on the lanes that do not test the gradient - would stand until the stores are
          // acknowledged)
          asm volatile("" : "+v"(lsc_again[0]), "+v"(lsc_again[1]), "+v"(lsc_again[2]), "+v"(lsc_again[3]));
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) H[i] = v[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) g[i] = v[10 + i];
      }
      SLS_K1_STAMP(2);
      SLS_PHASE("fresh_scale");
      if (fresh) {
        // first sweep of a solve: Jacobi scale of the line from its unscaled block, then continue in scaled line coordinates
        double sl[4];
        const double d[4] = { H[0], H[2], H[5], H[9] };
#pragma unroll
        for (int a = 0; a < 4; ++a) sl[a] = (pol.jacobi_scaling && line_active) ? 1.0 / (1.0 + sqrt(d[a])) : 1.0;
        if (line_ok && j == 0) {
          double* lsc = p.line_scale + (long long)ls * 4;
          for (int a = 0; a < 4; ++a) {
            lsc[a] = sl[a];
            if (line_active) acc_gmax = fmax(acc_gmax, fabs(g[a]));
          }
          if (line_active) acc_xn2 += un2;
        }
        H[0] *= sl[0] * sl[0]; H[1] *= sl[1] * sl[0]; H[2] *= sl[1] * sl[1]; H[3] *= sl[2] * sl[0]; H[4] *= sl[2] * sl[1];
        H[5] *= sl[2] * sl[2]; H[6] *= sl[3] * sl[0]; H[7] *= sl[3] * sl[1]; H[8] *= sl[3] * sl[2]; H[9] *= sl[3] * sl[3];
#pragma unroll
        for (int a = 0; a < 4; ++a) g[a] *= sl[a];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int a = 0; a < 4; ++a) Jl[4 * q + a] *= sl[a];
      }

      // ---- eliminate the line: A = H + D^2, A^-1 = K^T K; z = K^T (K g) = A^-1 g
      SLS_PHASE("factor4x4");
      double K[10], z[4] = { 0, 0, 0, 0 };
      {
        double D2[4], u[4];
        lm_diag4(H, pol, inv_radius, D2);
        bool okc = true;
        if (line_active) okc = chol4_inverse(H, D2, K);
        else { for (int q = 0; q < 10; ++q) K[q] = 0.0; }
        if (!okc) fail = 1;
        if (line_active) {
          u[0] = K[0] * g[0];
          u[1] = K[1] * g[0] + K[2] * g[1];
          u[2] = K[3] * g[0] + K[4] * g[1] + K[5] * g[2];
          u[3] = K[6] * g[0] + K[7] * g[1] + K[8] * g[2] + K[9] * g[3];
          z[0] = K[0] * u[0] + K[1] * u[1] + K[3] * u[2] + K[6] * u[3];
          z[1] = K[2] * u[1] + K[4] * u[2] + K[7] * u[3];
          z[2] = K[5] * u[2] + K[8] * u[3];
          z[3] = K[9] * u[3];
          if (need_grad && line_ok && j == 0) {       // only the launch after an accepted step tests the gradient
            for (int a = 0; a < 4; ++a) acc_gmax = fmax(acc_gmax, fabs(g[a] * fast_rcp(lsc_again[a])));
          }
          if (j == 0) {                                // the line's factor, for the back-substitution of this iteration
            double* le = p.line_elim + (long long)ls * p.line_elim_stride;
#pragma unroll
            for (int q = 0; q < 10; ++q) le[q] = K[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) { le[kLeD2 + q] = D2[q]; le[kLeG + q] = g[q]; }
            if (keep) {                                // ... and its undamped block, for the sweep after a rejected step
              double2* lh = reinterpret_cast<double2*>(p.line_h + (long long)ls * 10);
#pragma unroll
              for (int q = 0; q < 5; ++q) lh[q] = make_double2(H[2 * q], H[2 * q + 1]);
            }
          }
        }
      }

      // ---- row a of the observation's blocks: h = J_c'[:, a]^T J_l (1x4), F[a] = h K^T to the panel, b'[a] = g'[a] - h z, and the
      // camera record (J_c'^T J_c' lower triangle, g', b'); after a rejected step the slab keeps J_c'^T J_c' and g'.
      SLS_K1_STAMP(3);
      SLS_PHASE("f_rows");
      {
        JT Jc[24];
        double2* hkeep = reinterpret_cast<double2*>(p.fstore) + ((long long)t * (12 * 64) + lane);
        if constexpr (MIXED) {
#pragma unroll
          for (int q = 0; q < 12; ++q) { const float2 v2 = reinterpret_cast<const float2*>(slabF)[q]; Jc[2 * q] = (double)v2.x; Jc[2 * q + 1] = (double)v2.y; }
        } else {
#pragma unroll
          for (int q = 0; q < 12; ++q) { const double2 v2 = reinterpret_cast<const double2*>(slabF)[q]; Jc[2 * q] = v2.x; Jc[2 * q + 1] = v2.y; }
        }
        JT rsj[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rsj[r] = (JT)rs[r];
        double* rec = diag + (cam_free ? cf : 0) * kDiagRec;
        // (no skewed adds here - the packer's flag for lanes of a row that share a camera, see the diagonal block of
        // k_linearise_schur: with a third of that sweep's LDS atomics left it measures neutral, 1.224 / 1.222 ms)
        auto emit = [&](int off, double val) { if (cam_free) lds_add_rec(rec + off, val); };
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          JT gaj = 0, hj[4] = { 0, 0, 0, 0 };
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            gaj += Jc[6 * r + a] * rsj[r];
#pragma unroll
            for (int b = 0; b < 4; ++b) hj[b] += Jc[6 * r + a] * Jl[4 * r + b];
          }
          const double ga = (double)gaj, h[4] = { (double)hj[0], (double)hj[1], (double)hj[2], (double)hj[3] };
          const double f0 = h[0] * K[0];
          const double f1 = h[0] * K[1] + h[1] * K[2];
          const double f2 = h[0] * K[3] + h[1] * K[4] + h[2] * K[5];
          const double f3 = h[0] * K[6] + h[1] * K[7] + h[2] * K[8] + h[3] * K[9];
          reinterpret_cast<double2*>(slabF)[2 * a] = make_double2(f0, f1);
          reinterpret_cast<double2*>(slabF)[2 * a + 1] = make_double2(f2, f3);
          if (keep) {                                    // (every lane: whole 1 KB rows)
            hkeep[(2 * a) * 64] = make_double2(h[0], h[1]);
            hkeep[(2 * a + 1) * 64] = make_double2(h[2], h[3]);
          }
          const double fu = h[0] * z[0] + h[1] * z[1] + h[2] * z[2] + h[3] * z[3];
          emit(kDiagB + a, ga - fu);
          if (!same_point) {
            emit(kDiagG + a, ga);
#pragma unroll
            for (int b = 0; b <= a; ++b) {
              JT v = 0;
#pragma unroll
              for (int r = 0; r < 4; ++r) v += Jc[6 * r + a] * Jc[6 * r + b];
              emit(tri_index(a, b), (double)v);
            }
          }
        }
      }

      SLS_K1_STAMP(4);
      SLS_PHASE("prefetch_next");
      // the next tile's loads go out here: their latency overlaps the matrix-core phase
      nxt = resolve_tile(rq);
      prefetch_obs<FRESH, false>(p, nxt, cur, wd.obs_off, pfn, lane);

      matrix_phase(descv);
    }
  } else {
    // ---- the sweep after a rejected step: the point has not moved, so every J block is what the last linearising sweep computed.
    // Per observation h = J_c'^T J_l comes back from memory (12 x 16 bytes, one tile ahead), per line its block and gradient; the
    // new radius gives new D^2, K, z, the F rows h K^T for the panel and b' = g' - sum h z (the sum of g' is still in the slab: the
    // record gets - h z only and the write-out adds the two).  No residual, no Jacobian, no scan.
    struct Kept { double2 h[12]; double2 H[5]; double2 g[2]; int cam; };
    auto request = [&](int t, const TileCtx& c, Kept& kq) {
      const double2* hk = reinterpret_cast<const double2*>(p.fstore) + ((long long)(t < ck.tile_end ? t : ck.tile_begin) * (12 * 64) + lane);
#pragma unroll
      for (int q = 0; q < 12; ++q) kq.h[q] = hk[q * 64];
      const bool valid = c.line_ok && c.j < c.k;
      kq.cam = p.ob_cam[valid ? c.o0 + c.j : wd.obs_off];
      const int lsafe = c.line_ok ? c.ls : 0;
      const double2* lh = reinterpret_cast<const double2*>(p.line_h + (long long)lsafe * 10);
#pragma unroll
      for (int q = 0; q < 5; ++q) kq.H[q] = lh[q];
      const double2* lg = reinterpret_cast<const double2*>(p.line_elim + (long long)lsafe * p.line_elim_stride + kLeG);
      kq.g[0] = lg[0]; kq.g[1] = lg[1];
    };
    Kept kn;
    request(ck.tile_begin, nxt, kn);
    for (int t = ck.tile_begin; t < ck.tile_end; ++t) {
      SLS_PHASE("replay_head");
      const TileCtx tc = nxt;
      const Kept kp = kn;
      const unsigned descv = tc.desc;
      const int j = tc.j, ls = tc.ls, k = tc.k;
      const bool line_ok = tc.line_ok;
      const bool valid = line_ok && j < k;
      const bool line_free = line_ok && !(tc.lflags & 1);
      const int cf = camcf[kp.cam];
      const bool line_active = line_free && k > 0;
      const bool cam_free = valid && cf >= 0;
      const TileReq rq = request_tile(p, t + 1, ck.tile_end, lane);      // resolved where the next tile's blocks are requested
      SLS_PHASE("replay_factor");
      double K[10], z[4] = { 0, 0, 0, 0 };
      {
        double H[10], g[4], D2[4], u[4];
#pragma unroll
        for (int q = 0; q < 5; ++q) { H[2 * q] = kp.H[q].x; H[2 * q + 1] = kp.H[q].y; }
        g[0] = kp.g[0].x; g[1] = kp.g[0].y; g[2] = kp.g[1].x; g[3] = kp.g[1].y;
        lm_diag4(H, pol, inv_radius, D2);
        bool okc = true;
        if (line_active) okc = chol4_inverse(H, D2, K);
        else { for (int q = 0; q < 10; ++q) K[q] = 0.0; }
        if (!okc) fail = 1;
        if (line_active) {
          u[0] = K[0] * g[0];
          u[1] = K[1] * g[0] + K[2] * g[1];
          u[2] = K[3] * g[0] + K[4] * g[1] + K[5] * g[2];
          u[3] = K[6] * g[0] + K[7] * g[1] + K[8] * g[2] + K[9] * g[3];
          z[0] = K[0] * u[0] + K[1] * u[1] + K[3] * u[2] + K[6] * u[3];
          z[1] = K[2] * u[1] + K[4] * u[2] + K[7] * u[3];
          z[2] = K[5] * u[2] + K[8] * u[3];
          z[3] = K[9] * u[3];
          if (j == 0) {                                // the line's factor at the new radius (its gradient stays)
            double* le = p.line_elim + (long long)ls * p.line_elim_stride;
#pragma unroll
            for (int q = 0; q < 10; ++q) le[q] = K[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) le[kLeD2 + q] = D2[q];
          }
        }
      }
      SLS_PHASE("replay_rows");
      {
        double* slabF = panel + lane * kGpSlab;
        double* rec = diag + (cam_free ? cf : 0) * kDiagRec;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double h0 = kp.h[2 * a].x, h1 = kp.h[2 * a].y, h2 = kp.h[2 * a + 1].x, h3 = kp.h[2 * a + 1].y;
          const double f0 = h0 * K[0];
          const double f1 = h0 * K[1] + h1 * K[2];
          const double f2 = h0 * K[3] + h1 * K[4] + h2 * K[5];
          const double f3 = h0 * K[6] + h1 * K[7] + h2 * K[8] + h3 * K[9];
          reinterpret_cast<double2*>(slabF)[2 * a] = make_double2(f0, f1);
          reinterpret_cast<double2*>(slabF)[2 * a + 1] = make_double2(f2, f3);
          const double fu = h0 * z[0] + h1 * z[1] + h2 * z[2] + h3 * z[3];
          if (cam_free) lds_add_rec(rec + kDiagB + a, -fu);
        }
      }
      SLS_PHASE("replay_request");
      nxt = resolve_tile(rq);
      request(t + 1, nxt, kn);                         // in flight across the matrix-core phase
      matrix_phase(descv);
    }
  }
  SLS_K1_STAMP(7);
  SLS_PHASE("epilogue");
  if (cur_a >= 0) {
    grouped_flush_tile(slab, acc[0], 0, 0, cur_a, n, lane); grouped_flush_tile(slab, acc[1], 1, 0, cur_a, n, lane);
    grouped_flush_tile(slab, acc[2], 1, 1, cur_a, n, lane); grouped_flush_tile(slab, acc[3], 2, 0, cur_a, n, lane);
    grouped_flush_tile(slab, acc[4], 2, 1, cur_a, n, lane); grouped_flush_tile(slab, acc[5], 2, 2, cur_a, n, lane);
  }

  // ---- write-out: camera records, scalars
  __syncthreads();
  double* drec = slab + kPTiles * kPTileDoubles;
  for (int q = lane; q < ncf * kDiagRec; q += 64) {
    const int e = q % kDiagRec;
    if (same_point && (e < kDiagB || e >= kDiagG)) continue;     // only b was accumulated: the slab keeps the rest
    drec[q] = replay ? diag[q] + drec[q + (kDiagG - kDiagB)] : diag[q];      // (replay: - sum h z was accumulated, g' is next to it)
  }
  const double c_sum = wave_sum(acc_cost), f_sum = wave_sum(acc_fixed), x_sum = wave_sum(acc_xn2);
  const double g_max = wave_max(acc_gmax);
  const int any_fail = __any(fail);
  if (lane == 0) {
    double* sc = drec + ncf * kDiagRec;
    sc[kScCost] = c_sum; sc[kScFixedCost] = f_sum; sc[kScGradMaxLine] = g_max; sc[kScXn2Line] = x_sum;
    sc[kScFail] = any_fail ? 1.0 : 0.0;
  }
  SLS_K1_STAMP(8);
  SLS_K1_WALL(31);
}

}  // namespace slslam
#endif  // SLSLAM_LBA_ELIMINATE_GROUPED_H_
