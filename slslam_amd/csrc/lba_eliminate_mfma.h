// slslam_amd/csrc/lba_eliminate_mfma.h — the elimination sweep of the line bundle adjustment with the Schur outer
// products on the matrix cores (windows with at most 10 free cameras: reduced system of order <= 60).
//
// What it replaces: the first observation sweep of an LM iteration — the part of ceres::Solve that evaluates the
// residual blocks LBAProblem::build wires up (reference src/lba_problem.cpp:54-93) and forms the normal equations —
// i.e. the same job as k_linearise_schur<false> (lba_kernels.h), with a different accumulation scheme:
//
//   * lane <-> observation, a line owns a run of lanes (unchanged): residual, Jacobians, Huber, the line's 4x4 block
//     by segmented DPP scans, K = chol(H_ll + D^2)^-1, F_i = (J_c,i^T J_l,i) K^T  (6x4 per observation);
//   * the wave lays its F blocks into an LDS panel, one 6-row slab per lane and one plane per column k of F
//     (plane stride == 16 mod 32 doubles: the operand reads below are bank-conflict free);
//   * per LINE the Schur complement term  - sum_{i,j} F_i F_j^T  is the rank-4 update  - X X^T  with X (60 x 4) the line's
//     F blocks stacked by free camera.  The 60 x 60 sum lives in the REGISTERS of the workgroup's waves as 16x16
//     accumulator tiles of v_mfma_f64_16x16x4_f64 (lower triangle: 10 tiles = 80 registers, split over the waves of
//     the workgroup); for a line, lane l of a wave fetches X[16 I + (l & 15)][l >> 4] for each 16-row block I the line
//     touches (a popcount over the line's free-camera mask locates the source lane's slab in the panel) and the wave
//     issues one MFMA per touched tile (4.4 per line on the bench window).  No atomics, no operand shuffles, and the
//     matrix pipe runs beside the VALU work of the SIMD's other wave;
//   * only the block-diagonal part J_c^T J_c, the gradient and b = g_c - F K g_l (33 values per observation) still go
//     through LDS atomics, into one record per free camera.
//
// Slab written per chunk (sys_doubles_mfma): [10 tiles x 256 doubles in accumulator order] [ncf x 33 camera records]
// [8 scalars]; k_reduced_solve assembles S = blockdiag(J_c^T J_c) - P from it.
#ifndef SLSLAM_LBA_ELIMINATE_MFMA_H_
#define SLSLAM_LBA_ELIMINATE_MFMA_H_

#include "lba_kernels.h"
#include "lba_gram.h"
#include "lba_eliminate_mfma_maps.h"

namespace slslam {

__host__ __device__ inline int lds_bytes_eliminate_mfma(int C, int n, int NW) {
  const int doubles = C * 13 /* kCamTabG */ + NW * (n / 6) * kDiagRec + NW * kPanelDoubles + NW * 8;
  return doubles * 8 + NW * 64 * 4 + NW * kGatherLines * 64 + 16 * 4 + ((C + 15) / 16) * 16;
}

// The operands X[16 I + (lane & 15)][lane >> 4], I = 0..3, of one line: four LDS reads, issued together.  `src` holds, one
// byte per block I, the lane whose slab has the row's camera (64: the camera does not see the line - the read lands in the
// zeroed pad of the plane).
__device__ __forceinline__ void fetch_operands(const double* panel, const int (&off)[4], unsigned src, double (&X)[4]) {
#pragma unroll
  for (int I = 0; I < 4; ++I) X[I] = panel[off[I] + 6 * (int)((src >> (8 * I)) & 0xffu)];
}
// `src` of a line from its descriptor (tiles with more lines than the gather table holds): the observation of free camera
// cf is the lane  first lane + (number of the line's free cameras below cf).
__device__ __forceinline__ unsigned sources_from_mask(int lane, unsigned d) {
  const unsigned M = d & 0x3ffu, first = (d >> 10) & 63u;
  unsigned src = 0u;
#pragma unroll
  for (int I = 0; I < 4; ++I) {
    const int cf = (16 * I + (lane & 15)) / 6;
    const unsigned pl = ((M >> cf) & 1u) ? first + (unsigned)__popc(M & ((1u << cf) - 1u)) : 64u;
    src |= pl << (8 * I);
  }
  return src;
}

// The rank-4 updates of one panel's lines into the accumulator tiles this wave owns, kMfmaBatch lines per round.
// Software-pipelined: while the MFMAs of one round are issued, the operands of the next round (requested before them) and
// the sources of the round after (requested behind those) are in flight: two LDS round trips hide behind ~2.2 MFMAs per
// line and wave.  The descriptor's tile bits say which accumulators a line updates.
// tab: the sources come from the tile's gather table (else from the line masks).
// flags (timing experiments only): bit 1 = fetch the operands but issue no MFMA.
enum { kMfmaBatch = 2 };
template <int NW, int W>
__device__ __forceinline__ void mfma_panel(const double* panel, const unsigned* gtab, bool tab, unsigned descv, int nlines,
                                           const int (&off)[4], int lane, solve_acc_t (&acc)[ptile_slots(NW)], int flags) {
  if (nlines <= 0) return;
  const int r16 = lane & 15;
  auto sources = [&](int s) -> unsigned {
    s = s < nlines ? s : nlines - 1;
    return tab ? gtab[16 * s + r16] : sources_from_mask(lane, (unsigned)__builtin_amdgcn_readlane((int)descv, s));
  };
  double Xn[kMfmaBatch][4];
  unsigned srcn[kMfmaBatch];
#pragma unroll
  for (int i = 0; i < kMfmaBatch; ++i) fetch_operands(panel, off, sources(i), Xn[i]);
#pragma unroll
  for (int i = 0; i < kMfmaBatch; ++i) srcn[i] = sources(kMfmaBatch + i);
  for (int s0 = 0; s0 < nlines; s0 += kMfmaBatch) {
    double X[kMfmaBatch][4];
    unsigned d[kMfmaBatch];
#pragma unroll
    for (int i = 0; i < kMfmaBatch; ++i) {
      d[i] = (unsigned)__builtin_amdgcn_readlane((int)descv, (s0 + i) & 63);      // (lanes past the tile's lines hold 0)
#pragma unroll
      for (int I = 0; I < 4; ++I) X[i][I] = Xn[i][I];
    }
    if (s0 + kMfmaBatch < nlines) {
#pragma unroll
      for (int i = 0; i < kMfmaBatch; ++i) fetch_operands(panel, off, srcn[i], Xn[i]);
#pragma unroll
      for (int i = 0; i < kMfmaBatch; ++i) srcn[i] = sources(s0 + 2 * kMfmaBatch + i);
    }
    if (flags & 2) continue;
#pragma unroll
    for (int i = 0; i < kMfmaBatch; ++i) {
      if ((d[i] >> 16) == 0u) continue;
#pragma unroll
      for (int e = 0; e < ptile_slots(NW); ++e) {          // in place: one accumulator register block per owned tile
        const int t = ptile_of(NW, W, e);
        if (d[i] & (1u << (16 + t)))
          acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[i][ptile_I(t)], X[i][ptile_J(t)], acc[e], 0, 0, 0);
      }
    }
  }
}

// Phase timing of the sweep (debug_flags bit 8, timing experiments only): wave-level cycle stamps at the phase boundaries,
// summed per wave into BatchPtrs::dbg_cycles[(chunk * NW + wave) * 16 + phase].
__device__ __forceinline__ unsigned long long wave_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define SLS_STAMP(i)                                                        \
  do {                                                                      \
    if (DBG && (dbg & 256)) {                                                   \
      __builtin_amdgcn_sched_barrier(0);                                    \
      const unsigned long long now_ = wave_clock();                         \
      tacc[i] += now_ - tlast; tlast = now_;                                \
      __builtin_amdgcn_sched_barrier(0);                                    \
    }                                                                       \
  } while (0)

enum { kCamTabG = 13 };            // doubles per camera in LDS: R[9] t[3] + pad (odd stride: conflict-free ds_read_b64)

// DBG: the instantiation the timing experiments launch (debug_flags != 0); the production instantiation carries none of it.
template <int NW, bool DBG>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2)))
void k_eliminate_mfma(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Chunk ck = p.chunks[blockIdx.x];
  if (ck.win < 0) return;                              // an unused entry of a refillable batch's chunk array (lba_types.h)
  const WinDesc wd = p.wins[ck.win];
  const LMState* st = p.state + ck.win;
  if (st->status != kRunning && !(DBG && pol.debug_flags)) return;     // (timing experiments keep sweeping windows whose garbage results ended them)
  const int cur = st->cur;
  const double inv_radius = 1.0 / st->radius;
  const bool need_grad = st->need_grad_check != 0;
  const bool same_point = st->same_point != 0;     // J_c'^T J_c' and g_c' in the slab are still those of this point
  const bool fresh = st->fresh != 0;               // this sweep is also Ceres' initial evaluation (see k_linearise_schur)
  const int n = wd.n, ncf = n / 6;
  double* camtab = smem;
  double* diag = camtab + wd.C * kCamTabG;                     // [NW][ncf][kDiagRec]: one copy per wave (ordered sum at the end)
  double* panels = diag + NW * ncf * kDiagRec;
  double* red = panels + NW * kPanelDoubles;                   // [NW][8] per-wave scalars
  unsigned* ldesc = (unsigned*)(red + NW * 8);                 // [NW][64]
  unsigned* gtabs = ldesc + NW * 64;                           // [NW][kGatherLines][16]: source lane of (line, row, block)
  int* nlin = (int*)(gtabs + NW * kGatherLines * 16);          // [NW] (16 ints reserved)
  signed char* camcf = (signed char*)(nlin + 16);
  for (int c = tid; c < wd.C; c += 64 * NW) {
    const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
    double w[3] = { x[0], x[1], x[2] }, R[9];
    cam_rotation<double>(w, R);
    double* ct = camtab + c * kCamTabG;
    for (int q = 0; q < 9; ++q) ct[q] = R[q];
    ct[9] = x[3]; ct[10] = x[4]; ct[11] = x[5];
    camcf[c] = (signed char)p.cam_cf[wd.cam_off + c];
  }
  for (int q = tid; q < NW * ncf * kDiagRec; q += 64 * NW) diag[q] = 0.0;
  for (int q = tid; q < NW * 64; q += 64 * NW)                 // the 16-double pad of every plane stays zero: what absent rows read
    panels[(q >> 6) * kPanelDoubles + ((q >> 4) & 3) * kPanelPlane + 384 + (q & 15)] = 0.0;
  __syncthreads();
  const int dbg = DBG ? pol.debug_flags : 0;       // timing experiments only (SLSLAM_DEBUG_ABLATE)

  solve_acc_t acc[ptile_slots(NW)];
#pragma unroll
  for (int e = 0; e < ptile_slots(NW); ++e) acc[e] = solve_acc_t{ 0.0, 0.0, 0.0, 0.0 };
  double* panel = panels + wave * kPanelDoubles;
  double* mydiag = diag + wave * ncf * kDiagRec;
  unsigned* mytab = gtabs + wave * kGatherLines * 16;

  double acc_cost = 0.0, acc_fixed = 0.0, acc_gmax = 0.0, acc_xn2 = 0.0;
  int fail = 0;
  unsigned long long tacc[DBG ? 10 : 1] = { 0 }, tlast = 0;
  if (DBG && (dbg & 256)) tlast = wave_clock();
  TileCtx nxt = fetch_tile(p, ck.tile_begin + wave, ck.tile_end, lane);
  ObsPref pfn;
  prefetch_obs<false, false>(p, nxt, cur, wd.obs_off, pfn, lane);
  for (int t0 = ck.tile_begin; t0 < ck.tile_end; t0 += NW) {
    const int t = t0 + wave;
    const bool active = t < ck.tile_end;            // wave-uniform
    if (active) {
      const TileCtx tc = nxt;
      const ObsPref pf = pfn;
      ldesc[wave * 64 + lane] = tc.desc;            // (line descriptors of the tile: TileCtx, lba_kernels.h)
      if (lane == 0) nlin[wave] = tc.nlines;
      const TileReq rq = request_tile(p, t + NW, ck.tile_end, lane);      // in flight while this tile is processed
      const SegCtx sg = make_seg(tc, lane);
      const int j = tc.j, ls = tc.ls, k = tc.k;
      const bool line_ok = tc.line_ok;
      const bool valid = line_ok && j < k;
      const bool line_free = line_ok && !(tc.lflags & 1);
      const int cf = camcf[pf.cam];
      const bool kept = valid && !(cf < 0 && !line_free);
      const bool line_active = line_free && k > 0;     // uniform over the line's run
      const bool cam_free = valid && cf >= 0;
      const bool elim = cam_free && line_free;         // this observation couples a free camera to a free line

      // ---- Gram data of the observation's four rows (lba_gram.h), Huber weight
      const double* ct = camtab + pf.cam * kCamTabG;
      double dc[3], e2[3], Q[3], W[21], w[6], zc[3];
      {
        double R[9], tt[3], P[3], r[4], cost;
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = ct[q];
        tt[0] = ct[9]; tt[1] = ct[10]; tt[2] = ct[11];
        obs_gram<double>(R, tt, pf.trig, pf.ob, pol.baseline, dc, e2, P, r, W, w);
        zc[0] = R[2]; zc[1] = R[5]; zc[2] = R[8];
        const double sr = huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &cost);
        const double w2 = valid ? sr * sr : 0.0;       // both factors of every product carry sqrt(rho')
#pragma unroll
        for (int q = 0; q < 21; ++q) W[q] *= w2;
#pragma unroll
        for (int q = 0; q < 6; ++q) w[q] *= w2;
        if (kept) acc_cost += cost;
        if (fresh && valid && !kept) acc_fixed += cost;
#pragma unroll
        for (int q = 0; q < 3; ++q) Q[q] = -pf.trig[6] * e2[q];
      }
      SLS_STAMP(0);
      // ---- camera side: J_c'^T J_c' and J_c'^T r in raw coordinates (the reduced solve applies JL and the column scale);
      // after a rejected step both are unchanged and the slab keeps them
      if (cam_free && !same_point && !(dbg & 4)) {
        double D[21], gc[6];
        gram_camera_block<double>(W, Q, dc, D);
        apply_mc<double>(Q, dc, w, gc);
        double* rec = mydiag + cf * kDiagRec;
#pragma unroll
        for (int q = 0; q < 21; ++q) lds_add(&rec[q], D[q]);
#pragma unroll
        for (int q = 0; q < 6; ++q) lds_add(&rec[kDiagG + q], gc[q]);
      }
      SLS_STAMP(1);
      // ---- line side: Y = W Ml^T, the line's 4x4 block and gradient summed over its run of lanes
      double Y[24], v[14];
      {
        double sl[4], Ml[24];
#pragma unroll
        for (int a = 0; a < 4; ++a) sl[a] = fresh ? 1.0 : pf.lsc[a];
        if (!(dbg & 16)) {
          line_rows<double>(dc, e2, zc, pf.trig[6], pf.trig[1], pf.trig[0], sl, Ml);
          gram_line<double>(W, w, Ml, Y, v, v + 10);
        } else {
          for (int q = 0; q < 24; ++q) Y[q] = W[q % 21];
          for (int q = 0; q < 14; ++q) v[q] = w[q % 6] + (q == 0 || q == 2 || q == 5 || q == 9 ? 1.0 : 0.0);
        }
        if (!line_free) {
#pragma unroll
          for (int q = 0; q < 14; ++q) v[q] = 0.0;
        }
      }
      SLS_STAMP(2);
      if (!(dbg & 8)) seg_sum_n<14>(v, sg);
      SLS_STAMP(3);
      double* H = v;
      double* g = v + 10;
      if (fresh) {
        // first sweep of a solve: Jacobi scale of the line from its unscaled block, then continue in scaled line coordinates
        double sl[4];
        const double d[4] = { H[0], H[2], H[5], H[9] };
#pragma unroll
        for (int a = 0; a < 4; ++a) sl[a] = (pol.jacobi_scaling && line_active) ? 1.0 / (1.0 + sqrt(d[a])) : 1.0;
        if (line_ok && j == 0) {
          double* lsc = p.line_scale + (long long)ls * 4;
          const double* ul = p.line_x + line_rec(p, ls, cur);
          for (int a = 0; a < 4; ++a) {
            lsc[a] = sl[a];
            if (line_active) { acc_gmax = fmax(acc_gmax, fabs(g[a])); acc_xn2 += ul[a] * ul[a]; }
          }
        }
        H[0] *= sl[0] * sl[0]; H[1] *= sl[1] * sl[0]; H[2] *= sl[1] * sl[1]; H[3] *= sl[2] * sl[0]; H[4] *= sl[2] * sl[1];
        H[5] *= sl[2] * sl[2]; H[6] *= sl[3] * sl[0]; H[7] *= sl[3] * sl[1]; H[8] *= sl[3] * sl[2]; H[9] *= sl[3] * sl[3];
#pragma unroll
        for (int a = 0; a < 4; ++a) g[a] *= sl[a];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int q = 0; q < 6; ++q) Y[6 * a + q] *= sl[a];
      }

      // ---- eliminate the line: A = H + D^2, A^-1 = K^T K
      double D2[4], K[10], u[4] = { 0, 0, 0, 0 };
      lm_diag4(H, pol, inv_radius, D2);
      bool okc = true;
      if (line_active && !(dbg & 32)) okc = chol4_inverse(H, D2, K);
      else if (line_active) { for (int q = 0; q < 10; ++q) K[q] = H[q]; }
      else { for (int q = 0; q < 10; ++q) K[q] = 0.0; }
      if (!okc) fail = 1;
      if (line_active) {
        u[0] = K[0] * g[0];
        u[1] = K[1] * g[0] + K[2] * g[1];
        u[2] = K[3] * g[0] + K[4] * g[1] + K[5] * g[2];
        u[3] = K[6] * g[0] + K[7] * g[1] + K[8] * g[2] + K[9] * g[3];
        if (need_grad && line_ok && j == 0) {
          for (int a = 0; a < 4; ++a) acc_gmax = fmax(acc_gmax, fabs(g[a] / pf.lsc[a]));
        }
        if (j == 0) {                                  // the line's factor, for the back-substitution of this iteration
          double* le = p.line_elim + (long long)ls * p.line_elim_stride;
#pragma unroll
          for (int q = 0; q < 10; ++q) le[q] = K[q];
#pragma unroll
          for (int q = 0; q < 4; ++q) { le[kLeD2 + q] = D2[q]; le[kLeG + q] = g[q]; }
        }
      }
      SLS_STAMP(4);
      // ---- Z = Y K^T in place (column m = sum_{b <= m} K[m][b] Y_b), F' = Mc Z to the panel, b' = Mc (w - Z u)
      // the tile's gather table: for every line and every row of the stacked system, the lane that holds the row's camera
      // (64: none).  Cleared by all lanes, then every coupled observation enters its six rows.
      if (tc.nlines <= kGatherLines) {
        for (int q = lane; q < tc.nlines * 16; q += 64) mytab[q] = 0x40404040u;
        if (elim) {
          unsigned char* tbytes = (unsigned char*)mytab + tc.slot * 64;
#pragma unroll
          for (int a = 0; a < 6; ++a) { const int r = 6 * cf + a; tbytes[(r & 15) * 4 + (r >> 4)] = (unsigned char)lane; }
        }
      }
      if (cam_free && !(dbg & 64)) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const double y0 = Y[q], y1 = Y[6 + q], y2 = Y[12 + q], y3 = Y[18 + q];
          Y[18 + q] = K[6] * y0 + K[7] * y1 + K[8] * y2 + K[9] * y3;
          Y[12 + q] = K[3] * y0 + K[4] * y1 + K[5] * y2;
          Y[6 + q] = K[1] * y0 + K[2] * y1;
          Y[q] = K[0] * y0;
        }
        if (elim) {
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            double f[6];
            apply_mc<double>(Q, dc, Y + 6 * m, f);
#pragma unroll
            for (int a = 0; a < 6; ++a) panel[panel_store_index(lane, a, m)] = f[a];
          }
        }
        double e[6], bq[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) e[q] = w[q] - (Y[q] * u[0] + Y[6 + q] * u[1] + Y[12 + q] * u[2] + Y[18 + q] * u[3]);
        apply_mc<double>(Q, dc, e, bq);
        double* rec = mydiag + cf * kDiagRec;
#pragma unroll
        for (int q = 0; q < 6; ++q) lds_add(&rec[kDiagB + q], bq[q]);
      }
      SLS_STAMP(5);
      // the next tile's loads go out here: their latency overlaps the matrix-core phase
      nxt = resolve_tile(rq);
      prefetch_obs<false, false>(p, nxt, cur, wd.obs_off, pfn, lane);
    } else {
      if (lane == 0) nlin[wave] = 0;
    }
    SLS_STAMP(6);
    __syncthreads();
    SLS_STAMP(7);
    // ---- every wave: rank-4 updates of all panels' lines into the tiles it owns
    {
      int l2 = lane;
      asm volatile("" : "+v"(l2));                   // keeps the fetch constants out of the registers live across the tile
      __builtin_amdgcn_s_setprio(1);                 // the few VALU slots this phase needs come first: they feed the matrix pipe
      int off[4];
#pragma unroll
      for (int I = 0; I < 4; ++I) { const int r = 16 * I + (l2 & 15); off[I] = (l2 >> 4) * kPanelPlane + (r - 6 * (r / 6)); }
#pragma unroll
      for (int w2 = 0; w2 < NW; ++w2) {
        const int nl = (dbg & 1) ? 0 : __builtin_amdgcn_readfirstlane(nlin[w2]);
        const unsigned dv = ldesc[w2 * 64 + lane];
        const double* pn = panels + w2 * kPanelDoubles;
        const unsigned* gt = gtabs + w2 * kGatherLines * 16;
        const bool tab = nl <= kGatherLines;
        if constexpr (NW == 1) {
          mfma_panel<1, 0>(pn, gt, tab, dv, nl, off, l2, acc, dbg);
        } else {
          if (wave == 0) mfma_panel<2, 0>(pn, gt, tab, dv, nl, off, l2, acc, dbg);
          else mfma_panel<2, 1>(pn, gt, tab, dv, nl, off, l2, acc, dbg);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    SLS_STAMP(8);
    __syncthreads();
    SLS_STAMP(9);
  }
  if (DBG && (dbg & 256) && lane == 0 && p.dbg_cycles) {
    for (int i = 0; i < (DBG ? 10 : 1); ++i) p.dbg_cycles[((long long)ck.id * NW + wave) * 16 + i] = tacc[i];
  }

  // ---- write-out: accumulator tiles (coalesced 512-byte rows), camera records, scalars
  double* slab = p.slab + ck.slab_off;
#pragma unroll
  for (int e = 0; e < ptile_slots(NW); ++e) {
    const int t = NW == 1 ? e : (wave == 0 ? ptile_of(2, 0, e) : ptile_of(2, 1, e));
#pragma unroll
    for (int q = 0; q < 4; ++q) slab[(t * 4 + q) * 64 + lane] = acc[e][q];
  }
  double* drec = slab + kPTiles * kPTileDoubles;
  for (int q = tid; q < ncf * kDiagRec; q += 64 * NW) {
    const int e = q % kDiagRec;
    if (same_point && (e < kDiagB || e >= kDiagG)) continue;     // only b was accumulated: the slab keeps the rest
    double v = diag[q];
    if (NW == 2) v += diag[ncf * kDiagRec + q];                  // fixed order: bitwise reproducible
    drec[q] = v;
  }
  const double c_sum = wave_sum(acc_cost), f_sum = wave_sum(acc_fixed), x_sum = wave_sum(acc_xn2);
  const double g_max = wave_max(acc_gmax);
  const int any_fail = __any(fail);
  if (lane == 0) {
    double* r = red + wave * 8;
    r[0] = c_sum; r[1] = f_sum; r[2] = g_max; r[3] = x_sum; r[4] = any_fail ? 1.0 : 0.0;
  }
  __syncthreads();
  if (tid == 0) {
    double c = 0.0, f = 0.0, gm = 0.0, x = 0.0, fl = 0.0;
    for (int w2 = 0; w2 < NW; ++w2) {
      const double* r = red + w2 * 8;
      c += r[0]; f += r[1]; gm = fmax(gm, r[2]); x += r[3]; fl = fmax(fl, r[4]);
    }
    double* sc = drec + ncf * kDiagRec;
    sc[kScCost] = c; sc[kScFixedCost] = f; sc[kScGradMaxLine] = gm; sc[kScXn2Line] = x; sc[kScFail] = fl;
  }
}

}  // namespace slslam
#endif  // SLSLAM_LBA_ELIMINATE_MFMA_H_
