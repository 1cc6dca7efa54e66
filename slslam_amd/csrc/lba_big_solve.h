// slslam_amd/csrc/lba_big_solve.h - the damped reduced camera system of a window beyond the tiled sweeps (lba_big.h), n <= 256
// unknowns (the reference's W = 40 study: 40 free keyframes, n = 240), factorised and solved in ONE launch by ONE workgroup.
//
// Replaces, for such windows, the launch chain of the pose-graph path's blocked Cholesky (k_po_potrf_diag / k_po_panel_update /
// k_po_trisolve: eleven dependent launches per LM iteration at n = 240, 205 us, more than half of the device time of a W = 40
// solve) - and with it the dense factorisation Ceres' DENSE_SCHUR does inside ceres::Solve (reference src/lba_problem.cpp:99-110
// chooses the solver; what it does to a window is restated in oracle/lba_oracle.c: lba_solve_reduced).
//
// Layout: the lower triangle of S as 16 x 16 blocks, at most 136 of them, lives in the REGISTERS of the workgroup's eight waves
// (block b = i (i + 1) / 2 + j belongs to wave b % 8, slot b / 8: 15 or 17 slots of four doubles per lane, the result layout of
// v_mfma_f64_16x16x4_f64) from the one pass that reads S - applying the Jacobi congruence of a first sweep and the LM damping
// on the way, so nothing but the sweeps that build S ever writes it - until the solution is written: nothing of the factor
// goes back to memory.  Right-looking, one block column k at a time, two workgroup barriers per step:
//   panel     the waves that hold (i, k), i > k: L_ik = A_ik X_kk^T on the MFMA (operands staged through LDS), b_i -= L_ik y_k;
//   trailing  every wave: A_ij -= L_ik L_jk^T for its blocks with i >= j > k, operands from the panel in LDS.  LOOK-AHEAD: the
//             wave that holds (k + 1, k + 1) updates that tile first and factors it meanwhile (dense_tile.h: row per lane, DPP
//             broadcasts, the tile's inverse X from the same sweep; y = X b of the forward substitution) - the sixteen
//             sequential pivots of a tile are the long pole of a step - and puts off its other blocks' updates of this step
//             to the next phase (the panel is double-buffered; only the next panel's blocks are brought up to date at once).
// Then the backward substitution x_i = X_ii^T y_i, y_k -= L_ik^T x_i (k < i), block row by block row from the registers, the
// step statistics of the camera block, the candidate poses and their rotation tables.  No atomics: every sum has one owner
// and a fixed order (bitwise reproducible).
//
// Measured (house-sized W = 40 window, n = 240, tools/big_solve_phases.py): 113 us per launch against 205 us for the launch
// chain; by phase: load 7 %, panels 25 %, trailing with the look-ahead factorisation 55 % (the tiles themselves 29 %),
// backward 13 %.  A phase between two barriers costs ~0.5 us even when nearly empty (the unrolled slot loops are scalar branch
// chains and the kernel is larger than the instruction cache), so what is left is mostly the 60 phases.  Tried and dropped:
// blocks kept transposed so the panel solve multiplies straight from registers (panel phase -19 %, but the strided first read
// of S, the row-wise backward sums and the rest cost more: +7 % overall); two blocks per round of LDS reads in the trailing
// loop (no gain: the MFMA issue rate, not its latency, bounds that loop); four waves with 34 slots (accumulators spill).
#ifndef SLSLAM_LBA_BIG_SOLVE_H_
#define SLSLAM_LBA_BIG_SOLVE_H_

#include "dense_tile.h"
#include "lba_big.h"
#include "po_kernels.h"

namespace slslam {

enum { kBsvMaxN = 256, kBsvWaves = 8, kBsvSlots = 17, kBsvLd = 17, kBsvPark = 0 };   // 136 blocks of 16 over 8 waves; LDS tiles: 34-dword rows

// SLOTS: blocks per wave - 15 for n <= 240 (120 blocks: the reference's W = 40), 17 for n <= 256; the kernel sits at the
// register limit of two waves per SIMD and the two slots less keep the tile factorisation out of scratch
template <int SLOTS>
__global__ __launch_bounds__(64 * kBsvWaves) void k_big_solve(BatchPtrs p, BigPtrs bg, Policy pol) {
  __shared__ double Pan[2][16][16 * kBsvLd];   // block column k of the factor (buffer k & 1), one tile per block row
  __shared__ double Xs[16][16 * kBsvLd];       // inverses of the diagonal tiles
  __shared__ double Dt[16 * kBsvLd];           // the diagonal tile being factored
  __shared__ double yv[kBsvMaxN];              // right-hand side -> y -> solution
  __shared__ double dv[kBsvMaxN], sv[kBsvMaxN]; // LM damping of the diagonal, Jacobi scale of the first sweep (else 1)
  __shared__ v4f64 park[(kBsvPark > 0 ? kBsvPark : 1) * 64];        // blocks of the factoring wave, out of its registers for the duration
  __shared__ int failed;
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WinDesc wd = p.wins[w];
  const LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int n = wd.n, ld = big_ld(n), NB = (n + 15) >> 4;
  const double* S = bg.sys + bg.sys_off[w];
  double* yvec = bg.sys + bg.sys_off[w] + (long long)n * ld + 3LL * n;
  const bool timing = (pol.debug_flags & 512) && p.dbg_cycles;      // phase timing (timing experiments only, tools/solve_phases.py)
  unsigned long long tlast_ = timing ? solve_clock() : 0ull;
  const int am = lane & 15, ak = lane >> 4;            // MFMA operand coordinates (row / k) and result column; result rows ak + 4 q
  // (LDS set up before the loads of S are issued: the first diagonal tile is factored while the rest of them are in flight)
  // the first sweep of a solve built S in unscaled camera coordinates: the congruence with the Jacobi scale k_big_prepare derived
  // is applied to the copy in the registers, and so is the LM damping (every iteration) - k_big_prepare leaves S alone
  const bool rescale = bg.scal[(long long)w * kBgScal + kBgWasFresh] != 0.0;
  if (tid < kBsvMaxN) {
    yv[tid] = tid < n ? yvec[tid] : 0.0;
    dv[tid] = tid < n ? yvec[tid - n] : 0.0;                               // h: the damping term of the diagonal
    sv[tid] = (tid < n && rescale) ? yvec[n + tid] : 1.0;
  }
  if (tid == 0) failed = 0;
  for (int q = tid; q < 16 * 16 * kBsvLd; q += 64 * kBsvWaves) (&Xs[0][0])[q] = 0.0;      // (upper triangles stay zero)
  __syncthreads();
  // the blocks of this wave
  v4f64 acc[SLOTS];
  int bi[SLOTS], bj[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int b = s * kBsvWaves + wave;
    int i = 0;
    while (((i + 1) * (i + 2)) / 2 <= b) ++i;             // (wave-uniform: scalar code)
    const int j = b - (i * (i + 1)) / 2;
    bi[s] = i < NB ? i : -1;
    bj[s] = j;
    const int r0 = 16 * i + ak, c = 16 * j + am, off0 = r0 * ld + c;      // (n ld < 2^17: 32-bit offsets)
    if (16 * i + 16 <= n) {                              // whole block inside the matrix (wave-uniform): plain loads, all in flight at once
#pragma unroll                                         // (the upper triangle of a diagonal block holds zeros nobody reads)
      for (int q = 0; q < 4; ++q) acc[s][q] = S[off0 + 4 * q * ld];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = r0 + 4 * q;
        const bool in = i < NB && r < n && c <= r;
        const double v = S[in ? off0 + 4 * q * ld : 0];
        acc[s][q] = in ? v : (r == c) ? 1.0 : 0.0;     // rows beyond n: identity (scale 1, damping 0 there)
      }
    }
    if (rescale) {
      const double sc = sv[c];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[s][q] *= sv[r0 + 4 * q] * sc;
    }
    if (i == j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (r0 + 4 * q == c) acc[s][q] += dv[c];
    }
  }

  // the diagonal tile (k, k) - already carrying the updates of block columns < k - factored by the wave that holds it, with
  // y_k = X_kk b_k of the forward substitution (lower triangular: lane = (row am, columns 4 ak ..))
  auto factor_diag = [&](const int k, const v4f64 d) {
    const unsigned long long tf0 = timing ? solve_clock() : 0ull;
#pragma unroll
    for (int q = 0; q < 4; ++q) Dt[(ak + 4 * q) * kBsvLd + am] = d[q];
    int fail = 0;
    double* X = Xs[k];
    diag_tile_factor<double, false>(Dt, kBsvLd, lane, fail, [&](int r, int c, double v) { X[r * kBsvLd + c] = v; });
    if (__any(fail) && lane == 0) failed = 1;
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) t += X[am * kBsvLd + 4 * ak + c] * yv[16 * k + 4 * ak + c];
    t += __shfl_xor(t, 16);
    t += __shfl_xor(t, 32);
    if (ak == 0) yv[16 * k + am] = t;
    if (timing && lane == 0) p.dbg_cycles[(long long)w * 16 + 5] += solve_clock() - tf0;
  };
  SLS_SOLVE_STAMP(0);

  // (k = -1: only the first diagonal tile is factored - one copy of that code in the kernel, which is larger than the
  // instruction cache as it is)
  int deferred = -1;
  for (int k = -1; k < NB; ++k) {
    // ---- panel below the diagonal tile: L_ik = A_ik X_kk^T; forward substitution b_i -= L_ik y_k
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (bj[s] != k || bi[s] <= k) continue;
      int ib = bi[s];
      asm volatile("" : "+s"(ib));                       // (opaque: the slot's LDS addresses are formed here, not hoisted out of the k loop for all slots at once)
      double* P = Pan[k & 1][ib];
      const double* X = Xs[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) P[(ak + 4 * q) * kBsvLd + am] = acc[s][q];
      v4f64 l = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        l = __builtin_amdgcn_mfma_f64_16x16x4f64(P[am * kBsvLd + 4 * s4 + ak], X[am * kBsvLd + 4 * s4 + ak], l, 0, 0, 0);   // B[k][n] = X[n][k]
      acc[s] = l;
      const double yk = yv[16 * k + am];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        P[(ak + 4 * q) * kBsvLd + am] = l[q];
        double t = l[q] * yk;                            // row ak + 4 q of L_ik times y_k: sum over the 16 lanes of the row
        t += dpp_move<0xB1>(t); t += dpp_move<0x4E>(t); t += dpp_move<0x141>(t); t += dpp_move<0x140>(t);   // (DPP: no LDS round trips)
        if (am == 0) yv[16 * ib + ak + 4 * q] -= t;
      }
    }
    if (k >= 0) __syncthreads();
    SLS_SOLVE_STAMP(2);
    // ---- trailing update A_ij -= L_ik L_jk^T.  Look-ahead: the wave that holds the next diagonal tile updates it first and
    // factors it while the other waves work through their blocks - the 16 sequential pivots of a tile are the long pole of a
    // step - and of its other blocks it only brings the next panel's up to date: the rest of this step's updates wait for the
    // next phase (the panel is double-buffered), when another wave is the one factoring.
    auto update = [&](v4f64& c, const int kk, int ib, int jb) {
      asm volatile("" : "+s"(ib), "+s"(jb));
      const double* Pi = Pan[kk & 1][ib];
      const double* Pj = Pan[kk & 1][jb];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pi[am * kBsvLd + 4 * s4 + ak], Pj[am * kBsvLd + 4 * s4 + ak], c, 0, 0, 0);
    };
    const int bd = ((k + 1) * (k + 2)) / 2 + k + 1;
    const int pend = deferred;                                   // -1, or k - 1: that step is still missing on this wave's blocks with j > k
    const bool factoring = k + 1 < NB && wave == (bd & (kBsvWaves - 1));
    if (factoring) {
      v4f64 d = acc[0];
#pragma unroll
      for (int s = 0; s < SLOTS; ++s)
        if (s == bd / kBsvWaves) {
          if (pend >= 0) update(acc[s], pend, k + 1, k + 1);
          if (k >= 0) update(acc[s], k, k + 1, k + 1);
          d = acc[s];
        }
      factor_diag(k + 1, d);
    }
    // the blocks of this wave: a pending older step first (its panel buffer is the one the next panel is written to), then this
    // step - which the factoring wave only applies to the next panel's blocks (j = k + 1) and leaves pending on the others
    const int jhi = factoring ? k + 1 : NB;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (k < 0 || bj[s] <= k || bi[s] < 0 || (bi[s] == k + 1 && bj[s] == k + 1)) continue;
      if (pend >= 0) update(acc[s], pend, bi[s], bj[s]);
      if (bj[s] <= jhi) update(acc[s], k, bi[s], bj[s]);
    }
    deferred = (factoring && k >= 0) ? k : -1;
    __syncthreads();
    SLS_SOLVE_STAMP(3);
    // (factor_diag(k + 1) wrote Dt, Xs[k + 1], y_(k+1): nothing the trailing update reads; the next panel's stores go to the
    // buffer of step k - 1, whose pending updates every wave has applied by now)
  }
  // ---- backward substitution
  for (int i = NB - 1; i >= 0; --i) {
    __syncthreads();
    if (wave == ((((i * (i + 1)) / 2) + i) & (kBsvWaves - 1))) {
      const double* X = Xs[i];
      double t = 0.0;                                    // x_i = X_ii^T y_i: lane (column am, rows 4 ak ..)
#pragma unroll
      for (int r = 0; r < 4; ++r) t += X[(4 * ak + r) * kBsvLd + am] * yv[16 * i + 4 * ak + r];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (ak == 0) yv[16 * i + am] = t;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (bi[s] != i || bj[s] >= i) continue;
      int jb = bj[s];
      asm volatile("" : "+s"(jb));
      double t = 0.0;                                    // y_k -= L_ik^T x_i: column am of the block, rows ak + 4 q
#pragma unroll
      for (int q = 0; q < 4; ++q) t += acc[s][q] * yv[16 * i + ak + 4 * q];
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (ak == 0) yv[16 * jb + am] -= t;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  SLS_SOLVE_STAMP(4);
  if (tid < n) yvec[tid] = yv[tid];
  __syncthreads();
  // ---- what k_big_finish and k_big_cameras(candidate) do on the launch-chain path: step statistics of the camera block,
  // candidate poses, their rotation / Jacobian table
  if (wave == 0) big_finish(p, bg, w, lane, failed);
  __syncthreads();
  const int cand = 1 - st->cur;
  for (int c = tid; c < wd.C; c += 64 * kBsvWaves) big_camera_entry(p, bg, wd.cam_off + c, cand, 1);
}

}  // namespace slslam
#endif  // SLSLAM_LBA_BIG_SOLVE_H_
