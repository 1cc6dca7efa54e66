// slslam_amd/csrc/dense_tile.h - Cholesky of one 16 x 16 diagonal tile in the registers of a wave (gfx950).
//
// Shared by the reduced camera solve of the line bundle adjustment (k_reduced_solve, lba_kernels.h) and the blocked
// Cholesky of the pose-graph path (k_po_potrf_diag, po_kernels.h): both replace the dense factorisations Ceres' linear
// solvers do inside ceres::Solve (reference src/lba_problem.cpp:99-110, src/po_problem.cpp:79-85 choose them).
//
// Lane r of every 16-lane row of the wave holds ROW r of the symmetric tile (a[16]; the four rows of the wave carry the
// same values) and four columns of row r of the identity that the same row operations turn into L^-1 (row group
// g = lane >> 4 owns columns 4g .. 4g+3).  Elimination form of the right-looking Cholesky, 16 steps: the pivot row reaches
// the lanes through DPP row_newbcast (VALU moves: no LDS round trip on the dependent chain), row r > j takes
//   a_r -= (a_r[j] / piv) a_j,   e_r -= (a_r[j] / piv) e_j        (one fma per element),
// column j becomes L[., j] = a[j] / sqrt(piv); the rows of L^-1 get their 1 / L[r][r] (returned in ipown) at the end.
#ifndef SLSLAM_DENSE_TILE_H_
#define SLSLAM_DENSE_TILE_H_

#include <hip/hip_runtime.h>

namespace slslam {

// lane J of every 16-lane row -> all lanes of the row.  fp64: ONE v_mov_b64_dpp (row_newbcast is the control the DP ALU's DPP
// accepts); the clang builtin only takes 32-bit values, the LLVM intrinsic is reached by its name.
extern "C" __device__ double slslam_update_dpp_f64(double old, double src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
    __asm("llvm.amdgcn.update.dpp.f64");
template <int J>
__device__ __forceinline__ double tile_bcast(double v) {
  return slslam_update_dpp_f64(0.0, v, 0x150 + J, 0xF, 0xF, true);
}
template <int J>
__device__ __forceinline__ float tile_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + J, 0xF, 0xF, true));
}
// 1 / sqrt(x) of a positive, finite x (the caller has tested the pivot): v_rsq_f64 and the one refinement step the device
// library's rsqrt() applies - the same values - without its handling of 0 / inf / nan, which sits on the dependent chain of the
// sixteen pivots.  No sqrt, no divide on that chain.
__device__ __forceinline__ double tile_rsqrt(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y0, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}
__device__ __forceinline__ float tile_rsqrt(float x) { return rsqrtf(x); }

template <int JC, typename T>
__device__ __forceinline__ void diag_tile_steps(T (&a)[16], T (&e)[4], T& ipown, int r, int& fail) {
  if constexpr (JC < 16) {
    const T piv = tile_bcast<JC>(a[JC]);
    const bool okp = piv > T(0) && isfinite(piv);
    if (!okp) fail = 1;
    const T ip = tile_rsqrt(okp ? piv : T(1));
    const T below = (r > JC) ? a[JC] : T(0);                             // (off the pivot chain: ready before ip is)
    const T nm = below * (ip * -ip);                                     // -a[JC] / piv for the rows below the pivot, 0 elsewhere
    a[JC] *= ip;                                                         // L[r][JC] for r >= JC
    ipown = (r == JC) ? ip : ipown;
#pragma unroll
    for (int c = JC + 1; c < 16; ++c) a[c] = fma(nm, tile_bcast<JC>(a[c]), a[c]);
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = fma(nm, tile_bcast<JC>(e[k]), e[k]);
    diag_tile_steps<JC + 1, T>(a, e, ipown, r, fail);
  }
}

// The whole tile: D points at its (0, 0) element in LDS (leading dimension ld, LOWER triangle valid on entry).  On exit the
// lower triangle holds L (STORE_L; a caller that only needs the inverse skips it); the inverse goes where `put_inverse(row, col, value)` sends it (col <= row).  One full wave.
template <typename T, bool STORE_L = true, typename PutInv>
__device__ __forceinline__ void diag_tile_factor(T* D, int ld, int lane, int& fail, PutInv put_inverse) {
  const int r = lane & 15, g = lane >> 4;
  T a[16], e[4], ipown = T(1);
#pragma unroll
  for (int c = 0; c < 16; ++c) a[c] = (r >= c) ? D[r * ld + c] : D[c * ld + r];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = (r == 4 * g + k) ? T(1) : T(0);
  diag_tile_steps<0, T>(a, e, ipown, r, fail);
  // (a[] is indexed with compile-time constants only: one masked store per column - a select chain over the row group is
  // turned into an indexed load from a stack copy of a[] by the compiler)
  if (STORE_L) {
#pragma unroll
    for (int c = 0; c < 16; ++c)
      if (g == (c >> 2) && c <= r) D[r * ld + c] = a[c];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * g + k;
    if (c <= r) put_inverse(r, c, e[k] * ipown);
  }
}

}  // namespace slslam
#endif
