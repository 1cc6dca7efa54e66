// slslam_amd/csrc/lba_kernels.h — hand-written CDNA4 (gfx950) kernels of the batched line
// bundle adjustment.  One LM iteration of every window of the batch is four launches:
//
//   k_linearise_schur  one 64-lane wave per chunk of a window's lines; lane <-> observation,
//                      a line owns a run of lanes (segmented DPP scans), per-wave private partial of the
//                      reduced camera system in LDS (ds_add_f64); keeps each line's 4x4 factor
//   k_reduced_solve    one workgroup per window: ordered reduction of the chunk partials, LM damping,
//                      blocked in-LDS Cholesky of the (6 Cf)^2 system on v_mfma_f64_16x16x4_f64,
//                      candidate camera poses
//   k_backsub          second sweep over the observations: back-substitutes every line with the kept factor,
//                      writes the candidate line parameters (+ their sin/cos table) and the step statistics
//                      and, with the observation still in registers, the cost at the candidate point
//   k_lm_update        per window: gain ratio, accept/reject, radius update, convergence tests
//
// What this replaces: everything ceres::Solve does for the problem LBAProblem::build wires up
// (reference src/lba_problem.cpp:54-132, call sites src/slam.cpp:663,944).  Ceres evaluates
// Jet<double,10> functors block by block, forms the sparse normal equations and factors them with
// CHOLMOD on one thread; here the block structure (4x4 line blocks, 6x4 off-diagonal blocks,
// 6x6 camera blocks) is exploited directly; Jacobians are recomputed per sweep instead of stored, only the
// per-line 4x4 factors (144 B per line) travel from the first sweep to the second.
#ifndef SLSLAM_LBA_KERNELS_H_
#define SLSLAM_LBA_KERNELS_H_

#include <hip/hip_runtime.h>
#include "lba_math.h"
#include "lba_types.h"
#include "dense_tile.h"
#include "lba_eliminate_mfma_maps.h"

// ISA audit (tools/isa_audit.py): the SLSLAM_ISA_MARKERS build leaves '; @PHASE name' comments in the assembly and pins the
// schedule at each of them, so that the instructions of the tile loop can be counted per phase.  Not a product build.
#if defined(SLSLAM_ISA_MARKERS)
#define SLS_PHASE(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; @PHASE " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SLS_PHASE(name) do { } while (0)
#endif

namespace slslam {

enum { kCamTab = 21 };   // doubles per camera in LDS: R[9] JL[9] t[3]; odd stride (42 dwords): conflict-free
                         // ds_read_b64 across cameras.  The Jacobi scale (6 per FREE camera) is a second table.

// The wave's partial of the reduced camera system, in LDS and (verbatim) in its HBM slab.  Laid out by
// BLOCK, with odd strides in 8-byte units so that the ds_add_f64 of lanes working on different
// cameras / camera pairs fall on different banks (a packed lower triangle puts the 45 pair blocks of
// a 10-camera window on only 8 distinct bank offsets):
//   camera record cf  (kCamAcc = 39):  diagonal block, lower triangle [21] | b[6] | g[6] | hdiag[6]
//   pair block (cj > ci) (kPairAcc = 37): 6x6, row a of cj, column b of ci [36] | pad
// 20 cameras / 10 free: 3360 + 480 + 16440 + 24 = 20 304 B -> 8 workgroups per CU.
enum { kCamAcc = 39, kPairAcc = 37, kRecB = 21, kRecG = 27, kRecH = 33 };
enum { kMfmaTiles = kPTiles, kMfmaRec = kDiagRec };   // slab of the matrix-core elimination: 10 accumulator tiles | camera records
__host__ __device__ inline int sys_doubles(int n) { const int cf = n / 6; return cf * kCamAcc + ((cf * (cf - 1)) / 2) * kPairAcc; }
__device__ __forceinline__ int pair_base(int ncf, int cj, int ci) { return ncf * kCamAcc + ((cj * (cj - 1)) / 2 + ci) * kPairAcc; }

__device__ __forceinline__ int tri_index(int r, int c) { return (r * (r + 1)) / 2 + c; }  // r >= c

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);   // every lane has a valid source: no `old` operand needed
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// row_shr: lane i reads lane i - d of its 16-lane row; lanes whose source is outside the row read 0
template <int CTRL>
__device__ __forceinline__ double dpp_shift0(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm64(double v, int byte_addr) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// Where a lane sits in its line's run of lanes (Tile / lane_map in lba_types.h).
struct SegCtx {
  int j;            // position in the run (0 for idle lanes)
  int first4;       // byte address (lane * 4) of the run's first lane
  int rl4;          // byte address of the last lane of the run inside this lane's row
  int r0, r1;       // first and last row of the run
  int kk;           // min(run length, 4): lanes of the run that evaluate sin/cos
  int max_run;      // longest in-row run of the tile (wave-uniform)
  int rounds;       // sin/cos rounds of the tile (wave-uniform): 1 when every run has >= 4 lanes
  bool multirow;    // some run of the tile spans several rows (wave-uniform)
};

// Sum over the lanes of a line's run, result in every lane of the run (bitwise identical across the run).
// Inclusive prefix along the 16-lane row restricted to the lane's own run (row_shr by 1, 2, 4, 8 on the VALU; the
// source is in the same run iff j >= d), then every lane fetches the total from the run's last lane in the row.
// Runs that span rows (lines with more than 16 observations, rare) add their row totals in row order.
// N values at once, step-major: one wave-uniform branch per step for the whole batch.
// FMA_MASK: the mask rides on an fma (x += t * 1.0 or t * 0.0) instead of two v_cndmask per value.  A non-finite t
// from a neighbouring line would leak through 0 * t, but a non-finite block already invalidates the whole window's
// step, so the outcome is the same.  (Not used by the elimination kernel: there the fma form costs the few registers
// that separate 2 waves per SIMD from 1.)
// The non-fma form masks the shifted value with an integer AND on both halves (m ? ~0 : 0): the AND takes the DPP
// modifier itself (v_and_b32_dpp), so a masked step of one value is two VALU instructions + the add instead of two moves +
// two v_cndmask + the add, and a masked-out term is exactly +0.0 as before.
template <int CTRL>
__device__ __forceinline__ double dpp_shift0_and(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true) & mask;
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true) & mask;
  return __hiloint2double(hi, lo);
}
template <int CTRL, bool FMA_MASK>
__device__ __forceinline__ double seg_step(double v, bool m) {
  if (FMA_MASK) return fma(dpp_shift0<CTRL>(v), m ? 1.0 : 0.0, v);
  int mask = m ? -1 : 0;
  asm("" : "+v"(mask));            // opaque: keeps the AND an AND (the optimiser would turn it back into selects)
  return v + dpp_shift0_and<CTRL>(v, mask);
}
// row_shl: lane i reads lane i + d of its 16-lane row.  The run's total sits in its last lane (of the row) after the prefix scan;
// it travels back down the run in 1, 2, 4, 8-lane hops on the VALU (a lane takes the hop iff its source is still inside the run)
// instead of one ds_bpermute round trip per value: the sweeps are bound by their LDS instructions, not by VALU ones.
// One v_cndmask_b32 with the DPP modifier per register, in place: vcc = (room < d) keeps the lane's own value.  (Inline
// assembly: the compiler does not form the DPP select; s_nop 1 covers the VALU-write -> DPP-read wait states.)
#define SLS_BACK_ASM(SHL)                                                                                                     \
  asm volatile("s_nop 1\n\tv_cmp_gt_i32 vcc, " #SHL ", %28\n\t"                                                               \
               "v_cndmask_b32_dpp %0, %0, %0, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %1, %1, %1, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %2, %2, %2, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %3, %3, %3, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %4, %4, %4, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %5, %5, %5, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %6, %6, %6, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %7, %7, %7, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %8, %8, %8, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %9, %9, %9, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"               \
               "v_cndmask_b32_dpp %10, %10, %10, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %11, %11, %11, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %12, %12, %12, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %13, %13, %13, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %14, %14, %14, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %15, %15, %15, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %16, %16, %16, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %17, %17, %17, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %18, %18, %18, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %19, %19, %19, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %20, %20, %20, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %21, %21, %21, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %22, %22, %22, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %23, %23, %23, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %24, %24, %24, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %25, %25, %25, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %26, %26, %26, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"            \
               "v_cndmask_b32_dpp %27, %27, %27, vcc row_shl:" #SHL " row_mask:0xf bank_mask:0xf bound_ctrl:1"                   \
               : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "+v"(w[8]), "+v"(w[9]),   \
                 "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]), "+v"(w[16]), "+v"(w[17]), "+v"(w[18]),      \
                 "+v"(w[19]), "+v"(w[20]), "+v"(w[21]), "+v"(w[22]), "+v"(w[23]), "+v"(w[24]), "+v"(w[25]), "+v"(w[26]), "+v"(w[27])       \
               : "v"(room) : "vcc")
// the run totals of 14 values (28 dwords) brought back from the run's last lane to every lane of the run
__device__ __forceinline__ void seg_back_14(double (&v)[14], int room, int max_run) {
  int w[28];
#pragma unroll
  for (int q = 0; q < 14; ++q) { w[2 * q] = __double2loint(v[q]); w[2 * q + 1] = __double2hiint(v[q]); }
  if (max_run > 1) SLS_BACK_ASM(1);
  if (max_run > 2) SLS_BACK_ASM(2);
  if (max_run > 4) SLS_BACK_ASM(4);
  if (max_run > 8) SLS_BACK_ASM(8);
#pragma unroll
  for (int q = 0; q < 14; ++q) v[q] = __hiloint2double(w[2 * q + 1], w[2 * q]);
}
template <int N, bool FMA_MASK = false, bool BACK_DPP = false>
__device__ __forceinline__ void seg_sum_n(double (&v)[N], const SegCtx& s) {
  SLS_PHASE("seg_scan_1_2_4");
  if (s.max_run > 1) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = seg_step<0x111, FMA_MASK>(v[q], s.j >= 1);
  }
  if (s.max_run > 2) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = seg_step<0x112, FMA_MASK>(v[q], s.j >= 2);
  }
  if (s.max_run > 4) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = seg_step<0x114, FMA_MASK>(v[q], s.j >= 4);
  }
  SLS_PHASE("seg_scan_8");
  if (s.max_run > 8) {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = seg_step<0x118, FMA_MASK>(v[q], s.j >= 8);
  }
  SLS_PHASE("seg_total");
  if (BACK_DPP && N == 14) {
    seg_back_14(reinterpret_cast<double (&)[14]>(v), (s.rl4 >> 2) - (int)(threadIdx.x & 63), s.max_run);
  } else {
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = bperm64(v[q], s.rl4);
  }
  SLS_PHASE("seg_multirow");
  if (s.multirow) {
    const bool spans = s.r1 > s.r0;
#pragma unroll
    for (int q = 0; q < N; ++q) {
      double tot = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double x = bperm64(v[q], 64 * r);      // lane 16 r belongs to the run whenever the run covers row r
        if (r >= s.r0 && r <= s.r1) tot += x;
      }
      if (spans) v[q] = tot;
    }
  }
}
// 1 / x for the gradient-norm test (x = Jacobi scale in (0, 1]): hardware estimate + two Newton steps (within ~1 ulp) instead of the
// ~13-instruction IEEE division; the norm only meets a tolerance and the trace.
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// Timing experiments of the elimination sweep (SLSLAM_EXTRA_FLAGS=-DSLSLAM_ABLATE=<bits>, tools/gpu_variant_bench.sh; results are
// WRONG when set): 1 pair-block atomics dropped (products still computed), 4 camera-record atomics dropped, 8 pair-block
// atomics replaced by plain stores, 16 camera-record atomics replaced by stores, 64 back-substitution: every lane reads the first
// line's factor record (are those loads waited for?).
#if !defined(SLSLAM_ABLATE)
#define SLSLAM_ABLATE 0
#endif
__device__ __forceinline__ void keep_alive(double v) { asm volatile("" : : "v"(v)); }
__device__ __forceinline__ void lds_add_pair(double* p, double v) {
  if (SLSLAM_ABLATE & 1) keep_alive(v);
  else if (SLSLAM_ABLATE & 8) *(volatile double*)p = v;
  else lds_add(p, v);
}
__device__ __forceinline__ void lds_add_rec(double* p, double v) {
  if (SLSLAM_ABLATE & 4) keep_alive(v);
  else if (SLSLAM_ABLATE & 16) *(volatile double*)p = v;
  else lds_add(p, v);
}

// ------------------------------------------------------------------------------------------
// Everything a lane knows about its observation and its line after linearisation.
struct LaneLin {
  double rs[4];      // robustified residual
  double Jc[24];     // robustified, Jacobi-scaled camera Jacobian (row-major 4x6)
  double Jl[16];     // robustified, Jacobi-scaled line Jacobian (row-major 4x4)
  double cost;       // rho/2 of this block
  int cam;           // window-local camera id
  int cf;            // free-camera index or -1
  bool valid, kept, line_free;
};

// What a lane needs to know before it can touch its observation: fetched one tile AHEAD so that
// the dependent chain  tile descriptor -> line_ptr -> observation  is off the critical path.
struct TileCtx {
  int flags, ls, j, o0, k, lflags, nitems, item_off;
  bool skew;                // lane_map bit 15: this lane adds its camera-record entries one step late (see the diagonal block)
  int slot, nlines;         // the lane's line slot in the tile (matrix-core sweep), lines of the tile (wave-uniform)
  bool line_ok;
  unsigned desc;            // the tile's lane-th line descriptor (matrix-core sweeps; 0 past the tile's lines)
};
// One 16-byte load per lane (BatchPtrs.lane_ctx, resolved by the host: sorted line | first observation | j, k, flags | descriptor) and the
// tile record: nothing here depends on a loaded value, and nothing is decoded here - request_tile only issues the loads, resolve_tile
// (called where the next tile's observations are requested, most of a tile later) waits for them.  The chain
// tile -> lane map -> line pointer this replaces cost every tile of every sweep three exposed round trips to memory
// (s_waitcnt vmcnt(0) between dependent loads in the middle of the tile loop).
struct TileReq { int4 r; int2 items; int nl; bool live; };
__device__ __forceinline__ TileReq request_tile(const BatchPtrs& p, int t, int t_end, int lane) {
  // (no branch around the loads: the copies that merge a conditional load with its default would be uses, i.e. waits; past the end of
  // the chunk the previous tile is read again and resolve_tile discards it)
  TileReq q;
  q.live = t < t_end;
  const int tt = q.live ? t : (t_end > 0 ? t_end - 1 : 0);
  q.r = reinterpret_cast<const int4*>(p.lane_ctx)[(long long)tt * 64 + lane];
  const int* tl = reinterpret_cast<const int*>(p.tiles + tt);    // Tile: line_begin | nlines, flags | item_off | nitems (separate loads: a sweep
  q.nl = tl[1];                                                  // that does not use a part does not carry it)
  q.items = *reinterpret_cast<const int2*>(tl + 2);
#if defined(SLS_BLOCKING_TILE_FETCH)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // timing control: ONE exposed round trip per tile (the old chain had three)
#endif
  return q;
}
__device__ __forceinline__ TileCtx resolve_tile(const TileReq& q) {
  TileCtx c;
  int rx = q.r.x, ry = q.r.y, rz = q.r.z, rw = q.r.w;
  asm volatile("" : "+v"(rx), "+v"(ry), "+v"(rz), "+v"(rw));      // the decode (and with it the wait for the load) stays HERE
  if (!q.live) { rx = 0; ry = 0; rz = (int)(0xffu << 24 | 1u << 15); rw = 0; }      // an idle lane: no line, "constant"
  const unsigned m = (unsigned)rz;
  c.ls = rx; c.o0 = ry; c.desc = (unsigned)rw;
  c.j = (int)(m & 0x3fu); c.k = (int)((m >> 6) & 0x7fu); c.line_ok = ((m >> 13) & 1u) != 0u; c.skew = ((m >> 14) & 1u) != 0u;
  c.lflags = (int)((m >> 15) & 1u); c.slot = (int)(m >> 24);
  c.flags = q.live ? __builtin_amdgcn_readfirstlane((int)((m >> 16) & 0xffu)) : 0;          // (wave-uniform values to the scalar registers)
  c.nlines = q.live ? __builtin_amdgcn_readfirstlane((int)(short)(q.nl & 0xffff)) : 0;
  c.item_off = q.live ? __builtin_amdgcn_readfirstlane(q.items.x) : 0; c.nitems = q.live ? __builtin_amdgcn_readfirstlane(q.items.y) : 0;
  return c;
}
__device__ __forceinline__ TileCtx fetch_tile(const BatchPtrs& p, int t, int t_end, int lane) {
  return resolve_tile(request_tile(p, t, t_end, lane));
}
__device__ __forceinline__ SegCtx make_seg(const TileCtx& c, int lane) {
  SegCtx s;
  const int run = c.k > 1 ? c.k : 1;
  const int first = lane - c.j, last = first + run - 1;
  s.j = c.j;
  s.first4 = first * 4;
  s.rl4 = (last < (lane | 15) ? last : (lane | 15)) * 4;
  s.r0 = first >> 4; s.r1 = last >> 4;
  s.kk = run < 4 ? run : 4;
  s.max_run = tile_max_run(c.flags);
  s.rounds = tile_trig_rounds(c.flags);
  s.multirow = (c.flags & kTileMultiRow) != 0;
  return s;
}

// What a lane reads from HBM for its observation: the four endpoint pairs, the camera id and the sin/cos table
// of its line.  Requested for tile t + 1 in the low-register-pressure tail of tile t (after the Jacobians are
// dead), so that the HBM latency of the next tile overlaps the pair products of the current one.
struct ObsPref {
  double ob[8];
  double trig[7];
  double lsc[4];      // Jacobi scale of the line's columns
  double u[4];        // the line's parameters (back-substitution only)
  int cam;
  int items01;        // elimination only: this lane's work items of the first two pair passes, (li, lj) bytes x 2
};
template <bool WITH_U, bool WITH_ITEMS = false>
__device__ __forceinline__ void prefetch_obs(const BatchPtrs& p, const TileCtx& c, int cur, int safe_obs, ObsPref& f, int lane = 0) {
  const bool valid = c.line_ok && c.j < c.k;
#if defined(SLS_ABLATE_OBS_CACHED)
  const int o = safe_obs + (c.j & 7);                 // timing experiment (results WRONG): every tile reads the same few observations and lines - cache hits
  const int lsafe = c.j & 7;
#else
  const int o = valid ? c.o0 + c.j : safe_obs;
  const int lsafe = c.line_ok ? c.ls : 0;
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) {     // (x,y) endpoint pairs: one 16-byte load per plane
    const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
    f.ob[2 * q] = e.x; f.ob[2 * q + 1] = e.y;
  }
  f.cam = p.ob_cam[o];
  const double* lrec = p.line_x + line_rec(p, lsafe, cur);
#pragma unroll
  for (int q = 0; q < 7; ++q) f.trig[q] = lrec[4 + q];
  const double* ls = p.line_scale + (long long)lsafe * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) f.lsc[q] = ls[q];
  if (WITH_U) {
#pragma unroll
    for (int q = 0; q < 4; ++q) f.u[q] = lrec[q];
  }
  if (WITH_ITEMS) {
    const unsigned short* it16 = reinterpret_cast<const unsigned short*>(p.items);     // one item = 2 bytes (li, lj)
    const int i0 = lane < c.nitems ? c.item_off + lane : 0, i1 = 64 + lane < c.nitems ? c.item_off + 64 + lane : 0;
    f.items01 = (int)it16[i0] | ((int)it16[i1] << 16);
  }
}

// Load one observation + its line record and linearise it.  cur selects the parameter buffer.
// SCALED: apply the Jacobi column scaling (false for the initial evaluation and the test hook).
template <bool SCALED>
__device__ __forceinline__ void lane_linearise(const BatchPtrs& p, const Policy& pol, const double* camtab,
                                               const double* camscale, const signed char* camcf, int ls, int j, int k, int o0, bool line_ok,
                                               int lflags, int cur, int safe_obs, LaneLin& L, double (&ob)[8],
                                               const ObsPref* pf = nullptr, bool unit_line = false) {
  L.valid = line_ok && j < k;
  const int lsafe = line_ok ? ls : 0;
  double trig[7];
  if (pf) {
#pragma unroll
    for (int q = 0; q < 8; ++q) ob[q] = pf->ob[q];
#pragma unroll
    for (int q = 0; q < 7; ++q) trig[q] = pf->trig[q];
    L.cam = pf->cam;
  } else {
    const int o = L.valid ? o0 + j : safe_obs;
#pragma unroll
    for (int q = 0; q < 4; ++q) {     // (x,y) endpoint pairs: one 16-byte load per plane
      const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
      ob[2 * q] = e.x; ob[2 * q + 1] = e.y;
    }
    L.cam = p.ob_cam[o];
    const double* lrec = p.line_x + line_rec(p, lsafe, cur);
#pragma unroll
    for (int q = 0; q < 7; ++q) trig[q] = lrec[4 + q];
  }
  L.line_free = line_ok && !(lflags & 1);
  const double* ct = camtab + L.cam * kCamTab;
  double R[9], JL[9], t[3];
#pragma unroll
  for (int q = 0; q < 9; ++q) { R[q] = ct[q]; JL[q] = ct[9 + q]; }
#pragma unroll
  for (int q = 0; q < 3; ++q) t[q] = ct[18 + q];
  L.cf = camcf[L.cam];
  L.kept = L.valid && !(L.cf < 0 && !L.line_free);
  const double* cs = camscale + (L.cf >= 0 ? L.cf : 0) * 6;
  double cp[3], dv[3], dcp[12], ddv[9], r[4];
  line_points_jac<double>(trig, cp, dv, dcp, ddv);
  obs_linearise<double>(R, JL, t, cp, dv, dcp, ddv, ob, pol.baseline, r, L.Jc, L.Jl);
  const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
  const double sr = huber_scale<double>(s, pol.huber_delta, &L.cost);
#pragma unroll
  for (int q = 0; q < 4; ++q) L.rs[q] = r[q] * sr;
  if (SCALED) {
    const double* lsc = pf ? pf->lsc : p.line_scale + (long long)lsafe * 4;
    double sc[6], sl[4];
#pragma unroll
    for (int a = 0; a < 6; ++a) sc[a] = cs[a] * sr;
#pragma unroll
    for (int a = 0; a < 4; ++a) sl[a] = unit_line ? sr : lsc[a] * sr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int a = 0; a < 6; ++a) L.Jc[6 * q + a] *= sc[a];
#pragma unroll
      for (int a = 0; a < 4; ++a) L.Jl[4 * q + a] *= sl[a];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 24; ++q) L.Jc[q] *= sr;
#pragma unroll
    for (int q = 0; q < 16; ++q) L.Jl[q] *= sr;
  }
}

// Back-substitution form of lane_linearise: the residual (for the robust weight) and this observation's term of
// w = sum_i H_cl,i^T y_c, i.e. J_l^T (J_c y_c) with the robustified, Jacobi-scaled Jacobians - contracted on the fly
// (obs_backsub_w), neither Jacobian is formed.  The camera table of the back-substitution holds
// R[9] t[3] | JL (s_w o y_w) [3] | s_t o y_t [3] per camera (kBsTab doubles).
enum { kBsTab = 19 };
struct LaneBs {
  double cost;       // rho/2 of this block at the accepted point (unused by the caller today)
  int cam, cf;
  bool valid, kept, line_free;
};
__device__ __forceinline__ void lane_linearise_bs(const Policy& pol, const double* bstab, const signed char* camcf,
                                                  int j, int k, bool line_ok, int lflags, LaneBs& L, double (&ob)[8],
                                                  double (&w)[4], const ObsPref& pf) {
  L.valid = line_ok && j < k;
  double trig[7];
#pragma unroll
  for (int q = 0; q < 8; ++q) ob[q] = pf.ob[q];
#pragma unroll
  for (int q = 0; q < 7; ++q) trig[q] = pf.trig[q];
  L.cam = pf.cam;
  L.line_free = line_ok && !(lflags & 1);
  const double* ct = bstab + L.cam * kBsTab;
  double R[9], t[3], vw[3], yt[3];
#pragma unroll
  for (int q = 0; q < 9; ++q) R[q] = ct[q];
#pragma unroll
  for (int q = 0; q < 3; ++q) { t[q] = ct[9 + q]; vw[q] = ct[12 + q]; yt[q] = ct[15 + q]; }
  L.cf = camcf[L.cam];
  L.kept = L.valid && !(L.cf < 0 && !L.line_free);
  double cp[3], dv[3], dcp[12], ddv[9], r[4];
  line_points_jac<double>(trig, cp, dv, dcp, ddv);
  obs_backsub_w<double>(R, t, vw, yt, cp, dv, dcp, ddv, ob, pol.baseline, r, w);
  const double s = r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3];
  const double sr = huber_scale<double>(s, pol.huber_delta, &L.cost);
  const double sr2 = sr * sr;                    // both Jacobians carry the factor sqrt(rho')
#pragma unroll
  for (int a = 0; a < 4; ++a) w[a] *= pf.lsc[a] * sr2;
}

// Per-line normal-equation block, summed over the line's run of lanes (every lane of the run
// ends with the same values): H = sum Jl^T Jl (lower triangle, 10 values), g = sum Jl^T r.
// No mask on the lanes' terms: a run never reads outside itself (the scan steps are masked by the position in the run), every
// lane of a free line with observations carries one of them, and the sums of the other runs (idle lanes, lines without
// observations, constant lines) are never used - line_active guards every consumer.
__device__ __forceinline__ void line_block(const LaneLin& L, const SegCtx& sg, double H[10], double g[4]) {
  double v[14];
  int q = 0;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      double h = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) h += L.Jl[4 * r + a] * L.Jl[4 * r + b];
      v[q++] = h;
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double ga = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) ga += L.Jl[4 * r + a] * L.rs[r];
    v[10 + a] = ga;
  }
#if defined(SLSLAM_SEG_TOTAL_BPERMUTE)        // round-2 form, for comparison: 28 ds_bpermute per tile
  seg_sum_n<14>(v, sg);
#else
  seg_sum_n<14, false, true>(v, sg);
#endif
#pragma unroll
  for (int i = 0; i < 10; ++i) H[i] = v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) g[i] = v[10 + i];
}

// A = H + diag(D2), A = Lc Lc^T, K = Lc^-1 (lower, 10 values, same packing as H).
// Returns false if A is not positive definite (only possible with non-finite input).
__device__ __forceinline__ bool chol4_inverse(const double H[10], const double D2[4], double K[10]) {
  // packing: (0,0)=0 (1,0)=1 (1,1)=2 (2,0)=3 (2,1)=4 (2,2)=5 (3,0)=6 (3,1)=7 (3,2)=8 (3,3)=9
  const double a00 = H[0] + D2[0], a10 = H[1], a11 = H[2] + D2[1], a20 = H[3], a21 = H[4],
               a22 = H[5] + D2[2], a30 = H[6], a31 = H[7], a32 = H[8], a33 = H[9] + D2[3];
  bool ok = a00 > 0.0;
  const double i0 = inv_sqrt<double>(a00);
  const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
  const double d1 = a11 - l10 * l10;
  ok = ok && d1 > 0.0;
  const double i1 = inv_sqrt<double>(d1);
  const double l21 = (a21 - l20 * l10) * i1, l31 = (a31 - l30 * l10) * i1;
  const double d2 = a22 - l20 * l20 - l21 * l21;
  ok = ok && d2 > 0.0;
  const double i2 = inv_sqrt<double>(d2);
  const double l32 = (a32 - l30 * l20 - l31 * l21) * i2;
  const double d3 = a33 - l30 * l30 - l31 * l31 - l32 * l32;
  ok = ok && d3 > 0.0;
  const double i3 = inv_sqrt<double>(d3);
  // inverse of the lower-triangular factor (diagonal of Lc is 1/i_k)
  K[0] = i0; K[2] = i1; K[5] = i2; K[9] = i3;
  K[1] = -l10 * K[0] * i1;
  K[3] = -(l20 * K[0] + l21 * K[1]) * i2;
  K[4] = -l21 * K[2] * i2;
  K[6] = -(l30 * K[0] + l31 * K[1] + l32 * K[3]) * i3;
  K[7] = -(l31 * K[2] + l32 * K[4]) * i3;
  K[8] = -l32 * K[5] * i3;
  return ok && isfinite(d3) && isfinite(i3);
}

// inv_radius = 1 / radius, computed once per wave: four fp64 divisions per tile would cost ~50 instructions
__device__ __forceinline__ void lm_diag4(const double H[10], const Policy& pol, double inv_radius, double D2[4]) {
  const double d[4] = { H[0], H[2], H[5], H[9] };
#pragma unroll
  for (int a = 0; a < 4; ++a) D2[a] = fmin(fmax(d[a], pol.min_lm_diagonal), pol.max_lm_diagonal) * inv_radius;
}

// F = (Jc^T Jl) K^T  (6x4, row-major):  the line's share of the elimination, so that
// H_cl A^-1 H_lc = F F^T  and  H_cl A^-1 g_l = F (K g_l).
__device__ __forceinline__ void lane_F(const LaneLin& L, const double K[10], double F[24]) {
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double h[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) s += L.Jc[6 * r + a] * L.Jl[4 * r + b];
      h[b] = s;
    }
    F[4 * a + 0] = h[0] * K[0];
    F[4 * a + 1] = h[0] * K[1] + h[1] * K[2];
    F[4 * a + 2] = h[0] * K[3] + h[1] * K[4] + h[2] * K[5];
    F[4 * a + 3] = h[0] * K[6] + h[1] * K[7] + h[2] * K[8] + h[3] * K[9];
  }
}

// Camera table of one window in LDS.  WITH_JAC: R and JL at buffer `buf`; else R only.
// mode 0: built here (three trig calls per camera).  With BatchPtrs.cam_tab: mode 1 builds it and leaves a copy in memory (the
// first sweep of a solve), mode 2 reads the copy (every later sweep: the reduced solve left the candidate point's table in the
// other buffer, which the accepted point is if the step was taken) - for a window alone on the chip, whose chunks are one tile
// each, building the tables was 12 k of an iteration's 155 k cycles.
template <bool WITH_JAC>
__device__ __forceinline__ void load_cam_table(const BatchPtrs& p, const WinDesc& wd, int buf, int lane,
                                               double* camtab, double* camscale, signed char* camcf, bool unit_scale, int mode = 0) {
  for (int c = lane; c < wd.C; c += 64) {
    double* ct = camtab + c * kCamTab;
    double* gt = mode ? p.cam_tab + ((long long)(wd.cam_off + c) * 2 + buf) * kCamTab : nullptr;
    // (free index and Jacobi scale are requested with the table, not after it: one memory round trip for the lot)
    const int cf = p.cam_cf[wd.cam_off + c];
    double sc6[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) sc6[a] = unit_scale ? 1.0 : p.cam_scale[(long long)(wd.cam_off + c) * 6 + a];
    if (mode == 2) {
      for (int q = 0; q < kCamTab; ++q) ct[q] = gt[q];
    } else {
      const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + buf) * kCamRec;
      double w[3] = { x[0], x[1], x[2] }, R[9], JL[9];
      if (WITH_JAC) cam_prepare<double>(w, R, JL);
      else { cam_rotation<double>(w, R); for (int q = 0; q < 9; ++q) JL[q] = 0.0; }
      for (int q = 0; q < 9; ++q) { ct[q] = R[q]; ct[9 + q] = JL[q]; }
      ct[18] = x[3]; ct[19] = x[4]; ct[20] = x[5];
      if (mode == 1) for (int q = 0; q < kCamTab; ++q) gt[q] = ct[q];
    }
    if (cf >= 0)
      for (int a = 0; a < 6; ++a) camscale[6 * cf + a] = sc6[a];
    camcf[c] = (signed char)cf;
  }
}

__host__ __device__ inline int lds_doubles_linearise(int C, int n) {
  // camera table + scale table (one unused slot when no camera is free) + partial system ; free-index bytes appended
  return C * kCamTab + (n > 0 ? n : 6) + sys_doubles(n) + (C + 7) / 8;
}

// ------------------------------------------------------------------------------------------
// Kernel 1: linearise + per-line Schur elimination, partial reduced system per chunk.
// INIT = initial evaluation (Ceres: cost, gradient and column norms at x0 for the Jacobi scaling):
// no elimination, unit scaling, writes the per-line scale.
// (two waves per SIMD - 256 registers, arch + accumulation VGPRs together - is the occupancy the sweep is tuned for; without the
// bound the allocator parks a few values in AGPRs and the kernel drops to one wave per SIMD)
// FRESH: 1 = the first sweep of a solve (it is also Ceres' initial evaluation: every window of the launch is fresh), 0 = any later
// sweep (none is), -1 = read LMState.fresh.  The host knows which launch is which; the two compile-time forms keep the first
// sweep's extras (unit scales, line Jacobi scale, fixed cost, |x|) out of the steady sweep's registers.
// INVARIANT the host relies on for FRESH = 1 / 0 (enqueue_solve, lba_api.hip): every window that is kRunning at the first launch of a solve() is fresh
// (k_reset / finalize precede it), and none is afterwards - lm_step ends a window at max_num_iterations, so a later solve() on the same state finds
// no kRunning window.  A 'continue solving' entry point would have to launch the FRESH = -1 instantiation (k_reduced_solve, k_backsub and the camera-table
// mode still read LMState.fresh / cur from memory and would disagree with a hard-coded sweep).
// Timing experiment (build with -DSLSLAM_K1_TIMING=1 and run with SLSLAM_DEBUG_ABLATE != 0: tools/k1_phases.py): shader-clock stamps
// of one chunk's sweep, summed over chunks into dbg_cycles[chunk * 32 + i]
#if defined(SLSLAM_K1_TIMING) && SLSLAM_K1_TIMING
#define SLS_K1_STAMP(i) do { unsigned long long now_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
    if (lane == 0 && p.dbg_cycles) p.dbg_cycles[(long long)ck.id * 32 + (i)] += now_ - k1_t_; k1_t_ = now_; } while (0)
#define SLS_K1_STAMP_INIT unsigned long long k1_t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(k1_t_) :: "memory")
#else
#define SLS_K1_STAMP(i) do { } while (0)
#define SLS_K1_STAMP_INIT do { } while (0)
#endif
// -DSLSLAM_K1_WALL=1 (with SLSLAM_DEBUG_ABLATE != 0): constant-clock time (s_memrealtime, 100 MHz) at which a chunk's wave starts
// (word 30) / ends (word 31) its elimination sweep - the timeline of a launch, tools/chunk_timeline.py; no other stamp, so the sweep runs undisturbed
#if defined(SLSLAM_K1_WALL) && SLSLAM_K1_WALL
#define SLS_K1_WALL(slot) do { unsigned long long now_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_) :: "memory"); \
    if (lane == 0 && p.dbg_cycles) { p.dbg_cycles[(long long)ck.id * 32 + (slot)] = now_; \
      /* where the wave ran (word slot - 4: XCC_ID << 32 | HW_ID - wave, SIMD, CU, SH, SE): tools/chunk_classes.py groups the durations by it */ \
      unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw_), "=s"(xcc_)); \
      p.dbg_cycles[(long long)ck.id * 32 + (slot) - 4] = ((unsigned long long)xcc_ << 32) | hw_; } } while (0)
#else
#define SLS_K1_WALL(slot) do { } while (0)
#endif
template <bool INIT, int FRESH = -1>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_linearise_schur(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  SLS_K1_STAMP_INIT;
  const Chunk ck = p.chunks[blockIdx.x];
  if (ck.win < 0) return;                              // an unused entry of a refillable batch's chunk array (lba_types.h)
  const WinDesc wd = p.wins[ck.win];
  const LMState* st = p.state + ck.win;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const double radius = st->radius;
  const double inv_radius = 1.0 / radius;
  const bool need_grad = st->need_grad_check != 0;
  const bool same_point = !INIT && st->same_point != 0;   // g_c and diag(J_c^T J_c) in the slab are still those of this point
  const int n = wd.n, ncf = n / 6, nsys = sys_doubles(n);
  double* camtab = smem;
  double* camscale = camtab + wd.C * kCamTab;
  double* S = camscale + (n > 0 ? n : 6);
  signed char* camcf = (signed char*)(S + nsys);
  const bool fresh = !INIT && (FRESH < 0 ? st->fresh != 0 : FRESH == 1);   // this sweep is also the initial evaluation: see below
  // first tile: context and observations requested before the camera table is built (a chunk's set-up is a chain of round trips to
  // memory; on the latency path - chunks of a tile or two - it is a third of the sweep)
  TileCtx nxt = fetch_tile(p, ck.tile_begin, ck.tile_end, lane);
  ObsPref pfn;
  prefetch_obs<false, !INIT>(p, nxt, cur, wd.obs_off, pfn, lane);
  SLS_K1_STAMP(6);
  load_cam_table<true>(p, wd, cur, lane, camtab, camscale, camcf, INIT || fresh, p.cam_tab ? ((INIT || fresh) ? 1 : 2) : 0);
  SLS_K1_STAMP(7);
  for (int q = lane; q < nsys; q += 64) S[q] = 0.0;
  __syncthreads();
  SLS_K1_STAMP(0);

  double acc_cost = 0.0, acc_fixed = 0.0, acc_gmax = 0.0, acc_xn2 = 0.0;
  int fail = 0;
#if defined(SLSLAM_K1_TIMING) && SLSLAM_K1_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  SLS_K1_STAMP(1);
  for (int t = ck.tile_begin; t < ck.tile_end; ++t) {
    SLS_PHASE("tile_head");
    const TileCtx tc = nxt;
    const ObsPref pf = pfn;
    const TileReq rq = request_tile(p, t + 1, ck.tile_end, lane);      // in flight while this tile is processed (resolved at the prefetch)
    const SegCtx sg = make_seg(tc, lane);
    const int j = tc.j, ls = tc.ls, o0 = tc.o0, k = tc.k;
    const bool line_ok = tc.line_ok;
    LaneLin L;
    double ob[8];
    SLS_PHASE("linearise");
    lane_linearise<!INIT>(p, pol, camtab, camscale, camcf, ls, j, k, o0, line_ok, tc.lflags, cur, wd.obs_off, L, ob, &pf, fresh);
    if (L.kept) acc_cost += L.cost;
    if ((INIT || fresh) && L.valid && !L.kept) acc_fixed += L.cost;

    double H[10], g[4];
    SLS_PHASE("line_block");
    line_block(L, sg, H, g);
    const bool line_active = L.line_free && k > 0;   // uniform over the line's run

    if (INIT) {
      // Jacobi scaling of the line's columns: 1 / (1 + ||J_col||), estimated once at x0
      if (line_ok && j == 0) {
        const double d[4] = { H[0], H[2], H[5], H[9] };
        double* lsc = p.line_scale + (long long)ls * 4;
        const double* u = p.line_x + line_rec(p, ls, cur);
        for (int a = 0; a < 4; ++a) {
          lsc[a] = (pol.jacobi_scaling && line_active) ? 1.0 / (1.0 + sqrt(d[a])) : 1.0;
          if (line_active) { acc_gmax = fmax(acc_gmax, fabs(g[a])); acc_xn2 += u[a] * u[a]; }
        }
      }
      if (L.valid && L.cf >= 0) {
        double* rec = S + L.cf * kCamAcc;
        for (int a = 0; a < 6; ++a) {
          double ga = 0.0, ha = 0.0;
          for (int r = 0; r < 4; ++r) { ga += L.Jc[6 * r + a] * L.rs[r]; ha += L.Jc[6 * r + a] * L.Jc[6 * r + a]; }
          lds_add(&rec[kRecG + a], ga);
          lds_add(&rec[kRecH + a], ha);
        }
      }
      nxt = resolve_tile(rq);
      prefetch_obs<false, !INIT>(p, nxt, cur, wd.obs_off, pfn, lane);
      continue;
    }

    SLS_PHASE("fresh_scale");
    if (fresh) {
      // first sweep of a solve: H, g are those of the UNSCALED line columns.  Jacobi scale of the line from them
      // (1 / (1 + ||J_col||), Ceres: once at x0), the line's share of the initial gradient norm and of |x|, then the
      // block, the gradient and the lane's line Jacobian go to scaled coordinates and the elimination proceeds
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
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int a = 0; a < 4; ++a) L.Jl[4 * q + a] *= sl[a];
    }

    // ---- eliminate the line: A = H + D^2, A^-1 = K^T K
    SLS_PHASE("factor4x4");
    double D2[4], K[10], u[4] = { 0, 0, 0, 0 }, F[24];
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
      if (need_grad && line_ok && j == 0) {       // only the launch after an accepted step tests the gradient
        for (int a = 0; a < 4; ++a) acc_gmax = fmax(acc_gmax, fabs(g[a] * fast_rcp(pf.lsc[a])));
      }
    }
    SLS_PHASE("lane_F");
    const bool cam_free = L.valid && L.cf >= 0;
    const bool elim = cam_free && L.line_free;       // this observation couples a free camera to a free line
    lane_F(L, K, F);                                  // K = 0 for a constant line: F = 0 where the diagonal block reads it; lanes
                                                      // without a free camera never have their F read
    // keep what the back-substitution needs (24 doubles per coupled observation, 22 per line) so
    // that it does not have to linearise again
    SLS_PHASE("keep_factor");
    if (pol.store_f && elim) {
      const long long o = (long long)o0 + j;
#pragma unroll
      for (int q = 0; q < 12; ++q)
        reinterpret_cast<double2*>(p.fstore)[(long long)q * p.ob_stride + o] = make_double2(F[2 * q], F[2 * q + 1]);
    }
    // the per-line factor is kept for the back-substitution of the same iteration (176 B per line against
    // ~500 B of observations): it does not have to rebuild and refactor the 4x4 block
    if (line_active && j == 0) {
      double* le = p.line_elim + (long long)ls * p.line_elim_stride;
#pragma unroll
      for (int q = 0; q < 10; ++q) le[q] = K[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) { le[kLeD2 + q] = D2[q]; le[kLeG + q] = g[q]; }
      if (pol.store_f) {
#pragma unroll
        for (int q = 0; q < 4; ++q) le[kLeU + q] = u[q];      // only the streaming back-substitution wants K g
      }
    }

    SLS_PHASE("prefetch_next");
    nxt = resolve_tile(rq);
    prefetch_obs<false, !INIT>(p, nxt, cur, wd.obs_off, pfn, lane);
    __builtin_amdgcn_sched_barrier(0);
    SLS_PHASE("diag_block");
    if (cam_free) {
      // Lanes of one 16-lane row whose observations belong to the same camera add to the same addresses, and the LDS takes
      // such lanes one after the other.  The packer marks every second of them (TileCtx.skew): a marked lane adds each entry ONE
      // step later than its neighbours - in any one ds_add_f64 the two then name different entries of the record.  The
      // delayed value waits in two registers; selects and address arithmetic are VALU work, which this sweep has to spare.
      // (Three levels - the third same-camera lane of a row two steps late - measured slower: 1.26 -> 1.52 ms.)
      double* rec = S + L.cf * kCamAcc;
#if defined(SLSLAM_NO_SKEW)
      const bool skew = false;
#else
      const bool skew = tc.skew;
#endif
      double pval = 0.0;
      int poff = kRecB;
      auto emit = [&](int off, double val) {
        lds_add_rec(rec + (skew ? poff : off), skew ? pval : val);
        pval = val; poff = off;
      };
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double ga = 0.0, ha = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ga += L.Jc[6 * r + a] * L.rs[r]; ha += L.Jc[6 * r + a] * L.Jc[6 * r + a]; }
        const double fu = F[4 * a] * u[0] + F[4 * a + 1] * u[1] + F[4 * a + 2] * u[2] + F[4 * a + 3] * u[3];
        if (!same_point) {
          emit(kRecG + a, ga);
          emit(kRecH + a, ha);
        }
        emit(kRecB + a, ga - fu);
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double v = 0.0;
#pragma unroll
          for (int r = 0; r < 4; ++r) v += L.Jc[6 * r + a] * L.Jc[6 * r + b];
#pragma unroll
          for (int m = 0; m < 4; ++m) v -= F[4 * a + m] * F[4 * b + m];
          emit(tri_index(a, b), v);
        }
      }
      if (skew) lds_add_rec(rec + poff, pval);             // the marked lanes' last entry
    }

    // ---- off-diagonal camera pairs of the tile, balanced over the lanes
    SLS_PHASE("pair_loop_ctl");
    for (int base_it = 0; base_it < tc.nitems; base_it += 64) {
      SLS_PHASE("pair_gather");
      const int it = base_it + lane;
      const bool has = it < tc.nitems;
      int li = 0, lj = 0;
      if (base_it < 128) {                          // the first two passes' items came with the tile's request
        const int w16 = base_it == 0 ? (pf.items01 & 0xffff) : ((pf.items01 >> 16) & 0xffff);
        if (has) { li = w16 & 0xff; lj = w16 >> 8; }
      } else if (has) {
        li = p.items[2 * (long long)(tc.item_off + it)]; lj = p.items[2 * (long long)(tc.item_off + it) + 1];
      }
      double Fi[24], Fj[24];
#pragma unroll
      for (int q = 0; q < 24; ++q) { Fi[q] = __shfl(F[q], li); Fj[q] = __shfl(F[q], lj); }
      const int ci = __shfl(L.cf, li), cj = __shfl(L.cf, lj);
      SLS_PHASE("pair_product");
      if (has) {
        if (cj != ci) {        // cj > ci by construction
          double* blk = S + pair_base(ncf, cj, ci);
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
              double v = 0.0;                       // - F_j F_i^T: the sign rides on the fma's operand modifier
#pragma unroll
              for (int m = 0; m < 4; ++m) v -= Fj[4 * a + m] * Fi[4 * b + m];
              lds_add_pair(&blk[6 * a + b], v);
            }
        } else {               // the same camera observes the line twice: symmetric part
          SLS_PHASE("pair_same_camera");
          for (int a = 0; a < 6; ++a)
            for (int b = 0; b <= a; ++b) {
              double v = 0.0;
              for (int m = 0; m < 4; ++m) v += Fj[4 * a + m] * Fi[4 * b + m] + Fi[4 * a + m] * Fj[4 * b + m];
              lds_add(&S[ci * kCamAcc + tri_index(a, b)], -v);
            }
        }
      }
      SLS_PHASE("pair_loop_ctl");
    }
  }
  SLS_PHASE("epilogue");
  SLS_K1_STAMP(2);

  __syncthreads();
  double* slab = p.slab + ck.slab_off;
  // (four entries per lane and round: the LDS reads of a round are in flight together; the test for the entries a rejected step
  // must not overwrite is taken out of the common case - it was most of the instructions of this loop, 2.5 us of a window's 20 us
  // sweep when its chunk is one tile)
  if (!same_point) {
    for (int q0 = lane; q0 < nsys; q0 += 256) {
      double v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = S[q0 + 64 * e < nsys ? q0 + 64 * e : 0];
#pragma unroll
      for (int e = 0; e < 4; ++e) if (q0 + 64 * e < nsys) slab[q0 + 64 * e] = v[e];
    }
  } else {
    for (int q = lane; q < nsys; q += 64) {
      // after a rejected step the gradient / column-norm entries of the camera records were not accumulated: keep the slab's
      if (q < ncf * kCamAcc && (q % kCamAcc) >= kRecG) continue;
      slab[q] = S[q];
    }
  }
  SLS_K1_STAMP(3);
  const double c_sum = wave_sum(acc_cost), f_sum = wave_sum(acc_fixed), x_sum = wave_sum(acc_xn2);
  const double g_max = wave_max(acc_gmax);
  const int any_fail = __any(fail);
  SLS_K1_STAMP(4);
  if (lane == 0) {
    double* sc = slab + nsys;
    sc[kScCost] = c_sum; sc[kScFixedCost] = f_sum; sc[kScGradMaxLine] = g_max; sc[kScXn2Line] = x_sum;
    sc[kScFail] = any_fail ? 1.0 : 0.0;
  }
  SLS_K1_STAMP(5);
}

// ------------------------------------------------------------------------------------------
// Kernel 2: reduced camera system of one window, one 256-thread workgroup (4 waves).
//   1. ordered sum of the window's chunk partials (bitwise reproducible; every thread owns entries,
//      the loads of one entry over the chunks are independent and in flight together)
//   2. gradient-tolerance test Ceres does right after accepting a step
//   3. S += D_c^2, blocked right-looking Cholesky in LDS with 16x16 tiles: diagonal tile in the registers of
//      one wave together with its triangular inverse, then the panel (L21 = A21 L11^-T) and
//      the trailing update (A22 -= L21 L21^T) on v_mfma_f64_16x16x4_f64, tiles dealt to the 4 waves —
//      the one dense contraction of the LBA path (n = 6 Cf = 60: 64 MFMA per factorisation)
//   4. block forward / backward substitution with the diagonal-tile inverses
//   5. step statistics of the camera block, candidate camera poses
__device__ __forceinline__ void push_trace(BatchPtrs& p, int w, LMState* st, const IterRec& r) {
  if (st->ntrace < kMaxTrace) p.trace[(long long)w * kMaxTrace + st->ntrace] = r;
  st->ntrace++;
}

__host__ __device__ inline int solve_pad(int n) { return ((n + 15) / 16) * 16; }
__host__ __device__ inline int solve_stride(int n) { return ((solve_pad(n) + 30) / 32) * 32 + 1; }  // == 1 (mod 32) doubles
// Where entry q of a chunk partial (camera records | pair blocks, see kCamAcc / kPairAcc) lives in the reduced solve's LDS image
// A (N x ld, lower triangle) | b | g | hdiag; 0xFFFF = padding.  One table per distinct n of a batch, built on the host.
inline void sys_map_build(int n, unsigned short* out) {
  const int ncf = n / 6, N = solve_pad(n), ld = solve_stride(n);
  for (int q = 0; q < sys_doubles(n); ++q) {
    int off = 0xFFFF;
    if (q < ncf * kCamAcc) {
      const int cf = q / kCamAcc, ee = q - cf * kCamAcc;
      if (ee < kRecB) {
        int a = 0;
        while (((a + 1) * (a + 2)) / 2 <= ee) ++a;
        off = (6 * cf + a) * ld + 6 * cf + (ee - (a * (a + 1)) / 2);
      } else {
        const int v2 = (ee - kRecB) / 6, a = (ee - kRecB) - 6 * v2;      // 0 = b, 1 = g, 2 = hdiag
        off = N * ld + v2 * N + 6 * cf + a;
      }
    } else {
      const int pq = q - ncf * kCamAcc, pr = pq / kPairAcc, ee = pq - pr * kPairAcc;
      if (ee < 36) {
        int cj = 1;
        while (((cj + 1) * cj) / 2 <= pr) ++cj;
        const int ci = pr - (cj * (cj - 1)) / 2;
        off = (6 * cj + ee / 6) * ld + 6 * ci + (ee % 6);
      }
    }
    out[q] = (unsigned short)off;
  }
}
// The inverse of a diagonal tile's factor is kept IN the tile: strictly lower part transposed into the tile's
// (otherwise unused) strict upper triangle, its diagonal 1 / L[r][r] in a vector: 36.5 KB of LDS for n = 60,
// i.e. 4 workgroups per CU and the whole 1024-window batch resident in one round.
__host__ __device__ inline int lds_doubles_solve(int n) {
  const int N = solve_pad(n);
  return N * solve_stride(n) + 6 * N + 16 + 9 * (n / 6);     // + JL of every free camera (matrix-core sweep: T_c is applied here)
}
typedef double solve_acc_t __attribute__((ext_vector_type(4)));

// Phase timing of the reduced solve (debug_flags bit 9, timing experiments only): wave 0 stamps the shader clock at the phase
// boundaries, dbg_cycles[win * 16 + phase] accumulates (slslam_debug_phase_cycles reads it).
__device__ __forceinline__ unsigned long long solve_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
#define SLS_SOLVE_STAMP(i)                                                                   \
  do {                                                                                       \
    if (timing) { const unsigned long long now_ = solve_clock(); if (tid == 0) p.dbg_cycles[(long long)w * 16 + (i)] += now_ - tlast_; tlast_ = now_; } \
  } while (0)


// Sum of a window's chunk partials, spread over the chip: launched ahead of k_reduced_solve when a window has many chunks
// (a single window is cut into ~50 so that its sweeps fill the CUs; one workgroup reading all of them is bound by the
// load bandwidth of its CU, ~10 B per cycle: 0.8 MB = 30 us on the latency path of every iteration).  One thread per entry,
// partials added in chunk order (the sum k_reduced_solve would form); per-chunk scalars: cost / fixed cost / |x|^2 summed,
// gradient norm and failure flag maxed.  Result: one slab per window in slab_sum, same layout.
// kSlabReduceSub lanes share an entry: lane j adds the partials of the chunks j, j + kSlabReduceSub, ... in increasing order, then the
// lanes' sums are added in lane order - a fixed association, so the result is reproducible (and the same whatever batch the window is
// in); one thread walking 200 partials was 18 us of dependent round trips on the latency path of a 2000-line window.
enum { kSlabReduceSub = 16 };
__global__ __launch_bounds__(256) void k_slab_reduce(BatchPtrs p) {
  const int w = blockIdx.x;
  const WinDesc wd = p.wins[w];
  // (a window of at most 8 chunks is summed by its reduced solve itself, as in a batch without this launch: what a window's
  // result depends on is its own chunk count, not the company it keeps)
  if (p.state[w].status != kRunning || wd.nchunks <= 8) return;
  const int nsys = p.elim_mode == 1 ? sys_doubles_mfma(wd.n) : sys_doubles(wd.n);
  const long long sstride = (long long)nsys + kSlabScalars;
  const int t = blockIdx.y * 256 + threadIdx.x;
  const int q = t / kSlabReduceSub, j = t - q * kSlabReduceSub;
  const bool live = q < nsys + kSlabScalars;                  // (whole groups of lanes: the shuffles below want their partners)
  const double* src = p.slab + wd.slab_off + (live ? q : 0);
  const bool is_max = q == nsys + kScGradMaxLine || q == nsys + kScFail;
  double s = 0.0;
  for (int k0 = j; k0 < wd.nchunks; k0 += 8 * kSlabReduceSub) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (k0 + u * kSlabReduceSub < wd.nchunks) ? src[(long long)(k0 + u * kSlabReduceSub) * sstride] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s = is_max ? fmax(s, v[u]) : s + v[u];
  }
  // the group's sums, in lane order
  const int base = (threadIdx.x & 63) & ~(kSlabReduceSub - 1);
  double tot = __shfl(s, base);
#pragma unroll
  for (int i = 1; i < kSlabReduceSub; ++i) { const double o = __shfl(s, base + i); tot = is_max ? fmax(tot, o) : tot + o; }
  if (!live || j != 0) return;
  s = tot;
  if (p.slab_sum_image) {
    // straight into the layout the reduced solve keeps in LDS: its first phase was a chain of dependent global reads (partials,
    // then the map, then the scalars: ~2 us each for a window alone on the chip) and is now one sweep of independent loads
    const int N = solve_pad(wd.n), ext = N * solve_stride(wd.n) + 6 * N;
    double* img = p.slab_sum + (long long)w * p.slab_sum_stride;
    if (q >= nsys) img[ext + (q - nsys)] = s;
    else {
      const unsigned off = (p.sys_map + wd.map_off)[q];
      if (off != 0xFFFFu) img[off] = s;
    }
    return;
  }
  p.slab_sum[(long long)w * p.slab_sum_stride + q] = s;
}

// RESIDENT = 4: four workgroups per CU - the LDS limit - need <= 128 registers: the whole 1024-window batch is resident in one
// round.  RESIDENT = 1: a batch of at most one window per CU (the reference's call protocol: one window) has the CU to itself and
// the tile factorisation on its critical path gets the registers it wants (no scratch).
// the lane index, remade on every call (asm volatile: never merged into one long-lived value)
__device__ __forceinline__ int sls_fresh_lane() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  return x;
}

template <int RESIDENT>
__global__ __launch_bounds__(256, RESIDENT) void k_reduced_solve(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // (the thread's indices are remade where they are used - the lane from v_mbcnt behind an asm the compiler does not merge, the wave kept in a scalar
  // register: as one value that lives from the first line to the last the thread index was spilled under the 128-register cap of the
  // four-workgroups-per-CU form and reloaded from scratch 53 times along the solve)
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#define lane sls_fresh_lane()
#define tid (wave * 64 + sls_fresh_lane())
  const int w = blockIdx.x;
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int n = wd.n, ncf = n / 6, nsys = sys_doubles(n), N = solve_pad(n), nt = N / 16, ld = solve_stride(n);
  double* A = smem;                       // N x ld, lower triangle used
  double* bvec = A + N * ld;              // b | g | hdiag | y | tmp | 1 / diag(L), N each
  double* gvec = bvec + N;
  double* hvec = gvec + N;
  double* yvec = hvec + N;
  double* tvec = yvec + N;
  double* ivec = tvec + N;
  double* red = ivec + N;                 // 16 scratch doubles
  const int cur = st->cur;
  const double radius = st->radius;
  const int need_grad_check = st->need_grad_check;
  const double abs_grad_tol = st->abs_grad_tol;
  const int fresh = st->fresh;            // read before the barriers below: wave 0 clears it in step 1b
  const bool timing = (pol.debug_flags & 512) && p.dbg_cycles;
  unsigned long long tlast_ = timing ? solve_clock() : 0ull;
  // The latency form (RESIDENT == 1: at most one window per CU) asks NOW for what its last phase needs of the window's cameras -
  // pose, Jacobi scale, free index: known before the solve, read after it as a chain of dependent round trips otherwise
  // (index, then per component the pose entry behind the store of the previous one: ~3 k of the launch's 50 k cycles).
  constexpr bool kPreloadCams = RESIDENT == 1;
  double pre_x[6], pre_s[6];
  int pre_cf = -1;
  if (kPreloadCams && wave < 2) {
    const int c = lane < wd.C ? lane : 0;
    pre_cf = p.cam_cf[wd.cam_off + c];
    const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
    const double* sc = p.cam_scale + (long long)(wd.cam_off + c) * 6;
#pragma unroll
    for (int a = 0; a < 6; ++a) { pre_x[a] = x[a]; pre_s[a] = sc[a]; }
  }

  // ---- 1. ordered reduction over the window's chunk partials (uniform stride between consecutive slabs)
  const bool mfma_slab = p.elim_mode == 1;          // slab layout of lba_eliminate_mfma.h
  const bool presummed = p.slab_sum != nullptr && wd.nchunks > 8;       // k_slab_reduce ran for this window: one partial
  const bool image = presummed && p.slab_sum_image;                      // ... already in the layout of this kernel's LDS
  const int nsys_slab = mfma_slab ? kMfmaTiles * 256 + ncf * kMfmaRec : image ? N * ld + 6 * N : nsys;
  const double* slab_base = presummed ? p.slab_sum : p.slab;
  const long long slab0 = presummed ? (long long)w * p.slab_sum_stride : (wd.nchunks > 0 ? wd.slab_off : 0);
  const long long sstride = (long long)nsys_slab + kSlabScalars;
  const int nchunks = presummed ? 1 : wd.nchunks;
  // (the first chunk scalars of every thread are requested ahead of the image / the zeroing, so that their round trip overlaps it)
  // (latency form only: the batch form - four workgroups per CU, 128 registers - has other waves to run meanwhile and no register for it)
  constexpr bool kEarlyScalars = RESIDENT == 1;
  const bool have_sc0 = kEarlyScalars && tid < nchunks;
  double sc0_gmax = 0.0, sc0_fail = 0.0;
  if (kEarlyScalars) {
    const double* sc0 = slab_base + slab0 + (long long)(have_sc0 ? tid : 0) * sstride + nsys_slab;
    sc0_gmax = sc0[kScGradMaxLine]; sc0_fail = sc0[kScFail];
  }
  if (image) {
    const double2* src = reinterpret_cast<const double2*>(slab_base + slab0);
    double2* dst = reinterpret_cast<double2*>(smem);
    for (int q = tid; q < (N * ld + 6 * N) / 2; q += 256) dst[q] = src[q];
  } else {
    for (int q = tid; q < N * ld; q += 256) A[q] = 0.0;
    for (int q = tid; q < 6 * N; q += 256) bvec[q] = 0.0;
  }
  __syncthreads();
  const unsigned short* smap = p.sys_map + wd.map_off;
  if (mfma_slab) {
    // The matrix-core sweep leaves everything in RAW camera coordinates (J_c' = [tau | gP]: no SO(3) left Jacobian, no
    // Jacobi scale).  (1) camera records: D' = J_c'^T J_c' (diagonal blocks), b', g'
    for (int q = tid; q < ncf * kMfmaRec; q += 256) {
      const double* src = slab_base + slab0 + kMfmaTiles * 256 + q;
      double s = 0.0;
      for (int k0 = 0; k0 < nchunks; k0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(k0 + u < nchunks ? k0 + u : nchunks - 1) * sstride];      // (no load under a condition: all eight in flight together)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k0 + u < nchunks) ? v[u] : 0.0;
      }
      const int cf = q / kMfmaRec, e = q - cf * kMfmaRec;
      if (e < 21) {
        const int a = e >= 15 ? 5 : e >= 10 ? 4 : e >= 6 ? 3 : e >= 3 ? 2 : e >= 1 ? 1 : 0, b = e - tri_index(a, 0);
        A[(6 * cf + a) * ld + 6 * cf + b] = s;
      } else if (e < 27) {
        bvec[6 * cf + (e - 21)] = s;
      } else {
        gvec[6 * cf + (e - 27)] = s;
      }
    }
    // the cameras' left Jacobians JL(w) at the accepted point: T_c = diag(JL diag(s_w), diag(s_t)), J_c = J_c' T_c
    double* jlm = red + 16;                                // [ncf][9]
    for (int c = tid; c < wd.C; c += 256) {
      const int cf = p.cam_cf[wd.cam_off + c];
      if (cf < 0) continue;
      const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
      double wv[3] = { x[0], x[1], x[2] }, R[9], JL[9];
      cam_prepare<double>(wv, R, JL);
      for (int q = 0; q < 9; ++q) jlm[9 * cf + q] = JL[q];
    }
    __syncthreads();
    // (2) diag(J_c^T J_c) with unit scale (LM diagonal, Jacobi scale), b and g through T0 = diag(JL, I)
    for (int q = tid; q < n; q += 256) {
      const int cf = q / 6, a = q - 6 * cf;
      const double* JL = jlm + 9 * cf;
      double h;
      if (a >= 3) h = A[q * ld + q];
      else {
        h = 0.0;
        for (int i = 0; i < 3; ++i)
          for (int j2 = 0; j2 < 3; ++j2) {
            const int r = 6 * cf + (i >= j2 ? i : j2), c = 6 * cf + (i >= j2 ? j2 : i);
            h += JL[3 * i + a] * A[r * ld + c] * JL[3 * j2 + a];
          }
      }
      hvec[q] = h;
      if (a < 3) {
        double sb = 0.0, sg = 0.0;
        for (int i = 0; i < 3; ++i) { sb += JL[3 * i + a] * bvec[6 * cf + i]; sg += JL[3 * i + a] * gvec[6 * cf + i]; }
        tvec[q] = sb; yvec[q] = sg;                         // staged: the loop reads all three raw entries of a camera
      }
    }
    __syncthreads();
    for (int q = tid; q < n; q += 256) if (q % 6 < 3) { bvec[q] = tvec[q]; gvec[q] = yvec[q]; }
    // (3) S' = blockdiag(D') - P: accumulator tiles of P = sum_lines X X^T (tile t = (I, J), J <= I; entry q * 64 + lane is
    // row (lane >> 4) + 4 q, column lane & 15)
#pragma unroll 2
    for (int q = tid; q < kMfmaTiles * 256; q += 256) {
      int row, col;
      acc_row_col(q >> 8, (q & 255) >> 6, q & 63, &row, &col);
      if (row >= n || col > row) continue;
      const double* src = slab_base + slab0 + q;
      double s = 0.0;
      for (int k0 = 0; k0 < nchunks; k0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(k0 + u < nchunks ? k0 + u : nchunks - 1) * sstride];      // (no load under a condition: all eight in flight together)
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (k0 + u < nchunks) ? v[u] : 0.0;
      }
      A[row * ld + col] -= s;
    }
    __syncthreads();
    // (4) S = T0^T S' T0, block by block: diagonal blocks mirrored to full 6x6 first (the upper triangle of A is free
    // until the factorisation), then every block's rows times JL_ci (w columns), then its columns times JL_cj^T (w rows)
    for (int q = tid; q < ncf * 15; q += 256) {
      const int cf = q / 15, e = q - 15 * cf;
      const int a = e >= 10 ? 5 : e >= 6 ? 4 : e >= 3 ? 3 : e >= 1 ? 2 : 1, b = e - ((a - 1) * a) / 2;   // a > b
      A[(6 * cf + b) * ld + 6 * cf + a] = A[(6 * cf + a) * ld + 6 * cf + b];
    }
    __syncthreads();
    const int nblk = (ncf * (ncf + 1)) / 2;
    for (int q = tid; q < nblk * 6; q += 256) {
      const int blk = q / 6, a = q - 6 * blk;
      int cj = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
      while ((cj * (cj + 1)) / 2 > blk) --cj;
      while (((cj + 1) * (cj + 2)) / 2 <= blk) ++cj;
      const int ci = blk - (cj * (cj + 1)) / 2;
      double* row = A + (6 * cj + a) * ld + 6 * ci;
      const double* JL = jlm + 9 * ci;
      const double x0 = row[0], x1 = row[1], x2 = row[2];
      for (int c = 0; c < 3; ++c) row[c] = x0 * JL[c] + x1 * JL[3 + c] + x2 * JL[6 + c];
    }
    __syncthreads();
    for (int q = tid; q < nblk * 6; q += 256) {
      const int blk = q / 6, b = q - 6 * blk;
      int cj = (int)((sqrt(8.0 * blk + 1.0) - 1.0) * 0.5);
      while ((cj * (cj + 1)) / 2 > blk) --cj;
      while (((cj + 1) * (cj + 2)) / 2 <= blk) ++cj;
      const int ci = blk - (cj * (cj + 1)) / 2;
      double* col = A + (6 * cj) * ld + 6 * ci + b;
      const double* JL = jlm + 9 * cj;
      const double x0 = col[0], x1 = col[ld], x2 = col[2 * ld];
      for (int r = 0; r < 3; ++r) col[r * ld] = JL[r] * x0 + JL[3 + r] * x1 + JL[6 + r] * x2;
    }
    __syncthreads();
    // (5) Jacobi scale of the camera columns, kept since the first iteration (the first one derives it below)
    if (!fresh) {
      for (int q = tid; q < 6 * wd.C; q += 256) {
        const int c = q / 6, cf = p.cam_cf[wd.cam_off + c];
        if (cf >= 0) tvec[6 * cf + (q - 6 * c)] = p.cam_scale[(long long)(wd.cam_off + c) * 6 + (q - 6 * c)];
      }
      __syncthreads();
      for (int q = tid; q < n * ld; q += 256) {
        const int r = q / ld, c = q - r * ld;
        if (c <= r) A[q] *= tvec[r] * tvec[c];
      }
      for (int q = tid; q < n; q += 256) {
        const double sc = tvec[q];
        bvec[q] *= sc; gvec[q] *= sc; hvec[q] *= sc * sc;
      }
    }
  } else if (!image) {
  // eight entries of this thread at a time: the loads of an entry's partials (up to 8 chunks per round) of all eight are
  // in flight together; per entry the partials are added in chunk order, so the result does not depend on the grouping
  for (int q0 = tid; q0 < nsys; q0 += 8 * 256) {
    double acc8[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (nchunks <= 8) {
#pragma unroll
      for (int e0 = 0; e0 < 8; e0 += 2) {            // two entries x eight chunks in flight (the batched case: ~6 chunks)
        double v[2][8];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int q = q0 + 256 * (e0 + e);
            v[e][u] = (q < nsys && u < nchunks) ? slab_base[slab0 + (long long)u * sstride + q] : 0.0;
          }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int u = 0; u < 8; ++u) acc8[e0 + e] += v[e][u];
      }
    } else
    for (int k0 = 0; k0 < nchunks; k0 += 4) {
      double v[8][4];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int q = q0 + 256 * e;
          v[e][u] = (q < nsys && k0 + u < nchunks) ? slab_base[slab0 + (long long)(k0 + u) * sstride + q] : 0.0;
        }
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc8[e] += v[e][u];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int q = q0 + 256 * e;
      if (q >= nsys) continue;
      const unsigned off = smap[q];                   // where the entry lives in A | b | g | hdiag (host-built, sys_map_build)
      if (off != 0xFFFFu) smem[off] = acc8[e];
    }
  }
  }
  double gmax_line = 0.0;
  int fail = 0;
  {
    // per-chunk scalars: the threads share the chunks (a single window has ~50 of them), then a block-wide max / any
    double gm = have_sc0 ? sc0_gmax : 0.0;
    int fl = (have_sc0 && sc0_fail != 0.0) ? 1 : 0;
    for (int k = kEarlyScalars ? tid + 256 : tid; k < nchunks; k += 256) {
      const double* sc = slab_base + slab0 + (long long)k * sstride + nsys_slab;
      gm = fmax(gm, sc[kScGradMaxLine]);
      if (sc[kScFail] != 0.0) fl = 1;
    }
    gm = wave_max(gm);
    fl = __any(fl) ? 1 : 0;
    if (lane == 0) { red[4 + wave] = gm; red[8 + wave] = (double)fl; }
    __syncthreads();
    gmax_line = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    fail = (red[8] + red[9] + red[10] + red[11]) != 0.0 ? 1 : 0;
  }
  __syncthreads();

  SLS_SOLVE_STAMP(0);
  // ---- 1b. first iteration of a solve: the elimination sweep ran with unit camera scale, so the system just read is in
  // UNSCALED camera coordinates.  Do what Ceres' initial evaluation does (cost, gradient max-norm, |x|, Jacobi scale of the
  // camera columns from diag(J^T J) at x0, trace record 0, the tests that can end a solve before its first step), then
  // bring S, b, g, diag to scaled coordinates (a congruence with diag(scale)) and carry on as in every other iteration.
  if (fresh) {
    if (wave == 0) {
      double cost = 0.0, fixed = 0.0, xn2 = 0.0, gmax = 0.0;
      for (int k = lane; k < nchunks; k += 64) {
        const double* sc = slab_base + slab0 + (long long)k * sstride + nsys_slab;
        cost += sc[kScCost]; fixed += sc[kScFixedCost]; xn2 += sc[kScXn2Line];
      }
      if (lane == 0) gmax = gmax_line;
      for (int q = lane; q < N; q += 64) tvec[q] = 1.0;
      for (int q = lane; q < 6 * wd.C; q += 64) {
        const int c = q / 6, a = q - 6 * c;
        const int cf = p.cam_cf[wd.cam_off + c];
        double sc = 1.0;
        if (cf >= 0) {
          const double x = p.cam_x[((long long)(wd.cam_off + c) * 2 + cur) * kCamRec + a];
          gmax = fmax(gmax, fabs(gvec[6 * cf + a]));
          xn2 += x * x;
          if (pol.jacobi_scaling) sc = 1.0 / (1.0 + sqrt(hvec[6 * cf + a]));
          tvec[6 * cf + a] = sc;
        }
        p.cam_scale[(long long)(wd.cam_off + c) * 6 + a] = sc;
      }
      cost = wave_sum(cost); fixed = wave_sum(fixed); xn2 = wave_sum(xn2); gmax = wave_max(gmax);
      if (lane == 0) {
        st->cost = cost; st->fixed_cost = fixed; st->initial_cost = cost + fixed; st->min_cost = cost + fixed;
        st->x_norm = sqrt(xn2);
        st->grad_max = gmax;
        st->abs_grad_tol = pol.gradient_tolerance * (gmax > 1e-12 ? gmax : 1e-12);
        st->need_grad_check = 0;
        st->fresh = 0;
        int status = kRunning;
        if (wd.nfree_params == 0) status = 2;                    // FUNCTION_TOLERANCE: no free blocks
        else if (!isfinite(cost)) status = 4;
        else if (gmax <= st->abs_grad_tol) status = 1;
        if (status == kRunning) {
          IterRec rec;
          rec.pad = 0;
          rec.iteration = 0; rec.step_is_valid = 0; rec.step_is_successful = 0;
          rec.cost = cost + fixed; rec.cost_change = 0; rec.gradient_max_norm = gmax; rec.step_norm = 0;
          rec.relative_decrease = 0; rec.trust_region_radius = st->radius; rec.model_cost_change = 0;
          push_trace(p, w, st, rec);
        }
        st->status = status;
        red[1] = (double)status;
      }
    }
    __syncthreads();
    if (red[1] != (double)kRunning) return;
    for (int q = tid; q < n * ld; q += 256) {
      const int r = q / ld, c = q - r * ld;
      if (c <= r) A[q] *= tvec[r] * tvec[c];
    }
    for (int q = tid; q < n; q += 256) {
      const double sc = tvec[q];
      bvec[q] *= sc; gvec[q] *= sc; hvec[q] *= sc * sc;
    }
    __syncthreads();
  }

  // ---- 2. gradient max-norm at the accepted point: gvec holds the SCALED gradient J'^T r, the true
  // gradient is g / scale.  Ceres tests it right after accepting a step.
  if (need_grad_check) {
    if (wave == 0) {
      double gm = 0.0;
      for (int c = lane; c < wd.C; c += 64) {
        const int cf = p.cam_cf[wd.cam_off + c];
        if (cf < 0) continue;
        for (int a = 0; a < 6; ++a)
          gm = fmax(gm, fabs(gvec[6 * cf + a] / p.cam_scale[(long long)(wd.cam_off + c) * 6 + a]));
      }
      gm = fmax(wave_max(gm), gmax_line);
      if (lane == 0) {
        red[0] = gm;
        st->grad_max = gm;
        st->need_grad_check = 0;
        if (st->ntrace > 0 && st->ntrace <= kMaxTrace)
          p.trace[(long long)w * kMaxTrace + st->ntrace - 1].gradient_max_norm = gm;
        if (gm <= abs_grad_tol) st->status = 1 /* SLSLAM_GRADIENT_TOLERANCE */;
      }
    }
    __syncthreads();
    if (red[0] <= abs_grad_tol) return;
  }

  SLS_SOLVE_STAMP(1);
  // ---- 3. LM damping of the camera columns: D^2 = clamp(diag(J'^T J')) / radius; identity on the padding
  for (int q = tid; q < N; q += 256) {
    if (q < n) {
      const double d2 = fmin(fmax(hvec[q], pol.min_lm_diagonal), pol.max_lm_diagonal) / radius;
      hvec[q] = d2;
      A[q * ld + q] += d2;
    } else {
      A[q * ld + q] = 1.0;
    }
    yvec[q] = bvec[q];                          // right-hand side, turned into y = L^-1 b block row by block row during the factorisation
  }
  __syncthreads();

  SLS_SOLVE_STAMP(2);
  const int er = tid >> 4, ec = tid & 15;       // this thread's element of a 16x16 tile
  // (a)+(b) a diagonal tile in the registers of wave 0 (dense_tile.h): L^-1 strictly lower part transposed into the tile's upper
  // triangle, its diagonal into ivec (L of the diagonal tile itself is never read again: only its inverse is kept)
  auto factor_tile = [&](const int kb) {
    double* D = A + (16 * kb) * ld + 16 * kb;
    diag_tile_factor<double, false>(D, ld, lane, fail, [&](int r, int c, double v) {
      if (c < r) D[c * ld + r] = v; else ivec[16 * kb + r] = v;
    });
  };
  // A(i,j) -= L(i,kb) L(j,kb)^T, one tile by one wave
  auto update_tile = [&](const int kb, const int i, const int jt) {
    double* C = A + (16 * i) * ld + 16 * jt;
    const double* Pi = A + (16 * i) * ld + 16 * kb;
    const double* Pj = A + (16 * jt) * ld + 16 * kb;
    solve_acc_t acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = C[((lane >> 4) + 4 * q) * ld + (lane & 15)];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const double a = -Pi[(lane & 15) * ld + 4 * s4 + (lane >> 4)];
      const double b = Pj[(lane & 15) * ld + 4 * s4 + (lane >> 4)];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) C[((lane >> 4) + 4 * q) * ld + (lane & 15)] = acc[q];
  };
  // Forward substitution on the fly (one wave each): y_kb = Linv_kb r_kb once block row kb of the right-hand side carries the
  // contributions of all block columns before it, and r_i -= L(i,kb) y_kb for a block row below - in the shadow of the tile
  // factorisation, where three of the four waves would otherwise wait.  lane = (row lane & 15, columns 4 (lane >> 4) ..)
  auto tile_y = [&](const int kb) {
    const double* D = A + (16 * kb) * ld + 16 * kb;
    const int r = lane & 15, g4 = 4 * (lane >> 4);
    double sacc = 0.0;
#pragma unroll
    for (int c = g4; c < g4 + 4; ++c) sacc += (c < r ? D[c * ld + r] : (c == r ? ivec[16 * kb + r] : 0.0)) * yvec[16 * kb + c];
    sacc += __shfl_xor(sacc, 16); sacc += __shfl_xor(sacc, 32);
    if (lane < 16) yvec[16 * kb + r] = sacc;
  };
  auto rhs_update = [&](const int kb, const int i) {
    const double* T = A + (16 * i) * ld + 16 * kb;
    const int r = lane & 15, g4 = 4 * (lane >> 4);
    double sacc = 0.0;
#pragma unroll
    for (int c = g4; c < g4 + 4; ++c) sacc += T[r * ld + c] * yvec[16 * kb + c];
    sacc += __shfl_xor(sacc, 16); sacc += __shfl_xor(sacc, 32);
    if (lane < 16) yvec[16 * i + r] -= sacc;
  };
  if (wave == 0 && nt > 0) factor_tile(0);
  __syncthreads();
  SLS_SOLVE_STAMP(3);
  for (int kb = 0; kb < nt; ++kb) {
    const double* D = A + (16 * kb) * ld + 16 * kb;   // diagonal tile: its inverse
    // y of this block row (its right-hand side is complete since the last phase, its tile's inverse too): by the wave with the
    // fewest panel tiles, off wave 0's path
    if (wave == 3) tile_y(kb);
    // (c) panel: L(i,kb) = A(i,kb) Linv^T, one tile per wave at a time
    for (int i = kb + 1 + wave; i < nt; i += 4) {
      double* T = A + (16 * i) * ld + 16 * kb;
      solve_acc_t acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const double a = T[(lane & 15) * ld + 4 * s4 + (lane >> 4)];          // A[m][k]
        const int bn = lane & 15, bk = 4 * s4 + (lane >> 4);                  // B[k][n] = Linv[n][k], zero above the diagonal
        const double b = bn > bk ? D[bk * ld + bn] : (bn == bk ? ivec[16 * kb + bn] : 0.0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) T[((lane >> 4) + 4 * q) * ld + (lane & 15)] = acc[q];
    }
    __syncthreads();
    SLS_SOLVE_STAMP(4);
    // (d) trailing update: A(i,j) -= L(i,kb) L(j,kb)^T for kb < j <= i.  LOOK-AHEAD: wave 0 brings the next diagonal tile up to
    // date and factors it - the sixteen sequential pivots of a tile are the long pole of a step - while the other three waves
    // share the rest of the tiles round-robin (every tile by one wave, the same arithmetic as before)
    if (kb + 1 < nt) {
      if (wave == 0) {
        update_tile(kb, kb + 1, kb + 1);
        factor_tile(kb + 1);
      } else {
        int tile = 0;
        for (int i = kb + 1; i < nt; ++i)
          for (int jt = kb + 1; jt <= i; ++jt) {
            if (i == kb + 1) continue;                  // (kb + 1, kb + 1): wave 0's
            if ((tile++ % 3) + 1 == wave) update_tile(kb, i, jt);
          }
        for (int i = kb + 1 + (wave - 1); i < nt; i += 3) rhs_update(kb, i);
      }
    }
    __syncthreads();
    SLS_SOLVE_STAMP(9);
  }

  SLS_SOLVE_STAMP(5);
  // ---- 4. block substitution (forward: done above)
  SLS_SOLVE_STAMP(6);
  // backward: x_k = Linv_k^T (y_k - sum_{r >= 16(k+1)} L[r,k]^T x_r), in place in yvec
  for (int kb = nt - 1; kb >= 0; --kb) {
    {
      double sacc = 0.0;                                   // thread (er = column within tile, ec = row part)
      for (int r = 16 * (kb + 1) + ec; r < N; r += 16) sacc += A[r * ld + 16 * kb + er] * yvec[r];
      sacc += dpp_move<0xB1>(sacc); sacc += dpp_move<0x4E>(sacc); sacc += dpp_move<0x141>(sacc); sacc += dpp_move<0x140>(sacc);
      if (ec == 0) tvec[er] = yvec[16 * kb + er] - sacc;
    }
    __syncthreads();
    {
      const double* Dk = A + (16 * kb) * ld + 16 * kb;                                            // x = Linv^T t: Linv[ec][er]
      const double li = ec > er ? Dk[er * ld + ec] : (ec == er ? ivec[16 * kb + er] : 0.0);
      double sacc = li * tvec[ec];
      sacc += dpp_move<0xB1>(sacc); sacc += dpp_move<0x4E>(sacc); sacc += dpp_move<0x141>(sacc); sacc += dpp_move<0x140>(sacc);
      if (ec == 0) yvec[16 * kb + er] = sacc;
    }
    __syncthreads();
  }

  SLS_SOLVE_STAMP(7);
  // ---- 5. step statistics of the camera block and candidate camera poses (wave 0); next to it, wave 1: the candidate poses'
  // rotation / Jacobian table for the sweeps that follow (BatchPtrs.cam_tab)
  if (wave == 1 && p.cam_tab) {
    for (int c = lane; c < wd.C; c += 64) {
      const bool pre = kPreloadCams && c == lane;          // (kMaxCams = 64: the loop runs once)
      const int cf = pre ? pre_cf : p.cam_cf[wd.cam_off + c];
      const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
      double xc[6];
      for (int a = 0; a < 6; ++a) {
        double v = pre ? pre_x[a] : x[a];
        if (cf >= 0) {                                      // (the candidate pose, statement for statement as wave 0 forms it)
          const double d = -yvec[6 * cf + a] * ((pre && !fresh) ? pre_s[a] : p.cam_scale[(long long)(wd.cam_off + c) * 6 + a]);
          const double xn = v + d;
          v = xn;
        }
        xc[a] = v;
      }
      double R[9], JL[9];
      cam_prepare<double>(xc, R, JL);
      double* gt = p.cam_tab + ((long long)(wd.cam_off + c) * 2 + (1 - cur)) * kCamTab;
      for (int q = 0; q < 9; ++q) { gt[q] = R[q]; gt[9 + q] = JL[q]; }
      gt[18] = xc[3]; gt[19] = xc[4]; gt[20] = xc[5];
    }
  }
  if (wave != 0) return;
  double model = 0.0, dn2 = 0.0, xn2 = 0.0;
  int bad = 0;
  for (int q = lane; q < n; q += 64) {
    const double y = yvec[q];
    if (!isfinite(y)) bad = 1;
    model += 0.5 * y * (gvec[q] + hvec[q] * y);
    p.ysys[wd.sys_off + q] = y;
  }
  for (int c = lane; c < wd.C; c += 64) {
    const bool pre = kPreloadCams && c == lane;
    const int cf = pre ? pre_cf : p.cam_cf[wd.cam_off + c];
    const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
    double* xc = p.cam_x + ((long long)(wd.cam_off + c) * 2 + (1 - cur)) * kCamRec;
    for (int a = 0; a < 6; ++a) {
      double v = pre ? pre_x[a] : x[a];
      if (cf >= 0) {
        const double d = -yvec[6 * cf + a] * ((pre && !fresh) ? pre_s[a] : p.cam_scale[(long long)(wd.cam_off + c) * 6 + a]);
        const double xn = v + d;
        const double dd = v - xn;
        dn2 += dd * dd;
        xn2 += xn * xn;
        v = xn;
      }
      xc[a] = v;
    }
  }
  model = wave_sum(model); dn2 = wave_sum(dn2); xn2 = wave_sum(xn2);
  const int any_bad = __any(bad | fail);
  if (lane == 0) {
    st->cam_model = model; st->cam_dn2 = dn2; st->cam_xn2 = xn2;
    st->solve_failed = any_bad ? 1 : 0;
  }
  SLS_SOLVE_STAMP(8);
}
#undef lane
#undef tid

// ------------------------------------------------------------------------------------------
// Kernel 3, variant A (default): back-substitution  y_l = A^-1 (g_l - sum_i H_cl,i^T y_c) = K^T K (g_l - w),
// candidate line parameters, line part of the step statistics AND the cost at the candidate point.
// Re-linearises the chunk (the second of the algorithmically necessary observation sweeps,
// SURVEY.md 8d) instead of spilling Jacobian blocks to HBM; because the observation is still in
// registers when the line's candidate parameters become known, the candidate residual is
// evaluated right here: no third sweep over the observations and no separate sin/cos pass.
enum { kCandTab = 13 };   // doubles per camera of the candidate table: R[9] t[3]; odd stride in 8-byte units
__host__ __device__ inline int lds_doubles_backsub(int C, int n) { (void)n; return C * (kBsTab + kCandTab) + 8 + (C + 7) / 8; }

// sin/cos table of a candidate line, computed cooperatively: every lane of a line's run holds the same u[4];
// lane j < 4 of the run evaluates the sin/cos of angle j and the run shares the results (ds_bpermute from the
// run's first lanes).  Runs shorter than 4 lanes (lines with 1-3 observations, kept in tiles of their own by the
// packer) take 2 or 4 rounds: lane j evaluates angles j, j + kk, ...  Same expressions as line_trig().
__device__ __forceinline__ void seg_line_trig(const double u[4], const SegCtx& s, double trig[7]) {
  double sc[8];      // sin, cos of the four angles
  if (s.rounds == 1) {
    const int a = s.j < 3 ? s.j : 3;
    const double ang = a == 0 ? u[0] : a == 1 ? u[1] : a == 2 ? u[2] : u[3];
    const double sv = sin(ang), cv = cos(ang);
#pragma unroll
    for (int q = 0; q < 4; ++q) { sc[2 * q] = bperm64(sv, s.first4 + 4 * q); sc[2 * q + 1] = bperm64(cv, s.first4 + 4 * q); }
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) sc[q] = 0.0;
    for (int r = 0; r < s.rounds; ++r) {
      const int a = s.j + r * s.kk;                       // this lane's angle of round r (if < 4)
      const int as = a < 3 ? a : 3;
      const double ang = as == 0 ? u[0] : as == 1 ? u[1] : as == 2 ? u[2] : u[3];
      const double sv = sin(ang), cv = cos(ang);
#pragma unroll
      for (int q = 0; q < 4; ++q) {                       // angle q of this run comes from lane q % kk in round q / kk
        const int src = s.first4 + 4 * (q % s.kk);
        const double xs = bperm64(sv, src), xc = bperm64(cv, src);
        if (q / s.kk == r) { sc[2 * q] = xs; sc[2 * q + 1] = xc; }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) trig[q] = sc[q];
  trig[6] = sc[7] / sc[6];
}

// sin / cos table of a candidate line from the table of the accepted line and the step: the LM step of a line parameter is
// small, so  sin(a + e) = sin a cos e + cos a sin e  with sin e, cos e from their Taylor series (|e| <= 2^-4: the terms
// beyond e^11 / e^12 are below 1e-25) costs ~20 fp64 operations per angle where the library's full-range sin + cos cost ~150;
// cot(t + e) = (d cos e - sin e) / (cos e + d sin e).  Every lane updates all four angles of its line (no lane exchange).
// The caller falls back to seg_line_trig (library sin / cos of the new angles) when a step of its wave is larger.
__device__ __forceinline__ void sincos_small(double e, double* se, double* ce) {
  const double z = e * e;
  double sp = -2.5052108385441720e-08;                 // -1/11!
  sp = fma(sp, z, 2.7557319223985893e-06);             //  1/9!
  sp = fma(sp, z, -1.9841269841269841e-04);            // -1/7!
  sp = fma(sp, z, 8.3333333333333332e-03);             //  1/5!
  sp = fma(sp, z, -1.6666666666666666e-01);            // -1/3!
  *se = fma(sp * z, e, e);
  double cp = 2.0876756987868100e-09;                  //  1/12!
  cp = fma(cp, z, -2.7557319223985888e-07);            // -1/10!
  cp = fma(cp, z, 2.4801587301587302e-05);             //  1/8!
  cp = fma(cp, z, -1.3888888888888889e-03);            // -1/6!
  cp = fma(cp, z, 4.1666666666666664e-02);             //  1/4!
  cp = fma(cp, z, -0.5);
  *ce = fma(cp, z, 1.0);
}
__device__ __forceinline__ void line_trig_step(const double trig0[7], const double e[4], double trig[7]) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double se, ce;
    sincos_small(e[q], &se, &ce);
    const double s0 = trig0[2 * q], c0 = trig0[2 * q + 1];
    trig[2 * q] = fma(s0, ce, c0 * se);
    trig[2 * q + 1] = fma(c0, ce, -(s0 * se));
  }
  double se, ce;
  sincos_small(e[3], &se, &ce);
  const double d = trig0[6];
  trig[6] = fma(d, ce, -se) / fma(d, se, ce);
}

// One or two waves per chunk workgroup (blockDim.x = 64 or 128): wave w takes the tiles tile_begin + w, + nw, ...
#if defined(SLSLAM_BACKSUB_WAVES_PER_EU)       // occupancy experiments (SLSLAM_EXTRA_FLAGS=-DSLSLAM_BACKSUB_WAVES_PER_EU=3)
#define SLS_BACKSUB_OCC __attribute__((amdgpu_waves_per_eu(SLSLAM_BACKSUB_WAVES_PER_EU, SLSLAM_BACKSUB_WAVES_PER_EU)))
#else
#define SLS_BACKSUB_OCC
#endif
__global__ __launch_bounds__(128) SLS_BACKSUB_OCC void k_backsub(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  SLS_K1_STAMP_INIT;
  // (experiment SLSLAM_DEBUG_ABLATE bit 15: the back-substitution walks the chunk array from its END - it then starts on the data the elimination
  // sweep touched last, which the memory-side cache still holds)
  const Chunk ck = p.chunks[(pol.debug_flags & 32768) ? gridDim.x - 1 - blockIdx.x : blockIdx.x];
  if (ck.win < 0) return;                              // an unused entry of a refillable batch's chunk array (lba_types.h)
  SLS_K1_WALL(28);
  const WinDesc wd = p.wins[ck.win];
  const LMState* st = p.state + ck.win;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const int n = wd.n;
  const bool refresh_table = (st->n_success & 15) == 15;
  const TileReq rq0 = request_tile(p, ck.tile_begin + wave, ck.tile_end, lane);      // the first tile's context travels while the tables are built
  double* bstab = smem;
  double* candtab = bstab + wd.C * kBsTab;
  double* red = candtab + wd.C * kCandTab;             // [2][4] per-wave sums
  signed char* camcf = (signed char*)(red + 8);
  // the first tile's observations are requested before the tables are built (see k_eliminate_grouped: a chunk's set-up is a chain of
  // round trips to memory, and this link can overlap the next ones)
  TileCtx nxt = resolve_tile(rq0);
  ObsPref pfn;
  prefetch_obs<true>(p, nxt, cur, wd.obs_off, pfn);
  for (int c = tid; c < wd.C; c += 64 * nw) {
    // accepted pose: R, t and the camera step folded through JL and the Jacobi scale;
    // candidate pose (written by k_reduced_solve): R, t for the cost at the candidate point
    const double* x = p.cam_x + ((long long)(wd.cam_off + c) * 2 + cur) * kCamRec;
    const double* xc = p.cam_x + ((long long)(wd.cam_off + c) * 2 + (1 - cur)) * kCamRec;
    double R[9], JL[9], tacc[3], Rc[9], tc[3];
    if (p.cam_tab) {                                     // both points' tables are in memory (load_cam_table, k_reduced_solve)
      const double* ga = p.cam_tab + ((long long)(wd.cam_off + c) * 2 + cur) * kCamTab;
      const double* gc = p.cam_tab + ((long long)(wd.cam_off + c) * 2 + (1 - cur)) * kCamTab;
      for (int q = 0; q < 9; ++q) { R[q] = ga[q]; JL[q] = ga[9 + q]; Rc[q] = gc[q]; }
      for (int q = 0; q < 3; ++q) { tacc[q] = ga[18 + q]; tc[q] = gc[18 + q]; }
    } else {
      double w[3] = { x[0], x[1], x[2] };
      cam_prepare<double>(w, R, JL);
      tacc[0] = x[3]; tacc[1] = x[4]; tacc[2] = x[5];
      double wc[3] = { xc[0], xc[1], xc[2] };
      cam_rotation<double>(wc, Rc);
      tc[0] = xc[3]; tc[1] = xc[4]; tc[2] = xc[5];
    }
    double* bt = bstab + c * kBsTab;
    for (int q = 0; q < 9; ++q) bt[q] = R[q];
    bt[9] = tacc[0]; bt[10] = tacc[1]; bt[11] = tacc[2];
    const int cf = p.cam_cf[wd.cam_off + c];
    double yw[3] = { 0, 0, 0 }, yt[3] = { 0, 0, 0 };
    if (cf >= 0) {
      const double* sc = p.cam_scale + (long long)(wd.cam_off + c) * 6;
      const double* y = p.ysys + wd.sys_off + 6 * cf;
      for (int a = 0; a < 3; ++a) { yw[a] = sc[a] * y[a]; yt[a] = sc[3 + a] * y[3 + a]; }
    }
    for (int i = 0; i < 3; ++i) {
      bt[12 + i] = JL[3 * i] * yw[0] + JL[3 * i + 1] * yw[1] + JL[3 * i + 2] * yw[2];   // (JL yw)[i]:  (tau^T JL) . yw = tau . (JL yw)
      bt[15 + i] = yt[i];
    }
    camcf[c] = (signed char)cf;
    double* ct = candtab + c * kCandTab;
    for (int q = 0; q < 9; ++q) ct[q] = Rc[q];
    ct[9] = tc[0]; ct[10] = tc[1]; ct[11] = tc[2];
  }
  __syncthreads();
  SLS_K1_STAMP(8);

  double acc_model = 0.0, acc_dn2 = 0.0, acc_xn2 = 0.0, acc_cost = 0.0;
#if defined(SLSLAM_K1_TIMING) && SLSLAM_K1_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  SLS_K1_STAMP(9);
  for (int t = ck.tile_begin + wave; t < ck.tile_end; t += nw) {
    SLS_PHASE("bs_tile_head");
    const TileCtx tc = nxt;
    const ObsPref pf = pfn;
    const TileReq rq = request_tile(p, t + nw, ck.tile_end, lane);      // resolved at the prefetch below
    const SegCtx sg = make_seg(tc, lane);
    const int j = tc.j, ls = tc.ls, k = tc.k;
    const bool line_ok = tc.line_ok;
    // what the elimination kernel kept for this line at this linearisation point and radius: K = chol(H_ll + D^2)^-1,
    // D^2, g_l (the same values this sweep would recompute); every lane of the line's run reads the same record
    double K[10], D2[4], g[4];
    {
      // (timing experiment 64: the record read from the chunk's first line only - same instructions, L2 hits: is the sweep waiting
      // for these loads?)
      const double* le = p.line_elim + (long long)((SLSLAM_ABLATE & 64) ? 0 : (tc.line_ok ? tc.ls : 0)) * p.line_elim_stride;
#pragma unroll
      for (int q = 0; q < 10; ++q) K[q] = le[q];
#pragma unroll
      for (int q = 0; q < 4; ++q) { D2[q] = le[kLeD2 + q]; g[q] = le[kLeG + q]; }
    }
    LaneBs L;
    double ob[8], wo[4];
    SLS_PHASE("bs_linearise_w");
    lane_linearise_bs(pol, bstab, camcf, j, k, line_ok, tc.lflags, L, ob, wo, pf);
    const bool line_active = L.line_free && k > 0;
    // w = sum_i (Jc_i^T Jl_i)^T y_c[cam_i] = sum_i Jl_i^T (Jc_i y_c)
    double wv[4] = { 0, 0, 0, 0 };
    if (L.valid && L.cf >= 0 && L.line_free) {
#pragma unroll
      for (int a = 0; a < 4; ++a) wv[a] = wo[a];
    }
    // the next tile's loads go out here (the linearisation's temporaries are dead), see prefetch_obs
    SLS_PHASE("bs_prefetch_next");
    nxt = resolve_tile(rq);
    prefetch_obs<true>(p, nxt, cur, wd.obs_off, pfn);
    __builtin_amdgcn_sched_barrier(0);
    seg_sum_n<4, true>(wv, sg);        // (the DPP form of the run total measured slower here: 0.505 -> 0.518 ms; this sweep has no atomics)
    SLS_PHASE("bs_line_step");
    // every lane of the run holds the same H, g, w: all of them take the step (the candidate
    // parameters are needed by every lane below); lane 0 of the run writes and accumulates
    const int lsafe = line_ok ? ls : 0;
    double xn[4] = { pf.u[0], pf.u[1], pf.u[2], pf.u[3] };
    const bool head = line_ok && j == 0;
    if (line_active) {
      // z = K (g - w);  y = K^T z
      const double e0 = g[0] - wv[0], e1 = g[1] - wv[1], e2 = g[2] - wv[2], e3 = g[3] - wv[3];
      const double z0 = K[0] * e0;
      const double z1 = K[1] * e0 + K[2] * e1;
      const double z2 = K[3] * e0 + K[4] * e1 + K[5] * e2;
      const double z3 = K[6] * e0 + K[7] * e1 + K[8] * e2 + K[9] * e3;
      double y[4];
      y[0] = K[0] * z0 + K[1] * z1 + K[3] * z2 + K[6] * z3;
      y[1] = K[2] * z1 + K[4] * z2 + K[7] * z3;
      y[2] = K[5] * z2 + K[8] * z3;
      y[3] = K[9] * z3;
      for (int a = 0; a < 4; ++a) {
        const double v = xn[a] - y[a] * pf.lsc[a];
        const double dd = xn[a] - v;
        if (head) {
          acc_model += 0.5 * y[a] * (g[a] + D2[a] * y[a]);
          acc_dn2 += dd * dd;
          acc_xn2 += v * v;
        }
        xn[a] = v;
      }
    }
    double trig[7];
    SLS_PHASE("bs_candidate_trig");
    {
      // (the step actually taken: new minus old, exact in floating point for small steps)
      const double e[4] = { xn[0] - pf.u[0], xn[1] - pf.u[1], xn[2] - pf.u[2], xn[3] - pf.u[3] };
      const double emax = fmax(fmax(fabs(e[0]), fabs(e[1])), fmax(fabs(e[2]), fabs(e[3])));
      // library sin / cos of the new angles when a step of the wave's lines is large - and on every 16th accepted step of the
      // window, so that the round-off the angle-addition updates leave in the table (a few ulp per accepted step) does not
      // accumulate over a long solve (max_num_iterations >> 10); idle lanes carry no line and do not vote
      if (__any(line_ok && !(emax <= 0.0625)) || refresh_table) seg_line_trig(xn, sg, trig);
      else line_trig_step(pf.trig, e, trig);
    }
    SLS_PHASE("bs_store_candidate");
    if (head) {
      double* xc = p.line_x + line_rec(p, ls, (1 - cur));
      for (int a = 0; a < 4; ++a) xc[a] = xn[a];
      for (int a = 0; a < 7; ++a) xc[4 + a] = trig[a];
    }
    SLS_PHASE("bs_candidate_cost");
    // cost of this observation at the candidate point (cameras from the reduced solve, line from above)
    {
      const double* ct = candtab + L.cam * kCandTab;
      double R[9], tt[3], cp[3], dv[3], r[4], c;
#pragma unroll
      for (int q = 0; q < 9; ++q) R[q] = ct[q];
      tt[0] = ct[9]; tt[1] = ct[10]; tt[2] = ct[11];
      line_points<double>(trig, cp, dv);
      obs_residual<double>(R, tt, cp, dv, ob, pol.baseline, r);
      huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &c);
      if (L.kept) acc_cost += c;
    }
  }
  SLS_PHASE("epilogue");
  SLS_K1_STAMP(10);
  const double m = wave_sum(acc_model), d = wave_sum(acc_dn2), x = wave_sum(acc_xn2), cs = wave_sum(acc_cost);
  if (nw > 1) {
    if (lane == 0) { red[4 * wave] = m; red[4 * wave + 1] = d; red[4 * wave + 2] = x; red[4 * wave + 3] = cs; }
    __syncthreads();
    if (tid == 0) {
      double* bp = p.bs_part + (long long)ck.id * kBsStride;
      bp[kBsModel] = red[0] + red[4]; bp[kBsDn2] = red[1] + red[5]; bp[kBsXn2] = red[2] + red[6];
      p.cost_part[ck.id] = red[3] + red[7];
    }
  } else if (lane == 0) {
    double* bp = p.bs_part + (long long)ck.id * kBsStride;
    bp[kBsModel] = m; bp[kBsDn2] = d; bp[kBsXn2] = x;
    p.cost_part[ck.id] = cs;
  }
  SLS_K1_STAMP(11);
  SLS_K1_WALL(29);
}

// ------------------------------------------------------------------------------------------
// Kernel 3, variant B (Policy.store_f = 1): back-substitution from the F blocks and per-line factors
// the elimination kernel spilled to HBM,  y_l = K^T (u - sum_i F_i^T y_c,i).  No second
// linearisation, but +192 B per coupled observation written and read back: HBM traffic of the
// dominant kernel becomes ~4x its algorithmic bytes.  Measured (profiles/): a wash in wall time
// against variant A, so A (recompute, traffic == algorithmic) is the default.
__host__ __device__ inline int lds_doubles_backsub_stream(int C, int n) { return n + (C + 1) / 2 + 2; }

__global__ __launch_bounds__(64) void k_backsub_stream(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const Chunk ck = p.chunks[blockIdx.x];
  if (ck.win < 0) return;                              // an unused entry of a refillable batch's chunk array (lba_types.h)
  const WinDesc wd = p.wins[ck.win];
  const LMState* st = p.state + ck.win;
  if (st->status != kRunning) return;
  const int cur = st->cur;
  const int n = wd.n;
  double* yc = smem;
  int* camcf = (int*)(yc + n);
  for (int q = lane; q < n; q += 64) yc[q] = p.ysys[wd.sys_off + q];
  for (int c = lane; c < wd.C; c += 64) camcf[c] = p.cam_cf[wd.cam_off + c];
  __syncthreads();

  double acc_model = 0.0, acc_dn2 = 0.0, acc_xn2 = 0.0;
  for (int t = ck.tile_begin; t < ck.tile_end; ++t) {
    const TileCtx tc = fetch_tile(p, t, ck.tile_end, lane);
    const SegCtx sg = make_seg(tc, lane);
    const int j = tc.j, ls = tc.ls, o0 = tc.o0, k = tc.k, lflags = tc.lflags;
    const bool line_ok = tc.line_ok;
    const bool line_active = line_ok && !(lflags & 1) && k > 0;
    const bool valid = line_ok && j < k;
    double v[4] = { 0, 0, 0, 0 };
    if (valid && line_active) {
      const long long o = (long long)o0 + j;
      const int cf = camcf[p.ob_cam[o]];
      if (cf >= 0) {
        const double* y = yc + 6 * cf;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double ya = y[a];
          const double2 f01 = reinterpret_cast<const double2*>(p.fstore)[(long long)(2 * a) * p.ob_stride + o];
          const double2 f23 = reinterpret_cast<const double2*>(p.fstore)[(long long)(2 * a + 1) * p.ob_stride + o];
          v[0] += f01.x * ya; v[1] += f01.y * ya; v[2] += f23.x * ya; v[3] += f23.y * ya;
        }
      }
    }
    seg_sum_n<4>(v, sg);
    if (line_ok && j == 0) {
      const double* xl = p.line_x + line_rec(p, ls, cur);
      double* xc = p.line_x + line_rec(p, ls, (1 - cur));
      double xn[4] = { xl[0], xl[1], xl[2], xl[3] };
      if (line_active) {
        const double* le = p.line_elim + (long long)ls * p.line_elim_stride;
        double K[10];
#pragma unroll
        for (int q = 0; q < 10; ++q) K[q] = le[q];
        const double z0 = le[kLeU] - v[0], z1 = le[kLeU + 1] - v[1], z2 = le[kLeU + 2] - v[2], z3 = le[kLeU + 3] - v[3];
        double y[4];
        y[0] = K[0] * z0 + K[1] * z1 + K[3] * z2 + K[6] * z3;
        y[1] = K[2] * z1 + K[4] * z2 + K[7] * z3;
        y[2] = K[5] * z2 + K[8] * z3;
        y[3] = K[9] * z3;
        const double* lsc = p.line_scale + (long long)ls * 4;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          acc_model += 0.5 * y[a] * (le[kLeG + a] + le[kLeD2 + a] * y[a]);
          const double vv = xn[a] - y[a] * lsc[a];
          const double dd = xn[a] - vv;
          acc_dn2 += dd * dd;
          acc_xn2 += vv * vv;
          xn[a] = vv;
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) xc[a] = xn[a];
    }
  }
  const double m = wave_sum(acc_model), d = wave_sum(acc_dn2), x = wave_sum(acc_xn2);
  if (lane == 0) {
    double* bp = p.bs_part + (long long)ck.id * kBsStride;
    bp[kBsModel] = m; bp[kBsDn2] = d; bp[kBsXn2] = x;
  }
}

// ------------------------------------------------------------------------------------------
// Kernel 4: sin/cos table of one parameter buffer of every line; lane <-> line.
// which: 0 = the accepted buffer (initialisation), 1 = the candidate buffer.
__global__ __launch_bounds__(256) void k_line_trig(BatchPtrs p, int which) {
  const int ls = blockIdx.x * blockDim.x + threadIdx.x;
  if (ls >= p.nline) return;
  const int lw = p.line_win[ls];
  if (lw < 0) return;                                  // a record beyond the batch's lines (room for refills)
  const LMState* st = p.state + lw;
  if (st->status != kRunning) return;
  const int buf = which ? 1 - st->cur : st->cur;
  double* rec = p.line_x + line_rec(p, ls, buf);
  double u[4] = { rec[0], rec[1], rec[2], rec[3] }, trig[7];
  line_trig<double>(u, trig);
  for (int q = 0; q < 7; ++q) rec[4 + q] = trig[q];
}

// ------------------------------------------------------------------------------------------
// Kernel 5: cost of the reduced program at the candidate point (residuals only).
__host__ __device__ inline int lds_doubles_cost(int C, int n) { return C * kCamTab + (n > 0 ? n : 6) + (C + 7) / 8; }

__global__ __launch_bounds__(64) void k_candidate_cost(BatchPtrs p, Policy pol) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const Chunk ck = p.chunks[blockIdx.x];
  if (ck.win < 0) return;                              // an unused entry of a refillable batch's chunk array (lba_types.h)
  const WinDesc wd = p.wins[ck.win];
  const LMState* st = p.state + ck.win;
  if (st->status != kRunning) return;
  const int cand = 1 - st->cur;
  double* camtab = smem;
  double* camscale = camtab + wd.C * kCamTab;
  signed char* camcf = (signed char*)(camscale + (wd.n > 0 ? wd.n : 6));
  load_cam_table<false>(p, wd, cand, lane, camtab, camscale, camcf, true);
  __syncthreads();
  double acc = 0.0;
  for (int t = ck.tile_begin; t < ck.tile_end; ++t) {
    const TileCtx tc = fetch_tile(p, t, ck.tile_end, lane);
    const int j = tc.j, ls = tc.ls, o0 = tc.o0, k = tc.k;
    const bool line_ok = tc.line_ok;
    const bool valid = line_ok && j < k;
    const int o = valid ? o0 + j : wd.obs_off;
    double ob[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double2 e = reinterpret_cast<const double2*>(p.ob)[(long long)q * p.ob_stride + o];
      ob[2 * q] = e.x; ob[2 * q + 1] = e.y;
    }
    const int cam = p.ob_cam[o];
    const int lsafe = line_ok ? ls : 0;
    const double* lrec = p.line_x + line_rec(p, lsafe, cand);
    double trig[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) trig[q] = lrec[4 + q];
    const bool line_free = !(p.line_flags[lsafe] & 1);
    const double* ct = camtab + cam * kCamTab;
    double R[9], tt[3], cp[3], dv[3], r[4], c;
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = ct[q];
    tt[0] = ct[18]; tt[1] = ct[19]; tt[2] = ct[20];
    line_points<double>(trig, cp, dv);
    obs_residual<double>(R, tt, cp, dv, ob, pol.baseline, r);
    huber_scale<double>(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3], pol.huber_delta, &c);
    const bool kept = valid && !(camcf[cam] < 0 && !line_free);
    if (kept) acc += c;
  }
  const double s = wave_sum(acc);
  if (lane == 0) p.cost_part[ck.id] = s;
}

// ------------------------------------------------------------------------------------------
// Kernel 6: the trust-region bookkeeping of one window (one lane) after an LM iteration.

// After the initial evaluation (Ceres: cost, gradient and column norms at x0): one wave per window,
// lane <-> camera column; the chunk partials of a column are summed in chunk order with the loads in flight together.
__global__ __launch_bounds__(64) void k_lm_init(BatchPtrs p, Policy pol) {
  const int w = blockIdx.x, lane = threadIdx.x;
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
  if (st->status != kRunning) return;
  const int n = wd.n, nsys = sys_doubles(n);
  const long long slab0 = wd.nchunks > 0 ? wd.slab_off : 0;     // a window without lines has no chunk
  const long long sstride = (long long)nsys + kSlabScalars;
  double cost = 0.0, fixed = 0.0, gmax = 0.0, xn2 = 0.0;
  for (int c = lane; c < wd.nchunks; c += 64) {
    const double* sc = p.slab + slab0 + c * sstride + nsys;
    cost += sc[kScCost]; fixed += sc[kScFixedCost]; gmax = fmax(gmax, sc[kScGradMaxLine]); xn2 += sc[kScXn2Line];
  }
  // cost / fixed cost: chunk order matters for bitwise reproducibility -> ordered sum by lane 0 below when > 64 chunks never happens;
  // with <= 64 chunks every lane holds at most one term and the xor-tree below is a fixed order
  for (int q = lane; q < 6 * wd.C; q += 64) {
    const int c = q / 6, a = q - 6 * c;
    const int cf = p.cam_cf[wd.cam_off + c];
    double sc = 1.0;
    if (cf >= 0) {
      const double* src = p.slab + slab0 + cf * kCamAcc;
      double g = 0.0, h = 0.0;
      for (int k0 = 0; k0 < wd.nchunks; k0 += 8) {
        double vg[8], vh[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool in = k0 + u < wd.nchunks;
          const double* sl = src + (long long)(in ? k0 + u : k0) * sstride;
          vg[u] = in ? sl[kRecG + a] : 0.0; vh[u] = in ? sl[kRecH + a] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { g += vg[u]; h += vh[u]; }
      }
      const double x = p.cam_x[((long long)(wd.cam_off + c) * 2 + st->cur) * kCamRec + a];
      gmax = fmax(gmax, fabs(g));
      xn2 += x * x;
      if (pol.jacobi_scaling) sc = 1.0 / (1.0 + sqrt(h));
    }
    p.cam_scale[(long long)(wd.cam_off + c) * 6 + a] = sc;
  }
  cost = wave_sum(cost); fixed = wave_sum(fixed); xn2 = wave_sum(xn2); gmax = wave_max(gmax);
  if (lane != 0) return;
  st->cost = cost; st->fixed_cost = fixed; st->initial_cost = cost + fixed; st->min_cost = cost + fixed;
  st->x_norm = sqrt(xn2);
  st->grad_max = gmax;
  const double g0 = gmax > 1e-12 ? gmax : 1e-12;
  st->abs_grad_tol = pol.gradient_tolerance * g0;
  st->need_grad_check = 0;
  st->fresh = 0;
  if (wd.nfree_params == 0) { st->status = 2 /* FUNCTION_TOLERANCE: no free blocks */; return; }
  if (!isfinite(cost)) { st->status = 4; return; }
  if (gmax <= st->abs_grad_tol) { st->status = 1; return; }
  IterRec rec;
  rec.pad = 0;
  rec.iteration = 0; rec.step_is_valid = 0; rec.step_is_successful = 0;
  rec.cost = cost + fixed; rec.cost_change = 0; rec.gradient_max_norm = gmax; rec.step_norm = 0;
  rec.relative_decrease = 0; rec.trust_region_radius = st->radius; rec.model_cost_change = 0;
  push_trace(p, w, st, rec);
  if (st->iter >= pol.max_num_iterations) st->status = 0;
}

// One trust-region step's bookkeeping (Ceres 1.7 TrustRegionMinimizer; policy table in DESIGN.md section 5): given the cost at the
// candidate point and the step statistics, accept or reject, move the radius, record the iteration, test the stopping rules.
__device__ __forceinline__ void lm_step(BatchPtrs& p, const Policy& pol, int w, LMState* st, double new_cost, double model,
                                        double dn2, double xn2) {
  IterRec rec;
  rec.pad = 0;
  const double cost = st->cost;
  rec.iteration = st->iter + 1;
  rec.step_is_valid = 0; rec.step_is_successful = 0;
  rec.model_cost_change = model;
  rec.cost_change = 0; rec.step_norm = 0; rec.relative_decrease = 0;
  rec.gradient_max_norm = st->grad_max;
  bool valid = !st->solve_failed && !(model < 0.0);
  if (!isfinite(new_cost)) new_cost = 1.7976931348623157e308;
  if (!valid) {
    if (++st->n_invalid >= pol.max_invalid) { st->status = 4; return; }
  } else {
    st->n_invalid = 0;
    rec.step_is_valid = 1;
    rec.step_norm = sqrt(dn2);
    if (rec.step_norm <= pol.parameter_tolerance * (st->x_norm + pol.parameter_tolerance)) { st->status = 3; return; }
    rec.cost_change = cost - new_cost;
    if (fabs(rec.cost_change) < pol.function_tolerance * cost) { st->status = 2; return; }
    rec.relative_decrease = rec.cost_change / model;
    rec.step_is_successful = rec.relative_decrease > pol.min_relative_decrease;
  }
  if (rec.step_is_successful) {
    st->n_success++;
    const double q = 2.0 * rec.relative_decrease - 1.0;
    double f = 1.0 - q * q * q;
    if (f < 1.0 / 3.0) f = 1.0 / 3.0;
    st->radius = fmin(st->radius / f, pol.max_radius);
    st->decrease_factor = 2.0;
    st->cur = 1 - st->cur;
    st->cost = new_cost;
    st->x_norm = sqrt(xn2);
    st->need_grad_check = 1;      // the next linearisation supplies the gradient at the new point
    st->same_point = 0;
  } else {
    // only the radius changes: gradient and column norms of the cameras stay valid - except after the very first
    // sweep of a solve, whose camera entries are in unscaled coordinates (it doubled as the initial evaluation)
    st->same_point = st->iter > 0 ? 1 : 0;
    st->n_unsuccess++;
    if (rec.step_is_valid) { st->radius = st->radius / st->decrease_factor; st->decrease_factor *= 2.0; }
    else st->radius *= 0.5;
  }
  rec.cost = st->cost + st->fixed_cost;
  rec.trust_region_radius = st->radius;
  if (rec.cost < st->min_cost) st->min_cost = rec.cost;
  push_trace(p, w, st, rec);
  st->iter = rec.iteration;
  atomicAdd(p.iter_counter, 1ULL);   // one trust-region step (successful or not), as slam.cpp:949-950 counts
  if (st->radius < pol.min_radius) { st->status = 5; return; }
  if (st->iter >= pol.max_num_iterations) { st->status = 0; return; }
  atomicAdd(p.active_counter, 1u);    // still running: lets the host stop enqueueing long solves early
}

// Small batches (a window cut into ~50 chunks, this kernel on the latency path of every iteration): one wave per window,
// the lanes fetch the window's chunk partials together, fixed-shape tree sum, lane 0 does the bookkeeping.
__global__ __launch_bounds__(64) void k_lm_update_wave(BatchPtrs p, Policy pol) {
  const int w = blockIdx.x, lane = threadIdx.x;
  if (w >= p.nwin) return;
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
  if (st->status != kRunning) return;
  if (wd.nchunks <= 8) {                                  // the sums of k_lm_update, in its order (see k_slab_reduce)
    if (lane == 0) {
      double new_cost = 0.0, model = st->cam_model, dn2 = st->cam_dn2, xn2 = st->cam_xn2;
      for (int c = 0; c < wd.nchunks; ++c) {
        new_cost += p.cost_part[wd.chunk_off + c];
        const double* bp = p.bs_part + (long long)(wd.chunk_off + c) * kBsStride;
        model += bp[kBsModel]; dn2 += bp[kBsDn2]; xn2 += bp[kBsXn2];
      }
      lm_step(p, pol, w, st, new_cost, model, dn2, xn2);
    }
    return;
  }
  double new_cost = 0.0, model = 0.0, dn2 = 0.0, xn2 = 0.0;
  for (int c = lane; c < wd.nchunks; c += 64) {
    new_cost += p.cost_part[wd.chunk_off + c];
    const double* bp = p.bs_part + (long long)(wd.chunk_off + c) * kBsStride;
    model += bp[kBsModel]; dn2 += bp[kBsDn2]; xn2 += bp[kBsXn2];
  }
  new_cost = wave_sum(new_cost); model = wave_sum(model); dn2 = wave_sum(dn2); xn2 = wave_sum(xn2);
  if (lane == 0) lm_step(p, pol, w, st, new_cost, model + st->cam_model, dn2 + st->cam_dn2, xn2 + st->cam_xn2);
}
// Large batches (a handful of chunks per window): one lane per window.
__global__ __launch_bounds__(64) void k_lm_update(BatchPtrs p, Policy pol) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= p.nwin) return;
  const WinDesc wd = p.wins[w];
  LMState* st = p.state + w;
  if (st->status != kRunning) return;
  double new_cost = 0.0, model = st->cam_model, dn2 = st->cam_dn2, xn2 = st->cam_xn2;
  for (int c = 0; c < wd.nchunks; ++c) {
    new_cost += p.cost_part[wd.chunk_off + c];
    const double* bp = p.bs_part + (long long)(wd.chunk_off + c) * kBsStride;
    model += bp[kBsModel]; dn2 += bp[kBsDn2]; xn2 += bp[kBsXn2];
  }
  lm_step(p, pol, w, st, new_cost, model, dn2, xn2);
}

// ------------------------------------------------------------------------------------------
// Test hook: per-observation robustified residuals / Jacobians (unscaled) in the caller's order.
__global__ __launch_bounds__(64) void k_debug_linearise(BatchPtrs p, Policy pol, int w, const int* ob_orig,
                                                        double* out_r, double* out_jc, double* out_jl, double* out_cost) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x;
  const WinDesc wd = p.wins[w];
  const LMState* st = p.state + w;
  const int cur = st->cur;
  double* camtab = smem;
  double* camscale = camtab + wd.C * kCamTab;
  signed char* camcf = (signed char*)(camscale + (wd.n > 0 ? wd.n : 6));
  load_cam_table<true>(p, wd, cur, lane, camtab, camscale, camcf, true);
  __syncthreads();
  double acc = 0.0;
  for (int l = 0; l < wd.L; ++l) {
    const int ls = wd.line_off + l;
    const int o0 = p.line_ptr[ls], k = p.line_ptr[ls + 1] - o0;
    for (int j0 = 0; j0 < k; j0 += 64) {
      const int j = j0 + lane;
      LaneLin L;
      double ob[8];
      lane_linearise<false>(p, pol, camtab, camscale, camcf, ls, j, k, o0, true, p.line_flags[ls], cur, wd.obs_off, L, ob);
      if (L.valid) {
        const int orig = ob_orig[o0 + j];
        for (int q = 0; q < 4; ++q) out_r[4 * (long long)orig + q] = L.rs[q];
        for (int q = 0; q < 24; ++q) out_jc[24 * (long long)orig + q] = L.Jc[q];
        for (int q = 0; q < 16; ++q) out_jl[16 * (long long)orig + q] = L.Jl[q];
        acc += L.cost;
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) *out_cost = acc;
}

// Restores the initial parameters and LM state of every window (repeated solves of the same
// inputs).  lane <-> parameter block / window.
__global__ __launch_bounds__(256) void k_reset(BatchPtrs p, Policy pol) {
  // thread <-> one LINE (its whole record in the accepted buffer - the four parameters and their sin / cos table, what k_line_trig used to add
  // in a launch of its own at the start of every solve - and the parameters in the candidate buffer: consecutive threads write consecutive 88-byte
  // records, whole cache lines instead of 32-byte pieces of them: 157 + 88 us -> one launch), then one thread per camera parameter, per window state
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long ncp = 6LL * p.ncam, nlp = (long long)p.nline;
  if (i < nlp) {
    const long long ls = i;
    if (p.line_win[ls] >= 0) {                           // (a record beyond the batch's lines - room for refills - is left alone)
      const double2* u0 = reinterpret_cast<const double2*>(p.line_u0 + 4 * ls);
      const double2 a01 = u0[0], a23 = u0[1];
      double u[4] = { a01.x, a01.y, a23.x, a23.y }, trig[7];
      line_trig<double>(u, trig);
      double* r0 = p.line_x + line_rec(p, ls, 0);       // (88-byte records: 8-byte aligned only)
      double* r1 = p.line_x + line_rec(p, ls, 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) { r0[q] = u[q]; r1[q] = u[q]; }
#pragma unroll
      for (int q = 0; q < 7; ++q) r0[4 + q] = trig[q];
    }
  } else if (i < nlp + ncp) {
    const long long q = i - nlp, c = q / 6;
    const double v = p.cam_x0[q];
    p.cam_x[c * 2 * kCamRec + (q - 6 * c)] = v; p.cam_x[c * 2 * kCamRec + kCamRec + (q - 6 * c)] = v;
  } else if (i < nlp + ncp + p.nwin) {
    LMState* st = p.state + (i - nlp - ncp);
    LMState z;
    z.radius = pol.initial_radius; z.decrease_factor = 2.0; z.cost = 0; z.x_norm = 0; z.fixed_cost = 0;
    z.initial_cost = 0; z.min_cost = 0; z.abs_grad_tol = 0; z.grad_max = 0; z.cam_model = 0; z.cam_dn2 = 0; z.cam_xn2 = 0;
    z.status = kRunning; z.cur = 0; z.iter = 0; z.n_success = 0; z.n_unsuccess = 0; z.n_invalid = 0;
    z.solve_failed = 0; z.need_grad_check = 0; z.ntrace = 0; z.same_point = 0; z.fresh = 1; z.pad = 0;
    *st = z;
  }
}

// Gathers the accepted parameters of every window into the caller's layout
// [6C | 4L] per window, windows concatenated.  lane <-> parameter block.
// A window that ended in NUMERICAL_FAILURE hands back its INITIAL values: Ceres leaves the user's parameters untouched then (the host
// mirror of a batch keeps a copy for that; a batch built on the device has none, and a collective on the exported vector wants them too).
__global__ __launch_bounds__(256) void k_export(BatchPtrs p, const long long* win_param_off, const int* cam_win,
                                                const int* line_orig, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < p.ncam) {
    const int w = cam_win[i];
    if (w < 0) return;                                 // a record beyond the batch's cameras (room for refills)
    const WinDesc wd = p.wins[w];
    const int cur = p.state[w].cur;
    const double* x = p.state[w].status == kNumericalFailure ? p.cam_x0 + (long long)i * kCamRec : p.cam_x + ((long long)i * 2 + cur) * kCamRec;
    double* o = out + win_param_off[w] + 6 * (long long)(i - wd.cam_off);
    for (int a = 0; a < 6; ++a) o[a] = x[a];
  } else if (i < p.ncam + p.nline) {
    const int ls = i - p.ncam;
    const int w = p.line_win[ls];
    if (w < 0) return;
    const WinDesc wd = p.wins[w];
    const int cur = p.state[w].cur;
    const double* x = p.state[w].status == kNumericalFailure ? p.line_u0 + 4LL * ls : p.line_x + line_rec(p, ls, cur);
    double* o = out + win_param_off[w] + 6 * (long long)wd.C + 4 * (long long)line_orig[ls];
    for (int a = 0; a < 4; ++a) o[a] = x[a];
  }
}

}  // namespace slslam
#endif  // SLSLAM_LBA_KERNELS_H_
