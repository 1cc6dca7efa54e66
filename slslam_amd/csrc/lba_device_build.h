// slslam_amd/csrc/lba_device_build.h — the LBAProblem::build stage ON THE DEVICE (round 6).
//
// What the reference does per window before the solve - wire M residual blocks from five arrays, src/lba_problem.cpp:54-93, fed by
// src/slam.cpp:899-921 - is lba_pack.cpp::pack_window + lba_api.hip::plan_layout / fill_window on the host: validation, counting sorts,
// grouping by first free camera, best-fit bin packing of the lines into 16-lane rows, lane layout, tile descriptors, chunk cuts.  For a
// STREAM of windows (BASELINE config 4) that host work bounded the product (17.5 ms of packing on 16 host threads per 1024-window batch
// against 15 ms of GPU).  Here the same build runs as four kernels on the arrays as the caller holds them:
//
//   k_ingest        reads the callers' arrays where the copy engine put them in HBM (or, scattered page-locked arrays: few workgroups read
//                   them straight from PINNED host memory, 57 GB/s with 32 workgroups, tools/micro/zero_copy_bench.hip) - observations with the
//                   NaN / Inf test, the index arrays narrowed to one 32-bit word per observation with the range test, cameras and lines
//   k_build_lines   one workgroup per window: counting, constness, free-camera masks, the counting sort as a bitonic sort of
//                   (cell | line) keys
//   k_build_rows    TWO WAVES per window, one per length class (four windows to a CU): the best-fit packing of lines into rows - inherently
//                   sequential: a wave walks its lines, the 17 open-row lists live one per lane (ballots pick the list, the rows of a list
//                   are chained through LDS)
//   k_build_order   one workgroup per window: row order, line order, line pointers, the stable sort of every line's observations by camera
//   k_build_layout  one workgroup: prefix sums over the windows, the (graded) chunk cuts and the dispatch order of plan_layout, the
//                   fit test against the room the batch's arrays have
//   k_build_tiles   thread <-> tile over inputs staged in LDS, then wave <-> tile for the stable descriptor sort: lane map, skew flags,
//                   line descriptors (sorted per tile for the grouped sweep), pair items
//
// The host packer stays the SPECIFICATION: tests/test_gpu_device_build.py compares every array this file emits with pack_window's,
// byte for byte (tests/golden/packer_digest.json included), so a window's solved bytes do not depend on who built it.
// A window this path cannot take (bad index, NaN, a camera that sees a line twice, more than 64 observations of a line, more than
// 20 free cameras) is flagged in BuildWin.status, emitted EMPTY (nothing of it is launched) and left to the caller's host path.
#ifndef SLSLAM_LBA_DEVICE_BUILD_H_
#define SLSLAM_LBA_DEVICE_BUILD_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lba_types.h"
#include "lba_kernels.h"
#include "lba_eliminate_grouped_maps.h"
#include "lba_eliminate_mfma_maps.h"

namespace slslam {

// One per window of a refill, written by the host (pointers the DEVICE can read: pinned host memory or device memory).
struct RawWin {
  const int* cam;             // [M] camera_index, or nullptr when `packed` is given
  const int* line;            // [M] line_index
  const int* fixed;           // [2M] fixed_index
  const uint32_t* packed;     // [M] line | camera << 16 | camera constant << 24 | line constant << 25 (the narrowed form), or nullptr
  const double* obs;          // [8M]
  const double* params_in;    // [6C + 4L] the initial parameters (where the ingest reads them)
  double* params;             // [6C + 4L] where the solved parameters go when they are written in place (the caller's array), or null
  long long param_off;        // offset of the window in the batch's exported parameter vector
  int C, L, M;
  int cam_off, line_off, obs_off;     // the window's place in the batch arrays
  int pad;
};

enum { kBuildInvalid = 1, kBuildHostPath = 2, kBuildNoFit = 4 };     // BuildWin.status bits
struct BuildWin {
  int status;                 // 0, or why the window was not built (emitted empty)
  int Cf, ntiles, nitems, nfree_params, nkept;
  int nchunks, graded;        // (k_build_layout) the cut: chunk count and, for graded sizes, the number of slot rounds
};

struct BuildLine;
// What the stages of a window's build hand to each other (device only).
struct BuildMid {
  unsigned long long freeset, constset;   // cameras: free (used and not constant) | constant
  int ok;                     // 0: the window was flagged, the later stages skip it
  int Cf, free_lines, nitems;
  int n_long, n_cls0;         // (k_build_lines) lines with more than 16 observations | of at least 4 lanes among the others: their runs in the packing order
  int nrows, ntr, ntp, cls_row[4];        // (k_build_rows) rows made | row entries and tile ranges of the long lines | row range of each length class
};

// Everything the build kernels touch, by value.
struct BuildPtrs {
  const RawWin* raw;
  BuildWin* bw;
  int nwin;
  int grouping;
  // ingest destinations
  double* ob_raw;             // [8 nobs] observations in the caller's order
  int obs_in_place;           // the observations stay where RawWin.obs has them (device memory): the ingest only tests them
  uint32_t* raw_idx;          // [nobs] narrowed indices
  double* line_raw;           // [4 nline] line parameters in the caller's order
  // per-line scratch (sorted order)
  uint8_t* lflags;            // bit 0: first line of a row entry, bit 1: first line of a tile
  uint32_t* fmask;            // free cameras that see the line
  int* line_pos;              // [nline] caller's line -> sorted position in its window (the export in the caller's order), may be null
  // between the stages of a window's build (k_build_lines -> k_build_rows -> k_build_order), at the window's line offset
  uint32_t* mid_keys;         // [nline] the lines in packing order: sort key << 16 | line
  BuildLine* mid_li;          // [nline] per line: free-camera mask | flags, lanes, pair items
  uint4* mid_rows;            // [nline] the rows
  uint16_t* mid_next;         // [nline] next line of a line's row
  uint16_t* mid_trows;        // [nline + 8 nwin] row entries in tile order (the long lines')
  uint16_t* mid_tptr;         // [nline + 8 nwin] row range of each tile (the long lines')
  BuildMid* mid;              // [nwin]
  // batch arrays (as BatchPtrs / slslam_lba_batch)
  WinDesc* wins; Tile* tiles; Chunk* chunks; uint8_t* items; uint16_t* lane_map; uint32_t* line_desc;
  double* cam_x0; int* cam_cf; int* cam_win;
  double* line_x0; int* line_ptr; int* line_flags; int* line_win; int* line_orig;
  int* ob_cam; int* ob_orig;
  long long* param_off;
  int* item_base;             // [nwin] first pair item of each window (k_build_layout -> k_build_tiles)
  int* totals;                // [8]: tiles | items | chunks | fit (1 ok) | lines | observations
  unsigned long long* dbg;    // timing experiments (tools/build_phases.py): constant-clock (100 MHz) stamps of window 0's build, or null
};

// How k_build_layout cuts (the batch-wide numbers slslam_lba_batch_finalize resolved: plan_layout with frozen = true) and what has to fit.
struct LayoutArgs {
  int chunks_per_window, reproducible, auto_rounds, auto_cpw, elim_waves, elim_mode, equal_chunks;
  int cap_tiles, cap_items, cap_chunks, cap_maxn, slab_sum, slab_sum_image;
  long long cap_slab, cap_sys, slab_sum_stride;
  int nline, nobs;            // totals of the refill (host-known): the end of the line pointers
  int max_free;               // the batch's sweep takes windows with at most this many free cameras
  int sys_map_off[kMaxFreeCams + 2];   // per free-camera count: the table in BatchPtrs.sys_map, or -1
};

// ------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long bad_exponent(double v) {
  const unsigned long long x = (unsigned long long)__double_as_longlong(v);
  return ((x & 0x7ff0000000000000ull) + 0x0010000000000000ull) & 0x8000000000000000ull;     // exponent field all ones <=> NaN / Inf
}

// Few workgroups, every thread with four 16-byte loads in flight: enough to fill the host link, and not a wave slot more than that
// (the solves of the stream's other batches run beside this kernel).
__global__ __launch_bounds__(256) void k_ingest(BuildPtrs P) {
  const int tid = threadIdx.x;
  for (int w = blockIdx.x; w < P.nwin; w += gridDim.x) {
    const RawWin r = P.raw[w];
    unsigned long long bad = 0;
    // observations: 8 M doubles
    {
      const long long n2 = 4LL * r.M;
      double* dst = P.ob_raw + 8LL * r.obs_off;
      if ((reinterpret_cast<uintptr_t>(r.obs) & 15u) == 0) {
        const double2* src = reinterpret_cast<const double2*>(r.obs);
        double2* d2 = reinterpret_cast<double2*>(dst);
        const bool copy = !P.obs_in_place;
        for (long long i = tid; i < n2; i += 1024) {
          double2 v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) if (i + 256 * u < n2) v[u] = src[i + 256 * u];
#pragma unroll
          for (int u = 0; u < 4; ++u) if (i + 256 * u < n2) { if (copy) d2[i + 256 * u] = v[u]; bad |= bad_exponent(v[u].x) | bad_exponent(v[u].y); }
        }
      } else {
        for (long long i = tid; i < 2 * n2; i += 256) { const double v = r.obs[i]; if (!P.obs_in_place) dst[i] = v; bad |= bad_exponent(v); }
      }
    }
    // indices, narrowed: line | camera << 16 | camera constant << 24 | line constant << 25
    {
      uint32_t* dst = P.raw_idx + r.obs_off;
      if (r.packed) {
        for (int i = tid; i < r.M; i += 256) {
          const uint32_t v = r.packed[i];
          if ((int)(v & 0xffffu) >= r.L || (int)((v >> 16) & 0xffu) >= r.C || (v >> 26)) bad = 1;
          dst[i] = v;
        }
      } else {
        const int2* fx = reinterpret_cast<const int2*>(r.fixed);
        const bool fx8 = (reinterpret_cast<uintptr_t>(r.fixed) & 7u) == 0;
        for (int i = tid; i < r.M; i += 256) {
          const int c = r.cam[i], l = r.line[i];
          int f0, f1;
          if (fx8) { const int2 f = fx[i]; f0 = f.x; f1 = f.y; } else { f0 = r.fixed[2 * i]; f1 = r.fixed[2 * i + 1]; }
          if (c < 0 || c >= r.C || l < 0 || l >= r.L) bad = 1;
          dst[i] = ((uint32_t)l & 0xffffu) | ((uint32_t)c & 0xffu) << 16 | (f0 ? 1u << 24 : 0u) | (f1 ? 1u << 25 : 0u);
        }
      }
    }
    // parameters: cameras to their place, lines in the caller's order (k_build_order sorts them)
    for (int i = tid; i < 6 * r.C; i += 256) { const double v = r.params_in[i]; P.cam_x0[6LL * r.cam_off + i] = v; bad |= bad_exponent(v); }
    for (int i = tid; i < 4 * r.L; i += 256) { const double v = r.params_in[6 * r.C + i]; P.line_raw[4LL * r.line_off + i] = v; bad |= bad_exponent(v); }
    if (bad) atomicOr(&P.bw[w].status, (int)kBuildInvalid);
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// LDS of k_build_lines / k_build_order for a window of L lines (Lp = L rounded up to a power of two, at least 256), by phase:
//   A  [Lp]     u32   pass 1: observation count (bits 0-23) | constant (bit 31) per line; then the sort keys; row sort keys; first line of
//                     every row entry; last: position of every line (u16)
//   LI [L]      8 B   pass 1: camera mask (u64); then per line: free-camera mask | flags, lanes, pair items, next line of its row
//   R  [L]      16 B  rows: lanes used, lines, pair items | first, last line | list links | free-camera mask; when the rows are dead: line
//                     pointers [L + 1] and fill counters [L] (u32)
//   T  [2L + 8] u16   tile_rows [L + 2] | tile_ptr [L + 6]
//   part [257]  u32   scan partials
__host__ __device__ inline int build_pow2(int L) { int p = 256; while (p < L) p <<= 1; return p; }
__host__ __device__ inline size_t build_lines_lds_bytes(int L) { return 4 * (size_t)build_pow2(L) + 8 * (size_t)(L > 0 ? L : 1) + 64; }      // k_build_lines: A | CM
__host__ __device__ inline size_t build_lds_bytes(int L) {                                                                                // k_build_order: everything
  const size_t Lq = (size_t)(L > 0 ? L : 1);
  return 4 * (size_t)build_pow2(L) + 8 * Lq + 16 * Lq + 2 * (2 * Lq + 8) + 4 * 260;
}

struct BuildRow { uint8_t used, nl; uint16_t items; uint16_t head, tail; uint16_t lprev, lnext; uint32_t mask; };
static_assert(sizeof(BuildRow) == 16, "row record");
struct BuildLine { uint32_t fm; uint8_t len; uint8_t items; uint16_t next; };       // fm: free-camera mask (bits 0-19) | no observation (bit 30) | constant (bit 31)
static_assert(sizeof(BuildLine) == 8, "line record");
enum { kNoRow = 0xffff, kBuildBuckets = 42 };
enum : uint32_t { kLineConst = 0x80000000u, kLineEmpty = 0x40000000u, kLineMask = 0x000fffffu };

// exclusive prefix sum of a[0 .. n) in place (LDS), 256 threads; returns the total.  part: 257 words of LDS.
__device__ inline unsigned block_scan_excl(unsigned* a, int n, unsigned* part) {
  const int tid = threadIdx.x;
  const int per = (n + 255) / 256, lo = min(n, tid * per), hi = min(n, lo + per);
  unsigned s = 0;
  for (int i = lo; i < hi; ++i) s += a[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) { unsigned run = 0; for (int t = 0; t < 256; ++t) { const unsigned v = part[t]; part[t] = run; run += v; } part[256] = run; }
  __syncthreads();
  unsigned run = part[tid];
  for (int i = lo; i < hi; ++i) { const unsigned v = a[i]; a[i] = run; run += v; }
  const unsigned total = part[256];
  __syncthreads();
  return total;
}

// ascending bitonic sort of a[0 .. n), n a power of two >= 256, 256 threads
__device__ inline void block_bitonic_sort(unsigned* a, int n) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= n; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n; i += 256) {
        const int ix = i ^ j;
        if (ix > i) {
          const unsigned x = a[i], y = a[ix];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[ix] = x; }
        }
      }
      __syncthreads();
    }
}

// key of a line's group (lba_pack.cpp::group_key): first free camera that sees it x 2, + 1 unless its camera range spans more than 8 cameras
__device__ __forceinline__ int build_group_key(uint32_t fm) {
  const unsigned m = (fm & kLineConst) ? 0u : (fm & kLineMask);
  if (!m) return 1000;
  const int a = __ffs((int)m) - 1, hi = 31 - __clz((int)m);
  return 2 * a + ((hi - a + 1) > 8 ? 0 : 1);
}

__global__ __launch_bounds__(256) void k_build_lines(BuildPtrs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char build_smem[];
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const RawWin r = P.raw[w];
  const int C = r.C, L = r.L, M = r.M, grouping = P.grouping;
  const int Lp = build_pow2(L), Lq = L > 0 ? L : 1;
  unsigned* A = reinterpret_cast<unsigned*>(build_smem);
  unsigned long long* CM = reinterpret_cast<unsigned long long*>(A + Lp);
  BuildLine* LI = reinterpret_cast<BuildLine*>(CM);
  __shared__ unsigned long long s_cam_used, s_cam_const;
  __shared__ int s_flags, s_free_lines, s_nitems, s_nlong, s_ncls0;
  __shared__ signed char s_cam_cf[64];
#define BUILD_STAMP(i) do { if (P.dbg && w == 0 && tid == 0) P.dbg[i] = (unsigned long long)wall_clock64(); } while (0)
  BUILD_STAMP(0);
  BuildWin* bw = P.bw + w;
  if (tid == 0) { s_cam_used = 0; s_cam_const = 0; s_flags = 0; s_free_lines = 0; s_nitems = 0; s_nlong = 0; s_ncls0 = 0; }
  for (int l = tid; l < L; l += 256) { A[l] = 0; CM[l] = 0; }
  __syncthreads();
  const bool alive = bw->status == 0 && C <= kMaxCams;        // (else: the ingest flagged the window - a bad index, a non-finite value)
  const uint32_t* ridx = P.raw_idx + r.obs_off;

  // ---- pass 1 over the observations: counts, constness, which cameras see which line (lba_pack.cpp:58-84)
  if (alive) {
    unsigned long long used = 0, cconst = 0;
    for (int i = tid; i < M; i += 256) {
      const uint32_t v = ridx[i];
      const int l = (int)(v & 0xffffu), c = (int)((v >> 16) & 0xffu);
      const unsigned long long bit = 1ull << c;
      const unsigned long long old = atomicOr(&CM[l], bit);
      if (old & bit) atomicOr(&s_flags, 1);                     // a camera sees a line twice: the host path's business
      used |= bit;
      if (v & (1u << 24)) cconst |= bit;
      atomicAdd(&A[l], 1u);
      if (v & (1u << 25)) atomicOr(&A[l], (unsigned)kLineConst);
    }
    if (used) atomicOr(&s_cam_used, used);
    if (cconst) atomicOr(&s_cam_const, cconst);
  }
  __syncthreads();
  BUILD_STAMP(1);
  // free cameras: used and not constant, numbered in camera order
  const unsigned long long freeset = s_cam_used & ~s_cam_const;
  const unsigned long long constset = s_cam_const;
  (void)lane;
  const int Cf = __popcll(freeset);
  if (tid < 64) s_cam_cf[tid] = (tid < C && ((freeset >> tid) & 1ull)) ? (signed char)__popcll(freeset & ((1ull << tid) - 1ull)) : (signed char)-1;
  if (alive) for (int l = tid; l < L; l += 256) if ((A[l] & 0xffffffu) > 64u) atomicOr(&s_flags, 2);     // a line with more than 64 observations
  __syncthreads();
  const bool ok = alive && s_flags == 0 && Cf <= kMaxFreeCams;
  if (!ok) {
    // emitted empty: its records belong to no window (the per-line / per-camera kernels skip them), nothing of it is launched
    // (their line pointers still say where the window's observations start: the last line of the window before reads its end there)
    for (int l = tid; l < L; l += 256) { P.line_win[r.line_off + l] = -1; P.line_ptr[r.line_off + l] = r.obs_off; }
    for (int c = tid; c < C; c += 256) { P.cam_win[r.cam_off + c] = -1; P.cam_cf[r.cam_off + c] = -1; }
    if (tid == 0) {
      P.mid[w].ok = 0;
      if (bw->status == 0) bw->status = kBuildHostPath;
      bw->Cf = 0; bw->ntiles = 0; bw->nitems = 0; bw->nfree_params = 0; bw->nkept = 0;
    }
    return;
  }
  for (int c = tid; c < C; c += 256) { P.cam_cf[r.cam_off + c] = s_cam_cf[c]; P.cam_win[r.cam_off + c] = w; }

  // ---- per line: free-camera mask, lanes, pair items; the key of the counting sort (long lines first, by falling length; then length
  // class, group bucket, falling length - lba_pack.cpp:173-217), ties in line order
  for (int l = tid; l < Lp; l += 256) {
    unsigned key = 0xffffffffu;
    if (l < L) {
      const unsigned cw = A[l];
      const int cnt = (int)(cw & 0xffffffu);
      const bool is_const = (cw & kLineConst) != 0;
      unsigned long long m = CM[l] & freeset;
      unsigned fm = 0;
      while (m) { const int c = __ffsll((long long)m) - 1; m &= m - 1; fm |= 1u << s_cam_cf[c]; }
      const int kfree = __popc(fm);
      const int len = cnt > 1 ? cnt : 1;
      const int itm = is_const ? 0 : (kfree * (kfree - 1)) / 2;
      BuildLine li;
      li.fm = fm | (is_const ? (uint32_t)kLineConst : 0u) | (cnt == 0 ? (uint32_t)kLineEmpty : 0u);
      li.len = (uint8_t)len; li.items = (uint8_t)itm; li.next = (uint16_t)kNoRow;
      LI[l] = li;
      if (!is_const && cnt > 0) atomicAdd(&s_free_lines, 1);
      if (!grouping && itm) atomicAdd(&s_nitems, itm);
      unsigned ck;
      if (cnt > 16) { ck = (unsigned)(64 - cnt); atomicAdd(&s_nlong, 1); }
      else {
        const int cls = len < 4 ? 1 : 0, nb = grouping ? kBuildBuckets : 1;
        if (!cls) atomicAdd(&s_ncls0, 1);
        int bq = 0;
        if (grouping) { const int gk = build_group_key(li.fm); bq = gk >= 1000 ? kBuildBuckets - 1 : gk; }
        ck = 64u + (unsigned)((cls * nb + bq) * 16 + (16 - len));
      }
      key = ck << 16 | (unsigned)l;
    }
    A[l] = key;
  }
  __syncthreads();
  BUILD_STAMP(2);
  block_bitonic_sort(A, Lp);
  BUILD_STAMP(3);
  // ---- to the next stage: the sorted keys, the line records, the window's scalars
  unsigned* g_keys = P.mid_keys + r.line_off;
  unsigned long long* g_li = reinterpret_cast<unsigned long long*>(P.mid_li + r.line_off);
  for (int l = tid; l < L; l += 256) { g_keys[l] = A[l]; g_li[l] = CM[l]; }
  if (tid == 0) {
    BuildMid& m = P.mid[w];
    m.freeset = freeset; m.constset = constset; m.Cf = Cf; m.free_lines = s_free_lines; m.nitems = s_nitems; m.n_long = s_nlong; m.n_cls0 = s_ncls0; m.ok = 1;
  }
#undef BUILD_STAMP
}

// The rows: best fit over the sorted lines (lba_pack.cpp:153-172) - the one sequential stage of the build, so it is written for latency and
// runs TWO WAVES per window: the two length classes (lines of at least 4 lanes; shorter ones) are packed independently of each other
// (lba_pack.cpp:221-251: the open lists are cleared between them), wave 0 takes the long lines and class 0, wave 1 class 1, four windows to a
// CU (36 KB of LDS each for 2000 lines: the whole batch of 1024 windows is resident at once).
//   * the 17 lists of open rows (by the lanes a row has left) live one per lane: tail and and-mask of lane q are list q's, read with
//     v_readlane and written under a lane compare (the list index is wave-uniform: no LDS round trip); the rows of a list are chained
//     through R[].lprev / lnext (a row sits in at most one list; a list is walked from its tail);
//   * the sorted lines come 64 at a time, one block ahead of the walk: lane j fetches line b + j's record, the walk reads it with v_readlane;
//   * what is left per line is ONE dependent LDS read - the 16-byte record of the chosen row - and one block of stores by lane 0.
// Wave 1 keeps its rows at the top of R going down (row j of class 1 at R[L - 1 - j]): the two waves never meet (a row holds at least one
// line).  LDS: R [L] 16 B row records | NX [L] u16 next line of a line's row.
__host__ __device__ inline size_t build_rows_lds_bytes(int L) { const size_t Lq = (size_t)(L > 0 ? L : 1); return 16 * Lq + 2 * Lq + 64; }
__global__ __launch_bounds__(128) void k_build_rows(BuildPtrs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rows_smem[];
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RawWin r = P.raw[w];
  const int L = r.L, grouping = P.grouping;
  const int Lq = L > 0 ? L : 1;
  const BuildMid mid = P.mid[w];
  if (!mid.ok) return;
  BuildRow* R = reinterpret_cast<BuildRow*>(rows_smem);
  uint4* R4 = reinterpret_cast<uint4*>(R);
  uint16_t* NX = reinterpret_cast<uint16_t*>(R + Lq);
  const unsigned* A = P.mid_keys + r.line_off;
  const BuildLine* LI = P.mid_li + r.line_off;
  uint16_t* tile_rows = P.mid_trows + r.line_off + 8 * w;
  uint16_t* tile_ptr = P.mid_tptr + r.line_off + 8 * w;
  __shared__ int s_rows0, s_ntr, s_ntp, s_nbig;
#define BUILD_STAMP(i) do { if (P.dbg && w == 0 && tid == 0) P.dbg[i] = (unsigned long long)wall_clock64(); } while (0)
  BUILD_STAMP(3);
  for (int l = tid; l < L; l += 128) NX[l] = (uint16_t)kNoRow;
  __syncthreads();
#define RL(v, i) __builtin_amdgcn_readlane((int)(v), (i))
#define WL(v, val, i) do { if (lane == (i)) (v) = (val); } while (0)      /* (no v_writelane builtin in this compiler: a compare + select) */
#define LDS_FENCE() asm volatile("" ::: "memory")
  // (plain LDS pointers + compiler barriers: `volatile` turns every access into a FLAT instruction with a full wait behind it)
  const int top = wave ? L - 1 : 0, sgn = wave ? -1 : 1;          // row j of this wave lives at R[top + sgn * j]
#define PH(j) (top + sgn * (j))
  int q_tail = -1;
  unsigned q_and = ~0u;
  int nrows = 0, ntr = 0, ntp = 1;
  int i = wave ? mid.n_long + mid.n_cls0 : 0;
  const int iend = wave ? L : mid.n_long + mid.n_cls0;
  if (wave == 0) {
    // long lines: a row entry of their own that spans (k + 15) / 16 rows of one tile
    if (lane == 0) tile_ptr[0] = 0;
    int used_rows = 4;
    for (; i < mid.n_long; ++i) {
      const unsigned key = A[i];
      const int l = (int)(key & 0xffffu);
      const BuildLine li = LI[l];
      const int len = li.len, nr = (len + 15) / 16;
      if (used_rows + nr > 4) { if (ntr > 0) { if (lane == 0) tile_ptr[ntp] = (uint16_t)ntr; ++ntp; } used_rows = 0; }
      const int rr = nrows++;
      if (lane == 0) {
        uint4 v; v.x = (unsigned)len | 1u << 8 | (unsigned)li.items << 16; v.y = (unsigned)l | (unsigned)l << 16; v.z = (unsigned)kNoRow | (unsigned)kNoRow << 16; v.w = li.fm & kLineMask;
        R4[rr] = v;
        tile_rows[ntr] = (uint16_t)rr;
      }
      ++ntr;
      used_rows += nr;
    }
    if (ntr > 0) { if (lane == 0) tile_ptr[ntp] = (uint16_t)ntr; ++ntp; }
    i = __builtin_amdgcn_readfirstlane(i);
  }
  const int nbig = nrows;
  LDS_FENCE();
  int cur_bucket = -1, prev_first = nrows;
  // lane j: the record of sorted line base + j - fetched from memory one block of 64 lines ahead of the walk
  auto fetch = [&](int base, unsigned& k, unsigned& f, unsigned& q) {
    k = 0xffffffffu; f = 0; q = 0;
    if (base + lane < iend) {
      k = A[base + lane];
      const BuildLine li = LI[k & 0xffffu];
      f = li.fm; q = (unsigned)li.len | (unsigned)li.items << 8;
    }
  };
  unsigned nx_key, nx_fm, nx_li;
  fetch(i, nx_key, nx_fm, nx_li);
  for (int base = i; base < iend; base += 64) {
    const unsigned my_key = nx_key, my_fm = nx_fm, my_li = nx_li;
    fetch(base + 64, nx_key, nx_fm, nx_li);
    const int nhere = min(64, iend - base);
    for (int j = 0; j < nhere; ++j) {
      LDS_FENCE();
      const int jj = __builtin_amdgcn_readfirstlane(j);
      const unsigned key = (unsigned)RL(my_key, jj);
      const int l = (int)(key & 0xffffu);
      const int bucket = ((int)(key >> 16) - 64) >> 4;
      if (bucket != cur_bucket) {
        if (grouping) {
          // only the rows the previous group left open stay in the lists (lba_pack.cpp:229-236); a list is walked from its tail
          for (int q = 0; q <= 16; ++q) {
            int cur = RL(q_tail, q);
            int after = -1;                                    // the row that follows `cur` in the list after the removals so far
            while (cur >= 0) {
              const uint4 v = R4[PH(cur)];
              const int pv = (int)(v.z & 0xffffu), pvi = pv == kNoRow ? -1 : pv;
              if (cur < prev_first) {
                // unlink: the neighbours close ranks
                if (lane == 0) { if (pvi >= 0) R[PH(pvi)].lnext = (uint16_t)(after < 0 ? kNoRow : after); if (after >= 0) R[PH(after)].lprev = (uint16_t)pv; }
                if (after < 0) WL(q_tail, pvi, q);
              } else after = cur;
              LDS_FENCE();
              cur = pvi;
            }
          }
          prev_first = nrows;
        }
        cur_bucket = bucket;
      }
      const unsigned li = (unsigned)RL(my_li, jj), fmw = (unsigned)RL(my_fm, jj);
      const int len = (int)(li & 0xffu), itm = (int)(li >> 8);
      const unsigned fm = fmw & kLineMask;
      int rr = -1, room = 0;
      unsigned rx = 0, ry = 0, rz = 0, rw = 0;
      const bool fits = lane >= len && lane <= 16 && q_tail >= 0;
      // pass 0: the fullest open row that holds the line and shares no free camera with it (at most 32 candidates per list, newest first;
      // a list whose and-mask meets the line's cameras cannot hold such a row and is not read)
      unsigned long long m0 = __ballot(fits && !(q_and & fm));
      while (m0 && rr < 0) {
        room = __ffsll((long long)m0) - 1;
        m0 &= m0 - 1;
        int cur = RL(q_tail, room);
        for (int steps = 0; cur >= 0 && steps < 32; ++steps) {
          const uint4 v = R4[PH(cur)];
          if (!(v.w & fm)) { rr = cur; rx = v.x; ry = v.y; rz = v.z; rw = v.w; break; }
          const int pv = (int)(v.z & 0xffffu);
          cur = pv == kNoRow ? -1 : pv;
        }
      }
      if (rr < 0) {
        // pass 1: the fullest open row that holds it
        const unsigned long long m1 = __ballot(fits);
        if (m1) {
          room = __ffsll((long long)m1) - 1;
          rr = RL(q_tail, room);
          const uint4 v = R4[PH(rr)];
          rx = v.x; ry = v.y; rz = v.z; rw = v.w;
        }
      }
      // what lane 0 stores for this line: [neighbour p].lnext, [neighbour n].lprev, NX[tail], [list tail t].lnext, the row's record
      int st_p = -1, st_n = -1, st_tail = -1, st_t = -1, st_pv = 0, st_nv = 0;
      if (rr >= 0) {
        rr = __builtin_amdgcn_readfirstlane(rr);
        rx = (unsigned)__builtin_amdgcn_readfirstlane((int)rx); ry = (unsigned)__builtin_amdgcn_readfirstlane((int)ry);
        rz = (unsigned)__builtin_amdgcn_readfirstlane((int)rz); rw = (unsigned)__builtin_amdgcn_readfirstlane((int)rw);
        // unlink from its list
        const int p = (int)(rz & 0xffffu), n = (int)(rz >> 16);
        if (p != kNoRow) { st_p = p; st_pv = n; }
        if (n != kNoRow) { st_n = n; st_nv = p; } else WL(q_tail, p == kNoRow ? -1 : p, room);
        // append (the row has a line already)
        st_tail = (int)(ry >> 16);
        ry = (ry & 0xffffu) | (unsigned)l << 16;
        rx = ((rx & 0xffu) + (unsigned)len) | (((rx >> 8) & 0xffu) + 1u) << 8 | ((rx >> 16) + (unsigned)itm) << 16;
        rw |= fm;
      } else {
        rr = nrows++;
        rx = (unsigned)len | 1u << 8 | (unsigned)itm << 16; ry = (unsigned)l | (unsigned)l << 16; rw = fm;
      }
      const int used = (int)(rx & 0xffu);
      if (used < 16) {
        // push: the row joins the list of its remaining room, at the tail
        const int room2 = 16 - used;
        const int t = RL(q_tail, room2);
        rz = (unsigned)(t < 0 ? kNoRow : t) | (unsigned)kNoRow << 16;
        st_t = t;
        WL(q_and, t < 0 ? rw : ((unsigned)RL(q_and, room2) & rw), room2);
        WL(q_tail, rr, room2);
      } else rz = (unsigned)kNoRow | (unsigned)kNoRow << 16;
      if (lane == 0) {
        if (st_p >= 0) R[PH(st_p)].lnext = (uint16_t)st_pv;
        if (st_n >= 0) R[PH(st_n)].lprev = (uint16_t)st_nv;
        if (st_tail >= 0) NX[st_tail] = (uint16_t)l;
        if (st_t >= 0) R[PH(st_t)].lnext = (uint16_t)rr;
        uint4 v; v.x = rx; v.y = ry; v.z = rz; v.w = rw;
        R4[PH(rr)] = v;
      }
    }
  }
  LDS_FENCE();
#undef RL
#undef WL
  if (lane == 0 && wave == 0) { s_rows0 = nrows; s_ntr = ntr; s_ntp = ntp; s_nbig = nbig; }
  __syncthreads();
  BUILD_STAMP(4);
  // ---- to the next stage: the rows (class 1's behind class 0's, each in the order they were made), the chains, the window's counts
  const int rows0 = s_rows0;
  uint4* g_rows = P.mid_rows + r.line_off;
  for (int q = lane; q < nrows; q += 64) g_rows[(wave ? rows0 : 0) + q] = R4[PH(q)];
  uint16_t* g_next = P.mid_next + r.line_off;
  for (int l = tid; l < L; l += 128) g_next[l] = NX[l];
  if (lane == 0 && wave == 1) {
    BuildMid& m = P.mid[w];
    m.nrows = rows0 + nrows; m.ntr = s_ntr; m.ntp = s_ntp;
    m.cls_row[0] = s_nbig; m.cls_row[1] = rows0; m.cls_row[2] = rows0; m.cls_row[3] = rows0 + nrows;
  }
#undef PH
#undef LDS_FENCE
#undef BUILD_STAMP
}

// Row order, tile ranges, line order, line pointers, the observations of every line sorted by camera, the window's line-level arrays
// (lba_pack.cpp:238-333): parallel again, 256 threads per window.  LDS as k_build_lines plus the rows and the tile tables read back.
__global__ __launch_bounds__(256) void k_build_order(BuildPtrs P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char build_smem[];
  const int w = blockIdx.x, tid = threadIdx.x;
  const RawWin r = P.raw[w];
  const int L = r.L, M = r.M, grouping = P.grouping;
  const int Lp = build_pow2(L), Lq = L > 0 ? L : 1;
  const BuildMid mid = P.mid[w];
  if (!mid.ok) return;
  unsigned* A = reinterpret_cast<unsigned*>(build_smem);
  unsigned long long* CM = reinterpret_cast<unsigned long long*>(A + Lp);
  BuildLine* LI = reinterpret_cast<BuildLine*>(CM);
  BuildRow* R = reinterpret_cast<BuildRow*>(CM + Lq);
  uint16_t* T = reinterpret_cast<uint16_t*>(R + Lq);
  uint16_t* tile_rows = T;
  uint16_t* tile_ptr = T + Lq + 2;
  unsigned* part = reinterpret_cast<unsigned*>(T + 2 * Lq + 8);
  __shared__ int s_ntr, s_ntiles, s_cls_row[4], s_nkept;
  __shared__ signed char s_cam_cf[64];
  BuildWin* bw = P.bw + w;
  const uint32_t* ridx = P.raw_idx + r.obs_off;
  const unsigned long long freeset = mid.freeset, constset = mid.constset;
  const int Cf = mid.Cf;
#define BUILD_STAMP(i) do { if (P.dbg && w == 0 && tid == 0) P.dbg[i] = (unsigned long long)wall_clock64(); } while (0)
  // the line records (with the chains the row stage made), the rows, the tile tables of the long lines
  {
    const unsigned long long* g_li = reinterpret_cast<const unsigned long long*>(P.mid_li + r.line_off);
    const uint16_t* g_next = P.mid_next + r.line_off;
    for (int l = tid; l < L; l += 256) { CM[l] = g_li[l]; }
    __syncthreads();
    for (int l = tid; l < L; l += 256) LI[l].next = g_next[l];
    const uint4* g_rows = P.mid_rows + r.line_off;
    for (int q = tid; q < mid.nrows; q += 256) reinterpret_cast<uint4*>(R)[q] = g_rows[q];
    const uint16_t* g_tr = P.mid_trows + r.line_off + 8 * w;
    const uint16_t* g_tp = P.mid_tptr + r.line_off + 8 * w;
    for (int q = tid; q < mid.ntr; q += 256) tile_rows[q] = g_tr[q];
    for (int q = tid; q < mid.ntp; q += 256) tile_ptr[q] = g_tp[q];
    if (tid < 64) s_cam_cf[tid] = (tid < r.C && ((freeset >> tid) & 1ull)) ? (signed char)__popcll(freeset & ((1ull << tid) - 1ull)) : (signed char)-1;
    if (tid == 0) { s_ntr = mid.ntr; s_ntiles = mid.ntp - 1; for (int q = 0; q < 4; ++q) s_cls_row[q] = mid.cls_row[q]; s_nkept = 0; }
  }
  __syncthreads();
  BUILD_STAMP(4);
  // ---- the rows of each class in tile order (lba_pack.cpp:238-283), then the tiles' row ranges
  for (int cls = 0; cls < 2; ++cls) {
    const int r0 = s_cls_row[2 * cls], r1 = s_cls_row[2 * cls + 1], nr = r1 - r0;
    const int np = build_pow2(nr);             // (<= Lp: a row holds at least one line)
    const bool by_key = grouping || cls == 0;  // grouped: by the group of a row's first line; default packing, class 0: by pair items, heavy first; class 1: as built
    if (nr > 0 && by_key) {
      for (int q = tid; q < np; q += 256) {
        unsigned key = 0xffffffffu;
        if (q < nr) {
          const int rr = r0 + q;
          unsigned rk;
          if (grouping) {
            const int h = R[rr].head;
            const int k0 = build_group_key(LI[h].fm);
            int mixed = 0;
            for (int l = h; l != kNoRow; l = LI[l].next) if (build_group_key(LI[l].fm) != k0) { mixed = 1; break; }
            rk = (unsigned)(k0 * 2 + mixed);
          } else rk = 65535u - R[rr].items;
          key = rk << 16 | (unsigned)q;
        }
        A[q] = key;
      }
      __syncthreads();
      block_bitonic_sort(A, np);
    }
    if (!grouping && cls == 0) {
      if (tid == 0) {
        // boustrophedon deal: heavy rows meet light rows (the rows in order of falling pair-item count)
        int ntr = s_ntr, ntp = s_ntiles + 1;
        const int Tn = (nr + 3) / 4;
        for (int t = 0; t < Tn; ++t) {
          for (int pass = 0; pass < 4; ++pass) {
            const int pos = (pass & 1) ? Tn - 1 - t : t, rix = pass * Tn + pos;
            if (rix < nr) tile_rows[ntr++] = (uint16_t)(r0 + (int)(A[rix] & 0xffffu));
          }
          tile_ptr[ntp++] = (uint16_t)ntr;
        }
        s_ntr = ntr; s_ntiles = ntp - 1;
      }
    } else {
      // four rows to a tile, in order
      const int first = s_ntr, tp0 = s_ntiles + 1, nt = (nr + 3) / 4;
      for (int q = tid; q < nr; q += 256) tile_rows[first + q] = (uint16_t)(by_key ? r0 + (int)(A[q] & 0xffffu) : r0 + q);
      for (int t = tid; t < nt; t += 256) tile_ptr[tp0 + t] = (uint16_t)(first + min(4 * (t + 1), nr));
      __syncthreads();
      if (tid == 0) { s_ntr = first + nr; s_ntiles = tp0 + nt - 1; }
    }
    __syncthreads();
  }
  const int ntr = s_ntr, ntiles = s_ntiles;

  BUILD_STAMP(5);
  // ---- line order: the rows in tile order, the lines of a row as they were appended
  int* g_line_orig = P.line_orig + r.line_off;
  uint8_t* g_lflags = P.lflags + r.line_off;
  for (int p = tid; p < ntr; p += 256) A[p] = R[tile_rows[p]].nl;
  __syncthreads();
  block_scan_excl(A, ntr, part);               // A[p]: sorted position of the first line of row entry p
  for (int p = tid; p < ntr; p += 256) {
    int s = (int)A[p];
    bool first = true;
    for (int l = R[tile_rows[p]].head; l != kNoRow; l = LI[l].next, ++s) {
      g_line_orig[s] = l;
      g_lflags[s] = first ? 1 : 0;
      first = false;
    }
  }
  __syncthreads();
  for (int t = tid; t < ntiles; t += 256) g_lflags[A[tile_ptr[t]]] = 3;      // first line of a tile (and of its first row entry)
  __syncthreads();
  // ---- line pointers (the rows are dead: their memory holds the pointers and the fill counters); position of every line
  unsigned* lptr = reinterpret_cast<unsigned*>(R);
  unsigned* fill = lptr + Lq + 1;
  uint16_t* line_pos = reinterpret_cast<uint16_t*>(A);
  for (int s = tid; s < L; s += 256) {
    const int l = g_line_orig[s];
    lptr[s] = (LI[l].fm & kLineEmpty) ? 0u : (unsigned)LI[l].len;
    fill[s] = 0;
  }
  __syncthreads();
  for (int s = tid; s < L; s += 256) { const int l = g_line_orig[s]; line_pos[l] = (uint16_t)s; if (P.line_pos) P.line_pos[r.line_off + l] = s; }
  block_scan_excl(lptr, L, part);
  if (tid == 0) lptr[L] = (unsigned)M;
  __syncthreads();

  BUILD_STAMP(6);
  // ---- the observations of every line, free cameras first (ascending free index), then the others by id; ties in the caller's order
  // (lba_pack.cpp:308-333): keys camera key << 24 | index dropped into the line's range, then sorted per line
  unsigned* okey = reinterpret_cast<unsigned*>(P.ob_orig + r.obs_off);
  int kept = 0;
  // (four index words in flight per thread: the loop is a chain of one global load, two LDS reads, an LDS atomic and a store per observation)
  for (int i0 = tid; i0 < M; i0 += 1024) {
    uint32_t vv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vv[u] = i0 + 256 * u < M ? ridx[i0 + 256 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + 256 * u;
      if (i >= M) break;
      const uint32_t v = vv[u];
      const int l = (int)(v & 0xffffu), c = (int)((v >> 16) & 0xffu);
      const int s = line_pos[l];
      const unsigned slot = atomicAdd(&fill[s], 1u);
      const int cf = s_cam_cf[c];
      const unsigned ck = cf >= 0 ? (unsigned)cf : (unsigned)(Cf + c);
      okey[lptr[s] + slot] = ck << 24 | (unsigned)i;
      if (!(((constset >> c) & 1ull) && (LI[l].fm & kLineConst))) ++kept;
    }
  }
  if (kept) atomicAdd(&s_nkept, kept);
  __syncthreads();
  BUILD_STAMP(7);
  int* g_ob_cam = P.ob_cam + r.obs_off;
  for (int s = tid; s < L; s += 256) {
    const int o0 = (int)lptr[s], k = (int)lptr[s + 1] - o0;
    unsigned* o = okey + o0;
    for (int a = 1; a < k; ++a) {
      const unsigned v = o[a];
      int b = a;
      while (b > 0 && o[b - 1] > v) { o[b] = o[b - 1]; --b; }
      o[b] = v;
    }
    for (int a = 0; a < k; ++a) {
      const unsigned i = o[a] & 0xffffffu;
      o[a] = i;
      g_ob_cam[o0 + a] = (int)((ridx[i] >> 16) & 0xffu);
    }
  }
  BUILD_STAMP(8);
  // ---- the window's line-level arrays
  for (int s = tid; s < L; s += 256) {
    const int l = g_line_orig[s];
    const long long g = (long long)r.line_off + s;
    P.line_ptr[g] = r.obs_off + (int)lptr[s];
    P.line_flags[g] = (LI[l].fm & kLineConst) ? 1 : 0;
    P.line_win[g] = w;
    P.fmask[g] = LI[l].fm & kLineMask;
    const double2* src = reinterpret_cast<const double2*>(P.line_raw + 4LL * (r.line_off + l));
    double2* dst = reinterpret_cast<double2*>(P.line_x0 + 4LL * g);
    dst[0] = src[0]; dst[1] = src[1];
  }
  if (tid == 0) {
    bw->Cf = Cf; bw->ntiles = ntiles; bw->nitems = mid.nitems; bw->nfree_params = 6 * Cf + 4 * mid.free_lines; bw->nkept = s_nkept;
  }
  BUILD_STAMP(9);
#undef BUILD_STAMP
}

// ------------------------------------------------------------------------------------------------------------------------------
// plan_layout (lba_api.hip) with frozen = true, on the device: one workgroup, thread <-> a run of consecutive windows.
__device__ inline int layout_window_chunks(const LayoutArgs& a, int ntiles, int* graded_chunks, int* graded_rounds, int* per_chunk) {
  int gc = 0, gr = 0, per = 1;
  if (a.chunks_per_window < 0) { gr = (-a.chunks_per_window) / 1000; gc = (-a.chunks_per_window) % 1000; per = 1; }
  else if (a.chunks_per_window > 0) per = max(1, (ntiles + a.chunks_per_window - 1) / a.chunks_per_window);
  else {
    long long rounds = a.auto_rounds, cpw = a.auto_cpw;
    if (a.reproducible) { cpw = max(1, (ntiles + 33) / 34); rounds = 3; }
    per = (int)max((long long)a.elim_waves, (ntiles + cpw - 1) / cpw);
    if (rounds >= 2 && cpw >= rounds && cpw < 1000 && a.elim_waves == 1 && ntiles >= 8 * cpw && !a.equal_chunks) { gc = (int)cpw; gr = (int)rounds; }
  }
  *graded_chunks = gc; *graded_rounds = gr; *per_chunk = per;
  if (ntiles <= 0) return 0;
  if (gc > 0) {
    if (ntiles < 2 * gc) { const int tpc = (ntiles + gc - 1) / gc; return (ntiles + tpc - 1) / tpc; }       // chunk_boundaries_graded falls back to equal chunks
    return gc;
  }
  return (ntiles + per - 1) / per;
}
__device__ inline int layout_weight(int c, int gc, int gr) {
  const int q = (int)(((long long)c * gr) / gc);
  return q == 0 ? 84 : q == 1 ? (gr == 2 ? 36 : 24) : max(1, 12 / (gr - 2));
}

enum { kLayoutMaxRank = 8 };
__global__ __launch_bounds__(256) void k_build_layout(BuildPtrs P, LayoutArgs a) {
  const int tid = threadIdx.x, B = P.nwin;
  const int per = (B + 255) / 256, lo = min(B, tid * per), hi = min(B, lo + per);
  enum { NQ = 5 + kLayoutMaxRank };              // tiles | chunks | sys | items | (unused) | chunks of rank 0 .. 7
  __shared__ long long s_part[NQ][257];
  __shared__ long long s_slab_part[257];
  __shared__ int s_bad, s_maxchunks;
  if (tid == 0) { s_bad = 0; s_maxchunks = 0; }
  __syncthreads();
  long long acc[NQ];
  for (int q = 0; q < NQ; ++q) acc[q] = 0;
  long long slab_acc = 0;
  int bad = 0, maxchunks = 0;
  for (int w = lo; w < hi; ++w) {
    BuildWin* bw = P.bw + w;
    const bool okw = bw->status == 0;
    const int ntiles = okw ? bw->ntiles : 0, n = okw ? 6 * bw->Cf : 0;
    int gc, gr, pc;
    const int nch = layout_window_chunks(a, ntiles, &gc, &gr, &pc);
    const bool graded = gc > 0 && nch == gc;
    if (a.chunks_per_window < 0 && (gr < 2 || gc < gr)) bad = 1;
    if (graded && gr > kLayoutMaxRank) bad = 1;
    bw->nchunks = nch; bw->graded = graded ? gr : 0;
    const int nsys = a.elim_mode == 1 ? sys_doubles_mfma(n) : sys_doubles(n);
    acc[0] += ntiles; acc[1] += nch; acc[2] += n; acc[3] += okw ? bw->nitems : 0;
    slab_acc += (long long)nch * (nsys + kSlabScalars);
    for (int c = 0; c < nch; ++c) acc[5 + (graded ? (int)(((long long)c * gr) / gc) : 0)] += 1;
    maxchunks = max(maxchunks, nch);
    if (okw) {
      if (n > a.cap_maxn || bw->Cf > a.max_free || a.sys_map_off[bw->Cf] < 0) bad = 1;
      if (a.slab_sum) {
        if (a.slab_sum_image) { const int N = solve_pad(n); const long long ext = (long long)N * solve_stride(n) + 6LL * N; if (((ext + kSlabScalars + 1) / 2) * 2 > a.slab_sum_stride) bad = 1; }
        else if ((long long)nsys + kSlabScalars > a.slab_sum_stride) bad = 1;
      } else if (nch > 8) bad = 1;
    }
  }
  for (int q = 0; q < NQ; ++q) s_part[q][tid] = acc[q];
  s_slab_part[tid] = slab_acc;
  if (bad) atomicOr(&s_bad, 1);
  atomicMax(&s_maxchunks, maxchunks);
  __syncthreads();
  if (tid < NQ + 1) {
    long long* v = tid < NQ ? s_part[tid] : s_slab_part;
    long long run = 0;
    for (int t = 0; t < 256; ++t) { const long long x = v[t]; v[t] = run; run += x; }
    v[256] = run;
  }
  __syncthreads();
  const long long tot_tiles = s_part[0][256], tot_chunks = s_part[1][256], tot_sys = s_part[2][256], tot_items = s_part[3][256], tot_slab = s_slab_part[256];
  const bool fit = !s_bad && tot_tiles <= a.cap_tiles && tot_chunks <= a.cap_chunks && tot_sys <= a.cap_sys && tot_items <= a.cap_items &&
                   tot_slab <= a.cap_slab && tot_slab <= 0x7fffffffLL;
  long long rank_base[kLayoutMaxRank];
  { long long run = 0; for (int q = 0; q < kLayoutMaxRank; ++q) { rank_base[q] = run; run += s_part[5 + q][256]; } }
  long long tile_off = s_part[0][tid], chunk_off = s_part[1][tid], sys_off = s_part[2][tid], item_off = s_part[3][tid], slab_off = s_slab_part[tid];
  long long rank_pos[kLayoutMaxRank];
  for (int q = 0; q < kLayoutMaxRank; ++q) rank_pos[q] = rank_base[q] + s_part[5 + q][tid];
  for (int w = lo; w < hi; ++w) {
    const RawWin r = P.raw[w];
    BuildWin* bw = P.bw + w;
    if (!fit && bw->status == 0) bw->status = kBuildNoFit;
    const bool okw = bw->status == 0;
    const int ntiles = okw ? bw->ntiles : 0, n = okw ? 6 * bw->Cf : 0, nch = okw ? bw->nchunks : 0;
    WinDesc wd;
    wd.C = okw ? r.C : 0; wd.Cf = okw ? bw->Cf : 0; wd.L = okw ? r.L : 0; wd.M = okw ? r.M : 0;
    wd.cam_off = r.cam_off; wd.line_off = r.line_off; wd.obs_off = r.obs_off;
    wd.tile_off = (int)tile_off; wd.ntiles = ntiles;
    wd.chunk_off = (int)chunk_off; wd.nchunks = nch;
    wd.n = n; wd.sys_off = (int)sys_off; wd.nfree_params = okw ? bw->nfree_params : 0; wd.nkept = okw ? bw->nkept : 0;
    wd.slab_off = (int)slab_off;
    wd.map_off = okw && a.sys_map_off[wd.Cf] >= 0 ? a.sys_map_off[wd.Cf] : 0;
    P.wins[w] = wd;
    P.param_off[w] = r.param_off;
    P.item_base[w] = (int)item_off;
    if (!fit) {
      // nothing of the refill is launched; a window's records belong to nobody
      for (int l = 0; l < r.L; ++l) P.line_win[r.line_off + l] = -1;
      for (int c = 0; c < r.C; ++c) { P.cam_win[r.cam_off + c] = -1; P.cam_cf[r.cam_off + c] = -1; }
    }
    if (fit && nch > 0) {
      const int nsys = a.elim_mode == 1 ? sys_doubles_mfma(n) : sys_doubles(n);
      const long long slab_stride = (long long)nsys + kSlabScalars;
      int gc, gr, pc;
      (void)layout_window_chunks(a, ntiles, &gc, &gr, &pc);
      const bool graded = bw->graded != 0;                     // the dispatch classes (plan_layout: win_graded)
      const bool graded_cut = gc > 0 && ntiles >= 2 * gc;      // the boundaries: chunk_boundaries_graded, or its fall-back to equal chunks (lba_pack.cpp)
      long long total_w = 0, acc_w = 0;
      if (graded_cut) for (int c = 0; c < nch; ++c) total_w += layout_weight(c, gc, gr);
      int prev = 0;
      for (int c = 0; c < nch; ++c) {
        int e;
        if (graded_cut) {
          acc_w += layout_weight(c, gc, gr);
          e = (int)((ntiles * acc_w + total_w / 2) / total_w);
          const int lo_e = prev + 1, hi_e = ntiles - (nch - 1 - c);
          if (e < lo_e) e = lo_e;
          if (e > hi_e) e = hi_e;
          if (c == nch - 1) e = ntiles;
        } else e = (int)(((long long)ntiles * (c + 1)) / nch);
        Chunk ck;
        ck.win = w; ck.tile_begin = (int)tile_off + prev; ck.tile_end = (int)tile_off + e;
        ck.slab_off = (int)(slab_off + c * slab_stride); ck.id = (int)chunk_off + c;
        const int rank = graded ? (int)(((long long)c * gr) / gc) : 0;
        P.chunks[rank_pos[rank]++] = ck;
        prev = e;
      }
      slab_off += nch * slab_stride;
    }
    tile_off += ntiles; chunk_off += nch; sys_off += n; item_off += okw ? bw->nitems : 0;
  }
  // the entries of the chunk array that belong to no chunk (the launches cover the whole array), the end of the line pointers
  const long long used = fit ? tot_chunks : 0;
  for (long long c = used + tid; c < a.cap_chunks; c += 256) { Chunk z; z.win = -1; z.tile_begin = 0; z.tile_end = 0; z.slab_off = 0; z.id = 0; P.chunks[c] = z; }
  if (tid == 0) {
    P.line_ptr[a.nline] = a.nobs;
    P.totals[0] = fit ? (int)tot_tiles : 0; P.totals[1] = fit ? (int)tot_items : 0; P.totals[2] = (int)used; P.totals[3] = fit ? 1 : 0;
    P.totals[4] = a.nline; P.totals[5] = a.nobs; P.totals[6] = s_maxchunks;
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Tiles of one window, thread <-> tile: what lba_pack.cpp:410-505 does per tile - lane map, skew flags, line descriptors, pair items.
// stage_m >= the window's observation count: the per-line and per-observation inputs of the tile loop are first brought into LDS in bulk
// (coalesced, every thread with several loads in flight) - the loop itself is a chain of dependent reads (line pointer -> camera of the
// observation -> free index of the camera -> seen counter), each a global round trip when read in place: 0.36 ms per batch, all of it latency.
__global__ __launch_bounds__(256) void k_build_tiles(BuildPtrs P, const int* cam_cf, int stage_m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tiles_smem[];
  const int w = blockIdx.x, tid = threadIdx.x;
  const WinDesc wd = P.wins[w];
  const int L = wd.L, nt = wd.ntiles, grouping = P.grouping;
  if (nt <= 0 || L <= 0) return;
#define TILE_STAMP(i) do { if (P.dbg && w == 0 && tid == 0) P.dbg[i] = (unsigned long long)wall_clock64(); } while (0)
  TILE_STAMP(10);
  unsigned* F = reinterpret_cast<unsigned*>(tiles_smem);                 // [L] tile-start flags -> tile of each line; then [nt + 1] items per tile
  unsigned* part = F + (L > nt + 1 ? L : nt + 1);                        // [257]
  unsigned* tbeg = part + 260;                                           // [nt + 1] first sorted line of each tile
  unsigned char* seen_all = reinterpret_cast<unsigned char*>(tbeg + nt + 2);     // [256][4 * kMaxFreeCams]
  const bool staged = stage_m >= wd.M;
  unsigned* lp = reinterpret_cast<unsigned*>(seen_all);                          // staged (no seen block: the parities live in registers): [L + 1] line pointers relative to the window
  unsigned* fm = lp + L + 1;                                                     // [L] free-camera masks
  unsigned* dl = grouping ? F : fm + L;                                          // [L] line descriptors (sorted per tile here, written out in bulk); grouped packing: no pair items, F is free
  unsigned char* fl = reinterpret_cast<unsigned char*>(fm + L + (grouping ? 0 : L));     // [L] row-entry / tile-start flags (bits 0-1) | constant (bit 2)
  signed char* cfb = reinterpret_cast<signed char*>(fl + ((L + 15) & ~15));      // [M] free index of every observation's camera, or -1
  __shared__ signed char s_ccf[64];
  const uint8_t* lfl = P.lflags + wd.line_off;
  for (int s = tid; s < L; s += 256) F[s] = (lfl[s] & 2) ? 1u : 0u;
  __syncthreads();
  block_scan_excl(F, L, part);
  for (int s = tid; s < L; s += 256) if (lfl[s] & 2) tbeg[F[s]] = (unsigned)s;
  if (tid == 0) tbeg[nt] = (unsigned)L;
  __syncthreads();
  const int* line_ptr = P.line_ptr + wd.line_off;
  const int* line_flags = P.line_flags + wd.line_off;
  const uint32_t* fmask = P.fmask + wd.line_off;
  const int* ccf = cam_cf + wd.cam_off;
  // pair items per tile (default packing only), then their offsets
  for (int t = tid; t <= nt; t += 256) {
    unsigned cnt = 0;
    if (t < nt && !grouping)
      for (int s = (int)tbeg[t]; s < (int)tbeg[t + 1]; ++s)
        if (!(line_flags[s] & 1)) { const int kf = __popc(fmask[s]); cnt += (unsigned)((kf * (kf - 1)) / 2); }
    F[t] = cnt;
  }
  __syncthreads();
  block_scan_excl(F, nt + 1, part);
  const int item_base = P.item_base[w];
  TILE_STAMP(11);
  unsigned char* seen = seen_all + tid * (4 * kMaxFreeCams);
  if (staged) {
    if (tid < 64) s_ccf[tid] = tid < wd.C ? (signed char)ccf[tid] : (signed char)-1;
    for (int s = tid; s < L; s += 256) {
      lp[s] = (unsigned)(line_ptr[s] - wd.obs_off);
      fm[s] = fmask[s];
      fl[s] = (unsigned char)((lfl[s] & 3) | ((line_flags[s] & 1) << 2));
    }
    if (tid == 0) lp[L] = (unsigned)wd.M;
    __syncthreads();
    const int* oc = P.ob_cam + wd.obs_off;
    for (int i0 = tid; i0 < wd.M; i0 += 1024) {
      int c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = i0 + 256 * u < wd.M ? oc[i0 + 256 * u] : 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i0 + 256 * u < wd.M) cfb[i0 + 256 * u] = s_ccf[c[u] & 63];
    }
    __syncthreads();
    TILE_STAMP(12);
    for (int t = tid; t < nt; t += 256) {
      const long long gt = (long long)wd.tile_off + t;
      uint16_t* map = P.lane_map + 64 * gt;
      unsigned sb0 = 0, sb1 = 0, sb2 = 0, sb3 = 0;                         // parity of the (row, free camera) counters: one word per row
      const unsigned it0 = grouping ? 0u : F[t], it1 = grouping ? 0u : F[t + 1];
      uint8_t* item_w = P.items + 2 * ((long long)item_base + it0);
      const int s0 = (int)tbeg[t], s1 = (int)tbeg[t + 1];
      // ONE pass over the tile's 64 lanes, the same trip count for every thread of a wave (a loop over lines with an inner loop over a
      // line's lanes costs a wave the longest line times the most lines of its 64 tiles: 68 us for this loop against 13 in this form).
      // Every lane position begins at most one line (a line takes at least one lane).
      int lane = 0, nl = 0, min_lanes = 64, max_run = 1, multi = 0;
      int s = s0, next_start = 0, end = 0, o0 = 0, k = 0;
      for (int q8 = 0; q8 < 64; q8 += 8) {
      unsigned pk[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
      for (int u8 = 0; u8 < 8; ++u8) {
        const int q = q8 + u8;
        if (s < s1 && q == next_start) {
          // line s begins at lane q
          const unsigned f = fl[s];
          lane = q;
          o0 = (int)lp[s]; k = (int)lp[s + 1] - o0;
          const int run = k > 1 ? k : 1;
          end = lane + run;
          min_lanes = min(min_lanes, run);
          max_run = max(max_run, min(run, 16));
          if (run > 16) multi = 1;
          const bool is_const = (f & 4) != 0;
          const unsigned fms = fm[s];
          if (!is_const && !grouping) {
            const int kf = __popc(fms);                                  // free-camera observations come first, one per camera
            for (int i2 = 0; i2 < kf; ++i2)
              for (int j2 = i2 + 1; j2 < kf; ++j2) { *item_w++ = (uint8_t)(lane + i2); *item_w++ = (uint8_t)(lane + j2); }
          }
          const uint32_t m = is_const ? 0u : (fms & 0x3ffu);
          uint32_t desc;
          if (!grouping) {
            bool blk[4] = { false, false, false, false };
            for (int cf = 0; cf < 10; ++cf)
              if ((m >> cf) & 1u) { blk[(6 * cf) / 16] = true; blk[(6 * cf + 5) / 16] = true; }
            uint32_t touched = 0;
            for (int I = 0, tt = 0; I < 4; ++I)
              for (int J = 0; J <= I; ++J, ++tt) if (blk[I] && blk[J]) touched |= 1u << tt;
            desc = m | ((uint32_t)lane << 10) | (touched << 16);
          } else {
            uint32_t a = 0, nblk = 0, wdt = 0, holes = 0;
            if (m) {
              a = (uint32_t)(__ffs((int)m) - 1);
              wdt = (uint32_t)(31 - __clz((int)m)) - a + 1u;
              nblk = (6u * wdt + 15u) / 16u;
              holes = (uint32_t)__popc(m) != wdt ? 1u : 0u;
            }
            desc = gp_desc(m, (uint32_t)lane, a, nblk, holes, wdt);
          }
          dl[s] = desc;
          ++s; ++nl;
          next_start = s < s1 ? ((fl[s] & 1) ? (end + 15) & ~15 : end) : 64;      // every row entry starts a row
        }
        unsigned v = 0x00FFu;                                            // (a lane no line uses)
        if (q < end) {
          // lane map with the skew flag (bit 15): every second lane of a 16-lane row that holds an observation of the same free camera
          const int j2 = q - lane;
          v = (unsigned)((nl - 1) | (j2 << 8));
          if (j2 < k) {
            const int cf = cfb[o0 + j2];
            if (cf >= 0) {
              const int row = q >> 4;
              const unsigned bit = 1u << cf, cur = row == 0 ? sb0 : row == 1 ? sb1 : row == 2 ? sb2 : sb3;
              if (cur & bit) v |= 0x8000u;
              if (row == 0) sb0 ^= bit; else if (row == 1) sb1 ^= bit; else if (row == 2) sb2 ^= bit; else sb3 ^= bit;
            }
          }
        }
        pk[u8 >> 1] |= (v & 0xffffu) << (16 * (u8 & 1));
      }
      // (eight lanes per store: 2-byte stores scattered over a wave's 64 tiles were the whole cost of this loop)
      reinterpret_cast<uint4*>(map)[q8 >> 3] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      }
      const int rounds_log2 = min_lanes >= 4 ? 0 : min_lanes >= 2 ? 1 : 2;
      Tile tl;
      tl.line_begin = wd.line_off + s0; tl.nlines = (int16_t)nl;
      tl.flags = (int16_t)(multi | (rounds_log2 << 1) | (max_run << 3));
      tl.item_off = item_base + (int)it0; tl.nitems = (int)(it1 - it0);
      P.tiles[gt] = tl;
    }
    __syncthreads();
    if (grouping) {
      // the grouped sweep walks a tile's descriptors group after group, inside a group by the number of blocks, lines without elimination
      // work last (a STABLE sort, lba_pack.cpp:484-496): wave <-> tile, lane <-> line; rank = keys below + equal keys before.  (One thread
      // per tile with an insertion sort cost a wave its slowest tile: 64 short lines, ~1000 dependent LDS steps - most of this kernel.)
      const int wv = tid >> 6, ln = tid & 63;
      for (int t = wv; t < nt; t += 4) {
        const int s0 = (int)tbeg[t], n = (int)tbeg[t + 1] - s0;
        const unsigned v = ln < n ? dl[s0 + ln] : 0u;
        const int key = ln < n ? (gp_mask(v) ? (int)(gp_group(v) * 8u + gp_blocks(v)) : 1 << 20) : 0x7fffffff;
        const int nxt = __shfl_down(key, 1);
        if (__ballot(ln + 1 < n && key > nxt) == 0ull) continue;         // in order already
        int rank = 0;
        for (int u = 0; u < n; ++u) {
          const int ku = __builtin_amdgcn_readlane(key, u);
          rank += (ku < key || (ku == key && u < ln)) ? 1 : 0;
        }
        if (ln < n) dl[s0 + rank] = v;
      }
      __syncthreads();
    }
    TILE_STAMP(13);
    for (int s = tid; s < L; s += 256) P.line_desc[wd.line_off + s] = dl[s];
    TILE_STAMP(14);
    return;
  }
  for (int t = tid; t < nt; t += 256) {
    const long long gt = (long long)wd.tile_off + t;
    uint16_t* map = P.lane_map + 64 * gt;
    for (int q = 0; q < 64; ++q) map[q] = (uint16_t)0x00FF;
    for (int q = 0; q < 4 * kMaxFreeCams; ++q) seen[q] = 0;
    uint8_t* item_w = P.items + 2 * ((long long)item_base + F[t]);
    const int s0 = (int)tbeg[t], s1 = (int)tbeg[t + 1];
    int lane = 0, nl = 0, min_lanes = 64, max_run = 1, multi = 0;
    for (int s = s0; s < s1; ++s, ++nl) {
      if (lfl[s] & 1) lane = (lane + 15) & ~15;                          // every row entry starts a row
      const int o0 = line_ptr[s] - wd.obs_off, k = (s + 1 < L ? line_ptr[s + 1] : wd.obs_off + wd.M) - line_ptr[s];
      const int run = k > 1 ? k : 1;
      for (int j = 0; j < run; ++j) map[lane + j] = (uint16_t)(nl | (j << 8));
      // skew flag (bit 15): every second lane of a 16-lane row that holds an observation of the same free camera
      for (int j = 0; j < k && j < 64; ++j) {
        const int cf = ccf[P.ob_cam[wd.obs_off + o0 + j]];
        if (cf < 0) continue;
        const int row = (lane + j) >> 4;
        if (seen[row * kMaxFreeCams + cf]++ & 1) map[lane + j] |= (uint16_t)0x8000;
      }
      min_lanes = min(min_lanes, run);
      max_run = max(max_run, min(run, 16));
      if (run > 16) multi = 1;
      const bool is_const = (line_flags[s] & 1) != 0;
      if (!is_const && !grouping) {
        const int kf = __popc(fmask[s]);                                 // free-camera observations come first, one per camera
        for (int i = 0; i < kf; ++i)
          for (int j = i + 1; j < kf; ++j) { *item_w++ = (uint8_t)(lane + i); *item_w++ = (uint8_t)(lane + j); }
      }
      const uint32_t m = is_const ? 0u : (fmask[s] & 0x3ffu);
      uint32_t desc;
      if (!grouping) {
        bool blk[4] = { false, false, false, false };
        for (int cf = 0; cf < 10; ++cf)
          if ((m >> cf) & 1u) { blk[(6 * cf) / 16] = true; blk[(6 * cf + 5) / 16] = true; }
        uint32_t touched = 0;
        for (int I = 0, tt = 0; I < 4; ++I)
          for (int J = 0; J <= I; ++J, ++tt) if (blk[I] && blk[J]) touched |= 1u << tt;
        desc = m | ((uint32_t)lane << 10) | (touched << 16);
      } else {
        uint32_t a = 0, nblk = 0, wdt = 0, holes = 0;
        if (m) {
          a = (uint32_t)(__ffs((int)m) - 1);
          wdt = (uint32_t)(31 - __clz((int)m)) - a + 1u;
          nblk = (6u * wdt + 15u) / 16u;
          holes = (uint32_t)__popc(m) != wdt ? 1u : 0u;
        }
        desc = gp_desc(m, (uint32_t)lane, a, nblk, holes, wdt);
      }
      P.line_desc[wd.line_off + s] = desc;
      lane += run;
    }
    if (grouping) {
      // the grouped sweep walks a tile's descriptors: group after group, inside a group by the number of blocks, lines without
      // elimination work last (a stable insertion sort, lba_pack.cpp:484-496)
      uint32_t* d = P.line_desc + wd.line_off + s0;
      for (int a2 = 1; a2 < nl; ++a2) {
        const uint32_t v = d[a2];
        const int kv = gp_mask(v) ? (int)(gp_group(v) * 8u + gp_blocks(v)) : 1 << 20;
        int b = a2;
        while (b > 0) {
          const uint32_t u = d[b - 1];
          const int ku = gp_mask(u) ? (int)(gp_group(u) * 8u + gp_blocks(u)) : 1 << 20;
          if (ku <= kv) break;
          d[b] = u; --b;
        }
        d[b] = v;
      }
    }
    const int rounds_log2 = min_lanes >= 4 ? 0 : min_lanes >= 2 ? 1 : 2;
    Tile tl;
    tl.line_begin = wd.line_off + s0; tl.nlines = (int16_t)nl;
    tl.flags = (int16_t)(multi | (rounds_log2 << 1) | (max_run << 3));
    tl.item_off = item_base + (int)F[t]; tl.nitems = (int)(F[t + 1] - F[t]);
    P.tiles[gt] = tl;
  }
}
#undef TILE_STAMP
__host__ __device__ inline size_t build_tiles_lds_bytes(int L, int ntiles_max, int stage_m = -1, int grouping = 0) {
  const size_t a = (size_t)(L > ntiles_max + 1 ? L : ntiles_max + 1);
  const size_t n = 4 * a + 4 * 260 + 4 * ((size_t)ntiles_max + 2) + 64;
  if (stage_m >= 0) return n + 4 * ((size_t)L + 1) + (grouping ? 4 : 8) * (size_t)L + (((size_t)L + 15) & ~(size_t)15) + (((size_t)stage_m + 15) & ~(size_t)15) + 64;
  return n + 256 * 4 * kMaxFreeCams;
}

}  // namespace slslam
#endif
