// slslam_amd/csrc/lba_eliminate_mfma_maps.h — index maps of the matrix-core elimination sweep (lba_eliminate_mfma.h):
// where an observation's F block sits in the LDS panel, where a lane finds its MFMA operand for a line, which entry of the
// stacked reduced system an accumulator register holds.  Shared by the kernels (elimination, reduced solve) and, compiled
// for the host, by the CPU test-suite, which replays a tile through them (tests/test_host_side.py).
#ifndef SLSLAM_LBA_ELIMINATE_MFMA_MAPS_H_
#define SLSLAM_LBA_ELIMINATE_MFMA_MAPS_H_

#include "lba_math.h"     // SLS_HD

namespace slslam {

enum { kPanelPlane = 400 };        // doubles per plane of the F panel: 64 lanes x 6 rows + 16 (stride == 16 mod 32)
enum { kPanelDoubles = 4 * kPanelPlane };        // the pad of every plane (rows 384..399) stays zero: the slab of "lane 64", what rows of
                                                 // cameras that do not see a line read
enum { kGatherLines = 32 };        // lines per tile the gather table holds (tiles with more derive the sources from the masks)
enum { kDiagRec = 33, kDiagB = 21, kDiagG = 27 };   // camera record: J_c'^T J_c' lower triangle [21] | b' [6] | g' [6]
enum { kPTiles = 10, kPTileDoubles = 256 };
enum { kMfmaMaxFree = 10 };

SLS_HD constexpr int sys_doubles_mfma(int n) { return kPTiles * kPTileDoubles + (n / 6) * kDiagRec; }

// free cameras with a row in the 16-row block I of the stacked system (camera cf owns rows 6 cf .. 6 cf + 5)
SLS_HD constexpr unsigned block_cam_mask(int I) {
  return I == 0 ? 0x007u : I == 1 ? 0x03cu : I == 2 ? 0x0e0u : 0x300u;
}
// accumulator tile (I, J), J <= I, numbered t = I (I + 1) / 2 + J
SLS_HD constexpr int ptile_I(int t) { return t >= 6 ? 3 : t >= 3 ? 2 : t >= 1 ? 1 : 0; }
SLS_HD constexpr int ptile_J(int t) { return t - (ptile_I(t) * (ptile_I(t) + 1)) / 2; }
// tiles owned by wave `w` of an NW-wave workgroup, slot e (-1: none).  Two waves: balanced by how often the lines of a
// sliding window touch each tile (lines are seen by runs of consecutive keyframes).
SLS_HD constexpr int ptile_of(int NW, int w, int e) {
  return NW == 1 ? (e < 10 ? e : -1)
       : w == 0 ? (e == 0 ? 0 : e == 1 ? 2 : e == 2 ? 3 : e == 3 ? 7 : e == 4 ? 6 : -1)      // (0,0) (1,1) (2,0) (3,1) (3,0)
                : (e == 0 ? 1 : e == 1 ? 4 : e == 2 ? 5 : e == 3 ? 8 : e == 4 ? 9 : -1);     // (1,0) (2,1) (2,2) (3,2) (3,3)
}
SLS_HD constexpr int ptile_slots(int NW) { return NW == 1 ? 10 : 5; }

// v_mfma_f64_16x16x4_f64 accumulator layout: register q of lane l holds row (l >> 4) + 4 q, column l & 15 of the tile
SLS_HD void acc_row_col(int t, int q, int lane, int* row, int* col) {
  *row = 16 * ptile_I(t) + (lane >> 4) + 4 * q;
  *col = 16 * ptile_J(t) + (lane & 15);
}

// entry (a, k) of the F block of the observation handled by lane `lane`: plane k, 6-row slab of the lane
SLS_HD constexpr int panel_store_index(int lane, int a, int k) { return k * kPanelPlane + lane * 6 + a; }

// Per-lane constants of the operand fetch for block I: row r = 16 I + (lane & 15) of the stacked system belongs to free
// camera cf = r / 6, entry a = r % 6; the lane wants column k = lane >> 4 of that camera's F block.
struct XLane {
  unsigned low[4];     // cameras below cf
  int cf[4];
  int off[4];          // k * kPanelPlane + a
};
SLS_HD XLane make_xlane(int lane) {
  XLane x;
  for (int I = 0; I < 4; ++I) {
    const int r = 16 * I + (lane & 15);
    const int cf = r / 6;
    x.cf[I] = cf;
    x.low[I] = (1u << cf) - 1u;
    x.off[I] = (lane >> 4) * kPanelPlane + (r - 6 * cf);
  }
  return x;
}
SLS_HD int popcount32(unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(v);
#else
  return __builtin_popcount(v);
#endif
}
// A line's free-camera observations are the first lanes of its run, ascending free index: the observation of camera cf
// is lane first_lane + (number of the line's free cameras below cf).  Returns the panel index of the lane's operand
// X[16 I + (lane & 15)][lane >> 4], `present` = the row's camera sees the line (else the operand is 0).
SLS_HD int panel_fetch(const XLane& xl, int I, unsigned mask, int first_lane6, bool* present) {
  *present = ((mask >> xl.cf[I]) & 1u) != 0u;
  return xl.off[I] + first_lane6 + 6 * popcount32(mask & xl.low[I]);
}
// (host replay of the device-side sources_from_mask / fetch_operands pair)
SLS_HD int panel_fetch_index(int lane, int I, unsigned mask, int first_lane) {
  const XLane xl = make_xlane(lane);
  bool present;
  const int idx = panel_fetch(xl, I, mask, 6 * first_lane, &present);
  return present ? idx : -1;
}

}  // namespace slslam
#endif  // SLSLAM_LBA_ELIMINATE_MFMA_MAPS_H_
