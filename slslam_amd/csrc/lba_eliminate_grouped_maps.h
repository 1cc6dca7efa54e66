// slslam_amd/csrc/lba_eliminate_grouped_maps.h — index maps of the grouped matrix-core elimination sweep
// (lba_eliminate_grouped.h): the line descriptor the packer writes for PackedWindow.grouping = 1, where an observation's F block
// sits in the LDS panel, where a lane finds its MFMA operand for a line, where a group-local accumulator entry goes in the chunk's
// slab.  Shared by the kernel and, compiled for the host, by the CPU test-suite, which replays whole windows through them
// (tests/test_host_side.py::test_grouped_elimination_maps_replay_windows).
#ifndef SLSLAM_LBA_ELIMINATE_GROUPED_MAPS_H_
#define SLSLAM_LBA_ELIMINATE_GROUPED_MAPS_H_

#if defined(__HIPCC__)
#define GP_HD __host__ __device__ inline
#else
#define GP_HD inline
#endif

namespace slslam {

enum { kGpSlab = 26 };                       // doubles per lane slab of the F panel: 6 rows x 4 columns + 2 (16-byte aligned rows, lane stride
                                             // 52 dwords: the b128 row stores of 64 lanes spread over the banks)
enum { kGpPanel = 65 * kGpSlab };            // 64 lanes + the zero slab (index 64) that absent cameras read
enum { kGpPersist = 6 };                     // accumulator tiles kept per group: block rows 0-2 of the group-local sum

// Line descriptor (grouping = 1): free-camera mask (bits 0-9; the observations of these cameras are the first lanes of the line's
// run, ascending free index) | first lane of the run << 10 | first free camera a << 16 | 16-row blocks nb of the rows
// 6 (hi - a + 1) the line touches, counted from a, << 20 | range has holes (some camera between a and hi does not see the
// line) << 23 | cameras in the range hi - a + 1 << 24.  0 mask: no elimination work (constant line / no free camera).
GP_HD constexpr unsigned gp_desc(unsigned mask, unsigned first_lane, unsigned a, unsigned nb, unsigned holes, unsigned width) {
  return mask | first_lane << 10 | a << 16 | nb << 20 | holes << 23 | width << 24;
}
GP_HD constexpr unsigned gp_mask(unsigned d) { return d & 0x3ffu; }
GP_HD constexpr unsigned gp_first(unsigned d) { return (d >> 10) & 63u; }
GP_HD constexpr unsigned gp_group(unsigned d) { return (d >> 16) & 15u; }
GP_HD constexpr unsigned gp_blocks(unsigned d) { return (d >> 20) & 7u; }
GP_HD constexpr unsigned gp_holes(unsigned d) { return (d >> 23) & 1u; }
GP_HD constexpr unsigned gp_width(unsigned d) { return (d >> 24) & 15u; }

// entry (a, k) of the F block of the observation handled by lane `lane` (doubles from the start of the panel)
GP_HD constexpr int gp_store_index(int lane, int a, int k) { return lane * kGpSlab + 4 * a + k; }

// Lane l wants X[16 r + (l & 15)][l >> 4] of a line: row rho = 16 r + (l & 15) of the group-local system belongs to the camera
// `slot` = rho / 6 places after the group's first one, entry rho % 6, column l >> 4 of that camera's F block.  gp_pre: its offset
// (doubles) from the slab of the camera at slot 0.
GP_HD constexpr int gp_slot(int lane, int r) { return ((16 * r + (lane & 15)) * 43) >> 8; }        // (16 r + (lane & 15)) / 6, exact below 64
GP_HD constexpr int gp_pre(int lane, int r) {
  return gp_slot(lane, r) * kGpSlab + ((16 * r + (lane & 15)) - 6 * gp_slot(lane, r)) * 4 + (lane >> 4);
}
// Panel index (doubles) of the operand of block r for the line with descriptor d; the zero slab when the row's camera does not
// see the line or lies past the line's last camera.  A line without holes has the observation of camera a + i in lane first + i.
GP_HD int gp_fetch_index(int lane, int r, unsigned d) {
  const int slot = gp_slot(lane, r);
  if (!gp_holes(d)) return slot < (int)gp_width(d) ? (int)gp_first(d) * kGpSlab + gp_pre(lane, r) : 64 * kGpSlab;
  const unsigned mask = gp_mask(d), cfb = gp_group(d) + (unsigned)slot;
  if (!((mask >> cfb) & 1u)) return 64 * kGpSlab;
  unsigned below = mask & ((1u << cfb) - 1u), cnt = 0;
  for (; below; below &= below - 1u) ++cnt;
  return (int)(gp_first(d) + cnt) * kGpSlab + gp_pre(lane, r) - slot * kGpSlab;
}
// Where register q of lane `lane` of the group-local accumulator tile (block row r, block column c <= r; group's first camera a)
// goes in the chunk's slab (layout of lba_eliminate_mfma_maps.h: tile (I, J) of the window's 64 x 64 system, entry q * 64 + lane
// = row (lane >> 4) + 4 q, column lane & 15); -1: outside the window's system or above the diagonal.
GP_HD int gp_flush_index(int r, int c, int a, int q, int lane, int n) {
  const int grow = 6 * a + 16 * r + (lane >> 4) + 4 * q, gcol = 6 * a + 16 * c + (lane & 15);
  if (grow >= n || gcol > grow) return -1;
  const int I = grow >> 4, J = gcol >> 4;
  return ((I * (I + 1)) / 2 + J) * 256 + ((grow & 15) >> 2) * 64 + (grow & 3) * 16 + (gcol & 15);
}

}  // namespace slslam
#endif
