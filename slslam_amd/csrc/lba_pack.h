// slslam_amd/csrc/lba_pack.h — host-side "build" stage of the LBA batch: validates the caller's
// arrays and reorders them into the HBM layout of lba_types.h.
//
// Mirrors what LBAProblem::build does for Ceres (reference src/lba_problem.cpp:54-93): one residual
// block per observation wired to camera block camera_index[i] and line block line_index[i]
// (:83-84), a block is constant if ANY observation flags it (:88-91).  Instead of M heap-allocated
// cost functions the build emits:
//   * lines bin-packed into the 16-lane rows of 64-lane tiles (a line owns one lane per observation),
//     observations grouped by line with the free-camera observations first (ascending free index),
//   * per tile the lane map and the list of off-diagonal camera-pair work items, and chunks (runs of
//     tiles handled by one wave).
// Pure host C++ (no HIP), so the CPU test-suite can check its invariants.
#ifndef SLSLAM_LBA_PACK_H_
#define SLSLAM_LBA_PACK_H_

#include <vector>
#include <cstdint>
#include "lba_types.h"
#include "../../include/slslam_hip.h"

namespace slslam {

struct PackedWindow {
  int C = 0, Cf = 0, L = 0, M = 0;
  int nfree_params = 0, nkept = 0;
  std::vector<int> cam_cf;          // [C]  free index or -1
  std::vector<double> cam_x;        // [C*6] initial
  std::vector<int> line_order;      // [L]  sorted position -> original line
  std::vector<int> line_flags;      // [L]  sorted; bit0 constant
  std::vector<double> line_u;       // [L*4] sorted; initial (a,b,g,t)
  std::vector<int> line_ptr;        // [L+1] sorted; window-local offsets into the sorted observations
  std::vector<int> ob_orig;         // [M]  sorted position -> original observation
  std::vector<int> ob_cam;          // [M]  sorted
  std::vector<double> ob;           // [8*M] four planes of (x, y) pairs: ob[(plane * M + o) * 2 + {0, 1}], plane = endpoint 0..3 of the observation
  std::vector<Tile> tiles;          // line_begin window-local; item_off window-local
  std::vector<uint16_t> lane_map;   // [64 per tile] lane -> line slot | position << 8 (0x00FF: idle)
  std::vector<uint8_t> items;       // 2 bytes per item
  std::vector<uint32_t> line_desc;  // [L] sorted: free-camera mask | first lane of the run << 10 | touched accumulator tiles << 16
                                    // (grouping = 1: ... | first free camera a << 16 | 16-row blocks of the line's camera range << 20)
  int grouping = 0;                 // 0: rows dealt to the tiles by pair-item count; 1: lines grouped by their first free camera (below)
  bool big = false;                 // beyond the tiled sweeps (> 20 free / 64 cameras, a line with > 64 observations): lba_big.h
  bool dup_free_obs = false;        // some free camera observes some line more than once (the reference's map never does)
  std::vector<double> params0;      // caller's original parameter vector (for lines/cams never touched)
};

// Returns SLSLAM_OK or an error status; on error `out` is unspecified.
// grouping = 1 (the grouped matrix-core elimination, lba_eliminate_grouped.h): the lines of a window follow each other by the
// FIRST free camera that sees them (a sliding window's lines are seen by runs of consecutive keyframes, so the reduced-system rows
// a line touches, counted from its first camera, fit a few 16-row blocks), the lines whose camera range needs a fourth block
// (more than 8 cameras) behind the others of their group; rows are bin-packed inside a group and fill the tiles in that order.
// ob_dest (optional): the four observation planes are written THERE (plane q: ob_dest->plane[q][2 o + {0, 1}], o = sorted position) instead of
// into out->ob, which stays empty - a batch that is refilled from host buffers packs straight into its pinned staging image
// (slslam_lba_batch_refill: one pass over the caller's observations, no second copy).
struct ObPlanes {
  double* plane[4];
  // raw != nullptr (a refill whose batch permutes the observations on the DEVICE, lba_api.hip::k_permute_obs): the window's observations are
  // copied there in the CALLER'S order, [8 M] doubles - one linear pass with the finite check - and the planes are not written
  double* raw = nullptr;
};
int pack_window(const slslam_lba_window* w, PackedWindow* out, int grouping = 0, const ObPlanes* ob_dest = nullptr);
// The same window packed again with another grouping (the caller's arrays are rebuilt from the packed ones).
int repack_window(const PackedWindow& P, int grouping, PackedWindow* out);

// Splits ntiles into chunks of at most tiles_per_chunk tiles; returns boundaries [nchunks+1].
std::vector<int> chunk_boundaries(int ntiles, int tiles_per_chunk);
// `nchunks` chunks whose tile counts are in the proportions weights[0 .. nchunks) (graded sizes: a launch whose wave slots each run several chunks
// ends with everybody's LAST chunk - the shorter that one is, the less the slots wait for the slowest); every chunk gets at least one tile
std::vector<int> chunk_boundaries_graded(int ntiles, int nchunks, const int* weights);

}  // namespace slslam
#endif
