// slslam_amd/csrc/lba_types.h — HBM data layout of a batch of LBA windows.
//
// Host code (lba_pack.cpp) builds these arrays once per batch ("LBAProblem::build" stage,
// reference src/lba_problem.cpp:54-93); the kernels in lba_kernels.hip only read them, except for
// the double-buffered parameter records and the per-window LM state.
#ifndef SLSLAM_LBA_TYPES_H_
#define SLSLAM_LBA_TYPES_H_

#include <stdint.h>

namespace slslam {

enum { kRunning = -1 };                 // LMState.status while the window is still iterating
enum { kNumericalFailure = 4 };         // = SLSLAM_NUMERICAL_FAILURE (include/slslam_hip.h)
enum { kMaxTrace = 64 };                // iteration records kept per window
enum { kLineRec = 11 };                 // doubles per line record: u[4], trig[7] (88 B, no pad: the sweeps stream whole records)
enum { kLineElim = 22 };               // doubles per line kept by the elimination: K[10], D2[4], g[4] and - for the streaming
                                       // back-substitution and the global-memory path only - u[4] = K g; the record stride is
                                       // BatchPtrs.line_elim_stride (18 without u: the back-substitution reads whole records)
enum { kLeD2 = 10, kLeG = 14, kLeU = 18 };
enum { kCamRec = 6 };                   // doubles per camera record: w[3], t[3]
enum { kSlabScalars = 8 };              // per-chunk scalars written by the linearise kernel
enum { kMaxCams = 64, kMaxFreeCams = 20 };

// per-chunk scalar slots (linearise kernel)
enum { kScCost = 0, kScFixedCost = 1, kScGradMaxLine = 2, kScXn2Line = 3, kScFail = 4 };
// per-chunk scalar slots (back-substitution kernel)
enum { kBsModel = 0, kBsDn2 = 1, kBsXn2 = 2, kBsStride = 4 };

struct WinDesc {
  int C, Cf, L, M;        // cameras, free cameras, lines, observations
  int cam_off;            // first camera record
  int line_off;           // first (sorted) line record
  int obs_off;            // first (sorted) observation
  int tile_off, ntiles;
  int chunk_off, nchunks;
  int n;                  // 6 * Cf, order of the reduced camera system
  int sys_off;            // offset (doubles) of this window's y_c vector
  int nfree_params;       // 6 Cf + 4 (free lines with >= 1 kept block)
  int nkept;              // residual blocks in the reduced program
  int slab_off;           // the window's first chunk slab (its chunks' slabs follow each other, in the order of their positions in the window)
  int map_off;            // this window's table in BatchPtrs.sys_map (entry of a chunk partial -> place in the reduced solve's LDS image)
};

// A tile is one 64-lane pass over a run of consecutive (sorted) lines.  Every line owns a run of
// max(k, 1) consecutive lanes (k = its observations), lane j of the run handles observation
// line_ptr[l] + j.  Runs are bin-packed into the four 16-lane rows of the wave so that a run only
// crosses a row boundary when it starts on one (lines with more than 16 observations), which keeps the
// per-line reductions on row-local DPP shifts.  lane_map[64 t + lane] = slot | j << 8 | skew << 15 with slot = line - line_begin
// (0xFF: idle lane), j < 64 the position in the run, skew: see the diagonal block of the elimination sweep.
enum { kTileMultiRow = 1 };             // Tile.flags bit 0: some line of the tile spans several rows
struct Tile {
  int line_begin;         // global sorted line index of the first line
  int16_t nlines;
  int16_t flags;          // bit 0 kTileMultiRow | bits 1-2: log2(sin/cos rounds) = lines with < 4 lanes | bits 3-7: longest in-row run
  int item_off;           // off-diagonal camera-pair work items of this tile
  int nitems;
};
constexpr int tile_trig_rounds(int flags) { return 1 << ((flags >> 1) & 3); }
constexpr int tile_max_run(int flags) { return (flags >> 3) & 31; }

struct Chunk {
  int win;                    // -1: an unused entry (a refillable batch launches as many chunk workgroups as its array has room for; the kernels return on these)
  int tile_begin, tile_end;   // global tile indices
  int slab_off;               // doubles; slab = [S tri(n)] [b n] [g n] [hdiag n] [scalars]
  int id;                     // WinDesc.chunk_off + position in the window: where the chunk's partial sums (bs_part, cost_part) go.  The ARRAY
                              // of chunks is in dispatch order - the long chunks of every window first, see finalize - so blockIdx.x is not the id
};

// Levenberg-Marquardt state of one window (restates the locals of Ceres 1.7
// TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy; policy table in DESIGN.md §5).
struct LMState {
  double radius;
  double decrease_factor;
  double cost;              // reduced-program cost at the accepted point
  double x_norm;
  double fixed_cost;
  double initial_cost;      // incl. fixed cost
  double min_cost;          // min over recorded iteration costs (incl. fixed cost)
  double abs_grad_tol;
  double grad_max;          // |g|_inf at the accepted point
  double cam_model, cam_dn2, cam_xn2;   // camera parts of the step statistics (reduced solve)
  int status;               // kRunning or a SLSLAM_* termination type
  int cur;                  // which parameter buffer holds the accepted point
  int iter;                 // iterations recorded so far
  int n_success, n_unsuccess, n_invalid;
  int solve_failed;
  int need_grad_check;      // gradient at the accepted point not yet tested
  int ntrace;
  int same_point;           // the last step was rejected: the next linearisation is at the same point (new radius only)
  int fresh;                // no evaluation yet: the first elimination sweep also plays the role of Ceres' initial evaluation
  int pad;
};

struct IterRec {            // same fields as slslam_iteration
  int iteration, step_is_valid, step_is_successful, pad;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius,
         model_cost_change;
};

struct Policy {             // numeric policy, by value into every kernel that needs it
  double huber_delta, baseline;
  double initial_radius, max_radius, min_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  int max_num_iterations, max_invalid, jacobi_scaling;
  int keep_jacobian;           // 1: the grouped elimination sweep leaves J_c^T J_l of every observation (BatchPtrs.fstore) and the line blocks
                               // (BatchPtrs.line_h) in memory and the sweep after a rejected step starts from them instead of linearising again
  int store_f;                 // 1: spill F blocks for the streaming back-substitution (variant B)
  int debug_flags;             // timing experiments only (environment SLSLAM_DEBUG_ABLATE; results are wrong when set): bit 0 skip the
                               // matrix-core phase, bit 1 fetch operands without MFMA, bit 2 skip the camera-record atomics
};

// Everything the kernels need, passed by value.
struct BatchPtrs {
  const WinDesc* wins;
  const Tile* tiles;
  const Chunk* chunks;
  const uint16_t* lane_map;   // [ntile][64] lane -> (line slot, position in the line's run), see Tile
  const int32_t* lane_ctx;    // [ntile][64][4] per lane of a tile: sorted line | first observation of the line | j (bits 0-5), k (6-12), lane has a line (13),
                              // skew (14), line flags (16-23), line slot (24-31) | the tile's lane-th line descriptor: lane_map, line_ptr, line_flags and
                              // line_desc resolved on the host, one 16-byte load per lane and tile (fetch_tile)
  const uint8_t* items;       // 2 bytes per item: (lane_i, lane_j), camera(lane_i) <= camera(lane_j)
  const uint32_t* line_desc;  // [nline] matrix-core elimination: free-camera mask (bits 0-9, 0 for a constant line) | first lane of the
                              // line's run in its tile << 10 | accumulator tiles the line updates << 16
  // cameras
  double* cam_x;              // [ncam][2][6]
  double* cam_scale;          // [ncam][6]
  double* cam_tab;            // [ncam][2][kCamTab] R | JL | t of the pose in cam_x[.][buf]: built once per point (first sweep of a solve; the reduced solve for the candidate) and read by the sweeps of every chunk instead of being rebuilt by each of them; nullptr: every sweep builds its own
  const int* cam_cf;          // [ncam] index among the window's free cameras, or -1
  // lines (sorted order)
  double* line_x;             // [2][nline][12]: buffer-major, so that a sweep reading one buffer of consecutive lines streams dense memory
  double* line_scale;         // [nline][4]
  const int* line_ptr;        // [nline+1] first sorted observation of each line
  const int* line_flags;      // [nline] bit0: constant
  const int* line_win;        // [nline]
  // observations (sorted, structure of arrays)
  const double* ob;           // [4][ob_stride] double2: (x,y) of the four observed endpoints
  const int* ob_cam;          // [nobs] window-local camera id
  long long ob_stride;
  // per-chunk / per-window work areas
  double* slab;               // linearise/Schur partials
  double* slab_sum;           // [nwin][slab_sum_stride] per-window sum of the chunk partials (k_slab_reduce), nullptr when unused
  long long slab_sum_stride;
  int slab_sum_image;         // 1: slab_sum holds, per window, the LDS image of the reduced solve (A | b | g | hdiag scattered by sys_map, zeros elsewhere, the scalars behind it) - the solve copies it in one sweep of loads (default sweeps); 0: the slab layout
  const unsigned short* sys_map;   // per distinct n: chunk-partial entry -> place in the reduced solve's LDS image (WinDesc.map_off)
  double* bs_part;            // [nchunk][kBsStride]
  double* cost_part;          // [nchunk]
  double* ysys;               // y_c per window (sys_off)
  double* fstore;             // [12][ob_stride] double2: F = (Jc^T Jl) K^T (6x4, row-major) of every coupled observation (store_f);
                              // keep_jacobian: [tile][12][64] double2, h = Jc'^T Jl (6x4, row-major, raw camera coordinates) of the observation of every lane - 12 KB a tile, contiguous
  double* line_elim;          // [nline][line_elim_stride]
  double* line_h;             // [nline][10] keep_jacobian: lower triangle of the line's block J_l^T J_l (scaled line coordinates, undamped)
  int line_elim_stride;
  LMState* state;
  IterRec* trace;             // [nwin][kMaxTrace]
  unsigned long long* iter_counter;   // LM iterations executed by the batch since the counter was cleared
  unsigned int* active_counter;       // windows still iterating after the last k_lm_update (host early-out)
  const double* cam_x0;       // [ncam][6] initial camera poses (reset)
  const double* line_u0;      // [nline][4] initial line parameters (reset)
  int nwin, nchunk, nline, ncam;
  unsigned long long* dbg_cycles;   // [nchunk * 2][16] phase timing of the matrix-core sweep (debug_flags bit 8), else unused
  int elim_mode;              // 0: per-wave LDS partial fed by ds_add_f64 (lba_kernels.h); 1: matrix-core elimination
                              // (lba_eliminate_mfma.h), slab layout sys_doubles_mfma()
  int elim_waves;             // waves per chunk workgroup of the sweeps (1 or 2)
};

// offset (doubles) of line ls's record in parameter buffer buf of BatchPtrs.line_x
#if defined(__HIPCC__)
__host__ __device__
#endif
inline long long line_rec(const BatchPtrs& p, long long ls, int buf) { return ((long long)buf * p.nline + ls) * kLineRec; }

}  // namespace slslam
#endif
