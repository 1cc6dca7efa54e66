/* include/slslam_hip.h — C ABI of the MI355X-native SLSLAM optimisation back-end.
 *
 * Drop-in boundary for the reference's two optimisation hot paths.  The reference is C++ and has
 * no FFI of its own; what a maintainer binds is the triple
 *     XProblem::build(ceres::Problem*)  /  XProblem::set_options(ceres::Solver::Options*)  /
 *     ceres::Solve(options, &problem, &summary)
 * at the three call sites reference src/slam.cpp:643-663 (motion_only_ba), :924-952
 * (bundle_adjustment) and :1283-1293 (pose_optimization).  Every entry point below cites the
 * reference interface it replaces.  Plain pointers and sizes only; no C++/torch types.
 *
 * The host-side C++ mirror of the reference classes (slslam_amd/host/lba_problem.h,
 * po_problem.h and the minimal ceres facade) marshals into these calls; see INTEGRATION.md.
 *
 * All solves run on the GPU in hand-written HIP kernels (slslam_amd/csrc).  There is no CPU
 * fallback: without a usable HIP device every compute entry point returns SLSLAM_ERR_NO_DEVICE.
 */
#ifndef SLSLAM_HIP_H_
#define SLSLAM_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
enum {
  SLSLAM_OK = 0,
  SLSLAM_ERR_INVALID_ARGUMENT = 1,
  SLSLAM_ERR_NO_DEVICE = 2,        /* no HIP device / HIP runtime error */
  SLSLAM_ERR_HIP = 3,
  SLSLAM_ERR_UNSUPPORTED = 4,      /* problem shape outside what the kernels support */
  SLSLAM_ERR_STATE = 5,            /* call sequence violated (e.g. solve before finalize) */
  SLSLAM_ERR_NO_MEMORY = 6         /* a host allocation failed while building a window (std::bad_alloc caught at the boundary) */
};

/* Termination of one solve: mirrors ceres::Solver::Summary::termination_type of Ceres 1.7.
 * The reference never inspects it (src/slam.cpp:663,944,1293); reported for diagnostics.
 * Known deviation: when max_num_iterations ends a solve right after an ACCEPTED step, Ceres 1.7 still evaluates the
 * gradient at the new point inside that iteration and can report GRADIENT_TOLERANCE; here the gradient of a point is
 * evaluated by the next linearisation, which never comes, so such a solve reports NO_CONVERGENCE and its last trace
 * record keeps the previous point's gradient_max_norm.  Parameters, costs and step counts are the same. */
enum {
  SLSLAM_NO_CONVERGENCE = 0,       /* max_num_iterations reached */
  SLSLAM_GRADIENT_TOLERANCE = 1,
  SLSLAM_FUNCTION_TOLERANCE = 2,
  SLSLAM_PARAMETER_TOLERANCE = 3,
  SLSLAM_NUMERICAL_FAILURE = 4,    /* parameters are left untouched, as Ceres does */
  SLSLAM_MIN_RADIUS = 5
};

/* ------------------------------------------------------------------ options
 * Replaces: ceres::Solver::Options as filled by LBAProblem::set_options
 * (reference src/lba_problem.cpp:95-132) and POProblem::set_options (src/po_problem.cpp:67-77),
 * plus the constants hard-coded in the hot path (baseline 0.12: src/lba_problem.h:101;
 * Huber scale 1/406.05: src/lba_problem.cpp:78) and the gflags read by it (FLAGS_robust,
 * src/lba_problem.cpp:35; FLAGS_max_num_iter, src/slam.cpp:647,928).
 * Trust-region constants are the Ceres 1.7 defaults the reference leaves untouched. */
typedef struct slslam_solver_options {
  int    max_num_iterations;            /* lba_param_t.num_iterations / POProblem n_iter        */
  double huber_delta;                   /* 1/406.05 when FLAGS_robust, <= 0 disables the loss   */
  double baseline;                      /* stereo baseline, 0.12                                */
  double initial_trust_region_radius;   /* 1e4   */
  double max_trust_region_radius;       /* 1e16  */
  double min_trust_region_radius;       /* 1e-32 */
  double min_relative_decrease;         /* 1e-3  */
  double min_lm_diagonal;               /* 1e-6  */
  double max_lm_diagonal;               /* 1e32  */
  int    max_num_consecutive_invalid_steps; /* 5 */
  double function_tolerance;            /* 1e-6  */
  double gradient_tolerance;            /* 1e-10 (relative to the initial max-norm, Ceres 1.7)  */
  double parameter_tolerance;           /* 1e-8  */
  int    jacobi_scaling;                /* 1     */
  int    use_graph;                     /* 1: replay the LM iteration as a captured hipGraph    */
  int    chunks_per_window;             /* 0 = auto; n > 0: n equal chunks (waves cooperating on one window); n < 0: -(1000 r + c), c chunks
                                           of GRADED sizes made for r rounds of the chip's wave slots - long chunks first, short ones last -
                                           as the automatic choice cuts the windows of a batch whose slots each run several chunks (what
                                           slslam_lba_batch_window_chunks reports for such a window: pass it back to get the same cut)     */
  int    reuse_elimination;             /* 0: the back-substitution re-linearises (HBM traffic ==
                                           algorithmic); 1: it streams Jacobian blocks the elimination
                                           spilled to HBM (+192 B per coupled observation)         */
  int    po_factor_fp32;                /* pose graph only: 1 = factor the normal matrix in fp32 (MFMA f32);
                                           residuals, gradient, costs and LM bookkeeping stay fp64
                                           (implies po_dense_factor)                                    */
  int    po_dense_factor;               /* pose graph only: 0 (default) = structured factorisation: chains of poses
                                           eliminated concurrently (block tridiagonal), dense MFMA Cholesky of the
                                           junction poses only; 1 = dense MFMA Cholesky of the whole normal matrix   */
  int    lba_fused_motion_only;         /* 1 (default): a batch whose windows all have one free camera and only constant lines
                                           (SLAM::motion_only_ba) is solved by one launch, one wave per window, the 6 x 6
                                           system in registers; 0: the general elimination / back-substitution path     */
  int    lba_elimination;               /* how the elimination sweep accumulates the reduced camera system: 0 (default) = 4 for a batch
                                           that fills the chip with long chunks (>= 16 tiles per resident wave) when its conditions
                                           hold, else 1; 1 = per-wave partial in LDS fed by fp64 atomics; 2 / 3 = Schur outer products on the matrix cores
                                           (v_mfma_f64_16x16x4_f64, accumulator tiles in registers) with one / two waves per chunk
                                           workgroup - for windows with <= 10 free cameras and one observation per (line, free
                                           camera), else the default sweep runs.  Same results to round-off; slower on MI355X
                                           (DESIGN.md section 7), kept as a measured alternative.  4 = matrix cores with GROUP-LOCAL
                                           accumulators: the lines of a window are packed by their first free camera and the wave
                                           keeps only the 48 x 48 sum of the current group in registers (same conditions)         */
  int    lba_keep_jacobian;             /* 0 (default): every elimination sweep linearises.  1: the sweep that follows a REJECTED step
                                           does not - the point has not moved, only the trust-region radius has: it replays the blocks
                                           J_c^T J_l of every observation and the line blocks the last linearising sweep left in HBM
                                           (+192 B per observation), as ceres::TrustRegionMinimizer evaluates the Jacobian only after
                                           a successful step.  Grouped sweep (lba_elimination 4) only; same results to round-off.
                                           Measured SLOWER on MI355X (the stores cost the linearising sweeps more than the replays
                                           save): kept as a tested alternative                              */
  int    refill_headroom_percent;       /* 0 (default): the batch holds exactly its windows.  > 0: a batch for a STREAM of windows - its device
                                           arrays get that much room beyond what the first windows need, a pinned host image of them is
                                           kept, results come back through pinned buffers - so that slslam_lba_batch_refill can replace
                                           all its windows without allocating, re-capturing the solve graph or synchronising          */
  int    host_threads;                  /* host threads for the per-window host work of a batch (packing = the LBAProblem::build stage,
                                           layout, result copies): 0 = automatic (up to 8 for batches of 64 windows or more), 1 = the
                                           calling thread only                                                                         */
  int    reproducible;                  /* 0 (default): the elimination sweep (lba_elimination = 0) and the cut of a window into chunks
                                           (chunks_per_window = 0) are chosen for the BATCH - fastest, but a window's result bytes then
                                           depend on its company (the sums run in another order).  1: both become functions of the window
                                           alone - lba_elimination = 0 means 1, a requested matrix-core sweep the batch cannot take is
                                           SLSLAM_ERR_UNSUPPORTED instead of a fallback, chunks of at most 34 tiles graded for three
                                           rounds of the wave slots (what the automatic choice gives a batch that fills the chip) - so
                                           that 1 / 2 / 4 / 8-rank runs and one-window batches return identical bytes with no caller
                                           bookkeeping                                                                                  */
  int    lba_precision;                 /* 0 (default): fp64 throughout, the reference's arithmetic.  1: MIXED - the steady elimination
                                           sweeps form the CAMERA Jacobian of an observation in packed fp32 (and park it as floats);
                                           geometry, residuals, the LINE Jacobian, every block product and accumulation (line blocks,
                                           reduced camera system, gradients), the cost, the candidate evaluation and all trust-region
                                           bookkeeping stay fp64, as does the first sweep of a solve.  Which parts may be float was
                                           measured (a float line Jacobian changes accept / reject decisions: its depth column carries
                                           d = cos t / sin t, reference src/lba_problem.h:63).  Grouped sweep only (lba_elimination 0 or
                                           4, at most 10 free cameras); opt-in; results within the tolerance stated in DESIGN.md 4 ("Mixed precision") and
                                           tests/test_gpu_lba.py::test_mixed_precision_solves of the fp64 path and of the oracle.
                                           MEASURED: buys nothing (0.997 x the fp64 step) - fp32 line Jacobians change LM accept / reject
                                           decisions, and what is left to fp32 is a few per cent of the sweep.  Kept as a tested option  */
  int    device_build;                  /* who runs the LBAProblem::build stage (reference src/lba_problem.cpp:54-93) of a REFILL
                                           (slslam_lba_batch_refill, slslam_lba_stream_submit): 0 (default) = the device when it can
                                           (csrc/lba_device_build.h: at most 64 cameras and 65534 lines per window; the windows' arrays are
                                           read by the GPU where they are when they lie in page-locked memory - slslam_pinned_alloc /
                                           slslam_pinned_register - else from a pinned staging copy the host threads make), with the host
                                           packer as the fallback; -1 = always the host packer (lba_pack.cpp, options.host_threads
                                           threads).  Same layout, same solved bytes either way                                          */
} slslam_solver_options;

/* Fills every field with the configuration the reference runs (robust loss on, 10 iterations). */
void slslam_default_options(slslam_solver_options* opt);

/* ------------------------------------------------------------------ summary
 * Replaces: the four ceres::Solver::Summary fields the reference reads back
 * (reference src/slam.cpp:949-952) + diagnostics. */
typedef struct slslam_summary {
  int    num_successful_steps;
  int    num_unsuccessful_steps;
  double initial_cost;
  double final_cost;
  double fixed_cost;
  int    termination_type;
  int    num_free_parameters;
  int    num_residual_blocks;
} slslam_summary;

/* One record per LM iteration (index 0 = initial evaluation), for parity tests. */
typedef struct slslam_iteration {
  int    iteration;
  int    step_is_valid;
  int    step_is_successful;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  double model_cost_change;
} slslam_iteration;

/* ------------------------------------------------------------------ LBA window
 * Replaces: ceres::lba_param_t (reference src/lba_problem.h:123-130) + the five arrays handed
 * to LBAProblem through set_line_index / set_camera_index / set_fixed_index / set_observations /
 * set_parameters (src/lba_problem.h:153-157); layouts as documented at src/lba_problem.h:188-196
 * and built at src/slam.cpp:899-920.  All pointers are HOST pointers owned by the caller. */
typedef struct slslam_lba_window {
  int num_cameras;                /* lba_param_t.num_cameras      */
  int num_lines;                  /* lba_param_t.num_lines        */
  int num_observations;           /* lba_param_t.num_observations */
  const int*    camera_index;     /* [M]                          */
  const int*    line_index;       /* [M]                          */
  const int*    fixed_index;      /* [2M]: [2i] camera constant, [2i+1] line constant */
  const double* observations;     /* [8M]: x0 y0 x1 y1 (camera 0), x2 y2 x3 y3 (camera 1) */
  double*       parameters;       /* [6C + 4L] in/out: cameras (w,t) then lines (a,b,g,t) */
} slslam_lba_window;

/* Replaces: LBAProblem::build + LBAProblem::set_options + ceres::Solve for ONE window
 * (reference src/slam.cpp:924-944 and :643-663).  Synchronous; parameters solved in place.
 * trace may be NULL; at most trace_cap records are written and *trace_len gets the count. */
int slslam_lba_solve(const slslam_lba_window* window, const slslam_solver_options* opt,
                     slslam_summary* summary, slslam_iteration* trace, int trace_cap, int* trace_len);

/* ---- batched form: many independent windows resident in HBM, solved in lock-step.
 * This is the throughput path (independent sliding windows / sequence shards);
 * each window goes through exactly the per-window algorithm of slslam_lba_solve. */
typedef struct slslam_lba_batch slslam_lba_batch;

/* device < 0 selects the current HIP device. */
int  slslam_lba_batch_create(int device, slslam_lba_batch** out);
void slslam_lba_batch_destroy(slslam_lba_batch* b);
/* Replaces LBAProblem::build for one more window: validates, copies and reorders the arrays on
 * the host (observations grouped by line).  Returns the window's index in *index. */
int  slslam_lba_batch_add(slslam_lba_batch* b, const slslam_lba_window* window, int* index);
/* Uploads every added window to HBM; no windows can be added afterwards (all of them can be REPLACED: slslam_lba_batch_refill). */
int  slslam_lba_batch_finalize(slslam_lba_batch* b, const slslam_solver_options* opt);
/* Replaces ceres::Solve for all windows: enqueues the complete LM solve on `stream`
 * (a hipStream_t passed as void*, NULL = default stream) and returns without synchronising. */
int  slslam_lba_batch_solve(slslam_lba_batch* b, void* stream);
/* Restores the initial parameters on the device (for repeated timing of the same inputs). */
int  slslam_lba_batch_reset(slslam_lba_batch* b, void* stream);
/* Trust-region steps (successful + unsuccessful, the count the reference accumulates at
 * src/slam.cpp:949-950) executed by all windows since the counter was last cleared; synchronises
 * the stream.  Lets a bench count iterations over several solves without per-window downloads. */
int  slslam_lba_batch_iterations(slslam_lba_batch* b, void* stream, long long* iterations, int clear);
/* Blocks until the stream's work is done and copies parameters + summaries back to the host. */
int  slslam_lba_batch_download(slslam_lba_batch* b, void* stream);
/* The two halves of download: _async enqueues the export and the device-to-host copies on `stream` and returns; _wait blocks until
 * they have arrived (results of a refillable batch land in pinned memory, so the copies overlap whatever else the host does). */
int  slslam_lba_batch_download_async(slslam_lba_batch* b, void* stream);
int  slslam_lba_batch_wait(slslam_lba_batch* b);
/* Replaces LBAProblem::build for ALL windows of a finalized batch at once - the next `n` windows of a stream take the place of the
 * present ones (reference: the five arrays a caller hands over per window, src/slam.cpp:899-921; n must equal the batch's window
 * count).  The batch must have been finalized with refill_headroom_percent > 0.  Packs the windows on options.host_threads host
 * threads straight into the batch's pinned host image, uploads the arrays with asynchronous copies on `stream`, rebuilds the tile
 * contexts and the initial parameter buffers on the device and resets the LM state - no allocation, no synchronisation, the captured
 * solve graph stays valid.  The windows are cut into chunks by the policy finalize resolved, so a refilled batch returns the bytes a
 * fresh batch of the same windows returns.  SLSLAM_ERR_UNSUPPORTED (batch unchanged, still solvable): the windows do not fit the
 * room the arrays have, or need another path / sweep - build a new batch then.
 * Host arrays are read before the call returns - EXCEPT arrays in page-locked memory (slslam_pinned_alloc / _register) when the device
 * builds (options.device_build = 0): those are read by the GPU after the call returns and must stay valid until the results have been
 * waited for.  When the device builds, what only the build can find out arrives with the results: a window with bad input, of a shape
 * for the host path, or a refill that does not fit the room is flagged - slslam_lba_batch_get_parameters / _get_summary of such a window
 * return SLSLAM_ERR_INVALID_ARGUMENT / SLSLAM_ERR_UNSUPPORTED (the other windows are solved); sizes that cannot fit are still refused here.
 * `stream` must be the stream the batch is solved on. */
int  slslam_lba_batch_refill(slslam_lba_batch* b, const slslam_lba_window* windows, int n, void* stream);
/* After download: solved parameters of window `index` in the caller's original layout. */
int  slslam_lba_batch_get_parameters(const slslam_lba_batch* b, int index, double* parameters);
int  slslam_lba_batch_get_summary(const slslam_lba_batch* b, int index, slslam_summary* summary);
int  slslam_lba_batch_get_trace(const slslam_lba_batch* b, int index, slslam_iteration* trace,
                                int trace_cap, int* trace_len);
/* Copies all solved parameters, windows concatenated in add order, into a DEVICE buffer
 * (for a collective on the results without a host round trip). */
int  slslam_lba_batch_export_device(slslam_lba_batch* b, double* device_out, void* stream);
/* Problem-size accounting for throughput reporting: totals over the batch. */
int  slslam_lba_batch_counts(const slslam_lba_batch* b, long long* num_windows, long long* num_cameras,
                             long long* num_free_cameras, long long* num_lines, long long* num_observations);
/* After finalize: which device path the batch takes.  One window beyond the tiled sweeps (more than 20 free / 64 cameras, a line
 * with more than 64 observations: the reference's W = 40 study) takes the global-memory path (built for single large windows,
 * not for batch throughput).  A batch that mixes such windows with ordinary ones is solved as two batches side by side - the
 * ordinary windows on the tiled sweeps, the oversize ones on the global-memory path on a stream of the batch's own, joined to
 * the caller's before the call returns control of it: SLSLAM_PATH_MIXED. */
enum { SLSLAM_PATH_TILED = 0, SLSLAM_PATH_FUSED_MOTION_ONLY = 1, SLSLAM_PATH_GLOBAL_MEMORY = 2, SLSLAM_PATH_MIXED = 3 };
int  slslam_lba_batch_path(const slslam_lba_batch* b, int* path);
/* After finalize: the elimination sweep the batch runs, as a value of slslam_solver_options.lba_elimination that reproduces it
 * (1, 2, 3 or 4; never 0): what the automatic choice resolved to, or what the request fell back to.  A window's result depends on
 * this value (the sweeps sum in different orders) - a caller that wants one window of a large batch again, bit for bit, in a batch
 * of its own passes it back together with the window's chunk count.  0 for batches that do not take the tiled sweeps. */
int  slslam_lba_batch_elimination(const slslam_lba_batch* b, int* mode);
/* After finalize: the number of chunks (waves cooperating on the window's observation sweeps) window `index` was cut into.  A
 * window's result is a function of its inputs, the options and this number only (the chunk partials are summed in chunk order):
 * a window solved again with chunks_per_window set to it - in any batch that keeps the chunk count at 8 or below, or above 8 -
 * reproduces its result bit for bit, which is how results are compared across the ranks of a multi-GPU run (bench.py).
 * NEGATIVE: -(1000 r + c): c chunks of graded sizes made for r rounds of the wave slots (chunks_per_window set to it asks for the same cut). */
int  slslam_lba_batch_window_chunks(const slslam_lba_batch* b, int index, int* num_chunks);
/* Device time (ms) spent in each kernel family during the last solve, measured with HIP events
 * on the solve stream while profiling is enabled (solves are then launched eagerly instead of
 * replaying the captured graph); times accumulate over solves until set_profiling is called again.
 * names: 0 linearise+schur, 1 reduced solve, 2 back-substitution (includes the candidate cost and the candidate
 * lines' sin/cos table unless reuse_elimination is set), 3 line trig, 4 candidate cost (reuse_elimination only),
 * 5 LM update, 6 init.  launches[i] receives the number of launches. */
int  slslam_lba_batch_set_profiling(slslam_lba_batch* b, int enable);
int  slslam_lba_batch_kernel_times(const slslam_lba_batch* b, double ms[8], int launches[8]);

/* Test hook: evaluate residuals / Jacobians (after the Huber corrector, before Jacobi scaling)
 * of window `index` at its CURRENT device parameters, returned in the caller's observation order:
 * residuals[4M], j_cam[24M] (row-major 4x6), j_line[16M] (row-major 4x4), cost[1]. */
int  slslam_lba_batch_linearise(slslam_lba_batch* b, int index, double* residuals, double* j_cam,
                                double* j_line, double* cost);

/* ---- page-locked host memory the GPU reads and writes IN PLACE.
 * Replaces: the `new int[...]` / `new double[...]` of the caller's five arrays per window (reference src/slam.cpp:899-903) for a caller
 * that streams windows: arrays allocated here (or in a range registered once with slslam_pinned_register - an arena the caller owns) are
 * read by the device straight over the host link during slslam_lba_batch_refill / slslam_lba_stream_submit (no host thread touches the
 * data; 57 GB/s measured, tools/micro/zero_copy_bench.hip) and the solved `parameters` are written back the same way.  Such arrays
 * must stay valid and unmodified until the results have been waited for / collected.  Arrays anywhere else keep working: the host threads
 * copy them into a pinned staging buffer first.  slslam_pinned_contains: 1 when [p, p + bytes) lies in one such range. */
int  slslam_pinned_alloc(size_t bytes, void** out);
int  slslam_pinned_free(void* p);
int  slslam_pinned_register(void* p, size_t bytes);
int  slslam_pinned_unregister(void* p);
int  slslam_pinned_contains(const void* p, size_t bytes);
/* The three index arrays of a window narrowed to ONE 32-bit word per observation - line | camera << 16 | camera constant << 24 |
 * line constant << 25 - the form the device build works on (16 -> 4 bytes per observation over the host link: what the staging copy
 * produces on the way).  SLSLAM_ERR_UNSUPPORTED when a camera index exceeds 255 or a line index 65534. */
int  slslam_pack_indices(int n, const int* camera_index, const int* line_index, const int* fixed_index, unsigned int* packed);

/* ---- a STREAM of windows (BASELINE config 4: many independent windows arriving in host memory): `depth` refillable batches in
 * flight on HIP streams of their own, so that packing (host threads), upload (copy engine), solve and download of consecutive
 * batches overlap.  submit() = LBAProblem::build + ceres::Solve for `n` windows, asynchronous: it returns when the windows' arrays
 * have been read or handed to the GPU (`parameters` of each window must stay valid - it is written by collect -, and so must arrays in
 * page-locked memory, which the GPU reads in place: slslam_pinned_alloc); collect() waits for that submit's
 * results and writes every window's solved parameters in place (the in/out contract of reference src/slam.cpp:957-972) and, when
 * `summaries` is not NULL, summaries[0 .. n).  Tickets must be collected before their slot comes round again (every `depth` submits:
 * SLSLAM_ERR_STATE otherwise).  options: as for a batch; host_threads (0 = up to 16) pack and copy; refill_headroom_percent 0 = 10. */
typedef struct slslam_lba_stream slslam_lba_stream;
int  slslam_lba_stream_create(int device, const slslam_solver_options* opt, int depth, slslam_lba_stream** out);
void slslam_lba_stream_destroy(slslam_lba_stream* s);
int  slslam_lba_stream_submit(slslam_lba_stream* s, const slslam_lba_window* windows, int n, int* ticket);
int  slslam_lba_stream_collect(slslam_lba_stream* s, int ticket, slslam_summary* summaries);
/* submit() for a caller whose packer writes the indices NARROWED: packed_index[i] (when not NULL) replaces window i's camera_index /
 * line_index / fixed_index by one 32-bit word per observation (slslam_pack_indices: what the loop at reference src/slam.cpp:904-912 would
 * store instead of four ints) - 68 instead of 80 bytes per observation over the host link, which is what a stream of windows is bound
 * by.  The words are validated on the device (or by the staging copy); everything else as submit(). */
int  slslam_lba_stream_submit_packed(slslam_lba_stream* s, const slslam_lba_window* windows, const unsigned int* const* packed_index, int n, int* ticket);
/* Host-side accounting since create: wall-clock ms the caller spent in submit (pack + layout + enqueue), waiting in collect, copying
 * results out; submits served by a refill / by building a batch; windows submitted; trust-region steps (successful + unsuccessful, the
 * count of reference src/slam.cpp:949-950) of the windows collected; host threads in use.  Any pointer may be NULL. */
int  slslam_lba_stream_stats(const slslam_lba_stream* s, double* ms_submit, double* ms_collect_wait, double* ms_collect_copy,
                             long long* refills, long long* builds, long long* windows, long long* lm_iterations, int* host_threads);
/* Of the refills: how many were built on the device, how many of those read the callers' page-locked observations and parameters in place (no
 * staging copy; the index arrays are narrowed by the host threads on the way); windows the
 * device build handed back to the host path at collect time (a camera that sees a line twice, a line with more than 64 observations, more
 * than 20 free cameras, no room in the slot's arrays).  A set that does not fit the slot AS A WHOLE (only the device knows the tiles it needs) is
 * packed by the host threads at collect time, solved as one batch that then takes the slot - its windows count here and as one `builds`.
 * Any pointer may be NULL. */
int  slslam_lba_stream_build_stats(const slslam_lba_stream* s, long long* device_builds, long long* zero_copy, long long* fallback_windows);
/* The batch that served `ticket` (valid until its slot is submitted to again; after collect: its traces, chunk cuts and sweep can be read
 * with the slslam_lba_batch_* getters).  Read only. */
int  slslam_lba_stream_batch(slslam_lba_stream* s, int ticket, slslam_lba_batch** batch);

/* ------------------------------------------------------------------ pose graph
 * Replaces: POProblem(size, n_iter) + set_pose_index_1/2 + set_constraints + set_parameters
 * (reference src/po_problem.h:112-130) and the arrays built at src/slam.cpp:1262-1280. */
typedef struct slslam_po_graph {
  int num_poses;                  /* kfs.size()                                        */
  int num_edges;                  /* POProblem::num_size()                             */
  const int*    pose_index_1;     /* [E]                                               */
  const int*    pose_index_2;     /* [E]                                               */
  const double* constraints;      /* [6E]: (w,t) of C = T_{n2<-n1}                     */
  double*       parameters;       /* [6N] in/out: (w,t) per pose; pose_index_1[0] is held constant */
} slslam_po_graph;

/* Replaces: POProblem::build + POProblem::set_options + ceres::Solve
 * (reference src/slam.cpp:1283-1293).  Synchronous; parameters solved in place.
 * opt->huber_delta and opt->baseline are ignored (no loss function: src/po_problem.cpp:27,55). */
int slslam_po_solve(const slslam_po_graph* graph, const slslam_solver_options* opt,
                    slslam_summary* summary, slslam_iteration* trace, int trace_cap, int* trace_len);

/* The symbolic analysis behind the structured pose-graph factorisation (host only, no device needed), for
 * inspection and tests: slot[N] = offset of each pose in the reduced vector (-1: the constant pose of edge 0 or a
 * pose no edge references), chains first, junction poses last; chain_start / chain_len / chain_left / chain_right
 * [*num_chains, capacity max_chains] describe the chains (left / right = slot of what the chain ends at - a junction or, for a piece of a
 * long path, a cut pose -, -1 when free).  Returns SLSLAM_ERR_INVALID_ARGUMENT on malformed graphs, SLSLAM_ERR_UNSUPPORTED if max_chains is too small. */
int slslam_po_structure(const slslam_po_graph* graph, int* slot, int max_chains, int* num_chains, int* chain_start,
                        int* chain_len, int* chain_left, int* chain_right, int* num_chain_unknowns, int* num_unknowns);
/* The chains come on several levels (a long path is cut into pieces of ~L^(1/levels) poses by single cut poses; a path's cut poses form a
 * chain of the next level, eliminated after the pieces, and so on - up to three levels): how many of the chains the calling thread's last
 * slslam_po_structure listed - the first ones - are level-1 chains (consecutive poses are graph neighbours); the others are higher-level
 * chains (consecutive poses are cut poses of one path; left / right = a cut pose of a still higher level or the path's junction). */
int slslam_po_structure_level1(void);

/* ------------------------------------------------------------------ RANSAC hypothesis scoring
 * (the per-frame cost centre next to the hot path.)
 * Replaces: the scoring loop of SLAM::ransac_motion (reference src/slam.cpp:396-413) with
 * SLAM::reprojection_error (src/slam.cpp:691-726) as its body: for every motion hypothesis and every
 * common line, the mean absolute endpoint-to-line distance in both stereo images, an inlier when it is
 * below error_thr = 5 / focal_length (src/parameter.h:56); hypotheses with |t| > 1 are skipped
 * (src/slam.cpp:398-399) and score -1 here.  The reference's float/double mix (float `sql`, float
 * error accumulator) is reproduced exactly, so scores and inlier sets are bit-identical. */
typedef struct slslam_ransac_frame {
  int num_hypotheses;             /* motions to score                                           */
  int num_lines;                  /* comm_size: lines visible in both frames                    */
  const double* poses;            /* [12 H]: R row-major (9) then t (3) of each motion pose_t    */
  const double* observations;     /* [8 K]: obs1 of each common line (current frame)            */
  const double* lines;            /* [6 K]: (closest point, direction) of each line, world frame */
} slslam_ransac_frame;

/* scores[H]: inlier count per hypothesis (-1 when skipped);
 * inlier_bits[H * ((K + 63) / 64)]: bit k of word k / 64 set when line k is an inlier (may be NULL).
 * Synchronous; host pointers. */
int slslam_ransac_score(const slslam_ransac_frame* frame, double baseline, double error_thr,
                        int* scores, unsigned long long* inlier_bits);

/* ------------------------------------------------------------------ RANSAC hypothesis generation + trial loop
 * Replaces: SLAM::vo_angle_axis_approx (reference src/slam.cpp:433-574) — the linearised stereo-line motion from
 * s = max_feat_num = 5 sampled correspondences — for EVERY pre-drawn trial at once (lane <-> trial), and
 * SLAM::ransac_motion (src/slam.cpp:322-427): generate, score, then replay the reference's adaptive trial loop
 * (`trial_cnt < ransac_trial && trial_cnt <= max_trials`, ransac_trial re-estimated whenever the best score
 * improves, :363, :415-423) in trial order over the scores, so that best pose, best score, inlier set and
 * trial_cnt are what the sequential loop returns for the same sample sequence.  The caller draws the samples
 * (the reference: rand.rand_sample(sample, comm_size, s) once per trial, :366). */
typedef struct slslam_ransac_trials {
  int num_trials;                 /* pre-drawn trials (<= max_trials + 1 are ever looked at)            */
  int sample_size;                /* s = max_feat_num (src/parameter.h:25), 1..16                       */
  int num_lines;                  /* comm_size                                                          */
  const int*    samples;          /* [num_trials * s] indices into the common-line arrays               */
  const double* observations0;    /* [8 K] obs0: previous frame                                         */
  const double* observations1;    /* [8 K] obs1: current frame                                          */
} slslam_ransac_trials;

/* poses[12 H] (R row-major | t), valid[H] = num_sol (0 when a degenerate sample made the reference return 0).
 * `baseline` is passed through as given (the reference calls it with -baseline, src/slam.cpp:392). */
int slslam_ransac_generate(const slslam_ransac_trials* trials, double baseline, double* poses, int* valid);

/* Whole ransac_motion.  lines [6 K] as in slslam_ransac_frame.  *best_score_io: in = the caller's running best
 * (reference: 0), out = best score; *trial_cnt = trials executed; best_pose[12] and best_inlier_bits[(K+63)/64]
 * (may be NULL) are written only when some trial beat the incoming best score. */
int slslam_ransac_motion(const slslam_ransac_trials* trials, const double* lines, double baseline, double error_thr,
                         double prob_free_outliers, int max_trials, int* best_score_io, int* trial_cnt,
                         double* best_pose, unsigned long long* best_inlier_bits);

/* slslam_ransac_motion for many frames at once (sequence replay, several cameras): one upload, two launches per
 * frame enqueued back to back, one download.  frames[F], lines[F] (pointer per frame), best_score_io[F] in/out,
 * trial_cnt[F], best_pose[12 F], best_inlier_bits[F] (pointer per frame; the array or an entry may be NULL). */
int slslam_ransac_motion_batch(int num_frames, const slslam_ransac_trials* frames, const double* const* lines,
                               double baseline, double error_thr, double prob_free_outliers, int max_trials,
                               int* best_score_io, int* trial_cnt, double* best_pose,
                               unsigned long long* const* best_inlier_bits);

/* ------------------------------------------------------------------ diagnostics (benches, timing experiments)
 * Not part of the reference's surface: the reference times its back-end with StopWatch accumulators around whole calls
 * (src/stopwatch.h:42-157, proc_2 / proc_3 at src/slam.cpp:1384-1386, :1237,1312); these give the device-side split. */
/* Per-thread switch: slslam_po_solve brackets its device work and every factorisation (+ triangular solve) with HIP events. */
int slslam_po_set_profiling(int enable);
/* Timing of the calling thread's last profiled slslam_po_solve: device time of the whole solve, of its slowest factorisation
 * (iterations enqueued after the solve has converged are no-ops), how many factorisations were enqueued, unknowns of the reduced program and of the junction block (0 with the dense factorisations). */
int slslam_po_last_timing(double* total_ms, double* factor_ms, int* factor_calls, int* unknowns, int* junction_unknowns);
/* Timing experiments of the matrix-core elimination sweep / the reduced solve (environment SLSLAM_DEBUG_ABLATE, bits 8 / 9):
 * wave-clock cycles per phase summed over the batch since finalize; out[16].  SLSLAM_ERR_STATE unless the variable was set. */
int slslam_debug_phase_cycles(slslam_lba_batch* batch, double* out);
/* The same buffer raw (timing builds, -DSLSLAM_K1_TIMING=1: 32 words per chunk - the elimination sweep's phase cycles and, in words
 * 30 / 31, the constant-clock (s_memrealtime, 100 MHz) times at which the chunk's wave started and ended its last sweep: what
 * tools/chunk_timeline.py reads).  Copies min(n, size) words; *size = words the buffer holds. */
int slslam_debug_read_cycles(slslam_lba_batch* batch, unsigned long long* out, long long n, long long* size);

/* Test hook of the device build (csrc/lba_device_build.h): ONE window through k_ingest / k_build_window / k_build_layout / k_build_tiles
 * alone; everything the host packer emits for the window comes back (counts: Cf, tiles, items, free parameters, kept residual blocks;
 * tiles: line_begin, nlines, flags, nitems per tile) so that tests compare the two byte for byte.  *status: 0 built, else the reason
 * bits (1 bad input, 2 a shape for the host path). */
int slslam_debug_device_pack(const slslam_lba_window* window, int grouping, int* counts, int* line_order, int* line_ptr, int* ob_orig,
                             int* ob_cam, int* tiles, unsigned char* items, int* cam_cf, int max_tiles, int max_items,
                             unsigned short* lane_map, unsigned* line_desc, int* status);
/* ... with the shader-clock stamps of k_build_window's phases (tools/build_phases.py): phase_clocks[0 .. 9], or NULL. */
int slslam_debug_device_pack_timed(const slslam_lba_window* window, int grouping, int* counts, int* line_order, int* line_ptr, int* ob_orig,
                                   int* ob_cam, int* tiles, unsigned char* items, int* cam_cf, int max_tiles, int max_items,
                                   unsigned short* lane_map, unsigned* line_desc, int* status, unsigned long long* phase_clocks);

/* ------------------------------------------------------------------ misc */
int         slslam_device_count(void);          /* 0 when no HIP device is usable */
/* The one-shot entry points (slslam_lba_solve, slslam_po_solve) keep the device block of their last call per host thread
 * and reuse it when it is large enough (allocation costs as much as a small solve).  Frees the calling thread's block. */
void        slslam_release_cached_memory(void);
const char* slslam_version(void);
const char* slslam_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* SLSLAM_HIP_H_ */
