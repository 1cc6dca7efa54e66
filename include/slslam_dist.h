/* include/slslam_dist.h — C ABI of the multi-GPU fan-out of the LBA path: one process per GPU, RCCL over xGMI.
 *
 * The reference has no multi-process code; what fans out is its per-window call
 * LBAProblem::build + ceres::Solve (src/slam.cpp:924-944) over INDEPENDENT windows, and what the ranks have to agree on afterwards is
 * what the caller accumulates per window at src/slam.cpp:949-952: the number of LM iterations and the initial / final costs.  So:
 * windows are split contiguously over the ranks (slslam_dist_shard_range - the rule slslam_amd/dist.py uses), every rank solves its
 * shard as one batch on its GPU with NO data-path collective, then ONE ncclAllReduce of the three sums and - when every rank needs
 * every result - ONE ncclAllGather of the solved parameter vectors.  libslslam_dist.so links librccl and libslslam_hip.so; a C++ host
 * (the mirror classes of slslam_amd/host) calls this instead of torch.distributed.  No CPU fallback; errors are slslam_hip.h status codes.
 */
#ifndef SLSLAM_DIST_H_
#define SLSLAM_DIST_H_

#include "slslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { SLSLAM_DIST_ID_BYTES = 128 };             /* sizeof(ncclUniqueId) */
typedef struct slslam_dist slslam_dist;

/* Rank 0: a fresh communicator id (ncclGetUniqueId), to be handed to the other ranks by whatever the launcher offers - a file, an
 * environment variable, MPI, a torch.distributed store. */
int  slslam_dist_unique_id(unsigned char id[SLSLAM_DIST_ID_BYTES]);
/* Every rank: joins the communicator on `device` (< 0: the current device).  Collective (ncclCommInitRank). */
int  slslam_dist_create(int rank, int world, int device, const unsigned char id[SLSLAM_DIST_ID_BYTES], slslam_dist** out);
void slslam_dist_destroy(slslam_dist* d);
int  slslam_dist_rank(const slslam_dist* d);
int  slslam_dist_world(const slslam_dist* d);
/* The shard of rank `rank` of `n` windows over `world` ranks: [*lo, *hi), contiguous, sizes differing by at most one. */
void slslam_dist_shard_range(long long n, int rank, int world, long long* lo, long long* hi);

/* Replaces, for this rank's `n_mine` windows (its shard of the job's list), LBAProblem::build + ceres::Solve per window (reference
 * src/slam.cpp:924-944) and the three running sums of :949-952 for the WHOLE job: parameters of every window of the shard are solved in
 * place; sums[0] = LM iterations (successful + unsuccessful), sums[1] = initial cost, sums[2] = final cost, over all ranks (one
 * ncclAllReduce on the solve stream, behind the solve).  gathered != NULL: also one ncclAllGather - gathered[r * slot + k] = the k-th
 * double of rank r's concatenated parameter vectors (windows in shard order, each [6C | 4L]), `slot` doubles per rank (the bound every
 * rank passes: >= the largest shard's parameter count, same on all ranks), count_per_rank[r] = doubles rank r filled.  HOST pointers.
 * Collective: every rank calls it with the same options, slot and gathered-ness.
 * Errors keep the ranks in step: a rank whose shard fails (bad window, allocation, ...) still enters the all-reduce - with zeros and an
 * error count as a fourth element - so nobody blocks; when that count is non-zero EVERY rank returns an error (its own status, or
 * SLSLAM_ERR_STATE on the healthy ranks: the sums would silently omit a shard) and NO rank enters the all-gather. */
int  slslam_dist_solve(slslam_dist* d, const slslam_lba_window* my_windows, int n_mine, const slslam_solver_options* opt,
                       double sums[3], double* gathered, long long slot, long long* count_per_rank);
/* Test hook: the next slslam_dist_solve on this rank solves its shard and then reports it as failed (exercises the path above). */
int  slslam_dist_debug_fail_next_shard(slslam_dist* d);

/* ---- the STREAMED form: a stream of sets of windows, every set sharded over the ranks (BASELINE config 4: "many independent 10-KF
 * windows sharded across 8 MI355X").  Every rank drives a slslam_lba_stream over ITS shard of every set - `depth` refillable batches in
 * flight, the LBAProblem::build stage on the device when the windows' arrays are page-locked (slslam_pinned_alloc), results written in
 * place - and the ranks meet once per set: collect() is the one ncclAllReduce of the three sums of reference src/slam.cpp:949-952 over
 * the whole set (sums[0] LM iterations, sums[1] initial cost, sums[2] final cost).  Replaces, per set and rank, the loop over
 * LBAProblem::build + ceres::Solve (src/slam.cpp:924-944) of the shard.  submit() is local (no collective); collect() is collective:
 * every rank calls it once per set, in the order of the submits, with the ticket its submit returned (-1 if its submit failed: the rank
 * still enters the all-reduce, and every rank returns an error for that set). */
typedef struct slslam_dist_stream slslam_dist_stream;
int  slslam_dist_stream_create(slslam_dist* d, const slslam_solver_options* opt, int depth, slslam_dist_stream** out);
void slslam_dist_stream_destroy(slslam_dist_stream* s);
int  slslam_dist_stream_submit(slslam_dist_stream* s, const slslam_lba_window* my_windows, int n_mine, int* ticket);
int  slslam_dist_stream_collect(slslam_dist_stream* s, int ticket, double sums[3]);

#ifdef __cplusplus
}
#endif
#endif /* SLSLAM_DIST_H_ */
