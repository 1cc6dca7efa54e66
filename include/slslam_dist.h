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

#ifdef __cplusplus
}
#endif
#endif /* SLSLAM_DIST_H_ */
