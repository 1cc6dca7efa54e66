// slslam_amd/dist_c/dist_api.cpp — include/slslam_dist.h: the LBA fan-out over the GPUs of a node, one process per GPU, RCCL over xGMI.
// Host C++ on the C ABI of libslslam_hip.so + two collectives; compiled with hipcc (host code only) against librccl.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/slslam_dist.h"

static_assert(sizeof(ncclUniqueId) == SLSLAM_DIST_ID_BYTES, "ncclUniqueId size");

#define DIST_HIP(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "slslam_dist: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? SLSLAM_ERR_NO_DEVICE : SLSLAM_ERR_HIP; \
    }                                                                                         \
  } while (0)
#define DIST_NCCL(expr)                                                                       \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != ncclSuccess) {                                                                  \
      std::fprintf(stderr, "slslam_dist: %s failed: %s (%s:%d)\n", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return SLSLAM_ERR_HIP;                                                                  \
    }                                                                                         \
  } while (0)

struct slslam_dist {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* d_sums = nullptr;          // [4]: LM steps | initial cost | final cost | ranks whose shard failed
  double* d_local = nullptr;         // this rank's parameters, `slot` doubles
  double* d_all = nullptr;           // world * slot
  long long* d_counts = nullptr;     // [world + 1]: all-gathered counts | this rank's count
  long long slot = 0;
  int fail_next_shard = 0;           // test hook: the next slslam_dist_solve reports its shard as failed (after solving it)
};

extern "C" int slslam_dist_unique_id(unsigned char id[SLSLAM_DIST_ID_BYTES]) {
  if (!id) return SLSLAM_ERR_INVALID_ARGUMENT;
  ncclUniqueId u;
  DIST_NCCL(ncclGetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return SLSLAM_OK;
}

extern "C" int slslam_dist_create(int rank, int world, int device, const unsigned char id[SLSLAM_DIST_ID_BYTES], slslam_dist** out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return SLSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return SLSLAM_ERR_NO_DEVICE; }
  if (device >= ndev) return SLSLAM_ERR_INVALID_ARGUMENT;
  DIST_HIP(hipSetDevice(device));
  slslam_dist* d = new (std::nothrow) slslam_dist();
  if (!d) return SLSLAM_ERR_HIP;
  d->rank = rank; d->world = world; d->device = device;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&d->comm, world, u, rank);
  if (r != ncclSuccess) { std::fprintf(stderr, "slslam_dist: ncclCommInitRank failed: %s\n", ncclGetErrorString(r)); delete d; return SLSLAM_ERR_HIP; }
  if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc((void**)&d->d_sums, 4 * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d->d_counts, (size_t)(world + 1) * sizeof(long long)) != hipSuccess) {
    slslam_dist_destroy(d);
    return SLSLAM_ERR_HIP;
  }
  *out = d;
  return SLSLAM_OK;
}

extern "C" void slslam_dist_destroy(slslam_dist* d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->stream) (void)hipStreamSynchronize(d->stream);
  if (d->comm) (void)ncclCommDestroy(d->comm);
  if (d->d_sums) (void)hipFree(d->d_sums);
  if (d->d_local) (void)hipFree(d->d_local);
  if (d->d_all) (void)hipFree(d->d_all);
  if (d->d_counts) (void)hipFree(d->d_counts);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
}

extern "C" int slslam_dist_rank(const slslam_dist* d) { return d ? d->rank : -1; }
extern "C" int slslam_dist_world(const slslam_dist* d) { return d ? d->world : 0; }

extern "C" void slslam_dist_shard_range(long long n, int rank, int world, long long* lo, long long* hi) {
  if (world < 1) world = 1;
  const long long base = n / world, rem = n % world;            // the first `rem` ranks hold one window more (slslam_amd/dist.py::shard_range)
  const long long a = rank * base + (rank < rem ? rank : rem);
  if (lo) *lo = a;
  if (hi) *hi = a + base + (rank < rem ? 1 : 0);
}

extern "C" int slslam_dist_solve(slslam_dist* d, const slslam_lba_window* w, int n, const slslam_solver_options* opt,
                                 double sums[3], double* gathered, long long slot, long long* count_per_rank) {
  if (!d || n < 0 || (n > 0 && !w) || !sums) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (gathered && slot <= 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  DIST_HIP(hipSetDevice(d->device));
  // ---- this rank's shard: one batch, no collective on the data path
  // From here to the last collective NOTHING returns: whatever fails on this rank is recorded in `rc`, the rank contributes zeros and an
  // error count to the all-reduce, and every rank learns from that count whether the all-gather is to be entered at all - a rank that
  // left early would leave the others blocked inside RCCL.
  double local[4] = { 0.0, 0.0, 0.0, 0.0 };          // LM steps | initial cost | final cost | ranks that failed
  long long my_count = 0;
  slslam_lba_batch* b = nullptr;
  // (every way out of this function first waits for the stream, whose asynchronous copies read and write this frame's variables and the
  // caller's arrays, and then gives the batch back)
  struct Guard {
    slslam_lba_batch*& b; hipStream_t s;
    ~Guard() { if (s) (void)hipStreamSynchronize(s); if (b) { slslam_lba_batch_destroy(b); b = nullptr; } }
  } guard{ b, d->stream };
  int rc = SLSLAM_OK;
  auto note = [&rc](hipError_t e, const char* what) {
    if (e == hipSuccess) return;
    std::fprintf(stderr, "slslam_dist: %s failed: %s\n", what, hipGetErrorString(e));
    if (rc == SLSLAM_OK) rc = SLSLAM_ERR_HIP;
  };
  if (n > 0) {
    rc = slslam_lba_batch_create(d->device, &b);
    for (int i = 0; i < n && rc == SLSLAM_OK; ++i) { rc = slslam_lba_batch_add(b, &w[i], nullptr); my_count += 6LL * w[i].num_cameras + 4LL * w[i].num_lines; }
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_finalize(b, opt);
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_solve(b, (void*)d->stream);
  }
  if (gathered) {
    if (my_count > slot && rc == SLSLAM_OK) rc = SLSLAM_ERR_INVALID_ARGUMENT;
    if (d->slot < slot) {
      if (d->d_local) (void)hipFree(d->d_local);
      if (d->d_all) (void)hipFree(d->d_all);
      d->d_local = d->d_all = nullptr; d->slot = 0;
      note(hipMalloc((void**)&d->d_local, (size_t)slot * sizeof(double)), "hipMalloc(gather slot)");
      if (d->d_local) note(hipMalloc((void**)&d->d_all, (size_t)slot * (size_t)d->world * sizeof(double)), "hipMalloc(gather buffer)");
      if (d->d_local && d->d_all) d->slot = slot;
    }
    if (d->slot >= slot) {
      note(hipMemsetAsync(d->d_local, 0, (size_t)slot * sizeof(double), d->stream), "hipMemsetAsync");
      if (b && rc == SLSLAM_OK) rc = slslam_lba_batch_export_device(b, d->d_local, (void*)d->stream);      // straight from the parameter buffers: no host round trip
    }
  }
  if (b && rc == SLSLAM_OK) {
    rc = slslam_lba_batch_download(b, (void*)d->stream);
    for (int i = 0; i < n && rc == SLSLAM_OK; ++i) {
      slslam_summary s;
      rc = slslam_lba_batch_get_parameters(b, i, w[i].parameters);
      if (rc == SLSLAM_OK) rc = slslam_lba_batch_get_summary(b, i, &s);
      if (rc == SLSLAM_OK) { local[0] += s.num_successful_steps + s.num_unsuccessful_steps; local[1] += s.initial_cost; local[2] += s.final_cost; }
    }
  }
  if (d->fail_next_shard) { d->fail_next_shard = 0; if (rc == SLSLAM_OK) rc = SLSLAM_ERR_HIP; }      // (test hook: slslam_dist_debug_fail_next_shard)
  if (rc != SLSLAM_OK) { local[0] = local[1] = local[2] = 0.0; local[3] = 1.0; }
  // ---- the ONE all-reduce of the run summary (reference src/slam.cpp:949-952) + the count of ranks whose shard failed ...
  note(hipMemcpyAsync(d->d_sums, local, sizeof(local), hipMemcpyHostToDevice, d->stream), "hipMemcpyAsync(sums)");
  double total[4] = { 0.0, 0.0, 0.0, 0.0 };
  {
    const ncclResult_t r = ncclAllReduce(d->d_sums, d->d_sums, 4, ncclDouble, ncclSum, d->comm, d->stream);
    if (r != ncclSuccess) { std::fprintf(stderr, "slslam_dist: ncclAllReduce failed: %s\n", ncclGetErrorString(r)); return SLSLAM_ERR_HIP; }   // (the communicator itself is gone: nothing left to keep in step)
  }
  note(hipMemcpyAsync(total, d->d_sums, sizeof(total), hipMemcpyDeviceToHost, d->stream), "hipMemcpyAsync(sums back)");
  note(hipStreamSynchronize(d->stream), "hipStreamSynchronize");
  sums[0] = total[0]; sums[1] = total[1]; sums[2] = total[2];
  // a failed shard anywhere: EVERY rank returns an error (the sums omit that shard) and NO rank enters the all-gather
  const bool any_failed = total[3] > 0.5;
  if (any_failed) return rc != SLSLAM_OK ? rc : SLSLAM_ERR_STATE;
  // ---- ... and, when asked for, the ONE all-gather of the results (+ the counts, 8 bytes per rank)
  if (gathered) {
    const long long mine = my_count;
    DIST_HIP(hipMemcpyAsync(d->d_counts + d->world, &mine, sizeof(mine), hipMemcpyHostToDevice, d->stream));
    DIST_NCCL(ncclAllGather(d->d_counts + d->world, d->d_counts, 1, ncclInt64, d->comm, d->stream));
    DIST_NCCL(ncclAllGather(d->d_local, d->d_all, (size_t)slot, ncclDouble, d->comm, d->stream));
    DIST_HIP(hipMemcpyAsync(gathered, d->d_all, (size_t)slot * (size_t)d->world * sizeof(double), hipMemcpyDeviceToHost, d->stream));
    if (count_per_rank) DIST_HIP(hipMemcpyAsync(count_per_rank, d->d_counts, (size_t)d->world * sizeof(long long), hipMemcpyDeviceToHost, d->stream));
  }
  DIST_HIP(hipStreamSynchronize(d->stream));
  return rc;
}

extern "C" int slslam_dist_debug_fail_next_shard(slslam_dist* d) {
  if (!d) return SLSLAM_ERR_INVALID_ARGUMENT;
  d->fail_next_shard = 1;
  return SLSLAM_OK;
}

// ------------------------------------------------------------------------------------------
// The STREAMED form of the fan-out (BASELINE config 4 as it reads: a stream of windows sharded over the GPUs of a node): every rank owns a
// stream object (slslam_lba_stream_*: refillable batches in flight, the build stage on the device when the caller's arrays are page-locked)
// over its shard of every set; what the ranks exchange per set is the ONE all-reduce of the three sums the reference's caller accumulates
// (src/slam.cpp:949-952) plus the count of ranks whose shard failed.
struct slslam_dist_stream {
  slslam_dist* d = nullptr;
  slslam_lba_stream* st = nullptr;
  int depth = 0;
  std::vector<int> n_of_ticket;            // windows of the shard behind each ticket in flight (by slot)
  std::vector<slslam_summary> summaries;
};

extern "C" int slslam_dist_stream_create(slslam_dist* d, const slslam_solver_options* opt, int depth, slslam_dist_stream** out) {
  if (!d || !out) return SLSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  slslam_dist_stream* s = new (std::nothrow) slslam_dist_stream();
  if (!s) return SLSLAM_ERR_NO_MEMORY;
  s->d = d; s->depth = depth;
  const int rc = slslam_lba_stream_create(d->device, opt, depth, &s->st);
  if (rc != SLSLAM_OK) { delete s; return rc; }
  s->n_of_ticket.assign((size_t)depth, 0);
  *out = s;
  return SLSLAM_OK;
}

extern "C" void slslam_dist_stream_destroy(slslam_dist_stream* s) {
  if (!s) return;
  if (s->st) slslam_lba_stream_destroy(s->st);
  delete s;
}

extern "C" int slslam_dist_stream_submit(slslam_dist_stream* s, const slslam_lba_window* my_windows, int n_mine, int* ticket) {
  if (!s || !ticket) return SLSLAM_ERR_INVALID_ARGUMENT;
  // no collective here: a rank whose submit fails reports it at collect time (ticket -1 is accepted there), so that the ranks stay in step
  int t = -1;
  const int rc = slslam_lba_stream_submit(s->st, my_windows, n_mine, &t);
  *ticket = rc == SLSLAM_OK ? t : -1;
  if (rc == SLSLAM_OK) s->n_of_ticket[(size_t)(t % s->depth)] = n_mine;
  return rc;
}

extern "C" int slslam_dist_stream_collect(slslam_dist_stream* s, int ticket, double sums[3]) {
  if (!s || !sums) return SLSLAM_ERR_INVALID_ARGUMENT;
  slslam_dist* d = s->d;
  DIST_HIP(hipSetDevice(d->device));
  double local[4] = { 0.0, 0.0, 0.0, 0.0 };
  int rc = ticket < 0 ? SLSLAM_ERR_STATE : SLSLAM_OK;      // (this rank's submit had failed: it still enters the all-reduce)
  if (rc == SLSLAM_OK) {
    const int n = s->n_of_ticket[(size_t)(ticket % s->depth)];
    s->summaries.resize((size_t)std::max(1, n));
    rc = slslam_lba_stream_collect(s->st, ticket, s->summaries.data());
    for (int i = 0; i < n && rc == SLSLAM_OK; ++i) {
      local[0] += s->summaries[(size_t)i].num_successful_steps + s->summaries[(size_t)i].num_unsuccessful_steps;
      local[1] += s->summaries[(size_t)i].initial_cost; local[2] += s->summaries[(size_t)i].final_cost;
    }
  }
  if (d->fail_next_shard) { d->fail_next_shard = 0; if (rc == SLSLAM_OK) rc = SLSLAM_ERR_HIP; }
  if (rc != SLSLAM_OK) { local[0] = local[1] = local[2] = 0.0; local[3] = 1.0; }
  hipError_t e = hipMemcpyAsync(d->d_sums, local, sizeof(local), hipMemcpyHostToDevice, d->stream);
  if (e != hipSuccess && rc == SLSLAM_OK) rc = SLSLAM_ERR_HIP;
  double total[4] = { 0.0, 0.0, 0.0, 0.0 };
  DIST_NCCL(ncclAllReduce(d->d_sums, d->d_sums, 4, ncclDouble, ncclSum, d->comm, d->stream));
  DIST_HIP(hipMemcpyAsync(total, d->d_sums, sizeof(total), hipMemcpyDeviceToHost, d->stream));
  DIST_HIP(hipStreamSynchronize(d->stream));
  sums[0] = total[0]; sums[1] = total[1]; sums[2] = total[2];
  if (total[3] > 0.5) return rc != SLSLAM_OK ? rc : SLSLAM_ERR_STATE;      // some rank's shard failed: every rank says so
  return SLSLAM_OK;
}
