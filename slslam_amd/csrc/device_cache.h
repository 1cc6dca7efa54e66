// slslam_amd/csrc/device_cache.h — one cached device block per host thread for the one-shot entry points
// (slslam_lba_solve, slslam_po_solve: the per-keyframe / per-loop-closure calls of the reference, src/slam.cpp:663,944,1293).
// A solve of one window takes ~1 ms of GPU time; hipMalloc + hipFree of its few MB take a comparable time, so the block
// of the previous call is kept and reused when it is large enough.  Batches the caller owns never go through here.
#ifndef SLSLAM_DEVICE_CACHE_H_
#define SLSLAM_DEVICE_CACHE_H_

#include <hip/hip_runtime.h>
#include <cstddef>

namespace slslam {

struct DeviceBlockCache {
  char* p = nullptr;
  size_t bytes = 0;
  int device = -1;
  bool in_use = false;
  enum : size_t { kMaxCached = (size_t)512 << 20 };     // larger blocks are not kept

  static DeviceBlockCache& mine() { static thread_local DeviceBlockCache c; return c; }   // never destroyed: no HIP call at thread exit

  // A block of at least `want` bytes on `device` (the current device).  hipSuccess or the allocation error.
  static hipError_t acquire(size_t want, int device, char** out) {
    DeviceBlockCache& c = mine();
    if (c.p && !c.in_use && c.device == device && c.bytes >= want) { c.in_use = true; *out = c.p; return hipSuccess; }
    return hipMalloc((void**)out, want);
  }
  // Hands a block back: it becomes (or stays) the cached one, or is freed.
  static void give_back(char* q, size_t have, int device) {
    if (!q) return;
    DeviceBlockCache& c = mine();
    if (q == c.p) { c.in_use = false; return; }
    if (!c.in_use && have <= kMaxCached && (c.p == nullptr || have > c.bytes || c.device != device)) {
      if (c.p) (void)hipFree(c.p);
      c.p = q; c.bytes = have; c.device = device;
      return;
    }
    (void)hipFree(q);
  }
  // Frees the calling thread's cached block (slslam_release_cached_memory).
  static void drop() {
    DeviceBlockCache& c = mine();
    if (c.p && !c.in_use) { (void)hipFree(c.p); c.p = nullptr; c.bytes = 0; c.device = -1; }
  }
};

}  // namespace slslam
#endif
