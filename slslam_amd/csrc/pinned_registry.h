// slslam_amd/csrc/pinned_registry.h — which host address ranges are page-locked and mapped for the device (include/slslam_hip.h:
// slslam_pinned_alloc / slslam_pinned_register).  A window whose five arrays (reference src/slam.cpp:899-921) lie in such ranges is
// read by the GPU where it is (lba_device_build.h::k_ingest, zero copy) and its results are written back the same way; anything else
// goes through a pinned staging copy made by the host threads.  The lookup is a binary search over a handful of ranges: a refill asks
// five times per window.
#ifndef SLSLAM_PINNED_REGISTRY_H_
#define SLSLAM_PINNED_REGISTRY_H_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <mutex>
#include <vector>

namespace slslam {

class PinnedRegistry {
 public:
  static PinnedRegistry& get() { static PinnedRegistry* r = new PinnedRegistry(); return *r; }     // never destroyed: no HIP call at process exit
  struct Range { uintptr_t lo, hi; bool owned; };

  hipError_t alloc(size_t bytes, void** out) {
    *out = nullptr;
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    add((uintptr_t)p, bytes ? bytes : 1, true);
    *out = p;
    return hipSuccess;
  }
  // 0: freed; 1: not one of ours
  int free(void* p) {
    if (!remove((uintptr_t)p, true)) return 1;
    (void)hipHostFree(p);
    return 0;
  }
  hipError_t register_range(void* p, size_t bytes) {
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return e;
    add((uintptr_t)p, bytes, false);
    return hipSuccess;
  }
  int unregister_range(void* p) {
    if (!remove((uintptr_t)p, false)) return 1;
    (void)hipHostUnregister(p);
    return 0;
  }
  // [p, p + bytes) lies inside one registered range
  bool contains(const void* p, size_t bytes) {
    if (!p) return false;
    const uintptr_t a = (uintptr_t)p;
    std::lock_guard<std::mutex> l(mu_);
    auto it = std::upper_bound(ranges_.begin(), ranges_.end(), a, [](uintptr_t v, const Range& r) { return v < r.lo; });
    if (it == ranges_.begin()) return false;
    --it;
    return a >= it->lo && a + bytes <= it->hi;
  }
  // a snapshot for many lookups without the lock (a refill: 5 arrays x 1024 windows)
  std::vector<Range> snapshot() { std::lock_guard<std::mutex> l(mu_); return ranges_; }
  static bool contains(const std::vector<Range>& rs, const void* p, size_t bytes) {
    if (!p) return false;
    const uintptr_t a = (uintptr_t)p;
    auto it = std::upper_bound(rs.begin(), rs.end(), a, [](uintptr_t v, const Range& r) { return v < r.lo; });
    if (it == rs.begin()) return false;
    --it;
    return a >= it->lo && a + bytes <= it->hi;
  }

 private:
  void add(uintptr_t lo, size_t bytes, bool owned) {
    std::lock_guard<std::mutex> l(mu_);
    Range r{ lo, lo + bytes, owned };
    ranges_.insert(std::upper_bound(ranges_.begin(), ranges_.end(), r, [](const Range& x, const Range& y) { return x.lo < y.lo; }), r);
  }
  bool remove(uintptr_t lo, bool owned) {
    std::lock_guard<std::mutex> l(mu_);
    for (size_t i = 0; i < ranges_.size(); ++i)
      if (ranges_[i].lo == lo && ranges_[i].owned == owned) { ranges_.erase(ranges_.begin() + (long)i); return true; }
    return false;
  }
  std::mutex mu_;
  std::vector<Range> ranges_;
};

}  // namespace slslam
#endif
