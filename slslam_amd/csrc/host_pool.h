// slslam_amd/csrc/host_pool.h — a small persistent pool of host threads for the per-window host work of a STREAM of windows
// (slslam_lba_batch_refill, slslam_lba_stream_*): packing a window (the LBAProblem::build stage, reference src/lba_problem.cpp:54-93)
// is independent of every other window, and at 1024 windows per batch one host thread packs for 20x longer than the GPU solves.
// run(n, fn): fn(i) for every i in [0, n), indices handed out dynamically, the calling thread takes part; returns when all are done -
// true unless some fn(i) threw (std::bad_alloc from a window's vectors): exceptions are caught where they are thrown, on whichever thread,
// counted, and never leave run() (they would cross the extern "C" boundary, or unwind run() while workers still use the caller's lambda).
#ifndef SLSLAM_HOST_POOL_H_
#define SLSLAM_HOST_POOL_H_

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace slslam {

class HostPool {
 public:
  explicit HostPool(int threads) {
    const int extra = threads > 1 ? threads - 1 : 0;           // the caller is one of the workers
    for (int t = 0; t < extra; ++t) workers_.emplace_back([this] { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> l(mu_); stop_ = true; ++epoch_; }
    cv_.notify_all();
    for (std::thread& t : workers_) t.join();
  }
  HostPool(const HostPool&) = delete;
  HostPool& operator=(const HostPool&) = delete;
  int threads() const { return (int)workers_.size() + 1; }

  bool run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return true;
    failed_.store(0);
    if (workers_.empty() || n == 1) {
      for (int i = 0; i < n; ++i) { try { fn(i); } catch (...) { failed_.fetch_add(1); } }
      return failed_.load() == 0;
    }
    {
      std::lock_guard<std::mutex> l(mu_);
      fn_ = &fn; n_ = n; next_.store(0); pending_ = (int)workers_.size(); ++epoch_;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> l(mu_);
    done_.wait(l, [this] { return pending_ == 0; });
    fn_ = nullptr;
    return failed_.load() == 0;
  }

 private:
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= n_) break;
      try { (*fn_)(i); } catch (...) { failed_.fetch_add(1); }
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_.wait(l, [&] { return epoch_ != seen; });
        seen = epoch_;
        if (stop_) return;
      }
      drain();
      std::lock_guard<std::mutex> l(mu_);
      if (--pending_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  std::atomic<int> next_{0}, failed_{0};
  int n_ = 0, pending_ = 0;
  unsigned long long epoch_ = 0;
  bool stop_ = false;
};

}  // namespace slslam
#endif
