// job_pool.h -- the host-side worker protocol of the multi-device index (vaqhip_multi.cpp): one
// persistent thread per shard; the caller hands every worker the same job (a callable taking the
// shard number), waits until ALL of them have reported, and reads their return codes.  A job is one
// PHASE: the caller decides between phases -- in particular a collective is only issued after every
// shard's part has succeeded, so a shard that fails can never leave its peers waiting inside an
// all-gather (an error return must not become a device hang).
// No HIP in here: tests/cpp/job_pool_test.cpp runs it under ThreadSanitizer / AddressSanitizer on the
// CPU build.
#ifndef VAQ_JOB_POOL_H_
#define VAQ_JOB_POOL_H_

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace vaq {

class JobPool {
public:
  JobPool() = default;
  JobPool(const JobPool &) = delete;
  JobPool &operator=(const JobPool &) = delete;
  ~JobPool() { stop(); }

  // start n workers (idempotent for the same n)
  void start(int n) {
    if (!threads_.empty()) return;
    rc_.assign((size_t)n, 0);
    for (int g = 0; g < n; g++) threads_.emplace_back([this, g] { loop(g); });
  }

  void stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_job_.notify_all();
    for (auto &t : threads_)
      if (t.joinable()) t.join();
    threads_.clear();
  }

  int size() const { return (int)threads_.size(); }

  // Run fn(g) on worker g for every g, wait for all of them; returns the first non-zero code (by
  // shard number) or 0.  rc(g) gives each worker's own code afterwards.  One run at a time (the
  // caller serialises; vaqhip_multi holds its mutex).
  int run(const std::function<int(int)> &fn) {
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      pending_ = (int)threads_.size();
      seq_++;
    }
    cv_job_.notify_all();
    {
      std::unique_lock<std::mutex> lk(mu_);
      cv_done_.wait(lk, [&] { return pending_ == 0; });
      fn_ = nullptr;
    }
    for (int r : rc_)
      if (r) return r;
    return 0;
  }

  int rc(int g) const { return rc_[(size_t)g]; }

private:
  void loop(int g) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(int)> *fn;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_job_.wait(lk, [&] { return quit_ || seq_ != seen; });
        if (quit_) return;
        seen = seq_;
        fn = fn_;
      }
      const int r = (*fn)(g);
      {
        std::lock_guard<std::mutex> lk(mu_);
        rc_[(size_t)g] = r;
        if (--pending_ == 0) cv_done_.notify_all();
      }
    }
  }

  std::mutex mu_;
  std::condition_variable cv_job_, cv_done_;
  std::vector<std::thread> threads_;
  std::vector<int> rc_;
  const std::function<int(int)> *fn_ = nullptr;
  uint64_t seq_ = 0;
  int pending_ = 0;
  bool quit_ = false;
};

} // namespace vaq
#endif
