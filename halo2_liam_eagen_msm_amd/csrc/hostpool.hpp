// A small persistent worker pool for the host-side tail of an MSM call: converting the few hundred raw device
// records and folding each window's (total, U_0..U_{L-1}) into its window sum are independent per window
// (~10 us each), only the final Horner over the windows is a serial chain.  The workers sleep on a condition
// variable between calls; the calling thread takes jobs too, so a pool of size 0 degenerates to a plain loop.
#pragma once
#include <sched.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lemsm {
namespace host {

class Pool {
 public:
  explicit Pool(int workers) {
    for (int i = 0; i < workers; i++) threads_.emplace_back([this] { worker(); });
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  int workers() const { return (int)threads_.size(); }

  // runs fn(0..njobs-1), each index exactly once, and returns when all are done
  void run(int njobs, const std::function<void(int)>& fn) {
    if (njobs <= 0) return;
    if (threads_.empty() || njobs == 1) { for (int i = 0; i < njobs; i++) fn(i); return; }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn; njobs_ = njobs; next_.store(0, std::memory_order_relaxed); done_.store(0, std::memory_order_relaxed);
      gen_++;
    }
    cv_.notify_all();
    work(&fn, njobs);
    while (done_.load(std::memory_order_acquire) < njobs) std::this_thread::yield();
    { std::lock_guard<std::mutex> lk(mu_); fn_ = nullptr; }       // late wakers find nothing to do
    // a worker that picked this job up holds &fn until it leaves work(): it can claim no index any more
    // (next_ >= njobs until the next run() resets it, which happens after this wait)
    while (active_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

  // host threads this process may use (affinity mask / cgroup share of the box), at least 1
  static int usable_cpus() {
    cpu_set_t set; CPU_ZERO(&set);
    int n = 0;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return n > 0 ? n : 1;
  }

 private:
  void work(const std::function<void(int)>* fn, int nj) {
    for (;;) {
      int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= nj) break;
      (*fn)(i);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  void worker() {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn; int nj;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        if (!fn_) continue;
        fn = fn_; nj = njobs_;
        active_.fetch_add(1, std::memory_order_acq_rel);
      }
      work(fn, nj);
      active_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  unsigned long gen_ = 0;
  bool stop_ = false;
  const std::function<void(int)>* fn_ = nullptr;
  int njobs_ = 0;
  std::atomic<int> next_{0}, done_{0}, active_{0};
};

}  // namespace host
}  // namespace lemsm
