// gamma.txt / theta.txt writer of the host front end (host/terastructure_main.cpp), in a header of its own so that
// tests/test_host_format_cpu.py can drive it against the fprintf("%.8f\t") loop it replaces.
#pragma once
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fast_format.h"

// ---- where the wall clock goes (timing.txt in the run directory; extension) ----------
struct Stopwatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double lap() {
    const auto t1 = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(t1 - t0).count();
    t0 = t1;
    return s;
  }
};
// ---- gamma.txt / theta.txt off the critical path ---------------------------------------
// save_gamma (src/snpsamplinge.cc:546-576) prints 2 N K doubles with fprintf("%.8f\t") while everything else waits: at
// N = 1M, K = 8 that is as long as the training of a whole report period.  Here the main thread only snapshots gamma / theta
// (two device-to-host copies) and hands them to ONE writer thread, which formats them -- fmt_fixed8: the bytes of "%.8f" --
// while the next period's schedules run on the GPU.  Jobs are written in order, one at a time (a later save of the same
// file name must land after the earlier one); at most one snapshot waits (128 MB each at N = 1M, K = 8).  drain() returns
// when every file handed over is complete and closed: before the process exits, on every path (stop rule, SIGTERM,
// -max-iter, an error).
struct SaveJob {
  std::string gamma_path, theta_path;
  std::vector<double> g, t;
  size_t n = 0, k = 0;
};
class ModelWriter {
 public:
  ~ModelWriter() { stop(); }
  // false: an earlier job failed (error() says why)
  bool submit(std::unique_ptr<SaveJob> job, double *waited_s) {
    Stopwatch sw;
    std::unique_lock<std::mutex> lk(mu_);
    if (!started_) {
      th_ = std::thread([this] { loop(); });
      started_ = true;
    }
    cv_done_.wait(lk, [this] { return !pending_; });  // (one snapshot in the queue at most)
    if (waited_s) *waited_s += sw.lap();
    if (!error_.empty()) return false;
    pending_ = std::move(job);
    cv_work_.notify_one();
    return true;
  }
  // every file handed over is complete and closed
  bool drain(double *waited_s = nullptr) {
    Stopwatch sw;
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this] { return !pending_ && !busy_; });
    if (waited_s) *waited_s += sw.lap();
    return error_.empty();
  }
  void stop() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      if (!started_) return;
      cv_done_.wait(lk, [this] { return !pending_ && !busy_; });
      quit_ = true;
      cv_work_.notify_one();
    }
    th_.join();
    started_ = false;
  }
  std::string error() {
    std::lock_guard<std::mutex> lk(mu_);
    return error_;
  }
  double busy_seconds() {
    std::lock_guard<std::mutex> lk(mu_);
    return busy_s_;
  }
  // the two files of one snapshot: per individual K x "%.8f\t" then "\n" (trailing tab, no id column)
  static bool write_files(const SaveJob &j, std::string *err) {
    FILE *f = fopen(j.gamma_path.c_str(), "w");
    FILE *h = fopen(j.theta_path.c_str(), "w");
    if (!f || !h) {
      *err = std::string("cannot open gamma/theta file:") + strerror(errno);
      if (f) fclose(f);
      if (h) fclose(h);
      return false;
    }
    bool ok = write_matrix(f, j.g.data(), j.n, j.k) && write_matrix(h, j.t.data(), j.n, j.k);
    ok = (fclose(f) == 0) && ok;
    ok = (fclose(h) == 0) && ok;
    if (!ok) *err = std::string("error writing gamma/theta file:") + strerror(errno);
    return ok;
  }
  static bool write_matrix(FILE *f, const double *v, size_t n, size_t k) {
    const size_t rows_per_block = std::max<size_t>(1, (size_t)(1u << 20) / (k * 24 + 1));
    std::vector<char> buf(rows_per_block * (k * (tsfmt::kMaxLen + 1) + 1));
    for (size_t i0 = 0; i0 < n; i0 += rows_per_block) {
      const size_t i1 = std::min(n, i0 + rows_per_block);
      char *p = buf.data();
      for (size_t i = i0; i < i1; ++i) {
        for (size_t j = 0; j < k; ++j) {
          p = tsfmt::fmt_fixed8(p, v[i * k + j]);
          *p++ = '\t';
        }
        *p++ = '\n';
      }
      if (fwrite(buf.data(), 1, (size_t)(p - buf.data()), f) != (size_t)(p - buf.data())) return false;
    }
    return true;
  }

 private:
  void loop() {
    for (;;) {
      std::unique_ptr<SaveJob> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_work_.wait(lk, [this] { return pending_ || quit_; });
        if (!pending_) return;
        job = std::move(pending_);
        busy_ = true;
        cv_done_.notify_all();  // (the queue slot is free again)
      }
      Stopwatch sw;
      std::string err;
      const bool ok = write_files(*job, &err);
      job.reset();
      {
        std::lock_guard<std::mutex> lk(mu_);
        busy_s_ += sw.lap();
        if (!ok && error_.empty()) error_ = err;
        busy_ = false;
        cv_done_.notify_all();
      }
    }
  }
  std::mutex mu_;
  std::condition_variable cv_work_, cv_done_;
  std::unique_ptr<SaveJob> pending_;
  std::thread th_;
  std::string error_;
  double busy_s_ = 0;
  bool started_ = false, busy_ = false, quit_ = false;
};

