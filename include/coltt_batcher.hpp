// coltt_batcher.hpp — the RPC micro-batcher of SURVEY.md §8f.4 as compiled code (header-only C++17).
//
// The reference serves ONE query per RPC (core/core.go:633-695, edge/edge.go:610-690), each on its own goroutine; a GPU wants
// batches.  Callers on any number of threads call Search(query, k) and block; a collector thread flushes when max_batch
// queries are waiting or max_wait elapsed since the first of them, issues ONE batched search and hands every caller its own
// rows.  Queries are grouped by k: HNSW uses ef = max(cfg.ef, k), so answers for different k are not prefixes of one another
// in general.  Same semantics as go/colttgpu/batcher.go (the Go source the maintainer compiles); this one is exercised by
// tests/cpp/batcher_test.cpp on a mock backend (CPU) and over coltt::Hnsw on the GPU.
//
// Backend = any callable  int(const float* queries, size_t nq, uint32_t k, uint64_t* ids, float* scores, uint32_t* counts)
// returning COLTT_OK or an error code, e.g. a lambda around coltt_hnsw_search / coltt_flat_search.
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace coltt {

struct BatchItem { uint64_t Id; float Score; };
struct BatchAnswer { int rc = 0; std::vector<BatchItem> items; };

class Batcher {
 public:
  using Backend = std::function<int(const float*, size_t, uint32_t, uint64_t*, float*, uint32_t*)>;

  Batcher(uint32_t dim, size_t max_batch, std::chrono::microseconds max_wait, Backend backend)
      : dim_(dim), max_batch_(max_batch ? max_batch : 1), max_wait_(max_wait), backend_(std::move(backend)),
        worker_([this] { loop(); }) {}
  ~Batcher() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    worker_.join();
  }
  Batcher(const Batcher&) = delete;
  Batcher& operator=(const Batcher&) = delete;

  // blocks until the batch this query rode in has been answered; the query is copied before returning to the collector
  BatchAnswer Search(const float* query, uint32_t k) {
    auto p = std::make_shared<Pending>();
    p->q.assign(query, query + dim_);
    p->k = k;
    std::future<BatchAnswer> f = p->done.get_future();
    {
      std::lock_guard<std::mutex> g(mu_);
      queue_.push_back(p);
    }
    cv_.notify_all();
    return f.get();
  }

  // statistics (for tests / tuning)
  uint64_t batches() const { std::lock_guard<std::mutex> g(mu_); return n_batches_; }
  uint64_t queries() const { std::lock_guard<std::mutex> g(mu_); return n_queries_; }
  size_t largest_batch() const { std::lock_guard<std::mutex> g(mu_); return largest_; }

 private:
  struct Pending { std::vector<float> q; uint32_t k = 0; std::promise<BatchAnswer> done; };

  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [this] { return stop_ || !queue_.empty(); });
      if (queue_.empty()) { if (stop_) return; continue; }
      // the first waiting query opens a batch for its k; wait for company until the batch is full or max_wait elapsed
      const uint32_t k = queue_.front()->k;
      const auto deadline = std::chrono::steady_clock::now() + max_wait_;
      while (!stop_ && count_k(k) < max_batch_) {
        if (cv_.wait_until(lk, deadline) == std::cv_status::timeout) break;
      }
      std::vector<std::shared_ptr<Pending>> batch;
      for (auto it = queue_.begin(); it != queue_.end() && batch.size() < max_batch_;) {
        if ((*it)->k == k) { batch.push_back(*it); it = queue_.erase(it); } else ++it;
      }
      n_batches_++; n_queries_ += batch.size(); if (batch.size() > largest_) largest_ = batch.size();
      lk.unlock();
      flush(batch, k);
      lk.lock();
    }
  }
  size_t count_k(uint32_t k) const { size_t c = 0; for (auto& p : queue_) c += p->k == k; return c; }

  void flush(std::vector<std::shared_ptr<Pending>>& batch, uint32_t k) {
    const size_t nq = batch.size();
    std::vector<float> flat(nq * dim_);
    for (size_t i = 0; i < nq; i++) std::memcpy(flat.data() + i * dim_, batch[i]->q.data(), dim_ * sizeof(float));
    std::vector<uint64_t> ids(nq * (size_t)(k ? k : 1));
    std::vector<float> sc(nq * (size_t)(k ? k : 1));
    std::vector<uint32_t> cnt(nq, 0);
    const int rc = k ? backend_(flat.data(), nq, k, ids.data(), sc.data(), cnt.data()) : 0;
    for (size_t i = 0; i < nq; i++) {
      BatchAnswer a; a.rc = rc;
      if (rc == 0) {
        const uint32_t n = cnt[i] < k ? cnt[i] : k;
        a.items.resize(n);
        for (uint32_t j = 0; j < n; j++) a.items[j] = {ids[i * k + j], sc[i * k + j]};
      }
      batch[i]->done.set_value(std::move(a));
    }
  }

  const uint32_t dim_; const size_t max_batch_; const std::chrono::microseconds max_wait_; Backend backend_;
  mutable std::mutex mu_; std::condition_variable cv_; std::deque<std::shared_ptr<Pending>> queue_;
  bool stop_ = false; uint64_t n_batches_ = 0, n_queries_ = 0; size_t largest_ = 0;
  std::thread worker_;  // last member: started after everything else is initialised
};

}  // namespace coltt
