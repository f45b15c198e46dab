// coltt::Batcher (include/coltt_batcher.hpp) on a mock backend: many caller threads, one query each at a time — every caller
// gets exactly the rows of ITS query, batches never exceed max_batch, never mix different k, coalescing really happens, a
// lone query is released by the timer, backend errors reach every caller of the batch.  No GPU needed.
#include <atomic>
#include <cstdio>
#include <set>

#include "coltt_batcher.hpp"

static std::atomic<int> fails{0};
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main() {
  const uint32_t dim = 8;
  std::atomic<size_t> max_seen{0}, calls{0};
  std::atomic<bool> fail_next{false};
  std::mutex km; std::set<uint32_t> ks_in_call;
  // answer for query q: ids tag*1000 + j, scores tag + j/16, where tag = (uint32)q[0]; count = min(k, tag % 7 + 1)
  auto backend = [&](const float* q, size_t nq, uint32_t k, uint64_t* ids, float* sc, uint32_t* cnt) -> int {
    calls++;
    size_t m = max_seen.load(); while (nq > m && !max_seen.compare_exchange_weak(m, nq)) {}
    { std::lock_guard<std::mutex> g(km); ks_in_call.insert(k); }
    std::this_thread::sleep_for(std::chrono::microseconds(300));   // a "kernel": callers pile up meanwhile
    if (fail_next.exchange(false)) return -5;
    for (size_t i = 0; i < nq; i++) {
      const uint32_t tag = (uint32_t)q[i * dim];
      const uint32_t n = std::min<uint32_t>(k, tag % 7 + 1);
      cnt[i] = n;
      for (uint32_t j = 0; j < n; j++) { ids[i * k + j] = (uint64_t)tag * 1000 + j; sc[i * k + j] = (float)tag + (float)j / 16.f; }
    }
    return 0;
  };
  {
    coltt::Batcher b(dim, 16, std::chrono::microseconds(2000), backend);
    const int T = 48, M = 40;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back([&, t] {
      for (int m = 0; m < M; m++) {
        const uint32_t tag = (uint32_t)(t * 1000 + m);
        const uint32_t k = (t % 3 == 0) ? 3u : 5u;                  // two different k in flight at once
        float q[dim]; for (uint32_t e = 0; e < dim; e++) q[e] = (float)tag + (float)e;
        coltt::BatchAnswer a = b.Search(q, k);
        EXPECT(a.rc == 0);
        EXPECT(a.items.size() == std::min<uint32_t>(k, tag % 7 + 1));
        for (size_t j = 0; j < a.items.size(); j++) { EXPECT(a.items[j].Id == (uint64_t)tag * 1000 + j); EXPECT(a.items[j].Score == (float)tag + (float)j / 16.f); }
      }
    });
    for (auto& x : th) x.join();
    EXPECT(b.queries() == (uint64_t)T * M);
    EXPECT(max_seen.load() <= 16 && b.largest_batch() <= 16);
    EXPECT(b.batches() < (uint64_t)T * M / 2);                      // coalescing happened (48 callers against a 300 us backend)
    EXPECT(b.largest_batch() >= 4);
    std::printf("batches %llu for %llu queries, largest %zu\n", (unsigned long long)b.batches(), (unsigned long long)b.queries(), b.largest_batch());
    // a lone query is not held hostage: released by the timer
    float q[dim] = {42, 0, 0, 0, 0, 0, 0, 0};
    auto t0 = std::chrono::steady_clock::now();
    coltt::BatchAnswer a = b.Search(q, 5);
    auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    EXPECT(a.rc == 0 && a.items.size() == 1 && a.items[0].Id == 42000);
    EXPECT(us >= 2000 && us < 200000);
    // backend error reaches the caller
    fail_next = true;
    a = b.Search(q, 5);
    EXPECT(a.rc == -5 && a.items.empty());
    // k == 0 answers empty without calling the backend
    size_t c0 = calls.load();
    a = b.Search(q, 0);
    EXPECT(a.rc == 0 && a.items.empty() && calls.load() == c0);
  }
  std::printf(fails.load() ? "FAILED %d checks\n" : "batcher ok\n", fails.load());
  return fails.load() ? 1 : 0;
}
