// C++ host-side mirror (include/coltt_gpu.hpp) exercised the way the reference's own Go tests exercise *vectorindex.Hnsw and
// the edge vector store — the Go toolchain is absent, so this is the compiled-language consumer of the C-ABI:
//   * TestHnswCommitLoad-style round trip (core/vectorindex/hnsw_commit_test.go:32-102, 127-181): insert, remove ~20 %,
//     Commit, Load into a fresh index, Len and every search result equal;
//   * ItemAlreadyExistsError / ItemNotFoundError (hnsw.go:38-41), empty index => empty result (hnsw.go:249-251);
//   * edge VertexSearch keeps the K farthest ascending (edge/priority_queue.go:39-69), dimension-mismatch error text
//     (none_vectorstore.go:86-88), SaveVertex -> LoadVertex round trip.
// Prints one line per check; the pytest wrapper (tests/test_gpu_cpp_mirror.py) also compares the printed search results with
// the Python binding's on the same data.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>

#include "coltt_gpu.hpp"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

static std::vector<float> vec(std::mt19937& g, int d) {
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> v(d);
  for (auto& x : v) x = nd(g);
  return v;
}

int main() {
  if (coltt_init(0) != COLTT_OK) { std::printf("no device: %s\n", coltt_last_error()); return 77; }
  const int d = 32, n = 500;
  std::mt19937 g(12345);
  std::uniform_real_distribution<float> U(1e-6f, 0.999999f);
  // ---- *vectorindex.Hnsw
  coltt::Hnsw a(d, COLTT_COSINE);
  EXPECT(a.Search(vec(g, d), 5).empty());                      // empty index => empty result, not an error
  std::vector<std::vector<float>> X;
  for (int i = 0; i < n; i++) { X.push_back(vec(g, d)); a.Insert(1000 + i, X.back(), a.RandomLevel(U(g))); }
  EXPECT(a.Len() == n);
  try { a.Insert(1000, X[0], 0); EXPECT(false); } catch (const coltt::ItemAlreadyExistsError&) {}
  try { a.Remove(999999); EXPECT(false); } catch (const coltt::ItemNotFoundError&) {}
  int removed = 0;
  for (int i = 0; i < n; i += 5) { a.Remove(1000 + i); removed++; }
  EXPECT(a.Len() == n - removed);
  std::vector<uint8_t> blob = a.Commit(true);
  {
    coltt::Hnsw wrong(d, COLTT_EUCLIDEAN);                     // a stream written with another distance is refused, not misread
    try { wrong.Load(blob, true); EXPECT(false); } catch (const coltt::Error& e) { EXPECT(e.code == COLTT_E_INVALID); }
  }
  coltt::Hnsw b(d, COLTT_COSINE);
  EXPECT(b.Load(blob, true) == (uint64_t)(n - removed));
  EXPECT(b.Len() == a.Len());
  EXPECT(b.Commit(true).size() == blob.size());
  coltt::Hnsw c(d, COLTT_COSINE);                             // Commit(Load(.)) is a fixed point: c == b slot for slot
  c.Reserve((uint64_t)n);                                     // sized up front: same index, one allocation
  EXPECT(c.Load(b.Commit(true), true) == (uint64_t)(n - removed));
  std::mt19937 gq(777);
  for (int q = 0; q < 20; q++) {
    auto qv = vec(gq, d);
    auto ra = a.Search(qv, 10), rb = b.Search(qv, 10), rc = c.Search(qv, 10);
    EXPECT(ra.size() == 10 && rb.size() == 10 && rc.size() == 10);
    int common = 0;
    for (size_t i = 0; i < rb.size() && i < rc.size(); i++) {
      EXPECT(rb[i].Id == rc[i].Id);
      EXPECT(std::memcmp(&rb[i].Score, &rc[i].Score, 4) == 0);
      EXPECT((rb[i].Id - 1000) % 5 != 0);                      // removed vertices never come back
      if (i) EXPECT(rb[i - 1].Score <= rb[i].Score);           // ascending by distance (hnsw.go:261-277)
      for (auto& x : ra) common += x.Id == rb[i].Id;
    }
    // a (insertion slot order, tombstones) and b (stream slot order) walk their neighbours in different canonical orders —
    // as two runs of the reference do with Go's random map order — so they agree on the neighbourhood, not on every bit
    EXPECT(common >= 7);
    if (q < 3) { std::printf("hnsw q%d:", q); for (auto& r : rb) std::printf(" %llu/%08x", (unsigned long long)r.Id, *(const unsigned*)&r.Score); std::printf("\n"); }
  }
  // ---- the rest of the *vectorindex.Hnsw surface core uses (core/core.go:232-236, 439, 512, 607): Config, Distance, Get,
  // GetVertex, BytesSize, functional options
  {
    coltt::Hnsw o(d, COLTT_EUCLIDEAN, coltt::HnswOptions().M(8).Ef(33).EfConstruction(50).SearchAlgorithm(1));
    coltt::ProtoConfig pc = o.Config();
    EXPECT(pc.SearchAlgorithm == "heuristic" && pc.M == 8 && pc.MMax == 8 && pc.MMax0 == 16 && pc.Ef == 33 && pc.EfConstruction == 50);
    EXPECT(std::fabs(pc.LevelMultiplier - 1.0f / std::log(8.0f)) < 1e-6f && pc.HeuristicKeepPruned && !pc.HeuristicExtendCandidates);
    EXPECT(o.Distance() == "l2-squared" && a.Distance() == "cosine-dot");
    try { coltt::Hnsw bad(d, COLTT_COSINE, coltt::HnswOptions().SearchAlgorithm(1).HeuristicExtendCandidates(true)); EXPECT(false); }
    catch (const coltt::Error& e) { EXPECT(e.code == COLTT_E_UNSUPPORTED); }   // undefined behaviour in the reference: rejected
    for (int i = 0; i < 40; i++) o.Insert(7 + i, X[i], i % 3);
    coltt::Vertex v = o.GetVertex(7 + 5);
    EXPECT(v.Id == 12 && v.Level == 2 && v.Vec.size() == (size_t)d && std::memcmp(v.Vec.data(), X[5].data(), d * 4) == 0);  // l2: stored as given
    EXPECT(o.GetVertex(7).Level == 0);                                          // the first vertex is forced to level 0
    try { o.Get(99999); EXPECT(false); } catch (const coltt::ItemNotFoundError&) {}
    o.Remove(12);
    try { o.GetVertex(12); EXPECT(false); } catch (const coltt::ItemNotFoundError&) {}
    EXPECT(o.BytesSize() > (uint64_t)o.Len() * d * 4);
    // cosine: Get returns the NORMALISED vector (Insert normalises, hnsw.go:105-107)
    coltt::Vector g0 = a.Get(1001); double nn = 0; for (float f : g0) nn += (double)f * f;
    EXPECT(std::fabs(nn - 1.0) < 1e-5);
  }
  // ---- edge.vectorspace
  coltt::VecSpace s(d, COLTT_COSINE, COLTT_Q_F16);
  for (int i = 0; i < n; i++) s.ChangedVertex(50 + i, X[i]);
  EXPECT(s.LoadSize() == n);
  try { s.ChangedVertex(1, std::vector<float>(d + 1)); EXPECT(false); }
  catch (const coltt::Error& e) { EXPECT(std::strstr(e.what(), "expect dimension: [32], but got [33]") != nullptr); }
  auto far = s.VertexSearch(X[3], 10);                         // the reference's queue keeps the K FARTHEST, ascending
  auto near = s.VertexSearch(X[3], 10, false, COLTT_SELECT_NEAREST);
  EXPECT(far.size() == 10 && near.size() == 10);
  EXPECT(near[0].Id == 53);
  for (size_t i = 1; i < far.size(); i++) EXPECT(far[i - 1].Score <= far[i].Score);
  EXPECT(near.back().Score <= far.front().Score);
  coltt::VecSpace t(d, COLTT_COSINE, COLTT_Q_F16);
  EXPECT(t.LoadVertex(s.SaveVertex()) == (uint64_t)n);
  auto near2 = t.VertexSearch(X[3], 10, false, COLTT_SELECT_NEAREST);
  for (size_t i = 0; i < near.size(); i++) EXPECT(near[i].Id == near2[i].Id && std::memcmp(&near[i].Score, &near2[i].Score, 4) == 0);
  auto filt = s.FilterableVertexSearch({51, 53, 60, 61}, X[3], 3, COLTT_SELECT_NEAREST);
  EXPECT(filt.size() == 3 && filt[0].Id == 53);
  // the remaining vectorspace methods (edge/vectorstore.go:36-48): metadata / inverted blobs / accessors
  {
    coltt::CollectionMetadata m; m.Dim = d; m.Distance = COLTT_COSINE; m.Quantization = COLTT_Q_F16; m.Versioning = true;
    m.IndexType["user_id"] = "primary,string";
    coltt::VecSpace u(m);
    EXPECT(u.Dim() == (uint32_t)d && u.Distance() == COLTT_COSINE && u.Quantization() == COLTT_Q_F16 && u.Versional() && u.Indexer().count("user_id") == 1);
    EXPECT(u.SaveVertexMetadata().find("\"quantization\":1") != std::string::npos);
    u.LoadVertexInverted({1, 2, 3}); EXPECT(u.SaveVertexInverted().size() == 3);
    coltt::CollectionMetadata w = m; w.Dim = d + 1;
    try { u.LoadVertexMetadata("c", w); EXPECT(false); } catch (const coltt::Error&) {}
    u.LoadVertexMetadata("c", m);
    s.RemoveVertex({53});
    EXPECT(s.LoadSize() == n - 1 && s.VertexSearch(X[3], 1, false, COLTT_SELECT_NEAREST)[0].Id != 53);
  }
  // ---- a collection group: three FLAT members on this one GPU answer like the single store (same stored bits, same order)
  {
    coltt::Group grp({0, 0, 0}, d, COLTT_COSINE, COLTT_Q_F16, COLTT_GROUP_FLAT);
    std::vector<uint64_t> gids; std::vector<float> flat;
    for (int i = 0; i < n; i++) { gids.push_back(50 + i); flat.insert(flat.end(), X[i].begin(), X[i].end()); }
    EXPECT(grp.ChangedVertex(gids, flat) == (uint64_t)n && grp.Len() == (uint64_t)n);
    EXPECT(grp.ShardOf(53) >= 0 && grp.ShardOf(53) < 3);
    coltt::VecSpace ref(d, COLTT_COSINE, COLTT_Q_F16);
    for (int i = 0; i < n; i++) ref.ChangedVertex(50 + i, X[i]);
    for (int sel : {COLTT_SELECT_NEAREST, COLTT_SELECT_REFERENCE}) {
      auto a = grp.Search(X[5], 10, sel);
      auto b = ref.VertexSearch(X[5], 10, false, sel);
      EXPECT(a.size() == b.size());
      for (size_t i = 0; i < a.size() && i < b.size(); i++) EXPECT(a[i].Id == b[i].Id && std::memcmp(&a[i].Score, &b[i].Score, 4) == 0);
    }
    grp.Remove({55});
    EXPECT(grp.Len() == (uint64_t)n - 1 && grp.Search(X[5], 1)[0].Id != 55);
  }
  // ---- the product quantiser (coltt_pq_*): parameters of models.ProductQuantizerParameters, call shape of playground/hnswpq_verification.go
  {
    coltt::ProductQuantizerParameters pp; pp.NumSubVectors = 8; pp.NumCentroids = 16;
    coltt::ProductQuantizer pq(d, COLTT_PQ_EUCLIDEAN, pp);
    try { pq.Search(X[0], 3); EXPECT(false); } catch (const coltt::Error& e) { EXPECT(e.code == COLTT_E_INVALID); }   // no codebooks yet
    std::vector<float> sample;
    for (int i = 0; i < 200; i++) sample.insert(sample.end(), X[i].begin(), X[i].end());
    pq.Fit(sample, 0);                                       // iteration 0 of the definition: centroid c of sub-space j = sub-vector j of sample vector c
    auto cb = pq.Codebooks();
    const int ds = d / 8;
    bool seeds = true;
    for (int j = 0; j < 8; j++) for (int c = 0; c < 16; c++) for (int e = 0; e < ds; e++) seeds = seeds && cb[((size_t)j * 16 + c) * ds + e] == X[c][j * ds + e];
    EXPECT(seeds);
    pq.Fit(sample, 3);
    auto c0 = pq.Encode(X[0]);
    EXPECT(c0.size() == 8 && c0 == pq.Encode(X[0]));
    for (auto code : c0) EXPECT(code < 16);
    for (int i = 0; i < n; i++) pq.Insert(7000 + i, X[i]);
    EXPECT(pq.Len() == (uint64_t)n);
    try { pq.SetCodebooks(cb); EXPECT(false); } catch (const coltt::Error& e) { EXPECT(e.code == COLTT_E_INVALID); }   // the stored codes belong to the codebooks
    try { pq.Insert(1, std::vector<float>(d + 1)); EXPECT(false); } catch (const coltt::Error&) {}
    auto all = pq.Search(X[3], (unsigned)n);
    EXPECT(all.size() == (size_t)n);
    std::vector<char> seen(n, 0);
    for (size_t i = 0; i < all.size(); i++) {
      EXPECT(all[i].Id >= 7000 && all[i].Id < 7000 + (uint64_t)n && !seen[all[i].Id - 7000]); seen[all[i].Id - 7000] = 1;
      if (i) EXPECT(all[i - 1].Score < all[i].Score || (all[i - 1].Score == all[i].Score && all[i - 1].Id < all[i].Id));   // ascending by (score, id)
    }
    auto top = pq.Search(X[3], 5);
    for (size_t i = 0; i < top.size(); i++) EXPECT(top[i].Id == all[i].Id && std::memcmp(&top[i].Score, &all[i].Score, 4) == 0);
    pq.Remove({top[0].Id});
    EXPECT(pq.Len() == (uint64_t)n - 1 && pq.Search(X[3], 1)[0].Id == all[1].Id);
  }
  std::printf(fails ? "FAILED %d checks\n" : "mirror ok\n", fails);
  return fails ? 1 : 0;
}
