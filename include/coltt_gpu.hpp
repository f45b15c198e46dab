// coltt_gpu.hpp — header-only C++ mirror of the reference's Go interfaces over the C-ABI (include/coltt_gpu.h).
//
// The reference is compiled code (Go) whose toolchain is absent from the build image, so the host side above the C-ABI is
// mirrored here in C++ with the reference's own names, argument meaning and error behaviour:
//   coltt::VecSpace  <->  edge.vectorspace            (edge/vectorstore.go:30-49)
//   coltt::Hnsw      <->  *vectorindex.Hnsw           (core/vectorindex/hnsw.go:43-54)
// Metadata maps stay with the caller: results carry ids and scores only.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "coltt_gpu.h"

namespace coltt {

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* m) : std::runtime_error(m), code(c) {}
};
struct ItemNotFoundError : Error { using Error::Error; };       // core/vectorindex/hnsw.go:39
struct ItemAlreadyExistsError : Error { using Error::Error; };  // core/vectorindex/hnsw.go:40

inline void check(int rc) {
  if (rc == COLTT_OK) return;
  if (rc == COLTT_E_NOT_FOUND) throw ItemNotFoundError(rc, coltt_last_error());
  if (rc == COLTT_E_EXISTS) throw ItemAlreadyExistsError(rc, coltt_last_error());
  throw Error(rc, coltt_last_error());
}

using Vector = std::vector<float>;                                 // edge.Vector (edge/constants.go:76)
struct SearchResultItem { uint64_t Id; float Score; };             // edge/priority_queue.go:27-31 minus Metadata
using SearchResult = std::vector<SearchResultItem>;

// edge.vectorspace implemented on the GPU
class VecSpace {
 public:
  VecSpace(uint32_t dim, int distance /*COLTT_COSINE|EUCLIDEAN*/, int quantization /*COLTT_Q_**/)
      : dim_(dim), distance_(distance), quant_(quantization) { check(coltt_flat_create(dim, distance, quantization, &h_)); }
  ~VecSpace() { if (h_) coltt_flat_destroy(h_); }
  VecSpace(const VecSpace&) = delete;
  VecSpace& operator=(const VecSpace&) = delete;

  // ChangedVertex(updateID, Id, ENode) — vector half (edge/none_vectorstore.go:66-103)
  void ChangedVertex(uint64_t id, const Vector& v) {
    if (v.size() != dim_)  // none_vectorstore.go:86-88
      throw Error(COLTT_E_INVALID, ("Dim Length UnmatchdError: expect dimension: [" + std::to_string(dim_) + "], but got [" +
                                    std::to_string(v.size()) + "]").c_str());
    check(coltt_flat_upsert(h_, &id, v.data(), 1));
  }
  void ChangedVertices(const std::vector<uint64_t>& ids, const float* rows) { check(coltt_flat_upsert(h_, ids.data(), rows, ids.size())); }
  // RemoveVertex after the inverted index resolved the drop filter to ids (none_vectorstore.go:105-127)
  void RemoveVertex(const std::vector<uint64_t>& ids) { check(coltt_flat_remove(h_, ids.data(), ids.size())); }
  // VertexSearch(target, topK, highCpu) (none_vectorstore.go:129-180); highCpu has no meaning on the GPU
  SearchResult VertexSearch(const Vector& target, int topK, bool /*highCpu*/ = false, int select = COLTT_SELECT_REFERENCE) const {
    std::vector<uint64_t> ids(topK); std::vector<float> sc(topK); uint32_t n = 0;
    check(coltt_flat_search(h_, target.data(), 1, (uint32_t)topK, select, COLTT_MODE_EXACT, ids.data(), sc.data(), &n));
    SearchResult r(n);
    for (uint32_t i = 0; i < n; i++) r[i] = {ids[i], sc[i]};
    return r;
  }
  // FilterableVertexSearch(filter, target, topK, highCpu) with the filter already evaluated to candidate ids (:182-253)
  SearchResult FilterableVertexSearch(const std::vector<uint64_t>& candidates, const Vector& target, int topK,
                                      int select = COLTT_SELECT_REFERENCE) const {
    std::vector<uint64_t> ids(topK); std::vector<float> sc(topK); uint32_t n = 0;
    check(coltt_flat_search_ids(h_, target.data(), 1, (uint32_t)topK, select, candidates.data(), candidates.size(), ids.data(), sc.data(), &n));
    SearchResult r(n);
    for (uint32_t i = 0; i < n; i++) r[i] = {ids[i], sc[i]};
    return r;
  }
  // SaveVertex / LoadVertex (none_vectorstore.go:308-516 and twins): the `.vertex` byte stream, metadata written as empty maps
  std::vector<uint8_t> SaveVertex() const {
    uint64_t n = 0;
    check(coltt_flat_save_vertex(h_, nullptr, nullptr, nullptr, 0, nullptr, 0, &n));
    std::vector<uint8_t> out(n);
    check(coltt_flat_save_vertex(h_, nullptr, nullptr, nullptr, 0, out.data(), out.size(), &n));
    return out;
  }
  uint64_t LoadVertex(const std::vector<uint8_t>& data) {
    uint64_t n = 0;
    check(coltt_flat_load_vertex(h_, data.data(), data.size(), &n, nullptr, nullptr, nullptr, 0));
    return n;
  }
  int Quantization() const { return quant_; }
  int Distance() const { return distance_; }
  uint32_t Dim() const { return dim_; }
  int64_t LoadSize() const { uint64_t n = 0; check(coltt_flat_len(h_, &n)); return (int64_t)n; }
  coltt_handle_t handle() const { return h_; }

 private:
  coltt_handle_t h_ = 0; uint32_t dim_; int distance_, quant_;
};

// *vectorindex.Hnsw implemented on the GPU
class Hnsw {
 public:
  // NewHnsw(dim, distancer, options...) (hnsw.go:56-73)
  Hnsw(uint32_t dim, int distance, const coltt_hnsw_cfg* cfg = nullptr, int quantization = COLTT_Q_NONE) : dim_(dim) {
    check(coltt_hnsw_create(dim, distance, quantization, cfg, &h_));
  }
  ~Hnsw() { if (h_) coltt_hnsw_destroy(h_); }
  Hnsw(const Hnsw&) = delete;
  Hnsw& operator=(const Hnsw&) = delete;

  void Insert(uint64_t id, const Vector& value, int vertexLevel) { check(coltt_hnsw_insert(h_, id, value.data(), vertexLevel)); }  // hnsw.go:104
  void Remove(uint64_t id) { check(coltt_hnsw_remove(h_, id)); }                                                                    // hnsw.go:191
  SearchResult Search(const Vector& query, unsigned k) const {                                                                       // hnsw.go:243
    std::vector<uint64_t> ids(k); std::vector<float> sc(k); uint32_t n = 0;
    check(coltt_hnsw_search(h_, query.data(), 1, k, 0, ids.data(), sc.data(), &n, nullptr));
    SearchResult r(n);
    for (uint32_t i = 0; i < n; i++) r[i] = {ids[i], sc[i]};
    return r;
  }
  int Len() const { uint64_t n = 0; check(coltt_hnsw_len(h_, &n)); return (int)n; }
  // Commit(w, header) / Load(r, header) (hnsw_commit.go:69-278): the reference's big-endian stream (metadata: empty maps)
  std::vector<uint8_t> Commit(bool header = true) const {
    uint64_t n = 0;
    check(coltt_hnsw_commit(h_, header, nullptr, nullptr, nullptr, 0, &n));
    std::vector<uint8_t> out(n);
    check(coltt_hnsw_commit(h_, header, nullptr, nullptr, out.data(), out.size(), &n));
    return out;
  }
  uint64_t Load(const std::vector<uint8_t>& data, bool header = true) {
    uint64_t n = 0;
    check(coltt_hnsw_load(h_, header, data.data(), data.size(), &n, nullptr, nullptr, nullptr, 0));
    return n;
  }
  // RandomLevel() (hnsw.go:280-282) for the caller's uniform draw u in (0,1)
  int RandomLevel(float u) const { int32_t lv = 0; check(coltt_hnsw_random_level(h_, u, &lv)); return lv; }
  uint32_t Dim() const { return dim_; }
  coltt_hnsw_cfg Config() const { coltt_hnsw_cfg c; check(coltt_hnsw_get_cfg(h_, &c)); return c; }
  coltt_handle_t handle() const { return h_; }

 private:
  coltt_handle_t h_ = 0; uint32_t dim_;
};

}  // namespace coltt
