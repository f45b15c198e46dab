// coltt_gpu.hpp — header-only C++ mirror of the reference's Go interfaces over the C-ABI (include/coltt_gpu.h).
//
// The reference is compiled code (Go) whose toolchain is absent from the build image, so the host side above the C-ABI is
// mirrored here in C++ with the reference's own names, argument meaning and error behaviour:
//   coltt::VecSpace  <->  edge.vectorspace            (edge/vectorstore.go:30-49)
//   coltt::Hnsw      <->  *vectorindex.Hnsw           (core/vectorindex/hnsw.go:43-54)
// Metadata maps stay with the caller: results carry ids and scores only.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
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

// edge.Metadata (edge/edge_metadata.go:25-55): what CreateCollection fixes for a collection.  IndexType carries the caller's
// index features by name (edge.IndexFeature is the inverted index's business — pkg/inverted is out of scope).
struct CollectionMetadata {
  uint32_t Dim = 0; int Distance = COLTT_COSINE; int Quantization = COLTT_Q_NONE; bool Versioning = false;
  std::map<std::string, std::string> IndexType;
};

// edge.vectorspace implemented on the GPU — all 16 methods of edge/vectorstore.go:30-49.  Metadata maps and the inverted
// index stay with the caller: ChangedVertex / RemoveVertex / FilterableVertexSearch take the ids the
// caller's index resolved, Save/LoadVertexInverted carry the caller's serialized index as an opaque blob, results carry
// ids and scores.  (The Go file go/edge/gpu_vectorstore.go is the variant that owns those parts inside package edge.)
class VecSpace {
 public:
  VecSpace(uint32_t dim, int distance /*COLTT_COSINE|EUCLIDEAN*/, int quantization /*COLTT_Q_**/)
      : dim_(dim), distance_(distance), quant_(quantization) {
    meta_.Dim = dim; meta_.Distance = distance; meta_.Quantization = quantization;
    check(coltt_flat_create(dim, distance, quantization, &h_));
  }
  explicit VecSpace(const CollectionMetadata& m) : VecSpace(m.Dim, m.Distance, m.Quantization) { meta_ = m; }
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
  // SaveVertexMetadata / LoadVertexMetadata (none_vectorstore.go:255-274): the collection metadata as JSON; loading re-checks
  // it against the device store (dim / distance / quantisation are fixed at creation).
  std::string SaveVertexMetadata() const {
    std::string j = "{\"dim\":" + std::to_string(meta_.Dim) + ",\"distance\":" + std::to_string(meta_.Distance) + ",\"quantization\":" +
                    std::to_string(meta_.Quantization) + ",\"versioning\":" + (meta_.Versioning ? "true" : "false") + ",\"index_type\":{";
    bool first = true;
    for (auto& kv : meta_.IndexType) { j += (first ? "\"" : ",\"") + kv.first + "\":\"" + kv.second + "\""; first = false; }
    return j + "}}";
  }
  void LoadVertexMetadata(const std::string& /*collectionName*/, const CollectionMetadata& m) {
    if (m.Dim != dim_ || m.Distance != distance_ || m.Quantization != quant_)
      throw Error(COLTT_E_INVALID, "LoadVertexMetadata: dim / distance / quantization differ from the device store's");
    meta_ = m;
  }
  // SaveVertexInverted / LoadVertexInverted (:276-283): the caller's inverted index, carried opaquely
  const std::vector<uint8_t>& SaveVertexInverted() const { return inverted_blob_; }
  void LoadVertexInverted(const std::vector<uint8_t>& data) { inverted_blob_ = data; }
  int Quantization() const { return quant_; }
  int Distance() const { return distance_; }
  uint32_t Dim() const { return dim_; }
  int64_t LoadSize() const { uint64_t n = 0; check(coltt_flat_len(h_, &n)); return (int64_t)n; }
  const std::map<std::string, std::string>& Indexer() const { return meta_.IndexType; }
  bool Versional() const { return meta_.Versioning; }
  coltt_handle_t handle() const { return h_; }

 private:
  coltt_handle_t h_ = 0; uint32_t dim_; int distance_, quant_;
  CollectionMetadata meta_; std::vector<uint8_t> inverted_blob_;
};

// functional options of NewHnsw (hnsw_config.go:57-109) as a fluent builder: HnswOptions().M(32).Ef(64).SearchAlgorithm(1)
struct HnswOptions {
  coltt_hnsw_cfg c{16, -1, -1, 20, 200, 0, -1.f, 0, 1};   // newHnswConfig defaults (hnsw_config.go:135-162)
  int quantization = COLTT_Q_NONE;
  HnswOptions& LevelMultiplier(float v) { c.level_multiplier = v; return *this; }
  HnswOptions& Ef(int v) { c.ef = v; return *this; }
  HnswOptions& EfConstruction(int v) { c.ef_construction = v; return *this; }
  HnswOptions& M(int v) { c.m = v; return *this; }
  HnswOptions& Mmax(int v) { c.m_max = v; return *this; }
  HnswOptions& Mmax0(int v) { c.m_max0 = v; return *this; }
  HnswOptions& SearchAlgorithm(int v /*0 HnswSearchSimple, 1 HnswSearchHeuristic; COLTT_HNSW_DIVERSE (2) is NOT reference behaviour, see coltt_gpu.h*/) { c.algo = v; return *this; }
  HnswOptions& HeuristicExtendCandidates(bool v) { c.extend_candidates = v; return *this; }
  HnswOptions& HeuristicKeepPruned(bool v) { c.keep_pruned = v; return *this; }
  HnswOptions& Quantization(int q) { quantization = q; return *this; }   // extension: BASELINE.json configs[4]
};
struct ProtoConfig {   // hnsw_config.go:123-133
  std::string SearchAlgorithm; float LevelMultiplier; int Ef, EfConstruction, M, MMax, MMax0; bool HeuristicExtendCandidates, HeuristicKeepPruned;
};
struct Vertex { uint64_t Id; Vector Vec; int Level; };   // what GetVertex hands out (hnsw_vertex.go:54-68), minus Metadata

// *vectorindex.Hnsw implemented on the GPU
class Hnsw {
 public:
  Hnsw(uint32_t dim, int distance, const HnswOptions& o) : Hnsw(dim, distance, &o.c, o.quantization) {}
  // NewHnsw(dim, distancer, options...) (hnsw.go:56-73)
  Hnsw(uint32_t dim, int distance, const coltt_hnsw_cfg* cfg = nullptr, int quantization = COLTT_Q_NONE) : dim_(dim), distance_(distance) {
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
  // the collection's size is known (bulk import, Load): every array allocated once; never shrinks, never limits an Insert
  void Reserve(uint64_t vertices, uint64_t upper_rows = 0) { check(coltt_hnsw_reserve(h_, vertices, upper_rows)); }
  // Commit(w, header) / Load(r, header) (hnsw_commit.go:69-278): the reference's big-endian stream (metadata: empty maps)
  std::vector<uint8_t> Commit(bool header = true) const {
    uint64_t n = 0;
    check(coltt_hnsw_commit(h_, header, nullptr, nullptr, 0, nullptr, 0, &n));
    std::vector<uint8_t> out(n);
    check(coltt_hnsw_commit(h_, header, nullptr, nullptr, 0, out.data(), out.size(), &n));
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
  coltt_hnsw_cfg RawConfig() const { coltt_hnsw_cfg c; check(coltt_hnsw_get_cfg(h_, &c)); return c; }
  ProtoConfig Config() const {   // hnsw.go:86-98
    coltt_hnsw_cfg c = RawConfig();
    return {c.algo == 0 ? "simple" : (c.algo == 1 ? "heuristic" : "diverse"), c.level_multiplier, c.ef, c.ef_construction, c.m, c.m_max, c.m_max0,
            c.extend_candidates != 0, c.keep_pruned != 0};
  }
  std::string Distance() const { return distance_ == COLTT_COSINE ? "cosine-dot" : "l2-squared"; }   // Space.Type(), space.go:69,101
  // Get(id) / GetVertex(id) (hnsw.go:169-189): the STORED (normalised) vector; ItemNotFoundError for unknown / removed ids
  Vector Get(uint64_t id) const { Vector v(dim_); check(coltt_hnsw_get(h_, id, v.data(), nullptr)); return v; }
  Vertex GetVertex(uint64_t id) const { Vertex x{id, Vector(dim_), 0}; int32_t lv = 0; check(coltt_hnsw_get(h_, id, x.Vec.data(), &lv)); x.Level = lv; return x; }
  // BytesSize() (hnsw.go:476-490): the reference's host-side estimate — pointers per vertex by level + vector bytes
  // (HNSW_VERTEX_EDGE_BYTES 12, HNSW_VERTEX_MUTEX_BYTES 24, hnsw_vertex.go:27-28); metadata bytes are the caller's to add
  uint64_t BytesSize() const {
    coltt_hnsw_cfg c = RawConfig();
    uint64_t ns = 0, nu = 0; int32_t ent = -1, el = 0;
    check(coltt_hnsw_export_raw(h_, &ns, &nu, &ent, &el, nullptr, nullptr, nullptr));
    const int maxLevel = ent >= 0 ? el : 10;
    double ptr = c.m_max0 * 12.0 + 24.0;
    for (int i = 1; i < maxLevel; i++) ptr += (c.m_max * 12.0 + 24.0) * std::exp((double)i / -(double)c.level_multiplier);
    return (uint64_t)std::floor((double)Len() * ptr) + (uint64_t)Len() * dim_ * 4;
  }
  coltt_handle_t handle() const { return h_; }

 private:
  coltt_handle_t h_ = 0; uint32_t dim_; int distance_ = COLTT_COSINE;
};

// One collection over the GPUs of a node (BASELINE.json configs[4]): ShardVertex routing (pkg/sharding/shard.go:34-41), one FLAT
// store or HNSW graph per device, per-member search + one RCCL all-gather (or, members sharing a device, host staging) + the
// local-queue-then-global-queue merge of edge/none_vectorstore.go:148-178.  Thin wrapper over coltt_group_*.
class Group {
 public:
  Group(const std::vector<int>& devices, uint32_t dim, int distance, int quantization, int kind /*COLTT_GROUP_FLAT | _HNSW*/,
        int layout = COLTT_LAYOUT_SHARD, const coltt_hnsw_cfg* cfg = nullptr, int exchange = COLTT_EXCHANGE_AUTO) : dim_(dim), kind_(kind) {
    coltt_group_opts o{}; o.kind = kind; o.layout = layout; o.exchange = exchange;
    check(coltt_group_create(devices.data(), (int)devices.size(), dim, distance, quantization, cfg, &o, &h_));
  }
  ~Group() { if (h_) coltt_group_destroy(h_); }
  Group(const Group&) = delete;
  Group& operator=(const Group&) = delete;
  int ShardOf(uint64_t id) const { int32_t s = 0; check(coltt_group_shard_of(h_, id, &s)); return s; }   // sharding.ShardVertex(id, world)
  coltt_handle_t Member(int i) const { coltt_handle_t m = 0; check(coltt_group_member(h_, i, &m)); return m; }
  uint64_t Len() const { uint64_t n = 0; check(coltt_group_len(h_, &n)); return n; }
  // ChangedVertex (FLAT groups) / Insert (HNSW groups) for n vertices, row-major vectors; returns how many this process hosts
  uint64_t ChangedVertex(const std::vector<uint64_t>& ids, const std::vector<float>& vecs) {
    if (vecs.size() != ids.size() * dim_) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    uint64_t kept = 0; check(coltt_group_upsert(h_, ids.data(), vecs.data(), ids.size(), &kept)); return kept;
  }
  uint64_t Insert(const std::vector<uint64_t>& ids, const std::vector<float>& vecs, const std::vector<int32_t>& levels, uint32_t batch = 1) {
    if (vecs.size() != ids.size() * dim_ || levels.size() != ids.size()) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    uint64_t kept = 0; check(coltt_group_insert(h_, ids.data(), vecs.data(), levels.data(), ids.size(), batch, &kept)); return kept;
  }
  void Remove(const std::vector<uint64_t>& ids) { check(coltt_group_remove(h_, ids.data(), ids.size())); }
  // one query over the whole collection (rows ascending by (score, id)); select / mode: FLAT groups, ef: HNSW groups
  SearchResult Search(const Vector& query, unsigned k, int select = COLTT_SELECT_NEAREST, int mode = COLTT_MODE_EXACT, uint32_t ef = 0) const {
    if (query.size() != dim_) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    std::vector<uint64_t> ids(k); std::vector<float> sc(k); uint32_t n = 0;
    check(coltt_group_search(h_, query.data(), 1, k, select, mode, ef, ids.data(), sc.data(), &n));
    SearchResult r(n);
    for (uint32_t i = 0; i < n; i++) r[i] = {ids[i], sc[i]};
    return r;
  }
  coltt_handle_t handle() const { return h_; }

 private:
  coltt_handle_t h_ = 0; uint32_t dim_; int kind_;
};


// Product quantiser (coltt_pq_*).  Parameters = models.ProductQuantizerParameters (pkg/models/hnsw_common.go:20-33);
// method names follow the call shape of playground/hnswpq_verification.go:90-105,154,190-199 (the package it drives, pkg/hnswpq, is not in
// the reference's tree).  distance = COLTT_PQ_COSINE / _EUCLIDEAN / _DOT: which pkg/distancepq function fills the query's table.
struct ProductQuantizerParameters { int NumCentroids = 256; int NumSubVectors = 8; int TriggerThreshold = 10000; };
class ProductQuantizer {
 public:
  ProductQuantizer(uint32_t dim, int distance, const ProductQuantizerParameters& p) : dim_(dim), params_(p) {
    check(coltt_pq_create(dim, distance, (uint32_t)p.NumSubVectors, (uint32_t)p.NumCentroids, &h_));
  }
  ~ProductQuantizer() { if (h_) coltt_pq_destroy(h_); }
  ProductQuantizer(const ProductQuantizer&) = delete;
  ProductQuantizer& operator=(const ProductQuantizer&) = delete;
  // training on a sample of n = sample.size() / dim vectors (n >= NumCentroids), deterministic Lloyd iterations
  void Fit(const std::vector<float>& sample, unsigned iterations = 8) {
    if (sample.size() % dim_) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    check(coltt_pq_train(h_, sample.data(), sample.size() / dim_, iterations));
  }
  void SetCodebooks(const std::vector<float>& cb) {
    if (cb.size() != (size_t)params_.NumCentroids * dim_) throw Error(COLTT_E_INVALID, "codebooks: [NumSubVectors][NumCentroids][dim / NumSubVectors] floats expected");
    check(coltt_pq_set_codebooks(h_, cb.data()));
  }
  std::vector<float> Codebooks() const { std::vector<float> cb((size_t)params_.NumCentroids * dim_); check(coltt_pq_get_codebooks(h_, cb.data())); return cb; }
  std::vector<uint8_t> Encode(const Vector& v) const {
    if (v.size() != dim_) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    std::vector<uint8_t> c((size_t)params_.NumSubVectors); check(coltt_pq_encode(h_, v.data(), 1, c.data())); return c;
  }
  void Insert(uint64_t id, const Vector& v) {
    if (v.size() != dim_) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    check(coltt_pq_upsert(h_, &id, v.data(), 1));
  }
  void InsertMany(const std::vector<uint64_t>& ids, const float* rows) { check(coltt_pq_upsert(h_, ids.data(), rows, ids.size())); }
  void Remove(const std::vector<uint64_t>& ids) { check(coltt_pq_remove(h_, ids.data(), ids.size())); }
  uint64_t Len() const { uint64_t n = 0; check(coltt_pq_len(h_, &n)); return n; }
  // the k nearest by the asymmetric distance, ascending by (score, id)
  SearchResult Search(const Vector& query, unsigned k) const {
    if (query.size() != dim_) throw Error(COLTT_E_INVALID, "Dim Length UnmatchdError");
    std::vector<uint64_t> ids(k); std::vector<float> sc(k); uint32_t n = 0;
    check(coltt_pq_search(h_, query.data(), 1, k, ids.data(), sc.data(), &n));
    SearchResult r(n);
    for (uint32_t i = 0; i < n; i++) r[i] = {ids[i], sc[i]};
    return r;
  }
  coltt_handle_t handle() const { return h_; }

 private:
  coltt_handle_t h_ = 0; uint32_t dim_; ProductQuantizerParameters params_;
};

}  // namespace coltt
