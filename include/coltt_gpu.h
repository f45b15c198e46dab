/* coltt_gpu.h — C-ABI of libcoltt_gpu.so: the MI355X-native ANN search hot path of sjy-dv/coltt.
 *
 * Drop-in boundary.  The reference has no FFI for this path (it is pure Go + Go
 * assembly); these entry points are what a cgo shim binds in order to put the GPU behind the
 * reference's own Go interfaces.  Every function cites the reference interface it replaces.
 *
 * Conventions
 *   - plain C, no torch / C++ types; all handles are opaque uint64.
 *   - the caller owns every input and output buffer; the library copies inputs before returning
 *     (cgo: no Go pointer is retained, no callbacks into Go).
 *   - return value: 0 = ok, <0 = error class (COLTT_E_*); message via coltt_last_error() (thread-local).
 *     Nothing aborts or throws across the boundary.
 *   - all entry points are thread-safe (cgo calls arrive on arbitrary OS threads).
 *   - "*_device" variants take pointers that already live in this GPU's HBM (hipMalloc'd or a
 *     torch.Tensor.data_ptr()); they exist so resident data never crosses PCIe.  The library works on its own
 *     HIP stream: a device buffer handed in must be complete (its producer stream synchronised) before the call,
 *     and results are complete when the call returns.
 */
#ifndef COLTT_GPU_H
#define COLTT_GPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef uint64_t coltt_handle_t;

enum { COLTT_OK = 0, COLTT_E_INVALID = -1, COLTT_E_EXISTS = -2, COLTT_E_NOT_FOUND = -3,
       COLTT_E_UNSUPPORTED = -4, COLTT_E_DEVICE = -5, COLTT_E_NOMEM = -6 };

/* edgepb.Distance (idl/proto/v4/edge.proto:70-73) */
enum { COLTT_COSINE = 0, COLTT_EUCLIDEAN = 1 };
/* distance.SpaceImpl.ManhattanDistance (pkg/distance/space.go:25-29,73-79; simd/cpp/avx.cpp:34-49): exposed by the reference's kernel
 * interface, used by none of its stores — served by coltt_distance_pairs only */
enum { COLTT_MANHATTAN = 2 };
/* edgepb.Quantization (idl/proto/v4/edge.proto:75-80).  NB the reference's "BF16" is IEEE binary16
 * (pkg/compresshelper/bf16.go:233-317 == float16.go:237-321) and its "F8" decodes to 8 distinct
 * values (float8.go:233-313); both are reproduced bit-for-bit. */
enum { COLTT_Q_NONE = 0, COLTT_Q_F16 = 1, COLTT_Q_F8 = 2, COLTT_Q_BF16 = 3 };
/* top-k direction.  REFERENCE = what edge.PriorityQueue really does: min-heap + pop-min keeps the K
 * LARGEST distances (edge/priority_queue.go:39-55), returned ascending (:57-69).  NEAREST = K smallest. */
enum { COLTT_SELECT_REFERENCE = 0, COLTT_SELECT_NEAREST = 1 };
/* FLAT arithmetic.  EXACT = the reference AVX summation order, bit-identical scores
 * (pkg/distance/simd/cpp/avx.cpp:15-32,51-75).  MFMA = matrix-core candidate generation followed by an
 * EXACT re-score of the survivors: candidates are kept within a margin of twice the proven error bound of
 * the approximate score, so the candidate set is a superset of the exact top-k and ids, ranks and score
 * bits EQUAL the EXACT mode's.  Served for cosine and Euclidean, f32 / 2-byte rows, 128 <= dim <= 4096;
 * anything else (f8 rows, smaller dims, the id-list scan) silently runs EXACT — same answers. */
enum { COLTT_MODE_EXACT = 0, COLTT_MODE_MFMA = 1 };

/* ---- process / device ------------------------------------------------------------------------ */
int coltt_init(int device);                 /* selects the HIP device for the calling process      */
int coltt_device_count(void);
const char* coltt_last_error(void);
const char* coltt_version(void);
/* The COLTT_* measurement / test knobs (INTEGRATION.md "Knobs") are read from the environment ONCE, into a process-wide snapshot,
 * when the library is first used; no search call reads the environment.  A program that changes one of them later calls this. */
int coltt_policy_reload(void);

/* ---- kernels exposed one-to-one (pkg/distance, pkg/compresshelper, pkg/sharding, pkg/distancepq) --- */
/* distance.Space.Distance(a,b) for n independent pairs: a,b are row-major [n][dim] host arrays
 * (pkg/distance/space.go:61-63,77-79,93-95; metric COLTT_COSINE, COLTT_EUCLIDEAN or COLTT_MANHATTAN).  order: 0 avx (default dispatch on AVX hosts), 1 sse, 2 native
 * (space.go:40-49). */
int coltt_distance_pairs(int metric, int order, const float* a, const float* b, size_t n, uint32_t dim,
                         float* out);
/* edge.Normalize / vectorindex.Normalize (edge/vectorstore.go:173-189; core/vectorindex/metadata.go:107-123) */
int coltt_normalize(const float* in, size_t n, uint32_t dim, float* out);
/* the same function for ONE vector on the host: no device, no allocation, cannot fail for valid pointers (what the Go layer's
 * vectorindex.Normalize / edge.Normalize call once per RPC; bit-identical to coltt_normalize and to the reference) */
int coltt_normalize_host(const float* in, uint32_t dim, float* out);
/* Quantization.Lower: compresshelper.Fromfloat32 / BF16Fromfloat32 / F8Fromfloat32 per element
 * (edge/f16_quantization.go:47-53; pkg/compresshelper/float16.go:124-126, bf16.go:120-122, float8.go:120-122) */
int coltt_quant_lower(int quant, const float* in, size_t n_elems, void* out_codes);
/* Float16.Float32 / BFloat16.Float32 / Float8.Float32 (float16.go:184-187, bf16.go:180-183, float8.go:180-183) */
int coltt_quant_raise(int quant, const void* codes, size_t n_elems, float* out);
/* sharding.ShardVertex (pkg/sharding/shard.go:34-41) */
int coltt_shard_vertex(const uint64_t* ids, size_t n, uint64_t shard_count, uint64_t* out);
/* distancepq: asm.Dot / asm.SquaredEuclideanDistance (pkg/distancepq/asm/dot.s:7-55, euclidean.s:7-65);
 * kind 0 dot, 1 squared-L2, 2 cosineDistance (1-dot), 3 dotProductDistance (-dot) (distance.go:36-42);
 * one query against n rows. */
int coltt_pq_float_scan(int kind, const float* query, const float* rows, size_t n, uint32_t dim, float* out);
/* hammingDistance / jaccardDistance (pkg/distancepq/distance.go:62-84); kind 0 hamming, 1 jaccard;
 * one query of `words` uint64 against n rows. */
int coltt_pq_bit_scan(int kind, const uint64_t* query, const uint64_t* rows, size_t n, uint32_t words, float* out);

/* ---- edge FLAT store: replaces {none,f16,f8,bf16}VecSpace behind edge.vectorspace
 *      (edge/vectorstore.go:30-49; constructors selected at edge/vectorstore.go:62-85) ------------- */
int coltt_flat_create(uint32_t dim, int metric, int quant, coltt_handle_t* out);
int coltt_flat_destroy(coltt_handle_t h);
int coltt_flat_reserve(coltt_handle_t h, uint64_t n_rows);
/* ChangedVertex, vector half (edge/none_vectorstore.go:86-101; f16_vectorstore.go:87-105): applies
 * Normalize (cosine) and Lower so the stored bits equal the reference's.  Existing ids are overwritten.
 * A vector whose length != dim is the caller's error (none_vectorstore.go:86-88) — dim is fixed here. */
int coltt_flat_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n);
/* same, vectors already in HBM; ids == NULL means ids are first_id, first_id+1, ... (append-only fast path) */
int coltt_flat_upsert_device(coltt_handle_t h, const uint64_t* ids, uint64_t first_id, const float* d_vecs, size_t n);
/* RemoveVertex, vector half (edge/none_vectorstore.go:118-124).  Unknown ids are ignored as in Go's delete(). */
int coltt_flat_remove(coltt_handle_t h, const uint64_t* ids, size_t n);
int coltt_flat_len(coltt_handle_t h, uint64_t* out);
/* stored (lowered) bits of one vertex — what SaveVertex serialises (none_vectorstore.go:308-390) */
int coltt_flat_get(coltt_handle_t h, uint64_t id, void* out_row);
/* stored rows [first_slot, first_slot + n) in scan order, and their ids — the bulk form of coltt_flat_get (what SaveVertex
 * walks); either output may be NULL */
int coltt_flat_fetch_rows(coltt_handle_t h, uint64_t first_slot, uint64_t n, void* out_rows, uint64_t* out_ids);
/* VertexSearch for a batch of queries (edge/none_vectorstore.go:129-180; f16_vectorstore.go:131-186).
 * out_ids/out_scores are [nq][k]; out_counts[q] = min(k, len).  Rows ascending by (score, id). */
int coltt_flat_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select, int mode,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_counts);
int coltt_flat_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, int select, int mode,
                             uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_counts);
/* how many <= 256-query groups went through the matrix cores since creation, and how many of those overflowed their candidate
 * list and were re-run in exact mode (adversarial data only; the answers are the exact mode's either way) */
int coltt_flat_stats(coltt_handle_t h, uint64_t* mfma_groups, uint64_t* mfma_fallbacks);
/* Running bounds of the stored ||row||^2 over EVERYTHING the store ever held (they are not recomputed on removal), and whether the cosine
 * matrix-core path is open: it is closed for a store that ever held a row with ||row||^2 outside [1/4, 4] (a zero vector, a loaded stream
 * that was never normalised) — such a store answers COLTT_MODE_MFMA batches through the exact scan: same answers, ~25x the time at batch
 * 256.  Nothing is logged when the latch closes; this is where a caller sees it.  An empty store reports NaN bounds. */
int coltt_flat_norm_bounds(coltt_handle_t h, float* out_min, float* out_max, int32_t* out_cosine_matrix_core_open);
/* searches of <= 4 queries and k <= 64 are served by ONE kernel launch whatever `mode` says (scan in exact order, per-wave and
 * per-block k best, selection by the last block to finish): the reference's one-query-per-RPC shape (edge/edge_search.go).  Same
 * answers as both modes; this counts them (COLTT_FLAT_ONE=0 in the environment turns the path off). */
int coltt_flat_one_launch_searches(coltt_handle_t h, uint64_t* out);
/* FilterableVertexSearch (edge/none_vectorstore.go:182-253): the candidate ids come from the roaring
 * index (pkg/inverted/search.go:113-119) on the Go side; ids not present are skipped (:201). */
int coltt_flat_search_ids(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select,
                          const uint64_t* cand_ids, size_t n_cand,
                          uint64_t* out_ids, float* out_scores, uint32_t* out_counts);
/* the same with an explicit mode: COLTT_MODE_MFMA generates candidates on the matrix cores from the GATHERED rows (batches of
 * filtered queries; same ids, ranks and score bits as COLTT_MODE_EXACT, which coltt_flat_search_ids uses) */
int coltt_flat_search_ids_mode(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select, int mode,
                               const uint64_t* cand_ids, size_t n_cand,
                               uint64_t* out_ids, float* out_scores, uint32_t* out_counts);

/* SaveVertex / LoadVertex (edge/none_vectorstore.go:308-516; f16_vectorstore.go:317-532 and the f8/bf16 twins): 16 shards
 * x {u64 count, count x {u64 key, u32 vecLen, vecLen x big-endian STORED code (f32 | u16 | u8), u32 metaCount, typed
 * pairs}}.  Vector codes go straight into the HBM rows (byte-swapped on the device, no re-normalisation); metadata is
 * opaque to the library: load reports each vertex's blob position in buf, save re-emits the blobs the caller passes
 * (meta_ids[i] -> meta_blobs[i], each starting with its u32 metaCount; others are written as empty maps). */
int coltt_flat_load_vertex(coltt_handle_t h, const uint8_t* buf, uint64_t len, uint64_t* out_n, uint64_t* out_ids,
                           uint64_t* out_meta_off, uint32_t* out_meta_len, uint64_t cap_n);
int coltt_flat_save_vertex(coltt_handle_t h, const uint64_t* meta_ids, const uint8_t* const* meta_blobs,
                           const uint32_t* meta_lens, uint64_t n_meta, uint8_t* out, uint64_t cap, uint64_t* out_len);

/* ---- experimental CFLAT: multi-vector weighted FLAT scan (experimental/multi_vector_vertex.go:60-137) -------------
 * A vertex carries n_fields f32 vectors ([n][n_fields][dim] on upload; every field is normalised for cosine, :65-67).
 * MultiVertexSearch: score = sum over included fields of scoreHelper(Distance(node[f], q[f])) * (float32(ratio[f])/100)
 * (:113-119); the K LARGEST scores are kept and returned DESCENDING by (score, id) (multi_priority_queue.go:46-77). */
int coltt_cflat_create(uint32_t dim, int metric, uint32_t n_fields, coltt_handle_t* out);
int coltt_cflat_destroy(coltt_handle_t h);
int coltt_cflat_len(coltt_handle_t h, uint64_t* out);
int coltt_cflat_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n);
int coltt_cflat_remove(coltt_handle_t h, const uint64_t* ids, size_t n);
int coltt_cflat_search(coltt_handle_t h, const float* queries, const uint32_t* ratios, const uint8_t* include, size_t nq,
                       uint32_t k, uint64_t* out_ids, float* out_scores, uint32_t* out_counts);

/* ---- core HNSW: replaces *vectorindex.Hnsw (core/vectorindex/hnsw.go:43-54) --------------------- */
typedef struct coltt_hnsw_cfg {       /* hnswConfig defaults: hnsw_config.go:135-162 */
  int32_t m;                          /* 16 */
  int32_t m_max;                      /* -1 -> m */
  int32_t m_max0;                     /* -1 -> 2m */
  int32_t ef;                         /* 20 */
  int32_t ef_construction;            /* 200 */
  int32_t algo;                       /* 0 HnswSearchSimple, 1 HnswSearchHeuristic (hnsw_config.go:27-33) — both keep the k nearest —, or
                                         COLTT_HNSW_DIVERSE (2): NOT reference behaviour, see below */
  float level_multiplier;             /* -1 -> 1/ln(m) */
  int32_t extend_candidates;          /* must be 0: the reference's extend path is undefined */
  int32_t keep_pruned;                /* 1 (dead code in the reference) */
} coltt_hnsw_cfg;

/* COLTT_HNSW_DIVERSE — an opt-in neighbour selection the reference does NOT have: its selectNeighborsHeuristic (hnsw.go:399-447) never
 * compares a candidate with the neighbours already chosen, so both of its modes build a plain k-nearest graph.  This mode applies the
 * diversity test of the HNSW paper (Algorithm 4, as hnswlib runs it) in Insert and in pruneNeighbors-on-overflow: candidates ascending
 * by (distance, slot); a candidate is chosen iff no already chosen neighbour is closer to it than the base vertex is; keep_pruned re-adds
 * the rejected ones, nearest first.  Defined in oracle/coltt_oracle.cpp (select_diverse); GPU graph == that definition bit for bit.
 * Search is unchanged (Hnsw.Search over whatever graph was built).  A graph built this way is NOT the reference's graph: the Go drop-in
 * refuses the value unless the caller opts in (go/vectorindex). */
#define COLTT_HNSW_DIVERSE 2

typedef struct coltt_hnsw_stats {     /* per search call, summed over the batch */
  uint64_t n_dist;                    /* distance evaluations */
  uint64_t n_exp;                     /* expanded candidates on level 0 */
  uint64_t n_hops;                    /* greedy hops on upper levels */
  uint64_t n_visit_resets;            /* visited-set resets (0 => traversal identical to the oracle's) */
} coltt_hnsw_stats;

/* NewHnsw(dim, distancer, options...) (hnsw.go:56-73).  quant != NONE stores 2-/1-byte codes and
 * evaluates distances as the edge quantised stores do (decode both, f32 distance) — an extension the
 * reference's core does not have (BASELINE.json configs[4]). */
int coltt_hnsw_create(uint32_t dim, int metric, int quant, const coltt_hnsw_cfg* cfg, coltt_handle_t* out);
int coltt_hnsw_destroy(coltt_handle_t h);
int coltt_hnsw_get_cfg(coltt_handle_t h, coltt_hnsw_cfg* out);
/* Hnsw.RandomLevel() (hnsw.go:280-282) for a caller-supplied uniform draw u in (0,1) — the value the shim's
 * rand.Float32() returned (gomath/rand.go:42-44: -Log(u) * levelMultiplier; math.go:52-62: float32 log, Floor through
 * float64).  The reference draws from the auto-seeded global math/rand, so the draw itself stays on the Go side.
 * u <= 0 (the reference's +Inf -> undefined int) or u >= 1 is COLTT_E_INVALID. */
int coltt_hnsw_random_level(coltt_handle_t h, float u, int32_t* out_level);
/* Load a graph built elsewhere (the oracle, Hnsw.Load's stream, another shard): slot-major arrays.
 * vectors are raw (Normalize/Lower are applied here for cosine/quant, as Insert does, hnsw.go:105-107).
 * rows = sum(level+1); row r of slot s, level l holds nbr[row_offsets[r] .. row_offsets[r+1]).  */
int coltt_hnsw_bulk_load(coltt_handle_t h, uint64_t n, const uint64_t* ids, const int32_t* levels,
                         const uint8_t* deleted, const float* vectors, const int64_t* row_offsets,
                         const int32_t* nbr, const float* nbr_dist, int32_t entry_slot);
/* Hnsw.Insert(id, value, metadata, vertexLevel) (hnsw.go:104-167); COLTT_E_EXISTS = ItemAlreadyExistsError */
int coltt_hnsw_insert(coltt_handle_t h, uint64_t id, const float* vec, int32_t level);
/* n Inserts in id order; `batch` vertices are linked against a frozen graph at a time (batch==1 is the
 * reference's sequential semantics).  d_vecs lives in HBM; ids == NULL means first_id + i. */
int coltt_hnsw_insert_batch_device(coltt_handle_t h, const uint64_t* ids, uint64_t first_id, const float* d_vecs,
                                   const int32_t* levels, size_t n, uint32_t batch);
/* Hnsw.Remove(id) (hnsw.go:191-241); COLTT_E_NOT_FOUND = ItemNotFoundError */
int coltt_hnsw_remove(coltt_handle_t h, uint64_t id);
int coltt_hnsw_len(coltt_handle_t h, uint64_t* out);
/* Hnsw.Search(ctx, query, k) for a batch (hnsw.go:243-278): ef = max(cfg.ef or ef_override, k); rows
 * ascending by distance; empty index => counts 0, not an error (hnsw.go:249-251). */
int coltt_hnsw_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, uint32_t ef_override,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats);
int coltt_hnsw_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, uint32_t ef_override,
                             uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_counts, coltt_hnsw_stats* stats);
/* Capacity for n_slots vertices (and n_upper_rows upper-level adjacency rows; 0 = the expectation for this index's M) in ONE allocation per
 * array, before the inserts.  Optional — Insert grows the arrays by half their size when they are full — but an index whose final size is
 * known should reserve it: growth copies every array (30 GB of rows at 10 M x 768 f32), and arrays allocated once from an empty heap get the
 * largest contiguous physical fragments, which the random row reads of a search feel in their TLB miss rate. */
int coltt_hnsw_reserve(coltt_handle_t h, uint64_t n_slots, uint64_t n_upper_rows);
/* Round 4: indexes whose rows are f32 / 2-byte codes of a byte length that is a multiple of 128 (dim 128, 256, 512, 768, 1024, 1536 ...)
 * keep a second, line-transposed copy of their rows (derived data, like the adjacency-carried norms) and evaluate the level-0
 * distances of a search with EIGHT lanes per row over it — whole 128-byte lines per load instruction — in the reference's summation
 * order: same ids, score bits and counters (coltt_amd/csrc/rows8.hpp).  COLTT_ROWS8=0 (at create) keeps an index without the copy,
 * COLTT_EV8=0 (per call) searches the pair-owned rows.  This reports how many search launches the eight-lane core served and whether
 * the copy is complete. */
int coltt_hnsw_rows8_searches(coltt_handle_t h, uint64_t* out_launches, int32_t* out_has_copy);
/* graph export in the bulk_load layout (what Hnsw.Commit serialises, hnsw_commit.go:69-162).
 * Call with NULL arrays to get sizes.  With any array non-NULL, *n_slots / *n_rows / *n_edges are IN-OUT: on entry the
 * capacities of the caller's arrays (slots: ids, levels, deleted; rows + 1: row_offsets; edges: nbr, nbr_dist) — normally the
 * sizes the first call returned — and if an Insert has grown the index past them in between, nothing is written, the needed
 * sizes are returned and the call fails with COLTT_E_INVALID (retry with larger arrays). */
int coltt_hnsw_export(coltt_handle_t h, uint64_t* n_slots, uint64_t* n_rows, uint64_t* n_edges, uint64_t* ids,
                      int32_t* levels, uint8_t* deleted, int64_t* row_offsets, int32_t* nbr, float* nbr_dist,
                      int32_t* entry_slot);
/* Hnsw.Commit(w, header) (core/vectorindex/hnsw_commit.go:69-162): the reference's big-endian stream — [config
 * (hnsw_config.go:179-203), u32 dim, u8 distIdx], u64 entrypoint id, 16 FNV shards of {u64 id, i32 level, dim x f32,
 * metadata}, then per vertex {u64 id, per level (top down) u32 n, n x (u64 neighbour id, f32 distance)}; removed
 * vertices and edges to them are skipped.  meta_blobs[slot]/meta_lens[slot] = the vertex's Metadata already in stream
 * encoding (metadata.go:31-74: u16 pairs, {u8 keylen, key, u16 vallen, msgpack}); NULL => empty maps.  n_meta = the length
 * of both arrays: slots >= n_meta (vertices inserted after the caller sized them) get empty metadata, never an out-of-bounds read.
 * out == NULL => only *out_len is computed.
 * Quantised indexes: a binary16 ("f16" / "bf16") index writes the f32 values its codes stand for; coltt_hnsw_load into an index of the
 * same quantisation encodes them back to the same codes (binary16 encode(decode(c)) == c), so Commit -> Load is bit-identical and the
 * stream stays in the reference's format.  "f8" indexes return COLTT_E_UNSUPPORTED (that codec does not survive decode -> encode). */
int coltt_hnsw_commit(coltt_handle_t h, int header, const uint8_t* const* meta_blobs, const uint32_t* meta_lens, uint64_t n_meta,
                      uint8_t* out, uint64_t cap, uint64_t* out_len);
/* level of the entrypoint (what Hnsw.BytesSize needs, hnsw.go:476-490), -1 for an empty index: no device traffic */
int coltt_hnsw_entry_level(coltt_handle_t h, int32_t* out_level);
/* Hnsw.Load(r, header) (hnsw_commit.go:164-278) straight into the HBM layout: vectors are byte-swapped on the device
 * and NOT re-normalised (as in the reference).  Slots follow stream order.  out_ids / out_meta_off / out_meta_len
 * (capacity cap_n, may be NULL) receive each vertex's id and the position of its metadata blob inside buf, so the
 * caller can decode the msgpack values itself. */
int coltt_hnsw_load(coltt_handle_t h, int header, const uint8_t* buf, uint64_t len, uint64_t* out_n, uint64_t* out_ids,
                    uint64_t* out_meta_off, uint32_t* out_meta_len, uint64_t cap_n);
/* the adjacency exactly as it lives in HBM: adj0 [n][m_max0], upper_off [n], adjU [n_upper][m_max],
 * padded with 0xffffffff.  NULL arrays => sizes only. */
int coltt_hnsw_export_raw(coltt_handle_t h, uint64_t* n_slots, uint64_t* n_upper_rows, int32_t* entry_slot,
                          int32_t* entry_level, uint32_t* adj0, uint32_t* upper_off, uint32_t* adjU);
/* Hnsw.Get(id) / Hnsw.GetVertex(id) (hnsw.go:169-189): the stored (normalised / lowered) vector and the level of one live
 * vertex; COLTT_E_NOT_FOUND = ItemNotFoundError.  Either output may be NULL. */
int coltt_hnsw_get(coltt_handle_t h, uint64_t id, void* out_row, int32_t* out_level);
/* the same read-back for a slot range (bulk): the stored row bytes. */
int coltt_hnsw_fetch_rows(coltt_handle_t h, uint64_t first_slot, uint64_t n, void* out_rows);
/* last kernel timing of the handle's search stream, measured with hipEvents (milliseconds) */
int coltt_last_kernel_ms(coltt_handle_t h, float* out_ms);
/* the FLAT half of the call above (coltt_last_kernel_ms forwards a FLAT store's handle here); exported, hence declared */
int coltt_last_kernel_ms_flat(coltt_handle_t h, float* out_ms);

/* ---- collection groups: ONE collection partitioned over the GPUs of a node (BASELINE.json north_star) ----
 * The reference has no multi-device code; its sharding rule and merge shape are the ones `highCpu` uses for its 16
 * in-process map shards: vertex `id` lives on shard sharding.ShardVertex(id, world) (pkg/sharding/shard.go:34-41), every
 * shard is searched with a local queue, the local results are merged into one queue (edge/none_vectorstore.go:148-178).
 * SHARD layout: each member is an ordinary FLAT store / HNSW index on its own GPU; a search = per-member search on per-member
 * streams + ONE all-gather of packed {u64 id, f32 score, u32 valid} records (RCCL over xGMI; ncclCommInitAll in one process,
 * ncclCommInitRank with one process per GPU) + host-side merge in the canonical (score, id) order.  REPLICA layout: every
 * member holds everything, a query batch is split across members, nothing is exchanged.
 * RCCL is dlopen'ed on first use (no link-time dependency).  EXCHANGE_HOST / AUTO on a group whose members share one device
 * (RCCL refuses a device twice) moves the records through pinned host memory instead; the merge is on the host either way. */
enum { COLTT_GROUP_FLAT = 0, COLTT_GROUP_HNSW = 1 };
enum { COLTT_LAYOUT_SHARD = 0, COLTT_LAYOUT_REPLICA = 1 };
enum { COLTT_EXCHANGE_AUTO = 0, COLTT_EXCHANGE_RCCL = 1, COLTT_EXCHANGE_HOST = 2,
       COLTT_EXCHANGE_SHM = 3 /* processes of ONE box: POSIX shared memory + a process-shared barrier (coltt_shm_*) */ };
#define COLTT_UNIQUE_ID_BYTES 128
typedef struct coltt_group_opts {
  int32_t kind;               /* COLTT_GROUP_FLAT (edge vectorspace) | COLTT_GROUP_HNSW (core vectorindex)                  */
  int32_t layout;             /* COLTT_LAYOUT_SHARD | COLTT_LAYOUT_REPLICA                                                  */
  int32_t exchange;           /* COLTT_EXCHANGE_AUTO | _RCCL (fail if unavailable) | _HOST (one process) | _SHM (processes of a box) */
  int32_t world_size;         /* shards in the whole collection; 0 => n_devices (single process)                            */
  int32_t rank_base;          /* shard number of devices[0]; this process hosts shards rank_base .. rank_base+n_devices-1   */
  const uint8_t* unique_id;   /* COLTT_UNIQUE_ID_BYTES from coltt_group_unique_id(), the same in every process; NULL if one */
} coltt_group_opts;
int coltt_group_unique_id(uint8_t* out /*[COLTT_UNIQUE_ID_BYTES]*/);   /* ncclGetUniqueId, or random bytes when librccl is absent (SHM only) */
/* All-gather between the processes of one box through POSIX shared memory (host-only: no device call) — the transport of
 * COLTT_EXCHANGE_SHM groups, same packed records and same merge as the RCCL path; exported so the multi-process rendezvous,
 * barrier and chunking can be exercised without a GPU.  Every process passes the same unique_id, world and bytes_per_rank; it
 * hosts ranks rank_base .. rank_base + n_local - 1.  open() returns once all `world` ranks have attached (timeout:
 * COLTT_SHM_TIMEOUT_S, default 120 s).  allgather: local = n_local x bytes, out = world x bytes (rank-major), bytes <= bytes_per_rank
 * and equal in every process; collective — every process must make the same sequence of calls. */
int coltt_shm_open(const uint8_t* unique_id, int world, int n_local, int rank_base, uint64_t bytes_per_rank, coltt_handle_t* out);
int coltt_shm_allgather(coltt_handle_t h, const void* local, uint64_t bytes, void* out);
int coltt_shm_close(coltt_handle_t h);
int coltt_group_create(const int* devices, int n_devices, uint32_t dim, int metric, int quant, const coltt_hnsw_cfg* cfg,
                       const coltt_group_opts* opts, coltt_handle_t* out);
int coltt_group_destroy(coltt_handle_t h);
int coltt_group_info(coltt_handle_t h, int32_t* n_local, int32_t* world, int32_t* exchange_in_use, int32_t* rank_base);
/* the i-th local member's own handle (a coltt_flat_* / coltt_hnsw_* handle): device-resident ingest goes straight to it */
int coltt_group_member(coltt_handle_t h, int i, coltt_handle_t* out);
int coltt_group_shard_of(coltt_handle_t h, uint64_t id, int32_t* out_shard);
/* ChangedVertex (FLAT) / Insert (HNSW) routed by ShardVertex(id, world).  A process is OFFERED every vertex and keeps those
 * whose shard it hosts (*out_kept); a replica group gives every vertex to every member. */
int coltt_group_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n, uint64_t* out_kept);
int coltt_group_insert(coltt_handle_t h, const uint64_t* ids, const float* vecs, const int32_t* levels, size_t n, uint32_t batch,
                       uint64_t* out_kept);
int coltt_group_remove(coltt_handle_t h, const uint64_t* ids, size_t n);
int coltt_group_len(coltt_handle_t h, uint64_t* out);   /* vertices hosted by THIS process (replica: of one member) */
/* VertexSearch / Hnsw.Search over the whole collection; results on the host, rows ascending by (score, id).
 * select / mode: FLAT only; ef_override: HNSW only.  In a multi-process group every process must make the same call. */
int coltt_group_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select, int mode, uint32_t ef_override,
                       uint64_t* out_ids, float* out_scores, uint32_t* out_counts);
/* same, the query batch already resident on every local member's device: d_queries_per_member[i] -> [nq][dim] f32 */
int coltt_group_search_device(coltt_handle_t h, const float* const* d_queries_per_member, size_t nq, uint32_t k, int select, int mode,
                              uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts);
/* Streaming form of a SHARD-layout search (the exchange "issued on a comm stream and overlapped with the next batch";
 * the shape of the reference's local-queue-then-merge scan, edge/none_vectorstore.go:148-178, with the merge taken off the critical
 * path).  _begin returns when every local member has searched the batch and the batch's exchange (pack + ONE all-gather + D2H on the
 * members' comm streams) and host merge have been queued behind the earlier batches'; _end blocks until the merged answers are in the
 * out arrays handed to _begin (they must stay valid until then — also when the group is destroyed with the batch in flight: destroy drains
 * its queue first) and returns that batch's status.  Exactly one of queries (host) /
 * d_queries_per_member (device) is non-NULL.  At most 3 batches are between their search and the end of their merge (a 4th _begin waits for a slot); a ticket that is never
 * ended is dropped when the group is destroyed.  Multi-process groups: every process
 * makes the same _begin calls in the same order (the collectives are issued in ticket order). */
int coltt_group_search_begin(coltt_handle_t h, const float* queries, const float* const* d_queries_per_member, size_t nq, uint32_t k,
                             int select, int mode, uint32_t ef_override, uint64_t* out_ids, float* out_scores, uint32_t* out_counts,
                             uint64_t* out_ticket);
int coltt_group_search_end(coltt_handle_t h, uint64_t ticket);
/* cumulative wall-clock of this group's finished shard-search batches: out4 = {batches, search_ms (stage A: the members' searches),
 * exchange_ms (pack + all-gather + D2H), merge_ms (host merge)} */
int coltt_group_timing(coltt_handle_t h, double* out4);
/* the host-side final merge on its own (no device needed): recs = [world][nq][k] packed 16-byte records {u64 id, f32 score,
 * u32 valid}, each shard's valid records ascending by (score, id); nearest != 0: the k smallest of the union, else the k
 * largest (edge.PriorityQueue semantics), both returned ascending. */
int coltt_group_merge_host(const void* recs, int world, size_t nq, uint32_t k, int nearest, uint64_t* out_ids, float* out_scores,
                           uint32_t* out_counts);
/* Failure of ONE rank's shard search (round 6).  The `valid` word of a packed record carries, above bit 0, the STATUS of the rank that
 * packed it (bits 8..31; 0 = its shard search succeeded).  A rank whose search fails still takes part in the batch's exchange, with a
 * block of status records: no peer is left waiting in an all-gather that never comes (the shape of a failing shard goroutine that still
 * signals its WaitGroup, edge/none_vectorstore.go:148-178), every rank returns an error for THAT batch (coltt_group_search /
 * _search_end) and the group stays usable.  The RCCL exchange is waited for with a deadline (COLTT_EXCHANGE_TIMEOUT_S, default 120 s):
 * a peer that died or never made the call turns into an error, and the group refuses further shard searches.
 * This helper (no device needed) returns the first rank of recs = [world][per] records whose block carries a status, or -1. */
int coltt_group_first_failed_rank_host(const void* recs, int world, size_t per, uint32_t* out_status);
/* sharding.ShardVertex on the host (pkg/sharding/shard.go:34-41): the routing rule of a group, no device needed */
uint64_t coltt_shard_vertex_host(uint64_t id, uint64_t shard_count);

/* ---- product-quantised store: codebooks, Encode, the per-query distance table and the ADC scan -----------
 * The reference declares the parameters (models.ProductQuantizerParameters, pkg/models/hnsw_common.go:20-33: NumCentroids in
 * [2,256] -> one uint8 code per sub-vector, NumSubVectors >= 2) and ships the arithmetic (pkg/distancepq/distance.go:30-42 over
 * asm/dot.s:7-55 and asm/euclidean.s:7-65), but the package that drove them (pkg/hnswpq, imported by
 * playground/hnswpq_verification.go:29; call shape :69-73, 90-105, 154, 190-199) is not in its tree.  These entry points are what a
 * binding of that package's quantiser would call; their results are DEFINED from the distancepq arithmetic (oracle/coltt_oracle.cpp,
 * "Product quantiser"):
 *   Encode  code[j] = argmin_c SquaredEuclideanDistance(x_j, centroid[j][c])  (strict <, c ascending, from MaxFloat32)
 *   LUT     lut[j][c] = distFn(q_j, centroid[j][c]);   score = sum over j = 0..m-1 of lut[j][code[j]], f32, in j order
 *   top-k   the k smallest by (score bits, id), ascending.
 * metric = which distancepq function is distFn: */
enum { COLTT_PQ_COSINE = 0 /* cosineDistance = 1 - Dot */, COLTT_PQ_EUCLIDEAN = 1 /* euclideanDistance = SQUARED L2 */,
       COLTT_PQ_DOT = 2 /* dotProductDistance = -Dot */ };
int coltt_pq_create(uint32_t dim, int metric, uint32_t num_subvectors, uint32_t num_centroids, coltt_handle_t* out);
int coltt_pq_destroy(coltt_handle_t h);
/* codebooks trained elsewhere: [num_subvectors][num_centroids][dim / num_subvectors] f32, row-major.  Refused once rows are stored. */
int coltt_pq_set_codebooks(coltt_handle_t h, const float* codebooks);
int coltt_pq_get_codebooks(coltt_handle_t h, float* out_codebooks);
/* the quantiser's training step on a sample of n >= num_centroids vectors (TriggerThreshold, hnsw_common.go:29-32): deterministic
 * Lloyd iterations — centroid c of sub-space j starts as sub-vector j of sample vector c; assignment = Encode; update = f32 sum in
 * sample order / float32(count); an empty cluster keeps its centroid.  Refused once rows are stored. */
int coltt_pq_train(coltt_handle_t h, const float* vecs, size_t n, uint32_t iterations);
/* Encode without storing: out_codes [n][num_subvectors] */
int coltt_pq_encode(coltt_handle_t h, const float* vecs, size_t n, uint8_t* out_codes);
/* store vectors (encoded on the GPU) / ready codes under ids; existing ids are overwritten; codes >= num_centroids are refused */
int coltt_pq_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n);
int coltt_pq_upsert_device(coltt_handle_t h, const uint64_t* ids, uint64_t first_id, const float* d_vecs, size_t n);
int coltt_pq_upsert_codes(coltt_handle_t h, const uint64_t* ids, const uint8_t* codes, size_t n);
int coltt_pq_remove(coltt_handle_t h, const uint64_t* ids, size_t n);
int coltt_pq_len(coltt_handle_t h, uint64_t* out);
/* stored codes of rows [first_slot, first_slot + n) in scan order ([n][num_subvectors]) and their ids; either output may be NULL */
int coltt_pq_fetch_codes(coltt_handle_t h, uint64_t first_slot, uint64_t n, uint8_t* out_codes, uint64_t* out_ids);
/* the distance table of one query: out_lut [num_subvectors][num_centroids] */
int coltt_pq_lut(coltt_handle_t h, const float* query, float* out_lut);
/* ADC search of a batch: out_ids / out_scores [nq][k], out_counts[q] = min(k, len); rows ascending by (score, id) */
int coltt_pq_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, uint64_t* out_ids, float* out_scores,
                    uint32_t* out_counts);
int coltt_pq_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                           uint32_t* d_out_counts);
/* hipEvent times of the most recent search of this store: the whole call's kernels, and the scan launch over the last (largest)
 * segment alone — the kernel the roofline of the PQ leg is quoted on — with the number of rows that launch covered */
int coltt_pq_last_kernel_ms(coltt_handle_t h, float* out_search_ms, float* out_scan_ms, uint64_t* out_scan_rows);


/* ---- product-quantised HNSW: Hnsw.Search over the quantiser's codes with an exact re-rank ------------------------------------
 * The reference's only PQ call shape (playground/hnswpq_verification.go:69-105: hnswpq.NewProductQuantizationHnsw(), m = 32, 256
 * centroids, pre-train, Fit, search on the codes; UPDATE-LOG.md:190-194) — its package pkg/hnswpq is absent, so this is a DEFINITION
 * (oracle/coltt_oracle.cpp "Product-quantised HNSW"), assembled from the pinned pieces: the graph and traversal of
 * core/vectorindex/hnsw.go:243-278,320-389 and the quantiser above (pkg/distancepq/distance.go:30-42):
 *   codes    Encode(stored row as the index's distance sees it), one row-major code per slot, kept up to date by Insert / Load
 *   d(q, v)  = S_lo + S_hi, the f32 sums (j order, from +0.0) of float32(binary16(lut[j][code_v[j]])) over the first ceil(P / 2) and the remaining
 *            16-byte pieces of the code row (P = ceil(m / 16)): the quantiser's table over the query the index's distance sees (normalised /
 *            lowered), every entry rounded to binary16 (nearest even) — d only ranks, and a 2-byte table doubles the walk's resident traversals
 *   walk     Hnsw.Search with d in place of Distance() (entrypoint, upper levels, searchLevel(ef)); ties by (d bits, slot).  Bounded visiting:
 *            once the result set is full, a neighbour whose d is not below lowerBound is skipped before the visited test (it can never be
 *            admitted: the bound only falls) — the result sets are exactly the unbounded walk's, n_dist counts what passed the bound and was fresh
 *   re-rank  the min(max(rerank, k), |result set|) nearest by d (rerank = 0: the whole result set) are re-scored with the index's
 *            exact-order distance; the k smallest by (exact score, slot) are returned with their exact scores.
 * attach snapshots the (trained) quantiser's codebooks — later changes to `pq` do not reach the index — and encodes every stored row.
 * Supported: f32 / binary16 rows; euclideanDistance tables on any index, cosineDistance tables on a cosine index; <= 128 sub-vectors. */
int coltt_hnsw_pq_attach(coltt_handle_t hnsw, coltt_handle_t pq);
int coltt_hnsw_pq_info(coltt_handle_t hnsw, uint32_t* out_num_subvectors, uint32_t* out_num_centroids, int32_t* out_pq_metric, uint64_t* out_coded_slots);
/* codes of slots [first_slot, first_slot + n): out_codes [n][num_subvectors] */
int coltt_hnsw_pq_fetch_codes(coltt_handle_t hnsw, uint64_t first_slot, uint64_t n, uint8_t* out_codes);
/* stats: n_dist = table-distance evaluations that counted (entrypoint, upper levels, every fresh neighbour while the set fills, afterwards the fresh
 * neighbours under the bound), n_exp / n_hops as Hnsw.Search; *out_n_exact (may be NULL) = exact re-rank evaluations */
int coltt_hnsw_pq_search(coltt_handle_t hnsw, const float* queries, size_t nq, uint32_t k, uint32_t ef_override_or_0, uint32_t rerank,
                         uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats, uint64_t* out_n_exact);
int coltt_hnsw_pq_search_device(coltt_handle_t hnsw, const float* d_queries, size_t nq, uint32_t k, uint32_t ef_override_or_0, uint32_t rerank,
                                uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_counts, coltt_hnsw_stats* stats, uint64_t* out_n_exact);

#ifdef __cplusplus
}
#endif
#endif /* COLTT_GPU_H */
