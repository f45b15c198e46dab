// hnsw_dev.hpp — device-side HNSW traversal shared by the search kernel and the graph builder.
//
// One wave64 owns one query (or one vertex being inserted).  State in LDS, per wave:
//   qs   [dim] f32         the query as the distance kernel sees it (normalised / decoded)
//   res  [ef_pad] u64      result set, sorted ascending, merged in place;
//                          entry = d_bits<<32 | slot<<1 | expanded
//   vis  [hcap] u32        open-addressed visited set (slots), linear probing, EMPTY = 0xffffffff   (VISG = false)
// or, VISG = true, the visited set lives in HBM: one byte per slot in a region private to the workgroup, holding the epoch of
// the last traversal that visited the slot (no clearing between traversals; wiped when the 8-bit epoch wraps).  That costs one
// dependent global access per neighbour chunk and ~2 cache lines per visit, but frees 32-128 KB of LDS per wave: 8 waves per
// CU instead of 4 (ef 128), 2 (ef 256, efConstruction 200) or 1 (ef >= 512).
//
// Algorithm = the canonical closed form of Hnsw.searchLevel (core/vectorindex/hnsw.go:345-389) worked out in
// SURVEY.md §3.2 and restated on the CPU by oracle/coltt_oracle.cpp:search_level_canon:
//   * the candidate min-heap is "the unexpanded members of the result set" (every candidate is also pushed to
//     the result heap, hnsw.go:375-377, and a popped candidate worse than the worst result ends the loop, :359-361);
//   * lowerBound is sampled ONCE per popped candidate (:357) and not refreshed inside the neighbour loop (:374);
//   * in canonical neighbour order (ascending slot — Go's map order is random), the first (ef - len0) eligible
//     neighbours are admitted unconditionally, the others iff d < lowerBound; then the ef smallest are kept;
//   * ties are ordered by (distance, slot).
// Distances are evaluated 32 neighbours at a time (lane pair per row) in the reference's AVX summation order.
#pragma once
#include "exact.hpp"

namespace coltt {
namespace dev {

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
constexpr uint32_t VIS_EMPTY = 0xffffffffu;
constexpr uint32_t NBR_NONE = 0xffffffffu;

struct GraphView {
  const uint8_t* rows; size_t stride; const float* norms; const uint64_t* ids;
  const uint8_t* rows8;               // rows8.hpp: line-transposed copy of rows for the eight-lanes-per-row distance core, or null
  uint32_t* adj0; float* adj0_d;      // [cap][mMax0]   level-0 rows, ascending slot, padded with NBR_NONE
  float* adj0_n;                      // [cap][mMax0]   norms[adj0[..]] — cosine only, else null: the neighbours' ||row||^2 ride with the
                                      //                adjacency row instead of one 4-byte gather (= one cache line) per evaluation
  const uint32_t* upper_off;          // [cap]          first upper row of a slot (levels 1..L consecutive)
  uint32_t* adjU; float* adjU_d;      // [ucap][mMax]
  const uint32_t* del_bits;           // tombstones (hnswVertex.deleted, hnsw_vertex.go:70-76) or null when none
  uint32_t mMax, mMax0;
  int dim;
};

struct WaveCtx {
  float* qs; unsigned long long* res0; uint32_t* vis;  // res buffer b = res0 + b * ef_pad
  float* qp; uint32_t* scr;           // rows8.hpp kernels: the query in rows8 order, and 96 words of scratch (slot | norm | distance of <= 32 fresh rows)
  uint32_t ef_pad, hcap_mask, hcap;
  float qnorm;
  // counters (wave-uniform)
  uint32_t n_dist, n_exp, n_hops, n_resets;
  uint8_t* visg; size_t vis_bytes; uint32_t epoch;  // VISG: this workgroup's byte-per-slot region, its size, current epoch
  uint32_t* bloom; uint32_t bloom_words, bloom_shift;  // hnsw_walk2.hpp: LDS Bloom filter in front of the byte map (words = 2^(32-shift))
#ifdef COLTT_PHASE_TIMING
  unsigned long long pt[8], t_last;  // shader-clock ticks per traversal phase (diagnostic build only)
#endif
  uint32_t err;  // watchdog: 1 visited-set probe overflow, 2 expansion budget, 3 greedy hop budget (every loop is bounded)
};

// hipcc treats `lane`-derived predicates as loop-invariant and may thread such divergent branches through a loop's
// back edge, which splits the wave around convergent operations (ballot / shuffle / readlane) — observed as an
// endless loop in the work-fetch of the search kernel.  Re-reading the lane id through an opaque asm at the top of
// every traversal loop iteration makes the predicates non-invariant and keeps the wave converged.
#ifdef COLTT_PHASE_TIMING
#define COLTT_PT(W, K) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); (W).pt[K] += t_ - (W).t_last; (W).t_last = t_; }
#else
#define COLTT_PT(W, K)
#endif
// Cross-lane LDS hand-off inside ONE wave (the traversal kernels run single-wave workgroups).  LDS operations of a wave
// execute in order, so a later read by one lane sees an earlier write by another without any hardware wait; what is needed is
// only that the compiler keeps the order.  __syncthreads() would also drain vmcnt — i.e. wait for every global load in
// flight, which is exactly what the adjacency prefetch must not do.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int opaque_lane(int lane) { asm volatile("" : "+v"(lane)); return lane; }

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int j) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, j);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), j);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
  uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, 64);
  uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, 64);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { unsigned long long o = shfl_xor_u64(v, m); v = o < v ? o : v; }
  return v;
}

__device__ __forceinline__ bool is_deleted(const GraphView& g, uint32_t slot) {
  return g.del_bits && ((g.del_bits[slot >> 5] >> (slot & 31)) & 1u);
}

__device__ __forceinline__ const uint32_t* adj_row(const GraphView& g, uint32_t slot, int level, uint32_t& width) {
  if (level == 0) { width = g.mMax0; return g.adj0 + (size_t)slot * g.mMax0; }
  width = g.mMax;
  return g.adjU + ((size_t)g.upper_off[slot] + (uint32_t)(level - 1)) * g.mMax;
}

__device__ __forceinline__ uint32_t vis_hash(uint32_t slot, uint32_t mask) { return (slot * 2654435761u) >> 7 & mask; }

// test-and-set; true = newly inserted (was unvisited)
__device__ __forceinline__ bool vis_insert(uint32_t* vis, uint32_t mask, uint32_t slot) {
  uint32_t h = vis_hash(slot, mask);
  for (uint32_t probes = 0; probes <= mask; probes++) {
    uint32_t old = atomicCAS(&vis[h], VIS_EMPTY, slot);
    if (old == VIS_EMPTY) return true;
    if (old == slot) return false;
    h = (h + 1) & mask;
  }
  return false;  // table full: cannot happen (it is reset at 75 % load); treated as "visited"
}

__device__ __forceinline__ void vis_clear(WaveCtx& w, int lane) {
  for (uint32_t i = lane; i < w.hcap; i += 64) w.vis[i] = VIS_EMPTY;
}

// Burst depth of the row loads (exact.hpp: pair_distance), per kernel profile.  What buys HBM bandwidth is bytes in flight per CU =
// waves x bursts — but a lane pair completes a 128-byte line with FOUR consecutive loads, so every burst keeps U/4 lines per row x
// up to 32 rows x the resident waves alive in the CU's 32 KB vector L1, and 32 translations per load in its UTCL1.  Past that the
// lines are evicted between their quarters and re-fetched from L2.  tools/micro/rowscan.hip measures this very code over random rows
// of a 15 / 31 GB table without the walk around it (profiles/r03_rowscan_*.jsonl, TB/s at 4 waves per CU, 32 / 20 active pairs):
//   f32 rows  U = 8: 6.44 / 5.73   12: 6.35 / 6.38   16: 4.79 / 6.23   24: 3.48 / 3.69      (8 waves per CU: at most 5.3)
//   f16 rows  U = 12: 5.53 / 3.91  16: 6.02 / 4.46   24: 6.20 / 5.44   32: 6.17 / 5.87     (8 waves per CU, U = 24: 6.06 / 6.01)
// and on the 10 M x 768 f32 walk itself (ms per 10 k queries, ef 128 / 256 / 1024): U = 16 (round 2; 24 in the HBM-visited kernel)
// 22.81 / 47.37 / 174.6, U = 12 22.06 / 43.89 / 165.6, U = 8 23.75 / 46.83 / 176.2.  Hence 12 for f32 rows in every profile.
// 2-byte rows: the LDS-visited kernel runs 1 wave per SIMD and keeps a WHOLE 768-dim row in flight (U = 48, 13.15 -> 11.88 ms at
// 2 M x 768); the HBM-visited kernels run 2 waves per SIMD and must stay under 256 registers (U = 24).
enum { PROF_BUILD = 0, PROF_SEARCH_LDS = 1, PROF_SEARCH_HBM = 2, PROF_SEARCH_HBM_DEEP = 4 };
#ifndef COLTT_U_F32      // measurement knob: burst depth of f32 rows, all profiles
#define COLTT_U_F32 12
#endif
template <int QUANT, int PROFILE> __device__ __forceinline__ constexpr int burst_depth() {
  if (QUANT == Q_NONE) return COLTT_U_F32;
  if (PROFILE == PROF_SEARCH_HBM_DEEP) return 48;   // one wave per SIMD: a whole 2-byte row in flight
  return PROFILE == PROF_SEARCH_LDS ? 48 : 24;
}
// R8: GraphView::rows is line-transposed (rows8.hpp; the index keeps ONE row array in that layout) — same values, same order, same bits
// the query given explicitly (natural-order f32 in LDS + its ||q||^2): the traversal's query, or a stored row standing in for one
// (the diverse neighbour selection compares stored rows with each other, hnsw.hip)
template <int METRIC, int QUANT, int PROFILE, bool R8 = false>
__device__ __forceinline__ float eval_pair_q(const GraphView& g, const float* __restrict__ qs, float qnorm, uint32_t slot, int half) {
  float rn = 0.f;
  if constexpr (METRIC == M_COS) rn = g.norms[slot];
  constexpr int U = burst_depth<QUANT, PROFILE>();
  if constexpr (R8) return pair_distance_r8<METRIC, QUANT, U / (QUANT == Q_NONE ? 4 : 8)>(g.rows + (size_t)slot * g.stride, qs, g.dim, qnorm, rn, half);
  else return pair_distance<METRIC, QUANT, U>(g.rows + (size_t)slot * g.stride, qs, g.dim, qnorm, rn, half);
}
template <int METRIC, int QUANT, int PROFILE, bool R8 = false>
__device__ __forceinline__ float eval_pair(const GraphView& g, const WaveCtx& w, uint32_t slot, int half) {
  return eval_pair_q<METRIC, QUANT, PROFILE, R8>(g, w.qs, w.qnorm, slot, half);
}

// greedyClosestNeighbor (hnsw.go:320-343) on `level`: move to the strict minimum until no neighbour improves.
template <int METRIC, int QUANT, int PROFILE, bool R8 = false>
__device__ __forceinline__ void greedy_level(const GraphView& g, WaveCtx& w, uint32_t& cur, float& curd, int level,
                                             int lane_in) {
  for (uint32_t hops = 0;; hops++) {
    const int lane = opaque_lane(lane_in);
    const int half = lane & 1, p = lane >> 1;
    if (hops > (1u << 20)) { w.err |= 4u; break; }
    uint32_t width;
    const uint32_t* row = adj_row(g, cur, level, width);
    unsigned long long best = ~0ull;
    uint32_t best_slot = NBR_NONE;
    for (uint32_t c0 = 0; c0 < width; c0 += 32) {
      uint32_t idx = c0 + p;
      uint32_t nb = idx < width ? row[idx] : NBR_NONE;
      bool valid = nb != NBR_NONE && !is_deleted(g, nb);
      float d = 0.f;
      if (valid) d = eval_pair<METRIC, QUANT, PROFILE, R8>(g, w, nb, half);
      w.n_dist += __popcll(__ballot(valid && half == 0));
      unsigned long long key = valid ? (((unsigned long long)__float_as_uint(d) << 32) | idx) : ~0ull;
      unsigned long long km = wave_min_u64(key);
      if (km < best) {
        best = km;
        int src = (int)(((uint32_t)km - c0) * 2);  // lane of the winning pair in this chunk
        best_slot = (uint32_t)__builtin_amdgcn_readlane((int)nb, src);
      }
    }
    w.n_hops++;
    float bd = __uint_as_float((uint32_t)(best >> 32));
    if (best != ~0ull && bd < curd) { cur = best_slot; curd = bd; }
    else break;
  }
}

// Rebuild the visited set from the current result set (bounded-memory fallback; see search_level).
__device__ __forceinline__ void vis_reset(WaveCtx& w, const unsigned long long* res, uint32_t len, int lane) {
  vis_clear(w, lane);
  wave_sync();
  for (uint32_t i = lane; i < len; i += 64) vis_insert(w.vis, w.hcap_mask, (uint32_t)res[i] >> 1);
  wave_sync();
}

// searchLevel (hnsw.go:345-389).  On return w.res[buf][0..len) holds the result set ascending by (d, slot).
// The wave must be the only one in its workgroup (wave_sync is a wave-level LDS fence).
template <int METRIC, int QUANT, bool VISG, int PROFILE, bool R8 = false>
__device__ __forceinline__ void search_level(const GraphView& g, WaveCtx& w, uint32_t ep, float epd, uint32_t ef,
                                             int level, int lane_in, uint32_t& out_len, int& out_buf) {
  int lane = lane_in;
  if constexpr (VISG) {
    if (++w.epoch > 255u) {  // 8-bit epoch wrapped: wipe the region (once per 255 traversals)
      for (size_t i = (size_t)lane * 16; i < w.vis_bytes; i += 64 * 16) *reinterpret_cast<u32x4v*>(w.visg + i) = u32x4v{0, 0, 0, 0};
      __threadfence();
      w.epoch = 1;
    }
  } else vis_clear(w, lane);
  int buf = 0;
  if (lane == 0) w.res0[0] = ((unsigned long long)__float_as_uint(epd) << 32) | ((unsigned long long)ep << 1);
  wave_sync();
  if (lane == 0) {
    if constexpr (VISG) __hip_atomic_store(w.visg + ep, (uint8_t)w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else vis_insert(w.vis, w.hcap_mask, ep);
  }
  uint32_t len = 1, vis_count = 1;
  uint32_t scan_lo = 0;   // every member before this index is expanded (pop scans start at its 64-entry chunk)
  bool had_reset = false;
  uint32_t pre_slot = NBR_NONE, pre_nb = NBR_NONE;
  wave_sync();
  for (uint32_t iters = 0;; iters++) {
    if (iters > (1u << 22)) { w.err |= 2u; break; }
    lane = opaque_lane(lane_in);
    const int half = lane & 1, p = lane >> 1;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    unsigned long long* res = w.res0 + (size_t)buf * w.ef_pad;
    // ---- pop: the smallest unexpanded member (cj = the unexpanded member after it, see the adjacency prefetch)
    int ci = -1, cj = -1;
    for (uint32_t base = scan_lo & ~63u; base < len; base += 64) {
      uint32_t i = base + lane;
      bool un = i < len && !(res[i] & 1ull);
      unsigned long long m = __ballot(un);
      if (m) {
        ci = (int)base + __builtin_ctzll(m);
        m &= m - 1;
        if (m) cj = (int)base + __builtin_ctzll(m);
        break;
      }
    }
    if (ci < 0) break;
    scan_lo = (uint32_t)ci + 1;
    COLTT_PT(w, 0)  // pop scan
    unsigned long long ce = res[ci];
    const unsigned long long runner_key = cj >= 0 ? (res[cj] & ~1ull) : ~0ull;
    float lower_bound = __uint_as_float((uint32_t)(res[len - 1] >> 32));
    wave_sync();
    if (lane == 0) res[ci] = ce | 1ull;
    const uint32_t cslot = (uint32_t)ce >> 1;
    uint32_t free_slots = ef - len;  // len <= ef
    w.n_exp++;
    if constexpr (!VISG) {
      if (vis_count + 64 > (w.hcap >> 2) * 3) {  // bounded visited set: forget everything but the result set
        wave_sync();
        vis_reset(w, res, len, lane);
        vis_count = len; had_reset = true; w.n_resets++;
      }
    }
    wave_sync();
    uint32_t width;
    const uint32_t* row = adj_row(g, cslot, level, width);
    // Adjacency prefetch (level 0).  The NEXT pop is known as soon as this expansion's distances are: it is the smaller of
    // the runner-up (the unexpanded member after the one popped now) and the best vertex admitted now.  Its adjacency row
    // is requested right then and flies during the merge and the next pop scan, which takes the dependent HBM round trip
    // (12 % of the walk for f16 rows) out of the chain.  Rows are frozen during a search: nothing else changes.
    const bool use_pre = level == 0 && pre_slot == cslot;
    const uint32_t pre_now = pre_nb;
    pre_slot = NBR_NONE;
    unsigned long long best_new = ~0ull;
    // Enabled for 2-/1-byte rows only: with f32 rows it measured SLOWER at 10 M x 768 on one box (search 22.4 -> 23.1 ms per
    // 10 k queries, build 39.8 -> 47.8 s; neutral at 2 M) — the walk is then at the memory system's limit and the early
    // request only reorders traffic.  -DCOLTT_NO_ADJ_PREFETCH turns it off everywhere.
#ifdef COLTT_NO_ADJ_PREFETCH
#define COLTT_PREFETCH_LEVEL(L) false
#else
#define COLTT_PREFETCH_LEVEL(L) (QUANT != Q_NONE && (L) == 0)
#endif
#define COLTT_PREFETCH_NEXT()                                                                        \
    if (COLTT_PREFETCH_LEVEL(level)) {                                                               \
      const unsigned long long nk_ = runner_key < best_new ? runner_key : best_new;                  \
      if (nk_ != ~0ull) {                                                                            \
        pre_slot = (uint32_t)nk_ >> 1;                                                               \
        pre_nb = (uint32_t)p < g.mMax0 ? g.adj0[(size_t)pre_slot * g.mMax0 + p] : NBR_NONE;          \
      }                                                                                              \
    }
    for (uint32_t c0 = 0; c0 < width; c0 += 32) {
      res = w.res0 + (size_t)buf * w.ef_pad;
      uint32_t idx = c0 + p;
      uint32_t nb = idx < width ? ((use_pre && c0 == 0) ? pre_now : row[idx]) : NBR_NONE;
      bool valid = nb != NBR_NONE && !is_deleted(g, nb);
#ifdef COLTT_PHASE_TIMING
      if (__ballot(valid) == 0xdeadbeefcafeull) w.err |= 64u;  // forces the adjacency values to have arrived
#endif
      COLTT_PT(w, 1)  // adjacency row
      int fresh_i = 0;
      if (valid && half == 0) {
        if constexpr (VISG) {
          // No two lanes hold the same slot (a row lists a neighbour once), so load + store is a race-free test-and-set.
          // Agent-scope atomics: served by L2, never by a stale L1 line; the store is complete (vmcnt) before the next
          // chunk's loads are issued because the distance loads issued after it are waited for first.
          const uint8_t v = __hip_atomic_load(w.visg + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          fresh_i = v != (uint8_t)w.epoch ? 1 : 0;
          if (fresh_i) __hip_atomic_store(w.visg + nb, (uint8_t)w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else fresh_i = vis_insert(w.vis, w.hcap_mask, nb) ? 1 : 0;
      }
      fresh_i = __builtin_amdgcn_mov_dpp(fresh_i, 0xA0, 0xf, 0xf, true);  // even lane's verdict to its pair: quad_perm [0,0,2,2]
      bool fresh = fresh_i != 0;
      if (had_reset) {
        // a forgotten vertex that is still in the result set must not be admitted twice
        unsigned long long fm = __ballot(fresh && half == 0);
        while (fm) {
          int j = __builtin_ctzll(fm); fm &= fm - 1;
          uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)nb, j);
          bool hit = false;
          for (uint32_t base = 0; base < len; base += 64) {
            uint32_t i = base + lane;
            hit |= (i < len && ((uint32_t)res[i] >> 1) == sj);
          }
          if (__ballot(hit) && (lane >> 1) == (j >> 1)) fresh = false;
        }
      }
      unsigned long long E = __ballot(fresh && half == 0);
      uint32_t nfresh = __popcll(E);
      COLTT_PT(w, 2)  // visited test-and-set
      const bool last_chunk = c0 + 32 >= width;
      if (nfresh == 0) { if (last_chunk) { COLTT_PREFETCH_NEXT() } continue; }
      vis_count += nfresh; w.n_dist += nfresh;
      float d = 0.f;
      if (fresh) d = eval_pair<METRIC, QUANT, PROFILE, R8>(g, w, nb, half);
      uint32_t rank = __popcll(E & lt_mask);
      bool adm = fresh && half == 0 && (rank < free_slots || d < lower_bound);
#ifdef COLTT_PHASE_TIMING
      if (__ballot(adm) == 0xdeadbeefcafeull) w.err |= 64u;  // forces the distances
#endif
      COLTT_PT(w, 3)  // row reads + distances
      free_slots = free_slots > nfresh ? free_slots - nfresh : 0;
      unsigned long long A = __ballot(adm);
      uint32_t m = __popcll(A);
      unsigned long long mykey = adm ? (((unsigned long long)__float_as_uint(d) << 32) | ((unsigned long long)nb << 1)) : ~0ull;
      uint32_t myrank = 0;  // rank of my key among the admitted ones (readlane broadcasts: no LDS round trips)
      {
        unsigned long long am = A;
        while (am) {
          int j = __builtin_ctzll(am); am &= am - 1;
          unsigned long long kj = readlane_u64(mykey, j);
          myrank += (kj < mykey) ? 1u : 0u;
        }
      }
      if (m) {  // the smallest admitted key is the one of rank 0
        const unsigned long long z = __ballot(adm && myrank == 0);
        const unsigned long long mn = readlane_u64(mykey, __builtin_ctzll(z));
        best_new = mn < best_new ? mn : best_new;
      }
      if (last_chunk) { COLTT_PREFETCH_NEXT() }
      if (m == 0) continue;
      // ---- merge the m admitted (d, slot) into the sorted result set, keep the ef smallest.  IN PLACE, from the tail down to
      // the chunk of the smallest new key: members before it do not move (at ef 1024 most admissions land near the tail, so
      // most of the set is never touched).  A member at index i moves to i + #{new keys that sort before it}; new key j sorts
      // before it iff its insertion point pos_j <= i (keys are distinct), so the shift is counted on 32-bit positions.
      // Descending chunk order makes the move safe: targets lie at most m <= 32 entries above, i.e. in entries already moved.
      uint32_t mypos = 0xffffffffu;
      if (adm) {  // lower_bound over the sorted result set
        uint32_t lo = 0, hi = len;
        while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (res[mid] < mykey) lo = mid + 1; else hi = mid; }
        mypos = lo;
      }
      const uint32_t minpos = (uint32_t)__builtin_amdgcn_readlane((int)mypos, __builtin_ctzll(__ballot(adm && myrank == 0)));
      scan_lo = minpos < scan_lo ? minpos : scan_lo;   // the new keys are unexpanded
      wave_sync();
      if (len) {
        for (int base = (int)((len - 1) & ~63u); base >= (int)(minpos & ~63u); base -= 64) {
          const uint32_t i = (uint32_t)base + lane;
          const unsigned long long e = i < len ? res[i] : ~0ull;
          uint32_t shift = 0;
          unsigned long long am = A;
          while (am) {
            int j = __builtin_ctzll(am); am &= am - 1;
            const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)mypos, j);
            shift += (pj <= i) ? 1u : 0u;
          }
          wave_sync();   // every lane holds its member before any lane overwrites one
          const uint32_t np = i + shift;
          if (i < len && shift && np < ef) res[np] = e;
        }
      }
      wave_sync();
      {
        uint32_t np = mypos + myrank;
        if (adm && np < ef) res[np] = mykey;
      }
      len = len + m < ef ? len + m : ef;
      wave_sync();
      COLTT_PT(w, 4)  // merge
    }
  }
#undef COLTT_PREFETCH_NEXT
  out_len = len;
  out_buf = buf;
}


}  // namespace dev
}  // namespace coltt
