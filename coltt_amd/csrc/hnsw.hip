// hnsw.hip — core/vectorindex.Hnsw on the GPU (core/vectorindex/hnsw.go).
//
// HBM layout (one index = one collection shard on one GPU), everything addressed by dense slot (u32 =
// insertion index, never reused — the oracle uses the same numbering as its canonical order):
//   rows      [cap][stride]     stored vectors (normalised for cosine; 4/2/1-byte codes)
//   norms     [cap] f32         ||row||^2 in AVX order
//   ids       [cap] u64         slot -> id                       (absent in dense-id mode)
//   adj0      [cap][mMax0] u32  level-0 neighbours, ascending slot, padded 0xffffffff   (+ adj0_d f32 distances)
//   upper_off [cap] u32         first upper row of the slot (a vertex of level L owns L consecutive rows)
//   adjU      [ucap][mMax] u32  levels >= 1                                                (+ adjU_d)
//   del_bits  [cap/32] u32      tombstones (hnswVertex.deleted)
#include <algorithm>
#include <atomic>
#include <cmath>

#include "common.hpp"
#include "exact.hpp"
#include "hnsw_dev.hpp"
#include "hnsw_walk2.hpp"
#include "hnsw_lat.hpp"
#include "hnsw_pq.hpp"
#include "prep.hpp"

#include "hnsw_kernels.hpp"   // the search kernels (one-wave walk, large-ef walk, latency kernel, walk over PQ codes + re-rank)

using namespace coltt;
using namespace coltt::dev;
using namespace coltt::kern;

namespace {

// ---------------------------------------------------------------------------------------------------
// Graph construction (Hnsw.Insert, hnsw.go:104-167) for a batch of new vertices against the frozen graph.
// Phase A (this kernel, one wave per new vertex): greedy descent above the vertex level (:126-130), then per
// level searchLevel(efConstruction) + the m nearest (:132-140); the vertex's own rows are written here
// (nobody can reach a new vertex yet: it has no in-edges) and one link request per selected neighbour is
// queued: req[r] = {row id of the neighbour's level-l row, new slot, distance}, chained per target row through
// head[]/next[] with atomicExch.
// ---------------------------------------------------------------------------------------------------
struct BuildReq { uint32_t rid, from; float d; uint32_t next; };

template <int METRIC, int QUANT, bool VISG, bool R8 = false>
__global__ __launch_bounds__(64) void hnsw_build_search_kernel(GraphView g, int32_t entry, int32_t entry_level, uint32_t base,
                                                              uint32_t count, const int32_t* __restrict__ levels, uint32_t M,
                                                              uint32_t efc, uint32_t ef_pad, uint32_t hcap, uint64_t cap_slots,
                                                              uint32_t* __restrict__ counter, uint32_t* __restrict__ req_count,
                                                              BuildReq* __restrict__ req, uint32_t* __restrict__ head,
                                                              unsigned long long* __restrict__ stats, uint8_t* __restrict__ visg,
                                                              size_t vis_stride, uint32_t* __restrict__ vis_epoch,
                                                              uint32_t diverse /* 0 | 1 | 3 = keepPruned */, uint32_t ext_off) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  WaveCtx w;
  size_t off = ((size_t)g.dim * 4 + 15) & ~(size_t)15;
  // diverse selection (algo 2): a second query buffer (the candidate's stored row), the chosen and the rejected keys — behind the walk's LDS
  float* const qs2 = reinterpret_cast<float*>(smem + ext_off);
  unsigned long long* const sel = reinterpret_cast<unsigned long long*>(smem + ext_off + off);
  unsigned long long* const prn = sel + 64;
  w.qs = reinterpret_cast<float*>(smem);
  w.res0 = reinterpret_cast<unsigned long long*>(smem + off);
  w.vis = reinterpret_cast<uint32_t*>(w.res0 + (size_t)ef_pad);
  w.ef_pad = ef_pad; w.hcap = hcap; w.hcap_mask = hcap - 1;
  w.visg = nullptr; w.vis_bytes = 0; w.epoch = 0;
  if constexpr (VISG) { w.visg = visg + (size_t)blockIdx.x * vis_stride; w.vis_bytes = vis_stride; w.epoch = vis_epoch[blockIdx.x]; }
  for (;;) {
    const uint32_t bt = atomicAdd(counter, lane == 0 ? 1u : 0u);  // branch-free, see hnsw_search_kernel
    const uint32_t bi = (uint32_t)__shfl((int)bt, 0, 64);
    if (bi >= count) break;
    const uint32_t vi = base + bi;
    const int lv = levels[bi];
    w.n_dist = w.n_exp = w.n_hops = w.n_resets = 0; w.err = 0;
    wave_sync();
    {  // the query is the vertex's own stored row, decoded to f32
      const uint8_t* row = g.rows + (size_t)vi * g.stride;
      for (int e = lane; e < g.dim; e += 64) {
        if constexpr (R8) w.qs[e] = load1<QUANT>(row, r8_index<QUANT>(e)); else w.qs[e] = load1<QUANT>(row, e);
      }
    }
    w.qnorm = METRIC == M_COS ? g.norms[vi] : 0.f;
    wave_sync();
    uint32_t cur = (uint32_t)entry;
    float curd = eval_pair<METRIC, QUANT, PROF_BUILD, R8>(g, w, cur, lane & 1);
    curd = __shfl(curd, 0, 64);
    w.n_dist += 1;
    for (int l = entry_level; l > lv; l--) greedy_level<METRIC, QUANT, PROF_BUILD, R8>(g, w, cur, curd, l, lane);
    for (int l = entry_level < lv ? entry_level : lv; l >= 0; l--) {
      uint32_t len; int buf;
      w.n_dist += 1;
      search_level<METRIC, QUANT, VISG, PROF_BUILD, R8>(g, w, cur, curd, efc, l, lane, len, buf);
      const unsigned long long* res = w.res0 + (size_t)buf * ef_pad;
      uint32_t m = len < M ? len : M;
      if (diverse) {
        // algo 2 — NOT reference behaviour; the oracle's select_diverse (coltt_oracle.cpp) is the definition: walk the result set in
        // ascending (distance, slot) order while fewer than M are chosen; a candidate is chosen iff no already chosen r has
        // D(candidate as the query, r) < d(new vertex, candidate); every chosen r is evaluated (lane pair per r, side by side).
        uint32_t ns = 0, np = 0;
        for (uint32_t ci = 0; ci < len && ns < M; ci++) {
          const int ln = opaque_lane(lane);
          const int half = ln & 1, p = ln >> 1;
          const unsigned long long ce = res[ci];
          const uint32_t es = (uint32_t)ce >> 1;
          const float ed = __uint_as_float((uint32_t)(ce >> 32));
          bool bad = false;
          if (ns) {
            const uint8_t* crow = g.rows + (size_t)es * g.stride;
            for (int t = ln; t < g.dim; t += 64) {
              if constexpr (R8) qs2[t] = load1<QUANT>(crow, r8_index<QUANT>(t)); else qs2[t] = load1<QUANT>(crow, t);
            }
            const float qn2 = METRIC == M_COS ? g.norms[es] : 0.f;
            wave_sync();
            for (uint32_t c0 = 0; c0 < ns; c0 += 32) {
              const uint32_t idx = c0 + (uint32_t)p;
              const bool valid = idx < ns;
              float dd = 0.f;
              if (valid) dd = eval_pair_q<METRIC, QUANT, PROF_BUILD, R8>(g, qs2, qn2, (uint32_t)sel[idx] >> 1, half);
              bad = bad || (valid && half == 0 && dd < ed);
            }
            w.n_dist += ns;
          }
          const bool rejected = __ballot(bad) != 0ull;
          wave_sync();   // every lane is done with qs2 / sel before either is written again
          if (!rejected) { if (ln == 0) sel[ns] = ce; ns++; }
          else if (np < 64u) { if (ln == 0) prn[np] = ce; np++; }   // at most M <= 64 rejected ones can ever be re-added
          wave_sync();
        }
        if (diverse & 2u) {   // heuristicKeepPruned: the rejected, nearest first, fill the row up to M
          for (uint32_t i = 0; i < np && ns < M; i++) { if (lane == 0) sel[ns] = prn[i]; ns++; }
          wave_sync();
        }
        m = ns;
        res = sel;
      }
      // the m nearest (or the m chosen), re-ordered by slot (canonical row order)
      unsigned long long e = (uint32_t)lane < m ? res[lane] : ~0ull;
      uint32_t myslot = (uint32_t)e >> 1;
      float myd = __uint_as_float((uint32_t)(e >> 32));
      uint32_t rank = 0;
      for (uint32_t j = 0; j < m; j++) {
        uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)myslot, (int)j);
        rank += sj < myslot ? 1u : 0u;
      }
      uint32_t width;
      uint32_t* row = const_cast<uint32_t*>(adj_row(g, vi, l, width));
      float* drow = (l == 0 ? g.adj0_d + (size_t)vi * g.mMax0 : g.adjU_d + ((size_t)g.upper_off[vi] + (uint32_t)(l - 1)) * g.mMax);
      const uint32_t rt = atomicAdd(req_count, lane == 0 ? m : 0u);  // branch-free, see hnsw_search_kernel
      const uint32_t r0 = (uint32_t)__shfl((int)rt, 0, 64);
      if ((uint32_t)lane < m) {
        row[rank] = myslot; drow[rank] = myd;
        if (l == 0 && g.adj0_n) g.adj0_n[(size_t)vi * g.mMax0 + rank] = g.norms[myslot];
        uint32_t rid = l == 0 ? myslot : (uint32_t)cap_slots + g.upper_off[myslot] + (uint32_t)(l - 1);
        uint32_t r = r0 + lane;
        uint32_t old = atomicExch(&head[rid], r);
        req[r].rid = rid; req[r].from = vi; req[r].d = myd; req[r].next = old;
      }
      // next level starts from the nearest result (hnsw.go:145 `entrypoint = neighbor`, last popped = nearest; the nearest candidate is
      // always chosen first by the diverse selection, so sel[0] == the result set's first member)
      unsigned long long e0 = res[0];
      cur = (uint32_t)e0 >> 1; curd = __uint_as_float((uint32_t)(e0 >> 32));
      wave_sync();
    }
    if (lane == 0) {
      atomicAdd(&stats[0], (unsigned long long)w.n_dist);
      atomicAdd(&stats[1], (unsigned long long)w.n_exp);
      atomicAdd(&stats[2], (unsigned long long)w.n_hops);
      atomicAdd(&stats[3], (unsigned long long)w.n_resets);
      if (w.err) atomicOr(&stats[4], (unsigned long long)w.err);
    }
  }
  if constexpr (VISG) { if (lane == 0) vis_epoch[blockIdx.x] = w.epoch; }
}

// Phase B: apply the queued links to the neighbours' rows (hnsw.go:151-158: neighbor.addEdge + pruneNeighbors when
// the row exceeds mMax).  Exactly one request per touched row arrived first (next == NONE): its thread owns the
// row.  Closed form of "add one by one, prune on overflow" (order-independent): if existing + added <= width the
// row is the union; otherwise the deleted neighbours are dropped (pruneNeighbors skips them, hnsw.go:454-456) and
// the `width` nearest by (stored distance, slot) are kept (selectNeighbors, :391-397).  Rows stay sorted by slot.
constexpr int LINK_W = 64;    // widest row the builder supports
constexpr int LINK_CH = 64;   // requests merged per round
__global__ void hnsw_link_kernel(GraphView g, uint64_t cap_slots, const BuildReq* __restrict__ req, uint32_t n_req,
                                 uint32_t* __restrict__ head) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_req) return;
  if (req[r].next != NBR_NONE) return;
  const uint32_t rid = req[r].rid;
  const bool lvl0 = rid < cap_slots;
  const uint32_t W = lvl0 ? g.mMax0 : g.mMax;
  uint32_t* row = lvl0 ? g.adj0 + (size_t)rid * g.mMax0 : g.adjU + (size_t)(rid - cap_slots) * g.mMax;
  float* drow = lvl0 ? g.adj0_d + (size_t)rid * g.mMax0 : g.adjU_d + (size_t)(rid - cap_slots) * g.mMax;
  uint32_t s[LINK_W + LINK_CH]; float d[LINK_W + LINK_CH];
  uint32_t n = 0;
  for (; n < W && row[n] != NBR_NONE; n++) { s[n] = row[n]; d[n] = drow[n]; }
  uint32_t cur = head[rid];
  head[rid] = NBR_NONE;
  while (cur != NBR_NONE) {
    uint32_t take = 0;
    while (cur != NBR_NONE && take < LINK_CH) { s[n + take] = req[cur].from; d[n + take] = req[cur].d; cur = req[cur].next; take++; }
    uint32_t tot = n + take;
    if (tot > W) {
      uint32_t m = 0;  // drop tombstoned neighbours
      for (uint32_t i = 0; i < tot; i++) if (!is_deleted(g, s[i])) { s[m] = s[i]; d[m] = d[i]; m++; }
      // insertion sort by (d, slot)
      for (uint32_t i = 1; i < m; i++) {
        uint32_t si = s[i]; float di = d[i]; uint32_t j = i;
        while (j > 0 && (d[j - 1] > di || (d[j - 1] == di && s[j - 1] > si))) { s[j] = s[j - 1]; d[j] = d[j - 1]; j--; }
        s[j] = si; d[j] = di;
      }
      n = m < W ? m : W;
    } else n = tot;
  }
  for (uint32_t i = 1; i < n; i++) {  // back to canonical slot order
    uint32_t si = s[i]; float di = d[i]; uint32_t j = i;
    while (j > 0 && s[j - 1] > si) { s[j] = s[j - 1]; d[j] = d[j - 1]; j--; }
    s[j] = si; d[j] = di;
  }
  for (uint32_t i = 0; i < W; i++) { row[i] = i < n ? s[i] : NBR_NONE; drow[i] = i < n ? d[i] : 0.f; }
  if (lvl0 && g.adj0_n) {
    float* nrow = g.adj0_n + (size_t)rid * g.mMax0;
    for (uint32_t i = 0; i < W; i++) nrow[i] = i < n ? g.norms[s[i]] : 0.f;
  }
}

// Phase B of the diverse mode (algo 2 — NOT reference behaviour; definition: the oracle's select_diverse + the batch rule of
// hnsw_insert_batch): a row receives ALL the batch's links and is pruned ONCE if it overflows — drop the tombstoned neighbours; if the
// live ones still do not fit, walk them in ascending (stored distance, slot) order and choose a candidate iff no already chosen r has
// D(candidate's stored row as the query, r's stored row) < the candidate's stored distance, until W are chosen; keepPruned re-adds the
// rejected, nearest first.  One WAVE per touched row (the owner request's wave; the others leave at once): candidates in LDS, the
// candidate under test decoded into LDS as a query, the chosen rows evaluated side by side by lane pairs — the builder's own distance code.
constexpr uint32_t LINKD_CAND = 1024;   // candidates per row (existing + the batch's links); more trips stats[4] |= 16 (lower the batch size)
template <int METRIC, int QUANT, bool R8>
__global__ __launch_bounds__(64) void hnsw_link_diverse_kernel(GraphView g, uint64_t cap_slots, const BuildReq* __restrict__ req, uint32_t n_req,
                                                              uint32_t* __restrict__ head, uint32_t keep_pruned,
                                                              unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t r = blockIdx.x;
  if (r >= n_req) return;
  if (req[r].next != NBR_NONE) return;
  const int lane_in = threadIdx.x;
  const size_t off = ((size_t)g.dim * 4 + 15) & ~(size_t)15;
  float* const qs2 = reinterpret_cast<float*>(smem);
  unsigned long long* const cand = reinterpret_cast<unsigned long long*>(smem + off);   // [LINKD_CAND] as gathered
  unsigned long long* const sorted = cand + LINKD_CAND;                                  // [LINKD_CAND] ascending (d bits, slot)
  unsigned long long* const sel = sorted + LINKD_CAND;                                   // [64] chosen
  unsigned long long* const prn = sel + 64;                                              // [64] rejected (first W of them)
  const uint32_t rid = req[r].rid;
  const bool lvl0 = rid < cap_slots;
  const uint32_t W = lvl0 ? g.mMax0 : g.mMax;
  uint32_t* row = lvl0 ? g.adj0 + (size_t)rid * g.mMax0 : g.adjU + (size_t)(rid - cap_slots) * g.mMax;
  float* drow = lvl0 ? g.adj0_d + (size_t)rid * g.mMax0 : g.adjU_d + (size_t)(rid - cap_slots) * g.mMax;
  int lane = lane_in;
  // ---- gather: the row's entries, then the chained requests (the chain is short: one lane walks it)
  uint32_t n = 0;
  {
    const uint32_t sl = (uint32_t)lane < W ? row[lane] : NBR_NONE;
    const float dl = (uint32_t)lane < W ? drow[lane] : 0.f;
    const unsigned long long have = __ballot(sl != NBR_NONE);
    if (sl != NBR_NONE) cand[__popcll(have & ((1ull << lane) - 1ull))] = ((unsigned long long)__float_as_uint(dl) << 32) | sl;
    n = (uint32_t)__popcll(have);
  }
  uint32_t tot = n;
  if (lane == 0) {
    uint32_t cur = head[rid];
    head[rid] = NBR_NONE;
    while (cur != NBR_NONE) {
      if (tot < LINKD_CAND) cand[tot] = ((unsigned long long)__float_as_uint(req[cur].d) << 32) | req[cur].from;
      tot++;
      cur = req[cur].next;
    }
  }
  tot = (uint32_t)__shfl((int)tot, 0, 64);
  if (tot > LINKD_CAND) { if (lane == 0) atomicOr(&stats[4], 16ull); return; }   // the host fails the batch loudly
  wave_sync();
  uint32_t ns;            // entries of the new row, in sel[]
  if (tot <= W) {
    for (uint32_t i = lane; i < tot; i += 64) sel[i] = cand[i];
    ns = tot;
  } else {
    // drop the tombstoned neighbours (pruneNeighbors skips them, hnsw.go:454-456), rank-sort the live ones by key
    uint32_t live = 0;
    for (uint32_t b = 0; b < tot; b += 64) {
      const uint32_t i = b + lane;
      const unsigned long long k = i < tot ? cand[i] : ~0ull;
      const bool ok = i < tot && !is_deleted(g, (uint32_t)k);
      const unsigned long long mk = __ballot(ok);
      wave_sync();   // chunk b has been read by every lane before its front part is overwritten (live <= b)
      if (ok) cand[live + __popcll(mk & ((1ull << lane) - 1ull))] = k;
      live += (uint32_t)__popcll(mk);
      wave_sync();
    }
    for (uint32_t i = lane; i < live; i += 64) {
      const unsigned long long k = cand[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < live; j++) rank += cand[j] < k ? 1u : 0u;   // keys are distinct (a slot appears once in a row)
      sorted[rank] = k;
    }
    wave_sync();
    if (live <= W) {
      for (uint32_t i = lane; i < live; i += 64) sel[i] = sorted[i];
      ns = live;
    } else {
      uint32_t nsel = 0, np = 0;
      unsigned long long n_eval = 0;
      for (uint32_t ci = 0; ci < live && nsel < W; ci++) {
        lane = opaque_lane(lane_in);
        const int half = lane & 1, p = lane >> 1;
        const unsigned long long ce = sorted[ci];
        const uint32_t es = (uint32_t)ce;
        const float ed = __uint_as_float((uint32_t)(ce >> 32));
        bool bad = false;
        if (nsel) {
          const uint8_t* crow = g.rows + (size_t)es * g.stride;
          for (int t = lane; t < g.dim; t += 64) {
            if constexpr (R8) qs2[t] = load1<QUANT>(crow, r8_index<QUANT>(t)); else qs2[t] = load1<QUANT>(crow, t);
          }
          const float qn2 = METRIC == M_COS ? g.norms[es] : 0.f;
          wave_sync();
          for (uint32_t c0 = 0; c0 < nsel; c0 += 32) {
            const uint32_t idx = c0 + (uint32_t)p;
            const bool valid = idx < nsel;
            float dd = 0.f;
            if (valid) dd = eval_pair_q<METRIC, QUANT, PROF_BUILD, R8>(g, qs2, qn2, (uint32_t)sel[idx], half);
            bad = bad || (valid && half == 0 && dd < ed);
          }
          n_eval += nsel;
        }
        const bool rejected = __ballot(bad) != 0ull;
        wave_sync();
        if (!rejected) { if (lane == 0) sel[nsel] = ce; nsel++; }
        else if (np < 64u) { if (lane == 0) prn[np] = ce; np++; }
        wave_sync();
      }
      if (keep_pruned) {
        for (uint32_t i = 0; i < np && nsel < W; i++) { if (lane == 0) sel[nsel] = prn[i]; nsel++; }
      }
      if (lane == 0 && n_eval) atomicAdd(&stats[0], n_eval);   // the oracle's n_dist counts these evaluations
      ns = nsel;
    }
  }
  wave_sync();
  // ---- the new row, ascending by slot (ns <= W <= 64: one entry per lane, ranked by readlane broadcasts)
  lane = opaque_lane(lane_in);
  const unsigned long long mine = (uint32_t)lane < ns ? sel[lane] : ~0ull;
  const uint32_t myslot = (uint32_t)mine;
  uint32_t rank = 0;
  for (uint32_t j = 0; j < ns; j++) {
    const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)myslot, (int)j);
    rank += sj < myslot ? 1u : 0u;
  }
  wave_sync();
  float* nrow = (lvl0 && g.adj0_n) ? g.adj0_n + (size_t)rid * g.mMax0 : nullptr;
  if ((uint32_t)lane < ns) {
    row[rank] = myslot; drow[rank] = __uint_as_float((uint32_t)(mine >> 32));
    if (nrow) nrow[rank] = g.norms[myslot];
  }
  if ((uint32_t)lane >= ns && (uint32_t)lane < W) {
    row[lane] = NBR_NONE; drow[lane] = 0.f;
    if (nrow) nrow[lane] = 0.f;
  }
}

// Before the diverse link pass: the longest request chain of the batch (+ the widest row) -> stats[5].  A row that would receive more candidates than the link
// kernel holds makes the host retry the batch in halves BEFORE anything was applied (insert_core): the index stays exactly as it was.
__global__ void hnsw_link_count_kernel(const BuildReq* __restrict__ req, uint32_t n_req, const uint32_t* __restrict__ head, uint64_t cap_slots,
                                       uint32_t mMax0, uint32_t mMax, unsigned long long* __restrict__ stats) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_req || req[r].next != NBR_NONE) return;
  const uint32_t rid = req[r].rid;
  uint32_t c = 0;
  for (uint32_t cur = head[rid]; cur != NBR_NONE; cur = req[cur].next) c++;
  atomicMax(&stats[5], (unsigned long long)(c + (rid < cap_slots ? mMax0 : mMax)));
}

// adj0_n[slot][j] = norms[adj0[slot][j]] for the level-0 rows of slots [first, first + n): after a bulk install of the topology
__global__ void adj_norms_kernel(GraphView g, uint64_t first, uint64_t n) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * g.mMax0) return;
  const uint64_t i = first * g.mMax0 + t;
  const uint32_t nb = g.adj0[i];
  g.adj0_n[i] = nb != NBR_NONE ? g.norms[nb] : 0.f;
}

__global__ void fill_u32_kernel(uint32_t* p, size_t n, uint32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void add_base_kernel(uint64_t* ids, size_t n, uint64_t base) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] += base;
}

// Per-call search context: searches hold the index lock shared and run concurrently, each on its own stream and workspaces.
struct HCtx {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBuf w_qraw, w_qeff, w_qn, w_out_ids, w_out_sc, w_out_cnt, w_misc, w_pack;
  DevBuf w_surv, w_scnt, w_keys;   // product-quantised walk: survivors (slots), their count, their exact keys
  DevBuf w_mbox;                   // latency kernel: the walking workgroups' hint mailboxes (hnsw_lat.hpp: cache-warming helper workgroups)
  PinnedBuf h_in, h_out;   // small calls: see PinnedBuf
  int init() {  // the caller has selected the index's device
    COLTT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    COLTT_HIP(hipEventCreate(&ev0));
    COLTT_HIP(hipEventCreate(&ev1));
    return COLTT_OK;
  }
  ~HCtx() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

struct Hnsw : Object {
  uint32_t dim = 0; int metric = 0, quant = 0; size_t stride = 0;
  coltt_hnsw_cfg cfg{};
  uint64_t n = 0 /*slots*/, live = 0, cap = 0, n_upper = 0, ucap = 0;
  int32_t entry = -1, entry_level = 0;
  bool any_deleted = false;
  DevBuf rows, norms, ids, adj0, adj0_d, adj0_n, upper_off, adjU, adjU_d, del_bits;
  // rows8.hpp (round 5: ONE row array).  r8: `rows` itself is stored LINE-TRANSPOSED — every 128-byte line rewritten so that its 16-byte chunk r
  // holds residue r's consecutive AVX steps — for the shapes the eight-lanes-per-row core covers (f32 / 2-byte rows whose byte length is a
  // multiple of 128, dim >= 256 unless COLTT_ROWS8=2; never with COLTT_ROWS8=0).  Every reader knows the layout: the eight-lane core reads whole
  // lines, the pair-owned core reads its four chunks of a line (exact.hpp: pair_distance_r8 — same values, same order, same bits), the builder,
  // Commit / Get / fetch and the quantiser's Encode un-permute element indices (r8_index).  Writers go through a natural-order staging block.
  bool r8 = false;
  DevBuf w_stage;   // natural-order rows of the batch being ingested (r8 indexes)
  std::atomic<uint64_t> ev8_launches{0};   // search launches whose level-0 distances came from rows8
  // hnsw_pq.hpp: a snapshot of a trained product quantiser and one row-major code per slot (derived data, maintained like rows8:
  // writers encode the slots they added before they release the exclusive lock)
  bool pq_on = false; PqShape pq_shape; uint32_t pq_row = 0 /* bytes per code row: m rounded up to 16 */; uint64_t pq_done = 0;
  DevBuf pq_cb, pq_codes, pq_stage;
  // Neighbourhood blocks of the product-quantised walk (hnsw_pq.hpp: AdcEval<.., NBR>): pq_nbr[slot][p][pq_row] = pq_codes[adj0[slot][p]] (zeros for an
  // empty position) — derived from the level-0 rows and the codes, rebuilt LAZILY by the first product-quantised search after a mutation (writers only
  // raise pq_nbr_stale under their exclusive lock; searches hold the lock shared and settle the rebuild among themselves under pq_nbr_mu).
  DevBuf pq_nbr; bool pq_nbr_ok = false; std::atomic<bool> pq_nbr_stale{true}; std::mutex pq_nbr_mu;
  bool dense = true; uint64_t dense_base = 0;
  std::unordered_map<uint64_t, uint32_t> id2slot;
  std::vector<uint64_t> h_ids;       // !dense
  std::vector<int32_t> h_levels;     // per slot
  std::vector<uint32_t> h_upper_off; // per slot
  std::vector<uint32_t> h_del;       // bitmap mirror
  hipStream_t stream = nullptr;        // mutations (exclusive lock); searches use their context's stream
  std::atomic<float> last_ms{0.f};     // kernel time of the most recently finished search call
  CtxPool<HCtx> pool;
  DevBuf w_raw, w_misc;
  DevBuf b_head, b_req, b_levels; uint64_t head_cap = 0;  // builder scratch
  // HBM visited set (hnsw_dev.hpp, VISG): vis_regions regions of vis_stride bytes; concurrent searches lease disjoint
  // contiguous runs of regions (vis_busy), the builder (exclusive lock) uses all of them.
  DevBuf w_visg, w_vepoch; uint64_t vis_stride = 0; uint32_t vis_regions = 0, vis_want = 0;
  std::mutex vis_mu; std::condition_variable vis_cv; std::vector<uint8_t> vis_busy;
  coltt_hnsw_stats build_stats{};
  ~Hnsw() override {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamDestroy(stream);
  }
  GraphView view() const {
    GraphView g;
    g.rows = rows.as<uint8_t>(); g.stride = stride; g.norms = norms.as<float>();
    g.rows8 = r8 ? rows.as<uint8_t>() : nullptr;   // the same array: which core reads it is the kernel's choice
    g.ids = dense ? nullptr : ids.as<uint64_t>();
    g.adj0 = adj0.as<uint32_t>(); g.adj0_d = adj0_d.as<float>(); g.upper_off = upper_off.as<uint32_t>();
    g.adj0_n = metric == COLTT_COSINE ? adj0_n.as<float>() : nullptr;
    g.adjU = adjU.as<uint32_t>(); g.adjU_d = adjU_d.as<float>();
    g.del_bits = any_deleted ? del_bits.as<uint32_t>() : nullptr;
    g.mMax = (uint32_t)cfg.m_max; g.mMax0 = (uint32_t)cfg.m_max0; g.dim = (int)dim;
    return g;
  }
  int reserve(uint64_t slots, uint64_t upper_rows) {
    if (slots > cap) {
      uint64_t nc = std::max<uint64_t>({slots, cap + cap / 2, 1024});
      COLTT_TRY(rows.reserve(nc * stride, true, stream));
      COLTT_TRY(norms.reserve(nc * 4, true, stream));
      if (!dense) COLTT_TRY(ids.reserve(nc * 8, true, stream));
      COLTT_TRY(adj0.reserve(nc * cfg.m_max0 * 4, true, stream));
      COLTT_TRY(adj0_d.reserve(nc * cfg.m_max0 * 4, true, stream));
      if (metric == COLTT_COSINE) COLTT_TRY(adj0_n.reserve(nc * cfg.m_max0 * 4, true, stream));
      COLTT_TRY(upper_off.reserve(nc * 4, true, stream));
      size_t old_words = (cap + 31) / 32, new_words = (nc + 31) / 32;
      COLTT_TRY(del_bits.reserve(new_words * 4, true, stream));
      if (new_words > old_words) COLTT_HIP(hipMemsetAsync(del_bits.as<uint32_t>() + old_words, 0, (new_words - old_words) * 4, stream));
      cap = nc;
    }
    if (upper_rows > ucap) {
      uint64_t nc = std::max<uint64_t>({upper_rows, ucap + ucap / 2, 256});
      COLTT_TRY(adjU.reserve(nc * cfg.m_max * 4, true, stream));
      COLTT_TRY(adjU_d.reserve(nc * cfg.m_max * 4, true, stream));
      ucap = nc;
    }
    return COLTT_OK;
  }
};

int prep_rows_any(Hnsw* x, const float* d_raw, uint64_t n, uint64_t slot_base, bool normalize) {
  if (n == 0) return COLTT_OK;
  int nrm = normalize ? 1 : 0;
  uint8_t* R = x->rows.as<uint8_t>();
  float* N = x->norms.as<float>();
  if (x->r8) {   // Normalize + Lower into a natural-order staging block, norms from it (AVX order), then the line transposition into the row array
    COLTT_TRY(x->w_stage.reserve(n * x->stride));
    uint8_t* S = x->w_stage.as<uint8_t>();
    const uint64_t chunks = (uint64_t)x->stride / 16;
    if (x->quant == COLTT_Q_NONE) {
      launch_prep_rows<Q_NONE>(x->stream, d_raw, n, (int)x->dim, nrm, nullptr, 0, S, x->stride);
      row_norms_kernel<Q_NONE><<<ceil_div(n * 2, 256), 256, 0, x->stream>>>(S, x->stride, nullptr, 0, n, (int)x->dim, N + slot_base);
      rows8_permute_kernel<Q_NONE><<<ceil_div(n * chunks, 256), 256, 0, x->stream>>>(S, R + slot_base * x->stride, x->stride, (int)x->dim, n);
    } else {
      launch_prep_rows<Q_F16>(x->stream, d_raw, n, (int)x->dim, nrm, nullptr, 0, S, x->stride);
      row_norms_kernel<Q_F16><<<ceil_div(n * 2, 256), 256, 0, x->stream>>>(S, x->stride, nullptr, 0, n, (int)x->dim, N + slot_base);
      rows8_permute_kernel<Q_F16><<<ceil_div(n * chunks, 256), 256, 0, x->stream>>>(S, R + slot_base * x->stride, x->stride, (int)x->dim, n);
    }
    COLTT_HIP(hipGetLastError());
    return COLTT_OK;
  }
#define COLTT_PREP(Q)                                                                                                  \
  do {                                                                                                                 \
    launch_prep_rows<Q>(x->stream, d_raw, n, (int)x->dim, nrm, nullptr, slot_base, R, x->stride);                     \
    row_norms_kernel<Q><<<ceil_div(n * 2, 256), 256, 0, x->stream>>>(R, x->stride, nullptr, slot_base, n, (int)x->dim, N);     \
  } while (0)
  COLTT_DISPATCH_QUANT(x->quant, COLTT_PREP)
#undef COLTT_PREP
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// derived per-slot data of the slots a writer added (the product quantiser's codes); completes on the device before it returns
int sync_pq(Hnsw* x);
// stored rows [first, first + m) -> the f32 values the index's distance sees (what the quantiser encodes), packed [m][dim]
template <int QUANT, bool R8>
__global__ void rows_to_f32_kernel(const uint8_t* __restrict__ rows, size_t stride, uint64_t first, uint64_t m, int dim, float* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * (uint64_t)dim) return;
  const uint64_t i = t / dim; const int e = (int)(t - i * dim);
  int pos = e;
  if constexpr (R8) pos = r8_index<QUANT>(e);
  out[t] = load1<QUANT>(rows + (first + i) * stride, pos);
}
// the stored codes of rows [first, first + m) in NATURAL element order, packed [m][dim * elem bytes] (read-backs of a line-transposed index)
template <int QUANT>
__global__ void rows_natural_kernel(const uint8_t* __restrict__ rows, size_t stride, uint64_t first, uint64_t m, int dim, uint8_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * (uint64_t)dim) return;
  const uint64_t i = t / dim; const int e = (int)(t - i * dim);
  const uint8_t* row = rows + (first + i) * stride;
  const int pos = r8_index<QUANT>(e);
  if constexpr (QUANT == Q_NONE) reinterpret_cast<uint32_t*>(out)[t] = reinterpret_cast<const uint32_t*>(row)[pos];
  else reinterpret_cast<unsigned short*>(out)[t] = reinterpret_cast<const unsigned short*>(row)[pos];
}
// codes of the slots added since the last call (hnsw_pq.hpp); completes on the device before it returns
int sync_pq(Hnsw* x) {
  if (!x->pq_on || x->pq_done >= x->n) { if (x->pq_done > x->n) x->pq_done = x->n; return COLTT_OK; }
  const size_t old = x->pq_codes.cap;
  COLTT_TRY(x->pq_codes.reserve(std::max<uint64_t>(x->cap, x->n) * x->pq_row, true, x->stream));
  if (x->pq_codes.cap > old) COLTT_HIP(hipMemsetAsync(x->pq_codes.as<uint8_t>() + old, 0, x->pq_codes.cap - old, x->stream));   // the bytes j >= m of a row read as code 0
  const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)x->dim * 4));
  COLTT_TRY(x->pq_stage.reserve(std::min<uint64_t>(chunk, x->n - x->pq_done) * x->dim * 4));
  for (uint64_t b = x->pq_done; b < x->n; b += chunk) {
    const uint64_t m = std::min<uint64_t>(chunk, x->n - b);
    if (x->quant == COLTT_Q_NONE) {
      if (x->r8) rows_to_f32_kernel<Q_NONE, true><<<ceil_div(m * x->dim, 256), 256, 0, x->stream>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, x->pq_stage.as<float>());
      else rows_to_f32_kernel<Q_NONE, false><<<ceil_div(m * x->dim, 256), 256, 0, x->stream>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, x->pq_stage.as<float>());
    } else {   // (attach refuses "f8" rows)
      if (x->r8) rows_to_f32_kernel<Q_F16, true><<<ceil_div(m * x->dim, 256), 256, 0, x->stream>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, x->pq_stage.as<float>());
      else rows_to_f32_kernel<Q_F16, false><<<ceil_div(m * x->dim, 256), 256, 0, x->stream>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, x->pq_stage.as<float>());
    }
    COLTT_HIP(hipGetLastError());
    COLTT_TRY(pq_encode_rowmajor(x->stream, x->pq_cb.as<float>(), x->pq_shape, x->pq_stage.as<float>(), m, x->pq_codes.as<uint8_t>() + b * x->pq_row, x->pq_row));
  }
  COLTT_HIP(hipStreamSynchronize(x->stream));
  x->pq_done = x->n;
  return COLTT_OK;
}
// does this shape have a rows8 copy (rows8.hpp: f32 / 2-byte rows whose byte length is a multiple of 128)?
bool rows8_shape(uint32_t dim, int quant) {
  const int pol = policy().rows8;
  if (quant == COLTT_Q_F8 || pol == 0) return false;
  // Short rows stay on the lane pairs: the eight-lane core pays a fixed LDS hand-off per chunk of neighbours, which a 128-element row does
  // not amortise (1 M x 128 f32, ef 20 — the reference's published point: 1.41 ms per 10 k queries against 1.17, one query 0.130 against
  // 0.103 ms; 2 M x 256 f16, ef 64: 4.24 against 5.17 ms, +22 % — profiles/r04h_ev8_short_rows.json).  COLTT_ROWS8=2 lifts the limit.
  if (pol == 1 && dim < 256) return false;
  return ((size_t)dim * quant_bytes(quant)) % 128 == 0;
}

int prep_queries_any(Hnsw* x, HCtx* c, const float* d_qraw, size_t nq, uint32_t* zero64 = nullptr) {
  COLTT_TRY(c->w_qeff.reserve(nq * x->dim * 4));
  COLTT_TRY(c->w_qn.reserve(nq * 4));
  int norm = x->metric == COLTT_COSINE;
  float* qe = c->w_qeff.as<float>();
#define COLTT_PQ(Q) launch_prep_queries<Q>(c->stream, d_qraw, nq, (int)x->dim, norm, qe, c->w_qn.as<float>(), zero64)
  COLTT_DISPATCH_QUANT(x->quant, COLTT_PQ)
#undef COLTT_PQ
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

struct SearchGeom { uint32_t ef, ef_pad, hcap; size_t lds; bool visg; uint32_t max_grid; int w2 = -1; uint32_t bloom_words = 0; bool w2_lds = false; bool ev8 = false; };  // w2: hnsw_walk2.hpp variant (OPT bits | 8 = deep profile), -1 = hnsw_dev.hpp:search_level

#ifndef COLTT_VISG_MIN_EF
#define COLTT_VISG_MIN_EF 128
#endif
constexpr uint32_t VIS_MAX_REGIONS = 3072;  // 12 waves on each of 256 CUs: the product-quantised walk (3 per SIMD)
constexpr uint32_t VIS_ROW_REGIONS = 2048;  // 8 waves per CU: the row walks and the builder

// (Re)allocate the HBM visited set for the current slot capacity: one byte per slot and workgroup, zeroed, epochs reset.
// Sized against a quarter of the device memory; if that buys fewer than one region per CU the LDS hash is used instead.
int ensure_visg(Hnsw* x) {
  // Callers hold the index lock (shared or exclusive).  The slot capacity only changes under the exclusive lock, and every
  // search that wants the byte map passes through here first, so a re-allocation can never race a traversal that uses it.
  std::lock_guard<std::mutex> vg(x->vis_mu);
  const uint64_t stride = (std::max<uint64_t>(x->cap, 1) + 1023) & ~1023ull;
  // the row walks keep at most 8 traversals per CU (2048 regions); the product-quantised walk fits up to 12 (3072): only an index that carries a
  // quantiser pays for those (10 M slots x 3072 regions pinned 30 GB instead of 20 for everybody)
  const uint32_t want_max = x->pq_on ? VIS_MAX_REGIONS : VIS_ROW_REGIONS;
  if (x->vis_stride == stride && x->vis_want == want_max) return COLTT_OK;
  if (x->w_visg.p) { (void)hipFree(x->w_visg.p); x->w_visg.p = nullptr; x->w_visg.cap = 0; }
  size_t free_b = 0, total_b = 0;
  COLTT_HIP(hipMemGetInfo(&free_b, &total_b));
  uint64_t budget = std::min<uint64_t>(total_b / 4, free_b / 2);
  if (policy().visg_budget_mb >= 0) budget = std::min<uint64_t>(budget, (uint64_t)policy().visg_budget_mb << 20);  // test knob
  const uint64_t regions = std::min<uint64_t>(want_max, budget / stride);
  x->vis_stride = stride; x->vis_want = want_max;
  x->vis_regions = 0;
  x->vis_busy.clear();
  if (regions < 256) {
    fprintf(stderr, "[coltt_gpu] hnsw: HBM visited workspace would get only %llu regions of %llu B (budget %llu B) — using the LDS hash "
                    "visited set for ef > %d as well\n", (unsigned long long)regions, (unsigned long long)stride, (unsigned long long)budget, COLTT_VISG_MIN_EF);
    return COLTT_OK;
  }
  if (regions < want_max)
    fprintf(stderr, "[coltt_gpu] hnsw: HBM visited workspace capped at %llu of %u regions (%llu B each, budget %llu B): fewer resident "
                    "traversals per launch\n", (unsigned long long)regions, want_max, (unsigned long long)stride, (unsigned long long)budget);
  COLTT_TRY(x->w_visg.reserve(regions * stride));
  COLTT_TRY(x->w_vepoch.reserve(VIS_MAX_REGIONS * 4));
  COLTT_HIP(hipMemsetAsync(x->w_visg.p, 0, regions * stride, x->stream));
  COLTT_HIP(hipMemsetAsync(x->w_vepoch.p, 0, VIS_MAX_REGIONS * 4, x->stream));
  COLTT_HIP(hipStreamSynchronize(x->stream));  // searches run on other streams
  x->vis_regions = (uint32_t)regions;
  x->vis_busy.assign(regions, 0);
  return COLTT_OK;
}

// Lease a contiguous run of visited-set regions for one search launch: `want` if a free run that long exists, else the
// longest free run; blocks while every region is out.  Released by the RegionLease destructor.
struct RegionLease {
  Hnsw* x = nullptr; uint32_t base = 0, count = 0;
  ~RegionLease() {
    if (!x || !count) return;
    { std::lock_guard<std::mutex> g(x->vis_mu); for (uint32_t i = 0; i < count; i++) x->vis_busy[base + i] = 0; }
    x->vis_cv.notify_all();
  }
};
void acquire_regions(Hnsw* x, uint32_t want, RegionLease& out) {
  std::unique_lock<std::mutex> lk(x->vis_mu);
  const uint32_t R = (uint32_t)x->vis_busy.size();
  if (want > R) want = R;
  for (;;) {
    uint32_t best_b = 0, best_l = 0;
    for (uint32_t i = 0; i < R;) {
      if (x->vis_busy[i]) { i++; continue; }
      uint32_t j = i; while (j < R && !x->vis_busy[j] && j - i < want) j++;
      if (j - i > best_l) { best_l = j - i; best_b = i; }
      if (best_l == want) break;
      i = j;
    }
    if (best_l > 0) {
      for (uint32_t i = 0; i < best_l; i++) x->vis_busy[best_b + i] = 1;
      out.x = x; out.base = best_b; out.count = best_l;
      return;
    }
    x->vis_cv.wait(lk);
  }
}

// COLTT_VISG=0 / 1 forces the LDS hash / the HBM byte map (measurement and test knob, read at every call); default: the
// byte map above ef 128.  Measured on 10 M x 768 f16 (lowrank:32), queries/s LDS hash -> byte map: ef 128 728 k -> 724 k,
// ef 256 246 k -> 404 k, ef 512 62 k -> 212 k, ef 1024 25 k -> 105 k; build (efConstruction 200) 36 s -> 22 s.
int visg_policy() { return policy().visg; }

// does this ef want the HBM visited set (policy only; whether the workspace could be had is vis_regions > 0)
bool wants_visg(uint32_t ef) {
  const int pol = visg_policy();
  return pol < 0 ? ef > COLTT_VISG_MIN_EF : pol != 0;
}

// Which level-0 walk serves the HBM-visited searches.  COLTT_WALK2=off: hnsw_dev.hpp:search_level (the round-2 kernel);
// COLTT_WALK2=<n>: hnsw_walk2.hpp with OPT = n & 7 (1 Bloom, 2 delta result set, 4 adjacency-carried norms), n & 8 = the deep
// profile (one wave per SIMD, whole 2-byte row in flight).  Measurement and test knob, read at every call.
#ifndef COLTT_WALK2_DEFAULT
#define COLTT_WALK2_DEFAULT 7
#endif
int walk2_policy() { const int v = policy().walk2; return v == 7 ? COLTT_WALK2_DEFAULT : v; }

// The walk of the LDS-visited searches (ef <= 128 by default).  COLTT_WALK2_LDS=off: hnsw_dev.hpp:search_level; 2 / 4 / 6: hnsw_walk2.hpp
// with the delta result set / adjacency-carried norms / both.  A traversal that would overflow the hash table re-runs the call on
// search_level (which re-seeds the table from the result set) — never seen on the benchmark collections.
// 10 M x 768 f32, 10 000 queries, ms per launch (profiles/r03_walk_lds_f32_10m.json): ef 128 off 22.39, 2 22.25, 4 21.19, 6 21.28;
// ef 64 off 11.80, 2 11.77, 4 11.10, 6 11.25 — the norms riding with the adjacency rows are the gain; the delta set costs a little
// when the whole result set is two 64-entry chunks.  The default build carries 4 (2 and 6: -DCOLTT_WALK_EXPERIMENTS).
#ifndef COLTT_WALK2_LDS_DEFAULT
#define COLTT_WALK2_LDS_DEFAULT 4
#endif
int walk2_lds_policy() { const int v = policy().walk2_lds; return v == 4 ? COLTT_WALK2_LDS_DEFAULT : v; }

// resident waves per CU of the walk2 profiles: see waves_per_cu_cap
size_t waves_per_cu_cap(int quant);

// COLTT_EV8=0: level-0 distances from the pair-owned rows even when the index carries the line-transposed copy (A/B and test knob)
bool ev8_policy() { return policy().ev8; }

SearchGeom search_geom(Hnsw* x, uint32_t ef, bool for_search = false, bool no_w2_lds = false) {
  SearchGeom s;
  s.ef = ef;
  s.ef_pad = (ef + 63) & ~63u;
  const size_t qbytes = ((size_t)x->dim * 4 + 15) & ~(size_t)15;
  // eight lanes per row (rows8.hpp): a second, permuted copy of the query and 96 words of scratch per wave.  Served by the shipped
  // walk2 variants only (LDS hash: 4; HBM map: 6 / 7).
  const bool vis_hbm = wants_visg(ef) && x->vis_stride != 0 && x->vis_regions > 0;
  const bool want8 = for_search && x->r8 && x->n > 0 && ev8_policy() && x->cfg.m_max0 <= 1024 &&
                     (vis_hbm ? (walk2_policy() == 6 || walk2_policy() == 7) : (!no_w2_lds && walk2_lds_policy() == 4));
  const size_t fixed = qbytes + (want8 ? 96 * 4 : 0) + (size_t)s.ef_pad * 8;   // query (+ the eight-lane core's scratch) + result set (merged in place)
  // LDS visited set: sized so that a typical traversal (a few dozen evaluations per result slot) never resets
  s.hcap = std::min<uint32_t>(32768u, std::max<uint32_t>(8192u, next_pow2(ef * 48u)));
  // large ef x dim: shrink it until the wave's state fits the CU's 160 KiB (the reset-and-reseed path keeps results exact;
  // it needs 0.75 * hcap > ef + 64)
  while (fixed + (size_t)s.hcap * 4 > 160 * 1024 && s.hcap > 4096 && (s.hcap / 2) * 3 / 4 > ef + 64) s.hcap /= 2;
  s.visg = wants_visg(ef) && x->vis_stride != 0 && x->vis_regions > 0;
  if (s.visg) { s.hcap = 64; s.lds = fixed; s.max_grid = x->vis_regions; }
  else { s.lds = fixed + (size_t)s.hcap * 4; s.max_grid = 0xffffffffu; }
  if (s.visg && for_search && x->cfg.m_max0 <= 1024) {
    s.w2 = walk2_policy();
    if (s.w2 >= 0 && (s.w2 & 1)) {
      // Bloom filter: the largest power of two that keeps the profile's resident waves (>= 2 KiB, <= 32 KiB), else none
      const size_t waves = (s.w2 & 8) ? 4 : waves_per_cu_cap(x->quant);
      const size_t budget = (160 * 1024) / waves;
      size_t kb = 32;
      if (policy().bloom_kb > 0) kb = (size_t)policy().bloom_kb;
      else while (kb >= 2 && fixed + kb * 1024 > budget) kb >>= 1;
      size_t p2 = 1; while (p2 * 2 <= kb) p2 *= 2;   // power of two
      if (kb >= 2 && fixed + p2 * 1024 <= 160 * 1024) { s.bloom_words = (uint32_t)(p2 * 256); s.lds = fixed + p2 * 1024; }
      else s.w2 &= ~1;
    }
  }
  if (!s.visg && for_search && !no_w2_lds && x->cfg.m_max0 <= 1024) {
    // small ef: the same walk over the LDS hash (delta result set, adjacency-carried norms; no Bloom filter — the hash is on chip)
    const int pol = walk2_lds_policy();
    if (pol >= 0) { s.w2 = pol; s.w2_lds = true; }
  }
  s.ev8 = want8 && s.w2 >= 0;
  return s;
}

// Resident waves per CU.  2-byte rows want all 8 (two per SIMD); f32 rows are faster with 4: eight 3-KB-per-row streams per
// CU measured slower both in search (ef 256: 160 k vs 197 k q/s) and in the 10 M build (61 s vs 50 s).  COLTT_WAVES_PER_CU
// overrides (measurement knob).
size_t waves_per_cu_cap(int quant) {
  if (policy().waves_per_cu > 0) return (size_t)policy().waves_per_cu;
  return quant == Q_NONE ? 4 : 8;
}

uint32_t resident_waves(const SearchGeom& sg, int quant) {
  const size_t cap = (sg.w2 >= 0 && (sg.w2 & 8)) ? 4 : waves_per_cu_cap(quant);
  return 256u * (uint32_t)std::max<size_t>(1, std::min<size_t>(cap, (160 * 1024) / sg.lds));
}

// Non-temporal row loads for the eight-lane walks (exact.hpp: row_ld): worth it when the row array is far larger than L2 + MALL, costly when later queries
// would have found the rows there.  COLTT_ROWS_NT = 0 / 1 forces, COLTT_ROWS_NT_MIN_MB moves the threshold (default 12 GiB of rows: 768-d rows, same box — f16 0.5 M -24 %, 1 M -10 %,
// 2 M -2 %, 4 M +0.8 %, 10 M +-0; f32 1 M -4 %, 3 M +0.7 %, 10 M +4.5 %: profiles/r06ag_nt_rows_ab.md).
bool rows_nt(const Hnsw* x) {
  const Policy p = policy();
  const unsigned long long bytes = (unsigned long long)x->n * (unsigned long long)x->stride, least = (unsigned long long)p.rows_nt_min_mb << 20;
  const bool on = p.rows_nt >= 0 ? p.rows_nt != 0 : bytes >= least;
  static const bool dbg = [] { const char* e = getenv("COLTT_DEBUG_ROWS_NT"); return e && *e == '1'; }();   // diagnostics: which twin a launch takes, and why
  if (dbg) fprintf(stderr, "[rows_nt] slots=%llu stride=%zu row bytes=%llu threshold=%llu knob=%d -> %s\n", (unsigned long long)x->n, (size_t)x->stride, bytes, least, p.rows_nt, on ? "nt" : "default");
  return on;
}

// hnsw_walk2.hpp kernels.  The default build carries the shipped variant (and its Bloom-less twin for geometries whose LDS
// has no room for the filter); -DCOLTT_WALK_EXPERIMENTS adds every OPT x profile combination for A/B runs (tools/walk_sweep.py).
template <int METRIC, int QUANT>
int launch_search2(Hnsw* x, HCtx* c, const SearchGeom& sg, uint32_t grid, uint32_t region_base, uint32_t nq, uint32_t k, uint32_t* counter,
                   uint64_t* oi, float* os, uint32_t* oc, unsigned long long* stats) {
  typedef void (*kern_t)(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*,
                         uint64_t*, float*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
  kern_t kern = nullptr;
  const bool nt = rows_nt(x);   // non-temporal row loads (eight-lane kernels): see exact.hpp: row_ld
#define COLTT_W2(V, PROF, OPT) case V: kern = hnsw_search2_kernel<METRIC, QUANT, PROF, OPT>; break;
  if (sg.w2_lds) {
    switch (sg.w2) {
#ifdef COLTT_WALK_EXPERIMENTS
      case 2: kern = hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_LDS, 2, VIS_LDS>; break;
      case 6: kern = hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_LDS, 6, VIS_LDS>; break;
#endif
      case 4:
        kern = hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_LDS, 4, VIS_LDS>;
        if constexpr (QUANT != Q_F8) {
          if (sg.ev8) kern = nt ? hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_LDS, 4, VIS_LDS, false, true, false, true> : hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_LDS, 4, VIS_LDS, false, true>;
          else if (x->r8) kern = hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_LDS, 4, VIS_LDS, false, false, true>;   // the pair-owned core over the line-transposed rows
        }
        break;   // (adjacency prefetch for f32 rows in small batches: measured, no gain — 1 M x 128, ef 20, one query 105 vs 111 us)
      default: break;
    }
  } else
#ifdef COLTT_WALK_EXPERIMENTS
  if constexpr (METRIC == M_COS && QUANT != Q_F8) {
    switch (sg.w2) {
      COLTT_W2(0, PROF_SEARCH_HBM, 0) COLTT_W2(1, PROF_SEARCH_HBM, 1) COLTT_W2(2, PROF_SEARCH_HBM, 2) COLTT_W2(3, PROF_SEARCH_HBM, 3)
      COLTT_W2(4, PROF_SEARCH_HBM, 4) COLTT_W2(5, PROF_SEARCH_HBM, 5) COLTT_W2(6, PROF_SEARCH_HBM, 6) COLTT_W2(7, PROF_SEARCH_HBM, 7)
      COLTT_W2(8, PROF_SEARCH_HBM_DEEP, 0) COLTT_W2(9, PROF_SEARCH_HBM_DEEP, 1) COLTT_W2(10, PROF_SEARCH_HBM_DEEP, 2) COLTT_W2(11, PROF_SEARCH_HBM_DEEP, 3)
      COLTT_W2(12, PROF_SEARCH_HBM_DEEP, 4) COLTT_W2(13, PROF_SEARCH_HBM_DEEP, 5) COLTT_W2(14, PROF_SEARCH_HBM_DEEP, 6) COLTT_W2(15, PROF_SEARCH_HBM_DEEP, 7)
      default: break;
    }
  } else
#endif
  {
    switch (sg.w2) {
      COLTT_W2(6, PROF_SEARCH_HBM, 6) COLTT_W2(7, PROF_SEARCH_HBM, 7)
      default: break;
    }
    if constexpr (QUANT != Q_F8) {
      if (sg.ev8 && sg.w2 == 6) kern = nt ? hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_HBM, 6, VIS_HBM, false, true, false, true> : hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_HBM, 6, VIS_HBM, false, true>;
      if (sg.ev8 && sg.w2 == 7) kern = nt ? hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_HBM, 7, VIS_HBM, false, true, false, true> : hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_HBM, 7, VIS_HBM, false, true>;
      if (!sg.ev8 && x->r8 && sg.w2 == 6) kern = hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_HBM, 6, VIS_HBM, false, false, true>;
      if (!sg.ev8 && x->r8 && sg.w2 == 7) kern = hnsw_search2_kernel<METRIC, QUANT, PROF_SEARCH_HBM, 7, VIS_HBM, false, false, true>;
    }
  }
#undef COLTT_W2
  if (!kern) return fail(COLTT_E_UNSUPPORTED, "hnsw_search: walk variant %d is not compiled into this build (COLTT_WALK2)", sg.w2);
  if (x->r8 && !sg.ev8 && !(sg.w2_lds ? sg.w2 == 4 : (sg.w2 == 6 || sg.w2 == 7)))
    return fail(COLTT_E_UNSUPPORTED, "hnsw_search: walk variant %d has no instance for line-transposed rows (create the index with COLTT_ROWS8=0 for this experiment)", sg.w2);
  if (sg.ev8) x->ev8_launches.fetch_add(1);
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg.lds));
  kern<<<grid, 64, sg.lds, c->stream>>>(x->view(), x->entry, x->entry_level, c->w_qeff.as<float>(), c->w_qn.as<float>(), nq,
                                        k, sg.ef, sg.ef_pad, sg.w2_lds ? sg.hcap : sg.bloom_words, counter, oi, os, oc, stats,
                                        x->w_visg.as<uint8_t>() + (size_t)region_base * x->vis_stride, (size_t)x->vis_stride,
                                        x->w_vepoch.as<uint32_t>() + region_base);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

template <int METRIC, int QUANT>
int launch_search(Hnsw* x, HCtx* c, const SearchGeom& sg, uint32_t grid, uint32_t region_base, uint32_t nq, uint32_t k, uint32_t* counter,
                  uint64_t* oi, float* os, uint32_t* oc, unsigned long long* stats) {
  auto kern = sg.visg ? hnsw_search_kernel<METRIC, QUANT, true> : hnsw_search_kernel<METRIC, QUANT, false>;
  if constexpr (QUANT != Q_F8) { if (x->r8) kern = sg.visg ? hnsw_search_kernel<METRIC, QUANT, true, true> : hnsw_search_kernel<METRIC, QUANT, false, true>; }
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg.lds));
  kern<<<grid, 64, sg.lds, c->stream>>>(x->view(), x->entry, x->entry_level, c->w_qeff.as<float>(), c->w_qn.as<float>(), nq,
                                        k, sg.ef, sg.ef_pad, sg.hcap, counter, oi, os, oc, stats,
                                        x->w_visg.as<uint8_t>() + (size_t)region_base * x->vis_stride, (size_t)x->vis_stride,
                                        x->w_vepoch.as<uint32_t>() + region_base);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// latency launch: one 4-wave workgroup per query in flight, at most one per CU
template <int METRIC, int QUANT>
int launch_search_lat(Hnsw* x, HCtx* c, const SearchGeom& sg, uint32_t grid, uint32_t region_base, uint32_t nq, uint32_t k, uint32_t* counter,
                      uint64_t* oi, float* os, uint32_t* oc, unsigned long long* stats) {
  (void)region_base;
  // COLTT_LAT_SEQ=1: the sequential walk (search_level2 + LatEval) also for one-chunk rows — the A/B partner of the pipelined one
  const bool seq = policy().lat_seq || x->cfg.m_max0 > 32;
  // the latency kernel reads the index's ONE row array in the layout it has (line-transposed: lines evaluated out of their registers; natural:
  // staged and transposed through LDS).  COLTT_EV8 chooses between distance cores over the same bytes in the THROUGHPUT kernels; here the layout
  // alone decides (the A/B partner is an index created with COLTT_ROWS8=0).  One kernel instance per (layout, walk): TP = the row's 128-byte lines
  // as a compile-time constant for the common shapes (24: 768 x f32; 12: 768 x 2 bytes, 384 x f32), -1 any line-transposed row, 0 natural order.
  GraphView gv = x->view();
  typedef void (*lat_kern_t)(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint64_t*, float*, uint32_t*, unsigned long long*,
                             unsigned long long*, int);
  lat_kern_t kern;
  if constexpr (QUANT == Q_F8) kern = seq ? (lat_kern_t)hnsw_search_lat_kernel<METRIC, QUANT, LAT_TP_STAGED, true> : (lat_kern_t)hnsw_search_lat_kernel<METRIC, QUANT, LAT_TP_STAGED, false>;
  else {
    const int lines = gv.rows8 ? (int)(x->stride >> 7) : 0;
    if (gv.rows8) x->ev8_launches.fetch_add(1);
    if (!gv.rows8) kern = seq ? (lat_kern_t)hnsw_search_lat_kernel<METRIC, QUANT, LAT_TP_STAGED, true> : (lat_kern_t)hnsw_search_lat_kernel<METRIC, QUANT, LAT_TP_STAGED, false>;
    else if (seq) kern = hnsw_search_lat_kernel<METRIC, QUANT, LAT_TP_R8_ANY, true>;
    else if (lines == 24) kern = hnsw_search_lat_kernel<METRIC, QUANT, 24, false>;
    else if (lines == 12) kern = hnsw_search_lat_kernel<METRIC, QUANT, 12, false>;
    else kern = hnsw_search_lat_kernel<METRIC, QUANT, LAT_TP_R8_ANY, false>;
  }
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg.lds));
  // Batches of at most LAT_MASTERS queries (the reference's one-query RPC above all) get helper workgroups that warm the XCD's L2 with the rows the walk
  // will want next (hnsw_lat.hpp).  COLTT_LAT_HELPERS = 1 .. 3 per walking workgroup; OFF by default — exact, but 3-5 % slower than no helpers at all
  // (profiles/r06f_latency_helpers_ab.md); not with the sequential walk.
  unsigned long long* mbox = nullptr;
  const int helpers = seq ? 0 : std::min(policy().lat_helpers, LAT_HELPERS_MAX);
  if (helpers > 0 && nq <= (uint32_t)LAT_MASTERS) {
    const size_t mb = (size_t)LAT_MASTERS * (LAT_HELPERS_MAX + 1) * 8;
    COLTT_TRY(c->w_mbox.reserve(mb));
    COLTT_HIP(hipMemsetAsync(c->w_mbox.p, 0, mb, c->stream));
    mbox = c->w_mbox.as<unsigned long long>();
    grid = (uint32_t)LAT_MASTERS * (uint32_t)(helpers + 1);
  }
  kern<<<grid, 256, sg.lds, c->stream>>>(gv, x->entry, x->entry_level, c->w_qeff.as<float>(), c->w_qn.as<float>(), nq,
                                         k, sg.ef, sg.ef_pad, sg.hcap, counter, oi, os, oc, stats, mbox, helpers);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// Batches of at most COLTT_LAT_MAX_NQ queries (default below; COLTT_MW_MAX_NQ is the round-2 name of the knob) take the
// 256-thread latency kernel (hnsw_lat.hpp).
// Measured at 10 M x 768 f32, ef 128 (profiles/r03_latency.json): 1 query 1.04 ms against 1.43 ms on the one-wave kernel, 128 queries
// 1.45 against 1.97 ms; beyond ~256 queries every CU already holds a workgroup and the one-wave kernel's 1024+ resident traversals win.
#ifndef COLTT_LAT_MAX_NQ_DEFAULT
#define COLTT_LAT_MAX_NQ_DEFAULT 256
#endif
#ifndef COLTT_LAT_MIN_STRIDE
#define COLTT_LAT_MIN_STRIDE 1024   // bytes per stored row below which a lane pair streams the whole row in one burst anyway
#endif
uint32_t lat_max_nq() { const Policy p = policy(); return p.lat_knob_set ? p.lat_max_nq : (uint32_t)COLTT_LAT_MAX_NQ_DEFAULT; }

int search_common(Hnsw* x, HCtx* c, const float* queries, bool on_device, size_t nq, uint32_t k, uint32_t ef_override,
                  uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats, int force = 0) {
  const bool force_single_wave = (force & 1) != 0;   // force: 1 = not the latency kernel, 2 = not the walk2 LDS-hash kernel (both after an err 8)
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (nq == 0) return COLTT_OK;
  if (k == 0) return fail(COLTT_E_INVALID, "hnsw_search: k must be >= 1");
  uint64_t* d_oi = out_ids; float* d_os = out_scores; uint32_t* d_oc = out_counts;
  // small host-buffer calls: the work counter, the traversal counters and the three answer arrays live in ONE device block that
  // comes back in one copy through page-locked staging (layout: 256 B counter + stats | ids | scores | counts)
  const size_t pack_bytes = 256 + nq * k * 12 + nq * 4;
  const bool packed = !on_device && pack_bytes <= SMALL_CALL_BYTES && nq * x->dim * 4 <= SMALL_CALL_BYTES && small_call_staging();
  if (packed) {
    COLTT_TRY(c->w_pack.reserve(SMALL_CALL_BYTES + 256));
    COLTT_TRY(c->h_out.reserve(SMALL_CALL_BYTES + 256));
    COLTT_TRY(c->h_in.reserve(SMALL_CALL_BYTES));
    uint8_t* b = c->w_pack.as<uint8_t>();
    d_oi = reinterpret_cast<uint64_t*>(b + 256); d_os = reinterpret_cast<float*>(b + 256 + nq * k * 8); d_oc = reinterpret_cast<uint32_t*>(b + 256 + nq * k * 12);
  } else if (!on_device) {
    COLTT_TRY(c->w_out_ids.reserve(nq * k * 8));
    COLTT_TRY(c->w_out_sc.reserve(nq * k * 4));
    COLTT_TRY(c->w_out_cnt.reserve(nq * 4));
    d_oi = c->w_out_ids.as<uint64_t>(); d_os = c->w_out_sc.as<float>(); d_oc = c->w_out_cnt.as<uint32_t>();
  }
  if (x->entry < 0) {  // empty index => empty result, not an error (hnsw.go:249-251)
    COLTT_HIP(hipMemsetAsync(d_oc, 0, nq * 4, c->stream));
    if (!on_device) COLTT_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipStreamSynchronize(c->stream));
    return COLTT_OK;
  }
  uint32_t ef = std::max<uint32_t>(ef_override ? ef_override : (uint32_t)x->cfg.ef, k);  // gomath.MaxInt(ef, k), hnsw.go:258
  if (ef > 4096) return fail(COLTT_E_UNSUPPORTED, "hnsw_search: ef=%u > 4096", ef);
  if (wants_visg(ef)) COLTT_TRY(ensure_visg(x));  // lazily: N bytes x <= 2048 regions are only worth having for ef > 128
  SearchGeom sg = search_geom(x, ef, true, (force & 2) != 0);
  if (sg.lds > 160 * 1024) return fail(COLTT_E_UNSUPPORTED, "hnsw_search: dim/ef need %zu B of LDS (> 160 KiB)", sg.lds);
  // latency kernel for small batches.  Its LDS hash must never need the reset path, so it gets the largest table that fits beside the
  // staging area (one workgroup per CU); if even that is too small for this ef the one-wave kernel serves the call.
  bool mw = nq <= lat_max_nq() && !force_single_wave;
  if (mw) {
    SearchGeom m = sg; m.w2 = -1; m.w2_lds = false; m.bloom_words = 0; m.visg = false;
    // LDS: query + result set + exchange words + the staging area (32 padded rows) + the visited hash
    // (line-transposed rows are evaluated out of the registers they land in: no staging area — hnsw_lat.hpp: lat_eval_chunk, TP != 0)
    const bool lat_r8 = x->r8 && x->quant != COLTT_Q_F8;
    const size_t fixed = ((lat_q_floats((int)x->dim) * 4 + 15) & ~(size_t)15) + (size_t)m.ef_pad * 8 + sizeof(LatShared) + (lat_r8 ? (size_t)0 : (size_t)LAT_ROWS * (x->stride + LAT_PAD));
    if (x->stride > LAT_MAX_STRIDE) mw = false;               // rows too long to hold (or stage) 32 at a time
    else if (x->stride < COLTT_LAT_MIN_STRIDE && !policy().lat_knob_set) mw = false;   // short rows: the one-wave kernel (unless asked for)
    else {
      m.hcap = 32768; while (fixed + (size_t)m.hcap * 4 > 160 * 1024 && m.hcap > 1024) m.hcap /= 2;
      m.lds = fixed + (size_t)m.hcap * 4;
      if (m.lds > 160 * 1024 || (m.hcap / 4) * 3 < ef * 34u + 64u) mw = false;        // table too small to be sure: one-wave kernel
    }
    if (mw) sg = m;
  }
  uint32_t grid = mw ? std::min<uint32_t>((uint32_t)nq, 256u) : std::min<uint32_t>((uint32_t)std::min<size_t>(nq, 0xffffffffu), resident_waves(sg, x->quant));
  RegionLease lease;
  if (sg.visg) { acquire_regions(x, grid, lease); grid = lease.count; }
  const float* d_q = queries;
  if (!on_device) {
    COLTT_TRY(c->w_qraw.reserve(nq * x->dim * 4));
    const void* src = queries;
    if (packed) { std::memcpy(c->h_in.p, queries, nq * x->dim * 4); src = c->h_in.p; }
    COLTT_HIP(hipMemcpyAsync(c->w_qraw.p, src, nq * x->dim * 4, hipMemcpyHostToDevice, c->stream));
    d_q = c->w_qraw.as<float>();
  }
  COLTT_TRY(c->w_misc.reserve(256));
  if (!packed) COLTT_TRY(c->h_out.reserve(256));   // the traversal counters come back through page-locked memory (a pageable target makes the copy a blocking, staged one)
  uint8_t* misc = packed ? c->w_pack.as<uint8_t>() : c->w_misc.as<uint8_t>();
  uint32_t* counter = reinterpret_cast<uint32_t*>(misc);
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(misc + 16);
  COLTT_TRY(prep_queries_any(x, c, d_q, nq, reinterpret_cast<uint32_t*>(misc)));   // ... which also clears the 256 bytes of counters (no memset of their own)
  COLTT_HIP(hipEventRecord(c->ev0, c->stream));
  int rc;
#define COLTT_LS_ARGS x, c, sg, grid, lease.base, (uint32_t)nq, k, counter, d_oi, d_os, d_oc, d_stats
#define COLTT_LS(Q) rc = mw ? (x->metric == COLTT_COSINE ? launch_search_lat<M_COS, Q>(COLTT_LS_ARGS) : launch_search_lat<M_L2, Q>(COLTT_LS_ARGS)) \
                     : sg.w2 >= 0 ? (x->metric == COLTT_COSINE ? launch_search2<M_COS, Q>(COLTT_LS_ARGS) : launch_search2<M_L2, Q>(COLTT_LS_ARGS)) \
                            : (x->metric == COLTT_COSINE ? launch_search<M_COS, Q>(COLTT_LS_ARGS) : launch_search<M_L2, Q>(COLTT_LS_ARGS))
  COLTT_DISPATCH_QUANT(x->quant, COLTT_LS)
#undef COLTT_LS
#undef COLTT_LS_ARGS
  COLTT_TRY(rc);
  COLTT_HIP(hipEventRecord(c->ev1, c->stream));
  if (x->dense && x->dense_base) add_base_kernel<<<ceil_div(nq * k, 256), 256, 0, c->stream>>>(d_oi, nq * k, x->dense_base);
  unsigned long long h_stats[5] = {0, 0, 0, 0, 0};
  if (packed) COLTT_HIP(hipMemcpyAsync(c->h_out.p, c->w_pack.p, pack_bytes, hipMemcpyDeviceToHost, c->stream));
  else {
    if (!on_device) {
      COLTT_HIP(hipMemcpyAsync(out_ids, d_oi, nq * k * 8, hipMemcpyDeviceToHost, c->stream));
      COLTT_HIP(hipMemcpyAsync(out_scores, d_os, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
      COLTT_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, c->stream));
    }
    COLTT_HIP(hipMemcpyAsync(c->h_out.p, d_stats, 40, hipMemcpyDeviceToHost, c->stream));
  }
#ifdef COLTT_PHASE_TIMING
  unsigned long long h_pt[8] = {0};
  COLTT_HIP(hipMemcpyAsync(h_pt, d_stats + 8, 64, hipMemcpyDeviceToHost, c->stream));
#endif
  COLTT_HIP(hipStreamSynchronize(c->stream));  // the lease (destructor) outlives the kernel
  if (!packed) std::memcpy(h_stats, c->h_out.p, 40);
  if (packed) {
    const uint8_t* hb = c->h_out.as<uint8_t>();
    std::memcpy(h_stats, hb + 16, 40);
    std::memcpy(out_ids, hb + 256, nq * k * 8);
    std::memcpy(out_scores, hb + 256 + nq * k * 8, nq * k * 4);
    std::memcpy(out_counts, hb + 256 + nq * k * 12, nq * 4);
  }
#ifdef COLTT_PHASE_TIMING
  {
    static const char* nm_w[8] = {"pop", "adjacency", "visited", "rows+dist", "merge", "prologue(upper levels)", "writeout", "-"};
    static const char* nm_l[8] = {"pop+publish", "adjacency", "visited+fetch+stage", "eval", "barrier2+admission", "-", "prologue(upper levels)", "-"};
    const char* const* nm = mw ? nm_l : nm_w;
    double tot = 0; for (int i = 0; i < 7; i++) tot += (double)h_pt[i];
    fprintf(stderr, "[phase] nq=%zu ef=%u visg=%d:", nq, sg.ef, (int)sg.visg);
    for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h_pt[i] / tot);
    fprintf(stderr, "  | ticks/query %.0f\n", tot / (double)nq);
  }
#endif
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  x->last_ms.store(ms);
  if (mw && (h_stats[4] & 8ull))  // the multi-wave kernel's visited table would have needed a reset: same call on the single-wave kernel
    return search_common(x, c, queries, on_device, nq, k, ef_override, out_ids, out_scores, out_counts, stats, force | 1);
  if (sg.w2_lds && (h_stats[4] & 8ull))  // same for the walk2 kernel with the LDS hash: hnsw_dev.hpp:search_level has the reset-and-reseed path
    return search_common(x, c, queries, on_device, nq, k, ef_override, out_ids, out_scores, out_counts, stats, force | 3);
  if (h_stats[4]) return fail(COLTT_E_DEVICE, "hnsw_search: traversal watchdog tripped (code %llu)", h_stats[4]);
  if (stats) { stats->n_dist = h_stats[0]; stats->n_exp = h_stats[1]; stats->n_hops = h_stats[2]; stats->n_visit_resets = h_stats[3]; }
  return COLTT_OK;
}


// ---- Hnsw.Search over product-quantiser codes + exact re-rank (hnsw_pq.hpp) --------------------------------------------------------
// neighbourhood blocks: one thread per 16-byte piece of nbr[slot][p][row_bytes]
__global__ void pq_nbr_build_kernel(const uint32_t* __restrict__ adj0, const uint8_t* __restrict__ codes, uint8_t* __restrict__ nbr, uint64_t total, uint32_t row_bytes) {
  const uint32_t pieces = row_bytes >> 4;
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const uint64_t e = t / pieces; const uint32_t pc = (uint32_t)(t - e * pieces);   // e = slot * width + p
  const uint32_t nb = adj0[e];
  u32x4v v = {0u, 0u, 0u, 0u};
  if (nb != NBR_NONE) v = *reinterpret_cast<const u32x4v*>(codes + (size_t)nb * row_bytes + (size_t)pc * 16);
  *reinterpret_cast<u32x4v*>(nbr + e * row_bytes + (size_t)pc * 16) = v;
}
// Called by a search (shared lock held).  *use: the blocks are there and current.  They cost n x mMax0 x pq_row bytes (20.5 GB at 10 M x 32 x 64): an index
// that cannot afford them (less than twice that free) keeps gathering by neighbour slot.
int ensure_pq_nbr(Hnsw* x, hipStream_t stream, bool* use) {
  *use = false;
  if (!policy().pq_nbr || !x->pq_on || x->n == 0) return COLTT_OK;
  std::lock_guard<std::mutex> g(x->pq_nbr_mu);
  if (!x->pq_nbr_stale.load()) { *use = x->pq_nbr_ok; return COLTT_OK; }
  const uint32_t W = (uint32_t)x->cfg.m_max0;
  const size_t bytes = (size_t)std::max<uint64_t>(x->cap, x->n) * W * x->pq_row;
  x->pq_nbr_ok = false;
  if (x->pq_nbr.cap < bytes) {
    size_t free_b = 0, total_b = 0;
    COLTT_HIP(hipMemGetInfo(&free_b, &total_b));
    if (bytes > free_b / 2 || x->pq_nbr.reserve(bytes) != COLTT_OK) { (void)hipGetLastError(); x->pq_nbr_stale.store(false); return COLTT_OK; }
  }
  const uint64_t pieces = x->n * W * (x->pq_row >> 4);
  const uint64_t step = 840ull << 20;   // a multiple of every pieces-per-row (1 .. 8); grid.x stays under 2^31
  for (uint64_t b = 0; b < pieces; b += step) {
    const uint64_t m = std::min<uint64_t>(step, pieces - b);
    const uint64_t e0 = b / (x->pq_row >> 4);
    pq_nbr_build_kernel<<<(unsigned)ceil_div(m, 256), 256, 0, stream>>>(x->adj0.as<uint32_t>() + e0, x->pq_codes.as<uint8_t>(), x->pq_nbr.as<uint8_t>() + e0 * x->pq_row, m, x->pq_row);
  }
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipStreamSynchronize(stream));
  x->pq_nbr_ok = true; x->pq_nbr_stale.store(false);
  *use = true;
  return COLTT_OK;
}
// log2 of a table row in LDS: the centroid count rounded up to a power of two, at least 16 (codes >= C never occur)
uint32_t pq_lut_shift(const Hnsw* x) { uint32_t sh = 4; while ((1u << sh) < x->pq_shape.C) sh++; return sh; }
// Resident traversals per CU of the product-quantised walk (120 VGPRs: four waves per SIMD would fit; the LDS decides below that — 16 instead of 12
// measured nothing, profiles/r05s_pq_ab.md).
// The walk over the byte map carries NO Bloom filter (the row walks do, hnsw_walk2.hpp): the probe of the byte map is one wave-wide load that is
// issued whenever ANY lane's filter bits are set — and some ~10 of an expansion's 32 neighbours have been visited, so it is issued practically
// always; what the filter saves is probing lanes (HBM sectors), which this walk has to spare, and what it costs is an LDS atomic round trip in
// front of every probe plus 2-4 KiB of LDS per traversal (one resident wave per CU at ef 1 408).  Same box, 10 M x 768 f16, 64 x 32 quantiser
// (profiles/r05s_pq_ab.md): ef 1 024 410 -> 440 k queries/s, ef 1 408 277 -> 316 k.
constexpr size_t PQ_WAVES_CAP = 12;
struct PqGeom { uint32_t ef, ef_pad, vis_words; size_t lds; int variant; /* 0: LDS hash | 2: byte map + delta result set */ uint32_t per_cu; };
bool pq_geom(Hnsw* x, uint32_t ef, bool force_hbm, PqGeom& out) {
  PqGeom s{};
  s.ef = ef; s.ef_pad = (ef + 63) & ~63u;
  const size_t fixed = (size_t)s.ef_pad * 8 + ((size_t)pq_walk_table_rows(x->pq_row >> 4) << pq_lut_shift(x)) * 2;   // result set | binary16 table, pair-interleaved (the query stays in HBM: only the re-rank reads it)
  if (fixed > 160 * 1024) return false;
  const bool hbm_ok = x->vis_stride != 0 && x->vis_regions > 0;
  // LDS hash: as search_geom sizes it; it must never need the reset path (err 8 -> the call is re-run over the byte map)
  uint32_t hcap = std::min<uint32_t>(32768u, std::max<uint32_t>(8192u, next_pow2(ef * 48u)));
  const bool lds_fits = fixed + (size_t)hcap * 4 <= 160 * 1024;
  if (!force_hbm && !(wants_visg(ef) && hbm_ok) && lds_fits) { s.variant = 0; s.vis_words = hcap; s.lds = fixed + (size_t)hcap * 4; }
  else {
    if (!hbm_ok) return false;
    s.variant = 2; s.vis_words = 0; s.lds = fixed;
  }
  {
    const Policy pol = policy();
    const size_t cap = pol.pq_waves > 0 ? (size_t)pol.pq_waves : PQ_WAVES_CAP;
    s.per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(cap, (160 * 1024) / s.lds));
  }
  out = s;
  return true;
}

int launch_pq_walk(Hnsw* x, HCtx* c, const PqGeom& sg, bool nbr, uint32_t grid, uint32_t region_base, const unsigned short* lut, uint32_t nq, uint32_t k,
                   uint32_t rerank, uint32_t* counter, uint32_t* surv, uint32_t* surv_cnt, unsigned long long* stats) {
  const uint32_t sh = pq_lut_shift(x);
  // table row length x code-row pieces as compile-time constants for the common quantisers: 64 sub-vectors x 16 / 32 centroids (LS 4 / 5, 4 pieces),
  // 32 x 256 — the reference's shape, playground/hnswpq_verification.go:69-73 — (LS 8, 2 pieces), 96 x 256 (LS 8, 6 pieces); anything else runs the
  // forms that read the shape at run time.  Byte map (variant 2) with or without the neighbourhood blocks; the LDS-hash variant gathers.
  const uint32_t np = x->pq_row >> 4;
  typedef void (*pq_kern_t)(GraphView, int32_t, int32_t, const unsigned short*, const uint8_t*, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
                            uint32_t*, uint32_t*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
  pq_kern_t kern;
#define COLTT_PQK(LS, NP) (sg.variant == 0 ? (pq_kern_t)hnsw_pq_search_kernel<0, VIS_LDS, LS, NP, false> : nbr ? (pq_kern_t)hnsw_pq_search_kernel<2, VIS_HBM, LS, NP, true> : (pq_kern_t)hnsw_pq_search_kernel<2, VIS_HBM, LS, NP, false>)
  if (sh == 5 && np == 4) kern = COLTT_PQK(5, 4);
  else if (sh == 4 && np == 4) kern = COLTT_PQK(4, 4);
  else if (sh == 8 && np == 2) kern = COLTT_PQK(8, 2);
  else if (sh == 8 && np == 6) kern = COLTT_PQK(8, 6);
  else if (sh == 4) kern = COLTT_PQK(4, 0);
  else if (sh == 5) kern = COLTT_PQK(5, 0);
  else if (sh == 8) kern = COLTT_PQK(8, 0);
  else kern = COLTT_PQK(0, 0);
#undef COLTT_PQK
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sg.lds));
  kern<<<grid, 64, sg.lds, c->stream>>>(x->view(), x->entry, x->entry_level, lut, x->pq_codes.as<uint8_t>(), nbr ? x->pq_nbr.as<uint8_t>() : nullptr, x->pq_row, sh, nq, k, sg.ef, sg.ef_pad, rerank,
                                        sg.vis_words, counter, surv, surv_cnt, stats, x->w_visg.as<uint8_t>() + (size_t)region_base * x->vis_stride,
                                        (size_t)x->vis_stride, x->w_vepoch.as<uint32_t>() + region_base);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}
template <int METRIC, int QUANT>
int launch_pq_rerank(Hnsw* x, HCtx* c, const PqGeom& sg, uint32_t q0, uint32_t nq, uint32_t k, const uint32_t* surv, const uint32_t* surv_cnt,
                     unsigned long long* keys, uint64_t* oi, float* os, uint32_t* oc) {
  const dim3 grid(ceil_div(sg.ef_pad, 32), nq);
  const GraphView g = x->view();
  const float* qe = c->w_qeff.as<float>() + (size_t)q0 * x->dim; const float* qn = c->w_qn.as<float>() + q0;
  if (x->r8) hnsw_pq_rerank_kernel<METRIC, QUANT, true><<<grid, 64, 0, c->stream>>>(g, qe, qn, surv, surv_cnt, sg.ef_pad, keys);
  else hnsw_pq_rerank_kernel<METRIC, QUANT, false><<<grid, 64, 0, c->stream>>>(g, qe, qn, surv, surv_cnt, sg.ef_pad, keys);
  const size_t lds = (size_t)sg.ef_pad * 8;
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(hnsw_pq_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hnsw_pq_select_kernel<<<nq, 64, lds, c->stream>>>(keys, surv_cnt, sg.ef_pad, k, g.ids, oi + (size_t)q0 * k, os + (size_t)q0 * k, oc + q0);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// one attempt; *retry_hbm: the LDS hash would have needed its reset path — the caller runs the same call over the byte map
int pq_search_once(Hnsw* x, HCtx* c, const float* queries, bool on_device, size_t nq, uint32_t k, uint32_t ef_override, uint32_t rerank,
                   uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats, uint64_t* out_n_exact, bool force_hbm, bool* retry_hbm) {
  *retry_hbm = false;
  if (stats) std::memset(stats, 0, sizeof(*stats));
  if (out_n_exact) *out_n_exact = 0;
  if (nq == 0) return COLTT_OK;
  if (k == 0) return fail(COLTT_E_INVALID, "hnsw_pq_search: k must be >= 1");
  if (!x->pq_on) return fail(COLTT_E_INVALID, "hnsw_pq_search: the index carries no product-quantiser codes (coltt_hnsw_pq_attach)");
  if (x->pq_done != x->n) return fail(COLTT_E_DEVICE, "hnsw_pq_search: codes cover %llu of %llu slots", (unsigned long long)x->pq_done, (unsigned long long)x->n);
  uint64_t* d_oi = out_ids; float* d_os = out_scores; uint32_t* d_oc = out_counts;
  if (!on_device) {
    COLTT_TRY(c->w_out_ids.reserve(nq * k * 8)); COLTT_TRY(c->w_out_sc.reserve(nq * k * 4)); COLTT_TRY(c->w_out_cnt.reserve(nq * 4));
    d_oi = c->w_out_ids.as<uint64_t>(); d_os = c->w_out_sc.as<float>(); d_oc = c->w_out_cnt.as<uint32_t>();
  }
  if (x->entry < 0) {  // empty index => empty result, not an error (hnsw.go:249-251)
    COLTT_HIP(hipMemsetAsync(d_oc, 0, nq * 4, c->stream));
    if (!on_device) COLTT_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipStreamSynchronize(c->stream));
    return COLTT_OK;
  }
  const uint32_t ef = std::max<uint32_t>(ef_override ? ef_override : (uint32_t)x->cfg.ef, k);  // gomath.MaxInt(ef, k), hnsw.go:258
  if (ef > 4096) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_search: ef=%u > 4096", ef);
  if (wants_visg(ef) || force_hbm) COLTT_TRY(ensure_visg(x));
  PqGeom sg;
  bool have = pq_geom(x, ef, force_hbm, sg);
  if (!have && !force_hbm) { COLTT_TRY(ensure_visg(x)); have = pq_geom(x, ef, true, sg); }   // no room for the LDS hash beside the table: the byte map
  if (!have) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_search: dim %u / ef %u / %u sub-vectors need more than the CU's 160 KiB of LDS", x->dim, ef, x->pq_shape.m);
  uint32_t grid = (uint32_t)std::min<size_t>(nq, (size_t)256 * sg.per_cu);
  RegionLease lease;
  if (sg.variant != 0) { acquire_regions(x, grid, lease); grid = lease.count; }
  const float* d_q = queries;
  if (!on_device) {
    COLTT_TRY(c->w_qraw.reserve(nq * x->dim * 4));
    COLTT_HIP(hipMemcpyAsync(c->w_qraw.p, queries, nq * x->dim * 4, hipMemcpyHostToDevice, c->stream));
    d_q = c->w_qraw.as<float>();
  }
  COLTT_TRY(prep_queries_any(x, c, d_q, nq));
  bool nbr = false;
  if (sg.variant != 0) COLTT_TRY(ensure_pq_nbr(x, c->stream, &nbr));
  // tables for a group of queries at a time: [group][row_bytes][1 << shift] binary16 (4 KiB per query for 64 x 32), up to 256 MiB of them — a launch should
  // hold several queries per resident wave so that the work counter balances the tail (a 10 000-query call once ran as five 2 048-query launches on
  // 2 048 waves: every launch as long as its slowest traversal, profiles/r05i_bench_kernel_stats_by_grid.csv); at most 32 768 queries (the re-rank's grid.y)
  const uint32_t lsh = pq_lut_shift(x);
  const size_t lut_q = ((size_t)x->pq_row << lsh) * 2;
  // ... and the survivors' slots + exact keys (12 bytes per result-set entry and query) stay under 256 MiB as well (1.5 GB per pooled context at ef 4096)
  const size_t group = std::max<size_t>(1, std::min<size_t>({nq, (256ull << 20) / lut_q, (256ull << 20) / ((size_t)sg.ef_pad * 12), (size_t)32768}));
  COLTT_TRY(c->w_pack.reserve(group * lut_q + group * 4));   // the tables, then one word per query: its table's maximum (pq.hip: table scale)
  COLTT_TRY(c->w_surv.reserve(group * sg.ef_pad * 4)); COLTT_TRY(c->w_scnt.reserve(group * 4)); COLTT_TRY(c->w_keys.reserve(group * sg.ef_pad * 8));
  COLTT_TRY(c->w_misc.reserve(256));
  uint8_t* misc = c->w_misc.as<uint8_t>();
  uint32_t* counter = reinterpret_cast<uint32_t*>(misc);
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(misc + 16);
  COLTT_HIP(hipMemsetAsync(misc, 0, 256, c->stream));
  COLTT_HIP(hipEventRecord(c->ev0, c->stream));
  for (size_t q0 = 0; q0 < nq; q0 += group) {
    const size_t gn = std::min(group, nq - q0);
    COLTT_TRY(pq_lut16_batch(c->stream, x->pq_cb.as<float>(), x->pq_shape, c->w_qeff.as<float>() + q0 * x->dim, gn, x->pq_row, lsh,
                             reinterpret_cast<uint32_t*>(c->w_pack.as<uint8_t>() + group * lut_q), c->w_pack.as<unsigned short>()));
    if (q0) COLTT_HIP(hipMemsetAsync(counter, 0, 4, c->stream));
    COLTT_TRY(launch_pq_walk(x, c, sg, nbr, (uint32_t)std::min<size_t>(grid, gn), lease.base, c->w_pack.as<unsigned short>(), (uint32_t)gn, k, rerank, counter, c->w_surv.as<uint32_t>(),
                             c->w_scnt.as<uint32_t>(), d_stats));
    int rc;
#define COLTT_LP_ARGS x, c, sg, (uint32_t)q0, (uint32_t)gn, k, c->w_surv.as<uint32_t>(), c->w_scnt.as<uint32_t>(), c->w_keys.as<unsigned long long>(), d_oi, d_os, d_oc
#define COLTT_LP(Q) rc = x->metric == COLTT_COSINE ? launch_pq_rerank<M_COS, Q>(COLTT_LP_ARGS) : launch_pq_rerank<M_L2, Q>(COLTT_LP_ARGS)
    if (x->quant == COLTT_Q_NONE) { COLTT_LP(Q_NONE); } else { COLTT_LP(Q_F16); }
#undef COLTT_LP
#undef COLTT_LP_ARGS
    COLTT_TRY(rc);
  }
  COLTT_HIP(hipEventRecord(c->ev1, c->stream));
  if (x->dense && x->dense_base) add_base_kernel<<<ceil_div(nq * k, 256), 256, 0, c->stream>>>(d_oi, nq * k, x->dense_base);
  unsigned long long h_stats[5] = {0, 0, 0, 0, 0};
  if (!on_device) {
    COLTT_HIP(hipMemcpyAsync(out_ids, d_oi, nq * k * 8, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipMemcpyAsync(out_scores, d_os, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, c->stream));
  }
  COLTT_HIP(hipMemcpyAsync(h_stats, d_stats, 40, hipMemcpyDeviceToHost, c->stream));
#ifdef COLTT_PHASE_TIMING
  unsigned long long h_pt[8] = {0};
  COLTT_HIP(hipMemcpyAsync(h_pt, d_stats + 8, 64, hipMemcpyDeviceToHost, c->stream));
#endif
  COLTT_HIP(hipStreamSynchronize(c->stream));  // the lease (destructor) outlives the kernel
#ifdef COLTT_PHASE_TIMING
  {
    static const char* nm[8] = {"pop", "adjacency", "visited", "codes+table sum", "admission/evict/flush", "prologue(table, upper levels)", "flush+re-rank+writeout", "-"};
    double tot = 0; for (int i = 0; i < 7; i++) tot += (double)h_pt[i];
    fprintf(stderr, "[phase pq] nq=%zu ef=%u variant=%d per_cu=%u:", nq, sg.ef, sg.variant, sg.per_cu);
    for (int i = 0; i < 7; i++) fprintf(stderr, " %s %.1f%%", nm[i], 100.0 * (double)h_pt[i] / tot);
    fprintf(stderr, "  | ticks/query %.0f, per expansion %.0f\n", tot / (double)nq, tot / (double)std::max<unsigned long long>(1, h_stats[1]));
  }
#endif
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  x->last_ms.store(ms);
  if (sg.variant == 0 && (h_stats[4] & 8ull)) { *retry_hbm = true; return COLTT_OK; }
  if (h_stats[4]) return fail(COLTT_E_DEVICE, "hnsw_pq_search: traversal watchdog tripped (code %llu)", h_stats[4]);
  if (stats) { stats->n_dist = h_stats[0]; stats->n_exp = h_stats[1]; stats->n_hops = h_stats[2]; stats->n_visit_resets = 0; }
  if (out_n_exact) *out_n_exact = h_stats[3];
  return COLTT_OK;
}
int pq_search_common(Hnsw* x, HCtx* c, const float* queries, bool on_device, size_t nq, uint32_t k, uint32_t ef_override, uint32_t rerank,
                     uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats, uint64_t* out_n_exact) {
  bool retry = false;
  COLTT_TRY(pq_search_once(x, c, queries, on_device, nq, k, ef_override, rerank, out_ids, out_scores, out_counts, stats, out_n_exact, false, &retry));
  if (retry) COLTT_TRY(pq_search_once(x, c, queries, on_device, nq, k, ef_override, rerank, out_ids, out_scores, out_counts, stats, out_n_exact, true, &retry));
  return COLTT_OK;
}

// pruneNeighbors as Remove calls it (hnsw.go:234-236): rebuild the row from its non-deleted entries (<= width, so
// nothing is trimmed).  One thread per (neighbour, level) row.
__global__ void hnsw_unlink_kernel(GraphView g, const uint32_t* __restrict__ nbs, const int32_t* __restrict__ lvl, uint32_t n) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  uint32_t W;
  uint32_t* row = const_cast<uint32_t*>(adj_row(g, nbs[t], lvl[t], W));
  float* drow = lvl[t] == 0 ? g.adj0_d + (size_t)nbs[t] * g.mMax0 : g.adjU_d + ((size_t)g.upper_off[nbs[t]] + (uint32_t)(lvl[t] - 1)) * g.mMax;
  float* nrow = (lvl[t] == 0 && g.adj0_n) ? g.adj0_n + (size_t)nbs[t] * g.mMax0 : nullptr;
  uint32_t m = 0;
  for (uint32_t i = 0; i < W && row[i] != NBR_NONE; i++) {
    uint32_t sl = row[i]; float dd = drow[i];
    if (!is_deleted(g, sl)) { row[m] = sl; drow[m] = dd; if (nrow) nrow[m] = nrow[i]; m++; }
  }
  for (; m < W; m++) { row[m] = NBR_NONE; drow[m] = 0.f; if (nrow) nrow[m] = 0.f; }
}

int hnsw_undense(Hnsw* x) {
  if (!x->dense) return COLTT_OK;
  x->h_ids.resize(x->n);
  x->id2slot.reserve(x->n * 2);
  for (uint64_t s = 0; s < x->n; s++) {
    x->h_ids[s] = x->dense_base + s;
    if (!(x->h_del.size() > (s >> 5) && ((x->h_del[s >> 5] >> (s & 31)) & 1u))) x->id2slot[x->dense_base + s] = (uint32_t)s;
  }
  x->dense = false;
  COLTT_TRY(x->ids.reserve(std::max<uint64_t>(x->cap, 1024) * 8, false, x->stream));
  if (x->n) COLTT_HIP(hipMemcpyAsync(x->ids.p, x->h_ids.data(), x->n * 8, hipMemcpyHostToDevice, x->stream));
  COLTT_HIP(hipStreamSynchronize(x->stream));
  return COLTT_OK;
}

template <int METRIC, int QUANT>
int launch_build(Hnsw* x, const SearchGeom& sg, uint32_t base, uint32_t count, const int32_t* d_levels, uint32_t* counter,
                 uint32_t* req_count, BuildReq* req, uint32_t* head, unsigned long long* stats) {
  auto kern = sg.visg ? hnsw_build_search_kernel<METRIC, QUANT, true> : hnsw_build_search_kernel<METRIC, QUANT, false>;
  if constexpr (QUANT != Q_F8) { if (x->r8) kern = sg.visg ? hnsw_build_search_kernel<METRIC, QUANT, true, true> : hnsw_build_search_kernel<METRIC, QUANT, false, true>; }
  // algo 2 (diverse selection): a second query buffer + the chosen / rejected keys behind the walk's LDS
  const uint32_t diverse = x->cfg.algo == 2 ? (x->cfg.keep_pruned ? 3u : 1u) : 0u;
  const size_t ext_off = (sg.lds + 15) & ~(size_t)15;
  const size_t lds = diverse ? ext_off + (((size_t)x->dim * 4 + 15) & ~(size_t)15) + 2 * 64 * 8 : sg.lds;
  if (lds > 160 * 1024) return fail(COLTT_E_UNSUPPORTED, "hnsw insert: dim/efConstruction need %zu B of LDS", lds);
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(waves_per_cu_cap(QUANT), (160 * 1024) / lds));
  uint32_t grid = std::min<uint32_t>({count, 256u * per_cu, sg.max_grid});
  kern<<<grid, 64, lds, x->stream>>>(x->view(), x->entry, x->entry_level, base, count, d_levels, (uint32_t)x->cfg.m,
                                     sg.ef, sg.ef_pad, sg.hcap, x->cap, counter, req_count, req, head, stats,
                                     x->w_visg.as<uint8_t>(), (size_t)x->vis_stride, x->w_vepoch.as<uint32_t>(), diverse, (uint32_t)ext_off);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

template <int METRIC, int QUANT>
int launch_link_diverse(Hnsw* x, const BuildReq* req, uint32_t n_req, uint32_t* head, unsigned long long* stats) {
  auto kern = hnsw_link_diverse_kernel<METRIC, QUANT, false>;
  if constexpr (QUANT != Q_F8) { if (x->r8) kern = hnsw_link_diverse_kernel<METRIC, QUANT, true>; }
  const size_t lds = (((size_t)x->dim * 4 + 15) & ~(size_t)15) + (size_t)(2 * LINKD_CAND + 128) * 8;
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<n_req, 64, lds, x->stream>>>(x->view(), x->cap, req, n_req, head, (uint32_t)(x->cfg.keep_pruned ? 1 : 0), stats);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// n Inserts (hnsw.go:104-167) in order, `batch` at a time against a frozen graph.  d_vecs: raw vectors in HBM.
int insert_core(Hnsw* x, const uint64_t* ids, uint64_t first_id, const float* d_vecs, const int32_t* levels, size_t n,
                uint32_t batch) {
  if (n == 0) return COLTT_OK;
  if (batch == 0) batch = 1;
  if (x->cfg.m_max > LINK_W || x->cfg.m_max0 > LINK_W) return fail(COLTT_E_UNSUPPORTED, "hnsw insert: builder supports mMax, mMax0 <= %d", LINK_W);
  if (x->cfg.m > 64) return fail(COLTT_E_UNSUPPORTED, "hnsw insert: builder supports m <= 64");
  if (x->n + n >= 0x7fffffffull) return fail(COLTT_E_UNSUPPORTED, "hnsw insert: more than 2^31-1 slots");
  for (size_t i = 0; i < n; i++) if (levels[i] < 0 || levels[i] > 60) return fail(COLTT_E_INVALID, "hnsw insert: level %d out of range", levels[i]);
  bool dense_ok = !ids && x->dense && (x->n == 0 || first_id == x->dense_base + x->n);
  if (dense_ok) { if (x->n == 0) x->dense_base = first_id; }
  else {
    COLTT_TRY(hnsw_undense(x));
    std::unordered_map<uint64_t, int> seen;
    for (size_t i = 0; i < n; i++) {  // storeVertex: ItemAlreadyExistsError (hnsw.go:293-295), checked before any mutation
      uint64_t id = ids ? ids[i] : first_id + i;
      if (x->id2slot.count(id) || !seen.emplace(id, 1).second) return fail(COLTT_E_EXISTS, "Item already exists");
    }
  }
  const uint32_t efc = (uint32_t)x->cfg.ef_construction;
  COLTT_TRY(x->w_misc.reserve(64));
  uint32_t* counter = x->w_misc.as<uint32_t>();
  uint32_t* req_count = counter + 1;
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(x->w_misc.as<uint8_t>() + 16);
  size_t i = 0;
  uint32_t shrink = 0;   // diverse mode: a batch that would overflow a row's candidate list is retried at half the size (nothing was applied)
  while (i < n) {
    // nil entrypoint (empty index, or Remove left no successor: hnsw.go:197-217 may CAS the entrypoint to nil): the vertex is
    // stored at level 0 and becomes the entrypoint, nothing is linked (hnsw.go:108-116)
    const bool first = x->entry < 0;
    uint32_t b = first ? 1u : (uint32_t)std::min<size_t>(batch, n - i);
    if (shrink) b = std::min(b, shrink);
    uint64_t up = 0;
    for (uint32_t j = 0; j < b; j++) up += first ? 0 : (uint64_t)levels[i + j];
    const uint64_t base = x->n;
    COLTT_TRY(x->reserve(base + b, x->n_upper + up));
    if (wants_visg(efc)) COLTT_TRY(ensure_visg(x));  // (re)sized here: the slot capacity may just have grown
    const SearchGeom sg = search_geom(x, efc);
    if (sg.lds > 160 * 1024) return fail(COLTT_E_UNSUPPORTED, "hnsw insert: dim/efConstruction need %zu B of LDS", sg.lds);
    // per-slot tables of the batch.  Host mirrors (h_levels, h_upper_off, h_ids, id2slot, h_del) are committed only after
    // the device work of the batch has succeeded: a failed batch leaves the index exactly as it was (slots >= n unreferenced).
    std::vector<uint32_t> uo(b);
    std::vector<int32_t> lv(b);
    std::vector<uint64_t> nid(x->dense ? 0 : b);
    uint64_t u = x->n_upper;
    for (uint32_t j = 0; j < b; j++) {
      lv[j] = first ? 0 : levels[i + j];
      uo[j] = lv[j] > 0 ? (uint32_t)u : NBR_NONE;
      u += (uint64_t)lv[j];
      if (!x->dense) nid[j] = ids ? ids[i + j] : first_id + i + j;
    }
    auto commit_host = [&]() {
      for (uint32_t j = 0; j < b; j++) {
        x->h_levels.push_back(lv[j]);
        x->h_upper_off.push_back(uo[j]);
        if (!x->dense) { x->h_ids.push_back(nid[j]); x->id2slot[nid[j]] = (uint32_t)(base + j); }
      }
      x->h_del.resize((base + b + 31) / 32, 0u);
    };
    COLTT_HIP(hipMemcpyAsync(x->upper_off.as<uint32_t>() + base, uo.data(), (size_t)b * 4, hipMemcpyHostToDevice, x->stream));
    if (!x->dense) COLTT_HIP(hipMemcpyAsync(x->ids.as<uint64_t>() + base, nid.data(), (size_t)b * 8, hipMemcpyHostToDevice, x->stream));
    COLTT_HIP(hipMemsetAsync(x->adj0.as<uint32_t>() + base * x->cfg.m_max0, 0xff, (size_t)b * x->cfg.m_max0 * 4, x->stream));
    COLTT_HIP(hipMemsetAsync(x->adj0_d.as<float>() + base * x->cfg.m_max0, 0, (size_t)b * x->cfg.m_max0 * 4, x->stream));
    if (x->metric == COLTT_COSINE) COLTT_HIP(hipMemsetAsync(x->adj0_n.as<float>() + base * x->cfg.m_max0, 0, (size_t)b * x->cfg.m_max0 * 4, x->stream));
    if (up) {
      COLTT_HIP(hipMemsetAsync(x->adjU.as<uint32_t>() + x->n_upper * x->cfg.m_max, 0xff, (size_t)up * x->cfg.m_max * 4, x->stream));
      COLTT_HIP(hipMemsetAsync(x->adjU_d.as<float>() + x->n_upper * x->cfg.m_max, 0, (size_t)up * x->cfg.m_max * 4, x->stream));
    }
    COLTT_TRY(prep_rows_any(x, d_vecs + i * x->dim, b, base, x->metric == COLTT_COSINE));
    if (first) {
      COLTT_HIP(hipStreamSynchronize(x->stream));
      commit_host();
      x->entry = (int32_t)base; x->entry_level = 0;
      x->n += 1; x->live += 1; i += 1;
      continue;
    }
    // builder scratch: head[] over every adjacency row, request queue sized for the worst case
    uint64_t need_head = x->cap + x->ucap;
    if (need_head > x->head_cap) {
      COLTT_TRY(x->b_head.reserve(need_head * 4));
      fill_u32_kernel<<<ceil_div(need_head, 256), 256, 0, x->stream>>>(x->b_head.as<uint32_t>(), need_head, NBR_NONE);
      x->head_cap = need_head;
    }
    uint64_t max_req = 0;
    for (uint32_t j = 0; j < b; j++) max_req += (uint64_t)(std::min<int32_t>(lv[j], x->entry_level) + 1) * (uint64_t)x->cfg.m;
    COLTT_TRY(x->b_req.reserve(std::max<uint64_t>(max_req, 1) * sizeof(BuildReq)));
    COLTT_TRY(x->b_levels.reserve((size_t)b * 4));
    COLTT_HIP(hipMemcpyAsync(x->b_levels.p, lv.data(), (size_t)b * 4, hipMemcpyHostToDevice, x->stream));
    COLTT_HIP(hipMemsetAsync(x->w_misc.p, 0, 64, x->stream));
    int rc;
#define COLTT_LB_ARGS x, sg, (uint32_t)base, b, x->b_levels.as<int32_t>(), counter, req_count, x->b_req.as<BuildReq>(), x->b_head.as<uint32_t>(), d_stats
#define COLTT_LB(Q) rc = x->metric == COLTT_COSINE ? launch_build<M_COS, Q>(COLTT_LB_ARGS) : launch_build<M_L2, Q>(COLTT_LB_ARGS)
    COLTT_DISPATCH_QUANT(x->quant, COLTT_LB)
#undef COLTT_LB
#undef COLTT_LB_ARGS
    COLTT_TRY(rc);
    struct { uint32_t counter, n_req, pad0, pad1; unsigned long long st[6]; } hm;
    COLTT_HIP(hipMemcpyAsync(&hm, x->w_misc.p, sizeof(hm), hipMemcpyDeviceToHost, x->stream));
    COLTT_HIP(hipStreamSynchronize(x->stream));
    if (hm.st[4]) {  // phase A queued link requests on head[] before tripping: drop them so the next batch starts clean
      fill_u32_kernel<<<ceil_div(x->head_cap, 256), 256, 0, x->stream>>>(x->b_head.as<uint32_t>(), x->head_cap, NBR_NONE);
      (void)hipStreamSynchronize(x->stream);
      return fail(COLTT_E_DEVICE, "hnsw insert: traversal watchdog tripped (code %llu)", hm.st[4]);
    }
    if (hm.n_req && x->cfg.algo == 2) {
      // diverse mode: one wave per touched row; its evaluations are counted.  A row that would receive more candidates than the link kernel holds (a hub
      // chosen by > ~1 000 vertices of ONE batch) is found BEFORE anything is applied, and the batch is retried in halves: a smaller batch is another
      // legal schedule of the same Inserts, and the index is untouched by the attempt (the new slots' own rows are rewritten by the retry)
      hnsw_link_count_kernel<<<ceil_div(hm.n_req, 256), 256, 0, x->stream>>>(x->b_req.as<BuildReq>(), hm.n_req, x->b_head.as<uint32_t>(), x->cap, (uint32_t)x->cfg.m_max0,
                                                                            (uint32_t)x->cfg.m_max, d_stats);
      COLTT_HIP(hipMemcpyAsync(&hm, x->w_misc.p, sizeof(hm), hipMemcpyDeviceToHost, x->stream));
      COLTT_HIP(hipStreamSynchronize(x->stream));
      if (hm.st[5] > LINKD_CAND && b > 1) {
        fill_u32_kernel<<<ceil_div(x->head_cap, 256), 256, 0, x->stream>>>(x->b_head.as<uint32_t>(), x->head_cap, NBR_NONE);
        COLTT_HIP(hipStreamSynchronize(x->stream));
        shrink = std::max(1u, b / 2);
        continue;
      }
#define COLTT_LD_ARGS x, x->b_req.as<BuildReq>(), hm.n_req, x->b_head.as<uint32_t>(), d_stats
#define COLTT_LD(Q) rc = x->metric == COLTT_COSINE ? launch_link_diverse<M_COS, Q>(COLTT_LD_ARGS) : launch_link_diverse<M_L2, Q>(COLTT_LD_ARGS)
      COLTT_DISPATCH_QUANT(x->quant, COLTT_LD)
#undef COLTT_LD
#undef COLTT_LD_ARGS
      COLTT_TRY(rc);
      COLTT_HIP(hipMemcpyAsync(&hm, x->w_misc.p, sizeof(hm), hipMemcpyDeviceToHost, x->stream));
      COLTT_HIP(hipStreamSynchronize(x->stream));
      if (hm.st[4]) {
        fill_u32_kernel<<<ceil_div(x->head_cap, 256), 256, 0, x->stream>>>(x->b_head.as<uint32_t>(), x->head_cap, NBR_NONE);
        (void)hipStreamSynchronize(x->stream);
        return fail(COLTT_E_DEVICE, "hnsw insert (diverse selection): a row overflowed the link kernel's %u candidates behind the pre-check", LINKD_CAND);   // (cannot happen)
      }
      shrink = 0;
    } else if (hm.n_req) {
      hnsw_link_kernel<<<ceil_div(hm.n_req, 64), 64, 0, x->stream>>>(x->view(), x->cap, x->b_req.as<BuildReq>(), hm.n_req, x->b_head.as<uint32_t>());
      COLTT_HIP(hipGetLastError());
    }
    commit_host();
    x->build_stats.n_dist += hm.st[0]; x->build_stats.n_exp += hm.st[1]; x->build_stats.n_hops += hm.st[2]; x->build_stats.n_visit_resets += hm.st[3];
    for (uint32_t j = 0; j < b; j++)  // entrypoint CAS in insertion order (hnsw.go:161-164)
      if (lv[j] > x->entry_level) { x->entry = (int32_t)(base + j); x->entry_level = lv[j]; }
    x->n += b; x->live += b; x->n_upper += up; i += b;
  }
  COLTT_HIP(hipStreamSynchronize(x->stream));
  return sync_pq(x);
}

// the derived neighbour-norm rows of every slot (bulk installs; Insert / Remove maintain them incrementally in their kernels)
int fill_adj_norms(Hnsw* x) {
  if (x->metric != COLTT_COSINE || x->n == 0) return COLTT_OK;
  adj_norms_kernel<<<ceil_div(x->n * x->cfg.m_max0, 256), 256, 0, x->stream>>>(x->view(), 0, x->n);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipStreamSynchronize(x->stream));
  return COLTT_OK;
}

// A failed (re)load must not leave a half-installed index behind: fall back to the empty index (memory-safe, searchable).
void make_empty(Hnsw* x) {
  x->n = 0; x->live = 0; x->n_upper = 0; x->entry = -1; x->entry_level = 0; x->any_deleted = false; x->pq_done = 0;
  x->h_levels.clear(); x->h_upper_off.clear(); x->h_del.clear(); x->h_ids.clear(); x->id2slot.clear();
  x->dense = true; x->dense_base = 0;
}

// Installs graph topology + id tables (everything of a bulk load except the vectors) under configuration `c`.
// Transactional: the stream is validated and laid out on the host FIRST; the object (cfg included) is only touched once
// nothing but a device failure can go wrong, and a device failure leaves an EMPTY index, never a half-installed one.
int graph_install(Hnsw* x, const coltt_hnsw_cfg& c, uint64_t n, const uint64_t* ids, const int32_t* levels, const uint8_t* deleted,
                  const int64_t* row_offsets, const int32_t* nbr, const float* nbr_dist, int32_t entry_slot) {
  if (n >= 0x7fffffffull) return fail(COLTT_E_UNSUPPORTED, "hnsw_bulk_load: more than 2^31-1 slots");
  if (n && (entry_slot < -1 || entry_slot >= (int64_t)n)) return fail(COLTT_E_INVALID, "hnsw_bulk_load: entry slot out of range");
  const uint32_t W0 = (uint32_t)c.m_max0, WU = (uint32_t)c.m_max;
  uint64_t n_upper = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (levels[i] < 0 || levels[i] > 60) return fail(COLTT_E_INVALID, "hnsw_bulk_load: level %d out of range", levels[i]);
    n_upper += (uint64_t)levels[i];
  }
  std::vector<uint32_t> a0((size_t)n * W0, NBR_NONE), aU((size_t)n_upper * WU, NBR_NONE), uo(n, NBR_NONE);
  std::vector<float> d0((size_t)n * W0, 0.f), dU((size_t)n_upper * WU, 0.f);
  std::vector<uint32_t> h_del((n + 31) / 32, 0u);
  bool any_deleted = false; uint64_t live = 0;
  uint64_t row = 0, up = 0;
  std::vector<std::pair<uint32_t, float>> tmp;
  for (uint64_t i = 0; i < n; i++) {
    bool del = deleted && deleted[i];
    if (del) { h_del[i >> 5] |= 1u << (i & 31); any_deleted = true; } else live++;
    if (levels[i] > 0) { uo[i] = (uint32_t)up; }
    for (int l = 0; l <= levels[i]; l++, row++) {
      int64_t b = row_offsets[row], e = row_offsets[row + 1];
      uint32_t W = l == 0 ? W0 : WU;
      if (e < b || e - b > (int64_t)W) return fail(COLTT_E_INVALID, "hnsw_bulk_load: slot %llu level %d has %lld edges > width %u", (unsigned long long)i, l, (long long)(e - b), W);
      tmp.clear();
      for (int64_t j = b; j < e; j++) {
        if (nbr[j] < 0 || (uint64_t)nbr[j] >= n) return fail(COLTT_E_INVALID, "hnsw_bulk_load: neighbour slot out of range");
        if (levels[nbr[j]] < l) return fail(COLTT_E_INVALID, "hnsw_bulk_load: slot %llu level %d lists a neighbour of lower level", (unsigned long long)i, l);
        tmp.push_back({(uint32_t)nbr[j], nbr_dist ? nbr_dist[j] : 0.f});
      }
      std::sort(tmp.begin(), tmp.end());
      uint32_t* ar = l == 0 ? &a0[(size_t)i * W0] : &aU[(size_t)(up + l - 1) * WU];
      float* dr = l == 0 ? &d0[(size_t)i * W0] : &dU[(size_t)(up + l - 1) * WU];
      for (size_t j = 0; j < tmp.size(); j++) { ar[j] = tmp[j].first; dr[j] = tmp[j].second; }
    }
    up += (uint64_t)levels[i];
  }
  std::unordered_map<uint64_t, uint32_t> id2slot;
  if (ids) {
    id2slot.reserve(n * 2);
    for (uint64_t i = 0; i < n; i++)
      if (!(deleted && deleted[i]) && !id2slot.emplace(ids[i], (uint32_t)i).second) return fail(COLTT_E_INVALID, "hnsw_bulk_load: duplicate id");
  }
  // ---- from here on the object changes.  Buffers only grow; capacities are re-derived under the new row widths.
  const coltt_hnsw_cfg old_cfg = x->cfg;
  const uint64_t old_cap = x->cap, old_ucap = x->ucap;
  const bool old_dense = x->dense;
  x->cfg = c; x->cap = 0; x->ucap = 0; x->dense = (ids == nullptr);
  x->vis_stride = 0;  // the HBM visited workspace is sized by the slot capacity: re-made lazily
  int rc = x->reserve(n, n_upper);
  if (rc != COLTT_OK) { x->cfg = old_cfg; x->cap = old_cap; x->ucap = old_ucap; x->dense = old_dense; return rc; }  // nothing was written
  auto upload = [&]() -> int {
    if (!x->dense) COLTT_HIP(hipMemcpyAsync(x->ids.p, ids, n * 8, hipMemcpyHostToDevice, x->stream));
    if (n) {
      COLTT_HIP(hipMemcpyAsync(x->adj0.p, a0.data(), a0.size() * 4, hipMemcpyHostToDevice, x->stream));
      COLTT_HIP(hipMemcpyAsync(x->adj0_d.p, d0.data(), d0.size() * 4, hipMemcpyHostToDevice, x->stream));
      COLTT_HIP(hipMemcpyAsync(x->upper_off.p, uo.data(), n * 4, hipMemcpyHostToDevice, x->stream));
      COLTT_HIP(hipMemcpyAsync(x->del_bits.p, h_del.data(), h_del.size() * 4, hipMemcpyHostToDevice, x->stream));
      if (n_upper) {
        COLTT_HIP(hipMemcpyAsync(x->adjU.p, aU.data(), aU.size() * 4, hipMemcpyHostToDevice, x->stream));
        COLTT_HIP(hipMemcpyAsync(x->adjU_d.p, dU.data(), dU.size() * 4, hipMemcpyHostToDevice, x->stream));
      }
    }
    COLTT_HIP(hipStreamSynchronize(x->stream));
    return COLTT_OK;
  };
  rc = upload();
  x->id2slot.clear(); x->h_ids.clear(); x->dense_base = 0;
  if (rc != COLTT_OK) { make_empty(x); return rc; }  // device failure mid-install: leave an empty (safe) index
  x->h_levels.assign(levels, levels + n);
  x->h_upper_off = std::move(uo);
  x->h_del = std::move(h_del);
  x->any_deleted = any_deleted; x->live = live;
  if (!x->dense) { x->h_ids.assign(ids, ids + n); x->id2slot = std::move(id2slot); }
  x->n = n; x->n_upper = n_upper; x->pq_done = 0;   // every row is about to be rewritten (the callers upload the vectors next)
  x->entry = n ? entry_slot : -1;
  x->entry_level = x->entry >= 0 ? levels[x->entry] : 0;
  return COLTT_OK;
}

}  // namespace

extern "C" {

int coltt_hnsw_create(uint32_t dim, int metric, int quant, const coltt_hnsw_cfg* cfg, coltt_handle_t* out) {
  return coltt::hnsw_create_on(-1 /* the process default device, selected after the arguments have been validated */, dim, metric, quant, cfg, out);
}

}  // extern "C"

// NewHnsw on an explicit device (collection groups place one shard per GPU)
int coltt::hnsw_create_on(int device, uint32_t dim, int metric, int quant, const coltt_hnsw_cfg* cfg, coltt_handle_t* out) {
  if (!out) return fail(COLTT_E_INVALID, "hnsw_create: out is NULL");
  if (dim == 0 || dim > 8192) return fail(COLTT_E_INVALID, "hnsw_create: dim %u outside [1,8192]", dim);
  if (metric != COLTT_COSINE && metric != COLTT_EUCLIDEAN) return fail(COLTT_E_INVALID, "hnsw_create: bad metric %d", metric);
  if (quant < COLTT_Q_NONE || quant > COLTT_Q_BF16) return fail(COLTT_E_UNSUPPORTED, "not support quantization type");
  auto x = std::make_shared<Hnsw>();
  x->dim = dim; x->metric = metric; x->quant = quant;
  x->stride = ((size_t)dim * quant_bytes(quant) + 15) & ~(size_t)15;
  coltt_hnsw_cfg c = cfg ? *cfg : coltt_hnsw_cfg{16, -1, -1, 20, 200, 0, -1.f, 0, 1};
  if (c.m <= 0) return fail(COLTT_E_INVALID, "hnsw_create: m must be > 0");
  // newHnswConfig defaults (hnsw_config.go:150-160)
  if (c.level_multiplier == -1.f) c.level_multiplier = 1.0f / (float)std::log((double)(float)c.m);
  if (c.m_max == -1) c.m_max = c.m;
  if (c.m_max0 == -1) c.m_max0 = 2 * c.m;
  if (c.algo != 0 && c.algo != 1 && c.algo != COLTT_HNSW_DIVERSE) return fail(COLTT_E_INVALID, "hnsw_create: unknown search algorithm %d", c.algo);
  if (c.algo != 0 && c.extend_candidates)
    return fail(COLTT_E_UNSUPPORTED, "hnsw_create: HeuristicExtendCandidates is undefined behaviour in the reference "
                                     "(priority_queue.go:109-122 aliases the heap array); rejected");
  if (c.m_max < c.m || c.m_max0 < c.m || c.m_max0 > 1024) return fail(COLTT_E_INVALID, "hnsw_create: need m <= mMax, m <= mMax0 <= 1024");
  if (c.ef <= 0 || c.ef_construction <= 0) return fail(COLTT_E_INVALID, "hnsw_create: ef and efConstruction must be > 0");
  x->cfg = c;
  x->r8 = rows8_shape(dim, quant);
  COLTT_DEVICE(device); device = coltt_dev_scope_.device();
  x->device = device;
  COLTT_HIP(hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking));
  *out = Registry::get().add(x);
  return COLTT_OK;
}

extern "C" {

int coltt_hnsw_destroy(coltt_handle_t h) {
  if (!Registry::get().erase(h)) return fail(COLTT_E_NOT_FOUND, "hnsw_destroy: unknown handle");
  return COLTT_OK;
}

int coltt_hnsw_get_cfg(coltt_handle_t h, coltt_hnsw_cfg* out) {
  auto x = lookup<Hnsw>(h);
  if (!x || !out) return fail(COLTT_E_NOT_FOUND, "hnsw_get_cfg: unknown handle");
  ReadLock g(x->rw);
  *out = x->cfg;
  return COLTT_OK;
}

int coltt_hnsw_random_level(coltt_handle_t h, float u, int32_t* out_level) {
  auto x = lookup<Hnsw>(h);
  if (!x || !out_level) return fail(COLTT_E_NOT_FOUND, "hnsw_random_level: unknown handle");
  ReadLock g(x->rw);
  if (!(u > 0.f && u < 1.f)) return fail(COLTT_E_INVALID, "hnsw_random_level: u must be in (0,1)");
  const float lg = (float)std::log((double)u);       // gomath.Log
  const float v = -lg * x->cfg.level_multiplier;      // RandomExponential (f32 multiply)
  *out_level = (int32_t)std::floor((double)v);        // gomath.Floor
  return COLTT_OK;
}

int coltt_hnsw_len(coltt_handle_t h, uint64_t* out) {
  auto x = lookup<Hnsw>(h);
  if (!x || !out) return fail(COLTT_E_NOT_FOUND, "hnsw_len: unknown handle");
  ReadLock g(x->rw);
  *out = x->live;
  return COLTT_OK;
}

int coltt_hnsw_bulk_load(coltt_handle_t h, uint64_t n, const uint64_t* ids, const int32_t* levels, const uint8_t* deleted,
                         const float* vectors, const int64_t* row_offsets, const int32_t* nbr, const float* nbr_dist,
                         int32_t entry_slot) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_bulk_load: unknown handle");
  if (n && (!levels || !vectors || !row_offsets)) return fail(COLTT_E_INVALID, "hnsw_bulk_load: NULL input");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  x->pq_nbr_stale.store(true);
  COLTT_TRY(graph_install(x.get(), x->cfg, n, ids, levels, deleted, row_offsets, nbr, nbr_dist, entry_slot));
  auto upload_vectors = [&]() -> int {
    // vectors in chunks through a staging buffer: Normalize (cosine) + Lower, as Insert does (hnsw.go:105-107)
    const uint64_t chunk = std::max<uint64_t>(1, (256ull << 20) / ((uint64_t)x->dim * 4));
    COLTT_TRY(x->w_raw.reserve(std::min<uint64_t>(chunk, n) * x->dim * 4));
    for (uint64_t b = 0; b < n; b += chunk) {
      uint64_t m = std::min<uint64_t>(chunk, n - b);
      COLTT_HIP(hipMemcpyAsync(x->w_raw.p, vectors + b * x->dim, m * x->dim * 4, hipMemcpyHostToDevice, x->stream));
      COLTT_TRY(prep_rows_any(x.get(), x->w_raw.as<float>(), m, b, x->metric == COLTT_COSINE));
      COLTT_HIP(hipStreamSynchronize(x->stream));
    }
    return COLTT_OK;
  };
  if (n) { int rc = upload_vectors(); if (rc == COLTT_OK) rc = fill_adj_norms(x.get()); if (rc == COLTT_OK) rc = sync_pq(x.get()); if (rc != COLTT_OK) { make_empty(x.get()); return rc; } }
  return COLTT_OK;
}

// ---- Hnsw.Commit / Hnsw.Load (core/vectorindex/hnsw_commit.go:69-278) ------------------------------------------------
namespace {
struct BEW {
  uint8_t* p; uint64_t cap, n = 0;
  void put(const void* s, size_t k) { if (p && n + k <= cap) std::memcpy(p + n, s, k); n += k; }
  void u8(uint8_t v) { put(&v, 1); }
  void u16(uint16_t v) { uint8_t b[2] = {(uint8_t)(v >> 8), (uint8_t)v}; put(b, 2); }
  void u32(uint32_t v) { uint8_t b[4]; for (int i = 0; i < 4; i++) b[i] = (uint8_t)(v >> (8 * (3 - i))); put(b, 4); }
  void u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * (7 - i))); put(b, 8); }
  void f32(float f) { uint32_t u; std::memcpy(&u, &f, 4); u32(u); }
};
struct BER {
  const uint8_t* p; uint64_t n, i = 0; bool ok = true;
  bool need(uint64_t k) { if (i + k > n) { ok = false; return false; } return true; }
  uint8_t u8() { if (!need(1)) return 0; return p[i++]; }
  uint16_t u16() { if (!need(2)) return 0; uint16_t v = (uint16_t)((p[i] << 8) | p[i + 1]); i += 2; return v; }
  uint32_t u32() { if (!need(4)) return 0; uint32_t v = 0; for (int k = 0; k < 4; k++) v = (v << 8) | p[i + k]; i += 4; return v; }
  uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | p[i + k]; i += 8; return v; }
  float f32() { uint32_t u = u32(); float f; std::memcpy(&f, &u, 4); return f; }
};
// big-endian f32 vectors at arbitrary byte offsets of the uploaded stream chunk -> f32 staging rows (Vector.Load, edge/constants.go:115-122)
__global__ void be_rows_kernel(const uint8_t* __restrict__ chunk, const uint64_t* __restrict__ offs, uint64_t m, int dim, float* __restrict__ out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * (uint64_t)dim) return;
  uint64_t i = t / dim; int e = (int)(t - i * dim);
  const uint8_t* s = chunk + offs[i] + (size_t)e * 4;
  uint32_t u = ((uint32_t)s[0] << 24) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 8) | s[3];
  out[t] = __uint_as_float(u);
}
// stored 2-byte codes -> the f32 values they stand for (exact), for the f32 vertex section of Commit
__global__ void decode_rows16_kernel(const uint8_t* __restrict__ rows, size_t stride, uint64_t first, uint64_t m, int dim, float* __restrict__ out) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * (uint64_t)dim) return;
  uint64_t i = t / dim; int e = (int)(t - i * dim);
  out[t] = dev::f16bits_to_f32(*reinterpret_cast<const unsigned short*>(rows + (first + i) * stride + (size_t)e * 2));
}
}  // namespace

int coltt_hnsw_load(coltt_handle_t h, int header, const uint8_t* buf, uint64_t len, uint64_t* out_n, uint64_t* out_ids,
                    uint64_t* out_meta_off, uint32_t* out_meta_len, uint64_t cap_n) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_load: unknown handle");
  if (!buf && len) return fail(COLTT_E_INVALID, "hnsw_load: NULL buffer");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  x->pq_nbr_stale.store(true);
  BER r{buf, len};
  coltt_hnsw_cfg c = x->cfg;  // committed by graph_install only after the whole stream has been parsed and validated
  if (header) {  // hnswConfig.load (hnsw_config.go:205-245), dim, distIdx (hnsw_commit.go:165-183)
    c.algo = (int32_t)r.u32(); c.level_multiplier = r.f32(); c.ef = (int32_t)r.u32(); c.ef_construction = (int32_t)r.u32();
    c.m = (int32_t)r.u32(); c.m_max = (int32_t)r.u32(); c.m_max0 = (int32_t)r.u32();
    uint32_t dim = r.u32(); uint8_t di = r.u8();
    if (!r.ok) return fail(COLTT_E_INVALID, "hnsw_load: truncated header");
    if (di != 1 && di != 2) return fail(COLTT_E_INVALID, "Invalid space type");  // InvalidSpaceTypeErr
    if (dim != x->dim) return fail(COLTT_E_INVALID, "hnsw_load: stream dim %u != index dim %u", dim, x->dim);
    if ((di == 1) != (x->metric == COLTT_COSINE)) return fail(COLTT_E_INVALID, "hnsw_load: stream distance differs from the index's");
    if (c.m <= 0 || c.m_max < c.m || c.m_max0 < c.m || c.m_max0 > 1024 || c.ef <= 0 || c.ef_construction <= 0 || (c.algo != 0 && c.algo != 1 && c.algo != COLTT_HNSW_DIVERSE))
      return fail(COLTT_E_INVALID, "hnsw_load: invalid config in stream");
  }
  std::vector<uint64_t> ids, voff, moff; std::vector<int32_t> levels; std::vector<uint32_t> mlen;
  uint64_t entry_id = 0; bool empty = r.i >= len;
  if (!empty) {
    entry_id = r.u64();
    for (int s = 0; s < 16 && r.ok; s++) {
      uint32_t cnt = r.u32();
      for (uint32_t i = 0; i < cnt && r.ok; i++) {
        ids.push_back(r.u64()); levels.push_back((int32_t)r.u32());
        voff.push_back(r.i); r.need((uint64_t)x->dim * 4); r.i += (uint64_t)x->dim * 4;
        uint64_t m0 = r.i; uint16_t pairs = r.u16();
        for (uint16_t p = 0; p < pairs && r.ok; p++) { uint8_t kl = r.u8(); r.need(kl); r.i += kl; uint16_t vl = r.u16(); r.need(vl); r.i += vl; }
        moff.push_back(m0); mlen.push_back((uint32_t)(r.i - m0));
        if (levels.back() < 0 || levels.back() > 60) return fail(COLTT_E_INVALID, "hnsw_load: bad level in stream");
      }
    }
  }
  if (!r.ok) return fail(COLTT_E_INVALID, "hnsw_load: truncated vertex section");
  const uint64_t n = ids.size();
  std::unordered_map<uint64_t, int32_t> slot; slot.reserve(n * 2);
  for (uint64_t i = 0; i < n; i++) if (!slot.emplace(ids[i], (int32_t)i).second) return fail(COLTT_E_INVALID, "hnsw_load: duplicate id in stream");
  int32_t entry = -1;
  if (n) { auto it = slot.find(entry_id); if (it == slot.end()) return fail(COLTT_E_INVALID, "hnsw_load: entrypoint id not in stream"); entry = it->second; }
  // edges -> per (slot, level) lists, then CSR in slot-major / level-minor order
  std::vector<int64_t> row_of(n + 1, 0);
  for (uint64_t i = 0; i < n; i++) row_of[i + 1] = row_of[i] + levels[i] + 1;
  std::vector<std::vector<std::pair<int32_t, float>>> rows((size_t)row_of[n]);
  for (uint64_t k = 0; k < n && r.ok; k++) {
    uint64_t id = r.u64();
    auto it = slot.find(id);
    if (!r.ok || it == slot.end()) return fail(COLTT_E_INVALID, "hnsw_load: edge record for an unknown vertex");
    int32_t s = it->second;
    for (int l = levels[s]; l >= 0; l--) {
      uint32_t c = r.u32();
      auto& lst = rows[(size_t)row_of[s] + l];
      for (uint32_t j = 0; j < c && r.ok; j++) {
        uint64_t nid = r.u64(); float d = r.f32();
        auto nit = slot.find(nid);
        if (nit == slot.end()) return fail(COLTT_E_INVALID, "hnsw_load: edge to an unknown vertex");
        lst.push_back({nit->second, d});
      }
    }
  }
  if (!r.ok) return fail(COLTT_E_INVALID, "hnsw_load: truncated edge section");
  std::vector<int64_t> offs(rows.size() + 1, 0);
  for (size_t i = 0; i < rows.size(); i++) offs[i + 1] = offs[i] + (int64_t)rows[i].size();
  std::vector<int32_t> nbr((size_t)offs.back()); std::vector<float> nd((size_t)offs.back());
  for (size_t i = 0; i < rows.size(); i++)
    for (size_t j = 0; j < rows[i].size(); j++) { nbr[(size_t)offs[i] + j] = rows[i][j].first; nd[(size_t)offs[i] + j] = rows[i][j].second; }
  COLTT_TRY(graph_install(x.get(), c, n, n ? ids.data() : nullptr, levels.data(), nullptr, offs.data(), nbr.data(), nd.data(), entry));
  // vectors: upload the stream in chunks, byte-swap on the device; stored vectors are NOT re-normalised (hnsw_commit.go:217-220)
  auto upload_vectors = [&]() -> int {
    const uint64_t rows_per = std::max<uint64_t>(1, (128ull << 20) / ((uint64_t)x->dim * 4));
    DevBuf d_chunk, d_offs;
    for (uint64_t b = 0; b < n; b += rows_per) {
      uint64_t m = std::min<uint64_t>(rows_per, n - b);
      uint64_t lo = voff[b], hi = voff[b + m - 1] + (uint64_t)x->dim * 4;
      std::vector<uint64_t> rel(m);
      for (uint64_t i = 0; i < m; i++) rel[i] = voff[b + i] - lo;
      COLTT_TRY(d_chunk.reserve(hi - lo)); COLTT_TRY(d_offs.reserve(m * 8)); COLTT_TRY(x->w_raw.reserve(m * x->dim * 4));
      COLTT_HIP(hipMemcpyAsync(d_chunk.p, buf + lo, hi - lo, hipMemcpyHostToDevice, x->stream));
      COLTT_HIP(hipMemcpyAsync(d_offs.p, rel.data(), m * 8, hipMemcpyHostToDevice, x->stream));
      be_rows_kernel<<<ceil_div(m * x->dim, 256), 256, 0, x->stream>>>(d_chunk.as<uint8_t>(), d_offs.as<uint64_t>(), m, (int)x->dim, x->w_raw.as<float>());
      COLTT_TRY(prep_rows_any(x.get(), x->w_raw.as<float>(), m, b, false));
      COLTT_HIP(hipStreamSynchronize(x->stream));
    }
    return COLTT_OK;
  };
  if (n) { int rc = upload_vectors(); if (rc == COLTT_OK) rc = fill_adj_norms(x.get()); if (rc == COLTT_OK) rc = sync_pq(x.get()); if (rc != COLTT_OK) { make_empty(x.get()); return rc; } }
  if (out_n) *out_n = n;
  for (uint64_t i = 0; i < n && i < cap_n; i++) {
    if (out_ids) out_ids[i] = ids[i];
    if (out_meta_off) out_meta_off[i] = moff[i];
    if (out_meta_len) out_meta_len[i] = mlen[i];
  }
  return COLTT_OK;
}

int coltt_hnsw_entry_level(coltt_handle_t h, int32_t* out_level) {
  auto x = lookup<Hnsw>(h);
  if (!x || !out_level) return fail(COLTT_E_NOT_FOUND, "hnsw_entry_level: unknown handle");
  ReadLock g(x->rw);
  *out_level = x->entry >= 0 ? x->entry_level : -1;
  return COLTT_OK;
}

int coltt_hnsw_commit(coltt_handle_t h, int header, const uint8_t* const* meta_blobs, const uint32_t* meta_lens, uint64_t n_meta,
                      uint8_t* out, uint64_t cap, uint64_t* out_len) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_commit: unknown handle");
  if (!out_len) return fail(COLTT_E_INVALID, "hnsw_commit: out_len is NULL");
  // The reference stream stores f32 vectors (hnsw_commit.go:100-112).  A binary16 ("f16" / "bf16") index writes the values its codes stand
  // for — exact — and Load into an index of the same quantisation encodes them back to the same codes (encode(decode(c)) == c for every
  // non-NaN binary16 code; tests/test_oracle.py checks all 65 536), so the round trip is bit-identical and the stream stays readable by
  // the reference.  The "f8" codec is not idempotent (decode -> encode -> decode changes 128 of its 256 codes): no stream can carry it.
  if (x->quant == COLTT_Q_F8) return fail(COLTT_E_UNSUPPORTED, "hnsw_commit: the reference stream stores f32 vectors and the f8 codec does not round-trip through them; f8 indexes are not committable");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  BEW w{out, out ? cap : 0};
  if (header) {  // hnswConfig.save (hnsw_config.go:179-203) + dim + distIdx (hnsw_commit.go:70-80)
    w.u32((uint32_t)x->cfg.algo); w.f32(x->cfg.level_multiplier); w.u32((uint32_t)x->cfg.ef); w.u32((uint32_t)x->cfg.ef_construction);
    w.u32((uint32_t)x->cfg.m); w.u32((uint32_t)x->cfg.m_max); w.u32((uint32_t)x->cfg.m_max0);
    w.u32(x->dim); w.u8(x->metric == COLTT_COSINE ? 1 : 2);
  }
  if (x->live == 0) { *out_len = w.n; return (out && w.n > cap) ? fail(COLTT_E_INVALID, "hnsw_commit: buffer too small") : COLTT_OK; }
  if (x->entry < 0) return fail(COLTT_E_INVALID, "hnsw_commit: no entrypoint");  // NoEntrypointErr
  const uint64_t n = x->n;
  const uint32_t W0 = (uint32_t)x->cfg.m_max0, WU = (uint32_t)x->cfg.m_max;
  auto id_of = [&](uint64_t s) { return x->dense ? x->dense_base + s : x->h_ids[s]; };
  auto dead = [&](uint64_t s) { return (x->h_del[s >> 5] >> (s & 31)) & 1u; };
  std::vector<std::vector<uint32_t>> shards(16);
  for (uint64_t s = 0; s < n; s++) if (!dead(s)) shards[shard_vertex(id_of(s), 16)].push_back((uint32_t)s);
  w.u64(id_of((uint64_t)x->entry));
  // section 1: vertices (rows fetched from HBM in blocks)
  std::vector<float> rowbuf;
  const uint64_t blk = std::max<uint64_t>(1, (64ull << 20) / ((uint64_t)x->dim * 4));
  std::vector<float> all;  // sized only when actually writing
  if (out) { all.resize(n * (size_t)x->dim);
    DevBuf d_dec;
    for (uint64_t b = 0; b < n; b += blk) { uint64_t m = std::min<uint64_t>(blk, n - b);
      if (x->quant == COLTT_Q_NONE && !x->r8)
        COLTT_HIP(hipMemcpy2D(all.data() + b * x->dim, (size_t)x->dim * 4, x->rows.as<uint8_t>() + b * x->stride, x->stride, (size_t)x->dim * 4, m, hipMemcpyDeviceToHost));
      else {
        COLTT_TRY(d_dec.reserve(m * x->dim * 4));
        if (x->quant == COLTT_Q_NONE) rows_to_f32_kernel<Q_NONE, true><<<ceil_div(m * x->dim, 256), 256, 0, nullptr>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, d_dec.as<float>());
        else if (x->r8) rows_to_f32_kernel<Q_F16, true><<<ceil_div(m * x->dim, 256), 256, 0, nullptr>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, d_dec.as<float>());
        else decode_rows16_kernel<<<ceil_div(m * x->dim, 256), 256, 0, nullptr>>>(x->rows.as<uint8_t>(), x->stride, b, m, (int)x->dim, d_dec.as<float>());
        COLTT_HIP(hipGetLastError());
        COLTT_HIP(hipMemcpy(all.data() + b * x->dim, d_dec.p, m * x->dim * 4, hipMemcpyDeviceToHost));
      } } }
  for (auto& sh : shards) {
    w.u32((uint32_t)sh.size());
    for (uint32_t s : sh) {
      w.u64(id_of(s)); w.u32((uint32_t)x->h_levels[s]);
      if (out) for (uint32_t e = 0; e < x->dim; e++) w.f32(all[(size_t)s * x->dim + e]); else w.n += (uint64_t)x->dim * 4;
      if (meta_blobs && meta_lens && s < n_meta && meta_blobs[s] && meta_lens[s] >= 2) w.put(meta_blobs[s], meta_lens[s]); else w.u16(0);
    }
  }
  // section 2: edges, highest level first, tombstoned neighbours skipped (hnsw_commit.go:133-157)
  std::vector<uint32_t> a0((size_t)n * W0), aU((size_t)x->n_upper * WU); std::vector<float> d0(a0.size()), dU(aU.size());
  if (n) { COLTT_HIP(hipMemcpy(a0.data(), x->adj0.p, a0.size() * 4, hipMemcpyDeviceToHost)); COLTT_HIP(hipMemcpy(d0.data(), x->adj0_d.p, d0.size() * 4, hipMemcpyDeviceToHost)); }
  if (x->n_upper) { COLTT_HIP(hipMemcpy(aU.data(), x->adjU.p, aU.size() * 4, hipMemcpyDeviceToHost)); COLTT_HIP(hipMemcpy(dU.data(), x->adjU_d.p, dU.size() * 4, hipMemcpyDeviceToHost)); }
  for (auto& sh : shards)
    for (uint32_t s : sh) {
      w.u64(id_of(s));
      for (int l = x->h_levels[s]; l >= 0; l--) {
        uint32_t W = l == 0 ? W0 : WU;
        const uint32_t* row = l == 0 ? &a0[(size_t)s * W0] : &aU[((size_t)x->h_upper_off[s] + l - 1) * WU];
        const float* dr = l == 0 ? &d0[(size_t)s * W0] : &dU[((size_t)x->h_upper_off[s] + l - 1) * WU];
        uint32_t c = 0;
        for (uint32_t j = 0; j < W && row[j] != NBR_NONE; j++) if (!dead(row[j])) c++;
        w.u32(c);
        for (uint32_t j = 0; j < W && row[j] != NBR_NONE; j++) { if (dead(row[j])) continue; w.u64(id_of(row[j])); w.f32(dr[j]); }
      }
    }
  *out_len = w.n;
  if (out && w.n > cap) return fail(COLTT_E_INVALID, "hnsw_commit: buffer of %llu bytes is too small for %llu", (unsigned long long)cap, (unsigned long long)w.n);
  return COLTT_OK;
}

int coltt_hnsw_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, uint32_t ef_override,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_search: unknown handle");
  if (nq && (!queries || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "hnsw_search: NULL buffer");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  CtxLease<HCtx> ctx(x->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return search_common(x.get(), ctx.c, queries, false, nq, k, ef_override, out_ids, out_scores, out_counts, stats);
}

int coltt_hnsw_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, uint32_t ef_override,
                             uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_counts, coltt_hnsw_stats* stats) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_search_device: unknown handle");
  if (nq && (!d_queries || !d_out_ids || !d_out_scores || !d_out_counts)) return fail(COLTT_E_INVALID, "hnsw_search_device: NULL buffer");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  CtxLease<HCtx> ctx(x->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return search_common(x.get(), ctx.c, d_queries, true, nq, k, ef_override, d_out_ids, d_out_scores, d_out_counts, stats);
}


int coltt_hnsw_insert(coltt_handle_t h, uint64_t id, const float* vec, int32_t level) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_insert: unknown handle");
  if (!vec) return fail(COLTT_E_INVALID, "hnsw_insert: NULL vector");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  x->pq_nbr_stale.store(true);
  COLTT_TRY(x->w_raw.reserve((size_t)x->dim * 4));
  COLTT_HIP(hipMemcpyAsync(x->w_raw.p, vec, (size_t)x->dim * 4, hipMemcpyHostToDevice, x->stream));
  return insert_core(x.get(), &id, 0, x->w_raw.as<float>(), &level, 1, 1);
}

int coltt_hnsw_insert_batch_device(coltt_handle_t h, const uint64_t* ids, uint64_t first_id, const float* d_vecs,
                                   const int32_t* levels, size_t n, uint32_t batch) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_insert_batch_device: unknown handle");
  if (n && (!d_vecs || !levels)) return fail(COLTT_E_INVALID, "hnsw_insert_batch_device: NULL input");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  x->pq_nbr_stale.store(true);
  return insert_core(x.get(), ids, first_id, d_vecs, levels, n, batch);
}

int coltt_hnsw_remove(coltt_handle_t h, uint64_t id) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_remove: unknown handle");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  x->pq_nbr_stale.store(true);   // the neighbours' level-0 rows are about to be re-pruned
  uint32_t vi;
  if (x->dense) {
    if (id < x->dense_base || id >= x->dense_base + x->n) return fail(COLTT_E_NOT_FOUND, "Item not found");
    vi = (uint32_t)(id - x->dense_base);
    if ((x->h_del[vi >> 5] >> (vi & 31)) & 1u) return fail(COLTT_E_NOT_FOUND, "Item not found");
  } else {
    auto it = x->id2slot.find(id);
    if (it == x->id2slot.end()) return fail(COLTT_E_NOT_FOUND, "Item not found");  // ItemNotFoundError (hnsw.go:317)
    vi = it->second;
    x->id2slot.erase(it);
  }
  // removeVertex: delete from the map, set the deleted flag (hnsw.go:304-318)
  x->h_del[vi >> 5] |= 1u << (vi & 31);
  x->any_deleted = true; x->live--;
  COLTT_HIP(hipMemcpyAsync(x->del_bits.as<uint32_t>() + (vi >> 5), &x->h_del[vi >> 5], 4, hipMemcpyHostToDevice, x->stream));
  // the removed vertex's own rows
  const int L = x->h_levels[vi];
  const uint32_t W0 = (uint32_t)x->cfg.m_max0, WU = (uint32_t)x->cfg.m_max;
  std::vector<std::vector<uint32_t>> nb(L + 1);
  std::vector<std::vector<float>> nd(L + 1);
  for (int l = 0; l <= L; l++) {
    uint32_t W = l == 0 ? W0 : WU;
    nb[l].resize(W); nd[l].resize(W);
    const uint32_t* src = l == 0 ? x->adj0.as<uint32_t>() + (size_t)vi * W0 : x->adjU.as<uint32_t>() + ((size_t)x->h_upper_off[vi] + l - 1) * WU;
    const float* dsrc = l == 0 ? x->adj0_d.as<float>() + (size_t)vi * W0 : x->adjU_d.as<float>() + ((size_t)x->h_upper_off[vi] + l - 1) * WU;
    COLTT_HIP(hipMemcpyAsync(nb[l].data(), src, W * 4, hipMemcpyDeviceToHost, x->stream));
    COLTT_HIP(hipMemcpyAsync(nd[l].data(), dsrc, W * 4, hipMemcpyDeviceToHost, x->stream));
  }
  COLTT_HIP(hipStreamSynchronize(x->stream));
  if (x->entry == (int32_t)vi) {  // new entrypoint: closest neighbour on the highest level that has any (hnsw.go:197-217)
    float minD = 3.40282346638528859811704183484516925440e+38f;
    int32_t closest = -1;
    for (int l = L; l >= 0; l--) {
      for (size_t j = 0; j < nb[l].size() && nb[l][j] != NBR_NONE; j++)
        if (nd[l][j] < minD) { minD = nd[l][j]; closest = (int32_t)nb[l][j]; }
      if (closest >= 0) break;
    }
    x->entry = closest;
    x->entry_level = closest >= 0 ? x->h_levels[closest] : 0;
  }
  // every neighbour drops the back-edge and is re-pruned (hnsw.go:219-238)
  std::vector<uint32_t> t_nb; std::vector<int32_t> t_lv;
  for (int l = L; l >= 0; l--)
    for (size_t j = 0; j < nb[l].size() && nb[l][j] != NBR_NONE; j++) { t_nb.push_back(nb[l][j]); t_lv.push_back(l); }
  if (!t_nb.empty()) {
    COLTT_TRY(x->b_req.reserve(t_nb.size() * 8));
    uint32_t* d_nb = x->b_req.as<uint32_t>();
    int32_t* d_lv = reinterpret_cast<int32_t*>(d_nb + t_nb.size());
    COLTT_HIP(hipMemcpyAsync(d_nb, t_nb.data(), t_nb.size() * 4, hipMemcpyHostToDevice, x->stream));
    COLTT_HIP(hipMemcpyAsync(d_lv, t_lv.data(), t_lv.size() * 4, hipMemcpyHostToDevice, x->stream));
    hnsw_unlink_kernel<<<ceil_div(t_nb.size(), 64), 64, 0, x->stream>>>(x->view(), d_nb, d_lv, (uint32_t)t_nb.size());
    COLTT_HIP(hipGetLastError());
  }
  COLTT_HIP(hipStreamSynchronize(x->stream));
  return COLTT_OK;
}

int coltt_hnsw_export(coltt_handle_t h, uint64_t* n_slots, uint64_t* n_rows, uint64_t* n_edges, uint64_t* ids,
                      int32_t* levels, uint8_t* deleted, int64_t* row_offsets, int32_t* nbr, float* nbr_dist,
                      int32_t* entry_slot) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_export: unknown handle");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  const uint64_t n = x->n;
  const uint32_t W0 = (uint32_t)x->cfg.m_max0, WU = (uint32_t)x->cfg.m_max;
  const bool filling = ids || levels || deleted || row_offsets || nbr || nbr_dist;
  if (filling) {
    // the caller sized its arrays from an earlier call; an Insert in between must not turn into a write past their end
    if (!n_slots || !n_rows || !n_edges) return fail(COLTT_E_INVALID, "hnsw_export: array capacities (n_slots, n_rows, n_edges) are required when arrays are passed");
    uint64_t need_rows = 0;
    for (uint64_t i = 0; i < n; i++) need_rows += (uint64_t)x->h_levels[i] + 1;
    const bool slots_short = (ids || levels || deleted) && *n_slots < n;
    const bool rows_short = row_offsets && *n_rows < need_rows;
    if (slots_short || rows_short) {
      const uint64_t cs = *n_slots, cr = *n_rows;
      *n_slots = n; *n_rows = need_rows;
      return fail(COLTT_E_INVALID, "hnsw_export: the index grew past the caller's arrays (%llu slots / %llu rows offered, %llu / %llu needed)",
                  (unsigned long long)cs, (unsigned long long)cr, (unsigned long long)n, (unsigned long long)need_rows);
    }
  }
  const uint64_t cap_edges = (filling && n_edges) ? *n_edges : 0;
  std::vector<uint32_t> a0((size_t)n * W0), aU((size_t)x->n_upper * WU);
  std::vector<float> d0, dU;
  if (n) COLTT_HIP(hipMemcpy(a0.data(), x->adj0.p, a0.size() * 4, hipMemcpyDeviceToHost));
  if (x->n_upper) COLTT_HIP(hipMemcpy(aU.data(), x->adjU.p, aU.size() * 4, hipMemcpyDeviceToHost));
  if (nbr_dist) {
    d0.resize(a0.size()); dU.resize(aU.size());
    if (n) COLTT_HIP(hipMemcpy(d0.data(), x->adj0_d.p, d0.size() * 4, hipMemcpyDeviceToHost));
    if (x->n_upper) COLTT_HIP(hipMemcpy(dU.data(), x->adjU_d.p, dU.size() * 4, hipMemcpyDeviceToHost));
  }
  uint64_t rows = 0, edges = 0;
  if (row_offsets) row_offsets[0] = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (ids) ids[i] = x->dense ? x->dense_base + i : x->h_ids[i];
    if (levels) levels[i] = x->h_levels[i];
    if (deleted) deleted[i] = (x->h_del[i >> 5] >> (i & 31)) & 1u;
    for (int l = 0; l <= x->h_levels[i]; l++) {
      uint32_t W = l == 0 ? W0 : WU;
      const uint32_t* r = l == 0 ? &a0[(size_t)i * W0] : &aU[((size_t)x->h_upper_off[i] + l - 1) * WU];
      const float* dr = nbr_dist ? (l == 0 ? &d0[(size_t)i * W0] : &dU[((size_t)x->h_upper_off[i] + l - 1) * WU]) : nullptr;
      for (uint32_t j = 0; j < W && r[j] != NBR_NONE; j++) {
        if (edges < cap_edges) {
          if (nbr) nbr[edges] = (int32_t)r[j];
          if (nbr_dist) nbr_dist[edges] = dr[j];
        }
        edges++;
      }
      rows++;
      if (row_offsets) row_offsets[rows] = (int64_t)edges;
    }
  }
  if (n_slots) *n_slots = n;
  if (n_rows) *n_rows = rows;
  if (n_edges) *n_edges = edges;
  if (entry_slot) *entry_slot = x->entry;
  if ((nbr || nbr_dist) && edges > cap_edges)
    return fail(COLTT_E_INVALID, "hnsw_export: the index grew past the caller's edge arrays (%llu offered, %llu needed)",
                (unsigned long long)cap_edges, (unsigned long long)edges);
  return COLTT_OK;
}

int coltt_hnsw_export_raw(coltt_handle_t h, uint64_t* n_slots, uint64_t* n_upper_rows, int32_t* entry_slot,
                          int32_t* entry_level, uint32_t* adj0, uint32_t* upper_off, uint32_t* adjU) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_export_raw: unknown handle");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  if (n_slots) *n_slots = x->n;
  if (n_upper_rows) *n_upper_rows = x->n_upper;
  if (entry_slot) *entry_slot = x->entry;
  if (entry_level) *entry_level = x->entry_level;
  if (adj0 && x->n) COLTT_HIP(hipMemcpy(adj0, x->adj0.p, (size_t)x->n * x->cfg.m_max0 * 4, hipMemcpyDeviceToHost));
  if (upper_off && x->n) COLTT_HIP(hipMemcpy(upper_off, x->upper_off.p, (size_t)x->n * 4, hipMemcpyDeviceToHost));
  if (adjU && x->n_upper) COLTT_HIP(hipMemcpy(adjU, x->adjU.p, (size_t)x->n_upper * x->cfg.m_max * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_hnsw_fetch_rows(coltt_handle_t h, uint64_t first_slot, uint64_t n, void* out_rows) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_fetch_rows: unknown handle");
  if (n == 0) return COLTT_OK;
  ReadLock g(x->rw);
  if (!out_rows || first_slot + n > x->n) return fail(COLTT_E_INVALID, "hnsw_fetch_rows: range outside [0,%llu)", (unsigned long long)x->n);
  COLTT_DEVICE(x->device);
  const size_t rb = (size_t)x->dim * quant_bytes(x->quant);
  if (x->r8) {   // line-transposed rows: natural element order is restored on the device, block by block
    DevBuf d_nat;
    const uint64_t blk = std::max<uint64_t>(1, (256ull << 20) / rb);
    COLTT_TRY(d_nat.reserve(std::min<uint64_t>(blk, n) * rb));
    for (uint64_t b = 0; b < n; b += blk) {
      const uint64_t m = std::min<uint64_t>(blk, n - b);
      if (x->quant == COLTT_Q_NONE) rows_natural_kernel<Q_NONE><<<ceil_div(m * x->dim, 256), 256, 0, nullptr>>>(x->rows.as<uint8_t>(), x->stride, first_slot + b, m, (int)x->dim, d_nat.as<uint8_t>());
      else rows_natural_kernel<Q_F16><<<ceil_div(m * x->dim, 256), 256, 0, nullptr>>>(x->rows.as<uint8_t>(), x->stride, first_slot + b, m, (int)x->dim, d_nat.as<uint8_t>());
      COLTT_HIP(hipGetLastError());
      COLTT_HIP(hipMemcpy(static_cast<uint8_t*>(out_rows) + b * rb, d_nat.p, m * rb, hipMemcpyDeviceToHost));
    }
    return COLTT_OK;
  }
  COLTT_HIP(hipMemcpy2D(out_rows, rb, x->rows.as<uint8_t>() + first_slot * x->stride, x->stride, rb, n, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_hnsw_get(coltt_handle_t h, uint64_t id, void* out_row, int32_t* out_level) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_get: unknown handle");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  uint64_t slot;
  if (x->dense) {
    if (id < x->dense_base || id >= x->dense_base + x->n) return fail(COLTT_E_NOT_FOUND, "Item not found");
    slot = id - x->dense_base;
    if ((x->h_del[slot >> 5] >> (slot & 31)) & 1u) return fail(COLTT_E_NOT_FOUND, "Item not found");
  } else {
    auto it = x->id2slot.find(id);
    if (it == x->id2slot.end()) return fail(COLTT_E_NOT_FOUND, "Item not found");  // ItemNotFoundError (hnsw.go:177,188)
    slot = it->second;
  }
  if (out_row && x->r8) {   // one line-transposed row: copied as stored, natural element order restored on the host
    std::vector<uint8_t> raw(x->stride);
    COLTT_HIP(hipMemcpy(raw.data(), x->rows.as<uint8_t>() + slot * x->stride, x->stride, hipMemcpyDeviceToHost));
    if (x->quant == COLTT_Q_NONE) for (uint32_t e = 0; e < x->dim; e++) static_cast<uint32_t*>(out_row)[e] = reinterpret_cast<const uint32_t*>(raw.data())[r8_index<Q_NONE>((int)e)];
    else for (uint32_t e = 0; e < x->dim; e++) static_cast<uint16_t*>(out_row)[e] = reinterpret_cast<const uint16_t*>(raw.data())[r8_index<Q_F16>((int)e)];
  } else if (out_row) COLTT_HIP(hipMemcpy(out_row, x->rows.as<uint8_t>() + slot * x->stride, (size_t)x->dim * quant_bytes(x->quant), hipMemcpyDeviceToHost));
  if (out_level) *out_level = x->h_levels[slot];
  return COLTT_OK;
}

int coltt_hnsw_reserve(coltt_handle_t h, uint64_t n_slots, uint64_t n_upper_rows) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_reserve: unknown handle");
  if (n_slots >= 0x7fffffffull) return fail(COLTT_E_UNSUPPORTED, "hnsw_reserve: more than 2^31-1 slots");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  // upper rows: a vertex of level L owns L of them; the level draw floor(-ln(u) / ln(M)) averages 1 / (M - 1) per vertex
  if (n_upper_rows == 0) n_upper_rows = n_slots / (uint64_t)std::max(2, x->cfg.m - 1) + n_slots / 64 + 1024;
  return x->reserve(n_slots, n_upper_rows);
}

// ---- product-quantised HNSW (hnsw_pq.hpp; the reference's call shape: playground/hnswpq_verification.go:69-105) -------------------
int coltt_hnsw_pq_attach(coltt_handle_t h, coltt_handle_t pq) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_pq_attach: unknown index handle");
  WriteLock g(x->rw);
  COLTT_DEVICE(x->device);
  x->pq_nbr_stale.store(true);
  if (x->quant == COLTT_Q_F8) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_attach: \"f8\" rows hold eight distinct values — nothing to quantise");
  PqShape sh;
  COLTT_TRY(pq_snapshot(pq, &sh, &x->pq_stage, x->stream));   // staged first: the index is untouched unless everything checks out
  if (sh.dim != x->dim) return fail(COLTT_E_INVALID, "hnsw_pq_attach: the quantiser is for dim %u, the index holds dim %u", sh.dim, x->dim);
  if (sh.metric == COLTT_PQ_DOT) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_attach: dotProductDistance tables are negative — the walk orders distances by their bits (squared L2, or 1 - dot on a cosine index)");
  if (sh.metric == COLTT_PQ_COSINE && x->metric != COLTT_COSINE) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_attach: cosineDistance tables need a cosine index (normalised rows: every table entry is >= 0)");
  if (x->dim % 4) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_attach: dim %u is not a multiple of 4 (the re-rank reads the prepared query rows with 16-byte loads)", x->dim);
  if (sh.metric == COLTT_PQ_COSINE) {
    // 1 - dot(q_j, centroid) must never be negative (the walk orders table distances by their bits): the index hands the walk normalised queries, so
    // ||centroid|| <= 1 is enough — true of a quantiser trained on the index's own (normalised) rows, not of one trained on raw vectors
    COLTT_TRY(x->w_misc.reserve(256));
    float nmax = 0.f;
    COLTT_TRY(pq_centroid_norm_max(x->stream, x->pq_stage.as<float>(), sh, x->w_misc.as<uint32_t>(), &nmax));
    if (!(nmax <= 1.0f + 1e-4f)) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_attach: cosineDistance quantiser with a centroid of squared norm %g > 1 — its tables can be negative (train it on the index's normalised rows)", (double)nmax);
  }
  const uint32_t row = (sh.m + 15u) & ~15u;
  if (row > 128) return fail(COLTT_E_UNSUPPORTED, "hnsw_pq_attach: %u sub-vectors — a query's table (%u KiB) must fit the CU's LDS beside the result set (<= 128)", sh.m, row / 2);
  const size_t bytes = (size_t)sh.m * sh.C * sh.dsub * 4;
  COLTT_TRY(x->pq_cb.reserve(bytes));
  COLTT_HIP(hipMemcpyAsync(x->pq_cb.p, x->pq_stage.p, bytes, hipMemcpyDeviceToDevice, x->stream));
  COLTT_HIP(hipStreamSynchronize(x->stream));
  x->pq_shape = sh; x->pq_row = row; x->pq_done = 0; x->pq_on = true;
  if (x->pq_codes.p) COLTT_HIP(hipMemsetAsync(x->pq_codes.p, 0, x->pq_codes.cap, x->stream));
  int rc = sync_pq(x.get());
  if (rc == COLTT_OK && x->vis_stride != 0) rc = ensure_visg(x.get());   // (exclusive lock: no traversal holds a region) 2048 -> 3072 regions for the table walk
  if (rc != COLTT_OK) x->pq_on = false;
  return rc;
}

int coltt_hnsw_pq_info(coltt_handle_t h, uint32_t* out_m, uint32_t* out_c, int32_t* out_metric, uint64_t* out_coded) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_pq_info: unknown handle");
  ReadLock g(x->rw);
  if (out_m) *out_m = x->pq_on ? x->pq_shape.m : 0;
  if (out_c) *out_c = x->pq_on ? x->pq_shape.C : 0;
  if (out_metric) *out_metric = x->pq_on ? x->pq_shape.metric : -1;
  if (out_coded) *out_coded = x->pq_on ? x->pq_done : 0;
  return COLTT_OK;
}

int coltt_hnsw_pq_fetch_codes(coltt_handle_t h, uint64_t first_slot, uint64_t n, uint8_t* out_codes) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_pq_fetch_codes: unknown handle");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  if (!x->pq_on) return fail(COLTT_E_INVALID, "hnsw_pq_fetch_codes: no quantiser attached");
  if (first_slot + n > x->pq_done) return fail(COLTT_E_INVALID, "hnsw_pq_fetch_codes: slots [%llu,%llu) outside the %llu coded ones", (unsigned long long)first_slot, (unsigned long long)(first_slot + n), (unsigned long long)x->pq_done);
  if (n == 0) return COLTT_OK;
  if (!out_codes) return fail(COLTT_E_INVALID, "hnsw_pq_fetch_codes: NULL out");
  COLTT_HIP(hipMemcpy2D(out_codes, x->pq_shape.m, x->pq_codes.as<uint8_t>() + first_slot * x->pq_row, x->pq_row, x->pq_shape.m, n, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_hnsw_pq_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, uint32_t ef_override, uint32_t rerank,
                         uint64_t* out_ids, float* out_scores, uint32_t* out_counts, coltt_hnsw_stats* stats, uint64_t* out_n_exact) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_pq_search: unknown handle");
  if (nq && (!queries || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "hnsw_pq_search: NULL buffer");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  CtxLease<HCtx> ctx(x->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return pq_search_common(x.get(), ctx.c, queries, false, nq, k, ef_override, rerank, out_ids, out_scores, out_counts, stats, out_n_exact);
}
int coltt_hnsw_pq_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, uint32_t ef_override, uint32_t rerank,
                                uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_counts, coltt_hnsw_stats* stats, uint64_t* out_n_exact) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_pq_search_device: unknown handle");
  if (nq && (!d_queries || !d_out_ids || !d_out_scores || !d_out_counts)) return fail(COLTT_E_INVALID, "hnsw_pq_search_device: NULL buffer");
  ReadLock g(x->rw);
  COLTT_DEVICE(x->device);
  CtxLease<HCtx> ctx(x->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return pq_search_common(x.get(), ctx.c, d_queries, true, nq, k, ef_override, rerank, d_out_ids, d_out_scores, d_out_counts, stats, out_n_exact);
}

int coltt_hnsw_rows8_searches(coltt_handle_t h, uint64_t* out_launches, int32_t* out_has_copy) {
  auto x = lookup<Hnsw>(h);
  if (!x) return fail(COLTT_E_NOT_FOUND, "hnsw_rows8_searches: unknown handle");
  ReadLock g(x->rw);
  if (out_launches) *out_launches = x->ev8_launches.load();
  if (out_has_copy) *out_has_copy = x->r8 ? 1 : 0;
  return COLTT_OK;
}

int coltt_last_kernel_ms(coltt_handle_t h, float* out_ms) {
  if (!out_ms) return fail(COLTT_E_INVALID, "last_kernel_ms: NULL out");
  if (auto x = lookup<Hnsw>(h)) { *out_ms = x->last_ms.load(); return COLTT_OK; }
  return coltt_last_kernel_ms_flat(h, out_ms);
}

}  // extern "C"
