// hnsw_kernels.hpp — the search kernels of hnsw.hip (Hnsw.Search, core/vectorindex/hnsw.go:243-278): the one-wave walk, the large-ef walk,
// the single-query latency kernel, the walk over product-quantiser codes with its exact re-rank.  A header of their own so that a scratch
// translation unit can instantiate a handful of them for ISA inspection (tools/isa/walks.hip, tools/isa_loops.py) in seconds instead of the
// whole of hnsw.hip.  Included by hnsw.hip inside its anonymous namespace users; nothing here is host code.
#pragma once
#include "common.hpp"
#include "exact.hpp"
#include "hnsw_dev.hpp"
#include "hnsw_walk2.hpp"
#include "hnsw_lat.hpp"
#include "hnsw_pq.hpp"

namespace coltt {
namespace kern {
using namespace coltt::dev;

// Hnsw.Search (hnsw.go:243-278) for a batch: one wave per query, queries pulled from a global counter.
template <int METRIC, int QUANT, bool VISG, bool R8 = false>   // R8: the index's ONE row array is line-transposed (rows8.hpp)
__global__ __launch_bounds__(64) void hnsw_search_kernel(GraphView g, int32_t entry, int32_t entry_level,
                                                        const float* __restrict__ q_eff, const float* __restrict__ qnorms,
                                                        uint32_t nq, uint32_t k, uint32_t ef, uint32_t ef_pad, uint32_t hcap,
                                                        uint32_t* __restrict__ counter, uint64_t* __restrict__ out_ids,
                                                        float* __restrict__ out_scores, uint32_t* __restrict__ out_counts,
                                                        unsigned long long* __restrict__ stats, uint8_t* __restrict__ visg,
                                                        size_t vis_stride, uint32_t* __restrict__ vis_epoch) {
  constexpr int PROF = VISG ? PROF_SEARCH_HBM : PROF_SEARCH_LDS;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  WaveCtx w;
  size_t off = ((size_t)g.dim * 4 + 15) & ~(size_t)15;
  w.qs = reinterpret_cast<float*>(smem);
  w.res0 = reinterpret_cast<unsigned long long*>(smem + off);
  w.vis = reinterpret_cast<uint32_t*>(w.res0 + (size_t)ef_pad);
  w.ef_pad = ef_pad; w.hcap = hcap; w.hcap_mask = hcap - 1;
  w.visg = nullptr; w.vis_bytes = 0; w.epoch = 0;
  if constexpr (VISG) { w.visg = visg + (size_t)blockIdx.x * vis_stride; w.vis_bytes = vis_stride; w.epoch = vis_epoch[blockIdx.x]; }
  for (;;) {
    // dynamic work fetch.  Branch-free on purpose: with `if (lane == 0) t = atomicAdd(..)` hipcc threads the
    // loop-invariant divergent branch through the back edge, lanes 1..63 re-enter the loop without lane 0 and the
    // cross-lane broadcast below reads a stale value forever.
    const uint32_t qt = atomicAdd(counter, lane == 0 ? 1u : 0u);
    const uint32_t qi = (uint32_t)__shfl((int)qt, 0, 64);
    if (qi >= nq) break;
    w.n_dist = w.n_exp = w.n_hops = w.n_resets = 0; w.err = 0;
#ifdef COLTT_PHASE_TIMING
    for (int i_ = 0; i_ < 8; i_++) w.pt[i_] = 0;
    w.t_last = __builtin_amdgcn_s_memtime();
#endif
    wave_sync();
    for (int e = lane; e < g.dim; e += 64) w.qs[e] = q_eff[(size_t)qi * g.dim + e];
    w.qnorm = qnorms[qi];
    wave_sync();
    // minDistance := Distance(query, entrypoint.vector) (hnsw.go:253)
    uint32_t cur = (uint32_t)entry;
    float curd = eval_pair<METRIC, QUANT, PROF, R8>(g, w, cur, lane & 1);
    curd = __shfl(curd, 0, 64);
    w.n_dist += 1;
    for (int l = entry_level; l > 0; l--) greedy_level<METRIC, QUANT, PROF, R8>(g, w, cur, curd, l, lane);  // :254-256
    COLTT_PT(w, 5)  // query load + entry distance + upper levels
    // searchLevel re-evaluates the entrypoint distance (hnsw.go:346)
    w.n_dist += 1;
    uint32_t len; int buf;
    search_level<METRIC, QUANT, VISG, PROF, R8>(g, w, cur, curd, ef, 0, lane, len, buf);  // :258-259
    // selectNeighbors + pop into result[n-1..0] (:261-277) == the k smallest, ascending
    uint32_t n = len < k ? len : k;
    const unsigned long long* res = w.res0 + (size_t)buf * ef_pad;
    for (uint32_t i = lane; i < n; i += 64) {
      unsigned long long e = res[i];
      uint32_t slot = (uint32_t)e >> 1;
      out_ids[(size_t)qi * k + i] = g.ids ? g.ids[slot] : (uint64_t)slot;
      out_scores[(size_t)qi * k + i] = __uint_as_float((uint32_t)(e >> 32));
    }
    if (lane == 0) {
      out_counts[qi] = n;
      atomicAdd(&stats[0], (unsigned long long)w.n_dist);
      atomicAdd(&stats[1], (unsigned long long)w.n_exp);
      atomicAdd(&stats[2], (unsigned long long)w.n_hops);
      atomicAdd(&stats[3], (unsigned long long)w.n_resets);
      if (w.err) atomicOr(&stats[4], (unsigned long long)w.err);
#ifdef COLTT_PHASE_TIMING
      COLTT_PT(w, 6)  // result write-out
      for (int i_ = 0; i_ < 8; i_++) atomicAdd(&stats[8 + i_], w.pt[i_]);
#endif
    }
  }
  if constexpr (VISG) { if (lane == 0) vis_epoch[blockIdx.x] = w.epoch; }
}


// Hnsw.Search for large ef (HBM visited map): the level-0 walk of hnsw_walk2.hpp — delta result set, LDS Bloom filter in front
// of the byte map, neighbour norms riding with the adjacency rows (OPT bits) — at two register/occupancy profiles.
// EV8: the level-0 distances come from the eight-lanes-per-row core over GraphView::rows8 (rows8.hpp); the upper levels and the
// entrypoint (a few dozen evaluations) stay on the pair-owned rows.
template <int METRIC, int QUANT, int PROFILE, int OPT, int VISMODE = VIS_HBM, bool APREF = false, bool EV8 = false, bool R8 = false, bool NT = false>   // NT (EV8 only): non-temporal row loads, for collections far larger than the caches (exact.hpp: row_ld); APREF: adjacency prefetch for f32 rows too (small batches); R8 (without EV8): the pair-owned core over line-transposed rows
#ifndef COLTT_OP_WAVES_PER_EU   // experiment knob: minimum waves per SIMD the register allocator must leave room for in the HBM-visited eight-lane 2-byte walk (the operating point)
#define COLTT_OP_WAVES_PER_EU 1
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((EV8 && QUANT != Q_NONE && VISMODE == VIS_HBM) ? COLTT_OP_WAVES_PER_EU : 1)))
void hnsw_search2_kernel(GraphView g, int32_t entry, int32_t entry_level,
                                                         const float* __restrict__ q_eff, const float* __restrict__ qnorms,
                                                         uint32_t nq, uint32_t k, uint32_t ef, uint32_t ef_pad, uint32_t bloom_words,
                                                         uint32_t* __restrict__ counter, uint64_t* __restrict__ out_ids,
                                                         float* __restrict__ out_scores, uint32_t* __restrict__ out_counts,
                                                         unsigned long long* __restrict__ stats, uint8_t* __restrict__ visg,
                                                         size_t vis_stride, uint32_t* __restrict__ vis_epoch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  WaveCtx w;
  size_t off = ((size_t)g.dim * 4 + 15) & ~(size_t)15;
  w.qs = reinterpret_cast<float*>(smem);
  w.qp = nullptr; w.scr = nullptr;
  if constexpr (EV8) {   // [query in rows8 order | scratch] then the result set (search_geom adds the same bytes); no natural-order copy
    w.qp = w.qs; w.qs = nullptr;
    w.scr = reinterpret_cast<uint32_t*>(smem + off);
    off += 96 * 4;
  }
  w.res0 = reinterpret_cast<unsigned long long*>(smem + off);
  w.ef_pad = ef_pad;
  if constexpr (VISMODE == VIS_LDS) {   // small ef: the LDS hash (bloom_words carries its capacity); a table that fills up is err 8
    w.vis = reinterpret_cast<uint32_t*>(w.res0 + (size_t)ef_pad);
    w.hcap = bloom_words; w.hcap_mask = bloom_words - 1;
    w.bloom = nullptr; w.bloom_words = 0; w.bloom_shift = 0;
    w.visg = nullptr; w.vis_bytes = 0; w.epoch = 0;
  } else {
    w.vis = nullptr;
    w.bloom = reinterpret_cast<uint32_t*>(w.res0 + (size_t)ef_pad);
    w.bloom_words = bloom_words; w.bloom_shift = 32u - (uint32_t)__builtin_ctz(bloom_words | 0x80000000u);
    w.hcap = 0; w.hcap_mask = 0;
    w.visg = visg + (size_t)blockIdx.x * vis_stride; w.vis_bytes = vis_stride; w.epoch = vis_epoch[blockIdx.x];
  }
  for (;;) {
    const uint32_t qt = atomicAdd(counter, lane == 0 ? 1u : 0u);  // branch-free work fetch, see hnsw_search_kernel
    const uint32_t qi = (uint32_t)__shfl((int)qt, 0, 64);
    if (qi >= nq) break;
    w.n_dist = w.n_exp = w.n_hops = w.n_resets = 0; w.err = 0;
#ifdef COLTT_PHASE_TIMING
    for (int i_ = 0; i_ < 8; i_++) w.pt[i_] = 0;
    w.t_last = __builtin_amdgcn_s_memtime();
#endif
    wave_sync();
    for (int e = lane; e < g.dim; e += 64) {
      const float v = q_eff[(size_t)qi * g.dim + e];
      if constexpr (EV8) w.qp[rows8_qindex<QUANT>(e)] = v; else w.qs[e] = v;
    }
    w.qnorm = qnorms[qi];
    wave_sync();
    uint32_t cur = (uint32_t)entry;
    float curd;
    constexpr bool H16 = EV8 && QUANT != Q_NONE && VISMODE == VIS_HBM;   // Group8Eval: rows x burst depth of the HBM-visited 2-byte kernels
    if constexpr (EV8) curd = Group8Eval<METRIC, QUANT, false, H16, NT>().one(g, w, cur, lane);
    else curd = eval_pair<METRIC, QUANT, PROFILE, R8>(g, w, cur, lane & 1);  // hnsw.go:253
    curd = __shfl(curd, 0, 64);
    w.n_dist += 1;
    for (int l = entry_level; l > 0; l--) {  // :254-256
      if constexpr (EV8) greedy_level8<METRIC, QUANT, H16, NT>(g, w, cur, curd, l, lane);
      else greedy_level<METRIC, QUANT, PROFILE, R8>(g, w, cur, curd, l, lane);
    }
    COLTT_PT(w, 5)
    w.n_dist += 1;  // searchLevel re-evaluates the entrypoint distance (hnsw.go:346)
    uint32_t len;
    if constexpr (EV8) search_level2<METRIC, QUANT, PROFILE, OPT, VISMODE, APREF>(g, w, cur, curd, ef, lane, len, Group8Eval<METRIC, QUANT, (OPT & W2_ADJN) != 0 && METRIC == M_COS, H16, NT>());
    else search_level2<METRIC, QUANT, PROFILE, OPT, VISMODE, APREF>(g, w, cur, curd, ef, lane, len, PairEval<METRIC, QUANT, PROFILE, (OPT & W2_ADJN) != 0 && METRIC == M_COS, R8>());  // :258-259
    const uint32_t n = len < k ? len : k;  // selectNeighbors + pop (:261-277) == the k smallest, ascending
    for (uint32_t i = lane; i < n; i += 64) {
      const unsigned long long e = w.res0[i];
      const uint32_t slot = (uint32_t)e >> 1;
      out_ids[(size_t)qi * k + i] = g.ids ? g.ids[slot] : (uint64_t)slot;
      out_scores[(size_t)qi * k + i] = __uint_as_float((uint32_t)(e >> 32));
    }
    if (lane == 0) {
      out_counts[qi] = n;
      atomicAdd(&stats[0], (unsigned long long)w.n_dist);
      atomicAdd(&stats[1], (unsigned long long)w.n_exp);
      atomicAdd(&stats[2], (unsigned long long)w.n_hops);
      if (w.err) atomicOr(&stats[4], (unsigned long long)w.err);
#ifdef COLTT_PHASE_TIMING
      COLTT_PT(w, 6)
      for (int i_ = 0; i_ < 8; i_++) atomicAdd(&stats[8 + i_], w.pt[i_]);
#endif
    }
  }
  if constexpr (VISMODE == VIS_HBM) { if (lane == 0) vis_epoch[blockIdx.x] = w.epoch; }
}


// Hnsw.Search with a 256-thread workgroup per query (hnsw_lat.hpp): the latency path for small batches — the reference serves one
// query per RPC (core/core.go:633-667).  One workgroup per CU, queries pulled from a global counter.  Same answers, score bits and
// counters as the one-wave kernel (the parity tests run both).
// TP: how the rows are read (hnsw_lat.hpp: lat_eval_chunk) — > 0 line-transposed rows of TP 128-byte lines, -1 line-transposed of any length, 0 natural order
// (staged through LDS).  SEQ: hnsw_walk2.hpp's level-0 walk on wave 0 with the chunks evaluated by all four waves (rows of more than one chunk, mMax0 > 32, and
// the COLTT_LAT_SEQ=1 A/B partner) instead of the walk that is software-pipelined over expansions.  One walk and one evaluation per instance (round 6).
template <int METRIC, int QUANT, int TP = 0, bool SEQ = false>
__global__ __launch_bounds__(256) void hnsw_search_lat_kernel(GraphView g, int32_t entry, int32_t entry_level,
                                                             const float* __restrict__ q_eff, const float* __restrict__ qnorms,
                                                             uint32_t nq, uint32_t k, uint32_t ef, uint32_t ef_pad, uint32_t hcap,
                                                             uint32_t* __restrict__ counter, uint64_t* __restrict__ out_ids,
                                                             float* __restrict__ out_scores, uint32_t* __restrict__ out_counts,
                                                             unsigned long long* __restrict__ stats, unsigned long long* __restrict__ mbox, int helpers) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // helped launches (mbox != null; batches of <= LAT_MASTERS queries): blocks 0 .. LAT_MASTERS - 1 walk, block m + 8 (h + 1) is walking block m's helper h
  unsigned long long* hint_box = nullptr;
  if (mbox) {
    const uint32_t m = blockIdx.x & (uint32_t)(LAT_MASTERS - 1);
    unsigned long long* const box = mbox + (size_t)m * (LAT_HELPERS_MAX + 1);
    if (blockIdx.x >= (uint32_t)LAT_MASTERS) {
      lat_helper_loop(g, box, (int)(blockIdx.x / LAT_MASTERS) - 1, reinterpret_cast<LatShared*>(smem));
      return;
    }
    hint_box = box;
  }
  WaveCtx w;
  size_t off = (lat_q_floats(g.dim) * 4 + 15) & ~(size_t)15;   // the query, residue-major (hnsw_lat.hpp: lat_n8p)
  w.qs = reinterpret_cast<float*>(smem);
  w.res0 = reinterpret_cast<unsigned long long*>(smem + off);
  LatShared* xs = reinterpret_cast<LatShared*>(w.res0 + (size_t)ef_pad);
  uint8_t* stage = reinterpret_cast<uint8_t*>(xs + 1);                                   // TP == 0: [32][stride + pad] rows of the chunk being evaluated
  w.vis = reinterpret_cast<uint32_t*>(stage + (TP == LAT_TP_STAGED ? (size_t)LAT_ROWS * (g.stride + LAT_PAD) : (size_t)0));
  w.ef_pad = ef_pad; w.hcap = hcap; w.hcap_mask = hcap - 1;
  w.visg = nullptr; w.vis_bytes = 0; w.epoch = 0; w.bloom = nullptr; w.bloom_words = 0; w.bloom_shift = 0;
  for (;;) {
    __syncthreads();   // everybody is done with the previous query's LDS state (and with ctl[1])
    if (threadIdx.x == 0) xs->ctl[1] = atomicAdd(counter, 1u);
    __syncthreads();
    const uint32_t qi = xs->ctl[1];
    if (qi >= nq) break;
    w.n_dist = w.n_exp = w.n_hops = w.n_resets = 0; w.err = 0;
#ifdef COLTT_PHASE_TIMING
    for (int i_ = 0; i_ < 8; i_++) w.pt[i_] = 0;
    w.t_last = __builtin_amdgcn_s_memtime();
#endif
    {
      const int n8 = g.dim >> 3, n8p = lat_n8p(g.dim);
      for (int e = threadIdx.x; e < g.dim; e += 256) {
        const float v = q_eff[(size_t)qi * g.dim + e];
        if (e < n8 * 8) w.qs[(size_t)(e & 7) * n8p + (e >> 3)] = v; else w.qs[(size_t)8 * n8p + (e - n8 * 8)] = v;
      }
    }
    w.qnorm = qnorms[qi];
    // minDistance := Distance(query, entrypoint.vector) (hnsw.go:253): a chunk with one live row
    if (threadIdx.x < LAT_ROWS) { xs->nb[threadIdx.x] = threadIdx.x == 0 ? (uint32_t)entry : NBR_NONE; xs->fresh[threadIdx.x] = threadIdx.x == 0 ? 1u : 0u; }
    lat_chunk<METRIC, QUANT, TP>(g, w, xs, stage, wave, lane);
    uint32_t cur = (uint32_t)entry;
    float curd = xs->d[0];
    w.n_dist += 1;
    __syncthreads();   // xs->d[0] has been read by every wave before the next chunk overwrites it
    for (int l = entry_level; l > 0; l--) greedy_level_lat<METRIC, QUANT, TP>(g, w, xs, stage, cur, curd, l, wave, lane);  // :254-256
#ifdef COLTT_PHASE_TIMING
    if (wave == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); w.pt[6] += t_ - w.t_last; w.t_last = t_; }   // query load + entry + upper levels
#endif
    w.n_dist += 1;  // searchLevel re-evaluates the entrypoint distance (hnsw.go:346)
    uint32_t len;
    if constexpr (!SEQ) {   // rows of one chunk (mMax0 <= 32: the host checks): the walk that is software-pipelined over expansions (hnsw_lat.hpp)
      search_level_lat3<METRIC, QUANT, TP>(g, w, xs, stage, cur, curd, ef, wave, lane, len, hint_box, helpers);  // :258-259
    } else if (wave == 0) {   // hnsw_walk2.hpp's level-0 walk on wave 0, the chunks evaluated by all four waves (hnsw_lat.hpp: LatEval)
      if (lane == 0) xs->ctl[0] = 1u;
      LatEval<METRIC, QUANT, TP> ev{xs, stage};
      search_level2<METRIC, QUANT, PROF_SEARCH_LDS, W2_DELTA, VIS_LDS, true>(g, w, cur, curd, ef, lane, len, ev);  // :258-259
      wave_sync();
      if (lane == 0) xs->ctl[0] = 0u;
      lds_barrier();   // releases the companions
    } else {
      len = 0;
      lat_companion<METRIC, QUANT, TP>(g, w, xs, stage, wave, lane);
    }
    if (wave == 0) {
      const uint32_t n = len < k ? len : k;
      for (uint32_t i = lane; i < n; i += 64) {
        const unsigned long long e = w.res0[i];
        const uint32_t slot = (uint32_t)e >> 1;
        out_ids[(size_t)qi * k + i] = g.ids ? g.ids[slot] : (uint64_t)slot;
        out_scores[(size_t)qi * k + i] = __uint_as_float((uint32_t)(e >> 32));
      }
      if (lane == 0) {
        out_counts[qi] = n;
        atomicAdd(&stats[0], (unsigned long long)w.n_dist);
        atomicAdd(&stats[1], (unsigned long long)w.n_exp);
        atomicAdd(&stats[2], (unsigned long long)w.n_hops);
        if (w.err) atomicOr(&stats[4], (unsigned long long)w.err);
#ifdef COLTT_PHASE_TIMING
        for (int i_ = 0; i_ < 8; i_++) atomicAdd(&stats[8 + i_], w.pt[i_]);
#endif
      }
    }
  }
  if (hint_box && threadIdx.x == 0) __hip_atomic_store(hint_box + LAT_HELPERS_MAX, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // releases this block's helpers
}



// Hnsw.Search over product-quantiser codes + exact re-rank (hnsw_pq.hpp): one wave per query, queries pulled from a global counter.
// OPT / VISMODE as hnsw_search2_kernel (0 + VIS_LDS: the LDS hash; 2 + VIS_HBM: byte map, delta result set; no Bloom filter — see pq_geom).
// The walk touches no stored row: its survivors (slots, nearest first by table distance) go to HBM and the exact re-rank is two small kernels of its
// own (below) — inside the walk kernel it was 19 % of the time (one 128-byte line per row in flight, the burst depth the walk's register budget left:
// profiles/r05p_phase_breakdown.txt) and tied 24 instances of this kernel to the row format.
// LS: the table's row length (log2) when it is one of the common ones (16, 32, 256 centroids), 0 = any; NP: 16-byte pieces per code row, 0 = any;
// NBR: level-0 code rows come from the neighbourhood blocks, requested with the adjacency row (hnsw_pq.hpp: AdcEval<LS, NP, NBR>).
template <int OPT, int VISMODE, int LS, int NP = 0, bool NBR = false>
// amdgpu_waves_per_eu(3): <= 168 VGPRs, three waves per SIMD — the walk is latency-bound, resident traversals are its throughput
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void hnsw_pq_search_kernel(GraphView g, int32_t entry, int32_t entry_level, const unsigned short* __restrict__ lut_g,
                                                            const uint8_t* __restrict__ codes, const uint8_t* __restrict__ nbrc, uint32_t row_bytes, uint32_t lut_shift, uint32_t nq, uint32_t k,
                                                            uint32_t ef, uint32_t ef_pad, uint32_t rerank, uint32_t vis_words,
                                                            uint32_t* __restrict__ counter, uint32_t* __restrict__ surv, uint32_t* __restrict__ surv_cnt,
                                                            unsigned long long* __restrict__ stats, uint8_t* __restrict__ visg,
                                                            size_t vis_stride, uint32_t* __restrict__ vis_epoch) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x;
  WaveCtx w;
  // LDS: [table | result set | visited hash or Bloom filter] — the table first: its lookups address it by immediate offsets (AdcEval<LS>).
  // No copy of the query: the walk only needs its table.
  unsigned short* const lut = reinterpret_cast<unsigned short*>(smem);
  if constexpr (LS != 0) {   // AdcEval<LS> addresses the table by absolute LDS offsets: this kernel has no static LDS, so its dynamic LDS starts at 0
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem != 0u) { if (lane == 0) atomicOr(&stats[4], 128ull); return; }
  }
  size_t off = ((size_t)pq_walk_table_rows(row_bytes >> 4) << lut_shift) * 2;   // the pair-interleaved table (hnsw_pq.hpp); a multiple of 512
  w.qs = nullptr; w.qp = nullptr; w.scr = nullptr;
  w.res0 = reinterpret_cast<unsigned long long*>(smem + off); off += (size_t)ef_pad * 8;
  w.ef_pad = ef_pad;
  if constexpr (VISMODE == VIS_LDS) {
    w.vis = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)vis_words * 4;
    w.hcap = vis_words; w.hcap_mask = vis_words - 1;
    w.bloom = nullptr; w.bloom_words = 0; w.bloom_shift = 0;
    w.visg = nullptr; w.vis_bytes = 0; w.epoch = 0;
  } else {
    w.vis = nullptr; w.hcap = 0; w.hcap_mask = 0;
    w.bloom = reinterpret_cast<uint32_t*>(smem + off); off += (size_t)vis_words * 4;
    w.bloom_words = vis_words; w.bloom_shift = 32u - (uint32_t)__builtin_ctz(vis_words | 0x80000000u);
    w.visg = visg + (size_t)blockIdx.x * vis_stride; w.vis_bytes = vis_stride; w.epoch = vis_epoch[blockIdx.x];
  }
  AdcEval<LS, NP, NBR> ev; ev.codes = codes; ev.row_bytes = row_bytes; ev.lut = lut; ev.lut_shift = lut_shift; ev.nbrc = nbrc; ev.nbr_stride = g.mMax0 * row_bytes;
  ev.hsel = (uint32_t)lane & 1u;
  for (;;) {
    const uint32_t qt = atomicAdd(counter, lane == 0 ? 1u : 0u);  // branch-free work fetch, see hnsw_search_kernel
    const uint32_t qi = (uint32_t)__shfl((int)qt, 0, 64);
    if (qi >= nq) break;
    w.n_dist = w.n_exp = w.n_hops = w.n_resets = 0; w.err = 0;
#ifdef COLTT_PHASE_TIMING
    for (int i_ = 0; i_ < 8; i_++) w.pt[i_] = 0;
    w.t_last = __builtin_amdgcn_s_memtime();
#endif
    wave_sync();
    {  // the query's table as pq_lut16_kernel wrote it — row_bytes rows of (1 << lut_shift) binary16 entries — copied 16 bytes per lane and step into the
      // PAIR-INTERLEAVED layout the lane pairs read (hnsw_pq.hpp): table row j -> LDS row 2 (j mod JS) + (j div JS), JS = 16 * ceil(pieces / 2)
      const u32x4v* src = reinterpret_cast<const u32x4v*>(lut_g + ((size_t)qi * row_bytes << lut_shift));
      u32x4v* dst = reinterpret_cast<u32x4v*>(lut);
      const uint32_t psh = lut_shift - 3;                    // log2 of the 16-byte pieces per table row (lut_shift >= 4)
      const uint32_t total = row_bytes << psh;               // 16-byte pieces (row_bytes is a multiple of 16)
      const uint32_t js = 16u * (((row_bytes >> 4) + 1u) >> 1);
      for (uint32_t i = (uint32_t)lane; i < total; i += 64) {
        const uint32_t j = i >> psh, within = i & ((1u << psh) - 1u);
        const uint32_t r = j < js ? 2u * j : 2u * (j - js) + 1u;
        dst[(r << psh) + within] = src[i];
      }
    }
    w.qnorm = 0.f;
    wave_sync();
    uint32_t cur = (uint32_t)entry;
    float curd = ev.adc(cur);   // minDistance := d(query, entrypoint) (hnsw.go:253), the same value in every lane
    w.n_dist += 1;
    for (int l = entry_level; l > 0; l--) greedy_level_adc(g, w, ev, cur, curd, l, lane);  // :254-256
    w.n_dist += 1;  // searchLevel re-evaluates the entrypoint distance (hnsw.go:346)
    COLTT_PT(w, 5)  // table load + entry + upper levels
    uint32_t len;
    search_level2<M_L2, Q_F16, PROF_SEARCH_HBM, OPT, VISMODE, true>(g, w, cur, curd, ef, lane, len, ev);  // :258-259 (M_L2: no norms ride along; Q_F16: the adjacency prefetch)
    uint32_t r = rerank == 0 ? len : (rerank > k ? rerank : k);
    r = r < len ? r : len;
    for (uint32_t i = (uint32_t)lane; i < r; i += 64) surv[(size_t)qi * ef_pad + i] = (uint32_t)w.res0[i] >> 1;   // the r nearest by table distance, in that order
    COLTT_PT(w, 6)  // final delta flush + survivors' write-out
    if (lane == 0) {
#ifdef COLTT_PHASE_TIMING
      for (int i_ = 0; i_ < 8; i_++) atomicAdd(&stats[8 + i_], w.pt[i_]);
#endif
      surv_cnt[qi] = r;
      atomicAdd(&stats[0], (unsigned long long)w.n_dist);
      atomicAdd(&stats[1], (unsigned long long)w.n_exp);
      atomicAdd(&stats[2], (unsigned long long)w.n_hops);
      atomicAdd(&stats[3], (unsigned long long)r);
      if (w.err) atomicOr(&stats[4], (unsigned long long)w.err);
    }
  }
  if constexpr (VISMODE == VIS_HBM) { if (lane == 0) vis_epoch[blockIdx.x] = w.epoch; }
}

// Exact re-rank, step 1: the index's distance (reference summation order, exact.hpp) of every survivor.  One wave per (query, 32 survivors): lane pair p
// owns survivor 32 * chunk + p; key = exact score bits << 32 | slot << 1, the walk's own key layout.  The query is read from the prepared batch.
template <int METRIC, int QUANT, bool R8>
__global__ __launch_bounds__(64) void hnsw_pq_rerank_kernel(GraphView g, const float* __restrict__ q_eff, const float* __restrict__ qnorms, const uint32_t* __restrict__ surv,
                                                            const uint32_t* __restrict__ surv_cnt, uint32_t ef_pad, unsigned long long* __restrict__ keys) {
  const uint32_t qi = blockIdx.y, r = surv_cnt[qi];
  if (blockIdx.x * 32u >= r) return;   // wave-uniform
  const int lane = threadIdx.x, half = lane & 1, p = lane >> 1;
  WaveCtx w;
  w.qs = const_cast<float*>(q_eff + (size_t)qi * g.dim); w.qnorm = qnorms[qi];
  const uint32_t i = blockIdx.x * 32u + (uint32_t)p;
  const bool valid = i < r;
  const uint32_t slot = surv[(size_t)qi * ef_pad + (valid ? i : blockIdx.x * 32u)];   // an idle pair re-evaluates a live survivor (DPP partners stay active)
  const float d = eval_pair<METRIC, QUANT, PROF_SEARCH_HBM, R8>(g, w, slot, half);
  if (valid && half == 0) keys[(size_t)qi * ef_pad + i] = ((unsigned long long)__float_as_uint(d) << 32) | ((unsigned long long)slot << 1);
}
// step 2: the k smallest keys of a query — (exact score bits, slot) order — by k rounds of a wave minimum over the keys staged in LDS
__global__ __launch_bounds__(64) void hnsw_pq_select_kernel(const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ surv_cnt, uint32_t ef_pad, uint32_t k,
                                                            const uint64_t* __restrict__ ids, uint64_t* __restrict__ out_ids, float* __restrict__ out_scores,
                                                            uint32_t* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  unsigned long long* const res = reinterpret_cast<unsigned long long*>(smem);
  const uint32_t qi = blockIdx.x, r = surv_cnt[qi];
  const int lane_in = threadIdx.x;
  for (uint32_t i = (uint32_t)lane_in; i < r; i += 64) res[i] = keys[(size_t)qi * ef_pad + i];
  wave_sync();
  const uint32_t n = r < k ? r : k;
  for (uint32_t t = 0; t < n; t++) {
    const int lane = opaque_lane(lane_in);
    unsigned long long best = ~0ull; uint32_t bi = 0;
    for (uint32_t i = (uint32_t)lane; i < r; i += 64) { const unsigned long long e = res[i]; if (e < best) { best = e; bi = i; } }
    const unsigned long long km = wave_min_u64(best);
    if (best == km && km != ~0ull) {   // keys are distinct (a slot appears once): exactly one lane
      const uint32_t slot = (uint32_t)km >> 1;
      out_ids[(size_t)qi * k + t] = ids ? ids[slot] : (uint64_t)slot;
      out_scores[(size_t)qi * k + t] = __uint_as_float((uint32_t)(km >> 32));
      res[bi] = ~0ull;
    }
    wave_sync();
  }
  if (lane_in == 0) out_counts[qi] = n;
}

}  // namespace kern
}  // namespace coltt
