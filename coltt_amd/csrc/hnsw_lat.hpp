// hnsw_lat.hpp — Hnsw.Search for ONE query on a 256-thread workgroup: the LATENCY path of the reference's RPC shape
// (core/core.go:633-667 and edge/edge.go:610-690 serve one query per call).
//
// With a single query in flight the chip is idle and an expansion is a chain of dependent round trips.  In the one-wave kernel the
// longest link is the row fetch: a lane pair streams its 3 KB row through 96 sequential 16-byte loads in bursts (the residue chains
// must be summed in index order), several HBM round trips per expansion.  Here the FETCH is decoupled from the arithmetic order and
// every wave computes:
//   * wave w owns neighbours 8w .. 8w+7 of the expansion; EIGHT lanes cover one row, lane j of the group loading the 16-byte
//     pieces j, j+8, j+16, ... — one whole 128-byte line per row per instruction, the whole row in flight at once: ONE round trip
//     for all 32 rows (96 KB at 768 x f32);
//   * the pieces are copied to LDS as they are and read back element-wise: lane j of the group owns residue j of the reference's
//     8-lane AVX accumulator (pkg/distance/simd/cpp/avx.cpp:15-32,51-75) and adds q[8g+j] * r[8g+j] for g = 0, 1, 2, ... strictly in
//     order (separate multiply and add); the eight partial sums are combined ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)) by a DPP butterfly
//     (IEEE addition commutes, so both sides of every exchange hold the same bits), then the scalar tail and the epilogue of
//     exact.hpp.  Same values, same order, same bits as pair_distance — the parity tests run both kernels;
//   * wave 0 keeps what is inherently serial — pop, visited test-and-set (LDS hash), admission — by running hnsw_walk2.hpp's
//     search_level2 with the delta result set (an admission is a register move) and this file's evaluator in place of the lane-pair
//     distance code; the neighbours' adjacency rows are fetched WITH their vectors, so the next candidate's row is on chip.
// Two workgroup barriers per expansion (LDS-only: loads requested ahead of time stay in flight across them).  The visited set is an LDS hash that is never reset here: a traversal that would need the
// reset reports err 8 and the host re-runs the call on the one-wave kernel (hnsw.hip: search_common).
#pragma once
#include "hnsw_walk2.hpp"

namespace coltt {
namespace dev {

constexpr int LAT_ROWS = 32;            // neighbours evaluated per chunk (8 per wave)
constexpr int LAT_MAX_STRIDE = 3200;    // bytes per stored row the staging area takes: f32 up to dim 800, 2-byte codes up to 1600
constexpr int LAT_PAD = 32;             // staging rows are padded by 8 dwords: the 8 rows of a wave land on disjoint LDS banks
constexpr int LAT_MAX_PIECES = LAT_MAX_STRIDE / 128;   // 16-byte pieces per lane per row (25)

struct LatShared {
  uint32_t nb[LAT_ROWS];      // neighbour slots of the chunk (NBR_NONE = none)
  uint32_t fresh[LAT_ROWS];   // 1 = evaluate
  float d[LAT_ROWS];          // distances
  uint32_t ctl[8];            // 0 state (1 go, 0 done), 1 query index, 2/3 greedy verdict
  uint32_t adjn[LAT_ROWS][32] __attribute__((aligned(16)));   // level-0 adjacency rows of the chunk's fresh neighbours (mMax0 <= 32): fetched WITH
                              // their vectors, so the next candidate's neighbour list is on chip the moment it is chosen
};
// The query as the residue lanes read it: qT[j * n8p + g] = q[8 g + j] (n8p = n8 rounded up to 4, + 4: every residue row starts on a
// 16-byte boundary and the eight rows on disjoint LDS banks), followed by the scalar tail q[8 n8 .. dim).
__device__ __host__ __forceinline__ int lat_n8p(int dim) { return (((dim >> 3) + 3) & ~3) + 4; }
__device__ __host__ __forceinline__ size_t lat_q_floats(int dim) { return (size_t)8 * lat_n8p(dim) + 8; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt — i.e. it would wait for the adjacency rows
// requested ahead of time, which are exactly the loads that must stay in flight across the barrier.  Nothing but LDS is handed
// from wave to wave in this kernel.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// element e of a staged row, decoded
template <int QUANT> __device__ __forceinline__ float lat_elem(const uint8_t* __restrict__ row, int e) {
  if constexpr (QUANT == Q_NONE) return *reinterpret_cast<const float*>(row + (size_t)e * 4);
  else if constexpr (QUANT == Q_F8) return __uint_as_float(f8bits_to_f32bits(row[e]));
  else return f16bits_to_f32(*reinterpret_cast<const unsigned short*>(row + (size_t)e * 2));
}

#ifdef COLTT_PHASE_TIMING   // diagnostic build: shader-clock ticks per phase of wave 0 (slots of WaveCtx::pt)
#define COLTT_LT(W, K) { if (wave == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); const_cast<WaveCtx&>(W).pt[K] += t_ - (W).t_last; const_cast<WaveCtx&>(W).t_last = t_; } }
#define COLTT_LT0(W, K) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); const_cast<WaveCtx&>(W).pt[K] += t_ - (W).t_last; const_cast<WaveCtx&>(W).t_last = t_; }
#else
#define COLTT_LT(W, K)
#define COLTT_LT0(W, K)
#endif
// Distances of the chunk's fresh neighbours (xs->nb / xs->fresh, written by wave 0 before the barrier) -> xs->d.
// Called by all four waves; no workgroup barrier inside (a wave stages and evaluates its own 8 rows).
// ADJ: also fetch the fresh neighbours' level-0 adjacency rows into xs->adjn (one 16-byte load per lane: 8 lanes = 32 ids).
struct LatNoMid { __device__ __forceinline__ void operator()() const {} };
// MID: called (by every wave) after the row loads have been ISSUED and before they are waited for — wave 0 runs the previous
// expansion's merge there, under the shadow of the fetch.
// TP (round 6): how the kernel instance reads the rows — each instance carries ONE of the two evaluations, not both behind a run-time branch
// (round 5's single function kept 25 per-piece predicates as live SGPR pairs from the fetch to the LDS store, every piece in its own EXEC
// region, both evaluations' state at once: 256 VGPRs + 33-42 AGPR spills + 306-386 scalar spills inside the expansion chain):
//   TP > 0   line-transposed rows (rows8.hpp) of exactly TP 128-byte lines: lane j of a row's group holds chunk j of every line = residue j's
//            consecutive steps; the pieces are evaluated straight out of the registers they landed in.  The line count is a compile-time
//            constant: one predicate (`fresh`) around straight-line code.
//   TP = -1  line-transposed rows of any length (the line count is wave-uniform: a scalar compare per line)
//   TP = 0   natural-order rows: staged through LDS as they are and read back element-wise
constexpr int LAT_TP_STAGED = 0, LAT_TP_R8_ANY = -1;
template <int METRIC, int QUANT, bool ADJ, int TP, class MID = LatNoMid>
__device__ __forceinline__ void lat_eval_chunk(const GraphView& g, const WaveCtx& w, LatShared* xs, uint8_t* stage, int wave, int lane, MID&& mid = MID()) {
  const int r = lane >> 3, j = lane & 7;            // row of this wave's eight, lane within the row
  const int idx = wave * 8 + r;
  const uint32_t nb = xs->nb[idx];
  const bool fresh = xs->fresh[idx] != 0u;
  const int n8 = g.dim >> 3, n8p = lat_n8p(g.dim);
  float rn = 0.f;
  float d = 0.f;
  const bool any = __ballot(fresh) != 0ull;
  if constexpr (TP != LAT_TP_STAGED) {
    static_assert(QUANT != Q_F8, "\"f8\" rows are never line-transposed");
    constexpr int NT = TP > 0 ? TP : LAT_MAX_PIECES;
    constexpr int S = QUANT == Q_NONE ? 4 : 8;      // steps per 128-byte line
    const int lines = TP > 0 ? TP : (int)(g.stride >> 7);   // wave-uniform
    float acc = 0.f;
    if (any) {
      // ---- fetch: every line of every fresh row of this wave in flight before the first one is used
      u32x4v tmp[NT];
      u32x4v adj = {NBR_NONE, NBR_NONE, NBR_NONE, NBR_NONE};
      const uint8_t* src = g.rows + (size_t)nb * g.stride + (size_t)j * 16;
      if (fresh) {
#pragma unroll
        for (int t = 0; t < NT; t++) { if (TP > 0 || t < lines) tmp[t] = *reinterpret_cast<const u32x4v*>(src + (size_t)t * 128); }
        if constexpr (ADJ) { if ((uint32_t)(j * 4) < g.mMax0) adj = *reinterpret_cast<const u32x4v*>(g.adj0 + (size_t)nb * g.mMax0 + j * 4); }
        if constexpr (METRIC == M_COS) rn = g.norms[nb];
      }
      mid();
      if (fresh) {
        const float* qT8 = w.qs + (size_t)j * n8p;
#pragma unroll
        for (int t = 0; t < NT; t++) {
          if (TP > 0 || t < lines) {
            if constexpr (QUANT == Q_NONE) {
              const f32x4 x = __builtin_bit_cast(f32x4, tmp[t]);
              const f32x4 q = *reinterpret_cast<const f32x4*>(qT8 + S * t);
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if constexpr (METRIC == M_COS) { const float pp = q[u] * x[u]; acc = acc + pp; }
                else { const float df = q[u] - x[u]; const float pp = df * df; acc = acc + pp; }
              }
            } else {
              const u32x2e lo = {tmp[t].x, tmp[t].y}, hi = {tmp[t].z, tmp[t].w};
              const f32x4 x0 = __builtin_convertvector(__builtin_bit_cast(f16x4, lo), f32x4);
              const f32x4 x1 = __builtin_convertvector(__builtin_bit_cast(f16x4, hi), f32x4);
              const f32x4 q0 = *reinterpret_cast<const f32x4*>(qT8 + S * t), q1 = *reinterpret_cast<const f32x4*>(qT8 + S * t + 4);
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if constexpr (METRIC == M_COS) { const float pp = q0[u] * x0[u]; acc = acc + pp; }
                else { const float df = q0[u] - x0[u]; const float pp = df * df; acc = acc + pp; }
              }
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if constexpr (METRIC == M_COS) { const float pp = q1[u] * x1[u]; acc = acc + pp; }
                else { const float df = q1[u] - x1[u]; const float pp = df * df; acc = acc + pp; }
              }
            }
          }
        }
        if constexpr (ADJ) *reinterpret_cast<u32x4v*>(&xs->adjn[idx][j * 4]) = adj;
      }
      COLTT_LT(w, 2)   // fetch issued + landed + evaluated
      // ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)): butterfly over the 8 lanes of the row (all lanes take part in the DPP moves)
      float s = acc;
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x141, 0xf, 0xf, true));   // row_half_mirror: lane i <-> 7 - i
      if (fresh) {   // (a line-transposed row has no scalar tail: its length is a multiple of 128 bytes, i.e. of 8 elements)
        if constexpr (METRIC == M_COS) d = cos_epilogue(s, w.qnorm, rn);
        else d = go_sqrt(s);
      }
    } else mid();
    if (j == 0) xs->d[idx] = d;
    COLTT_LT(w, 3)
    return;
  } else {
    const int lds_stride = (int)g.stride + LAT_PAD;
    uint8_t* srow = stage + (size_t)idx * lds_stride;
    const int pieces = (int)(g.stride >> 4);          // 16-byte pieces per row (the stride is a multiple of 16)
    if (any) {
      // ---- fetch: every piece of every fresh row of this wave in flight before the first one is stored
      u32x4v tmp[LAT_MAX_PIECES];
      u32x4v adj = {NBR_NONE, NBR_NONE, NBR_NONE, NBR_NONE};
      const uint8_t* src = g.rows + (size_t)nb * g.stride;
#pragma unroll
      for (int t = 0; t < LAT_MAX_PIECES; t++) {
        const int pc = t * 8 + j;
        if (fresh && pc < pieces) tmp[t] = *reinterpret_cast<const u32x4v*>(src + (size_t)pc * 16);
      }
      if constexpr (ADJ) { if (fresh && (uint32_t)(j * 4) < g.mMax0) adj = *reinterpret_cast<const u32x4v*>(g.adj0 + (size_t)nb * g.mMax0 + j * 4); }
      if constexpr (METRIC == M_COS) { if (fresh) rn = g.norms[nb]; }
      mid();
#pragma unroll
      for (int t = 0; t < LAT_MAX_PIECES; t++) {
        const int pc = t * 8 + j;
        if (fresh && pc < pieces) *reinterpret_cast<u32x4v*>(srow + (size_t)pc * 16) = tmp[t];
      }
      COLTT_LT(w, 2)   // fetch issued + landed + staged
      if constexpr (ADJ) { if (fresh) *reinterpret_cast<u32x4v*>(&xs->adjn[idx][j * 4]) = adj; }
    } else mid();
    wave_sync();   // the rows of this wave were staged by this wave
    if (any) {
      // ---- evaluate out of LDS: lane j = residue j of the 8-lane accumulator, groups in increasing order.  Blocks of 16 groups:
      // the 16 row elements and the 16 query elements (4 x ds_read_b128 of the transposed query) are requested together, then
      // the 16 multiply / add pairs run in order.
      float acc = 0.f;
      const float* qT = w.qs + (size_t)j * n8p;
      if (fresh) {
        int gq = 0;
        // The LDS reads of block b + 1 are in flight while block b's 16 multiply / add pairs run (the adds of a block otherwise wait for
        // its reads: ~1.1 us per chunk of 32 rows at dim 768 against ~0.4 us of dependent VALU).  A/B at 10 M x 768 f32, ef 128
        // (profiles/r04_latency_evalpipe_ab.txt): 1 query 1.032 -> 1.015 ms, 16 queries 1.260 -> 1.239, 128 queries 1.444 -> 1.425;
        // same answers and counters.  Adopted in round 4 (it was the -DCOLTT_LAT_EVAL_PIPE experiment of round 3).
#define COLTT_LAT_LD(G0, RV, QV)                                                                         \
        {                                                                                                  \
          _Pragma("unroll") for (int u = 0; u < 16; u++) RV[u] = lat_elem<QUANT>(srow, 8 * ((G0) + u) + j); \
          _Pragma("unroll") for (int u = 0; u < 4; u++) QV[u] = *reinterpret_cast<const f32x4*>(qT + (G0) + 4 * u); \
        }
#define COLTT_LAT_ACC(RV, QV)                                                                            \
        {                                                                                                  \
          _Pragma("unroll") for (int u = 0; u < 16; u++) {                                                 \
            const float qe = QV[u >> 2][u & 3];                                                            \
            if constexpr (METRIC == M_COS) { const float pp = qe * RV[u]; acc = acc + pp; }                \
            else { const float df = qe - RV[u]; const float pp = df * df; acc = acc + pp; }                \
          }                                                                                                \
        }
        {
          const int nblk = n8 >> 4;
          float rv0[16], rv1[16]; f32x4 qv0[4], qv1[4];
          if (nblk > 0) COLTT_LAT_LD(0, rv0, qv0)
          for (int b = 0; b < nblk; b += 2) {
            if (b + 1 < nblk) COLTT_LAT_LD(16 * (b + 1), rv1, qv1)
            COLTT_LAT_ACC(rv0, qv0)
            if (b + 1 < nblk) {
              if (b + 2 < nblk) COLTT_LAT_LD(16 * (b + 2), rv0, qv0)
              COLTT_LAT_ACC(rv1, qv1)
            }
          }
          gq = nblk << 4;
        }
#undef COLTT_LAT_LD
#undef COLTT_LAT_ACC
        for (; gq < n8; gq++) {
          const float rv = lat_elem<QUANT>(srow, 8 * gq + j);
          const float qe = qT[gq];
          if constexpr (METRIC == M_COS) { const float p = qe * rv; acc = acc + p; }
          else { const float df = qe - rv; const float p = df * df; acc = acc + p; }
        }
      }
      // ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)): butterfly over the 8 lanes of the row (all lanes take part in the DPP moves)
      float s = acc;
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
      s = s + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0x141, 0xf, 0xf, true));   // row_half_mirror: lane i <-> 7 - i
      if (fresh) {
        const float* qtail = w.qs + (size_t)8 * n8p;
        for (int e = n8 * 8; e < g.dim; e++) {  // scalar tail (avx.cpp:28-31,68-72), the same in every lane of the row
          const float rv = lat_elem<QUANT>(srow, e);
          const float qe = qtail[e - n8 * 8];
          if constexpr (METRIC == M_COS) s += qe * rv;
          else { const float df = qe - rv; s += df * df; }
        }
        if constexpr (METRIC == M_COS) d = cos_epilogue(s, w.qnorm, rn);
        else d = go_sqrt(s);
      }
    }
    if (j == 0) xs->d[idx] = d;
    COLTT_LT(w, 3)   // evaluation out of LDS
  }
}

// Run one chunk through the four waves: wave 0 has written xs->nb / xs->fresh.  Two barriers; afterwards xs->d is valid.
template <int METRIC, int QUANT, int TP>
__device__ __forceinline__ void lat_chunk(const GraphView& g, const WaveCtx& w, LatShared* xs, uint8_t* stage, int wave, int lane) {
  __syncthreads();
  lat_eval_chunk<METRIC, QUANT, false, TP>(g, w, xs, stage, wave, lane);
  __syncthreads();
}

// greedyClosestNeighbor (hnsw.go:320-343) on an upper level, all four waves.  cur / curd are meaningful on wave 0 and are
// handed to the other waves through the exchange words (every wave follows the same hop sequence).
template <int METRIC, int QUANT, int TP>
__device__ __forceinline__ void greedy_level_lat(const GraphView& g, WaveCtx& w, LatShared* xs, uint8_t* stage, uint32_t& cur, float& curd,
                                                 int level, int wave, int lane_in) {
  for (uint32_t hops = 0;; hops++) {
    const int lane = opaque_lane(lane_in);
    const int p = lane >> 1;
    uint32_t width;
    const uint32_t* row = adj_row(g, cur, level, width);
    unsigned long long best = ~0ull; uint32_t best_slot = NBR_NONE;
    if (hops > (1u << 20)) { w.err |= 4u; }
    for (uint32_t c0 = 0; c0 < width && hops <= (1u << 20); c0 += LAT_ROWS) {
      uint32_t nb = NBR_NONE; bool valid = false;
      if (wave == 0) {
        const uint32_t idx = c0 + (uint32_t)p;
        nb = idx < width ? row[idx] : NBR_NONE;
        valid = nb != NBR_NONE && !is_deleted(g, nb);
        if ((lane & 1) == 0) { xs->nb[p] = nb; xs->fresh[p] = valid ? 1u : 0u; }
      }
      lat_chunk<METRIC, QUANT, TP>(g, w, xs, stage, wave, lane);
      if (wave == 0) {
        const float d = xs->d[p];
        w.n_dist += __popcll(__ballot(valid && (lane & 1) == 0));
        const unsigned long long key = valid ? (((unsigned long long)__float_as_uint(d) << 32) | (c0 + (uint32_t)p)) : ~0ull;
        const unsigned long long km = wave_min_u64(key);
        if (km < best) {
          best = km;
          best_slot = (uint32_t)__builtin_amdgcn_readlane((int)nb, (int)(((uint32_t)km - c0) * 2));
        }
      }
    }
    // every wave must take the same branch: wave 0 publishes the verdict
    if (wave == 0) {
      w.n_hops++;
      const float bd = __uint_as_float((uint32_t)(best >> 32));
      const bool move = best != ~0ull && bd < curd && hops <= (1u << 20);
      if (move) { cur = best_slot; curd = bd; }
      if (lane == 0) { xs->ctl[2] = move ? 1u : 0u; xs->ctl[3] = cur; }
    }
    __syncthreads();
    const bool move = xs->ctl[2] != 0u;
    cur = xs->ctl[3];
    __syncthreads();   // ctl is rewritten by the next hop
    if (!move) break;
  }
}

// Level 0: wave 0 runs hnsw_walk2.hpp's search_level2 — delta result set (admissions are register moves, no per-chunk merge on the
// critical path), LDS-hash visited set, adjacency prefetch — with THIS evaluator: the chunk's neighbours are published, all four waves
// fetch and evaluate them (lat_eval_chunk), the distances come back through LDS.  The other waves sit in lat_companion().
// The chunk's adjacency rows come along with its vectors (xs->adjn), so the next candidate's neighbour list is on chip when it is
// chosen (search_level2: CHUNK_ADJ); the runner-up's row is requested at pop time.
template <int METRIC, int QUANT, int TP> struct LatEval {
  static constexpr bool CHUNK_ADJ = true;
  static constexpr bool SPEC = false;
  static constexpr bool RADJ = true;    // the runner-up's adjacency row is requested at pop time (the chunk's rows come along with its vectors)
  static constexpr bool ROWPF = false;
  static constexpr bool EARLY = false;
  static constexpr bool BOUNDED = false;
  static constexpr bool SPLIT = false;
  static constexpr bool SETCACHE = false;
  LatShared* xs; uint8_t* stage;
  __device__ __forceinline__ uint32_t chunk_adj(int idx, int p) const { return xs->adjn[idx][p]; }
  __device__ __forceinline__ void prefetch(uint32_t, bool, int) const {}
  __device__ __forceinline__ float operator()(const GraphView& g, const WaveCtx& w, uint32_t nb, bool fresh, float /*nrm*/, int half, int lane) const {
    const int p = lane >> 1;
    if (half == 0) { xs->nb[p] = nb; xs->fresh[p] = fresh ? 1u : 0u; }
    COLTT_LT0(w, 0)   // walk: pop + visited + admission of the previous chunk
    lds_barrier();
    if (g.mMax0 <= 32) lat_eval_chunk<METRIC, QUANT, true, TP>(g, w, xs, stage, 0, lane);
    else lat_eval_chunk<METRIC, QUANT, false, TP>(g, w, xs, stage, 0, lane);
    lds_barrier();
    COLTT_LT0(w, 4)   // barrier 2 (the slowest wave)
    return xs->d[p];
  }
};
// waves 1-3 while wave 0 walks: one round per published chunk until wave 0 clears ctl[0]
template <int METRIC, int QUANT, int TP>
__device__ __forceinline__ void lat_companion(const GraphView& g, const WaveCtx& w, LatShared* xs, uint8_t* stage, int wave, int lane) {
  for (;;) {
    lds_barrier();
    if (xs->ctl[0] == 0u) break;
    if (g.mMax0 <= 32) lat_eval_chunk<METRIC, QUANT, true, TP>(g, w, xs, stage, wave, lane);
    else lat_eval_chunk<METRIC, QUANT, false, TP>(g, w, xs, stage, wave, lane);
    lds_barrier();
  }
}

// ---- cache-warming helper workgroups (round 6: "more than one CU per query") -------------------------------------------------
// One query alone leaves 255 CUs idle, and what an expansion waits for is its fetch: ~25 fresh rows of 3 KB through ONE CU's load path (~1.4 us of
// transfer behind ~1 us of HBM + translation latency; profiles/r06e_latency_kernel_phases.txt: 79 % of the walk).  Handing the rows' EVALUATION to other
// CUs costs two cross-CU hand-offs per expansion (>= 0.5-1 us each): as much as it saves.  What costs the walk nothing is a HINT: the walking
// workgroup posts the slots of its best unexpanded candidates (the runner-up and the members behind it — one of them is the next candidate unless
// the expansion in flight admits a nearer vertex) in a mailbox in HBM; helper workgroups on OTHER CUs of the SAME XCD poll it, read the candidate's
// adjacency row and simply LOAD every byte of its neighbours' rows.  The rows land in the XCD's L2 (and the translations in its shared TLB level); when
// the walk gets there, its own fetch is an L2 hit.  Helpers never compute, never write anything the walk reads: the walk's loads, arithmetic, answers
// and counters are exactly what they are without helpers — a wrong or late hint only wastes a helper's bandwidth.
// MEASURED (GPU call F, profiles/r06f_latency_helpers_ab.md; 10 M x 768 f32, one index, one process, answers identical in every arm): one query, ef 128:
// 0.719 ms without helpers, 0.740 / 0.741 / 0.758 ms with 1 / 2 / 3; ef 512: 2.82 against 2.89 / 2.91 / 2.96.  The hint is late by construction — a
// helper needs the poll, the candidate's adjacency row and then the rows (~2-3 us) while the walk wants the runner-up's rows ~1.5 us after it posted
// them — and what an expansion waits for is mostly the 75-96 KB moving through the walking CU's own load path, which an L2 hit does not shorten.  OFF by
// default (COLTT_LAT_HELPERS=0); kept as the A/B partner of that record.
// Mailbox of walking workgroup m (m = blockIdx.x < LAT_MASTERS): LAT_HELPERS_MAX + 1 8-byte words — word h = (sequence number << 32 | slot) for helper
// h, the last word = done.  Single 8-byte relaxed agent-scope stores and loads (one granule: untorn); no ordering is needed — a helper acts on whatever
// slot it sees.  Block b runs on XCD b % 8 (observed placement, used for speed only: a helper on another XCD still warms the Infinity Cache), so walking
// workgroup m's helpers are the blocks m + 8, m + 16, ...
constexpr int LAT_MASTERS = 8;        // walking workgroups of a helped launch (one per XCD): batches of at most 8 queries
constexpr int LAT_HELPERS_MAX = 3;    // helper workgroups per walking workgroup
__device__ __forceinline__ void lat_post_hint(unsigned long long* box, int helpers, int h, uint32_t seq, uint32_t slot, int lane) {   // wave 0; one lane stores
  if (h < helpers && lane == 0) __hip_atomic_store(box + h, ((unsigned long long)seq << 32) | slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// a helper workgroup: until the walking workgroup is done (or a generous spin budget runs out — nothing waits for a helper)
__device__ __forceinline__ void lat_helper_loop(const GraphView& g, unsigned long long* box, int h, LatShared* xs) {
  const int tid = threadIdx.x;
  unsigned long long last = 0ull; uint32_t sink = 0u;
  for (uint32_t spins = 0; spins < (1u << 22); spins++) {
    const unsigned long long v = __hip_atomic_load(box + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long done = __hip_atomic_load(box + LAT_HELPERS_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done) break;
    if (v == last || v == 0ull) { __builtin_amdgcn_s_sleep(8); continue; }
    last = v;
    const uint32_t slot = (uint32_t)v;
    __syncthreads();   // (everybody is through with the previous hint's neighbour list)
    if (tid < LAT_ROWS) xs->nb[tid] = (uint32_t)tid < g.mMax0 ? g.adj0[(size_t)slot * g.mMax0 + tid] : NBR_NONE;
    __syncthreads();
    const uint32_t pieces = (uint32_t)(g.stride >> 4);
    for (int i = 0; i < LAT_ROWS; i++) {
      const uint32_t nb = xs->nb[i];
      if (nb == NBR_NONE) continue;   // workgroup-uniform
      const uint8_t* row = g.rows + (size_t)nb * g.stride;
      for (uint32_t pc = (uint32_t)tid; pc < pieces; pc += 256u) { const u32x4v x = *reinterpret_cast<const u32x4v*>(row + (size_t)pc * 16); sink ^= x.x ^ x.w; }
    }
    if (sink == 0x9e3779b9u && spins == 0xffffffffu) box[h] = sink;   // (never true: keeps the loads)
  }
}

// searchLevel on level 0 for rows of at most one chunk (mMax0 <= 32), SOFTWARE-PIPELINED over expansions, all four waves.
// With LatEval an expansion is  pop -> visited -> [fetch -> evaluate] -> admission / eviction  in sequence on wave 0, ~3.4 us of
// single-wave VALU / LDS work around a ~3.5 us fetch + evaluation (phase timing, profiles/r03_latency.json).  Here the NEXT candidate is
// chosen the moment the distances are known — it is min(runner-up, smallest admitted key), both available without touching the
// set — its neighbours (their adjacency row came along with the vectors) are tested and published at once, all four waves start
// fetching them, and wave 0 does the previous expansion's set work under the shadow of its own loads (the `mid` hook of
// lat_eval_chunk): admitted keys into the delta, eviction down to ef, lowerBound / free slots for the expansion in flight, and the
// next runner-up, whose adjacency row is requested right then.  Same set, same pops, same counters as hnsw_walk2.hpp:
//   * the set after the admissions is top-ef(S ∪ admitted); its smallest unexpanded member is the smaller of S's smallest unexpanded
//     member other than the candidate just expanded (the runner-up) and the smallest admitted key;
//   * if that member does not survive the truncation to ef (it is larger than the new worst member), the canonical loop would find
//     nothing to expand and stop: the speculative fetch is dropped, nothing of it is counted.
template <int METRIC, int QUANT, int TP>
__device__ __forceinline__ void search_level_lat3(const GraphView& g, WaveCtx& w, LatShared* xs, uint8_t* stage, uint32_t ep, float epd,
                                                  uint32_t ef, int wave, int lane_in, uint32_t& out_len, unsigned long long* hint_box, int hint_helpers) {
  int lane = lane_in;
  const int tid = wave * 64 + lane;
  unsigned long long* const res = w.res0;
  const uint32_t width = g.mMax0;
  for (uint32_t i = tid; i < w.hcap; i += 256) w.vis[i] = VIS_EMPTY;
  __syncthreads();
  uint32_t len = 1, vis_count = 1, scan_lo = 1;
  Delta dl; dl.clear();
  float lower_bound = epd; uint32_t free_slots = ef - 1;
  unsigned long long runner_key = ~0ull; uint32_t runner_nb = NBR_NONE; int runner_idx = -1, runner_dlane = -1;
  uint32_t nb = NBR_NONE; bool fresh = false;       // the expansion in flight (wave 0, lane pair p <-> neighbour p)
  unsigned long long pA = 0, pnext = ~0ull; uint32_t pkhi = 0, pklo = 0, pm = 0; bool locate = false, dead = false;
  uint32_t hint_seq = 0;   // sequence number of the hints posted to the helper workgroups

  auto absorb = [&]() {   // the admitted keys of the previous expansion: into the delta, then keep the ef smallest
    if (!pm) return;
    if (dl.n + pm > 64u) delta_flush(res, len, dl, scan_lo, lane);
    unsigned long long am = pA;
    while (am) {
      const int jj = __builtin_ctzll(am); am &= am - 1;
      dl.insert((uint32_t)__builtin_amdgcn_readlane((int)pkhi, jj), (uint32_t)__builtin_amdgcn_readlane((int)pklo, jj), lane);
    }
    pm = 0;
    const uint32_t total = len + dl.n;
    if (total > ef) evict_largest(res, len, dl, total - ef, lane);
  };
  auto mid = [&]() {      // wave 0, under the shadow of the fetch
    if (wave != 0) return;
    absorb();
    if (!locate) return;
    locate = false;
    unsigned long long worst = len ? (res[len - 1] & ~1ull) : 0ull;
    { const unsigned long long dmx = dl.max_key(); worst = dmx > worst ? dmx : worst; }
    if (pnext > worst) { dead = true; return; }   // the candidate in flight was truncated away: the canonical loop ends here
    w.n_exp++;
    lower_bound = __uint_as_float((uint32_t)(worst >> 32));
    free_slots = ef - (len + dl.n);
    // the runner-up: the smallest unexpanded member of main ∪ delta (the candidate in flight is already marked)
    int ci = -1;
    unsigned long long more = 0ull; uint32_t more_base = 0;   // the unexpanded main members behind the first one, in its 64-entry chunk (hints)
    for (uint32_t base = scan_lo & ~63u; base < len; base += 64) {
      const uint32_t i = base + lane;
      const bool un = i < len && !(res[i] & 1ull);
      const unsigned long long mm = __ballot(un);
      if (mm) { ci = (int)base + __builtin_ctzll(mm); more = mm & (mm - 1ull); more_base = base; break; }
    }
    scan_lo = ci >= 0 ? (uint32_t)ci : len;
    const unsigned long long kci = ci >= 0 ? res[ci] : ~0ull;
    unsigned long long kd = ~0ull; int dlane = -1;
    { const unsigned long long u = dl.unexpanded(lane); if (u) { dlane = __builtin_ctzll(u); kd = dl.key_at(dlane); } }
    const bool runner_in_delta = kd < kci;
    if (runner_in_delta) { runner_key = kd; runner_dlane = dlane; runner_idx = -1; }
    else { runner_key = kci; runner_idx = ci; runner_dlane = -1; }
    runner_nb = NBR_NONE;
    if (runner_key != ~0ull && (uint32_t)(lane >> 1) < width) runner_nb = g.adj0[(size_t)((uint32_t)runner_key >> 1) * width + (lane >> 1)];
    // hints for the helper workgroups: the runner-up, then the main array's next unexpanded members (the delta's are not looked for: a hint, not a promise)
    if (hint_box && runner_key != ~0ull) {
      hint_seq++;
      lat_post_hint(hint_box, hint_helpers, 0, hint_seq, (uint32_t)runner_key >> 1, lane);
      int hh = 1;
      if (runner_in_delta && ci >= 0) { lat_post_hint(hint_box, hint_helpers, hh, hint_seq, (uint32_t)kci >> 1, lane); hh++; }   // the runner-up came from the delta: the main array's first one is next in line
      while (more && hh < hint_helpers) {
        const int l1 = __builtin_ctzll(more); more &= more - 1ull;
        lat_post_hint(hint_box, hint_helpers, hh, hint_seq, (uint32_t)res[more_base + (uint32_t)l1] >> 1, lane); hh++;
      }
    }
  };

  if (wave == 0) {   // the entrypoint is popped at once (it is the only member)
    const int half = lane & 1, p = lane >> 1;
    if (lane == 0) { res[0] = ((unsigned long long)__float_as_uint(epd) << 32) | ((unsigned long long)ep << 1) | 1ull; vis_insert(w.vis, w.hcap_mask, ep); }
    w.n_exp++;
    nb = (uint32_t)p < width ? g.adj0[(size_t)ep * width + p] : NBR_NONE;
    const bool valid = nb != NBR_NONE && !is_deleted(g, nb);
    wave_sync();
    int fresh_i = 0;
    if (valid && half == 0) fresh_i = vis_insert(w.vis, w.hcap_mask, nb) ? 1 : 0;
    fresh_i = __builtin_amdgcn_mov_dpp(fresh_i, 0xA0, 0xf, 0xf, true);
    fresh = fresh_i != 0;
    if (half == 0) { xs->nb[p] = nb; xs->fresh[p] = (uint32_t)fresh_i; }
    if (lane == 0) xs->ctl[0] = 1u;
  }
  for (uint32_t iters = 0;; iters++) {
    lane = opaque_lane(lane_in);
    const int half = lane & 1, p = lane >> 1;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    COLTT_LT(w, 0)   // choose + visited + publish (wave 0)
    lds_barrier();
    if (xs->ctl[0] == 0u) break;
    lat_eval_chunk<METRIC, QUANT, true, TP>(g, w, xs, stage, wave, lane, mid);
    lds_barrier();
    COLTT_LT(w, 4)
    if (wave == 0) {
      if (dead || iters > (1u << 22)) { if (!dead) w.err |= 2u; if (lane == 0) xs->ctl[0] = 0u; continue; }
      // ---- which of the distances that just arrived are admitted (hnsw.go:374: the stale lowerBound, the free slots first)
      const unsigned long long E = __ballot(fresh && half == 0);
      const uint32_t nfresh = __popcll(E);
      unsigned long long A = 0, best_new = ~0ull; uint32_t m = 0, khi = 0, klo = 0; bool adm = false; int best_lane = -1;
      if (nfresh) {
        vis_count += nfresh; w.n_dist += nfresh;
        const float d = xs->d[p];
        const uint32_t rank = __popcll(E & lt_mask);
        adm = fresh && half == 0 && (rank < free_slots || d < lower_bound);
        A = __ballot(adm); m = __popcll(A);
        khi = __float_as_uint(d); klo = nb << 1;
        if (m) best_lane = wave_argmin_key(adm, khi, klo, best_new);
      }
      // ---- the next candidate, its neighbours, their visited test
      const unsigned long long nk = runner_key < best_new ? runner_key : best_new;
      bool go = nk != ~0ull;
      if (go && vis_count + 64 > (w.hcap >> 2) * 3) { w.err |= 8u; go = false; }   // would need the reset path: give up (host falls back)
      if (go) {
        pnext = nk; locate = true;
        if (runner_key < best_new) {   // the runner-up: mark it where it sits (nothing has moved since it was found)
          if (runner_dlane >= 0) { if (lane == runner_dlane) dl.lo |= 1u; }
          else if (lane == 0) res[runner_idx] = runner_key | 1ull;
          nb = runner_nb;
        } else {                       // a vertex admitted just now: it enters the set already marked; its adjacency row came with its vector
          if (lane == best_lane) klo |= 1u;
          nb = (uint32_t)p < width ? xs->adjn[best_lane >> 1][p] : NBR_NONE;
        }
        const bool valid = nb != NBR_NONE && !is_deleted(g, nb);
        int fresh_i = 0;
        if (valid && half == 0) fresh_i = vis_insert(w.vis, w.hcap_mask, nb) ? 1 : 0;
        fresh_i = __builtin_amdgcn_mov_dpp(fresh_i, 0xA0, 0xf, 0xf, true);
        fresh = fresh_i != 0;
        if (half == 0) { xs->nb[p] = nb; xs->fresh[p] = (uint32_t)fresh_i; }
      }
      pA = A; pm = m; pkhi = khi; pklo = klo;
      if (lane == 0) xs->ctl[0] = go ? 1u : 0u;
      COLTT_LT(w, 5)
    }
  }
  if (wave == 0) { absorb(); delta_flush(res, len, dl, scan_lo, lane); }   // the last expansion's admissions; one sorted array for the caller
  out_len = len;   // meaningful on wave 0
}

}  // namespace dev
}  // namespace coltt
