// select.hpp — top-k selection shared by the FLAT and CFLAT stores.
#pragma once
#include "exact.hpp"

namespace coltt {
namespace dev {

constexpr uint32_t K_MAX = 2048;  // largest top-k served by flat_select's LDS rank sort
constexpr uint32_t SELECT_SMALL = 512;  // candidate lists up to this long are rank-sorted whole (no radix passes)

// ---------------------------------------------------------------------------------------------------
// Selection: the bounded queue + ToSlice of edge.PriorityQueue (edge/priority_queue.go:39-69) in closed
// form.  REFERENCE keeps the K LARGEST (score,id) keys, NEAREST the K smallest; output ascending.
// One block per query: radix-select on the score key (4 x 8 bits), a second radix-select on the id among
// boundary ties only when they do not all fit, then an LDS rank sort of the <= K survivors.
// ---------------------------------------------------------------------------------------------------
static __device__ __forceinline__ uint64_t slot_id(const uint64_t* ids, uint64_t dense_base, uint32_t slot) {
  return ids ? ids[slot] : dense_base + slot;
}

// rank of (ki, ii) among the n4 (a multiple of 4; padded with (0xffffffff, ~0) sentinels, which rank after everything) entries of the
// LDS arrays: four entries per LDS round trip.  (One entry per iteration leaves every iteration waiting for its own two LDS reads:
// 95 us for a 512-entry list — most of a single-query FLAT search.)
typedef uint32_t sel_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long sel_u64x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t sel_rank(const uint32_t* __restrict__ keys, const uint64_t* __restrict__ ids, uint32_t n4, uint32_t ki, uint64_t ii) {
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n4; j += 4) {
    const sel_u32x4 kj = *reinterpret_cast<const sel_u32x4*>(keys + j);
    const sel_u64x2 ia = *reinterpret_cast<const sel_u64x2*>(ids + j), ib = *reinterpret_cast<const sel_u64x2*>(ids + j + 2);
    rank += ((kj.x < ki) || (kj.x == ki && ia.x < ii)) ? 1u : 0u;
    rank += ((kj.y < ki) || (kj.y == ki && ia.y < ii)) ? 1u : 0u;
    rank += ((kj.z < ki) || (kj.z == ki && ib.x < ii)) ? 1u : 0u;
    rank += ((kj.w < ki) || (kj.w == ki && ib.y < ii)) ? 1u : 0u;
  }
  return rank;
}

// digit search of a radix-select pass by the whole block: the first bin b with hist[0..b] >= need, and what is still needed inside it.
// (A serial scan of the 256 bins by one thread is 256 dependent LDS reads, ~7 us per pass; twelve passes in the worst case.)
static __device__ __forceinline__ void sel_find_digit(const uint32_t* __restrict__ hist, uint32_t need, uint32_t* __restrict__ wave_tot,
                                                      uint32_t* __restrict__ s_digit, uint32_t* __restrict__ s_need) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t h = hist[tid];
  uint32_t inc = h;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64);
    if (lane >= d) inc += up;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; w++) base += wave_tot[w];
  inc += base;
  const uint32_t exc = inc - h;
  if (exc < need && need <= inc) { *s_digit = (uint32_t)tid; *s_need = need - exc; }
}

// The selection of query q by ONE 256-thread block (every thread of the block calls it; the LDS arrays are reused by the next call
// after a __syncthreads()).  flat_select_kernel runs it with one block per query; the fused small-batch search (flat_one.hpp) runs
// it in the last block to finish.
template <uint32_t KCAP = K_MAX>   // LDS capacity: >= k and >= SELECT_SMALL (the one-launch kernel serves k <= 64 and sizes it 512)
static __device__ __forceinline__ void flat_select_block(
    const int q, unsigned long long* __restrict__ cand_all, uint32_t* __restrict__ cnt_all, uint32_t* __restrict__ thr_all,
    uint32_t cap, uint32_t k, int nearest, const uint64_t* __restrict__ ids, uint64_t dense_base,
    uint32_t* __restrict__ overflow, uint64_t* __restrict__ out_ids, float* __restrict__ out_scores,
    uint32_t* __restrict__ out_counts) {
  static_assert(KCAP >= SELECT_SMALL, "the short-list path keeps every candidate in LDS");
  __shared__ uint32_t hist[256];
  __shared__ __attribute__((aligned(16))) uint32_t sel_key[KCAP + 4];
  __shared__ uint32_t sel_slot[KCAP];
  __shared__ __attribute__((aligned(16))) uint64_t sel_id[KCAP + 4];
  __shared__ uint32_t s_digit, s_need, s_nsel, s_wtot[4];
  const int tid = threadIdx.x;
  unsigned long long* cand = cand_all + (size_t)q * cap;
  uint32_t c = cnt_all[q];
  if (c > cap) { if (tid == 0) atomicOr(overflow, 1u); c = cap; }
  const uint32_t kk = k < c ? k : c;
  if (kk == 0) {
    if (tid == 0) { out_counts[q] = 0; cnt_all[q] = 0; thr_all[q] = nearest ? 0xffffffffu : 0u; }
    return;
  }
  const uint32_t flip = nearest ? 0u : 0xffffffffu;  // key' = key ^ flip : we want the kk smallest key'
  // ---- short lists (what is left behind a tight threshold, and every matrix-core search after its exact re-score): one LDS
  // rank sort of ALL candidates by (key', id') replaces the radix passes — same total order, same survivors, ~5x less latency
  if (c <= SELECT_SMALL) {
    const uint64_t idflip_s = nearest ? 0ull : ~0ull;
    for (uint32_t i = tid; i < c; i += 256) {
      const unsigned long long e = cand[i];
      sel_key[i] = (uint32_t)(e >> 32) ^ flip; sel_slot[i] = (uint32_t)e; sel_id[i] = slot_id(ids, dense_base, (uint32_t)e) ^ idflip_s;
    }
    const uint32_t c4 = (c + 3u) & ~3u;
    if (tid < c4 - c) { sel_key[c + tid] = 0xffffffffu; sel_id[c + tid] = ~0ull; }
    __syncthreads();   // every candidate is in LDS: the list can be rewritten in place
    for (uint32_t i = tid; i < c; i += 256) {
      const uint32_t ki = sel_key[i]; const uint64_t ii = sel_id[i];
      const uint32_t rank = sel_rank(sel_key, sel_id, c4, ki, ii);
      if (rank >= kk) continue;
      const uint32_t pos = nearest ? rank : (kk - 1 - rank);
      const uint32_t key = ki ^ flip;
      out_ids[(size_t)q * k + pos] = ii ^ idflip_s;
      out_scores[(size_t)q * k + pos] = key_score(key);
      cand[pos] = ((unsigned long long)key << 32) | sel_slot[i];
      if (rank == kk - 1) thr_all[q] = (kk == k) ? key : (nearest ? 0xffffffffu : 0u);   // the boundary key is the next threshold
    }
    if (tid == 0) { out_counts[q] = kk; cnt_all[q] = kk; }
    return;
  }
  // ---- pass 1: radix-select the kk-th smallest key'
  uint32_t prefix = 0, mask = 0, need = kk;
  for (int pass = 3; pass >= 0; pass--) {
    const int shift = pass * 8;
    hist[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < c; i += 256) {
      uint32_t kp = (uint32_t)(cand[i] >> 32) ^ flip;
      if ((kp & mask) == prefix) atomicAdd(&hist[(kp >> shift) & 255u], 1u);
    }
    __syncthreads();
    sel_find_digit(hist, need, s_wtot, &s_digit, &s_need);
    __syncthreads();
    prefix |= s_digit << shift; mask |= 255u << shift; need = s_need;
    __syncthreads();
  }
  const uint32_t T = prefix;            // boundary key'
  const uint32_t ties = hist[s_digit];  // elements with key' == T (last pass histogram)
  // ---- boundary ties that do not all fit: radix-select on id' among them
  const uint64_t idflip = nearest ? 0ull : ~0ull;
  uint64_t idT = ~0ull;  // take ties with id' <= idT
  if (ties > need) {
    uint64_t ipre = 0, imask = 0; uint32_t ineed = need;
    for (int pass = 7; pass >= 0; pass--) {
      const int shift = pass * 8;
      __syncthreads();
      hist[tid] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < c; i += 256) {
        unsigned long long e = cand[i];
        if (((uint32_t)(e >> 32) ^ flip) != T) continue;
        uint64_t ip = slot_id(ids, dense_base, (uint32_t)e) ^ idflip;
        if ((ip & imask) == ipre) atomicAdd(&hist[(ip >> shift) & 255ull], 1u);
      }
      __syncthreads();
      sel_find_digit(hist, ineed, s_wtot, &s_digit, &s_need);
      __syncthreads();
      ipre |= (uint64_t)s_digit << shift; imask |= 255ull << shift; ineed = s_need;
    }
    idT = ipre;
  }
  // ---- gather the survivors
  if (tid == 0) s_nsel = 0;
  __syncthreads();
  for (uint32_t i = tid; i < c; i += 256) {
    unsigned long long e = cand[i];
    uint32_t kp = (uint32_t)(e >> 32) ^ flip;
    if (kp > T) continue;
    uint64_t ip = slot_id(ids, dense_base, (uint32_t)e) ^ idflip;
    if (kp == T && ip > idT) continue;
    uint32_t j = atomicAdd(&s_nsel, 1u);
    if (j < KCAP) { sel_key[j] = kp; sel_slot[j] = (uint32_t)e; sel_id[j] = ip; }
  }
  __syncthreads();
  const uint32_t ns = s_nsel < kk ? s_nsel : kk;  // == kk by construction
  const uint32_t ns4 = (ns + 3u) & ~3u;
  if (tid < ns4 - ns) { sel_key[ns + tid] = 0xffffffffu; sel_id[ns + tid] = ~0ull; }
  __syncthreads();
  // ---- rank sort by (key', id'); emit ascending by (score, id)
  for (uint32_t i = tid; i < ns; i += 256) {
    uint32_t ki = sel_key[i]; uint64_t ii = sel_id[i];
    const uint32_t rank = sel_rank(sel_key, sel_id, ns4, ki, ii);
    uint32_t pos = nearest ? rank : (ns - 1 - rank);
    uint32_t key = ki ^ flip;
    out_ids[(size_t)q * k + pos] = ii ^ idflip;
    out_scores[(size_t)q * k + pos] = key_score(key);
    cand[pos] = ((unsigned long long)key << 32) | sel_slot[i];
  }
  if (tid == 0) {
    out_counts[q] = ns; cnt_all[q] = ns;
    thr_all[q] = (ns == k) ? (T ^ flip) : (nearest ? 0xffffffffu : 0u);
  }
}

static __global__ __launch_bounds__(256) void flat_select_kernel(
    unsigned long long* __restrict__ cand_all, uint32_t* __restrict__ cnt_all, uint32_t* __restrict__ thr_all,
    uint32_t cap, uint32_t k, int nearest, const uint64_t* __restrict__ ids, uint64_t dense_base,
    uint32_t* __restrict__ overflow, uint64_t* __restrict__ out_ids, float* __restrict__ out_scores,
    uint32_t* __restrict__ out_counts) {
  flat_select_block<K_MAX>((int)blockIdx.x, cand_all, cnt_all, thr_all, cap, k, nearest, ids, dense_base, overflow, out_ids, out_scores, out_counts);
}

// group state: cnt[256] | thr[256] | overflow
static __global__ void init_group_kernel(uint32_t* cnt, uint32_t* thr, uint32_t* overflow, int nearest) {
  int q = threadIdx.x;
  if (q < 256) { cnt[q] = 0; thr[q] = nearest ? 0xffffffffu : 0u; }
  if (q == 0) *overflow = 0;
}



}  // namespace dev
}  // namespace coltt
