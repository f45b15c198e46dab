// flat_mfma.hpp — matrix-core candidate generation for the batched FLAT scan (kernel K2 of SURVEY.md §2).
//
// The reference scores one (query, vector) pair at a time (edge/none_vectorstore.go:136-147); a GPU serving a batch of B
// queries turns the scan into a [rows x D] x [D x B] GEMM.  MFMA accumulates in a different order than the reference's AVX
// kernel, so its scores are NOT the reference's bits.  They are used only to pick candidates:
//
//   1. flat_mfma_cos_kernel     : s~ = |1 - dot/sqrt(nq*nr)| for every (row, query) of a 128-row tile with
//      v_mfma_f32_32x32x16_f16 (f16 x f16 products are exact in f32; only the summation order differs), fused
//      threshold filter, survivors appended to the per-query candidate list — the score matrix is never materialised;
//   2. flat_pick_kernel        : k-th best s~ per query, keep everything within MARGIN of it, publish the bound as the
//      next scan threshold;
//   3. flat_rescore_kernel     : the survivors are re-scored with the exact-order kernel (exact.hpp), then the ordinary
//      flat_select_kernel picks the top-k.
//
// Exactness of the returned set: if |s~ - s| <= d for every pair, every true top-k member has s~ <= s~_(k) + 2d, so with
// MARGIN = 2d the candidate set is a superset of the exact top-k and the final answer (ids, ranks, score bits) is the
// exact path's.  For unit vectors of dimension D <= 4096 stored as binary16: products exact, f32 accumulation error
// <= D * 2^-24 * sum|a_i b_i| <= 2.5e-4 (worst case; ~1e-6 typical), epilogue ~1e-6  =>  d = 3e-4, MARGIN = 6e-4.
#pragma once
#include "exact.hpp"

namespace coltt {
namespace dev {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifndef COLTT_MF_BM
#define COLTT_MF_BM 128
#endif
constexpr int MF_BM = COLTT_MF_BM; // rows per workgroup tile
#ifndef COLTT_MF_BK
#define COLTT_MF_BK 32
#endif
constexpr int MF_BK = COLTT_MF_BK; // halves per K step; 32 (64 B of every row per step) keeps stages small => 2 workgroups per CU
constexpr int MF_CPR = MF_BK / 8; // 16-byte chunks per row per K step
constexpr int MF_LD = MF_BK + 8;  // padded LDS row (80 B = 5 slots, odd => conflict-free ds_read_b128 fragment reads)
constexpr float MF_MARGIN = 6e-4f;
// f32 rows rounded to binary16 for candidate generation: |x~ - x| <= 2^-11 |x| per element of row AND query =>
// |d dot| <= (2^-10 + 2^-22) sum|a_i b_i| <= 9.8e-4 for unit vectors, + accumulation 2.5e-4, + epilogue => d = 1.3e-3.
constexpr float MF_MARGIN_F32 = 2.6e-3f;

template <int BN> constexpr size_t mfma_lds_bytes() { return (size_t)2 * (MF_BM + BN) * MF_LD * 2 + MF_BM * 4; }

// queries as f16 [BN][dim]: exact for the 2-byte stores (q_eff went through Lower), rounded to nearest for f32 stores
// (also resets the group state cnt[256] | thr[256] | overflow: one launch less in the chain)
// qscale: the query is multiplied by it on its way to binary16 ("f8" stores: 2^24, flat.hip f8_expand_kernel), qn_out = qn_in * qscale^2
__global__ void mfma_prep_queries_kernel(const float* __restrict__ q_eff, int nq, int bn, int dim, int dimp, _Float16* __restrict__ q16,
                                         uint32_t* __restrict__ cnt, uint32_t* __restrict__ thr, uint32_t* __restrict__ overflow, int nearest,
                                         float qscale = 1.0f, const float* __restrict__ qn_in = nullptr, float* __restrict__ qn_out = nullptr) {
  if (blockIdx.x == 0) {
    if (threadIdx.x < 256) { cnt[threadIdx.x] = 0; thr[threadIdx.x] = nearest ? 0xffffffffu : 0u; }
    if (threadIdx.x == 0) *overflow = 0;
    if (qn_out && (int)threadIdx.x < nq) qn_out[threadIdx.x] = qn_in[threadIdx.x] * qscale * qscale;
  }
  // q16 = [bn][dimp]: queries beyond nq and columns beyond dim (K padded to whole 32-column steps) are zero
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)bn * dimp) return;
  const int q = (int)(i / dimp), e = (int)(i - (size_t)q * dimp);
  q16[i] = (q < nq && e < dim) ? (_Float16)(q_eff[(size_t)q * dim + e] * qscale) : (_Float16)0.f;
}

// ---- epilogue shared by both GEMM kernels -----------------------------------------------------------------------------
// A lane holds, per 32x32 block, the dots of ONE query against 16 rows.  Testing every element costs ~6 VALU + a branch each,
// and with two workgroups per CU little of that hides behind another wave's MFMAs.  So the block is filtered first: t_r = dot_r / ||row_r|| (= cos * ||q||), one max (or min/max) over the 16
// values, ONE compare against a per-query bound that is the exact test loosened by MF_BLOCK_SLACK; only blocks that may
// hold a survivor — rare once the threshold has tightened — run the exact per-element test.  Blocks containing a row whose
// norm is zero / inf / NaN always take the element path (`bad`), so NaN scores still reach the exact re-score.
constexpr float MF_BLOCK_SLACK = 1e-5f;

struct QCol {            // per-lane constants of one query column
  float iq, tf;          // 1/||q||, threshold on s~ (+-inf = everything / nothing passes)
  float lo, hi;          // block bounds on t = cos*||q||: nearest: hit iff max t >= lo;  farthest: hit iff min t <= lo or max t >= hi
  int qidx;
};

__device__ __forceinline__ QCol mf_query_col(int qidx, int nq, const float* __restrict__ qnorms, const uint32_t* __restrict__ thr, int nearest) {
  QCol c; c.qidx = qidx;
  const bool live = qidx < nq;
  const float nrm = live ? qnorms[qidx] : 1.f;
  c.iq = live ? rsqrtf(nrm) : 0.f;
  const float len = sqrtf(nrm);            // ||q||; NaN / 0 / inf propagate into lo/hi and make every block take the element path
  const uint32_t t = live ? thr[qidx] : (nearest ? 0u : 0xffffffffu);
  if (nearest) {
    c.tf = !live ? -1.f : (t == 0xffffffffu ? __builtin_inff() : key_score(t));
    c.lo = !live ? __builtin_inff() : (1.0f - c.tf - MF_BLOCK_SLACK) * len;
    c.hi = __builtin_inff();
  } else {
    c.tf = !live ? __builtin_inff() : (t == 0u ? -__builtin_inff() : key_score(t));
    c.lo = !live ? -__builtin_inff() : (1.0f - c.tf + MF_BLOCK_SLACK) * len;
    c.hi = !live ? __builtin_inff() : (1.0f + c.tf - MF_BLOCK_SLACK) * len;
  }
  return c;
}

// ir[g][j] = 1/||row|| of local row 8*g + j (+ 4 for the upper half-wave); rbase = global index of the lane's local row 0;
// ep = this lane's private 16-float LDS slot.  The element path is a ROLLED loop over values parked in LDS: unrolled it is
// ~500 instructions per block, 16 blocks per tile — the epilogue then no longer fits the instruction cache and every tile
// pays the misses even though the path is almost never taken.
__device__ __forceinline__ void mf_emit_block(const f32x16& acc, const f32x4 (&ir)[4], bool bad, const QCol& qc, int nearest, int nq,
                                              uint64_t rbase, uint64_t end, unsigned long long* __restrict__ cand,
                                              uint32_t* __restrict__ cnt, uint32_t cap, float* ep) {
  float t[16];
#pragma unroll
  for (int r = 0; r < 16; r++) t[r] = acc[r] * ir[r >> 2][r & 3];
  float mx = t[0], mn = t[0];
#pragma unroll
  for (int r = 1; r < 16; r++) mx = __builtin_fmaxf(mx, t[r]);
  bool hit;
  if (nearest) hit = !(mx < qc.lo);
  else {
#pragma unroll
    for (int r = 1; r < 16; r++) mn = __builtin_fminf(mn, t[r]);
    hit = !(mn > qc.lo) || !(mx < qc.hi);
  }
  if (!(hit || bad)) return;
#pragma unroll
  for (int g = 0; g < 4; g++) reinterpret_cast<f32x4*>(ep)[g] = f32x4{t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]};
#pragma unroll 1
  for (int r = 0; r < 16; r++) {
    float s = fabsf(1.0f - reinterpret_cast<volatile float*>(ep)[r] * qc.iq);
    const bool pass = nearest ? !(s > qc.tf) : !(s < qc.tf);
    if (pass) {
      const uint64_t gr = rbase + (r & 3) + 8 * (r >> 2);
      if (gr < end && qc.qidx < nq) {
        uint32_t idx = atomicAdd(&cnt[qc.qidx], 1u);
        if (idx < cap) cand[(size_t)qc.qidx * cap + idx] = ((unsigned long long)score_key(s) << 32) | (uint32_t)gr;
      }
    }
  }
}

__device__ __forceinline__ bool mf_bad_norms(const f32x4 (&ir)[4]) {
  bool bad = false;
#pragma unroll
  for (int g = 0; g < 4; g++)
#pragma unroll
    for (int j = 0; j < 4; j++) bad |= !(ir[g][j] > 0.f && ir[g][j] < __builtin_inff());
  return bad;
}

// Per query: the k-th best approximate key, keep every candidate within MF_MARGIN of it (compacted into `dst`), publish the
// bound as the next scan threshold.  One block of 256 threads per query.
__global__ __launch_bounds__(256) void flat_pick_kernel(const unsigned long long* __restrict__ src_all, unsigned long long* __restrict__ dst_all,
                                                       uint32_t* __restrict__ cnt_all, uint32_t* __restrict__ thr_all, uint32_t cap,
                                                       uint32_t k, int nearest, float margin, uint32_t* __restrict__ overflow,
                                                       const float* __restrict__ qn_l2 = nullptr, float max_row_norm = 0.f) {
  __shared__ uint32_t hist[256], wsum[4];
  __shared__ uint32_t s_digit, s_need, s_n;
  const int q = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* src = src_all + (size_t)q * cap;
  unsigned long long* dst = dst_all + (size_t)q * cap;
  uint32_t c = cnt_all[q];
  if (c > cap) { if (tid == 0) atomicOr(overflow, 1u); c = cap; }
  const uint32_t flip = nearest ? 0u : 0xffffffffu;
  uint32_t bound_key = 0xffffffffu;  // in key' space: keep key' <= bound_key
  // Lists of up to 4096 entries — every list but a pathological one: the seed segment is 1-4 Ki rows, later segments leave a few hundred —
  // are read from global memory ONCE into registers; the four digit passes and the compaction then run out of them (each used to be its
  // own trip through L2: 18.8 us for the 4 Ki-entry pick after the seed, profiles/r04w_f3_chain_timeline.txt).
  constexpr int PK_R = 16;
  const bool in_regs = c <= 256u * PK_R;
  unsigned long long ev[PK_R];
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < PK_R; u++) { const uint32_t i = (uint32_t)tid + 256u * u; ev[u] = 0ull; if (256u * u < c) ev[u] = i < c ? src[i] : 0ull; }   // (block-uniform guard: a short list costs its own length)
  }
  if (c > k) {
    uint32_t prefix = 0, mask = 0, need = k;
    for (int pass = 3; pass >= 0; pass--) {
      const int shift = pass * 8;
      hist[tid] = 0;
      __syncthreads();
      if (in_regs) {
#pragma unroll
        for (int u = 0; u < PK_R; u++) {
          if (256u * u >= c) break;
          const uint32_t kp = (uint32_t)(ev[u] >> 32) ^ flip;
          if ((uint32_t)tid + 256u * u < c && (kp & mask) == prefix) atomicAdd(&hist[(kp >> shift) & 255u], 1u);
        }
      } else {
        for (uint32_t i = tid; i < c; i += 256) {
          uint32_t kp = (uint32_t)(src[i] >> 32) ^ flip;
          if ((kp & mask) == prefix) atomicAdd(&hist[(kp >> shift) & 255u], 1u);
        }
      }
      __syncthreads();
      {  // the digit whose bucket holds the need-th key: inclusive prefix sum of the 256 buckets (a serial walk by one
         // thread is 256 dependent LDS reads = ~10 us per pass, 40 us per pick: more than the later segments' scans)
        const uint32_t h = hist[tid];
        uint32_t inc = h;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d); if ((tid & 63) >= d) inc += o; }
        if ((tid & 63) == 63) wsum[tid >> 6] = inc;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < (tid >> 6); w++) base += wsum[w];
        inc += base;
        if (inc >= need && inc - h < need) { s_digit = tid; s_need = need - (inc - h); }  // exactly one thread (need >= 1, total >= need)
      }
      __syncthreads();
      prefix |= s_digit << shift; mask |= 255u << shift; need = s_need;
      __syncthreads();
    }
    // k-th best approximate score, widened by the margin in the "worse" direction
    float t = key_score(prefix ^ flip);
    // Euclidean candidates carry s~^2 = ||q||^2 + ||r||^2 - 2 dot~; its error scales with the norms: margin * (||q||^2 + max ||r||^2)
    if (qn_l2) margin = qn_l2[q] <= 4.0e9f ? margin * (qn_l2[q] + max_row_norm) : __builtin_inff();   // out-of-range query: keep everything (-> exact fallback)
    float b = nearest ? t + margin : t - margin;
    bound_key = score_key(b) ^ flip;
    if (bound_key < prefix) bound_key = prefix;  // NaN / saturation guard
  }
  if (tid == 0) s_n = 0;
  __syncthreads();
  if (in_regs) {
#pragma unroll
    for (int u = 0; u < PK_R; u++) {
      if (256u * u >= c) break;
      if ((uint32_t)tid + 256u * u < c && ((uint32_t)(ev[u] >> 32) ^ flip) <= bound_key) dst[atomicAdd(&s_n, 1u)] = ev[u];
    }
  } else {
    for (uint32_t i = tid; i < c; i += 256) {
      unsigned long long e = src[i];
      if (((uint32_t)(e >> 32) ^ flip) <= bound_key) dst[atomicAdd(&s_n, 1u)] = e;
    }
  }
  __syncthreads();
  if (tid == 0) { cnt_all[q] = s_n; thr_all[q] = c > k ? (bound_key ^ flip) : (nearest ? 0xffffffffu : 0u); }
}

// Exact re-score of the survivors: lane pair per (query, candidate); the key's score half is replaced by the exact score.
// Grid-stride over the query's list, so the launch needs no host-side knowledge of the list lengths (no mid-chain sync).
template <int METRIC, int QUANT>
__global__ __launch_bounds__(64) void flat_rescore_kernel(const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms,
                                                         const float* __restrict__ q_eff, const float* __restrict__ qnorms, int dim,
                                                         unsigned long long* __restrict__ cand_all, const uint32_t* __restrict__ cnt_all,
                                                         uint32_t cap) {
  const int q = blockIdx.y;
  const int lane = threadIdx.x, half = lane & 1;
  const uint32_t c = cnt_all[q] < cap ? cnt_all[q] : cap;
  unsigned long long* cand = cand_all + (size_t)q * cap;
  for (uint32_t j0 = blockIdx.x * 32; j0 < c; j0 += gridDim.x * 32) {
    const uint32_t j = j0 + (lane >> 1);
    const bool valid = j < c;
    const uint32_t slot = (uint32_t)cand[valid ? j : 0];
    float d = pair_distance<METRIC, QUANT, 4>(rows + (size_t)slot * stride, q_eff + (size_t)q * dim, dim, qnorms[q], norms[slot], half);
    if (valid && half == 0) cand[j] = ((unsigned long long)score_key(d) << 32) | slot;
  }
}

}  // namespace dev
}  // namespace coltt
