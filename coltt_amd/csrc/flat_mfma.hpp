// flat_mfma.hpp — the batched FLAT scan through the matrix cores (kernel K2 of SURVEY.md §2): candidate generation by an f16 GEMM with a fused
// threshold test, pick, exact re-score.  ONE header since round 5 (it used to be a three-deep include chain flat_mfma.hpp -> flat_mfma2.hpp ->
// flat_mfma3.hpp, one file per kernel generation; only the third generation's kernel ships, and the earlier files had long been reduced to the
// layers it is built from).  Three sections, in dependency order:
//   [1] contract, margins, query preparation, the per-column constants, pick and exact re-score           (round 1)
//   [2] tile geometry, the LDS-DMA primitives, counted vmcnt waits, the one- and two-pass epilogues        (round 2)
//   [3] flat_mfma3_kernel: split row / query rings with specialised loader waves, gather mode             (round 3-4)
// (The superseded kernels were kept under tools/experiments/ for A/B builds until round 5; they are in the history only.)
//
// ---- [1] ----------------------------------------------------------------------------------------------------------------------------------
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


// ---- [2] the candidate GEMM as a persistent LDS-DMA pipeline ----------------------------------------------------------------------------------
// (second generation of the round-1 kernel; same contract: approximate scores only pick candidates, survivors are re-scored exactly)
//
// What bound the first kernel (PMC, 10 M x 768 f16, batch 256): bytes entering a CU — the 256 x D query matrix was re-read for
// every 128-row tile (3.0x the algorithmic bytes at the L2 -> L1 level, TCP_PENDING_STALL 75 % of busy), every byte was staged
// through VGPRs and written to LDS with ds_write_b128 (~79 B/clk/CU: as many LDS-pipe cycles as the fragment reads), and a
// two-step register prefetch is all the latency cover an in-order wave gets.  Here:
//   * tile = 256 rows x BN queries per workgroup of 8 waves (4 x 2, 64 x BN/2 per wave, 2 waves per SIMD): the query matrix
//     crosses L2 -> CU once per 256 rows, half as often;
//   * rows AND queries go global -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction): no staging registers,
//     no ds_write, no VALU in the feed path;
//   * a ring of NS stages (K step = 32 halves: 64 B of every row) runs NS-1 stages ahead ACROSS tile boundaries — the next
//     tile's first stages are in flight while the current tile's epilogue runs — with counted s_waitcnt vmcnt and ONE
//     s_barrier per K step;
//   * the DMA image is lane-linear (lane l writes 16 B at base + 16 l), so each lane FETCHES the chunk that belongs at its
//     position under an XOR swizzle (64-B rows: pos = chunk ^ ((row >> 2) & 3); 128-B f32 rows: pos = chunk ^ ((row >> 1) & 7)),
//     which makes every MFMA-fragment ds_read_b128 hit 16 distinct 16-byte bank slots per lane group (conflict-free).
// The DMA instructions are inline asm on purpose: hipcc's waitcnt pass treats __builtin_amdgcn_global_load_lds as aliasing
// every later ds_read and drains vmcnt to 0 in front of each barrier, which would serialise the ring.

constexpr int M2_BM = 256;   // rows per tile
constexpr int M2_BK = 32;    // halves per K step
constexpr int M2_NT = 512;   // 8 waves
constexpr int M2_TNORM = 320;  // floats per tile-parity buffer of raw ||row||^2 (256 + the 32-float overlap of the last wave + pad)

template <int BN, bool AF32> struct M2Geom {
  static constexpr int A_ROWB = M2_BK * (AF32 ? 4 : 2);            // bytes of one row per stage (64 | 128)
  static constexpr int A_STAGE = M2_BM * A_ROWB;                   // 16 KiB | 32 KiB
  static constexpr int B_STAGE = BN * M2_BK * 2;                   // 64 B per query
  static constexpr int STAGE = A_STAGE + B_STAGE;
  static constexpr int FIXED = 2 * M2_TNORM * 4 + M2_NT * 32;      // raw norms x2 + 8-float epilogue scratch per lane
  static constexpr int NS = (4 * STAGE + FIXED <= 160 * 1024) ? 4 : ((3 * STAGE + FIXED <= 160 * 1024) ? 3 : 2);
  static constexpr int NA_I = A_STAGE / 1024 / 8;                  // DMA instructions per wave per stage: rows
  static constexpr int NB_I = (B_STAGE / 1024 + 7) / 8;            //                                      queries (BN = 64: waves 4-7 duplicate)
  static constexpr int PER = NA_I + NB_I + 1;                      // + the raw-norm refresh
  static constexpr size_t LDS = (size_t)NS * STAGE + FIXED;
};

// lds_base is wave-uniform by construction; readfirstlane makes that explicit for the "s" constraint (hipcc otherwise hands
// the asm a VGPR whenever its uniformity analysis gives up, e.g. on the tile-parity flag)
template <bool NT> __device__ __forceinline__ void m2_dma16(const void* g, uint32_t lds_base) {
  lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
  if constexpr (NT) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(g), "s"(lds_base) : "memory");
  else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(lds_base) : "memory");
}
#ifndef COLTT_M2_NT
#define COLTT_M2_NT 0
#endif
constexpr bool M2_A_NT = COLTT_M2_NT != 0;   // streamed-once rows: non-temporal hint on the row DMA (measurement knob)
#ifndef COLTT_M2_ISSUE
#define COLTT_M2_ISSUE 0
#endif
// where a wave issues its DMA pieces inside a K step: 0 all right after the barrier; 1 waves 4-7 (the SIMD partners of
// waves 0-3) issue theirs between the two MFMA groups instead; 2 every wave spreads its pieces between MFMA groups
constexpr int M2_ISSUE = COLTT_M2_ISSUE;
__device__ __forceinline__ void m2_dma4(const void* g, uint32_t lds_base) {
  lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(g), "s"(lds_base) : "memory");
}
// s_waitcnt vmcnt(N) only (gfx9 encoding: vmcnt = imm[15:14]:imm[3:0], expcnt imm[6:4], lgkmcnt imm[11:8])
template <int N> __device__ __forceinline__ void m2_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// block filter + (rare) element path: as mf_emit_block, but (a) the parked values go through an 8-float private LDS slot in
// two halves and each half costs ONE atomicAdd (count first, reserve, then store) instead of one dependent atomic round trip
// per passing element, and (b) the unfiltered SEED segment takes no atomics at all: every score passes, so entry (row - begin)
// of the query's list is simply written in place and the count is the segment length.
// Euclidean columns (METRIC == M_L2): the candidate value is s~^2 = ||q||^2 + ||r||^2 - 2 dot (monotone in the distance; the
// exact re-score produces the reference's sqrt form).  QCol then means: iq = ||q||^2, tf = threshold on s~^2, lo / hi = bounds on
// t = ||r||^2 - 2 dot (nearest: hit iff min t <= lo; farthest: hit iff max t >= hi), and `ir` carries the RAW row norms.
__device__ __forceinline__ QCol m2_query_col_l2(int qidx, int nq, const float* __restrict__ qnorms, const uint32_t* __restrict__ thr, int nearest) {
  QCol c; c.qidx = qidx;
  const bool live = qidx < nq;
  const float nqv = live ? qnorms[qidx] : 0.f;
  c.iq = nqv;
  const uint32_t t = live ? thr[qidx] : (nearest ? 0u : 0xffffffffu);
  if (nearest) {
    c.tf = !live ? -__builtin_inff() : (t == 0xffffffffu ? __builtin_inff() : key_score(t));
    c.lo = !live ? -__builtin_inff() : (c.tf - nqv) + MF_BLOCK_SLACK * (fabsf(c.tf) + nqv);   // NaN norms: every compare fails -> `hit`
    c.hi = __builtin_inff();
  } else {
    c.tf = !live ? __builtin_inff() : (t == 0u ? -__builtin_inff() : key_score(t));
    c.lo = -__builtin_inff();
    c.hi = !live ? __builtin_inff() : (c.tf - nqv) - MF_BLOCK_SLACK * (fabsf(c.tf) + nqv);
  }
  return c;
}

template <bool SEED, int METRIC = M_COS>
__device__ __forceinline__ void m2_emit_block(const f32x16& acc, const f32x4 (&ir)[4], bool bad, const QCol& qc, int nearest, int nq,
                                              uint64_t rbase, uint64_t begin, uint64_t end, unsigned long long* __restrict__ cand,
                                              uint32_t* __restrict__ cnt, uint32_t cap, float* ep,
                                              const uint32_t* __restrict__ gather = nullptr) {  // gather: row numbers are positions of a slot list
  float t[16];
  // (v_pk_mul_f32 / v_pk_add_f32 on pairs of elements — half the VALU issue slots, same IEEE results — was measured in round 4 and
  //  changes nothing: C3 4.975 -> 4.953 ms, C2 0.645 -> 0.648 ms, profiles/r04_flat_c3_pkepi_ab.txt; the epilogue is not what binds.)
#pragma unroll
  for (int r = 0; r < 16; r++) t[r] = METRIC == M_COS ? acc[r] * ir[r >> 2][r & 3] : ir[r >> 2][r & 3] - 2.0f * acc[r];
  // the approximate candidate value of element r from its parked t
  auto value = [&](float tv) { return METRIC == M_COS ? fabsf(1.0f - tv * qc.iq) : qc.iq + tv; };
  if constexpr (SEED) {
    if (qc.qidx < nq) {
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const uint64_t gr = rbase + (r & 3) + 8 * (r >> 2);
        const float s = value(t[r]);
        if (gr < end) cand[(size_t)qc.qidx * cap + (uint32_t)(gr - begin)] = ((unsigned long long)score_key(s) << 32) | (gather ? gather[gr] : (uint32_t)gr);
      }
      if (rbase == begin) cnt[qc.qidx] = (uint32_t)(end - begin);
    }
    return;
  }
  float mx = t[0], mn = t[0];
#pragma unroll
  for (int r = 1; r < 16; r++) mx = __builtin_fmaxf(mx, t[r]);
  bool hit;
  if constexpr (METRIC == M_COS) {
    if (nearest) hit = !(mx < qc.lo);
    else {
#pragma unroll
      for (int r = 1; r < 16; r++) mn = __builtin_fminf(mn, t[r]);
      hit = !(mn > qc.lo) || !(mx < qc.hi);
    }
  } else {
    if (nearest) {
#pragma unroll
      for (int r = 1; r < 16; r++) mn = __builtin_fminf(mn, t[r]);
      hit = !(mn > qc.lo);
    } else hit = !(mx < qc.hi);
  }
  if (!(hit || bad)) return;
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    reinterpret_cast<f32x4*>(ep)[0] = h ? f32x4{t[8], t[9], t[10], t[11]} : f32x4{t[0], t[1], t[2], t[3]};
    reinterpret_cast<f32x4*>(ep)[1] = h ? f32x4{t[12], t[13], t[14], t[15]} : f32x4{t[4], t[5], t[6], t[7]};
    uint32_t mask = 0;
#pragma unroll 1
    for (int r8 = 0; r8 < 8; r8++) {
      const int r = h * 8 + r8;
      const float s = value(reinterpret_cast<volatile float*>(ep)[r8]);
      const bool pass = nearest ? !(s > qc.tf) : !(s < qc.tf);
      const uint64_t gr = rbase + (r & 3) + 8 * (r >> 2);
      if (pass && gr < end && qc.qidx < nq) mask |= 1u << r8;
    }
    if (mask) {
      uint32_t idx = atomicAdd(&cnt[qc.qidx], (uint32_t)__builtin_popcount(mask));
#pragma unroll 1
      for (int r8 = 0; r8 < 8; r8++) {
        if (!((mask >> r8) & 1u)) continue;
        const int r = h * 8 + r8;
        const float s = value(reinterpret_cast<volatile float*>(ep)[r8]);
        const uint64_t gr = rbase + (r & 3) + 8 * (r >> 2);
        if (idx < cap) cand[(size_t)qc.qidx * cap + idx] = ((unsigned long long)score_key(s) << 32) | (gather ? gather[gr] : (uint32_t)gr);
        idx++;
      }
    }
  }
}

// ---- two-pass form of the same epilogue (flat_mfma.hpp [3]) ---------------------------------------------------------------------------
// m2_emit_block pays one global atomic ROUND TRIP per half block that holds a survivor, inside the block loop: a wave with h such half
// blocks in a tile stalls h x ~2 us, and its workgroup waits for it at the next tile's first barrier.  Behind a loose threshold that is
// what a segment costs (1 M x 768 f32, batch 64, a 4 Ki-row seed followed by everything else: +0.24 ms for 156 k survivors,
// profiles/r04n_segment_schedule.md).  Here the tile's blocks are only TESTED first (a 16-bit survivor mask per block, in registers), the
// lane then reserves its survivors of a whole query column with ONE atomic per column — all columns' atomics in flight together — and a
// second pass over the blocks that had survivors writes them: one round trip per tile and wave.
template <int METRIC>
__device__ __forceinline__ uint32_t m2_test_block(const f32x16& acc, const f32x4 (&ir)[4], bool bad, const QCol& qc, int nearest, int nq,
                                                  uint64_t rbase, uint64_t end) {
  float t[16];
#pragma unroll
  for (int r = 0; r < 16; r++) t[r] = METRIC == M_COS ? acc[r] * ir[r >> 2][r & 3] : ir[r >> 2][r & 3] - 2.0f * acc[r];
  auto value = [&](float tv) { return METRIC == M_COS ? fabsf(1.0f - tv * qc.iq) : qc.iq + tv; };
  float mx = t[0], mn = t[0];
#pragma unroll
  for (int r = 1; r < 16; r++) mx = __builtin_fmaxf(mx, t[r]);
  bool hit;
  if constexpr (METRIC == M_COS) {
    if (nearest) hit = !(mx < qc.lo);
    else {
#pragma unroll
      for (int r = 1; r < 16; r++) mn = __builtin_fminf(mn, t[r]);
      hit = !(mn > qc.lo) || !(mx < qc.hi);
    }
  } else {
    if (nearest) {
#pragma unroll
      for (int r = 1; r < 16; r++) mn = __builtin_fminf(mn, t[r]);
      hit = !(mn > qc.lo);
    } else hit = !(mx < qc.hi);
  }
  if (!(hit || bad) || qc.qidx >= nq) return 0u;
  // The element test, UNROLLED in registers: no atomics and no stores here, ~6 VALU per element — the rolled loop over LDS-parked values
  // that m2_emit_block needs for its instruction-cache footprint cost ~100 dependent cycles per element, which is what a segment behind a
  // loose threshold (92 % of its blocks hit) was paying.
  const uint64_t left64 = end > rbase ? end - rbase : 0;
  const int left = left64 > 64 ? 64 : (int)left64;          // local rows 0 .. 27 of this lane's 16 exist up to `end`
  // (at batch 256 the unrolled test made the gathered-f32 instances spill and the power-bound C3 shape slow from 4.82 to 5.45 ms per batch,
  //  a rolled one still cost it 2.4 %: those instances stay on m2_emit_block, profiles/r04p_epilogue_ab.md)
  uint32_t mask = 0;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const float s = value(t[r]);
    const bool pass = nearest ? !(s > qc.tf) : !(s < qc.tf);
    mask |= (pass && ((r & 3) + 8 * (r >> 2)) < left) ? (1u << r) : 0u;
  }
  return mask;
}
// second pass: the survivors of one block (mask from m2_test_block, same acc / ir / qc) into cand[qidx][idx ...]; returns the next free index
template <int METRIC>
__device__ __forceinline__ uint32_t m2_store_block(const f32x16& acc, const f32x4 (&ir)[4], uint32_t mask, const QCol& qc, uint64_t rbase,
                                                   unsigned long long* __restrict__ cand, uint32_t idx, uint32_t cap, float* ep,
                                                   const uint32_t* __restrict__ gather) {
  float t[16];
#pragma unroll
  for (int r = 0; r < 16; r++) t[r] = METRIC == M_COS ? acc[r] * ir[r >> 2][r & 3] : ir[r >> 2][r & 3] - 2.0f * acc[r];
  auto value = [&](float tv) { return METRIC == M_COS ? fabsf(1.0f - tv * qc.iq) : qc.iq + tv; };
#pragma unroll 1
  for (int h = 0; h < 2; h++) {
    if (!((mask >> (8 * h)) & 0xffu)) continue;
    reinterpret_cast<f32x4*>(ep)[0] = h ? f32x4{t[8], t[9], t[10], t[11]} : f32x4{t[0], t[1], t[2], t[3]};
    reinterpret_cast<f32x4*>(ep)[1] = h ? f32x4{t[12], t[13], t[14], t[15]} : f32x4{t[4], t[5], t[6], t[7]};
#pragma unroll 1
    for (int r8 = 0; r8 < 8; r8++) {
      const int r = h * 8 + r8;
      if (!((mask >> r) & 1u)) continue;
      const float s = value(reinterpret_cast<volatile float*>(ep)[r8]);
      const uint64_t gr = rbase + (r & 3) + 8 * (r >> 2);
      if (idx < cap) cand[(size_t)qc.qidx * cap + idx] = ((unsigned long long)score_key(s) << 32) | (gather ? gather[gr] : (uint32_t)gr);
      idx++;
    }
  }
  return idx;
}


// ---- [3] split rings, specialised loader waves -----------------------------------------------------------------------------------------------
// (third generation: the LDS-DMA ring of section [2] with SPLIT rings and SPECIALISED loader waves)
//
// What the second generation showed (10 M x 768 f16, batch 256, one MI355X; ablations in profiles/r02_flat_mfma_ablation.txt):
// ds_read + MFMA alone 2.84 ms, DMA alone 3.38 ms, both together 4.98 ms.  The DMA stream sustains ~17 B/clk/CU with three
// 32 KB stages in flight per CU, i.e. a loaded memory latency of ~5 800 clk; the ring holds just enough bytes for that, so any
// delay in issuing (the compute phase) shows up one-for-one.  Half of the ring was spent on the QUERY tile, which is L2-resident
// and needs no such cover — but loads complete in issue order per wave, so one wave cannot run a deep prefetch for rows and a
// shallow one for queries at the same time.  Hence:
//   * waves 0-3 issue the ROW DMA only (ring of NSA slots, NSA-1 stages ahead: HBM latency), waves 4-7 the QUERY DMA and the
//     raw-norm refresh (ring of NSB slots, NSB-1 ahead: L2 latency); each wave's vmcnt queue is homogeneous, so both depths
//     are real.  f16 rows, batch 256: rows 5 x 16 KB, queries 3 x 16 KB = the same 128 KB, with 64 KB of rows in flight
//     instead of 48 KB;
//   * the row loaders issue right after the barrier, the query loaders between the two MFMA groups of the step: waves w and
//     w + 4 share a SIMD (dispatch order 0, 2, 1, 3), so a SIMD never has both of its waves in DMA issue at once;
//   * everything else (256 x BN tile, 8 waves as 4 x 2, XOR-swizzled lane-linear DMA image, one s_barrier per K step, epilogue
//     with the in-place seed segment and one atomic per half block) is flat_mfma.hpp [2]'s.

// saddr-form DMA: address = sbase (SGPR pair) + voffset (32-bit VGPR)
template <bool NT> __device__ __forceinline__ void m3_dma16s(uint32_t voffset, const void* sbase, uint32_t lds_base) {
  lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
  const uint64_t sb = (uint64_t)(uintptr_t)sbase;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32));
  const uint64_t sbu = ((uint64_t)hi << 32) | lo;
  if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(voffset), "s"(sbu), "s"(lds_base) : "memory");
  else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voffset), "s"(sbu), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void m3_dma4s(uint32_t voffset, const void* sbase, uint32_t lds_base) {
  lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
  const uint64_t sb = (uint64_t)(uintptr_t)sbase;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32));
  const uint64_t sbu = ((uint64_t)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voffset), "s"(sbu), "s"(lds_base) : "memory");
}

// GATHER: the rows of a tile are rows[gather[pos]] for consecutive positions pos (FilterableVertexSearch, edge/none_vectorstore.go:182-253):
// every wave keeps the slot numbers of its 64 tile rows for three tiles in LDS (IDS bytes, see flat_mfma3_kernel).
template <int BN, bool AF32, int BM = M2_BM, bool GATHER = false> struct M3Geom {
  static constexpr int A_ROWB = M2_BK * (AF32 ? 4 : 2);
  static constexpr int A_STAGE = BM * A_ROWB;                      // 16 KiB | 32 KiB (24 | 48 at BM = 384)
  static constexpr int B_STAGE = BN * M2_BK * 2;                   // 4 / 8 / 16 KiB
  static constexpr int TNORM = BM + 64;                            // floats per tile-parity buffer of raw ||row||^2
  static constexpr int NN_I = BM / 256 + (BM % 256 ? 1 : 0);       // raw-norm DMA instructions per query-loader wave per stage
  static constexpr int IDS = GATHER ? 8 * 3 * 64 * 4 : 0;          // [8 waves][3 tiles][64 rows] u32
  static constexpr int FIXED = 2 * TNORM * 4 + M2_NT * 32 + IDS;
  static constexpr bool TUNED = !AF32 && BN == 256;              // measurement overrides apply to the batch-256 f16 shape only
#ifdef COLTT_M3_NSB
  static constexpr int NSB = TUNED ? COLTT_M3_NSB : 3;
#else
  static constexpr int NSB = BM > 256 ? 2 : 3;
#endif
  static constexpr int NSA_FIT = (160 * 1024 - FIXED - NSB * B_STAGE) / A_STAGE;
#ifdef COLTT_M3_NSA
  static constexpr int NSA = TUNED ? COLTT_M3_NSA : (NSA_FIT > 8 ? 8 : NSA_FIT);
#else
  static constexpr int NSA = NSA_FIT > 8 ? 8 : NSA_FIT;
#endif
  static constexpr int NA_I = A_STAGE / 1024 / 4;                  // row DMA instructions per loader wave per stage (4 | 8)
  static constexpr int NB_I = B_STAGE / 1024 / 4;                  // query DMA instructions per loader wave per stage (1 | 2 | 4)
  static constexpr int A_BYTES = NSA * A_STAGE, B_BYTES = NSB * B_STAGE;
  static constexpr size_t LDS = (size_t)A_BYTES + B_BYTES + FIXED;
  static_assert(NSA >= 2 && NSB >= 2 && LDS <= 160 * 1024, "ring does not fit");
  static_assert(!GATHER || BM == 256, "gather mode: 64 tile rows per loader wave");
  static_assert((NSA - 2) * NA_I < 64 && (NSB - 2) * (NB_I + NN_I) < 64, "vmcnt range");
};

template <int BN, bool AF32, bool SEED, int BM = M2_BM, int METRIC = M_COS, bool GATHER = false>
__global__ __launch_bounds__(M2_NT, 2) void flat_mfma3_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, uint64_t begin, uint64_t end,
    const _Float16* __restrict__ q16, const float* __restrict__ qnorms, int nq, int dim, const uint32_t* __restrict__ thr,
    int nearest, unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap,
    const uint32_t* __restrict__ gather = nullptr) {
  typedef M3Geom<BN, AF32, BM, GATHER> G;
  constexpr int WN = 2;
  constexpr int WROWS = BM / 4;            // rows per wave row (4 x 2 wave grid)
  constexpr int TM = WROWS / 32, TN = BN / WN / 32;
  constexpr int NSA = G::NSA, NSB = G::NSB;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const bool row_loader = wave < 4;
  const int lw = wave & 3;  // index among the loaders of my kind
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  uint8_t* const ringA = smem;                                                   // [NSA][256 rows][A_ROWB]
  uint8_t* const ringB = smem + G::A_BYTES;                                      // [NSB][BN queries][64]
  float* const tnorm = reinterpret_cast<float*>(smem + G::A_BYTES + G::B_BYTES); // [2][M2_TNORM] raw ||row||^2
  float* const ep = reinterpret_cast<float*>(smem + G::A_BYTES + G::B_BYTES + 2 * G::TNORM * 4) + tid * 8;
  // GATHER: slot numbers of this wave's 64 tile rows (tile rows lw*64 .. lw*64+63: the rows a row loader fetches, the norms a query
  // loader refreshes), for three consecutive tiles of this workgroup
  uint32_t* const idbuf = reinterpret_cast<uint32_t*>(smem + G::A_BYTES + G::B_BYTES + 2 * G::TNORM * 4 + M2_NT * 32) + wave * (3 * 64);
  const int nk = dim / M2_BK;
  const uint64_t ntiles = (end - begin + BM - 1) / BM;
  if ((uint64_t)blockIdx.x >= ntiles) return;
  QCol qc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++)
    qc[tn] = METRIC == M_COS ? mf_query_col(wn * (BN / WN) + tn * 32 + (lane & 31), nq, qnorms, thr, nearest)
                             : m2_query_col_l2(wn * (BN / WN) + tn * 32 + (lane & 31), nq, qnorms, thr, nearest);

  // ---- loader state ---------------------------------------------------------------------------------------------------------
  // Every DMA is `global_load_lds_* voffset, sbase`: the per-lane part of the address is ONE 32-bit register per stream that
  // never changes (row-in-instruction x stride + swizzled chunk), everything that moves (tile, K step, instruction index) is
  // scalar arithmetic.  Rows past `end` in the last tile are fetched without clamping: the store keeps ROW_SLACK rows of
  // slack behind its capacity (flat.hip), their scores are dropped by the `row < end` test of the epilogue.
  constexpr int A_CPR = G::A_ROWB / 16, A_RPI = 64 / A_CPR;
  // (f32 rows: 8 rows per instruction, so bit 3 of the tile row — bit 2 of the swizzle — alternates with the instruction
  // index: odd instructions use voff_odd.  f16 rows: 16 rows per instruction, the swizzle bits never see the index.)
  uint32_t voff, voff_odd = 0;   // row loaders: rows stream; query loaders: queries stream
  if (row_loader) {
    const int lr = lane / A_CPR, p = lane % A_CPR;
    voff = (uint32_t)(lr * stride) + (uint32_t)((AF32 ? (p ^ ((lr >> 1) & 7)) : (p ^ ((lr >> 2) & 3))) * 16);
    if constexpr (AF32) voff_odd = (uint32_t)(lr * stride) + (uint32_t)((p ^ (((lr >> 1) & 7) | 4)) * 16);
  } else {
    const int q = lane / 4, p = lane % 4;
    voff = (uint32_t)(q * dim * 2) + (uint32_t)((p ^ ((q >> 2) & 3)) * 16);
  }
  const uint32_t nvoff = (uint32_t)lane * 4;
  uint64_t ld_tile = blockIdx.x; int ld_ks = 0; uint32_t ld_g = 0, ld_par = 0;
  const uint64_t last_tile = blockIdx.x + ((ntiles - 1 - blockIdx.x) / gridDim.x) * gridDim.x;
  // ---- GATHER: per-lane 64-bit source addresses.  Tile n (n-th tile of this workgroup) has its slot numbers in idbuf[n % 3]:
  // tiles 0-2 are fetched synchronously below (the prologue may already cross into them), tile n + 2 is requested — an LDS-DMA, no
  // register result that would have to be waited for — when the loader switches to tile n, i.e. >= 2 nk stages before it is read:
  // every wave waits for all but its last few DMAs at every K step, so the request has long landed by then.
  uint32_t ld_n = 0;                 // index of the loader's tile among this workgroup's tiles
  const uint8_t* gbase[GATHER ? G::NA_I : 1];
  const float* gnorm = nullptr;
  auto tile_of = [&](uint32_t n) { const uint64_t t = blockIdx.x + (uint64_t)n * gridDim.x; return t < ntiles ? t : last_tile; };
  auto ids_pos = [&](uint32_t n) {   // position (in the gather list) of tile row lw*64 + lane of this workgroup's n-th tile, clamped
    const uint64_t pos = begin + tile_of(n) * BM + (uint64_t)(lw * 64 + lane);
    return pos < end ? pos : end - 1;
  };
  auto take_ids = [&](uint32_t n) {  // addresses of the loader's rows / norms for tile n from idbuf (LDS)
    const uint32_t* ib = idbuf + (n % 3) * 64;
    if (row_loader) {
      const int lr = lane / A_CPR, p = lane % A_CPR;
#pragma unroll
      for (int i = 0; i < G::NA_I; i++) {
        const uint32_t slot = ib[i * A_RPI + lr];
        const uint32_t chunk = AF32 ? (uint32_t)((p ^ (((lr >> 1) & 7) | ((i & 1) ? 4 : 0))) * 16) : (uint32_t)((p ^ ((lr >> 2) & 3)) * 16);
        gbase[i] = rows + (size_t)slot * stride + chunk;
      }
    } else gnorm = norms + ib[lane];
  };
  if constexpr (GATHER) {
#pragma unroll
    for (uint32_t n = 0; n < 3; n++) idbuf[n * 64 + lane] = gather[ids_pos(n)];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    take_ids(0);
  }
  auto issue_stage = [&]() {   // my kind's share of the stage the loader points at, then advance
#ifndef COLTT_M2_NO_DMA
    const uint64_t row0 = begin + ld_tile * BM;
    if (row_loader) {
      const uint32_t slot = lds0 + (ld_g % NSA) * G::A_STAGE + (uint32_t)(lw * G::NA_I * 1024);
      if constexpr (GATHER) {
#pragma unroll
        for (int i = 0; i < G::NA_I; i++) m2_dma16<M2_A_NT || AF32>(gbase[i] + (size_t)ld_ks * G::A_ROWB, slot + (uint32_t)(i * 1024));
      } else {
        const uint8_t* sb = rows + (row0 + (uint64_t)(lw * G::NA_I * A_RPI)) * stride + (size_t)ld_ks * G::A_ROWB;
#pragma unroll
        for (int i = 0; i < G::NA_I; i++) m3_dma16s<M2_A_NT || AF32>((AF32 && (i & 1)) ? voff_odd : voff, sb + (size_t)i * A_RPI * stride, slot + (uint32_t)(i * 1024));
      }
    } else {
      const uint32_t slot = lds0 + G::A_BYTES + (ld_g % NSB) * G::B_STAGE + (uint32_t)(lw * G::NB_I * 1024);
      const uint8_t* sb = reinterpret_cast<const uint8_t*>(q16) + (size_t)(lw * G::NB_I * 16) * dim * 2 + (size_t)ld_ks * 64;
#pragma unroll
      for (int i = 0; i < G::NB_I; i++) m3_dma16s<false>(voff, sb + (size_t)i * 16 * dim * 2, slot + (uint32_t)(i * 1024));
#pragma unroll
      for (int i = 0; i < G::NN_I; i++) {   // raw norms of the tile being loaded: this wave refreshes BM/4 of them, 64 per DMA
        const int off = i == 0 ? 0 : (BM / 4 - 64);   // the last piece ends exactly at the wave's share (pieces may overlap)
        const uint32_t dst = lds0 + (uint32_t)(G::A_BYTES + G::B_BYTES) + ld_par * (G::TNORM * 4) + (uint32_t)((lw * (BM / 4) + off) * 4);
        if constexpr (GATHER) m2_dma4(gnorm, dst);   // norms[gather[pos]]: one address per lane (BM = 256: one piece)
        else m3_dma4s(nvoff, norms + row0 + (uint64_t)(lw * (BM / 4) + off), dst);
      }
    }
#endif
    ld_g++;
    if (++ld_ks == nk) {
      ld_ks = 0;
      if (ld_tile != last_tile) { ld_tile += gridDim.x; ld_par ^= 1u; }  // past the end: re-fetch the last tile (uniform vmcnt)
      if constexpr (GATHER) {   // next tile: its slot numbers are in LDS; request those of the tile after the next one
        ld_n++;
        take_ids(ld_n);
        m2_dma4(gather + ids_pos(ld_n + 2), lds0 + (uint32_t)(reinterpret_cast<uint8_t*>(idbuf + ((ld_n + 2) % 3) * 64) - smem));
      }
    }
  };
  uint32_t fa[2][AF32 ? 2 : 1], fb[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) {
    if constexpr (AF32) {
      const int sw = (lane >> 1) & 7, c0 = kk * 4 + (lane >> 5) * 2;
      fa[kk][0] = (uint32_t)((lane & 31) * 128 + ((c0 ^ sw) << 4));
      fa[kk][1] = (uint32_t)((lane & 31) * 128 + (((c0 + 1) ^ sw) << 4));
    } else {
      fa[kk][0] = (uint32_t)((lane & 31) * 64 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 2) & 3)) << 4));
    }
    fb[kk] = (uint32_t)((lane & 31) * 64 + (((kk * 2 + (lane >> 5)) ^ ((lane >> 2) & 3)) << 4));
  }
  {  // prologue: each kind fills all but one of its slots
    const int pre = row_loader ? NSA - 1 : NSB - 1;
#pragma unroll 1
    for (int s = 0; s < pre; s++) issue_stage();
  }

  uint32_t g = 0, par = 0;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1u) {
    const uint64_t row0 = begin + tile * BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.f;
#pragma unroll 1
    for (int ks = 0; ks < nk; ks++, g++) {
      // my share of stage g has landed (my later stages may still fly) ...
      if (row_loader) m2_wait_vmcnt<(NSA - 2) * G::NA_I>(); else m2_wait_vmcnt<(NSB - 2) * (G::NB_I + G::NN_I)>();
      __builtin_amdgcn_s_barrier();   // ... and so has everybody's; everybody is done reading stage g-1 = the slots refilled next
      if (row_loader) issue_stage();
#ifdef COLTT_M2_NO_MFMA
      if (!row_loader) issue_stage();
      continue;
#endif
      const uint8_t* Ab = ringA + (size_t)(g % NSA) * G::A_STAGE + (size_t)(wm * WROWS) * G::A_ROWB;
      const uint8_t* Bb = ringB + (size_t)(g % NSB) * G::B_STAGE + (size_t)(wn * (BN / WN)) * 64;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        half8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
          if constexpr (AF32) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(Ab + tm * 32 * 128 + fa[kk][0]);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(Ab + tm * 32 * 128 + fa[kk][1]);
            a[tm] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
          } else a[tm] = *reinterpret_cast<const half8*>(Ab + tm * 32 * 64 + fa[kk][0]);
        }
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
#ifdef COLTT_M3_FAKE_FEWER_READS   // ablation only (WRONG answers): a third fewer fragment reads per step, same MFMA count — what a 128 x 128 wave tile would save
          if (tn & 1) { b[tn] = b[tn - 1]; continue; }
#endif
          b[tn] = *reinterpret_cast<const half8*>(Bb + tn * 32 * 64 + fb[kk]);
        }
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        if (kk == 0 && !row_loader) { __builtin_amdgcn_sched_barrier(0); issue_stage(); __builtin_amdgcn_sched_barrier(0); }
      }
    }
    const float* tn_raw = tnorm + par * G::TNORM;
#ifdef COLTT_M2_NO_EPI
    {
      float sum_ = 0.f;
      _Pragma("unroll") for (int tm = 0; tm < TM; tm++) _Pragma("unroll") for (int tn = 0; tn < TN; tn++) _Pragma("unroll") for (int r = 0; r < 16; r++) sum_ += acc[tm][tn][r];
      if (sum_ == 12345.678f) cnt[0] = 1;
    }
    continue;
#endif
    auto row_scale = [&](int tm, f32x4 (&ir)[4]) {
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        const f32x4 raw = *reinterpret_cast<const f32x4*>(tn_raw + wm * WROWS + tm * 32 + 8 * gq + 4 * (lane >> 5));
        ir[gq] = METRIC == M_COS ? f32x4{rsqrtf(raw.x), rsqrtf(raw.y), rsqrtf(raw.z), rsqrtf(raw.w)} : raw;   // Euclidean: raw ||row||^2
      }
    };
    // Batches up to 128 take the two-pass epilogue (flat_mfma.hpp [2]: test, one reservation per column, store); the batch-256 instances keep
    // m2_emit_block: their long segments run behind tight thresholds, and the extra mask bookkeeping cost the power-bound C3 shape 2.4 %
    // (4.83 -> 4.95 ms per batch; C2 0.650 -> 0.627, filtered batch 64 0.216 -> 0.200: profiles/r04p_epilogue_ab.md).
    constexpr bool TWO_PASS = BN < 256;
    // survivor masks of the tile's blocks: 16 bits per block, the TM blocks of a query column packed into one register (TM <= 2)
    static_assert(TM <= 2, "mask packing: two 16-bit masks per register");
    uint32_t msk[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) msk[tn] = 0;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
      f32x4 ir[4];
      row_scale(tm, ir);
      bool bad;
      if constexpr (METRIC == M_COS) bad = mf_bad_norms(ir);
      else {  // a non-finite norm always takes the element path
        bad = false;
#pragma unroll
        for (int gq = 0; gq < 4; gq++)
#pragma unroll
          for (int j = 0; j < 4; j++) bad |= !(ir[gq][j] >= 0.f && ir[gq][j] < __builtin_inff());
      }
      const uint64_t rbase = row0 + wm * WROWS + tm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        if constexpr (SEED || !TWO_PASS) m2_emit_block<SEED, METRIC>(acc[tm][tn], ir, bad, qc[tn], nearest, nq, rbase, begin, end, cand, cnt, cap, ep, GATHER ? gather : nullptr);
        else msk[tn] |= m2_test_block<METRIC>(acc[tm][tn], ir, bad, qc[tn], nearest, nq, rbase, end) << (16 * tm);
      }
    }
    if constexpr (!SEED && TWO_PASS) {   // one reservation per query column and tile (flat_mfma.hpp [2]: two-pass epilogue), then the blocks that had survivors
      uint32_t any = 0;
#pragma unroll
      for (int tn = 0; tn < TN; tn++) any |= msk[tn];
      if (any) {
        uint32_t idx[TN];
#pragma unroll
        for (int tn = 0; tn < TN; tn++) idx[tn] = msk[tn] ? atomicAdd(&cnt[qc[tn].qidx], (uint32_t)__builtin_popcount(msk[tn])) : 0u;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
          if (!((any >> (16 * tm)) & 0xffffu)) continue;
          f32x4 ir[4];
          row_scale(tm, ir);
          const uint64_t rbase = row0 + wm * WROWS + tm * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int tn = 0; tn < TN; tn++) {
            const uint32_t m = (msk[tn] >> (16 * tm)) & 0xffffu;
            if (m) idx[tn] = m2_store_block<METRIC>(acc[tm][tn], ir, m, qc[tn], rbase, cand, idx[tn], cap, ep, GATHER ? gather : nullptr);
          }
        }
      }
    }
  }
  m2_wait_vmcnt<0>();
}

}  // namespace dev
}  // namespace coltt
