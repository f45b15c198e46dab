// flat_mfma2.hpp — the batched FLAT candidate GEMM as a persistent LDS-DMA pipeline (second generation of flat_mfma.hpp's
// flat_mfma_cos_kernel; same contract: approximate scores only pick candidates, survivors are re-scored exactly).
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
#pragma once
#include "flat_mfma.hpp"

namespace coltt {
namespace dev {

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

// ---- two-pass form of the same epilogue (flat_mfma3.hpp) ---------------------------------------------------------------------------
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

}  // namespace dev
}  // namespace coltt
