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
                                              uint32_t* __restrict__ cnt, uint32_t cap, float* ep) {
  float t[16];
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
        if (gr < end) cand[(size_t)qc.qidx * cap + (uint32_t)(gr - begin)] = ((unsigned long long)score_key(s) << 32) | (uint32_t)gr;
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
        if (idx < cap) cand[(size_t)qc.qidx * cap + idx] = ((unsigned long long)score_key(s) << 32) | (uint32_t)gr;
        idx++;
      }
    }
  }
}

template <int BN, bool AF32, bool SEED>
__global__ __launch_bounds__(M2_NT, 2) void flat_mfma2_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, uint64_t begin, uint64_t end,
    const _Float16* __restrict__ q16, const float* __restrict__ qnorms, int nq, int dim, const uint32_t* __restrict__ thr,
    int nearest, unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  typedef M2Geom<BN, AF32> G;
  constexpr int WN = 2;
  constexpr int TM = 2, TN = BN / WN / 32;   // wave tile 64 rows x BN/2 queries
  constexpr int NS = G::NS;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  uint8_t* const ring = smem;                                                             // [NS][A_STAGE | B_STAGE]
  float* const tnorm = reinterpret_cast<float*>(smem + (size_t)NS * G::STAGE);            // [2][M2_TNORM] raw ||row||^2
  float* const ep = reinterpret_cast<float*>(smem + (size_t)NS * G::STAGE + 2 * M2_TNORM * 4) + tid * 8;
  const int nk = dim / M2_BK;
  const uint64_t ntiles = (end - begin + M2_BM - 1) / M2_BM;
  if ((uint64_t)blockIdx.x >= ntiles) return;
  // per-lane query constants for its TN columns (plain loads: consumed before the first DMA is issued)
  QCol qc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++) qc[tn] = mf_query_col(wn * (BN / WN) + tn * 32 + (lane & 31), nq, qnorms, thr, nearest);

  // ---- loader state: this lane's share of one stage ------------------------------------------------------------------------
  // rows: instruction j = wave * NA_I + i covers RPI rows; lane l -> local row j * RPI + l / CPR, LDS position p = l % CPR,
  // fetched chunk c = p ^ swizzle(row)
  constexpr int A_CPR = G::A_ROWB / 16, A_RPI = 64 / A_CPR;
  int a_lrow[G::NA_I]; uint32_t a_coff[G::NA_I];
#pragma unroll
  for (int i = 0; i < G::NA_I; i++) {
    const int lr = (wave * G::NA_I + i) * A_RPI + lane / A_CPR, p = lane % A_CPR;
    a_lrow[i] = lr;
    a_coff[i] = (uint32_t)((AF32 ? (p ^ ((lr >> 1) & 7)) : (p ^ ((lr >> 2) & 3))) * 16);
  }
  const uint8_t* b_ptr[G::NB_I];
#pragma unroll
  for (int i = 0; i < G::NB_I; i++) {
    const int j = (wave * G::NB_I + i) % (G::B_STAGE / 1024);
    const int q = j * 16 + lane / 4, p = lane % 4;
    b_ptr[i] = reinterpret_cast<const uint8_t*>(q16) + (size_t)q * dim * 2 + (size_t)((p ^ ((q >> 2) & 3)) * 16);
  }
  uint64_t ld_tile = blockIdx.x; int ld_ks = 0; uint32_t ld_g = 0, ld_par = 0;
  const uint64_t last_tile = blockIdx.x + ((ntiles - 1 - blockIdx.x) / gridDim.x) * gridDim.x;
  // per-lane source pointers of the tile being loaded (recomputed when the loader moves to the next tile: the per-step cost
  // of a DMA is then one 64-bit add, not a clamp + 64-bit multiply)
  const uint8_t* a_ptr[G::NA_I]; const float* n_ptr;
  auto loader_tile = [&]() {
    const uint64_t row0 = begin + ld_tile * M2_BM;
#pragma unroll
    for (int i = 0; i < G::NA_I; i++) {
      uint64_t gr = row0 + (uint64_t)a_lrow[i]; if (gr >= end) gr = end - 1;
      a_ptr[i] = rows + gr * stride + a_coff[i];
    }
    uint64_t gr = row0 + (uint64_t)(wave * 32 + lane); if (gr >= end) gr = end - 1;
    n_ptr = norms + gr;
  };
  loader_tile();
  // one DMA of the stage being loaded; piece PER-1 (the raw-norm refresh) also advances the loader
  auto issue_piece = [&](int pc) {
#ifdef COLTT_M2_NO_DMA
    if (pc == G::PER - 1) ld_g++;
    return;
#endif
    const uint32_t slot = lds0 + (ld_g % NS) * G::STAGE;
    if (pc < G::NA_I) {
      m2_dma16<M2_A_NT>(a_ptr[pc] + (size_t)ld_ks * G::A_ROWB, slot + (uint32_t)((wave * G::NA_I + pc) * 1024));
    } else if (pc < G::NA_I + G::NB_I) {
      const int i = pc - G::NA_I;
      m2_dma16<false>(b_ptr[i] + (size_t)ld_ks * 64, slot + G::A_STAGE + (uint32_t)(((wave * G::NB_I + i) % (G::B_STAGE / 1024)) * 1024));
    } else {
      // raw norms of the tile being loaded: wave w refreshes floats [32 w, 32 w + 64) of this tile parity's buffer
      m2_dma4(n_ptr, lds0 + (uint32_t)(NS * G::STAGE) + ld_par * (M2_TNORM * 4) + (uint32_t)(wave * 128));
      ld_g++;
      if (++ld_ks == nk) {
        ld_ks = 0;
        if (ld_tile != last_tile) { ld_tile += gridDim.x; ld_par ^= 1u; loader_tile(); }  // past the end: keep re-fetching the last tile (uniform vmcnt)
      }
    }
  };
  auto issue_stage = [&]() {
#pragma unroll
    for (int pc = 0; pc < G::PER; pc++) issue_piece(pc);
  };
  // fragment offsets (lane constants): chunk position under the swizzle
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
#pragma unroll 1
  for (int s = 0; s < NS - 1; s++) issue_stage();

  uint32_t g = 0, par = 0;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1u) {
    const uint64_t row0 = begin + tile * M2_BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.f;
#pragma unroll 1
    for (int ks = 0; ks < nk; ks++, g++) {
      m2_wait_vmcnt<(NS - 2) * G::PER>();   // my share of stage g has landed (stages g+1 .. g+NS-2 may still fly)
      __builtin_amdgcn_s_barrier();         // everybody's share has; everybody is done reading stage g-1 = the slot refilled next
      if (M2_ISSUE == 0 || (M2_ISSUE == 1 && wave < 4)) issue_stage();
#ifdef COLTT_M2_NO_MFMA
      if (!(M2_ISSUE == 0 || (M2_ISSUE == 1 && wave < 4))) issue_stage();
      continue;
#endif
      const uint8_t* Ab = ring + (size_t)(g % NS) * G::STAGE + (size_t)(wm * 64) * G::A_ROWB;
      const uint8_t* Bb = ring + (size_t)(g % NS) * G::STAGE + G::A_STAGE + (size_t)(wn * (BN / WN)) * 64;
      int pc_next = 0;
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
        for (int tn = 0; tn < TN; tn++) b[tn] = *reinterpret_cast<const half8*>(Bb + tn * 32 * 64 + fb[kk]);
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
#pragma unroll
          for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
          if constexpr (M2_ISSUE == 2) {  // spread the pieces: ceil(PER / 4) after each of the 4 MFMA groups of a K step
            constexpr int PG = (G::PER + 3) / 4;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < PG; j++) if (pc_next < G::PER) issue_piece(pc_next++);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (M2_ISSUE == 1 && kk == 0 && wave >= 4) { __builtin_amdgcn_sched_barrier(0); issue_stage(); __builtin_amdgcn_sched_barrier(0); }
      }
    }
    // ---- epilogue (the next tile's first stages are already in flight).  The raw norms of THIS tile were refreshed by every
    // K step's DMA; the value is the same each time, so a refresh still in flight is harmless.
    const float* tn_raw = tnorm + par * M2_TNORM;
#ifdef COLTT_M2_NO_EPI
    {  // keep EVERY accumulator alive (dead ones would take their MFMAs with them)
      float sum_ = 0.f;
      _Pragma("unroll") for (int tm = 0; tm < TM; tm++) _Pragma("unroll") for (int tn = 0; tn < TN; tn++) _Pragma("unroll") for (int r = 0; r < 16; r++) sum_ += acc[tm][tn][r];
      if (sum_ == 12345.678f) cnt[0] = 1;
    }
    continue;
#endif
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
      f32x4 ir[4];
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        const f32x4 raw = *reinterpret_cast<const f32x4*>(tn_raw + wm * 64 + tm * 32 + 8 * gq + 4 * (lane >> 5));
        ir[gq] = f32x4{rsqrtf(raw.x), rsqrtf(raw.y), rsqrtf(raw.z), rsqrtf(raw.w)};
      }
      const bool bad = mf_bad_norms(ir);
      const uint64_t rbase = row0 + wm * 64 + tm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < TN; tn++) m2_emit_block<SEED>(acc[tm][tn], ir, bad, qc[tn], nearest, nq, rbase, begin, end, cand, cnt, cap, ep);
    }
  }
  m2_wait_vmcnt<0>();  // do not leave DMA writes in flight into an LDS allocation that is about to be handed on
}

}  // namespace dev
}  // namespace coltt
