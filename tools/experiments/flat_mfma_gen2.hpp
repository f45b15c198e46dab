// flat_mfma_gen2.hpp — EXPERIMENT, not part of the default library (-DCOLTT_EXPERIMENTS; COLTT_MFMA_GEN=2 selects it).
// Second-generation FLAT matrix-core kernel: 256 x B tile, 8 waves, one persistent workgroup per CU, LDS-DMA into ONE ring of
// 4 x 32 KB stages.  C3 5.63 ms; superseded by the split-ring kernel (coltt_amd/csrc/flat_mfma3.hpp), which reuses this
// generation's DMA / epilogue helpers (they stay in coltt_amd/csrc/flat_mfma2.hpp).
#pragma once
#include "../../coltt_amd/csrc/flat_mfma.hpp"

namespace coltt {
namespace dev {

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
