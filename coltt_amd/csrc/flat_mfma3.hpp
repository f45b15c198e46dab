// flat_mfma3.hpp — third generation of the batched FLAT candidate GEMM: the LDS-DMA ring of flat_mfma2.hpp with SPLIT rings
// and SPECIALISED loader waves.
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
//     with the in-place seed segment and one atomic per half block) is flat_mfma2.hpp's.
#pragma once
#include "flat_mfma2.hpp"

namespace coltt {
namespace dev {

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
    // Batches up to 128 take the two-pass epilogue (flat_mfma2.hpp: test, one reservation per column, store); the batch-256 instances keep
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
    if constexpr (!SEED && TWO_PASS) {   // one reservation per query column and tile (flat_mfma2.hpp: two-pass epilogue), then the blocks that had survivors
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
