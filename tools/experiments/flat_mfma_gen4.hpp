// EXPERIMENT, not part of the default library (-DCOLTT_EXPERIMENTS; COLTT_MFMA_GEN=4 selects it): bit-identical to generation 3, 4.93 ms at C3 (gen 3: 4.8).
// flat_mfma4.hpp — fourth generation of the batched FLAT candidate GEMM: flat_mfma3.hpp's split DMA rings with the fragment
// reads SOFTWARE-PIPELINED one K step ahead of the matrix cores.
//
// What the third generation showed (10 M x 768 f16, batch 256; profiles/r02_flat_mfma_ablation.txt): per K step a CU needs
// 1024 clk of matrix-pipe time, 768 clk of LDS read time for the fragments (8 waves x 12 ds_read_b128) and 256 clk of LDS write
// time for the DMA — but after each s_barrier all eight waves first read fragments (matrix pipe idle) and then all issue MFMAs
// (LDS idle): measured 1600 clk per step for compute alone, 2550 with the DMA stream, against 1930 for the DMA stream alone.
// Here every wave runs its fragment reads HALF A STEP ahead of its MFMAs, at no cost in registers (two half-step fragment sets —
// what one unpipelined step held): while the 8 MFMAs of (stage g, K half 0) run, the ds_reads of (g, half 1) are in flight; while
// the 8 MFMAs of (g, half 1) run, those of (g+1, half 0).  The LDS pipe and the matrix pipe overlap inside each wave and the
// barrier no longer separates a read phase from a math phase.  The ring protocol moves with it:
//   * the barrier sits in the MIDDLE of a step, after "my share of stage g+1 has landed" and "my reads of stage g have completed":
//     behind it stage g+1 is readable and the slots of stage g are free half a step earlier than before;
//   * the row loaders (waves 0-3) refill right behind the barrier;
//   * the QUERY tile no longer rides the LDS-DMA path.  Cycle accounting (-DCOLTT_M4_TIMING) showed a K step costing ~2200 clk
//     with the waves stalled in DMA *issue* (back-pressure), at a clock throttled to ~1.7 GHz by the matrix cores: the LDS-DMA
//     path sustains ~15 B/clk/CU whatever the bytes are, and rows + queries = 32 KB per step.  Waves 4-7 now fetch the
//     (L2-resident) query stage with global_load_dwordx4 into registers one step ahead and ds_write_b128 it into the same
//     swizzled image; only the rows (16 KB per step) and the raw norms (once per tile) use LDS-DMA;
//   * the tile epilogue runs with the DMA queue full and (g+1, half 0) parked in registers.
// Everything else (256 x BN tile, 4 x 2 waves, XOR-swizzled lane-linear DMA image, saddr-form DMA, seed segment in place, one
// atomic per half block) is flat_mfma3.hpp's.  The raw-norm parity buffers require dim >= 128 (flat.hip checks).
#pragma once
#include "../../coltt_amd/csrc/flat_mfma.hpp"

namespace coltt {
namespace dev {

// vmcnt <= N and lgkmcnt == 0 (gfx9 encoding)
template <int N> __device__ __forceinline__ void m4_wait_vm_lgkm0() {
  static_assert(N >= 0 && N < 64, "vmcnt");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (0 << 8));
}

template <int BN, bool AF32, bool SEED, int METRIC = M_COS>
__global__ __launch_bounds__(M2_NT, 2) void flat_mfma4_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, uint64_t begin, uint64_t end,
    const _Float16* __restrict__ q16, const float* __restrict__ qnorms, int nq, int dim, const uint32_t* __restrict__ thr,
    int nearest, unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  constexpr int BM = M2_BM;
  typedef M3Geom<BN, AF32, BM> G;
  constexpr int WN = 2;
  constexpr int WROWS = BM / 4;
  constexpr int TM = WROWS / 32, TN = BN / WN / 32;
  constexpr int NSA = G::NSA, NSB = G::NSB;
  static_assert((NSA - 1) * G::NA_I < 64 && NSB >= 3, "vmcnt range / query ring");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
  const bool row_loader = wave < 4;
  const int lw = wave & 3;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  uint8_t* const ringA = smem;
  uint8_t* const ringB = smem + G::A_BYTES;
  float* const tnorm = reinterpret_cast<float*>(smem + G::A_BYTES + G::B_BYTES);
  float* const ep = reinterpret_cast<float*>(smem + G::A_BYTES + G::B_BYTES + 2 * G::TNORM * 4) + tid * 8;
  const int nk = dim / M2_BK;
  const uint64_t ntiles = (end - begin + BM - 1) / BM;
  if ((uint64_t)blockIdx.x >= ntiles) return;
  QCol qc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++)
    qc[tn] = METRIC == M_COS ? mf_query_col(wn * (BN / WN) + tn * 32 + (lane & 31), nq, qnorms, thr, nearest)
                             : m2_query_col_l2(wn * (BN / WN) + tn * 32 + (lane & 31), nq, qnorms, thr, nearest);

  // ---- loader state (as flat_mfma3.hpp) ---------------------------------------------------------------------------------------
  constexpr int A_CPR = G::A_ROWB / 16, A_RPI = 64 / A_CPR;
  uint32_t voff, voff_odd = 0;
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
  auto advance = [&]() {
    ld_g++;
    if (++ld_ks == nk) {
      ld_ks = 0;
      if (ld_tile != last_tile) { ld_tile += gridDim.x; ld_par ^= 1u; }   // past the end: re-fetch the last tile (uniform queues)
    }
  };
  auto issue_stage = [&]() {   // row loaders: my share of the stage the loader points at -> LDS (DMA), then advance
    const uint64_t row0 = begin + ld_tile * BM;
    const uint32_t slot = lds0 + (ld_g % NSA) * G::A_STAGE + (uint32_t)(lw * G::NA_I * 1024);
    const uint8_t* sb = rows + (row0 + (uint64_t)(lw * G::NA_I * A_RPI)) * stride + (size_t)ld_ks * G::A_ROWB;
#pragma unroll
    for (int i = 0; i < G::NA_I; i++) m3_dma16s<M2_A_NT || AF32>((AF32 && (i & 1)) ? voff_odd : voff, sb + (size_t)i * A_RPI * stride, slot + (uint32_t)(i * 1024));
    advance();
  };
  // query loaders: the L2-resident query tile goes through REGISTERS (global_load_dwordx4, then ds_write_b128 one step later),
  // not through the LDS-DMA path: that path moves ~15 B/clk/CU however the bytes are split, and the rows need all of it.
  u32x4 qreg[G::NB_I];
  uint32_t q_slot = 0;   // LDS byte offset the registers go to
  auto q_load = [&]() {
    const uint8_t* sb = reinterpret_cast<const uint8_t*>(q16) + (size_t)(lw * G::NB_I * 16) * dim * 2 + (size_t)ld_ks * 64;
#pragma unroll
    for (int i = 0; i < G::NB_I; i++) qreg[i] = *reinterpret_cast<const u32x4*>(sb + (size_t)i * 16 * dim * 2 + voff);
    q_slot = (uint32_t)G::A_BYTES + (ld_g % NSB) * G::B_STAGE + (uint32_t)(lw * G::NB_I * 1024) + (uint32_t)lane * 16;
    if (ld_ks == 0) {   // first stage of a tile: its raw norms, BM/4 per wave, 64 per DMA, into the tile-parity buffer
      const uint64_t row0 = begin + ld_tile * BM;
#pragma unroll
      for (int i = 0; i < G::NN_I; i++) {
        const int off = i == 0 ? 0 : (BM / 4 - 64);
        m3_dma4s(nvoff, norms + row0 + (uint64_t)(lw * (BM / 4) + off),
                 lds0 + (uint32_t)(G::A_BYTES + G::B_BYTES) + ld_par * (G::TNORM * 4) + (uint32_t)((lw * (BM / 4) + off) * 4));
      }
    }
    advance();
  };
  auto q_store = [&]() {
#pragma unroll
    for (int i = 0; i < G::NB_I; i++) *reinterpret_cast<u32x4*>(smem + q_slot + (uint32_t)(i * 1024)) = qreg[i];
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
  struct Half { half8 a[TM]; half8 b[TN]; };   // the fragments of one K half (16 of the 32 columns of a stage)
  auto read_half = [&](Half& f, uint32_t gs, int kk) {
    const uint8_t* Ab = ringA + (size_t)(gs % NSA) * G::A_STAGE + (size_t)(wm * WROWS) * G::A_ROWB;
    const uint8_t* Bb = ringB + (size_t)(gs % NSB) * G::B_STAGE + (size_t)(wn * (BN / WN)) * 64;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
      if constexpr (AF32) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(Ab + tm * 32 * 128 + fa[kk][0]);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(Ab + tm * 32 * 128 + fa[kk][1]);
        f.a[tm] = half8{(_Float16)lo.x, (_Float16)lo.y, (_Float16)lo.z, (_Float16)lo.w, (_Float16)hi.x, (_Float16)hi.y, (_Float16)hi.z, (_Float16)hi.w};
      } else f.a[tm] = *reinterpret_cast<const half8*>(Ab + tm * 32 * 64 + fa[kk][0]);
    }
#pragma unroll
    for (int tn = 0; tn < TN; tn++) f.b[tn] = *reinterpret_cast<const half8*>(Bb + tn * 32 * 64 + fb[kk]);
  };
  constexpr int PER_A = G::NA_I;

  // ---- prologue: rows fill every slot; query stages 0 and 1 go to LDS, stage 2 stays in flight; park (stage 0, half 0) ---------
  if (row_loader) {
#pragma unroll 1
    for (int s = 0; s < NSA; s++) issue_stage();
  } else {
#pragma unroll 1
    for (int s = 0; s < 2; s++) { q_load(); m2_wait_vmcnt<0>(); q_store(); }
    q_load();
  }
  Half h0, h1;
  if (row_loader) m2_wait_vmcnt<(NSA - 1) * PER_A>(); else m4_wait_vm_lgkm0<63>();
  __builtin_amdgcn_s_barrier();
  read_half(h0, 0, 0);

#ifdef COLTT_M4_TIMING
  // per-wave cycle accounting (s_memtime), printed by two workgroups at the end: where does a K step go?
  unsigned long long t_q = 0, t_m0 = 0, t_w = 0, t_b = 0, t_i = 0, t_m1 = 0, t_e = 0, t_mark = __builtin_readcyclecounter();
  const unsigned long long t_begin = t_mark, w_begin = wall_clock64();
#define M4_LAP(acc_) { const unsigned long long now_ = __builtin_readcyclecounter(); acc_ += now_ - t_mark; t_mark = now_; }
#else
#define M4_LAP(acc_)
#endif
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
      // top of step g: h0 = (g, half 0); stage g has landed.  Query loaders hold stage g+2 in registers (loaded a step ago): it
      // goes to the slot stage g-1 left at the last barrier, and the loads of stage g+3 take the registers over.
      if (!row_loader) { m2_wait_vmcnt<0>(); q_store(); q_load(); __builtin_amdgcn_sched_barrier(0); }
      M4_LAP(t_q)
      read_half(h1, g, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0.a[tm], h0.b[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      M4_LAP(t_m0)
      // my share of stage g+1 has landed and my reads of stage g are complete ...
      if (row_loader) m4_wait_vm_lgkm0<(NSA - 2) * PER_A>(); else m4_wait_vm_lgkm0<63>();   // (query stores: lgkmcnt)
      M4_LAP(t_w)
      __builtin_amdgcn_s_barrier();   // ... and so for everybody: stage g+1 is readable, the slots of stage g are free
      M4_LAP(t_b)
      if (row_loader) { issue_stage(); __builtin_amdgcn_sched_barrier(0); }
      M4_LAP(t_i)
      read_half(h0, g + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int tm = 0; tm < TM; tm++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1.a[tm], h1.b[tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      M4_LAP(t_m1)
    }
    const float* tn_raw = tnorm + par * G::TNORM;
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
      f32x4 ir[4];
#pragma unroll
      for (int gq = 0; gq < 4; gq++) {
        const f32x4 raw = *reinterpret_cast<const f32x4*>(tn_raw + wm * WROWS + tm * 32 + 8 * gq + 4 * (lane >> 5));
        ir[gq] = METRIC == M_COS ? f32x4{rsqrtf(raw.x), rsqrtf(raw.y), rsqrtf(raw.z), rsqrtf(raw.w)} : raw;
      }
      bool bad;
      if constexpr (METRIC == M_COS) bad = mf_bad_norms(ir);
      else {
        bad = false;
#pragma unroll
        for (int gq = 0; gq < 4; gq++)
#pragma unroll
          for (int j = 0; j < 4; j++) bad |= !(ir[gq][j] >= 0.f && ir[gq][j] < __builtin_inff());
      }
      const uint64_t rbase = row0 + wm * WROWS + tm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < TN; tn++) m2_emit_block<SEED, METRIC>(acc[tm][tn], ir, bad, qc[tn], nearest, nq, rbase, begin, end, cand, cnt, cap, ep);
    }
    M4_LAP(t_e)
  }
  m2_wait_vmcnt<0>();
#ifdef COLTT_M4_TIMING
  M4_LAP(t_e)
  if ((blockIdx.x == 0 || blockIdx.x == 131) && lane == 0 && g > 100) {
    const unsigned long long tot = __builtin_readcyclecounter() - t_begin, wall = wall_clock64() - w_begin;
    printf("m4 wg %3d wave %d: %u steps, per step clk: q-issue %llu  math0 %llu  vm-wait %llu  barrier %llu  row-issue %llu  math1 %llu | epilogue/step %llu | total %llu clk = %llu x10ns (%.0f MHz)\n",
           (int)blockIdx.x, wave, g, t_q / g, t_m0 / g, t_w / g, t_b / g, t_i / g, t_m1 / g, t_e / g, tot, wall, (double)tot / (double)wall * 100.0);
  }
#endif
}

}  // namespace dev
}  // namespace coltt
