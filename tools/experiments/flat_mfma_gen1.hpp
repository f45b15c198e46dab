// flat_mfma_gen1.hpp — EXPERIMENT, not part of the default library (-DCOLTT_EXPERIMENTS compiles it in; COLTT_MFMA_GEN=1 selects it).
// First-generation FLAT matrix-core kernel (round 1): 128-row tiles, 4 waves, two workgroups per CU, global -> VGPR -> ds_write, padded
// LDS rows.  C3 (10 M x 768 f16, batch 256) 5.93 ms; superseded by the LDS-DMA ring kernels (coltt_amd/csrc/flat_mfma3.hpp, 4.8 ms).
// Kept for A/B runs (tools/flat_ab.py); shares the epilogue helpers of coltt_amd/csrc/flat_mfma.hpp.
#pragma once
#include "../../coltt_amd/csrc/flat_mfma.hpp"

namespace coltt {
namespace dev {

// 256 threads = 4 waves per workgroup, wave grid 2 x 2 over the 128 x BN tile; stages are small enough (61 KB of LDS at
// BN = 256) for TWO workgroups per CU, so one group's global-load / barrier stalls are covered by the other's MFMAs.
// Registers must stay <= 256 per lane and must not spill (a scratch reload is a VMEM op: it drains the in-order vmcnt queue).
constexpr int MF_NT = 256;
// AF32: the stored rows are f32 (COLTT_Q_NONE); they are rounded to binary16 on their way into LDS (candidate generation
// only — MF_MARGIN_F32 covers the rounding), so the same f16 matrix-core loop serves both row formats.
template <int BN, bool AF32>
__global__ __launch_bounds__(MF_NT, (MF_BM <= 64 ? 3 : (MF_BK <= 32 ? 2 : 1))) void flat_mfma_cos_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, uint64_t begin, uint64_t end,
    const _Float16* __restrict__ q16, const float* __restrict__ qnorms, int nq, int dim, const uint32_t* __restrict__ thr,
    int nearest, unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  constexpr int WN = 2, WM = 2;
  constexpr int TM = MF_BM / WM / 32, TN = BN / WN / 32;
  constexpr int NA = MF_BM * MF_CPR / MF_NT, NB = BN * MF_CPR / MF_NT;  // 16-byte chunks per thread per K step
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  _Float16* As = reinterpret_cast<_Float16*>(smem);                    // [2][MF_BM][MF_LD]
  _Float16* Bs = As + 2 * MF_BM * MF_LD;                               // [2][BN][MF_LD]
  float* tnorm = reinterpret_cast<float*>(Bs + 2 * BN * MF_LD);        // [MF_BM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN;
  const int nk = dim / MF_BK;
  // per-lane query constants for its TN columns: 1/||q||, the threshold as a float (+-inf = everything passes)
  QCol qc[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++) qc[tn] = mf_query_col(wn * (BN / WN) + tn * 32 + (lane & 31), nq, qnorms, thr, nearest);
  const uint64_t ntiles = (end - begin + MF_BM - 1) / MF_BM;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t row0 = begin + tile * MF_BM;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.f;
    __syncthreads();  // previous tile's epilogue is done with tnorm / LDS
    if (tid < MF_BM) { uint64_t r = row0 + tid; tnorm[tid] = rsqrtf(norms[r < end ? r : end - 1]); }  // 1/||row||; NaN rows never pass... see below
    // ---- global -> registers -> LDS staging.  Rows (HBM, ~2 us under load) are requested TWO K steps ahead, queries
    // (L2-resident) one step ahead: with one workgroup per CU, bytes in flight per CU are what buys HBM bandwidth.
    // (ext_vector types and macros on purpose: HIP's uint4 struct / lambda captures put these arrays in scratch.)
    u32x4 ra[NA], ra_hi[AF32 ? NA : 1], rb[NB];  // raw bits; f32 rows are converted at LDS-store time, not at load time
#define MF_GLOAD(KS)                                                                                         \
    {                                                                                                        \
      const int k0_ = (KS) * MF_BK;                                                                          \
      _Pragma("unroll") for (int i = 0; i < NA; i++) {                                                       \
        int c = tid + MF_NT * i, r = c / MF_CPR, c16 = c % MF_CPR;                                           \
        uint64_t gr = row0 + r; if (gr >= end) gr = end - 1;                                                 \
        if constexpr (AF32) {                                                                                \
          const uint8_t* p_ = rows + gr * stride + (size_t)(k0_ + c16 * 8) * 4;                              \
          ra[i] = *reinterpret_cast<const u32x4*>(p_); ra_hi[i] = *reinterpret_cast<const u32x4*>(p_ + 16);  \
        } else ra[i] = *reinterpret_cast<const u32x4*>(rows + gr * stride + (size_t)(k0_ + c16 * 8) * 2);    \
      }                                                                                                      \
      _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                       \
        int c = tid + MF_NT * i, q = c / MF_CPR, c16 = c % MF_CPR;                                           \
        rb[i] = *reinterpret_cast<const u32x4*>(q16 + (size_t)q * dim + k0_ + c16 * 8);                      \
      }                                                                                                      \
    }
#define MF_LSTORE(BUF)                                                                                       \
    {                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < NA; i++) {                                                       \
        int c = tid + MF_NT * i, r = c / MF_CPR, c16 = c % MF_CPR;                                           \
        u32x4 v_ = ra[i];                                                                                    \
        if constexpr (AF32) {                                                                                \
          f32x4 lo_ = __builtin_bit_cast(f32x4, ra[i]), hi_ = __builtin_bit_cast(f32x4, ra_hi[i]);           \
          half8 h_ = {(_Float16)lo_.x, (_Float16)lo_.y, (_Float16)lo_.z, (_Float16)lo_.w,                     \
                      (_Float16)hi_.x, (_Float16)hi_.y, (_Float16)hi_.z, (_Float16)hi_.w};                    \
          v_ = __builtin_bit_cast(u32x4, h_);                                                                \
        }                                                                                                    \
        *reinterpret_cast<u32x4*>(As + ((size_t)(BUF) * MF_BM + r) * MF_LD + c16 * 8) = v_;                  \
      }                                                                                                      \
      _Pragma("unroll") for (int i = 0; i < NB; i++) {                                                       \
        int c = tid + MF_NT * i, q = c / MF_CPR, c16 = c % MF_CPR;                                           \
        *reinterpret_cast<u32x4*>(Bs + ((size_t)(BUF) * BN + q) * MF_LD + c16 * 8) = rb[i];                  \
      }                                                                                                      \
    }
    MF_GLOAD(0);
    MF_LSTORE(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ks++) {
      const int buf = ks & 1;
      if (ks + 1 < nk) MF_GLOAD(ks + 1);
      const _Float16* Ab = As + (size_t)buf * MF_BM * MF_LD;
      const _Float16* Bb = Bs + (size_t)buf * BN * MF_LD;
#pragma unroll
      for (int kk = 0; kk < MF_BK / 16; kk++) {
        const int kofs = kk * 16 + (lane >> 5) * 8;
        half8 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; tm++) a[tm] = *reinterpret_cast<const half8*>(Ab + (size_t)(wm * (MF_BM / WM) + tm * 32 + (lane & 31)) * MF_LD + kofs);
#pragma unroll
        for (int tn = 0; tn < TN; tn++) b[tn] = *reinterpret_cast<const half8*>(Bb + (size_t)(wn * (BN / WN) + tn * 32 + (lane & 31)) * MF_LD + kofs);
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
      if (ks + 1 < nk) MF_LSTORE(buf ^ 1);
      __syncthreads();
    }
#undef MF_GLOAD
#undef MF_LSTORE
    // ---- epilogue: block filter, then (rarely) the per-element test s~ = |1 - dot * (1/||q||) * (1/||r||)| — see mf_emit_block
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
      f32x4 ir[4];  // 1/||row|| of this lane's 16 rows: 4 runs of 4 consecutive rows
#pragma unroll
      for (int g = 0; g < 4; g++) ir[g] = *reinterpret_cast<const f32x4*>(tnorm + wm * (MF_BM / WM) + tm * 32 + 8 * g + 4 * (lane >> 5));
      const bool bad = mf_bad_norms(ir);
      const uint64_t rbase = row0 + wm * (MF_BM / WM) + tm * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
        mf_emit_block(acc[tm][tn], ir, bad, qc[tn], nearest, nq, rbase, end, cand, cnt, cap, reinterpret_cast<float*>(smem) + tid * 16);
    }
  }
}

}  // namespace dev
}  // namespace coltt
