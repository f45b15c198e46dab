// flat_mfma.hpp — matrix-core candidate generation for the batched FLAT scan (kernel K2 of SURVEY.md §2).
//
// The reference scores one (query, vector) pair at a time (edge/none_vectorstore.go:136-147); a GPU serving a batch of B
// queries turns the scan into a [rows x D] x [D x B] GEMM.  MFMA accumulates in a different order than the reference's AVX
// kernel, so its scores are NOT the reference's bits.  They are used only to pick candidates:
//
//   1. flat_mfma_cos_f16_kernel : s~ = |1 - dot/sqrt(nq*nr)| for every (row, query) of a 128-row tile with
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

constexpr int MF_BM = 128;        // rows per workgroup tile
constexpr int MF_BK = 64;         // halves per K step (128 B of every row per step)
constexpr int MF_LD = MF_BK + 8;  // padded LDS row (144 B): conflict-free ds_read_b128 fragment reads
constexpr float MF_MARGIN = 6e-4f;

template <int BN> constexpr size_t mfma_lds_bytes() { return (size_t)2 * (MF_BM + BN) * MF_LD * 2 + MF_BM * 4; }

// queries as f16 [BN][dim]; q_eff holds values that are exactly representable in binary16 (they went through Lower)
__global__ void mfma_prep_queries_kernel(const float* __restrict__ q_eff, int nq, int bn, int dim, _Float16* __restrict__ q16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)bn * dim) return;
  int q = (int)(i / dim);
  q16[i] = q < nq ? (_Float16)q_eff[i] : (_Float16)0.f;
}

template <int BN>
__global__ __launch_bounds__(256) void flat_mfma_cos_f16_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, uint64_t begin, uint64_t end,
    const _Float16* __restrict__ q16, const float* __restrict__ qnorms, int nq, int dim, const uint32_t* __restrict__ thr,
    int nearest, unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  constexpr int TN = BN / 64;  // 32-wide query tiles per wave (wave grid 2 x 2)
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  _Float16* As = reinterpret_cast<_Float16*>(smem);                    // [2][MF_BM][MF_LD]
  _Float16* Bs = As + 2 * MF_BM * MF_LD;                               // [2][BN][MF_LD]
  float* tnorm = reinterpret_cast<float*>(Bs + 2 * BN * MF_LD);        // [MF_BM]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int nk = dim / MF_BK;
  // per-lane query constants for its TN columns
  float qn[TN]; uint32_t th[TN]; int qidx[TN];
#pragma unroll
  for (int tn = 0; tn < TN; tn++) {
    qidx[tn] = wn * (BN / 2) + tn * 32 + (lane & 31);
    qn[tn] = qidx[tn] < nq ? qnorms[qidx[tn]] : 1.f;
    th[tn] = qidx[tn] < nq ? thr[qidx[tn]] : 0u;
  }
  const uint64_t ntiles = (end - begin + MF_BM - 1) / MF_BM;
  for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint64_t row0 = begin + tile * MF_BM;
    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0.f;
    __syncthreads();  // previous tile's epilogue is done with tnorm / LDS
    if (tid < MF_BM) { uint64_t r = row0 + tid; tnorm[tid] = norms[r < end ? r : end - 1]; }
    // ---- global -> registers -> LDS staging of one K step
    uint4 ra[4], rb[BN / 32];
    auto gload = [&](int ks) {
      const int k0 = ks * MF_BK;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int c = tid + 256 * i, r = c >> 3, c16 = c & 7;
        uint64_t gr = row0 + r; if (gr >= end) gr = end - 1;
        ra[i] = *reinterpret_cast<const uint4*>(rows + gr * stride + (size_t)(k0 + c16 * 8) * 2);
      }
#pragma unroll
      for (int i = 0; i < BN / 32; i++) {
        int c = tid + 256 * i, q = c >> 3, c16 = c & 7;
        rb[i] = *reinterpret_cast<const uint4*>(q16 + (size_t)q * dim + k0 + c16 * 8);
      }
    };
    auto lstore = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        int c = tid + 256 * i, r = c >> 3, c16 = c & 7;
        *reinterpret_cast<uint4*>(As + ((size_t)buf * MF_BM + r) * MF_LD + c16 * 8) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < BN / 32; i++) {
        int c = tid + 256 * i, q = c >> 3, c16 = c & 7;
        *reinterpret_cast<uint4*>(Bs + ((size_t)buf * BN + q) * MF_LD + c16 * 8) = rb[i];
      }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ks++) {
      const int buf = ks & 1;
      if (ks + 1 < nk) gload(ks + 1);
      const _Float16* Ab = As + (size_t)buf * MF_BM * MF_LD;
      const _Float16* Bb = Bs + (size_t)buf * BN * MF_LD;
#pragma unroll
      for (int kk = 0; kk < MF_BK / 16; kk++) {
        const int kofs = kk * 16 + (lane >> 5) * 8;
        half8 a[2], b[TN];
#pragma unroll
        for (int tm = 0; tm < 2; tm++) a[tm] = *reinterpret_cast<const half8*>(Ab + (size_t)(wm * 64 + tm * 32 + (lane & 31)) * MF_LD + kofs);
#pragma unroll
        for (int tn = 0; tn < TN; tn++) b[tn] = *reinterpret_cast<const half8*>(Bb + (size_t)(wn * (BN / 2) + tn * 32 + (lane & 31)) * MF_LD + kofs);
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
      }
      if (ks + 1 < nk) lstore(buf ^ 1);
      __syncthreads();
    }
    // ---- epilogue: approximate cosine distance, threshold filter, candidate emission
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int rl = wm * 64 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const uint64_t gr = row0 + rl;
          float s = fabsf(1.0f - acc[tm][tn][r] * rsqrtf(qn[tn] * tnorm[rl]));
          uint32_t sk = score_key(s);
          bool pass = gr < end && qidx[tn] < nq && (nearest ? sk <= th[tn] : sk >= th[tn]);
          if (pass) {
            uint32_t idx = atomicAdd(&cnt[qidx[tn]], 1u);
            if (idx < cap) cand[(size_t)qidx[tn] * cap + idx] = ((unsigned long long)sk << 32) | (uint32_t)gr;
          }
        }
  }
}

// Per query: the k-th best approximate key, keep every candidate within MF_MARGIN of it (compacted into `dst`), publish the
// bound as the next scan threshold.  One block of 256 threads per query.
__global__ __launch_bounds__(256) void flat_pick_kernel(const unsigned long long* __restrict__ src_all, unsigned long long* __restrict__ dst_all,
                                                       uint32_t* __restrict__ cnt_all, uint32_t* __restrict__ thr_all, uint32_t cap,
                                                       uint32_t k, int nearest, float margin, uint32_t* __restrict__ overflow) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_digit, s_need, s_n;
  const int q = blockIdx.x, tid = threadIdx.x;
  const unsigned long long* src = src_all + (size_t)q * cap;
  unsigned long long* dst = dst_all + (size_t)q * cap;
  uint32_t c = cnt_all[q];
  if (c > cap) { if (tid == 0) atomicOr(overflow, 1u); c = cap; }
  const uint32_t flip = nearest ? 0u : 0xffffffffu;
  uint32_t bound_key = 0xffffffffu;  // in key' space: keep key' <= bound_key
  if (c > k) {
    uint32_t prefix = 0, mask = 0, need = k;
    for (int pass = 3; pass >= 0; pass--) {
      const int shift = pass * 8;
      hist[tid] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < c; i += 256) {
        uint32_t kp = (uint32_t)(src[i] >> 32) ^ flip;
        if ((kp & mask) == prefix) atomicAdd(&hist[(kp >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        uint32_t cum = 0, b = 0;
        for (; b < 256; b++) { if (cum + hist[b] >= need) break; cum += hist[b]; }
        s_digit = b; s_need = need - cum;
      }
      __syncthreads();
      prefix |= s_digit << shift; mask |= 255u << shift; need = s_need;
      __syncthreads();
    }
    // k-th best approximate score, widened by the margin in the "worse" direction
    float t = key_score(prefix ^ flip);
    float b = nearest ? t + margin : t - margin;
    bound_key = score_key(b) ^ flip;
    if (bound_key < prefix) bound_key = prefix;  // NaN / saturation guard
  }
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (uint32_t i = tid; i < c; i += 256) {
    unsigned long long e = src[i];
    if (((uint32_t)(e >> 32) ^ flip) <= bound_key) dst[atomicAdd(&s_n, 1u)] = e;
  }
  __syncthreads();
  if (tid == 0) { cnt_all[q] = s_n; thr_all[q] = c > k ? (bound_key ^ flip) : (nearest ? 0xffffffffu : 0u); }
}

// Exact re-score of the survivors: lane pair per (query, candidate); the key's score half is replaced by the exact score.
template <int QUANT>
__global__ __launch_bounds__(64) void flat_rescore_kernel(const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms,
                                                         const float* __restrict__ q_eff, const float* __restrict__ qnorms, int dim,
                                                         unsigned long long* __restrict__ cand_all, const uint32_t* __restrict__ cnt_all,
                                                         uint32_t cap) {
  const int q = blockIdx.y;
  const int lane = threadIdx.x, half = lane & 1;
  const uint32_t j = blockIdx.x * 32 + (lane >> 1);
  const uint32_t c = cnt_all[q] < cap ? cnt_all[q] : cap;
  if (blockIdx.x * 32 >= c) return;
  unsigned long long* cand = cand_all + (size_t)q * cap;
  const bool valid = j < c;
  const uint32_t slot = (uint32_t)cand[valid ? j : 0];
  float d = pair_distance<M_COS, QUANT, 4>(rows + (size_t)slot * stride, q_eff + (size_t)q * dim, dim, qnorms[q], norms[slot], half);
  if (valid && half == 0) cand[j] = ((unsigned long long)score_key(d) << 32) | slot;
}

}  // namespace dev
}  // namespace coltt
