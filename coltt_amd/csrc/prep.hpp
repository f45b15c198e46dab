// prep.hpp — ingest-side kernels shared by the FLAT store and the HNSW index: Normalize + Lower of stored
// vectors, AVX-order row norms, and the query-side Normalize / Lower / decode.
#pragma once
#include <algorithm>
#include "exact.hpp"

namespace coltt {
namespace dev {

// ---------------------------------------------------------------------------------------------------
// Normalize (edge/vectorstore.go:173-189) + Quantization.Lower (edge/f16_quantization.go:47-53 ...).
// One wave per vector.  The reference's norm is ONE sequential f32 chain (norm += v[i]*v[i]), so one lane walks it — over an
// LDS copy fetched with coalesced loads — and the element-wise half (divide, encode, store) runs on all 64 lanes.  (One
// thread per vector, each striding through its own row, ingested at 0.2 TB/s.)
// ---------------------------------------------------------------------------------------------------
constexpr int PQ_CHUNK = 4096;
// norm += v[0]^2, += v[1]^2, ... strictly in order (mul, then add: no FMA), reading LDS 16 values at a time so the chain
// pays the LDS latency once per 16 elements instead of once per element.
__device__ __forceinline__ float seq_sqsum(float norm, const float* v, int m) {
  int e = 0;
  for (; e + 16 <= m; e += 16) {
    f32x4 a = *reinterpret_cast<const f32x4*>(v + e), b = *reinterpret_cast<const f32x4*>(v + e + 4);
    f32x4 c = *reinterpret_cast<const f32x4*>(v + e + 8), d = *reinterpret_cast<const f32x4*>(v + e + 12);
    f32x4 pa = a * a, pb = b * b, pc = c * c, pd = d * d;
    norm += pa.x; norm += pa.y; norm += pa.z; norm += pa.w;
    norm += pb.x; norm += pb.y; norm += pb.z; norm += pb.w;
    norm += pc.x; norm += pc.y; norm += pc.z; norm += pc.w;
    norm += pd.x; norm += pd.y; norm += pd.z; norm += pd.w;
  }
  for (; e < m; e++) { float x = v[e]; norm += x * x; }
  return norm;
}
template <int QUANT>
__global__ __launch_bounds__(64) void prep_rows_kernel(const float* __restrict__ raw, uint64_t n, int dim, int normalize,
                                                       const uint32_t* __restrict__ slots, uint64_t slot_base,
                                                       uint8_t* __restrict__ rows, size_t stride) {
  __shared__ __attribute__((aligned(16))) float buf[PQ_CHUNK];
  __shared__ float s_norm;
  const uint64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const float* v = raw + i * (uint64_t)dim;
  const uint64_t slot = slots ? slots[i] : slot_base + i;
  uint8_t* out = rows + slot * stride;
  float norm = 0.f;
  bool zero = false;
  if (normalize) {
    for (int c0 = 0; c0 < dim; c0 += PQ_CHUNK) {
      const int m = min(PQ_CHUNK, dim - c0);
      for (int e = lane; e < m; e += 64) buf[e] = v[c0 + e];
      __syncthreads();
      if (lane == 0) norm = seq_sqsum(norm, buf, m);
      __syncthreads();
    }
    if (lane == 0) s_norm = norm;
    __syncthreads();
    norm = s_norm;
    zero = (norm == 0.f);
    norm = go_sqrt(norm);
  }
  for (int e = lane; e < dim; e += 64) {
    float x = v[e];
    if (normalize) x = zero ? 0.f : div_rn(x, norm);
    if constexpr (QUANT == Q_NONE) reinterpret_cast<float*>(out)[e] = x;
    else if constexpr (QUANT == Q_F8) out[e] = (uint8_t)f32bits_to_f8bits(__float_as_uint(x));
    else reinterpret_cast<unsigned short*>(out)[e] = (unsigned short)f32bits_to_f16bits(__float_as_uint(x));
  }
  for (size_t bb = (size_t)dim * elem_bytes<QUANT>() + lane; bb < stride; bb += 64) out[bb] = 0;
}
template <int QUANT>
inline void launch_prep_rows(hipStream_t st, const float* raw, uint64_t n, int dim, int normalize, const uint32_t* slots,
                             uint64_t slot_base, uint8_t* rows, size_t stride) {
  const uint64_t step = 1ull << 24;  // grid x 64 threads must stay below 2^32
  for (uint64_t o = 0; o < n; o += step) {
    const uint64_t m = std::min<uint64_t>(step, n - o);
    prep_rows_kernel<QUANT><<<(unsigned)m, 64, 0, st>>>(raw + o * (uint64_t)dim, m, dim, normalize, slots ? slots + o : nullptr,
                                                       slot_base + o, rows, stride);
  }
}

// ||row||^2 in AVX order, one lane pair per row.
template <int QUANT>
__global__ void row_norms_kernel(const uint8_t* __restrict__ rows, size_t stride, const uint32_t* __restrict__ slots,
                                 uint64_t slot_base, uint64_t n, int dim, float* __restrict__ norms, uint32_t* __restrict__ max_bits = nullptr) {
  uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  int half = threadIdx.x & 1;
  bool valid = pair < n;
  uint64_t slot = valid ? (slots ? slots[pair] : slot_base + pair) : (slots ? slots[0] : slot_base);
  float s = pair_sqnorm<QUANT>(rows + slot * stride, dim, half);
  if (valid && half == 0) {
    norms[slot] = s;
    // running upper bound of ||row||^2 over everything ever stored (bit pattern order == value order for s >= 0; NaN sorts
    // above +inf, so a store that ever held a non-finite row keeps its matrix-core Euclidean path switched off)
    // max_bits[1] = max of the COMPLEMENTED bits = the running lower bound (zero-initialised like the upper one): the cosine
    // matrix-core path needs every stored norm near 1 (flat.hip: search_prepared)
    if (max_bits) { atomicMax(max_bits, __float_as_uint(s)); atomicMax(max_bits + 1, ~__float_as_uint(s)); }
  }
}

// query side of VertexSearch: Normalize (none_vectorstore.go:131-133), Lower (f16_vectorstore.go:136) and the
// decode half of Similarity (f16_quantization.go:35-45) -> the f32 vector the distance kernel sees.
// One wave per query.  The norm is the reference's strictly sequential f32 sum (pkg/distance Normalize: a scalar loop), so
// one lane walks it — but over an LDS copy the whole wave fetched with coalesced loads, and the element-wise half
// (divide, Lower, decode) runs on all 64 lanes.  (One THREAD per query, each striding through its own row, cost 330 us for
// 256 x 768 queries: 5 % of a batched FLAT search.)
// qn != null (dim <= PQ_CHUNK): also ||q_eff||^2 in AVX order (what query_norms_kernel computes), from the LDS copy of the vector just
// written — one launch less on the single-query path, where every launch is ~10 us of a ~100 us search.
template <int QUANT>
__global__ __launch_bounds__(64) void prep_queries_kernel(const float* __restrict__ raw, uint64_t nq, int dim, int normalize,
                                                          float* __restrict__ q_eff, float* __restrict__ qn, uint32_t* __restrict__ zero64) {
  if (zero64 && blockIdx.x == 0) zero64[threadIdx.x] = 0u;   // the search's work counter + traversal counters (256 bytes): one stream operation less per call
  __shared__ __attribute__((aligned(16))) float buf[PQ_CHUNK];
  __shared__ float s_norm;
  const uint64_t i = blockIdx.x;
  const int lane = threadIdx.x;
  const float* v = raw + i * (uint64_t)dim;
  float* out = q_eff + i * (uint64_t)dim;
  float norm = 0.f;
  bool zero = false;
  if (normalize) {
    for (int c0 = 0; c0 < dim; c0 += PQ_CHUNK) {
      const int n = min(PQ_CHUNK, dim - c0);
      for (int e = lane; e < n; e += 64) buf[e] = v[c0 + e];
      __syncthreads();
      if (lane == 0) norm = seq_sqsum(norm, buf, n);
      __syncthreads();
    }
    if (lane == 0) s_norm = norm;
    __syncthreads();
    norm = s_norm;
    zero = (norm == 0.f);
    norm = go_sqrt(norm);
  }
  for (int e = lane; e < dim; e += 64) {
    float x = v[e];
    if (normalize) x = zero ? 0.f : div_rn(x, norm);
    if constexpr (QUANT == Q_F8) x = __uint_as_float(f8bits_to_f32bits(f32bits_to_f8bits(__float_as_uint(x))));
    else if constexpr (QUANT != Q_NONE) x = f16bits_to_f32(f32bits_to_f16bits(__float_as_uint(x)));
    out[e] = x;
    if (qn) buf[e] = x;
  }
  if (qn) {   // the norm_a accumulator of avx.cpp:51-75 for this query: lanes 0 / 1 are the two halves of the 8-lane register
    __syncthreads();
    if (lane < 2) {
      const int half = lane, n8 = dim >> 3;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int t = 0;
      for (; t + 4 <= n8; t += 4) {
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(buf + 8 * t + 4 * half), r1 = *reinterpret_cast<const f32x4*>(buf + 8 * (t + 1) + 4 * half);
        const f32x4 r2 = *reinterpret_cast<const f32x4*>(buf + 8 * (t + 2) + 4 * half), r3 = *reinterpret_cast<const f32x4*>(buf + 8 * (t + 3) + 4 * half);
        f32x4 p = r0 * r0; acc = acc + p; p = r1 * r1; acc = acc + p; p = r2 * r2; acc = acc + p; p = r3 * r3; acc = acc + p;
      }
      for (; t < n8; t++) { const f32x4 r = *reinterpret_cast<const f32x4*>(buf + 8 * t + 4 * half); const f32x4 p = r * r; acc = acc + p; }
      float sq = pair_hsum(acc, half);
      for (int e = n8 * 8; e < dim; e++) { const float r = buf[e]; sq += r * r; }
      if (lane == 0) qn[i] = sq;
    }
  }
}
static __global__ void query_norms_kernel(const float* __restrict__ q_eff, uint64_t nq, int dim, float* __restrict__ qn);
// q_eff <- Normalize / Lower / decode of the raw queries; qn (may be null) <- their AVX-order squared norms
// zero64 (may be null): 64 words the first workgroup clears — the caller's counters, so that they need no memset of their own
template <int QUANT>
inline void launch_prep_queries(hipStream_t st, const float* raw, uint64_t nq, int dim, int normalize, float* q_eff, float* qn = nullptr, uint32_t* zero64 = nullptr) {
  if (!nq) return;
  const bool fused = qn && dim <= PQ_CHUNK;
  prep_queries_kernel<QUANT><<<(unsigned)nq, 64, 0, st>>>(raw, nq, dim, normalize, q_eff, fused ? qn : nullptr, zero64);
  if (qn && !fused) query_norms_kernel<<<(unsigned)((nq * 2 + 255) / 256), 256, 0, st>>>(q_eff, nq, dim, qn);
}
static __global__ void query_norms_kernel(const float* __restrict__ q_eff, uint64_t nq, int dim, float* __restrict__ qn) {
  uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  int half = threadIdx.x & 1;
  bool valid = pair < nq;
  float s = pair_sqnorm_f32(q_eff + (valid ? pair : 0) * (uint64_t)dim, dim, half);
  if (valid && half == 0) qn[pair] = s;
}


}  // namespace dev
}  // namespace coltt
