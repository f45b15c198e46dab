// prep.hpp — ingest-side kernels shared by the FLAT store and the HNSW index: Normalize + Lower of stored
// vectors, AVX-order row norms, and the query-side Normalize / Lower / decode.
#pragma once
#include "exact.hpp"

namespace coltt {
namespace dev {

// ---------------------------------------------------------------------------------------------------
// Normalize (edge/vectorstore.go:173-189) + Quantization.Lower (edge/f16_quantization.go:47-53 ...).
// One thread per vector: the reference's norm is ONE sequential f32 chain (norm += v[i]*v[i]).
// ---------------------------------------------------------------------------------------------------
template <int QUANT>
__global__ void prep_rows_kernel(const float* __restrict__ raw, uint64_t n, int dim, int normalize,
                                 const uint32_t* __restrict__ slots, uint64_t slot_base, uint8_t* __restrict__ rows,
                                 size_t stride) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* v = raw + i * (uint64_t)dim;
  uint64_t slot = slots ? slots[i] : slot_base + i;
  uint8_t* out = rows + slot * stride;
  float norm = 0.f;
  bool zero = false;
  if (normalize) {
    for (int e = 0; e < dim; e++) { float x = v[e]; norm += x * x; }
    zero = (norm == 0.f);
    norm = go_sqrt(norm);
  }
  for (int e = 0; e < dim; e++) {
    float x = v[e];
    if (normalize) x = zero ? 0.f : div_rn(x, norm);
    if constexpr (QUANT == Q_NONE) reinterpret_cast<float*>(out)[e] = x;
    else if constexpr (QUANT == Q_F8) out[e] = (uint8_t)f32bits_to_f8bits(__float_as_uint(x));
    else reinterpret_cast<unsigned short*>(out)[e] = (unsigned short)f32bits_to_f16bits(__float_as_uint(x));
  }
  for (size_t b = (size_t)dim * elem_bytes<QUANT>(); b < stride; b++) out[b] = 0;
}

// ||row||^2 in AVX order, one lane pair per row.
template <int QUANT>
__global__ void row_norms_kernel(const uint8_t* __restrict__ rows, size_t stride, const uint32_t* __restrict__ slots,
                                 uint64_t slot_base, uint64_t n, int dim, float* __restrict__ norms) {
  uint64_t pair = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 1;
  int half = threadIdx.x & 1;
  bool valid = pair < n;
  uint64_t slot = valid ? (slots ? slots[pair] : slot_base + pair) : (slots ? slots[0] : slot_base);
  float s = pair_sqnorm<QUANT>(rows + slot * stride, dim, half);
  if (valid && half == 0) norms[slot] = s;
}

// query side of VertexSearch: Normalize (none_vectorstore.go:131-133), Lower (f16_vectorstore.go:136) and the
// decode half of Similarity (f16_quantization.go:35-45) -> the f32 vector the distance kernel sees.
// One wave per query.  The norm is the reference's strictly sequential f32 sum (pkg/distance Normalize: a scalar loop), so
// one lane walks it — but over an LDS copy the whole wave fetched with coalesced loads, and the element-wise half
// (divide, Lower, decode) runs on all 64 lanes.  (One THREAD per query, each striding through its own row, cost 330 us for
// 256 x 768 queries: 5 % of a batched FLAT search.)
constexpr int PQ_CHUNK = 4096;
template <int QUANT>
__global__ __launch_bounds__(64) void prep_queries_kernel(const float* __restrict__ raw, uint64_t nq, int dim, int normalize,
                                                          float* __restrict__ q_eff) {
  __shared__ float buf[PQ_CHUNK];
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
      if (lane == 0) for (int e = 0; e < n; e++) { float x = buf[e]; norm += x * x; }
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
  }
}
template <int QUANT>
inline void launch_prep_queries(hipStream_t st, const float* raw, uint64_t nq, int dim, int normalize, float* q_eff) {
  if (nq) prep_queries_kernel<QUANT><<<(unsigned)nq, 64, 0, st>>>(raw, nq, dim, normalize, q_eff);
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
