// kernels.hip — the reference's leaf kernels exposed one-to-one through the C-ABI:
// pkg/distance (AVX / SSE / native orders), Normalize, the three codecs, FNV sharding and the
// pkg/distancepq FMA / popcount kernels.  Host buffers in, host buffers out (staged through HBM).
#include "common.hpp"
#include "exact.hpp"
#include "prep.hpp"

using namespace coltt;
using namespace coltt::dev;

namespace {

// ---- pkg/distance, AVX order: the same pair-owned code path the scan and HNSW kernels use -------------
template <int METRIC>
__global__ __launch_bounds__(64) void pairs_avx_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                      int dim, float* __restrict__ out) {
  size_t pair = (size_t)blockIdx.x * 32 + (threadIdx.x >> 1);
  int half = threadIdx.x & 1;
  bool valid = pair < n;
  size_t i = valid ? pair : 0;
  const float* ai = a + i * dim;
  const float* bi = b + i * dim;
  float na = 0.f, nb = 0.f;
  if constexpr (METRIC == M_COS) { na = pair_sqnorm_f32(ai, dim, half); nb = pair_sqnorm_f32(bi, dim, half); }
  float d = pair_distance<METRIC, Q_NONE, 4>(reinterpret_cast<const uint8_t*>(bi), ai, dim, na, nb, half);
  if (valid && half == 0) out[pair] = d;
}

// ---- SSE (4-lane, l0+l1+l2+l3; sse.cpp:3-82) and native (scalar; native_impl.go:23-52) orders: one thread
// per pair, sequential emulation.  Not a hot path — the reference only selects these on non-AVX hosts
// (space.go:40-49); kept so that every SpaceImpl has a device twin.
template <int LANES>
__device__ float seq_l2sq(const float* a, const float* b, int dim) {
  float acc[LANES];
  for (int j = 0; j < LANES; j++) acc[j] = 0.f;
  int nl = (dim / LANES) * LANES;
  for (int i = 0; i < nl; i += LANES)
    for (int j = 0; j < LANES; j++) { float d = a[i + j] - b[i + j]; float m = d * d; acc[j] = acc[j] + m; }
  float r;
  if (LANES == 8) r = ((acc[0] + acc[1 % LANES]) + (acc[2 % LANES] + acc[3 % LANES])) + ((acc[4 % LANES] + acc[5 % LANES]) + (acc[6 % LANES] + acc[7 % LANES]));
  else if (LANES == 4) r = ((acc[0] + acc[1 % LANES]) + acc[2 % LANES]) + acc[3 % LANES];
  else r = acc[0];
  for (int i = nl; i < dim; i++) { float d = a[i] - b[i]; r += d * d; }
  return r;
}
// manhattan_distance (avx.cpp:34-49, sse.cpp:35-53): sqrt(d * d) per lane in the vector part (the rounded square, a correctly rounded
// sqrt — not |d|), abs(d) in the scalar tail; native_impl.go:33-40: gomath.Abs sequentially
template <int LANES>
__device__ float seq_l1(const float* a, const float* b, int dim) {
  float acc[LANES];
  for (int j = 0; j < LANES; j++) acc[j] = 0.f;
  int nl = LANES == 1 ? 0 : (dim / LANES) * LANES;
  for (int i = 0; i < nl; i += LANES)
    for (int j = 0; j < LANES; j++) { float d = a[i + j] - b[i + j]; float m = d * d; acc[j] = acc[j] + go_sqrt(m); }
  float r;
  if (LANES == 8) r = ((acc[0] + acc[1 % LANES]) + (acc[2 % LANES] + acc[3 % LANES])) + ((acc[4 % LANES] + acc[5 % LANES]) + (acc[6 % LANES] + acc[7 % LANES]));
  else if (LANES == 4) r = ((acc[0] + acc[1 % LANES]) + acc[2 % LANES]) + acc[3 % LANES];
  else r = 0.f;
  for (int i = nl; i < dim; i++) { float d = a[i] - b[i]; r += fabsf(d); }
  return r;
}
template <int LANES>
__device__ void seq_cos(const float* a, const float* b, int dim, float& dot, float& na, float& nb) {
  float d[LANES], x[LANES], y[LANES];
  for (int j = 0; j < LANES; j++) d[j] = x[j] = y[j] = 0.f;
  int nl = (dim / LANES) * LANES;
  for (int i = 0; i < nl; i += LANES)
    for (int j = 0; j < LANES; j++) {
      float v1 = a[i + j], v2 = b[i + j];
      float p = v1 * v2; d[j] = d[j] + p;
      float q = v1 * v1; x[j] = x[j] + q;
      float r = v2 * v2; y[j] = y[j] + r;
    }
  if (LANES == 4) {
    dot = ((d[0] + d[1 % LANES]) + d[2 % LANES]) + d[3 % LANES];
    na = ((x[0] + x[1 % LANES]) + x[2 % LANES]) + x[3 % LANES];
    nb = ((y[0] + y[1 % LANES]) + y[2 % LANES]) + y[3 % LANES];
  } else { dot = d[0]; na = x[0]; nb = y[0]; }
  for (int i = nl; i < dim; i++) { dot += a[i] * b[i]; na += a[i] * a[i]; nb += b[i] * b[i]; }
}
__global__ void pairs_seq_kernel(int metric, int order, const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                 int dim, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* ai = a + i * dim;
  const float* bi = b + i * dim;
  if (metric == COLTT_MANHATTAN) {
    out[i] = order == 0 ? seq_l1<8>(ai, bi, dim) : (order == 1 ? seq_l1<4>(ai, bi, dim) : seq_l1<1>(ai, bi, dim));
  } else if (metric == COLTT_EUCLIDEAN) {
    float s = order == 1 ? seq_l2sq<4>(ai, bi, dim) : seq_l2sq<1>(ai, bi, dim);
    out[i] = go_sqrt(s);
  } else {
    float dot, na, nb;
    if (order == 1) { seq_cos<4>(ai, bi, dim, dot, na, nb); out[i] = cos_epilogue(dot, na, nb); }
    else { seq_cos<1>(ai, bi, dim, dot, na, nb); out[i] = cos_epilogue_native(dot, na, nb); }
  }
}

// ---- codecs ---------------------------------------------------------------------------------------------
__global__ void lower_kernel(int quant, const float* __restrict__ in, size_t n, uint8_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t u = __float_as_uint(in[i]);
  if (quant == COLTT_Q_NONE) reinterpret_cast<float*>(out)[i] = in[i];
  else if (quant == COLTT_Q_F8) out[i] = (uint8_t)f32bits_to_f8bits(u);
  else reinterpret_cast<unsigned short*>(out)[i] = (unsigned short)f32bits_to_f16bits(u);
}
__global__ void raise_kernel(int quant, const uint8_t* __restrict__ in, size_t n, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (quant == COLTT_Q_NONE) out[i] = reinterpret_cast<const float*>(in)[i];
  else if (quant == COLTT_Q_F8) out[i] = __uint_as_float(f8bits_to_f32bits(in[i]));
  else out[i] = f16bits_to_f32(reinterpret_cast<const unsigned short*>(in)[i]);
}
__global__ void shard_kernel(const uint64_t* __restrict__ ids, size_t n, uint64_t c, uint64_t* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = shard_vertex(ids[i], c);
}

// ---- pkg/distancepq: asm.Dot / asm.SquaredEuclideanDistance (dot.s:7-55, euclidean.s:7-65) ------------
// Four 8-lane FMA accumulators over 32-float blocks => 32 lanes own one row, lane j owns element j of every
// block; scalar-FMA tail in "lane 0 of X4"; reduce ((Y0+Y1)+Y2)+Y3 -> lo128+hi128 -> +tail -> hadd, hadd.
template <int KIND>
__global__ __launch_bounds__(64) void pq_float_kernel(const float* __restrict__ query, const float* __restrict__ rows, size_t n,
                                                     int dim, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, sub = lane & 31;
  size_t r = (size_t)blockIdx.x * 2 + (lane >> 5);
  bool valid = r < n;
  const float* y = rows + (valid ? r : 0) * (size_t)dim;
  float acc = 0.f;
  int nblk = dim / 32;
  for (int b = 0; b < nblk; b++) {
    float xv = query[b * 32 + sub], yv = y[b * 32 + sub];
    if (KIND == 1) { float d = xv - yv; acc = fmaf(d, d, acc); }
    else acc = fmaf(xv, yv, acc);
  }
  float tail = 0.f;
  for (int e = nblk * 32; e < dim; e++) {
    float xv = query[e], yv = y[e];
    if (KIND == 1) { float d = xv - yv; tail = fmaf(d, d, tail); }
    else tail = fmaf(xv, yv, tail);
  }
  // lane sub = 8*reg + j
  float a1 = __shfl(acc, (lane & 32) | ((sub & 7) + 8), 64);
  float a2 = __shfl(acc, (lane & 32) | ((sub & 7) + 16), 64);
  float a3 = __shfl(acc, (lane & 32) | ((sub & 7) + 24), 64);
  float a0 = __shfl(acc, (lane & 32) | (sub & 7), 64);
  float s = ((a0 + a1) + a2) + a3;                       // valid for every j = sub & 7
  float hi = __shfl(s, (lane & 32) | (((sub & 3) + 4)), 64);
  float lo = __shfl(s, (lane & 32) | (sub & 3), 64);
  float t = lo + hi;                                     // t[j], j = sub & 3
  if ((sub & 3) == 0) t = t + tail; else t = t + 0.0f;   // VADDPS X0, X4 with X4 = {tail,0,0,0}
  float t0 = __shfl(t, (lane & 32) | 0, 64), t1 = __shfl(t, (lane & 32) | 1, 64);
  float t2 = __shfl(t, (lane & 32) | 2, 64), t3 = __shfl(t, (lane & 32) | 3, 64);
  float res = (t0 + t1) + (t2 + t3);
  if (KIND == 2) res = 1.0f - res;   // cosineDistance (distance.go:40-42)
  if (KIND == 3) res = -res;         // dotProductDistance (distance.go:36-38)
  if (valid && sub == 0) out[r] = res;
}
// hammingDistance / jaccardDistance (distance.go:62-84)
__global__ void pq_bit_kernel(int kind, const uint64_t* __restrict__ query, const uint64_t* __restrict__ rows, size_t n,
                              int words, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t* y = rows + i * (size_t)words;
  if (kind == 0) {
    long long d = 0;
    for (int w = 0; w < words; w++) d += __popcll(query[w] ^ y[w]);
    out[i] = (float)d;
  } else {
    long long in = 0, un = 0;
    for (int w = 0; w < words; w++) { in += __popcll(query[w] & y[w]); un += __popcll(query[w] | y[w]); }
    out[i] = un == 0 ? 0.f : 1.0f - div_rn((float)in, (float)un);
  }
}

struct Stage {  // scratch device buffers for one stateless call
  std::vector<void*> ptrs;
  ~Stage() { for (void* p : ptrs) (void)hipFree(p); }
  int alloc(void** p, size_t bytes) {
    COLTT_HIP(hipMalloc(p, bytes ? bytes : 16));
    ptrs.push_back(*p);
    return COLTT_OK;
  }
};

}  // namespace

extern "C" {

int coltt_distance_pairs(int metric, int order, const float* a, const float* b, size_t n, uint32_t dim, float* out) {
  if (n == 0) return COLTT_OK;
  if (!a || !b || !out || dim == 0) return fail(COLTT_E_INVALID, "distance_pairs: NULL/empty input");
  if (metric != COLTT_COSINE && metric != COLTT_EUCLIDEAN && metric != COLTT_MANHATTAN) return fail(COLTT_E_INVALID, "distance_pairs: bad metric");
  if (order < 0 || order > 2) return fail(COLTT_E_INVALID, "distance_pairs: order must be 0 (avx), 1 (sse) or 2 (native)");
  COLTT_DEVICE(-1);
  Stage st; float *da, *db, *dout;
  size_t bytes = n * dim * 4;
  // rows are padded by 16 floats so the 16-byte vector loads of the last row stay in bounds
  COLTT_TRY(st.alloc((void**)&da, bytes + 64)); COLTT_TRY(st.alloc((void**)&db, bytes + 64)); COLTT_TRY(st.alloc((void**)&dout, n * 4));
  COLTT_HIP(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
  COLTT_HIP(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
  if (metric == COLTT_MANHATTAN) {   // a leaf of distance.SpaceImpl with no caller in the reference: the sequential emulation serves all three orders
    pairs_seq_kernel<<<ceil_div(n, 64), 64>>>(metric, order, da, db, n, (int)dim, dout);
  } else if (order == 0 && dim % 4 == 0) {
    if (metric == COLTT_COSINE) pairs_avx_kernel<M_COS><<<ceil_div(n, 32), 64>>>(da, db, n, (int)dim, dout);
    else pairs_avx_kernel<M_L2><<<ceil_div(n, 32), 64>>>(da, db, n, (int)dim, dout);
  } else if (order == 0) {
    return fail(COLTT_E_UNSUPPORTED, "distance_pairs: AVX order over packed rows needs dim %% 4 == 0 (stores pad rows to 16 B)");
  } else {
    pairs_seq_kernel<<<ceil_div(n, 64), 64>>>(metric, order, da, db, n, (int)dim, dout);
  }
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_normalize(const float* in, size_t n, uint32_t dim, float* out) {
  if (n == 0) return COLTT_OK;
  if (!in || !out || dim == 0) return fail(COLTT_E_INVALID, "normalize: NULL/empty input");
  COLTT_DEVICE(-1);
  Stage st; float *di, *dout;
  COLTT_TRY(st.alloc((void**)&di, n * dim * 4)); COLTT_TRY(st.alloc((void**)&dout, n * dim * 4));
  COLTT_HIP(hipMemcpy(di, in, n * dim * 4, hipMemcpyHostToDevice));
  launch_prep_queries<Q_NONE>(nullptr, di, n, (int)dim, 1, dout);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out, dout, n * dim * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

// The same arithmetic on the HOST, for the one-vector-per-RPC callers (vectorindex.Normalize in the Go layer): no device, no stream, no
// allocation, cannot fail for valid pointers — the reference's Normalize is infallible too.  Sequential f32 sum (mul, then add: the
// library is built -ffp-contract=off and without fast-math, so neither FMA nor a re-associated reduction can appear), sqrt through
// float64, one IEEE divide per element; a zero vector comes back as zeros (metadata.go:107-123).
int coltt_normalize_host(const float* in, uint32_t dim, float* out) {
  if (dim == 0) return COLTT_OK;
  if (!in || !out) return fail(COLTT_E_INVALID, "normalize_host: NULL input");
  volatile float norm = 0.f;   // volatile: keeps the reduction scalar and in source order whatever the host vectoriser is told
  for (uint32_t i = 0; i < dim; i++) { const float p = in[i] * in[i]; norm = norm + p; }
  const float nsq = norm;
  if (nsq == 0.f) { for (uint32_t i = 0; i < dim; i++) out[i] = 0.f; return COLTT_OK; }
  const float nr = (float)std::sqrt((double)nsq);
  for (uint32_t i = 0; i < dim; i++) out[i] = in[i] / nr;
  return COLTT_OK;
}

int coltt_quant_lower(int quant, const float* in, size_t n, void* out_codes) {
  if (n == 0) return COLTT_OK;
  if (!in || !out_codes) return fail(COLTT_E_INVALID, "quant_lower: NULL input");
  if (quant < COLTT_Q_NONE || quant > COLTT_Q_BF16) return fail(COLTT_E_UNSUPPORTED, "not support quantization type");
  COLTT_DEVICE(-1);
  Stage st; float* di; uint8_t* dout;
  size_t ob = n * quant_bytes(quant);
  COLTT_TRY(st.alloc((void**)&di, n * 4)); COLTT_TRY(st.alloc((void**)&dout, ob));
  COLTT_HIP(hipMemcpy(di, in, n * 4, hipMemcpyHostToDevice));
  lower_kernel<<<ceil_div(n, 256), 256>>>(quant, di, n, dout);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out_codes, dout, ob, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_quant_raise(int quant, const void* codes, size_t n, float* out) {
  if (n == 0) return COLTT_OK;
  if (!codes || !out) return fail(COLTT_E_INVALID, "quant_raise: NULL input");
  if (quant < COLTT_Q_NONE || quant > COLTT_Q_BF16) return fail(COLTT_E_UNSUPPORTED, "not support quantization type");
  COLTT_DEVICE(-1);
  Stage st; uint8_t* di; float* dout;
  size_t ib = n * quant_bytes(quant);
  COLTT_TRY(st.alloc((void**)&di, ib)); COLTT_TRY(st.alloc((void**)&dout, n * 4));
  COLTT_HIP(hipMemcpy(di, codes, ib, hipMemcpyHostToDevice));
  raise_kernel<<<ceil_div(n, 256), 256>>>(quant, di, n, dout);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_shard_vertex(const uint64_t* ids, size_t n, uint64_t shard_count, uint64_t* out) {
  if (n == 0) return COLTT_OK;
  if (!ids || !out || shard_count == 0) return fail(COLTT_E_INVALID, "shard_vertex: NULL input or zero shard count");
  COLTT_DEVICE(-1);
  Stage st; uint64_t *di, *dout;
  COLTT_TRY(st.alloc((void**)&di, n * 8)); COLTT_TRY(st.alloc((void**)&dout, n * 8));
  COLTT_HIP(hipMemcpy(di, ids, n * 8, hipMemcpyHostToDevice));
  shard_kernel<<<ceil_div(n, 256), 256>>>(di, n, shard_count, dout);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_pq_float_scan(int kind, const float* query, const float* rows, size_t n, uint32_t dim, float* out) {
  if (n == 0) return COLTT_OK;
  if (!query || !rows || !out || dim == 0) return fail(COLTT_E_INVALID, "pq_float_scan: NULL/empty input");
  if (kind < 0 || kind > 3) return fail(COLTT_E_INVALID, "pq_float_scan: kind must be 0..3");
  COLTT_DEVICE(-1);
  Stage st; float *dq, *dr, *dout;
  COLTT_TRY(st.alloc((void**)&dq, dim * 4)); COLTT_TRY(st.alloc((void**)&dr, n * dim * 4)); COLTT_TRY(st.alloc((void**)&dout, n * 4));
  COLTT_HIP(hipMemcpy(dq, query, dim * 4, hipMemcpyHostToDevice));
  COLTT_HIP(hipMemcpy(dr, rows, n * dim * 4, hipMemcpyHostToDevice));
  uint32_t g = ceil_div(n, 2);
  if (kind == 0) pq_float_kernel<0><<<g, 64>>>(dq, dr, n, (int)dim, dout);
  else if (kind == 1) pq_float_kernel<1><<<g, 64>>>(dq, dr, n, (int)dim, dout);
  else if (kind == 2) pq_float_kernel<2><<<g, 64>>>(dq, dr, n, (int)dim, dout);
  else pq_float_kernel<3><<<g, 64>>>(dq, dr, n, (int)dim, dout);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_pq_bit_scan(int kind, const uint64_t* query, const uint64_t* rows, size_t n, uint32_t words, float* out) {
  if (n == 0) return COLTT_OK;
  if (!query || !rows || !out || words == 0) return fail(COLTT_E_INVALID, "pq_bit_scan: NULL/empty input");
  if (kind != 0 && kind != 1) return fail(COLTT_E_INVALID, "pq_bit_scan: kind must be 0 (hamming) or 1 (jaccard)");
  COLTT_DEVICE(-1);
  Stage st; uint64_t *dq, *dr; float* dout;
  COLTT_TRY(st.alloc((void**)&dq, words * 8)); COLTT_TRY(st.alloc((void**)&dr, n * words * 8)); COLTT_TRY(st.alloc((void**)&dout, n * 4));
  COLTT_HIP(hipMemcpy(dq, query, words * 8, hipMemcpyHostToDevice));
  COLTT_HIP(hipMemcpy(dr, rows, n * words * 8, hipMemcpyHostToDevice));
  pq_bit_kernel<<<ceil_div(n, 128), 128>>>(kind, dq, dr, n, (int)words, dout);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

}  // extern "C"
