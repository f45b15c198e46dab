// exact.hpp — gfx950 device code for the reference-order ("exact") arithmetic.
//
// The reference's AVX kernels (pkg/distance/simd/cpp/avx.cpp:15-32,51-75; shipped as
// pkg/distance/simd/avx/AVX_amd64.s) keep ONE 8-lane f32 accumulator per quantity: element i lands in
// partial sum (i mod 8), partial sums grow in increasing i with a separate multiply and add (no FMA),
// and are combined as ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)); a scalar tail follows.  To be bit-identical
// a GPU kernel has to keep that order.  Mapping used everywhere in this library:
//
//     one ROW (stored vector) is owned by a PAIR of adjacent lanes (2p, 2p+1) of a wave64:
//       lane 2p   ("half 0") owns residues 0..3, lane 2p+1 ("half 1") owns residues 4..7;
//       step t consumes elements [8t, 8t+8): half h loads the 4 elements 8t+4h .. 8t+4h+3
//       (16 B of f32, 8 B of f16 codes, 4 B of f8 codes) — a pair reads 32 contiguous bytes of f32,
//       32 rows are in flight per wave, and four 32-bit accumulators per lane form the 8-lane AVX register.
//
// The query is staged as f32 in LDS; both halves read it with broadcast ds_read_b128.
// Everything here is compiled with -ffp-contract=off; sqrt and divide go through f64 exactly as the Go
// code does (float32(math.Sqrt(float64(x))), pkg/distance/simd/avx/AVX_amd64.go:31,51), which is also
// correctly rounded for f32 regardless of compiler flags.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace coltt {
namespace dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

enum { Q_NONE = 0, Q_F16 = 1, Q_F8 = 2, Q_BF16 = 3 };
enum { M_COS = 0, M_L2 = 1 };

// ---- codecs (pkg/compresshelper) -----------------------------------------------------------------
// f32bitsToF16bits (float16.go:276-321) == f32bitsToBF16bits (bf16.go:272-317): integer restatement.
__device__ __forceinline__ uint32_t f32bits_to_f16bits(uint32_t u32) {
  uint32_t sign = u32 & 0x80000000u, exp = u32 & 0x7f800000u, coef = u32 & 0x007fffffu;
  if (exp == 0x7f800000u) {
    uint32_t nanBit = coef != 0 ? 0x0200u : 0u;
    return ((sign >> 16) | 0x7c00u | nanBit | (coef >> 13)) & 0xffffu;
  }
  uint32_t halfSign = sign >> 16;
  int32_t halfExp = (int32_t)(exp >> 23) - 127 + 15;
  if (halfExp >= 0x1f) return halfSign | 0x7c00u;
  if (halfExp <= 0) {
    if (14 - halfExp > 24) return halfSign;
    uint32_t c = coef | 0x00800000u;
    uint32_t halfCoef = c >> (uint32_t)(14 - halfExp);
    uint32_t roundBit = 1u << (uint32_t)(13 - halfExp);
    if ((c & roundBit) != 0 && (c & (3 * roundBit - 1)) != 0) halfCoef++;
    return (halfSign | halfCoef) & 0xffffu;
  }
  uint32_t uHalfExp = (uint32_t)halfExp << 10;
  uint32_t halfCoef = coef >> 13;
  if ((coef & 0x1000u) != 0 && (coef & 0x2fffu) != 0) return ((halfSign | uHalfExp | halfCoef) + 1) & 0xffffu;
  return halfSign | uHalfExp | halfCoef;
}
// f32bitsToF8bits (float8.go:270-313): same algorithm, sign taken from bit 23, every return cut to 8 bits.
__device__ __forceinline__ uint32_t f32bits_to_f8bits(uint32_t u32) {
  uint32_t sign = u32 & 0x800000u, exp = u32 & 0x7f800000u, coef = u32 & 0x007fffffu;
  if (exp == 0x7f800000u) {
    uint32_t nanBit = coef != 0 ? 0x0200u : 0u;
    return ((sign >> 8) | 0x7cu | nanBit | (coef >> 13)) & 0xffu;
  }
  uint32_t halfSign = sign >> 8;
  int32_t halfExp = (int32_t)(exp >> 23) - 127 + 15;
  if (halfExp >= 0x1f) return (halfSign | 0x7cu) & 0xffu;
  if (halfExp <= 0) {
    if (14 - halfExp > 24) return halfSign & 0xffu;
    uint32_t c = coef | 0x00800000u;
    uint32_t halfCoef = c >> (uint32_t)(14 - halfExp);
    uint32_t roundBit = 1u << (uint32_t)(13 - halfExp);
    if ((c & roundBit) != 0 && (c & (3 * roundBit - 1)) != 0) halfCoef++;
    return (halfSign | halfCoef) & 0xffu;
  }
  uint32_t uHalfExp = (uint32_t)halfExp << 10;
  uint32_t halfCoef = coef >> 13;
  if ((coef & 0x1000u) != 0 && (coef & 0x2fffu) != 0) return ((halfSign | uHalfExp | halfCoef) + 1) & 0xffu;
  return (halfSign | uHalfExp | halfCoef) & 0xffu;
}
// F8bitsToF32bits (float8.go:233-266): the exponent field is always read as 0, so only bits 0,1,7 of the
// code matter; the subnormal-normalisation loop yields 2^-24, 2^-23, 1.5*2^-23; "sign" lands on bit 15.
__device__ __forceinline__ uint32_t f8bits_to_f32bits(uint32_t in) {
  uint32_t m = in & 3u, s = (in & 0x80u) << 8;
  if (m == 0) return s;
  uint32_t base = 0x33800000u + ((m >> 1) ? (0x00800000u + ((m & 1u) << 22)) : 0u);
  return base | s;
}
// f16bitsToF32bits (float16.go:237-272) == BF16bitsToF32bits (bf16.go:233-268): IEEE binary16 -> binary32,
// exactly what v_cvt_f32_f16 computes (quiet-NaN payload included); checked over all 65 536 codes on the GPU.
__device__ __forceinline__ float f16bits_to_f32(uint32_t h) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)h);
}

template <int QUANT> __device__ __forceinline__ constexpr int elem_bytes() {
  return QUANT == Q_NONE ? 4 : (QUANT == Q_F8 ? 1 : 2);
}

// 4 consecutive decoded elements starting at element index e (e % 4 == 0) of a stored row.
template <int QUANT> __device__ __forceinline__ f32x4 load4(const uint8_t* __restrict__ row, int e) {
  if constexpr (QUANT == Q_NONE) {
    return *reinterpret_cast<const f32x4*>(row + (size_t)e * 4);
  } else if constexpr (QUANT == Q_F8) {
    uint32_t w = *reinterpret_cast<const uint32_t*>(row + e);
    f32x4 r;
    r.x = __uint_as_float(f8bits_to_f32bits(w & 0xffu));
    r.y = __uint_as_float(f8bits_to_f32bits((w >> 8) & 0xffu));
    r.z = __uint_as_float(f8bits_to_f32bits((w >> 16) & 0xffu));
    r.w = __uint_as_float(f8bits_to_f32bits(w >> 24));
    return r;
  } else {
    f16x4 h = *reinterpret_cast<const f16x4*>(row + (size_t)e * 2);
    return __builtin_convertvector(h, f32x4);
  }
}
// Raw (undecoded) 4-element chunk: keeping the RAW bits in the software-pipeline registers and decoding at the point of
// use is what lets the loads stay in flight — decoding at load time would wait for every load as it is issued.
template <int QUANT> struct Raw4 { typedef f32x4 type; };
template <> struct Raw4<Q_F16> { typedef f16x4 type; };
template <> struct Raw4<Q_BF16> { typedef f16x4 type; };
template <> struct Raw4<Q_F8> { typedef uint32_t type; };
template <int QUANT> __device__ __forceinline__ typename Raw4<QUANT>::type load_raw4(const uint8_t* __restrict__ row, int e) {
  return *reinterpret_cast<const typename Raw4<QUANT>::type*>(row + (size_t)e * elem_bytes<QUANT>());
}
template <int QUANT> __device__ __forceinline__ f32x4 decode4(typename Raw4<QUANT>::type v) {
  if constexpr (QUANT == Q_NONE) return v;
  else if constexpr (QUANT == Q_F8) {
    f32x4 r;
    r.x = __uint_as_float(f8bits_to_f32bits(v & 0xffu));
    r.y = __uint_as_float(f8bits_to_f32bits((v >> 8) & 0xffu));
    r.z = __uint_as_float(f8bits_to_f32bits((v >> 16) & 0xffu));
    r.w = __uint_as_float(f8bits_to_f32bits(v >> 24));
    return r;
  } else return __builtin_convertvector(v, f32x4);
}

template <int QUANT> __device__ __forceinline__ float load1(const uint8_t* __restrict__ row, int e) {
  if constexpr (QUANT == Q_NONE) return *reinterpret_cast<const float*>(row + (size_t)e * 4);
  else if constexpr (QUANT == Q_F8) return __uint_as_float(f8bits_to_f32bits(row[e]));
  else return f16bits_to_f32(*reinterpret_cast<const unsigned short*>(row + (size_t)e * 2));
}

// ---- epilogues -------------------------------------------------------------------------------------
// gomath.Sqrt / the Go wrappers: float32(math.Sqrt(float64(x)))
__device__ __forceinline__ float go_sqrt(float x) { return (float)sqrt((double)x); }
// correctly rounded f32 divide (f64 divide then round: innocuous double rounding, 53 >= 2*24+2)
__device__ __forceinline__ float div_rn(float a, float b) { return (float)((double)a / (double)b); }
// Cosine.Distance (space.go:93-95) over AVX_amd64.go:46-52: |1 - dot/float32(sqrt(float64(na*nb)))|
__device__ __forceinline__ float cos_epilogue(float dot, float na, float nb) {
  float nsq = na * nb;
  float d = 1.0f - div_rn(dot, go_sqrt(nsq));
  return fabsf(d);
}
// native_impl.go:41-52: 1 - dot/(Sqrt(na)*Sqrt(nb))
__device__ __forceinline__ float cos_epilogue_native(float dot, float na, float nb) {
  float den = go_sqrt(na) * go_sqrt(nb);
  return fabsf(1.0f - div_rn(dot, den));
}

__device__ __forceinline__ float xor1(float v) { return __shfl_xor(v, 1, 64); }

// hadd,hadd,lane0+lane4 (avx.cpp:4-8) for an accumulator split over a lane pair: each half first forms
// (r0+r1)+(r2+r3) of its own four residues, then low half + high half.  Result valid in BOTH lanes.
__device__ __forceinline__ float pair_hsum(f32x4 a, int half) {
  float s = (a.x + a.y) + (a.z + a.w);
  float o = xor1(s);
  return half == 0 ? (s + o) : (o + s);
}

// ||row||^2 in AVX order for the pair-owned row (the norm_b accumulator of avx.cpp:51-75, which depends
// on b only — precomputed once at upsert, bit-identical to recomputing it per pair as the reference does).
template <int QUANT>
__device__ __forceinline__ float pair_sqnorm(const uint8_t* __restrict__ row, int dim, int half) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int n8 = dim >> 3;
  for (int t = 0; t < n8; t++) {
    f32x4 r = load4<QUANT>(row, 8 * t + 4 * half);
    f32x4 p = r * r;
    acc = acc + p;
  }
  float s = pair_hsum(acc, half);
  for (int e = n8 * 8; e < dim; e++) { float r = load1<QUANT>(row, e); s += r * r; }
  return s;
}
// same for an f32 vector in LDS/global (the query: norm_a)
__device__ __forceinline__ float pair_sqnorm_f32(const float* __restrict__ v, int dim, int half) {
  return pair_sqnorm<Q_NONE>(reinterpret_cast<const uint8_t*>(v), dim, half);
}

// Distance(query, row) for the pair-owned row; q = f32 query in LDS.  Valid in both lanes of the pair.
// Row loads are software-pipelined in bursts of U steps: the next U raw chunks are requested back to back while the
// previous U are decoded and consumed (U..2U loads per lane in flight).  The traversal kernels run 4 waves per CU, so
// bytes in flight per wave are what buys HBM bandwidth (Little's law); U is tuned per element size.
template <int METRIC, int QUANT, int U = 8>
__device__ __forceinline__ float pair_distance(const uint8_t* __restrict__ row, const float* __restrict__ q, int dim,
                                               float qnorm, float rnorm, int half) {
  typedef typename Raw4<QUANT>::type raw_t;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int n8 = dim >> 3;
  const int nb = n8 / U;
  raw_t cur[U], nxt[U];
  if (nb > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = load_raw4<QUANT>(row, 8 * u + 4 * half);
  }
  for (int b = 0; b < nb; b++) {
    if (b + 1 < nb) {
#pragma unroll
      for (int u = 0; u < U; u++) nxt[u] = load_raw4<QUANT>(row, 8 * ((b + 1) * U + u) + 4 * half);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      f32x4 qq = *reinterpret_cast<const f32x4*>(q + 8 * (b * U + u) + 4 * half);
      f32x4 r = decode4<QUANT>(cur[u]);
      if constexpr (METRIC == M_COS) { f32x4 p = qq * r; acc = acc + p; }
      else { f32x4 d = qq - r; f32x4 p = d * d; acc = acc + p; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
  }
  for (int t = nb * U; t < n8; t++) {
    f32x4 r = decode4<QUANT>(load_raw4<QUANT>(row, 8 * t + 4 * half));
    f32x4 qq = *reinterpret_cast<const f32x4*>(q + 8 * t + 4 * half);
    if constexpr (METRIC == M_COS) { f32x4 p = qq * r; acc = acc + p; }
    else { f32x4 d = qq - r; f32x4 p = d * d; acc = acc + p; }
  }
  float s = pair_hsum(acc, half);
  for (int e = n8 * 8; e < dim; e++) {  // scalar tail (avx.cpp:28-31,68-72)
    float r = load1<QUANT>(row, e);
    if constexpr (METRIC == M_COS) s += q[e] * r;
    else { float d = q[e] - r; s += d * d; }
  }
  if constexpr (METRIC == M_COS) return cos_epilogue(s, qnorm, rnorm);
  else return go_sqrt(s);
}

// ---- misc --------------------------------------------------------------------------------------------
// sharding.ShardVertex (pkg/sharding/shard.go:34-41): FNV-1a-64 over the 8 LE bytes of the id
__device__ __host__ __forceinline__ uint64_t shard_vertex(uint64_t x, uint64_t c) {
  uint64_t h = 14695981039346656037ull;
  for (int i = 0; i < 8; i++) { h ^= (x >> (8 * i)) & 0xffull; h *= 1099511628211ull; }
  return h % c;
}

// total order on f32 scores used for top-k keys: non-negative floats order as their bit patterns; the
// general map below also orders negatives and puts NaN (positive) last.
__device__ __host__ __forceinline__ uint32_t score_key(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __host__ __forceinline__ float key_score(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __builtin_bit_cast(float, u);
}

}  // namespace dev
}  // namespace coltt
