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
#ifndef COLTT_PAIR_NT   // A/B knob: the non-temporal hint (row_ld below) on the PAIR-owned row loads too (FLAT exact scans, filtered scans, the PQ re-rank, the builder) — compile-time, every size.
                        // Measured SLOWER everywhere (GPU call AM: FLAT scans -40 %, C3 -12 %, PQ walk -7.5 %, build +20-40 %: these kernels re-read rows through L2), off
#define COLTT_PAIR_NT 0
#endif
template <int QUANT> __device__ __forceinline__ typename Raw4<QUANT>::type load_raw4(const uint8_t* __restrict__ row, int e) {
  typedef typename Raw4<QUANT>::type raw_t;
#if COLTT_PAIR_NT
  return __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(row + (size_t)e * elem_bytes<QUANT>()));
#else
  return *reinterpret_cast<const raw_t*>(row + (size_t)e * elem_bytes<QUANT>());
#endif
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

// partner lane's value (lane ^ 1) through DPP quad_perm [1,0,3,2]: a VALU move, not a trip through the LDS crossbar
__device__ __forceinline__ float xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

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

// 2-byte codes, wide loads.  With 8 B per lane a wave-wide load touches 32 rows x 16 B: the same number of vector-memory
// instructions (and cache-line lookups) as f32 rows for half the bytes, and the quantised walk ran at 0.52 of HBM peak where
// the f32 walk reaches 0.70.  Here a lane loads 16 B — the WHOLE 8-element group 2s + half — and the pair swaps the halves
// that belong to the other lane's residue chains through DPP (quad_perm [1,0,3,2]), so every chain still adds group 2s, then
// group 2s+1: same values, same order, half the load instructions.
__device__ __forceinline__ uint32_t dpp_swap1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
}
typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
// Stored rows are read ONCE per evaluation by ONE compute unit.  On a collection far larger than the caches (10 M x 768: 15-30 GB against 32 MB of L2 + 256 MB of
// MALL) a non-temporal hint on the row loads (`global_load_dwordx4 ... nt`) is worth +4.5 % on the headline walk (0.769 -> 0.804 of HBM peak) and +7 % on the
// operating-point walk, same box (GPU call AG, profiles/r06ag_nt_rows_ab.md; MI355X_MICROARCH.md "nt-weights"); on a SMALL collection whose rows are re-read out of
// L2 / MALL by the next queries (2 M x 768 x 2 B in eight shards, 10 000 queries per batch) the same hint costs 28 %.  So it is a template argument of the
// eight-lane kernels, chosen per launch from the size of the row array (hnsw.hip: rows_nt).  The product quantiser's neighbourhood blocks lose with it at any
// size measured (-7.5 %: hub vertices' blocks are re-read by other traversals): no hint there.
template <bool NT, class T> __device__ __forceinline__ T row_ld(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
typedef uint32_t u32x2e __attribute__((ext_vector_type(2)));
template <int METRIC>
__device__ __forceinline__ void h2_consume(f32x4& acc, u32x4e raw, const float* __restrict__ q, int s, int half) {
  // a = my residues of group 2s, b = my residues of group 2s + 1: the half I loaded myself or my partner's.  The swaps are
  // computed unconditionally, by all lanes: a DPP read under a partial EXEC mask would see its partner disabled.
  // (v_fma_mix_f32 — q * f16 + (-0.0), no separate convert — is bit-identical and 4 VALU shorter per step, but measured
  // 3 % slower: it is not packed and not full rate.)
  const uint32_t sx = dpp_swap1(raw.x), sy = dpp_swap1(raw.y), sz = dpp_swap1(raw.z), sw = dpp_swap1(raw.w);
  const u32x2e a = {half ? sz : raw.x, half ? sw : raw.y};
  const u32x2e b = {half ? raw.z : sx, half ? raw.w : sy};
  const f32x4 ra = __builtin_convertvector(__builtin_bit_cast(f16x4, a), f32x4);
  const f32x4 rb = __builtin_convertvector(__builtin_bit_cast(f16x4, b), f32x4);
  const f32x4 qa = *reinterpret_cast<const f32x4*>(q + 16 * s + 4 * half);
  const f32x4 qb = *reinterpret_cast<const f32x4*>(q + 16 * s + 8 + 4 * half);
  if constexpr (METRIC == M_COS) { f32x4 p = qa * ra; acc = acc + p; p = qb * rb; acc = acc + p; }
  else { f32x4 d = qa - ra; f32x4 p = d * d; acc = acc + p; d = qb - rb; p = d * d; acc = acc + p; }
}

// Software pipeline of a row walk: `n` steps in bursts of U — the next burst is requested back to back while the previous
// one is decoded and consumed (U..2U loads per lane in flight; bytes in flight per wave are what buys HBM bandwidth).  The
// remainder (n % U steps; the whole row when n < U, e.g. dim 128) goes four predicated steps at a time (a full predicated
// burst doubled the kernels' register count).
// Written as a macro over LOAD(step) / CONSUME(raw, step): lambdas and helper templates around the staging arrays cost 10 %
// (extra registers, lost overlap) in hipcc 7.2.
#define COLTT_BURST_WALK(U_, RAW_T, N_, LOAD, CONSUME)                                              \
  {                                                                                                 \
    const int n_ = (N_), nb_ = n_ / (U_);                                                           \
    RAW_T cur_[U_], nxt_[U_];                                                                       \
    if (nb_ > 0) {                                                                                  \
      _Pragma("unroll") for (int u = 0; u < (U_); u++) cur_[u] = LOAD(u);                           \
    }                                                                                               \
    for (int b_ = 0; b_ < nb_; b_++) {                                                              \
      if (b_ + 1 < nb_) {                                                                           \
        _Pragma("unroll") for (int u = 0; u < (U_); u++) nxt_[u] = LOAD((b_ + 1) * (U_) + u);       \
      }                                                                                             \
      _Pragma("unroll") for (int u = 0; u < (U_); u++) { CONSUME(cur_[u], b_ * (U_) + u); }         \
      _Pragma("unroll") for (int u = 0; u < (U_); u++) cur_[u] = nxt_[u];                           \
    }                                                                                               \
    for (int t_ = nb_ * (U_); t_ < n_; t_ += 4) {  /* remainder: 4 steps in flight at a time */   \
      RAW_T r_[4];                                                                                  \
      _Pragma("unroll") for (int u = 0; u < 4; u++) if (t_ + u < n_) r_[u] = LOAD(t_ + u);          \
      _Pragma("unroll") for (int u = 0; u < 4; u++) if (t_ + u < n_) { CONSUME(r_[u], t_ + u); }    \
    }                                                                                               \
  }

template <int METRIC, int QUANT>
__device__ __forceinline__ void narrow_consume(f32x4& acc, typename Raw4<QUANT>::type raw, const float* __restrict__ q, int t, int half) {
  f32x4 qq = *reinterpret_cast<const f32x4*>(q + 8 * t + 4 * half);
  f32x4 r = decode4<QUANT>(raw);
  if constexpr (METRIC == M_COS) { f32x4 p = qq * r; acc = acc + p; }
  else { f32x4 d = qq - r; f32x4 p = d * d; acc = acc + p; }
}

// Distance(query, row) for the pair-owned row; q = f32 query in LDS (or global).  Valid in both lanes of the pair.
template <int METRIC, int QUANT, int U = 8>
__device__ __forceinline__ float pair_distance(const uint8_t* __restrict__ row, const float* __restrict__ q, int dim,
                                               float qnorm, float rnorm, int half) {
  typedef typename Raw4<QUANT>::type raw_t;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int n8 = dim >> 3;
  if constexpr ((QUANT == Q_F16 || QUANT == Q_BF16) && U >= 2) {
    // wide walk over pairs of groups (every store keeps rows 16-byte aligned: row strides are rounded up to 16 B)
#define COLTT_LD_(S) (row_ld<COLTT_PAIR_NT != 0>(reinterpret_cast<const u32x4e*>(row + (size_t)(16 * (S) + 8 * half) * 2)))
#define COLTT_CS_(RAW, S) h2_consume<METRIC>(acc, RAW, q, S, half)
    COLTT_BURST_WALK(U / 2, u32x4e, n8 >> 1, COLTT_LD_, COLTT_CS_)
#undef COLTT_LD_
#undef COLTT_CS_
    if (n8 & 1) narrow_consume<METRIC, QUANT>(acc, load_raw4<QUANT>(row, 8 * (n8 - 1) + 4 * half), q, n8 - 1, half);  // odd group count
  } else {
#define COLTT_LD_(T) load_raw4<QUANT>(row, 8 * (T) + 4 * half)
#define COLTT_CS_(RAW, T) narrow_consume<METRIC, QUANT>(acc, RAW, q, T, half)
    COLTT_BURST_WALK(U, raw_t, n8, COLTT_LD_, COLTT_CS_)
#undef COLTT_LD_
#undef COLTT_CS_
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

// ---- the pair-owned core over LINE-TRANSPOSED rows (rows8.hpp: chunk r of every 128-byte line holds residue r's 4 (f32) / 8 (2-byte)
// consecutive AVX steps).  Half h of a pair owns residues 4h .. 4h+3 = the chunks 4h .. 4h+3 of a line: 64 CONTIGUOUS bytes per line and
// lane, the 16-byte chunk of residue r' in register c[r'].  Step t of the line is then (c[0][t], c[1][t], c[2][t], c[3][t]) — register
// renaming, no data movement — times the query's elements 8 * step + 4h .. + 3 (natural order in LDS, as for pair_distance): the same
// products added to the same four accumulators in the same order, hence the same bits.  Rows whose byte length is a multiple of 128 only
// (no scalar tail exists).  U = lines per burst.
struct Line4 { u32x4e c[4]; };
template <int METRIC, int QUANT>
__device__ __forceinline__ void r8_consume(f32x4& acc, const Line4& ln, const float* __restrict__ q, int L, int half) {
  if constexpr (QUANT == Q_NONE) {
    const f32x4 c0 = __builtin_bit_cast(f32x4, ln.c[0]), c1 = __builtin_bit_cast(f32x4, ln.c[1]), c2 = __builtin_bit_cast(f32x4, ln.c[2]), c3 = __builtin_bit_cast(f32x4, ln.c[3]);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      const f32x4 qq = *reinterpret_cast<const f32x4*>(q + 8 * (4 * L + t) + 4 * half);
      const f32x4 r = {c0[t], c1[t], c2[t], c3[t]};
      if constexpr (METRIC == M_COS) { const f32x4 p = qq * r; acc = acc + p; }
      else { const f32x4 d = qq - r; const f32x4 p = d * d; acc = acc + p; }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const f32x4 qq = *reinterpret_cast<const f32x4*>(q + 8 * (8 * L + t) + 4 * half);
      const int wd = t >> 1, sh = (t & 1) * 16;
      const f32x4 r = {f16bits_to_f32((ln.c[0][wd] >> sh) & 0xffffu), f16bits_to_f32((ln.c[1][wd] >> sh) & 0xffffu),
                       f16bits_to_f32((ln.c[2][wd] >> sh) & 0xffffu), f16bits_to_f32((ln.c[3][wd] >> sh) & 0xffffu)};
      if constexpr (METRIC == M_COS) { const f32x4 p = qq * r; acc = acc + p; }
      else { const f32x4 d = qq - r; const f32x4 p = d * d; acc = acc + p; }
    }
  }
}
__device__ __forceinline__ Line4 r8_load(const uint8_t* __restrict__ row, int L, int half) {
  const u32x4e* p = reinterpret_cast<const u32x4e*>(row + (size_t)L * 128 + (size_t)half * 64);
  Line4 ln;
  ln.c[0] = row_ld<COLTT_PAIR_NT != 0>(p); ln.c[1] = row_ld<COLTT_PAIR_NT != 0>(p + 1); ln.c[2] = row_ld<COLTT_PAIR_NT != 0>(p + 2); ln.c[3] = row_ld<COLTT_PAIR_NT != 0>(p + 3);
  return ln;
}
template <int METRIC, int QUANT, int U = 3>
__device__ __forceinline__ float pair_distance_r8(const uint8_t* __restrict__ row, const float* __restrict__ q, int dim, float qnorm, float rnorm, int half) {
  static_assert(QUANT == Q_NONE || QUANT == Q_F16 || QUANT == Q_BF16, "line-transposed rows exist for f32 and 2-byte codes");
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nl = (dim * elem_bytes<QUANT>()) >> 7;
#define COLTT_LD_(L_) r8_load(row, (L_), half)
#define COLTT_CS_(RAW, L_) r8_consume<METRIC, QUANT>(acc, RAW, q, (L_), half)
  COLTT_BURST_WALK(U, Line4, nl, COLTT_LD_, COLTT_CS_)
#undef COLTT_LD_
#undef COLTT_CS_
  const float s = pair_hsum(acc, half);
  if constexpr (METRIC == M_COS) return cos_epilogue(s, qnorm, rnorm);
  else return go_sqrt(s);
}
// element e of a line-transposed row (host read-backs, the builder's own query): rows8.hpp's index map, restated here for exact.hpp's users
template <int QUANT> __device__ __host__ __forceinline__ int r8_index(int e) {
  constexpr int S = QUANT == Q_NONE ? 4 : 8;
  const int step = e >> 3, r = e & 7;
  return ((step / S) * 8 + r) * S + (step % S);
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
