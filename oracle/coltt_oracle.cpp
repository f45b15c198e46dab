// coltt_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain C++ restatement of the reference's (sjy-dv/coltt @ 2025-03-28) ANN search hot path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
// the product (libcoltt_gpu.so) never links, loads or calls it.
//
// PARITY PIN: the reference's own tests pin no numerical search result (SURVEY.md §0 finding 10,
// §8c), the Go toolchain is absent, so the Go path cannot be run here.  What IS pinned:
//   * the distance kernels (orc_l2 / orc_cosine, order=avx|sse) are checked bit-for-bit against the
//     reference's own pkg/distance/simd/cpp/{avx,sse}.cpp compiled from where they lie
//     (oracle/_ref/libcoltt_ref_simd.so, recipe oracle/Makefile) — tests/test_oracle.py::test_distance_orders_equal_reference_sources;
//   * the codecs are checked against IEEE binary16 (numpy float16) over all 65 536 codes / random f32;
//   * FNV-1a against Python's own restatement of hash/fnv.
// Everything above that (heaps, FLAT scan, HNSW) is a line-by-line restatement with the reference
// file:line cited on every function; for those layers parity is "unpinned by the reference" and the
// golden fixtures under tests/golden/ are produced by this oracle.
//
// Build: see oracle/Makefile  (-O2 -mavx -mno-fma -ffp-contract=off : same instruction mix and
// summation order as pkg/distance/simd/avx/AVX_amd64.s).
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <atomic>
#include <chrono>
#include <queue>
#include <thread>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#if defined(__clang__) || defined(__GNUC__)
typedef float v8f __attribute__((vector_size(32)));
typedef float v4f __attribute__((vector_size(16)));
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// pkg/distance — summation orders
// ------------------------------------------------------------------------------------------------
enum { ORDER_AVX = 0, ORDER_SSE = 1, ORDER_NATIVE = 2 };

// _sum_vector for __m256: hadd,hadd, lane0+lane4  (pkg/distance/simd/cpp/avx.cpp:4-8)
static inline float hsum8(const v8f& v) {
  return ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}
// _sum_vector for __m128: v0+v1+v2+v3 left to right (pkg/distance/simd/cpp/sse.cpp:3-6)
static inline float hsum4(const v4f& v) { return ((v[0] + v[1]) + v[2]) + v[3]; }

static inline v8f ld8(const float* p) { v8f r; std::memcpy(&r, p, 32); return r; }
static inline v4f ld4(const float* p) { v4f r; std::memcpy(&r, p, 16); return r; }

// euclidean_distance_squared (avx.cpp:15-32 / sse.cpp:13-33); native_impl.go:23-30
static float l2sq(int order, const float* a, const float* b, size_t len) {
  if (order == ORDER_AVX) {
    v8f acc = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0, n8 = (len / 8) * 8;
    for (; i < n8; i += 8) { v8f d = ld8(a + i) - ld8(b + i); v8f m = d * d; acc = acc + m; }
    float r = hsum8(acc);
    for (; i < len; i++) { float d = a[i] - b[i]; r += d * d; }
    return r;
  } else if (order == ORDER_SSE) {
    v4f acc = {0, 0, 0, 0};
    size_t i = 0, n4 = (len / 4) * 4;
    for (; i < n4; i += 4) { v4f d = ld4(a + i) - ld4(b + i); v4f m = d * d; acc = acc + m; }
    float r = hsum4(acc);
    for (; i < len; i++) { float d = a[i] - b[i]; r += d * d; }
    return r;
  }
  float r = 0;
  for (size_t i = 0; i < len; i++) { float d = a[i] - b[i]; r += d * d; }
  return r;
}

// cosine_similarity_dot_norm (avx.cpp:51-75 / sse.cpp:54-82); native_impl.go:41-52
static void cos_parts(int order, const float* a, const float* b, size_t len, float* dot, float* na,
                      float* nb) {
  if (order == ORDER_AVX) {
    v8f d = {0, 0, 0, 0, 0, 0, 0, 0}, x = d, y = d;
    size_t i = 0, n8 = (len / 8) * 8;
    for (; i < n8; i += 8) {
      v8f v1 = ld8(a + i), v2 = ld8(b + i);
      v8f p = v1 * v2; d = d + p;
      v8f q = v1 * v1; x = x + q;
      v8f r = v2 * v2; y = y + r;
    }
    float ds = hsum8(d), xs = hsum8(x), ys = hsum8(y);
    for (; i < len; i++) { ds += a[i] * b[i]; xs += a[i] * a[i]; ys += b[i] * b[i]; }
    *dot = ds; *na = xs; *nb = ys;
  } else if (order == ORDER_SSE) {
    v4f d = {0, 0, 0, 0}, x = d, y = d;
    size_t i = 0, n4 = (len / 4) * 4;
    for (; i < n4; i += 4) {
      v4f v1 = ld4(a + i), v2 = ld4(b + i);
      v4f p = v1 * v2; d = d + p;
      v4f q = v1 * v1; x = x + q;
      v4f r = v2 * v2; y = y + r;
    }
    float ds = hsum4(d), xs = hsum4(x), ys = hsum4(y);
    for (; i < len; i++) { ds += a[i] * b[i]; xs += a[i] * a[i]; ys += b[i] * b[i]; }
    *dot = ds; *na = xs; *nb = ys;
  } else {
    float ds = 0, xs = 0, ys = 0;
    for (size_t i = 0; i < len; i++) { ds += a[i] * b[i]; xs += a[i] * a[i]; ys += b[i] * b[i]; }
    *dot = ds; *na = xs; *nb = ys;
  }
}

// manhattan_distance (avx.cpp:34-49 / sse.cpp:35-53): the vector part adds sqrt(d * d) per lane — NOT |d|: the square is rounded to
// f32 first (it underflows to 0 for |d| < 2^-75 and overflows to +Inf above 2^64) and _mm256_sqrt_ps is correctly rounded — the scalar
// tail adds abs(d); native_impl.go:33-40 adds gomath.Abs(d) sequentially.  Manhattan.Distance returns the sum itself (space.go:77-79).
static inline float sqrt_rn(float x) { return (float)std::sqrt((double)x); }   // correctly rounded f32 sqrt (53 >= 2 * 24 + 2)
static float manhattan(int order, const float* a, const float* b, size_t len) {
  if (order == ORDER_AVX) {
    v8f acc = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t i = 0, n8 = (len / 8) * 8;
    for (; i < n8; i += 8) { v8f d = ld8(a + i) - ld8(b + i); v8f m = d * d; for (int j = 0; j < 8; j++) acc[j] = acc[j] + sqrt_rn(m[j]); }
    float r = hsum8(acc);
    for (; i < len; i++) { float d = a[i] - b[i]; r += (d < 0) ? -d : d; }
    return r;
  } else if (order == ORDER_SSE) {
    v4f acc = {0, 0, 0, 0};
    size_t i = 0, n4 = (len / 4) * 4;
    for (; i < n4; i += 4) { v4f d = ld4(a + i) - ld4(b + i); v4f m = d * d; for (int j = 0; j < 4; j++) acc[j] = acc[j] + sqrt_rn(m[j]); }
    float r = hsum4(acc);
    for (; i < len; i++) { float d = a[i] - b[i]; r += (d < 0) ? -d : d; }
    return r;
  }
  float r = 0;
  for (size_t i = 0; i < len; i++) r += (float)std::fabs((double)(a[i] - b[i]));   // gomath.Abs
  return r;
}

// gomath.Sqrt: float32(math.Sqrt(float64(x)))  (pkg/gomath/math.go:48-50)
static inline float go_sqrt(float x) { return (float)std::sqrt((double)x); }
// gomath.Abs: float32(math.Abs(float64(x)))     (pkg/gomath/math.go:35-37)
static inline float go_abs(float x) { return (float)std::fabs((double)x); }

// Euclidean.Distance -> impl.EuclideanDistance (space.go:61-63; AVX_amd64.go:28-32; native_impl.go:23-30)
static float dist_l2(int order, const float* a, const float* b, size_t len) {
  return go_sqrt(l2sq(order, a, b, len));
}
// Cosine.Distance = Abs(impl.CosineDistance) (space.go:93-95; AVX_amd64.go:46-52; native_impl.go:41-52)
static float dist_cos(int order, const float* a, const float* b, size_t len) {
  float dot, na, nb;
  cos_parts(order, a, b, len, &dot, &na, &nb);
  float d;
  if (order == ORDER_NATIVE) d = 1.0f - dot / (go_sqrt(na) * go_sqrt(nb));
  else { float nsq = na * nb; d = 1.0f - dot / go_sqrt(nsq); }
  return go_abs(d);
}
enum { METRIC_COS = 0, METRIC_L2 = 1 };  // edgepb.Distance order (idl/proto/v4/edge.proto:70-73)
static inline float dist(int metric, int order, const float* a, const float* b, size_t len) {
  return metric == METRIC_COS ? dist_cos(order, a, b, len) : dist_l2(order, a, b, len);
}

// Normalize (edge/vectorstore.go:173-189 == core/vectorindex/metadata.go:107-123)
static void normalize(const float* v, float* out, size_t len) {
  float norm = 0;
  for (size_t i = 0; i < len; i++) norm += v[i] * v[i];
  if (norm == 0) { for (size_t i = 0; i < len; i++) out[i] = 0; return; }
  norm = go_sqrt(norm);
  for (size_t i = 0; i < len; i++) out[i] = v[i] / norm;
}

// ------------------------------------------------------------------------------------------------
// pkg/compresshelper — codecs
// ------------------------------------------------------------------------------------------------
static inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// f16bitsToF32bits (float16.go:237-272) == BF16bitsToF32bits (bf16.go:233-268)
static uint32_t f16bits_to_f32bits(uint16_t in) {
  uint32_t sign = (uint32_t)(in & 0x8000) << 16;
  uint32_t exp = (uint32_t)(in & 0x7c00) >> 10;
  uint32_t coef = (uint32_t)(in & 0x03ff) << 13;
  if (exp == 0x1f) {
    if (coef == 0) return sign | 0x7f800000u | coef;
    return sign | 0x7fc00000u | coef;
  }
  if (exp == 0) {
    if (coef == 0) return sign;
    exp++;
    while ((coef & 0x7f800000u) == 0) { coef <<= 1; exp--; }
    coef &= 0x007fffffu;
  }
  return sign | ((exp + (0x7f - 0xf)) << 23) | coef;
}
// f32bitsToF16bits (float16.go:276-321) == f32bitsToBF16bits (bf16.go:272-317)
static uint16_t f32bits_to_f16bits(uint32_t u32) {
  uint32_t sign = u32 & 0x80000000u, exp = u32 & 0x7f800000u, coef = u32 & 0x007fffffu;
  if (exp == 0x7f800000u) {
    uint32_t nanBit = coef != 0 ? 0x0200u : 0u;
    return (uint16_t)((sign >> 16) | 0x7c00u | nanBit | (coef >> 13));
  }
  uint32_t halfSign = sign >> 16;
  int32_t unbiasedExp = (int32_t)(exp >> 23) - 127;
  int32_t halfExp = unbiasedExp + 15;
  if (halfExp >= 0x1f) return (uint16_t)(halfSign | 0x7c00u);
  if (halfExp <= 0) {
    if (14 - halfExp > 24) return (uint16_t)halfSign;
    uint32_t c = coef | 0x00800000u;
    uint32_t halfCoef = c >> (uint32_t)(14 - halfExp);
    uint32_t roundBit = 1u << (uint32_t)(13 - halfExp);
    if ((c & roundBit) != 0 && (c & (3 * roundBit - 1)) != 0) halfCoef++;
    return (uint16_t)(halfSign | halfCoef);
  }
  uint32_t uHalfExp = (uint32_t)halfExp << 10;
  uint32_t halfCoef = coef >> 13;
  uint32_t roundBit = 0x00001000u;
  if ((coef & roundBit) != 0 && (coef & (3 * roundBit - 1)) != 0)
    return (uint16_t)((halfSign | uHalfExp | halfCoef) + 1);
  return (uint16_t)(halfSign | uHalfExp | halfCoef);
}
// F8bitsToF32bits (float8.go:233-266).  NB `(in&0x7c)>>10` is always 0 and the subnormal loop runs on
// a uint32 `exp` that wraps — restated with the same unsigned arithmetic.
static uint32_t f8bits_to_f32bits(uint8_t in) {
  uint32_t sign = (uint32_t)(in & 0x80) << 8;
  uint32_t exp = (uint32_t)(in & 0x7c) >> 10;
  uint32_t coef = (uint32_t)(in & 0x03) << 13;
  if (exp == 0x1f) {
    if (coef == 0) return sign | 0x7f800000u | coef;
    return sign | 0x7fc00000u | coef;
  }
  if (exp == 0) {
    if (coef == 0) return sign;
    exp++;
    while ((coef & 0x7f800000u) == 0) { coef <<= 1; exp--; }
    coef &= 0x007fffffu;
  }
  return sign | ((exp + (0x7f - 0xf)) << 23) | coef;
}
// f32bitsToF8bits (float8.go:270-313): the fp16 algorithm with `sign = u32 & 0x800000` and every
// return truncated to uint8.
static uint8_t f32bits_to_f8bits(uint32_t u32) {
  uint32_t sign = u32 & 0x800000u, exp = u32 & 0x7f800000u, coef = u32 & 0x007fffffu;
  if (exp == 0x7f800000u) {
    uint32_t nanBit = coef != 0 ? 0x0200u : 0u;
    return (uint8_t)((sign >> 8) | 0x7cu | nanBit | (coef >> 13));
  }
  uint32_t halfSign = sign >> 8;
  int32_t unbiasedExp = (int32_t)(exp >> 23) - 127;
  int32_t halfExp = unbiasedExp + 15;
  if (halfExp >= 0x1f) return (uint8_t)(halfSign | 0x7cu);
  if (halfExp <= 0) {
    if (14 - halfExp > 24) return (uint8_t)halfSign;
    uint32_t c = coef | 0x00800000u;
    uint32_t halfCoef = c >> (uint32_t)(14 - halfExp);
    uint32_t roundBit = 1u << (uint32_t)(13 - halfExp);
    if ((c & roundBit) != 0 && (c & (3 * roundBit - 1)) != 0) halfCoef++;
    return (uint8_t)(halfSign | halfCoef);
  }
  uint32_t uHalfExp = (uint32_t)halfExp << 10;
  uint32_t halfCoef = coef >> 13;
  uint32_t roundBit = 0x00001000u;
  if ((coef & roundBit) != 0 && (coef & (3 * roundBit - 1)) != 0)
    return (uint8_t)((halfSign | uHalfExp | halfCoef) + 1);
  return (uint8_t)(halfSign | uHalfExp | halfCoef);
}

enum { Q_NONE = 0, Q_F16 = 1, Q_F8 = 2, Q_BF16 = 3 };  // edgepb.Quantization (edge.proto:75-80)
static inline size_t quant_bytes(int q) { return q == Q_NONE ? 4 : (q == Q_F8 ? 1 : 2); }

// Quantization.Lower (edge/quantization.go:47-49; f16_quantization.go:47-53; f8_…:45-51; bf16_…:45-51)
static void lower(int q, const float* v, size_t len, uint8_t* out) {
  if (q == Q_NONE) { std::memcpy(out, v, len * 4); return; }
  if (q == Q_F8) { for (size_t i = 0; i < len; i++) out[i] = f32bits_to_f8bits(f2u(v[i])); return; }
  uint16_t* o = (uint16_t*)out;
  for (size_t i = 0; i < len; i++) o[i] = f32bits_to_f16bits(f2u(v[i]));
}
// the decode half of Quantization.Similarity (f16_quantization.go:35-45 etc.)
static void raise(int q, const uint8_t* in, size_t len, float* out) {
  if (q == Q_NONE) { std::memcpy(out, in, len * 4); return; }
  if (q == Q_F8) { for (size_t i = 0; i < len; i++) out[i] = u2f(f8bits_to_f32bits(in[i])); return; }
  const uint16_t* p = (const uint16_t*)in;
  for (size_t i = 0; i < len; i++) out[i] = u2f(f16bits_to_f32bits(p[i]));
}

// ------------------------------------------------------------------------------------------------
// pkg/sharding
// ------------------------------------------------------------------------------------------------
// ShardVertex (pkg/sharding/shard.go:34-41): hash/fnv New64a over the 8 little-endian bytes of id.
static uint64_t shard_vertex(uint64_t x, uint64_t c) {
  uint64_t h = 14695981039346656037ull;
  for (int i = 0; i < 8; i++) { h ^= (x >> (8 * i)) & 0xff; h *= 1099511628211ull; }
  return h % c;
}

// ------------------------------------------------------------------------------------------------
// pkg/distancepq — avo-generated FMA kernels (asm/dot.s:7-55, asm/euclidean.s:7-65)
// ------------------------------------------------------------------------------------------------
static float pq_dot(const float* x, const float* y, size_t len) {
  float acc[4][8] = {};
  size_t i = 0;
  for (; len - i >= 32; i += 32)
    for (int r = 0; r < 4; r++)
      for (int j = 0; j < 8; j++) acc[r][j] = std::fmaf(x[i + 8 * r + j], y[i + 8 * r + j], acc[r][j]);
  float tail = 0;  // X4 lane 0; VFMADD231SS
  for (; i < len; i++) tail = std::fmaf(x[i], y[i], tail);
  float s[8];
  for (int j = 0; j < 8; j++) s[j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
  float t[4];
  for (int j = 0; j < 4; j++) t[j] = s[j] + s[j + 4];  // VEXTRACTF128 + VADDPS
  t[0] = t[0] + tail;                                  // VADDPS X0, X4 (X4 = {tail,0,0,0})
  t[1] = t[1] + 0.0f; t[2] = t[2] + 0.0f; t[3] = t[3] + 0.0f;
  float h0 = t[0] + t[1], h1 = t[2] + t[3];            // VHADDPS
  return h0 + h1;                                      // VHADDPS
}
static float pq_l2sq(const float* x, const float* y, size_t len) {
  float acc[4][8] = {};
  size_t i = 0;
  for (; len - i >= 32; i += 32)
    for (int r = 0; r < 4; r++)
      for (int j = 0; j < 8; j++) {
        float d = x[i + 8 * r + j] - y[i + 8 * r + j];
        acc[r][j] = std::fmaf(d, d, acc[r][j]);
      }
  float tail = 0;
  for (; i < len; i++) { float d = x[i] - y[i]; tail = std::fmaf(d, d, tail); }
  float s[8];
  for (int j = 0; j < 8; j++) s[j] = ((acc[0][j] + acc[1][j]) + acc[2][j]) + acc[3][j];
  float t[4];
  for (int j = 0; j < 4; j++) t[j] = s[j] + s[j + 4];
  t[0] = t[0] + tail; t[1] = t[1] + 0.0f; t[2] = t[2] + 0.0f; t[3] = t[3] + 0.0f;
  float h0 = t[0] + t[1], h1 = t[2] + t[3];
  return h0 + h1;
}
// puredist.go:20-35
static float pq_dot_pure(const float* x, const float* y, size_t len) {
  float s = 0; for (size_t i = 0; i < len; i++) s += x[i] * y[i]; return s;
}
static float pq_l2sq_pure(const float* x, const float* y, size_t len) {
  float s = 0; for (size_t i = 0; i < len; i++) { float d = x[i] - y[i]; s += d * d; } return s;
}
// hammingDistance / jaccardDistance (pkg/distancepq/distance.go:62-84)
static float pq_hamming(const uint64_t* x, const uint64_t* y, size_t n) {
  long d = 0; for (size_t i = 0; i < n; i++) d += __builtin_popcountll(x[i] ^ y[i]); return (float)d;
}
static float pq_jaccard(const uint64_t* x, const uint64_t* y, size_t n) {
  long in = 0, un = 0;
  for (size_t i = 0; i < n; i++) { in += __builtin_popcountll(x[i] & y[i]); un += __builtin_popcountll(x[i] | y[i]); }
  if (un == 0) return 0;
  return 1 - (float)in / (float)un;
}

// ------------------------------------------------------------------------------------------------
// Go container/heap (go1.23 src/container/heap/heap.go) — Init/Push/Pop/up/down restated.
// Call sites: core/vectorindex/priority_queue.go:57-99; edge/priorityqueue/priority_queue.go:57-98.
// ------------------------------------------------------------------------------------------------
struct PQItem { float prio; int64_t val; };
template <bool MAX> struct GoHeap {
  std::vector<PQItem> a;
  // minPriorityQueue.Less: a[i].prio < a[j].prio ; maxPriorityQueue.Less: > (priority_queue.go:161-163,183-185)
  inline bool less(int i, int j) const { return MAX ? a[i].prio > a[j].prio : a[i].prio < a[j].prio; }
  void up(int j) {
    for (;;) {
      int i = (j - 1) / 2;
      if (i == j || !less(j, i)) break;
      std::swap(a[i], a[j]);
      j = i;
    }
  }
  bool down(int i0, int n) {
    int i = i0;
    for (;;) {
      int j1 = 2 * i + 1;
      if (j1 >= n || j1 < 0) break;
      int j = j1;
      int j2 = j1 + 1;
      if (j2 < n && less(j2, j1)) j = j2;
      if (!less(j, i)) break;
      std::swap(a[i], a[j]);
      i = j;
    }
    return i > i0;
  }
  void init() { int n = (int)a.size(); for (int i = n / 2 - 1; i >= 0; i--) down(i, n); }
  void push(PQItem x) { a.push_back(x); up((int)a.size() - 1); }
  PQItem pop() {
    int n = (int)a.size() - 1;
    std::swap(a[0], a[n]);
    down(0, n);
    PQItem x = a.back();
    a.pop_back();
    return x;
  }
  const PQItem& peek() const { return a[0]; }  // Peek = ToSlice()[0] (priority_queue.go:101-107)
  int len() const { return (int)a.size(); }
};

// ------------------------------------------------------------------------------------------------
// Canonical total order used wherever the reference's outcome depends on Go map iteration order or
// on heap/sort tie handling: key = (score bits as f32 compare, then 64-bit tiebreak).
// ------------------------------------------------------------------------------------------------
struct Scored { float score; uint64_t tie; };
// Canonical (score, id) order = order of the sortable key of the score's IEEE bits (negative values below positive ones, a NaN
// score — a zero-norm operand under cosine — after +Inf): a TOTAL order, equal to the value order for every ordinary score.  The
// reference's queue compares NaN priorities with `<` (false both ways): its order is then not defined, and mode 0 / 1 (the literal
// Go heap) keep exactly that behaviour.
static inline uint32_t score_sort_key(float f) {
  uint32_t u; std::memcpy(&u, &f, 4);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline bool scored_less(const Scored& x, const Scored& y) {
  const uint32_t kx = score_sort_key(x.score), ky = score_sort_key(y.score);
  if (kx != ky) return kx < ky;
  return x.tie < y.tie;
}

// ------------------------------------------------------------------------------------------------
// edge FLAT store  (edge/none_vectorstore.go, f16_/f8_/bf16_vectorstore.go)
// ------------------------------------------------------------------------------------------------
struct Flat {
  uint32_t dim; int metric; int quant; int order;
  // vertices[shard] : id -> stored (lowered) vector bytes.  std::map gives the canonical
  // ascending-id iteration that stands in for Go's random map order.
  std::map<uint64_t, std::vector<uint8_t>> shards[16];
};

// {none,f16,f8,bf16}VecSpace.ChangedVertex, vector part (none_vectorstore.go:86-101; f16_…:87-105)
static int flat_upsert(Flat* f, uint64_t id, const float* vec) {
  std::vector<float> tmp(f->dim);
  const float* v = vec;
  if (f->metric == METRIC_COS) { normalize(vec, tmp.data(), f->dim); v = tmp.data(); }
  std::vector<uint8_t> low(f->dim * quant_bytes(f->quant));
  lower(f->quant, v, f->dim, low.data());
  f->shards[shard_vertex(id, 16)][id] = std::move(low);
  return 0;
}

// Quantization.Similarity (quantization.go:43-45; f16_quantization.go:35-45): decode BOTH, then Distance
static float flat_similarity(const Flat* f, const uint8_t* x, const uint8_t* y, float* bx, float* by) {
  if (f->quant == Q_NONE) return dist(f->metric, f->order, (const float*)x, (const float*)y, f->dim);
  raise(f->quant, x, f->dim, bx);
  raise(f->quant, y, f->dim, by);
  return dist(f->metric, f->order, bx, by, f->dim);
}

// edge.PriorityQueue (edge/priority_queue.go:33-69): min-heap, pop-min when over capacity => keeps the
// K LARGEST scores; ToSlice sorts ascending.  `nearest` flips to a max-heap (the useful direction; not
// reference behaviour).
struct EdgePQ {
  GoHeap<false> mn; GoHeap<true> mx; int maxSize; bool nearest;
  void add(float score, uint64_t id) {
    if (!nearest) { mn.push({score, (int64_t)id}); if (mn.len() > maxSize) mn.pop(); }
    else { mx.push({score, (int64_t)id}); if (mx.len() > maxSize) mx.pop(); }
  }
  std::vector<PQItem> to_slice() const {
    std::vector<PQItem> r = nearest ? mx.a : mn.a;
    // sort.Slice is an unstable pdqsort; tie order is unspecified in the reference.  stable_sort here.
    std::stable_sort(r.begin(), r.end(), [](const PQItem& x, const PQItem& y) { return x.prio < y.prio; });
    return r;
  }
};

// mode 0: literal, highCpu=false (none_vectorstore.go:135-147)
// mode 1: literal, highCpu=true  (none_vectorstore.go:148-178): 16 local queues -> global queue
// mode 2: canonical closed form: the K extreme (score,id) pairs, ascending — what the GPU computes.
// cand/n_cand != null: FilterableVertexSearch's candidate walk (none_vectorstore.go:192-252)
static int flat_search(const Flat* f, const float* query, int topK, int nearest, int mode, bool use_cand,
                       const uint64_t* cand, size_t n_cand, uint64_t* out_ids, float* out_scores) {
  std::vector<float> q(f->dim), bx(f->dim), by(f->dim);
  if (f->metric == METRIC_COS) normalize(query, q.data(), f->dim);
  else std::memcpy(q.data(), query, f->dim * 4);
  std::vector<uint8_t> low(f->dim * quant_bytes(f->quant));
  lower(f->quant, q.data(), f->dim, low.data());  // f16_vectorstore.go:136 — the QUERY is lowered too

  std::vector<std::vector<uint64_t>> shard_cand(16);
  if (use_cand) for (size_t i = 0; i < n_cand; i++) shard_cand[shard_vertex(cand[i], 16)].push_back(cand[i]);

  auto scan_shard = [&](int s, auto&& emit) {
    if (use_cand) {
      for (uint64_t id : shard_cand[s]) {
        auto it = f->shards[s].find(id);
        if (it == f->shards[s].end()) continue;
        emit(id, flat_similarity(f, low.data(), it->second.data(), bx.data(), by.data()));
      }
    } else {
      for (auto& kv : f->shards[s])
        emit(kv.first, flat_similarity(f, low.data(), kv.second.data(), bx.data(), by.data()));
    }
  };

  std::vector<PQItem> res;
  if (mode == 2) {
    std::vector<Scored> all;
    for (int s = 0; s < 16; s++) scan_shard(s, [&](uint64_t id, float sc) { all.push_back({sc, id}); });
    std::sort(all.begin(), all.end(), scored_less);
    size_t n = all.size(), k = std::min((size_t)std::max(topK, 0), n);
    size_t lo = nearest ? 0 : n - k;
    for (size_t i = 0; i < k; i++) res.push_back({all[lo + i].score, (int64_t)all[lo + i].tie});
  } else if (mode == 0) {
    EdgePQ pq; pq.maxSize = topK; pq.nearest = nearest;
    for (int s = 0; s < 16; s++) scan_shard(s, [&](uint64_t id, float sc) { pq.add(sc, id); });
    res = pq.to_slice();
  } else {
    EdgePQ pq; pq.maxSize = topK; pq.nearest = nearest;
    std::vector<std::vector<PQItem>> local(16);
    for (int s = 0; s < 16; s++) {
      EdgePQ lp; lp.maxSize = topK; lp.nearest = nearest;
      scan_shard(s, [&](uint64_t id, float sc) { lp.add(sc, id); });
      local[s] = lp.to_slice();
    }
    for (int s = 0; s < 16; s++) for (auto& it : local[s]) pq.add(it.prio, (uint64_t)it.val);
    res = pq.to_slice();
  }
  for (size_t i = 0; i < res.size(); i++) { out_ids[i] = (uint64_t)res[i].val; out_scores[i] = res[i].prio; }
  return (int)res.size();
}

// ------------------------------------------------------------------------------------------------
// core/vectorindex HNSW
// ------------------------------------------------------------------------------------------------
struct HnswCfg {  // hnsw_config.go:135-162
  int32_t m, mMax, mMax0, ef, efConstruction, algo /*0 simple, 1 heuristic, 2 DIVERSE (not reference behaviour: a definition, see select_diverse)*/;
  float levelMultiplier;
  int32_t extendCandidates, keepPruned;
};
struct Edge { int32_t to; float d; };  // hnswEdgeSet entry (hnsw_vertex.go:27)
struct Vertex {                          // hnswVertex (hnsw_vertex.go:29-39)
  uint64_t id; int level; bool deleted;
  std::vector<float> own;                // reference-shaped: one heap allocation per vector
  const float* vec;                      // == own.data(), or a view into a caller-owned matrix
  // edges[l]: kept sorted by `to` (insertion index) — the canonical stand-in for Go map order.
  std::vector<std::vector<Edge>> edges;
};
struct HnswStats { uint64_t n_dist, n_exp, n_hops; };
struct Hnsw {
  uint32_t dim; int metric; int order; HnswCfg cfg;
  std::vector<Vertex> v;                         // slot = insertion index (never reused)
  std::unordered_map<uint64_t, int32_t> by_id;   // live vertices only (the 16 sharded maps, hnsw.go:49-50)
  int32_t entry = -1; uint64_t len = 0;
  bool canon_build = false;  // Insert/prune use the canonical closed forms (what the GPU builder runs)
  HnswStats st{};
  float D(const float* a, const float* b) { st.n_dist++; return dist(metric, order, a, b, dim); }
};

static void edge_set(std::vector<Edge>& es, int32_t to, float d) {  // map[k]=v
  auto it = std::lower_bound(es.begin(), es.end(), to, [](const Edge& e, int32_t t) { return e.to < t; });
  if (it != es.end() && it->to == to) it->d = d; else es.insert(it, {to, d});
}
static void edge_del(std::vector<Edge>& es, int32_t to) {
  auto it = std::lower_bound(es.begin(), es.end(), to, [](const Edge& e, int32_t t) { return e.to < t; });
  if (it != es.end() && it->to == to) es.erase(it);
}

// greedyClosestNeighbor (hnsw.go:320-343)
static void greedy(Hnsw* h, const float* q, int32_t& ep, float& minD, int level) {
  for (;;) {
    int32_t closest = -1;
    for (const Edge& e : h->v[ep].edges[level]) {
      const Vertex& n = h->v[e.to];
      if (n.deleted) continue;
      float d = h->D(q, n.vec);
      if (d < minD) { minD = d; closest = e.to; }
    }
    h->st.n_hops++;
    if (closest < 0) break;
    ep = closest;
  }
}

// searchLevel (hnsw.go:345-389), literal: Go heaps, stale lowerBound, canonical neighbour order.
static GoHeap<true> search_level_literal(Hnsw* h, const float* q, int32_t ep, int ef, int level) {
  float epd = h->D(q, h->v[ep].vec);
  GoHeap<false> cand; GoHeap<true> res;
  cand.push({epd, ep}); res.push({epd, ep});
  std::unordered_set<int32_t> visited; visited.reserve((size_t)ef * h->cfg.mMax0);
  visited.insert(ep);
  while (cand.len() > 0) {
    PQItem c = cand.pop();
    float lowerBound = res.peek().prio;
    if (c.prio > lowerBound) break;
    h->st.n_exp++;
    for (const Edge& e : h->v[c.val].edges[level]) {
      const Vertex& n = h->v[e.to];
      if (n.deleted) continue;
      if (!visited.insert(e.to).second) continue;
      float d = h->D(q, n.vec);
      if (d < lowerBound || res.len() < ef) {
        cand.push({d, e.to}); res.push({d, e.to});
        if (res.len() > ef) res.pop();
      }
    }
  }
  return res;
}

// searchLevel, canonical closed form: result set = sorted array with an `expanded`
// flag, key (distance, slot); the candidate queue is "the unexpanded members of the result set";
// per popped candidate, in canonical neighbour order, the first (ef-len0) eligible neighbours are
// admitted unconditionally, the rest iff d < lowerBound sampled before the loop; then keep the ef
// smallest.  This is the algorithm the HIP kernel runs.  Equal to the literal form whenever no two
// distinct vertices have bit-equal distances to the query.
struct RItem { float d; int32_t slot; bool expanded; };
// Order of the canonical result set = order of the key (IEEE bits of d, slot).  Distances are |1 - x| or a square root, i.e. never
// negative, so for every ordinary value this IS (d, slot); a NaN distance (a stored vector whose norm underflows to zero under
// cosine) sorts after +Inf instead of comparing "equal to everything", which keeps the order total.  The reference's heaps have no
// defined order for NaN priorities (Less is false both ways), so with NaN distances neither form reproduces the Go code.
static inline bool ritem_less(const RItem& a, const RItem& b) {
  uint32_t ka, kb;
  std::memcpy(&ka, &a.d, 4); std::memcpy(&kb, &b.d, 4);
  if (ka != kb) return ka < kb;
  return a.slot < b.slot;
}
static std::vector<RItem> search_level_canon(Hnsw* h, const float* q, int32_t ep, int ef, int level) {
  std::vector<RItem> res;
  res.push_back({h->D(q, h->v[ep].vec), ep, false});
  std::unordered_set<int32_t> visited; visited.insert(ep);
  std::vector<RItem> adm;
  for (;;) {
    int ci = -1;
    for (int i = 0; i < (int)res.size(); i++) if (!res[i].expanded) { ci = i; break; }
    if (ci < 0) break;
    res[ci].expanded = true;
    float lowerBound = res.back().d;
    int len0 = (int)res.size();
    int32_t c = res[ci].slot;
    h->st.n_exp++;
    adm.clear();
    int free_slots = ef - len0;
    for (const Edge& e : h->v[c].edges[level]) {
      const Vertex& n = h->v[e.to];
      if (n.deleted) continue;
      if (!visited.insert(e.to).second) continue;
      float d = h->D(q, n.vec);
      if (free_slots > 0) { adm.push_back({d, e.to, false}); free_slots--; }
      else if (d < lowerBound) adm.push_back({d, e.to, false});
    }
    for (auto& a : adm) res.insert(std::upper_bound(res.begin(), res.end(), a, ritem_less), a);
    if ((int)res.size() > ef) res.resize(ef);
  }
  return res;
}

// selectNeighbors (hnsw.go:391-397)
static void select_simple(GoHeap<true>& nb, int k) { while (nb.len() > k) nb.pop(); }

// selectNeighborsHeuristic (hnsw.go:399-447) with extendCandidates=false (true is rejected: the
// reference's Reverse() aliases the backing array — SURVEY.md §0 finding 8).  candidateVertices =
// neighbors.Reverse(): the SAME array re-typed as a min-heap and heap.Init'ed (priority_queue.go:109-122).
static GoHeap<true> select_heuristic(GoHeap<true>& nb, int k, bool keepPruned) {
  GoHeap<false> cand; cand.a = nb.a; cand.init();
  GoHeap<true> result;
  while (cand.len() > 0 && result.len() < k) result.push(cand.pop());
  if (keepPruned) { while (cand.len() > 0) { if (result.len() >= k) break; result.push(cand.pop()); } }
  return result;
}

// algo 2, "diverse" — NOT reference behaviour (the reference's selectNeighborsHeuristic, hnsw.go:399-447, never compares a candidate with
// the neighbours already chosen: it returns the k nearest).  A DEFINITION, after Algorithm 4 of the HNSW paper as hnswlib runs it
// (getNeighborsByHeuristic2), fixed here so that the GPU builder and this file agree bit for bit:
//   candidates `c` ascending by (distance-to-base bits, slot) — the order of the canonical result set;
//   walk them in that order while fewer than k are chosen: c is chosen iff NO already chosen r has D(c, r) < d(base, c), where
//   D(c, r) is the index's distance with c's stored vector as the query and r's as the stored row, and d(base, c) is the distance the
//   candidate list carries (searchLevel's for Insert, the stored edge distance for pruneNeighbors).  EVERY chosen r is evaluated (no
//   early exit: the GPU evaluates them side by side), so a tested candidate costs |chosen| evaluations;
//   keepPruned (hnsw_config.go: heuristicKeepPruned): the rejected candidates, nearest first, fill the result up to k.
// Returns the chosen items in candidate order (chosen first, then the re-added ones).
static std::vector<RItem> select_diverse(Hnsw* h, const std::vector<RItem>& cand, int k, bool keepPruned) {
  std::vector<RItem> chosen, pruned;
  for (const RItem& c : cand) {
    if ((int)chosen.size() >= k) break;
    bool good = true;
    for (const RItem& r : chosen) { float d = h->D(h->v[c.slot].vec, h->v[r.slot].vec); if (d < c.d) good = false; }
    if (good) chosen.push_back(c); else pruned.push_back(c);
  }
  if (keepPruned) for (const RItem& c : pruned) { if ((int)chosen.size() >= k) break; chosen.push_back(c); }
  return chosen;
}

// pruneNeighbors (hnsw.go:449-474)
static void prune(Hnsw* h, int32_t vi, int k, int level) {
  if (h->canon_build) {  // k nearest by (stored distance, slot); Simple == Heuristic(extend=false)
    std::vector<RItem> all;
    for (const Edge& e : h->v[vi].edges[level]) { if (h->v[e.to].deleted) continue; all.push_back({e.d, e.to, false}); }
    std::sort(all.begin(), all.end(), ritem_less);
    // diverse: the selection runs only when the live neighbours do not fit (hnswlib's rule; Remove's re-prune therefore only drops tombstones)
    if (h->cfg.algo == 2 && (int)all.size() > k) all = select_diverse(h, all, k, h->cfg.keepPruned != 0);
    if ((int)all.size() > k) all.resize(k);
    std::vector<Edge> ne;
    for (auto& r : all) ne.push_back({r.slot, r.d});
    std::sort(ne.begin(), ne.end(), [](const Edge& a, const Edge& b) { return a.to < b.to; });
    h->v[vi].edges[level] = std::move(ne);
    return;
  }
  GoHeap<true> nq;
  for (const Edge& e : h->v[vi].edges[level]) { if (h->v[e.to].deleted) continue; nq.push({e.d, e.to}); }
  if (h->cfg.algo == 0) select_simple(nq, k); else nq = select_heuristic(nq, k, h->cfg.keepPruned);
  std::vector<Edge> ne;
  for (auto& it : nq.a) ne.push_back({(int32_t)it.val, it.prio});
  std::sort(ne.begin(), ne.end(), [](const Edge& a, const Edge& b) { return a.to < b.to; });
  h->v[vi].edges[level] = std::move(ne);
}

// Hnsw.Insert (hnsw.go:104-167).  Returns 0 ok, -2 ItemAlreadyExistsError (hnsw.go:293-295).
static int hnsw_insert(Hnsw* h, uint64_t id, const float* value, int vertexLevel, bool view) {
  if (h->by_id.count(id)) return -2;
  Vertex nv; nv.id = id; nv.deleted = false;
  if (h->metric == METRIC_COS || !view) {
    nv.own.resize(h->dim);
    if (h->metric == METRIC_COS) normalize(value, nv.own.data(), h->dim);
    else std::memcpy(nv.own.data(), value, h->dim * 4);
  }
  int32_t vi = (int32_t)h->v.size();
  if (h->entry < 0) {  // first vertex is forced to level 0 (hnsw.go:108-117)
    nv.level = 0; nv.edges.resize(1);
    h->v.push_back(std::move(nv));
    h->v[vi].vec = h->v[vi].own.empty() ? value : h->v[vi].own.data();
    h->by_id[id] = vi; h->len++; h->entry = vi;
    return 0;
  }
  nv.level = vertexLevel; nv.edges.resize(vertexLevel + 1);
  h->v.push_back(std::move(nv));
  h->v[vi].vec = h->v[vi].own.empty() ? value : h->v[vi].own.data();
  h->by_id[id] = vi; h->len++;
  const float* vec = h->v[vi].vec;

  int32_t ep = h->entry;
  float minD = h->D(vec, h->v[ep].vec);
  for (int l = h->v[ep].level; l > vertexLevel; l--) greedy(h, vec, ep, minD, l);

  for (int l = std::min(h->v[ep].level, vertexLevel); l >= 0; l--) {
    GoHeap<true> nb;
    if (h->canon_build) {
      std::vector<RItem> r = search_level_canon(h, vec, ep, h->cfg.efConstruction, l);
      if (h->cfg.algo == 2) r = select_diverse(h, r, h->cfg.m, h->cfg.keepPruned != 0);   // r[0] stays the nearest candidate
      if ((int)r.size() > h->cfg.m) r.resize(h->cfg.m);
      for (auto& x : r) nb.a.push_back({x.d, x.slot});  // ascending array; popped back-to-front below
    } else nb = search_level_literal(h, vec, ep, h->cfg.efConstruction, l);
    if (h->canon_build) {
      int mMaxc = l == 0 ? h->cfg.mMax0 : h->cfg.mMax;
      for (int i = (int)nb.a.size() - 1; i >= 0; i--) {  // farthest first, nearest last (next entrypoint)
        int32_t ni = (int32_t)nb.a[i].val; ep = ni;
        edge_set(h->v[vi].edges[l], ni, nb.a[i].prio);
        edge_set(h->v[ni].edges[l], vi, nb.a[i].prio);
        if ((int)h->v[ni].edges[l].size() > mMaxc) prune(h, ni, mMaxc, l);
      }
      continue;
    }
    if (h->cfg.algo == 0) select_simple(nb, h->cfg.m); else nb = select_heuristic(nb, h->cfg.m, h->cfg.keepPruned);
    int mMax = l == 0 ? h->cfg.mMax0 : h->cfg.mMax;
    while (nb.len() > 0) {
      PQItem it = nb.pop();
      int32_t ni = (int32_t)it.val;
      ep = ni;
      edge_set(h->v[vi].edges[l], ni, it.prio);
      edge_set(h->v[ni].edges[l], vi, it.prio);
      if ((int)h->v[ni].edges[l].size() > mMax) prune(h, ni, mMax, l);
    }
  }
  if (h->entry >= 0 && h->v[vi].level > h->v[h->entry].level) h->entry = vi;
  return 0;
}

// Hnsw.Remove (hnsw.go:191-241).  Returns 0 ok, -3 ItemNotFoundError.
static int hnsw_remove(Hnsw* h, uint64_t id) {
  auto it = h->by_id.find(id);
  if (it == h->by_id.end()) return -3;
  int32_t vi = it->second;
  h->by_id.erase(it); h->len--; h->v[vi].deleted = true;  // removeVertex (hnsw.go:304-318)
  if (h->entry == vi) {
    float minD = 3.40282346638528859811704183484516925440e+38f;  // gomath.MaxFloat
    int32_t closest = -1;
    for (int l = h->v[vi].level; l >= 0; l--) {
      for (const Edge& e : h->v[vi].edges[l]) if (e.d < minD) { minD = e.d; closest = e.to; }
      if (closest >= 0) break;
    }
    h->entry = closest;
  }
  for (int l = h->v[vi].level; l >= 0; l--) {
    int mMax = l == 0 ? h->cfg.mMax0 : h->cfg.mMax;
    std::vector<int32_t> nbs;
    for (const Edge& e : h->v[vi].edges[l]) nbs.push_back(e.to);
    for (int32_t ni : nbs) { edge_del(h->v[ni].edges[l], vi); prune(h, ni, mMax, l); }
  }
  return 0;
}

// Hnsw.Search (hnsw.go:243-278).  mode 0 literal, 1 canonical.  Result ascending by distance.
static int hnsw_search(Hnsw* h, const float* query, int k, int mode, int ef_override,
                       uint64_t* out_ids, float* out_scores, int32_t* out_slots) {
  std::vector<float> qn;
  const float* q = query;
  if (h->metric == METRIC_COS) { qn.resize(h->dim); normalize(query, qn.data(), h->dim); q = qn.data(); }
  if (h->entry < 0) return 0;
  int32_t ep = h->entry;
  float minD = h->D(q, h->v[ep].vec);
  for (int l = h->v[ep].level; l > 0; l--) greedy(h, q, ep, minD, l);
  int ef = std::max(ef_override > 0 ? ef_override : h->cfg.ef, k);
  int n = 0;
  if (mode == 0) {
    GoHeap<true> nb = search_level_literal(h, q, ep, ef, 0);
    if (h->cfg.algo == 0) select_simple(nb, k); else nb = select_heuristic(nb, k, h->cfg.keepPruned);
    n = std::min(k, nb.len());
    for (int i = n - 1; i >= 0; i--) {
      PQItem it = nb.pop();
      out_ids[i] = h->v[it.val].id; out_scores[i] = it.prio; if (out_slots) out_slots[i] = (int32_t)it.val;
    }
  } else {
    std::vector<RItem> res = search_level_canon(h, q, ep, ef, 0);
    n = std::min(k, (int)res.size());
    for (int i = 0; i < n; i++) {
      out_ids[i] = h->v[res[i].slot].id; out_scores[i] = res[i].d; if (out_slots) out_slots[i] = res[i].slot;
    }
  }
  return n;
}


// Batched Insert (what the GPU builder runs).  The reference allows concurrent Inserts
// (per-vertex RWMutexes, hnsw.go:104-167) with unspecified interleaving; this fixes ONE interleaving:
// every vertex of the batch searches the graph as it was before the batch (phase A: hnsw.go:124-135 with
// the canonical searchLevel / k-nearest selection), then all links are applied (phase B: hnsw.go:142-159 —
// addEdge both ways, pruneNeighbors when a row exceeds mMax; the outcome is independent of the order in which
// the batch's links are applied).  batch == 1 is exactly the sequential canonical Insert.
static int hnsw_insert_batch(Hnsw* h, const uint64_t* ids, const float* vecs, const int32_t* levels, size_t n) {
  struct Link { int32_t from; int l; int32_t to; float d; };
  std::vector<Link> links;
  size_t i0 = 0;
  if (h->entry < 0 && n > 0) {  // first vertex: level forced to 0, becomes the entrypoint (hnsw.go:108-117)
    int rc = hnsw_insert(h, ids[0], vecs, 0, false);
    if (rc) return rc;
    i0 = 1;
  }
  int32_t base = (int32_t)h->v.size();
  for (size_t i = i0; i < n; i++) {
    if (h->by_id.count(ids[i])) return -2;
    Vertex nv; nv.id = ids[i]; nv.deleted = false; nv.level = levels[i]; nv.edges.resize(levels[i] + 1);
    nv.own.resize(h->dim);
    if (h->metric == METRIC_COS) normalize(vecs + i * h->dim, nv.own.data(), h->dim);
    else std::memcpy(nv.own.data(), vecs + i * h->dim, h->dim * 4);
    h->v.push_back(std::move(nv));
    h->v.back().vec = h->v.back().own.data();
  }
  for (size_t i = i0; i < n; i++) {  // phase A (frozen graph: new vertices are unreachable, they have no in-edges)
    int32_t vi = base + (int32_t)(i - i0);
    const float* vec = h->v[vi].vec;
    int32_t ep = h->entry;
    float minD = h->D(vec, h->v[ep].vec);
    for (int l = h->v[ep].level; l > levels[i]; l--) greedy(h, vec, ep, minD, l);
    for (int l = std::min(h->v[ep].level, (int)levels[i]); l >= 0; l--) {
      std::vector<RItem> r = search_level_canon(h, vec, ep, h->cfg.efConstruction, l);
      if (h->cfg.algo == 2) r = select_diverse(h, r, h->cfg.m, h->cfg.keepPruned != 0);
      if ((int)r.size() > h->cfg.m) r.resize(h->cfg.m);
      for (auto& x : r) links.push_back({vi, l, x.slot, x.d});
      ep = r[0].slot;
    }
  }
  for (size_t i = i0; i < n; i++) { int32_t vi = base + (int32_t)(i - i0); h->by_id[ids[i]] = vi; h->len++; }
  bool save = h->canon_build; h->canon_build = true;
  if (h->cfg.algo == 2) {
    // diverse selection is not associative, so "prune at every overflow" would depend on the order the batch's links arrive in.  The
    // definition for a batch: a row receives ALL the batch's links, then is pruned ONCE if it overflows (rows are independent of each
    // other: a prune reads the row's stored edge distances and stored vectors only).  batch == 1 adds one link per row: the sequential Insert.
    for (auto& k : links) { edge_set(h->v[k.from].edges[k.l], k.to, k.d); edge_set(h->v[k.to].edges[k.l], k.from, k.d); }
    for (auto& k : links) {
      int mMax = k.l == 0 ? h->cfg.mMax0 : h->cfg.mMax;
      if ((int)h->v[k.to].edges[k.l].size() > mMax) prune(h, k.to, mMax, k.l);
    }
  } else
  for (auto& k : links) {  // phase B
    int mMax = k.l == 0 ? h->cfg.mMax0 : h->cfg.mMax;
    edge_set(h->v[k.from].edges[k.l], k.to, k.d);
    edge_set(h->v[k.to].edges[k.l], k.from, k.d);
    if ((int)h->v[k.to].edges[k.l].size() > mMax) prune(h, k.to, mMax, k.l);
  }
  h->canon_build = save;
  for (size_t i = i0; i < n; i++) {  // entrypoint CAS in insertion order (hnsw.go:161-164)
    int32_t vi = base + (int32_t)(i - i0);
    if (h->v[vi].level > h->v[h->entry].level) h->entry = vi;
  }
  return 0;
}


// ------------------------------------------------------------------------------------------------
// "Contiguous" CPU-baseline variant: the same canonical Hnsw.Search over the GPU's own
// padded-array graph layout (adj0 [n][w0], upper_off [n], adjU [rows][wu], rows [n][dim] f32), so a
// 10M-vertex index exported from HBM can be timed on the host without re-materialising per-vertex objects.
// ------------------------------------------------------------------------------------------------
struct CsrGraph {
  const uint8_t* rows; const uint32_t* adj0; const uint32_t* upper_off; const uint32_t* adjU; const uint32_t* del_bits;
  uint32_t w0, wu, dim; int metric, order; int32_t entry, entry_level; int quant;
};
static inline bool csr_deleted(const CsrGraph& g, uint32_t s) { return g.del_bits && ((g.del_bits[s >> 5] >> (s & 31)) & 1u); }
static inline const uint32_t* csr_row(const CsrGraph& g, uint32_t s, int level, uint32_t& w) {
  if (level == 0) { w = g.w0; return g.adj0 + (size_t)s * g.w0; }
  w = g.wu; return g.adjU + ((size_t)g.upper_off[s] + (uint32_t)(level - 1)) * g.wu;
}
// Per-thread scratch reused across queries.  The reference allocates a fresh visited map and fresh heaps per query and leaves
// them to the Go GC (hnsw.go:346-352); doing the same with malloc/free from 256 threads serialises them on the kernel's
// address-space lock (heap trims / re-faults), which would make the all-cores baseline an artefact.  Set semantics are unchanged.
struct VisitedTable {  // open addressing, generation-stamped (O(1) clear), grows at 50 % load
  std::vector<uint32_t> key, gen; uint32_t cur = 0, mask = 0, count = 0;
  void reset(size_t expect) {
    size_t cap = 1024; while (cap < expect * 4) cap <<= 1;
    if (key.size() < cap) { key.assign(cap, 0); gen.assign(cap, 0); cur = 0; }
    mask = (uint32_t)key.size() - 1; count = 0;
    if (++cur == 0) { std::fill(gen.begin(), gen.end(), 0u); cur = 1; }
  }
  bool insert(uint32_t k) {  // true = newly inserted
    if ((size_t)count * 2 > key.size()) grow();
    uint32_t h = (k * 2654435761u) & mask;
    while (gen[h] == cur) { if (key[h] == k) return false; h = (h + 1) & mask; }
    gen[h] = cur; key[h] = k; count++;
    return true;
  }
  void grow() {
    std::vector<uint32_t> ok, og; ok.swap(key); og.swap(gen);
    const uint32_t oc = cur;
    key.assign(ok.size() * 2, 0); gen.assign(ok.size() * 2, 0); mask = (uint32_t)key.size() - 1; cur = 1; count = 0;
    for (size_t i = 0; i < ok.size(); i++) if (og[i] == oc) insert(ok[i]);
  }
};
struct CsrScratch { std::vector<float> qn, rowbuf; std::vector<uint8_t> qlow; std::vector<RItem> res, adm; VisitedTable visited; };

static int csr_search(const CsrGraph& g, const float* query, int k, int ef, int32_t* out_slots, float* out_scores, uint64_t* st) {
  static thread_local CsrScratch S;
  std::vector<float>& qn = S.qn; std::vector<float>& rowbuf = S.rowbuf; std::vector<uint8_t>& qlow = S.qlow;
  qn.resize(g.dim); rowbuf.resize(g.dim);
  const float* q = query;
  if (g.metric == METRIC_COS) { normalize(query, qn.data(), g.dim); q = qn.data(); }
  if (g.quant != Q_NONE) {  // the query is lowered too, then both operands are decoded per pair (f16_vectorstore.go:136, f16_quantization.go:35-45)
    qlow.resize((size_t)g.dim * quant_bytes(g.quant)); lower(g.quant, q, g.dim, qlow.data());
    raise(g.quant, qlow.data(), g.dim, qn.data()); q = qn.data();
  }
  if (g.entry < 0) return 0;
  uint64_t n_dist = 0, n_exp = 0, n_hops = 0;
  const size_t rb = (size_t)g.dim * quant_bytes(g.quant);
  auto D = [&](uint32_t s) {
    n_dist++;
    if (g.quant == Q_NONE) return dist(g.metric, g.order, q, (const float*)(g.rows + (size_t)s * rb), g.dim);
    raise(g.quant, g.rows + (size_t)s * rb, g.dim, rowbuf.data());
    return dist(g.metric, g.order, q, rowbuf.data(), g.dim);
  };
  uint32_t ep = (uint32_t)g.entry; float minD = D(ep);
  for (int l = g.entry_level; l > 0; l--) {
    for (;;) {
      int64_t closest = -1; uint32_t w; const uint32_t* row = csr_row(g, ep, l, w);
      for (uint32_t j = 0; j < w && row[j] != 0xffffffffu; j++) {
        if (csr_deleted(g, row[j])) continue;
        float d = D(row[j]);
        if (d < minD) { minD = d; closest = row[j]; }
      }
      n_hops++;
      if (closest < 0) break;
      ep = (uint32_t)closest;
    }
  }
  std::vector<RItem>& res = S.res; res.clear(); res.reserve((size_t)ef + g.w0 + 1); res.push_back({D(ep), (int32_t)ep, false});
  VisitedTable& visited = S.visited; visited.reset((size_t)ef * g.w0); visited.insert(ep);
  std::vector<RItem>& adm = S.adm;
  for (;;) {
    int ci = -1;
    for (int i = 0; i < (int)res.size(); i++) if (!res[i].expanded) { ci = i; break; }
    if (ci < 0) break;
    res[ci].expanded = true;
    float lowerBound = res.back().d; int free_slots = ef - (int)res.size(); uint32_t c = (uint32_t)res[ci].slot;
    n_exp++; adm.clear();
    uint32_t w; const uint32_t* row = csr_row(g, c, 0, w);
    for (uint32_t j = 0; j < w && row[j] != 0xffffffffu; j++) {
      uint32_t nb = row[j];
      if (csr_deleted(g, nb)) continue;
      if (!visited.insert(nb)) continue;
      float d = D(nb);
      if (free_slots > 0) { adm.push_back({d, (int32_t)nb, false}); free_slots--; }
      else if (d < lowerBound) adm.push_back({d, (int32_t)nb, false});
    }
    for (auto& a : adm) res.insert(std::upper_bound(res.begin(), res.end(), a, ritem_less), a);
    if ((int)res.size() > ef) res.resize(ef);
  }
  int n = std::min(k, (int)res.size());
  for (int i = 0; i < n; i++) { out_slots[i] = res[i].slot; out_scores[i] = res[i].d; }
  if (st) { st[0] += n_dist; st[1] += n_exp; st[2] += n_hops; }
  return n;
}


// ------------------------------------------------------------------------------------------------
// Hnsw.Commit / Hnsw.Load (core/vectorindex/hnsw_commit.go:69-278): big-endian stream
//   [header: config (hnsw_config.go:179-203: u32 algo, f32 levelMult, i32 ef, efC, m, mMax, mMax0), u32 dim, u8 distIdx (1 cosine, 2 l2)]
//   u64 entrypoint id | 16 shards x { u32 count, count x { u64 id, i32 level, dim x f32, metadata } }
//   then per vertex { u64 id, for l = level..0 { u32 n, n x { u64 neighbour id, f32 distance } } }   (deleted neighbours skipped)
// Metadata (metadata.go:31-105) = u16 pairs, each { u8 keylen, key, u16 vallen, msgpack value } — kept as an opaque blob here.
// Go iterates maps in random order; canonical order here = shard by shard (FNV ShardVertex(id,16)), ascending slot inside.
// ------------------------------------------------------------------------------------------------
struct BEWriter {
  std::vector<uint8_t>& b;
  void u8(uint8_t v) { b.push_back(v); }
  void u16(uint16_t v) { b.push_back(v >> 8); b.push_back(v & 0xff); }
  void u32(uint32_t v) { for (int i = 3; i >= 0; i--) b.push_back((v >> (8 * i)) & 0xff); }
  void u64(uint64_t v) { for (int i = 7; i >= 0; i--) b.push_back((v >> (8 * i)) & 0xff); }
  void f32(float f) { u32(f2u(f)); }
};
struct BEReader {
  const uint8_t* p; size_t n, i = 0; bool ok = true;
  bool need(size_t k) { if (i + k > n) { ok = false; return false; } return true; }
  uint8_t u8() { if (!need(1)) return 0; return p[i++]; }
  uint16_t u16() { if (!need(2)) return 0; uint16_t v = (uint16_t)((p[i] << 8) | p[i + 1]); i += 2; return v; }
  uint32_t u32() { if (!need(4)) return 0; uint32_t v = 0; for (int k = 0; k < 4; k++) v = (v << 8) | p[i + k]; i += 4; return v; }
  uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | p[i + k]; i += 8; return v; }
  float f32() { return u2f(u32()); }
};
static void hnsw_commit(Hnsw* h, bool header, std::vector<uint8_t>& out) {
  BEWriter w{out};
  if (header) {
    w.u32((uint32_t)h->cfg.algo); w.f32(h->cfg.levelMultiplier); w.u32((uint32_t)h->cfg.ef); w.u32((uint32_t)h->cfg.efConstruction);
    w.u32((uint32_t)h->cfg.m); w.u32((uint32_t)h->cfg.mMax); w.u32((uint32_t)h->cfg.mMax0);
    w.u32(h->dim); w.u8(h->metric == METRIC_COS ? 1 : 2);
  }
  if (h->len == 0) return;
  w.u64(h->v[h->entry].id);
  std::vector<std::vector<int32_t>> shards(16);
  for (size_t i = 0; i < h->v.size(); i++) if (!h->v[i].deleted) shards[shard_vertex(h->v[i].id, 16)].push_back((int32_t)i);
  for (auto& sh : shards) {
    w.u32((uint32_t)sh.size());
    for (int32_t vi : sh) {
      const Vertex& v = h->v[vi];
      w.u64(v.id); w.u32((uint32_t)(int32_t)v.level);
      for (uint32_t e = 0; e < h->dim; e++) w.f32(v.vec[e]);
      w.u16(0);  // empty Metadata map
    }
  }
  for (auto& sh : shards)
    for (int32_t vi : sh) {
      const Vertex& v = h->v[vi];
      w.u64(v.id);
      for (int l = v.level; l >= 0; l--) {
        uint32_t c = 0;
        for (const Edge& e : v.edges[l]) if (!h->v[e.to].deleted) c++;
        w.u32(c);
        for (const Edge& e : v.edges[l]) { if (h->v[e.to].deleted) continue; w.u64(h->v[e.to].id); w.f32(e.d); }
      }
    }
}
static int hnsw_load(Hnsw* h, bool header, const uint8_t* buf, size_t len) {
  BEReader r{buf, len};
  if (header) {
    h->cfg.algo = (int32_t)r.u32(); h->cfg.levelMultiplier = r.f32(); h->cfg.ef = (int32_t)r.u32(); h->cfg.efConstruction = (int32_t)r.u32();
    h->cfg.m = (int32_t)r.u32(); h->cfg.mMax = (int32_t)r.u32(); h->cfg.mMax0 = (int32_t)r.u32();
    h->dim = r.u32(); uint8_t di = r.u8();
    if (di != 1 && di != 2) return -1;  // InvalidSpaceTypeErr
    h->metric = di == 1 ? METRIC_COS : METRIC_L2;
  }
  h->v.clear(); h->by_id.clear(); h->len = 0; h->entry = -1;
  if (r.i >= len) return r.ok ? 0 : -1;
  uint64_t entry_id = r.u64();
  for (int s = 0; s < 16; s++) {
    uint32_t cnt = r.u32();
    for (uint32_t i = 0; i < cnt && r.ok; i++) {
      Vertex v; v.id = r.u64(); v.level = (int32_t)r.u32(); v.deleted = false;
      v.own.resize(h->dim);
      for (uint32_t e = 0; e < h->dim; e++) v.own[e] = r.f32();  // stored vectors are NOT re-normalised by Load
      uint16_t pairs = r.u16();
      for (uint16_t p = 0; p < pairs && r.ok; p++) { uint8_t kl = r.u8(); r.need(kl); r.i += kl; uint16_t vl = r.u16(); r.need(vl); r.i += vl; }
      v.edges.resize(v.level + 1);
      h->by_id[v.id] = (int32_t)h->v.size();
      h->v.push_back(std::move(v)); h->v.back().vec = h->v.back().own.data(); h->len++;
    }
  }
  if (!r.ok) return -1;
  auto it = h->by_id.find(entry_id);
  h->entry = it == h->by_id.end() ? -1 : it->second;
  for (size_t n = 0; n < h->v.size() && r.ok; n++) {
    uint64_t id = r.u64();
    auto vit = h->by_id.find(id);
    if (vit == h->by_id.end()) return -1;
    Vertex& v = h->v[vit->second];
    for (int l = v.level; l >= 0; l--) {
      uint32_t c = r.u32();
      for (uint32_t j = 0; j < c && r.ok; j++) {
        uint64_t nid = r.u64(); float d = r.f32();
        auto nit = h->by_id.find(nid);
        if (nit == h->by_id.end()) return -1;
        edge_set(v.edges[l], nit->second, d);
      }
    }
  }
  for (auto& v : h->v) v.vec = v.own.data();
  return r.ok ? 0 : -1;
}


// ------------------------------------------------------------------------------------------------
// edge SaveVertex / LoadVertex (edge/none_vectorstore.go:308-516; f16_vectorstore.go:317-532; f8/bf16 twins):
//   16 shards x { u64 count, count x { u64 key, u32 vecLen, vecLen x elem (BE f32 | u16 | u8 = the STORED codes),
//                                      u32 metaCount, metaCount x { u16 keylen, key, u8 tag, value } } }
//   tag 0 int64 BE, 1 {u16 len, bytes}, 2 float64 BE, 3 u8 bool.  Metadata is carried as an opaque blob here.
// Canonical order: shard 0..15, ascending id inside (std::map order).
// ------------------------------------------------------------------------------------------------
static void flat_save(const Flat* f, std::vector<uint8_t>& out) {
  BEWriter w{out};
  for (int s = 0; s < 16; s++) {
    w.u64(f->shards[s].size());
    for (auto& kv : f->shards[s]) {
      w.u64(kv.first); w.u32(f->dim);
      const uint8_t* p = kv.second.data();
      for (uint32_t e = 0; e < f->dim; e++) {
        if (f->quant == Q_NONE) { uint32_t u; std::memcpy(&u, p + 4 * e, 4); w.u32(u); }
        else if (f->quant == Q_F8) w.u8(p[e]);
        else { uint16_t u; std::memcpy(&u, p + 2 * e, 2); w.u16(u); }
      }
      w.u32(0);  // empty metadata map
    }
  }
}
static int flat_load(Flat* f, const uint8_t* buf, size_t len) {
  BEReader r{buf, len};
  for (auto& s : f->shards) s.clear();
  for (int s = 0; s < 16; s++) {
    uint64_t cnt = r.u64();
    for (uint64_t i = 0; i < cnt && r.ok; i++) {
      uint64_t key = r.u64(); uint32_t vl = r.u32();
      if (vl != f->dim) return -1;
      std::vector<uint8_t> codes((size_t)f->dim * quant_bytes(f->quant));
      for (uint32_t e = 0; e < vl; e++) {
        if (f->quant == Q_NONE) { uint32_t u = r.u32(); std::memcpy(codes.data() + 4 * e, &u, 4); }
        else if (f->quant == Q_F8) codes[e] = r.u8();
        else { uint16_t u = r.u16(); std::memcpy(codes.data() + 2 * e, &u, 2); }
      }
      uint32_t mc = r.u32();
      for (uint32_t m = 0; m < mc && r.ok; m++) {
        uint16_t kl = r.u16(); r.need(kl); r.i += kl; uint8_t tag = r.u8();
        if (tag == 0 || tag == 2) { r.need(8); r.i += 8; } else if (tag == 1) { uint16_t sl = r.u16(); r.need(sl); r.i += sl; } else if (tag == 3) { r.need(1); r.i += 1; } else return -1;
      }
      f->shards[s][key] = std::move(codes);  // LoadVertex keeps the shard of the stream, not a re-hash (none_vectorstore.go:505-513)
    }
  }
  return r.ok ? 0 : -1;
}


// ------------------------------------------------------------------------------------------------
// experimental CFLAT: multi-vector weighted FLAT scan (experimental/multi_vector_vertex.go:60-137)
// ------------------------------------------------------------------------------------------------
struct CFlat {
  uint32_t dim, nf; int metric, order;
  std::map<uint64_t, std::vector<float>> v;  // id -> nf * dim floats (normalised per field for cosine)
};
// scoreHelper (experimental, same as edge/edge_helper.go:143-148): cosine ((2-d)/2)*100 ; l2 float32(max(0, float64(100-d)))
static inline float score_helper(float d, int metric) {
  if (metric == METRIC_COS) return ((2.0f - d) / 2.0f) * 100.0f;
  return (float)std::fmax(0.0, (double)(100.0f - d));
}
// MultiVertexSearch (multi_vector_vertex.go:85-137): score = sum over included fields of scoreHelper(Distance(node, q)) *
// (float32(ratio)/100); the queue keeps the K LARGEST scores (min-heap + pop-min — correct for a similarity) and ToSlice
// sorts DESCENDING (experimental/multi_priority_queue.go:54-77).  Canonical tie order: (score, id) descending.
static int cflat_search(const CFlat* c, const float* q /*nf*dim*/, const uint32_t* ratio, const uint8_t* include, int topK,
                        uint64_t* out_ids, float* out_scores) {
  std::vector<float> qn((size_t)c->nf * c->dim);
  for (uint32_t f = 0; f < c->nf; f++) {
    if (include[f] && c->metric == METRIC_COS) normalize(q + (size_t)f * c->dim, qn.data() + (size_t)f * c->dim, c->dim);
    else std::memcpy(qn.data() + (size_t)f * c->dim, q + (size_t)f * c->dim, c->dim * 4);
  }
  std::vector<Scored> all;
  for (auto& kv : c->v) {
    float score = 0.f;
    for (uint32_t f = 0; f < c->nf; f++) {
      if (!include[f]) continue;
      float sim = dist(c->metric, c->order, kv.second.data() + (size_t)f * c->dim, qn.data() + (size_t)f * c->dim, c->dim);
      score += score_helper(sim, c->metric) * ((float)ratio[f] / 100.0f);
    }
    all.push_back({score, kv.first});
  }
  std::sort(all.begin(), all.end(), scored_less);
  size_t n = all.size(), k = std::min((size_t)std::max(topK, 0), n);
  for (size_t i = 0; i < k; i++) { out_ids[i] = all[n - 1 - i].tie; out_scores[i] = all[n - 1 - i].score; }
  return (int)k;
}

static inline uint64_t fnv_mix(uint64_t h, const void* p, size_t n) {
  const uint8_t* b = (const uint8_t*)p;
  for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

// splitmix64 counter RNG
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

}  // namespace

// ================================================================================================
// C API (ctypes)
// ================================================================================================
extern "C" {

float orc_l2(int order, const float* a, const float* b, size_t d) { return dist_l2(order, a, b, d); }
float orc_cosine(int order, const float* a, const float* b, size_t d) { return dist_cos(order, a, b, d); }
float orc_l2sq(int order, const float* a, const float* b, size_t d) { return l2sq(order, a, b, d); }
float orc_manhattan(int order, const float* a, const float* b, size_t d) { return manhattan(order, a, b, d); }
void orc_cosine_parts(int order, const float* a, const float* b, size_t d, float* dot, float* na, float* nb) {
  cos_parts(order, a, b, d, dot, na, nb);
}
void orc_normalize(const float* v, float* out, size_t d) { normalize(v, out, d); }
// one query against n contiguous rows (the contiguous CPU-baseline shape)
void orc_dist_rows(int metric, int order, const float* q, const float* rows, size_t n, size_t d, float* out) {
  for (size_t i = 0; i < n; i++) out[i] = dist(metric, order, q, rows + i * d, d);
}

void orc_f16_encode(const float* in, uint16_t* out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = f32bits_to_f16bits(f2u(in[i])); }
void orc_f16_decode(const uint16_t* in, float* out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = u2f(f16bits_to_f32bits(in[i])); }
void orc_f8_encode(const float* in, uint8_t* out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = f32bits_to_f8bits(f2u(in[i])); }
void orc_f8_decode(const uint8_t* in, float* out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = u2f(f8bits_to_f32bits(in[i])); }
void orc_lower(int quant, const float* v, size_t d, uint8_t* out) { lower(quant, v, d, out); }
void orc_raise(int quant, const uint8_t* in, size_t d, float* out) { raise(quant, in, d, out); }

uint64_t orc_shard_vertex(uint64_t id, uint64_t c) { return shard_vertex(id, c); }

float orc_pq_dot(const float* x, const float* y, size_t n) { return pq_dot(x, y, n); }
float orc_pq_l2sq(const float* x, const float* y, size_t n) { return pq_l2sq(x, y, n); }
float orc_pq_dot_pure(const float* x, const float* y, size_t n) { return pq_dot_pure(x, y, n); }
float orc_pq_l2sq_pure(const float* x, const float* y, size_t n) { return pq_l2sq_pure(x, y, n); }
float orc_pq_hamming(const uint64_t* x, const uint64_t* y, size_t n) { return pq_hamming(x, y, n); }
float orc_pq_jaccard(const uint64_t* x, const uint64_t* y, size_t n) { return pq_jaccard(x, y, n); }

// Go container/heap trace: ops[i] >= 0 -> Push(prios[ops[i]] , value ops[i]); ops[i] == -1 -> Pop.
// Writes the popped values to out_vals (in pop order) then the final array order to out_final.
int orc_heap_trace(int is_max, const float* prios, const int32_t* ops, size_t n_ops, int32_t* out_pops,
                   int32_t* out_final, int32_t* n_final) {
  int np = 0;
  if (is_max) {
    GoHeap<true> hp;
    for (size_t i = 0; i < n_ops; i++) { if (ops[i] >= 0) hp.push({prios[ops[i]], ops[i]}); else if (hp.len()) out_pops[np++] = (int32_t)hp.pop().val; }
    for (int i = 0; i < hp.len(); i++) out_final[i] = (int32_t)hp.a[i].val;
    *n_final = hp.len();
  } else {
    GoHeap<false> hp;
    for (size_t i = 0; i < n_ops; i++) { if (ops[i] >= 0) hp.push({prios[ops[i]], ops[i]}); else if (hp.len()) out_pops[np++] = (int32_t)hp.pop().val; }
    for (int i = 0; i < hp.len(); i++) out_final[i] = (int32_t)hp.a[i].val;
    *n_final = hp.len();
  }
  return np;
}

// Deterministic synthetic data, integer-exact so that any host/device restatement is bit-identical:
// element e of stream `seed` = (sum of twelve 16-bit uniforms - 393210) * 2^-16  (Irwin–Hall ~ N(0,1)).
void orc_fill_normal(uint64_t seed, uint64_t first, float* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint64_t e = first + i;
    uint64_t a = splitmix64(seed ^ (e * 3 + 0)), b = splitmix64(seed ^ (e * 3 + 1) ^ 0x5555555555555555ull),
             c = splitmix64(seed ^ (e * 3 + 2) ^ 0xAAAAAAAAAAAAAAAAull);
    uint32_t s = 0;
    for (int k = 0; k < 4; k++) { s += (a >> (16 * k)) & 0xffff; s += (b >> (16 * k)) & 0xffff; s += (c >> (16 * k)) & 0xffff; }
    out[i] = ((float)(int32_t)s - 393210.0f) * (1.0f / 65536.0f);
  }
}
// HNSW level draw: floor(-ln(U) * mult) (hnsw.go:280-282; gomath/rand.go:42-44; math.go:52-54,60-62)
// with U = (24 random bits + 1) / 2^24 in (0,1] from the counter stream (the reference uses the
// auto-seeded global math/rand, so no sequence is reproducible there).
// Hnsw.RandomLevel for a given uniform draw u (the value rand.Float32() returned): gomath.Floor(-gomath.Log(u) * mult)
int orc_level_from_u(float u, float mult) {
  float lg = (float)std::log((double)u);
  float x = -lg * mult;
  return (int)std::floor((double)x);
}
int orc_level(uint64_t seed, uint64_t i, float mult) {
  uint64_t r = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull));
  float u = (float)((r >> 40) + 1) * (1.0f / 16777216.0f);
  return orc_level_from_u(u, mult);
}

// ---- FLAT ----
void* orc_flat_create(uint32_t dim, int metric, int quant, int order) {
  Flat* f = new Flat(); f->dim = dim; f->metric = metric; f->quant = quant; f->order = order; return f;
}
void orc_flat_destroy(void* h) { delete (Flat*)h; }
int orc_flat_upsert(void* h, const uint64_t* ids, const float* vecs, size_t n) {
  Flat* f = (Flat*)h; for (size_t i = 0; i < n; i++) flat_upsert(f, ids[i], vecs + i * f->dim); return 0;
}
int orc_flat_remove(void* h, const uint64_t* ids, size_t n) {
  Flat* f = (Flat*)h; for (size_t i = 0; i < n; i++) f->shards[shard_vertex(ids[i], 16)].erase(ids[i]); return 0;
}
uint64_t orc_flat_len(void* h) { Flat* f = (Flat*)h; uint64_t n = 0; for (auto& s : f->shards) n += s.size(); return n; }
int orc_flat_get(void* h, uint64_t id, uint8_t* out) {
  Flat* f = (Flat*)h; auto& s = f->shards[shard_vertex(id, 16)]; auto it = s.find(id);
  if (it == s.end()) return -3;
  std::memcpy(out, it->second.data(), it->second.size());
  return 0;
}
int orc_flat_search(void* h, const float* query, int topK, int nearest, int mode, const uint64_t* cand,
                    size_t n_cand, int use_cand, uint64_t* out_ids, float* out_scores) {
  return flat_search((Flat*)h, query, topK, nearest, mode, use_cand != 0, cand, n_cand, out_ids, out_scores);
}


int64_t orc_flat_save(void* h, uint8_t* out, uint64_t cap) {
  std::vector<uint8_t> b; flat_save((Flat*)h, b);
  if (out && cap >= b.size()) std::memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
}
int orc_flat_load(void* h, const uint8_t* buf, uint64_t len) { return flat_load((Flat*)h, buf, len); }


void* orc_cflat_create(uint32_t dim, int metric, uint32_t nf, int order) { CFlat* c = new CFlat(); c->dim = dim; c->nf = nf; c->metric = metric; c->order = order; return c; }
void orc_cflat_destroy(void* h) { delete (CFlat*)h; }
// ChangedVertex (multi_vector_vertex.go:60-75): every field is normalised for cosine
int orc_cflat_upsert(void* h, const uint64_t* ids, const float* vecs, size_t n) {
  CFlat* c = (CFlat*)h; size_t per = (size_t)c->nf * c->dim;
  for (size_t i = 0; i < n; i++) {
    std::vector<float> row(per);
    for (uint32_t f = 0; f < c->nf; f++) {
      const float* src = vecs + i * per + (size_t)f * c->dim;
      if (c->metric == METRIC_COS) normalize(src, row.data() + (size_t)f * c->dim, c->dim); else std::memcpy(row.data() + (size_t)f * c->dim, src, c->dim * 4);
    }
    c->v[ids[i]] = std::move(row);
  }
  return 0;
}
int orc_cflat_remove(void* h, const uint64_t* ids, size_t n) { CFlat* c = (CFlat*)h; for (size_t i = 0; i < n; i++) c->v.erase(ids[i]); return 0; }
int orc_cflat_search(void* h, const float* q, const uint32_t* ratio, const uint8_t* include, int topK, uint64_t* out_ids, float* out_scores) {
  return cflat_search((CFlat*)h, q, ratio, include, topK, out_ids, out_scores);
}

// ---- HNSW ----
void* orc_hnsw_create(uint32_t dim, int metric, int order, const HnswCfg* cfg) {
  Hnsw* h = new Hnsw(); h->dim = dim; h->metric = metric; h->order = order; h->cfg = *cfg;
  if (h->cfg.levelMultiplier == -1) h->cfg.levelMultiplier = 1.0f / (float)std::log((double)(float)h->cfg.m);
  if (h->cfg.mMax == -1) h->cfg.mMax = h->cfg.m;
  if (h->cfg.mMax0 == -1) h->cfg.mMax0 = 2 * h->cfg.m;
  if (h->cfg.algo == 2) h->canon_build = true;   // the diverse mode is defined on the canonical forms only (select_diverse)
  return h;
}
void orc_hnsw_destroy(void* h) { delete (Hnsw*)h; }
void orc_hnsw_get_cfg(void* h, HnswCfg* out) { *out = ((Hnsw*)h)->cfg; }
int orc_hnsw_insert(void* h, uint64_t id, const float* vec, int level) {
  Hnsw* x = (Hnsw*)h;
  if (x->cfg.algo == 1 && x->cfg.extendCandidates) return -4;
  return hnsw_insert(x, id, vec, level, false);
}
void orc_hnsw_set_canonical(void* h, int on) { ((Hnsw*)h)->canon_build = on != 0 || ((Hnsw*)h)->cfg.algo == 2; }
int orc_hnsw_insert_batch(void* h, const uint64_t* ids, const float* vecs, const int32_t* levels, size_t n) {
  return hnsw_insert_batch((Hnsw*)h, ids, vecs, levels, n);
}
int orc_hnsw_remove(void* h, uint64_t id) { return hnsw_remove((Hnsw*)h, id); }
uint64_t orc_hnsw_len(void* h) { return ((Hnsw*)h)->len; }
int64_t orc_hnsw_slots(void* h) { return (int64_t)((Hnsw*)h)->v.size(); }
int32_t orc_hnsw_entry(void* h) { return ((Hnsw*)h)->entry; }
int orc_hnsw_search(void* h, const float* q, int k, int mode, int ef_override, uint64_t* out_ids,
                    float* out_scores, int32_t* out_slots, uint64_t* stats3) {
  Hnsw* x = (Hnsw*)h; x->st = HnswStats{};
  int n = hnsw_search(x, q, k, mode, ef_override, out_ids, out_scores, out_slots);
  if (stats3) { stats3[0] = x->st.n_dist; stats3[1] = x->st.n_exp; stats3[2] = x->st.n_hops; }
  return n;
}
// Export: per-slot arrays + CSR over (slot, level) in canonical (ascending slot) neighbour order.
// Pass null pointers to query sizes: returns total edge count; max_level via *out_max_level.
int64_t orc_hnsw_export(void* h, uint64_t* ids, int32_t* levels, uint8_t* deleted, float* vectors,
                        int64_t* row_offsets /* n_rows+1 where rows = sum(level+1) in slot-major, level-minor order */,
                        int32_t* nbr, float* nbr_dist, int32_t* out_max_level) {
  Hnsw* x = (Hnsw*)h; int64_t ne = 0, row = 0; int32_t ml = 0;
  if (row_offsets) row_offsets[0] = 0;
  for (size_t i = 0; i < x->v.size(); i++) {
    const Vertex& v = x->v[i];
    if (ids) ids[i] = v.id;
    if (levels) levels[i] = v.level;
    if (deleted) deleted[i] = v.deleted;
    if (vectors) std::memcpy(vectors + i * x->dim, v.vec, x->dim * 4);
    ml = std::max(ml, (int32_t)v.level);
    for (int l = 0; l <= v.level; l++) {
      for (const Edge& e : v.edges[l]) { if (nbr) nbr[ne] = e.to; if (nbr_dist) nbr_dist[ne] = e.d; ne++; }
      row++;
      if (row_offsets) row_offsets[row] = ne;
    }
  }
  if (out_max_level) *out_max_level = ml;
  return ne;
}
// Import a graph (same layout as export).  view != 0: `vectors` is NOT copied (caller keeps it alive) —
// the "contiguous" CPU-baseline variant; vectors must already be normalised for cosine.
int orc_hnsw_import(void* h, int64_t n, const uint64_t* ids, const int32_t* levels, const uint8_t* deleted,
                    const float* vectors, const int64_t* row_offsets, const int32_t* nbr, const float* nbr_dist,
                    int32_t entry, int view) {
  Hnsw* x = (Hnsw*)h; x->v.clear(); x->by_id.clear(); x->len = 0;
  x->v.resize(n); int64_t row = 0;
  for (int64_t i = 0; i < n; i++) {
    Vertex& v = x->v[i]; v.id = ids[i]; v.level = levels[i]; v.deleted = deleted ? deleted[i] : 0;
    if (view) v.vec = vectors + i * x->dim;
    else { v.own.assign(vectors + i * x->dim, vectors + (i + 1) * x->dim); v.vec = v.own.data(); }
    v.edges.resize(v.level + 1);
    for (int l = 0; l <= v.level; l++, row++) {
      for (int64_t e = row_offsets[row]; e < row_offsets[row + 1]; e++) v.edges[l].push_back({nbr[e], nbr_dist ? nbr_dist[e] : 0.f});
      std::sort(v.edges[l].begin(), v.edges[l].end(), [](const Edge& a, const Edge& b) { return a.to < b.to; });
    }
    if (!v.deleted) { x->by_id[v.id] = (int32_t)i; x->len++; }
  }
  x->entry = entry;
  return 0;
}

// nq queries, one after another on the calling thread (call from several threads for the all-cores figure).
int orc_csr_search(const void* rows, int quant, const uint32_t* adj0, const uint32_t* upper_off, const uint32_t* adjU,
                   const uint32_t* del_bits, uint32_t w0, uint32_t wu, uint32_t dim, int metric, int order, int32_t entry,
                   int32_t entry_level, const float* queries, size_t nq, int k, int ef, int32_t* out_slots, float* out_scores,
                   int32_t* out_counts, uint64_t* stats3) {
  CsrGraph g{(const uint8_t*)rows, adj0, upper_off, adjU, del_bits, w0, wu, dim, metric, order, entry, entry_level, quant};
  for (size_t i = 0; i < nq; i++)
    out_counts[i] = csr_search(g, queries + i * dim, k, ef, out_slots + i * k, out_scores + i * k, stats3);
  return 0;
}


// returns the stream length; writes it when out != null and cap suffices
int64_t orc_hnsw_commit(void* h, int header, uint8_t* out, uint64_t cap) {
  std::vector<uint8_t> b; hnsw_commit((Hnsw*)h, header != 0, b);
  if (out && cap >= b.size()) std::memcpy(out, b.data(), b.size());
  return (int64_t)b.size();
}
int orc_hnsw_load(void* h, int header, const uint8_t* buf, uint64_t len) { return hnsw_load((Hnsw*)h, header != 0, buf, len); }

uint64_t orc_hnsw_graph_hash(void* h) {
  Hnsw* x = (Hnsw*)h; uint64_t hs = 14695981039346656037ull;
  hs = fnv_mix(hs, &x->entry, 4);
  for (auto& v : x->v) {
    hs = fnv_mix(hs, &v.id, 8); int32_t lv = v.level; hs = fnv_mix(hs, &lv, 4); uint8_t d = v.deleted; hs = fnv_mix(hs, &d, 1);
    for (auto& es : v.edges) { int32_t c = (int32_t)es.size(); hs = fnv_mix(hs, &c, 4); for (auto& e : es) { hs = fnv_mix(hs, &e.to, 4); hs = fnv_mix(hs, &e.d, 4); } }
  }
  return hs;
}

}  // extern "C"

// ================================================================================================
// cpu_baseline drivers (bench.py only).  The reference serves one query per goroutine on the Go scheduler's threads
// (core/core.go:633-667, edge/edge.go:610-690) and `highCpu` splits ONE edge query over 16 goroutines
// (edge/none_vectorstore.go:148-178).  Here: native threads pinned one per allowed CPU, the corpus in a buffer whose pages
// are interleaved over the NUMA nodes (mbind, then a parallel first touch by the pinned threads), so the all-cores number
// is not an artefact of every page sitting on the node of one copy thread.
// ================================================================================================
static std::vector<int> allowed_cpus() {
  cpu_set_t set; CPU_ZERO(&set);
  std::vector<int> r;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) for (int c = 0; c < CPU_SETSIZE; c++) if (CPU_ISSET(c, &set)) r.push_back(c);
  if (r.empty()) { unsigned n = std::thread::hardware_concurrency(); for (unsigned c = 0; c < (n ? n : 1); c++) r.push_back((int)c); }
  return r;
}
// pin policy 1: thread t -> the t-th allowed CPU (dense: the first cores of the first socket, SMT siblings last on Linux's usual
// numbering); policy 2: SPREAD — thread t of T -> allowed[t * n / T]: evenly over sockets / CCDs / memory channels, which is what
// a T-thread run on an n-CPU host should get before it is called "the host's best".
static int pinned_cpu(const std::vector<int>& cpus, int t, int n_threads, int pin) {
  const size_t n = cpus.size();
  if (pin == 2 && n_threads > 0 && (size_t)n_threads < n) return cpus[((size_t)t * n) / (size_t)n_threads % n];
  return cpus[(size_t)t % n];
}
static void pin_self(const std::vector<int>& cpus, int t, int n_threads, int pin) {
  cpu_set_t set; CPU_ZERO(&set); CPU_SET(pinned_cpu(cpus, t, n_threads, pin), &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}
static std::atomic<int> g_pin_policy{1};   // what "pin != 0" means for the drivers below: 1 dense, 2 spread (orc_set_pin_policy)
template <class F> static double run_threads(int n_threads, int pin, F&& body) {
  if (pin) pin = g_pin_policy.load();
  const std::vector<int> cpus = allowed_cpus();
  std::atomic<int> ready{0}; std::atomic<bool> go{false};
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back([&, t]() {
      if (pin) pin_self(cpus, t, n_threads, pin);
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      body(t);
    });
  while (ready.load() < n_threads) std::this_thread::yield();
  auto t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto& x : th) x.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

extern "C" {

int orc_cpu_count(void) { return (int)allowed_cpus().size(); }
// the CPU ids a run of n_threads pinned threads uses under `pin` policy (provenance of the cpu_baseline legs)
void orc_set_pin_policy(int policy) { g_pin_policy.store(policy == 2 ? 2 : 1); }
int orc_pin_map(int n_threads, int pin, int* out, int cap) {
  const std::vector<int> cpus = allowed_cpus();
  int k = 0;
  for (int t = 0; t < n_threads && k < cap; t++) out[k++] = pinned_cpu(cpus, t, n_threads, pin);
  return k;
}
int orc_numa_nodes(void) {  // highest online node + 1 (from sysfs; 1 when unknown)
  FILE* f = fopen("/sys/devices/system/node/online", "r");
  if (!f) return 1;
  char buf[256] = {0}; size_t k = fread(buf, 1, sizeof(buf) - 1, f); fclose(f); (void)k;
  int hi = 0; for (char* p = buf; *p; p++) if (*p >= '0' && *p <= '9') { int v = (int)strtol(p, &p, 10); if (v > hi) hi = v; if (!*p) break; }
  return hi + 1;
}
// Page-interleaved anonymous mapping.  flags out: bit0 = mbind(MPOL_INTERLEAVE) accepted, bit1 = transparent huge pages advised.
void* orc_numa_alloc(size_t bytes, int n_threads, int* out_flags) {
  if (bytes == 0) bytes = 1;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return nullptr;
  int flags = 0;
  const int nodes = orc_numa_nodes();
  if (nodes > 1 && nodes <= 64) {
    unsigned long mask = nodes >= 64 ? ~0ul : ((1ul << nodes) - 1);
    if (syscall(SYS_mbind, p, bytes, 3 /*MPOL_INTERLEAVE*/, &mask, (unsigned long)nodes + 1, 0ul) == 0) flags |= 1;
  }
  if (madvise(p, bytes, MADV_HUGEPAGE) == 0) flags |= 2;
  // parallel first touch in 2 MiB blocks, block b by thread b % T (threads pinned round-robin over the allowed CPUs)
  const size_t blk = 2u << 20, nblk = (bytes + blk - 1) / blk;
  if (n_threads < 1) n_threads = 1;
  run_threads(n_threads, 1, [&](int t) {
    for (size_t b = (size_t)t; b < nblk; b += (size_t)n_threads) {
      uint8_t* q = (uint8_t*)p + b * blk; size_t len = std::min(blk, bytes - b * blk);
      for (size_t o = 0; o < len; o += 4096) q[o] = 0;
    }
  });
  if (out_flags) *out_flags = flags;
  return p;
}
void orc_numa_free(void* p, size_t bytes) { if (p) munmap(p, bytes ? bytes : 1); }

// Streaming-read bandwidth of `bytes` of `p` on n_threads pinned threads (GB/s): the DRAM ceiling the CPU legs are quoted against.
double orc_membw(const void* p, size_t bytes, int n_threads, int reps) {
  if (n_threads < 1) n_threads = 1;
  if (reps < 1) reps = 1;
  std::vector<double> sink((size_t)n_threads, 0.0);
  const size_t n8 = bytes / 32;
  double w = run_threads(n_threads, 1, [&](int t) {
    const v8f* a = (const v8f*)p; v8f acc = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < reps; r++) {
      const size_t lo = n8 * (size_t)t / (size_t)n_threads, hi = n8 * (size_t)(t + 1) / (size_t)n_threads;
      for (size_t i = lo; i < hi; i++) { v8f x; std::memcpy(&x, &a[i], 32); acc += x; }
    }
    sink[(size_t)t] = acc[0] + acc[3];
  });
  double chk = 0; for (double v : sink) chk += v;
  return (chk == 12345.678 ? 0.0 : 1.0) * (double)bytes * reps / w / 1e9;
}

// Hnsw.Search for nq queries on n_threads native threads (one query per thread at a time, work pulled from a shared counter).
// Returns the wall time of the parallel region through *wall_s.  stats3 is summed over all queries.
int orc_csr_search_mt(const void* rows, int quant, const uint32_t* adj0, const uint32_t* upper_off, const uint32_t* adjU,
                      const uint32_t* del_bits, uint32_t w0, uint32_t wu, uint32_t dim, int metric, int order, int32_t entry,
                      int32_t entry_level, const float* queries, size_t nq, int k, int ef, int32_t* out_slots, float* out_scores,
                      int32_t* out_counts, uint64_t* stats3, int n_threads, int pin, double* wall_s) {
  CsrGraph g{(const uint8_t*)rows, adj0, upper_off, adjU, del_bits, w0, wu, dim, metric, order, entry, entry_level, quant};
  if (n_threads < 1) n_threads = 1;
  std::atomic<size_t> next{0};
  std::vector<uint64_t> st((size_t)n_threads * 3, 0);
  double w = run_threads(n_threads, pin, [&](int t) {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= nq) break;
      out_counts[i] = csr_search(g, queries + i * dim, k, ef, out_slots + i * k, out_scores + i * k, &st[(size_t)t * 3]);
    }
  });
  if (stats3) for (int t = 0; t < n_threads; t++) for (int j = 0; j < 3; j++) stats3[j] += st[(size_t)t * 3 + j];
  if (wall_s) *wall_s = w;
  return 0;
}

// VertexSearch over CONTIGUOUS stored rows ("contiguous" variant of BASELINE.md §2; the reference walks 16 Go maps of
// heap-scattered vectors, edge/none_vectorstore.go:136-147).  Per (query, row): Quantization.Similarity exactly as the
// reference — shape 0 decodes BOTH operands per pair (f16_quantization.go:35-45: the value receiver defeats its buffer cache),
// shape 1 decodes the query once — then the bounded queue.  The queue keeps the K extreme (score, slot) pairs in the
// canonical order the GPU uses, so a sample can be compared bit-for-bit.
//   split == 1  : one query per thread (concurrent RPC handlers), queries pulled from a shared counter;
//   split == S>1: `highCpu` — every query is scanned by S threads over S contiguous row ranges with local queues, merged
//                 by thread 0 (none_vectorstore.go:148-178); n_threads must equal S.
struct TopK {
  int k; bool nearest; std::vector<Scored> h;  // heap with the WORST kept element on top
  bool worse(const Scored& a, const Scored& b) const { return nearest ? scored_less(b, a) : scored_less(a, b); }  // a worse than b
  void add(const Scored& x) {
    auto cmp = [&](const Scored& a, const Scored& b) { return worse(b, a); };  // max-heap on "worse"
    if ((int)h.size() < k) { h.push_back(x); std::push_heap(h.begin(), h.end(), cmp); return; }
    if (k == 0 || !worse(h.front(), x)) return;
    std::pop_heap(h.begin(), h.end(), cmp); h.back() = x; std::push_heap(h.begin(), h.end(), cmp);
  }
};
int orc_flat_scan_mt(const void* rows_v, int quant, uint64_t n, uint32_t dim, int metric, int order, const float* queries, size_t nq,
                     int k, int nearest, int shape, int split, int n_threads, int pin, uint64_t* out_slots, float* out_scores,
                     int32_t* out_counts, double* wall_s) {
  const uint8_t* rows = (const uint8_t*)rows_v;
  const size_t rb = (size_t)dim * quant_bytes(quant);
  if (n_threads < 1) n_threads = 1;
  if (split < 1) split = 1;
  if (split > 1 && n_threads != split) return -1;
  // per-query prepared operands: Normalize (cosine), Lower (f16_vectorstore.go:136)
  std::vector<float> qn((size_t)nq * dim); std::vector<uint8_t> qlow((size_t)nq * rb);
  for (size_t i = 0; i < nq; i++) {
    if (metric == METRIC_COS) normalize(queries + i * dim, &qn[i * dim], dim); else std::memcpy(&qn[i * dim], queries + i * dim, (size_t)dim * 4);
    lower(quant, &qn[i * dim], dim, &qlow[i * rb]);
    if (quant != Q_NONE) raise(quant, &qlow[i * rb], dim, &qn[i * dim]);  // the decoded query (shape 1 uses it directly)
  }
  // shape 2 = the reference's MEMORY shape as well (none_vectorstore.go:40-47, 135-178): 16 maps id -> ENode, every stored vector
  // its own heap allocation, scanned in map order; Similarity allocates both decoded operands per pair (f16_quantization.go:35-45).
  // Built here, outside the timed region; with split = 16 thread t scans map t, which is what highCpu does.
  std::vector<std::unordered_map<uint64_t, std::vector<uint8_t>>> shards;
  if (shape == 2) {
    shards.resize(16);
    for (uint64_t r = 0; r < n; r++) shards[(size_t)shard_vertex(r, 16)].emplace(r, std::vector<uint8_t>(rows + r * rb, rows + (r + 1) * rb));
    if (split != 1 && split != 16) return -1;
  }
  auto scan_shard = [&](size_t qi, size_t sh, TopK& tk) {
    const float* qd = &qn[qi * dim]; const uint8_t* ql = &qlow[qi * rb];
    for (const auto& kv : shards[sh]) {
      float sc;
      if (quant == Q_NONE) sc = dist(metric, order, qd, (const float*)kv.second.data(), dim);
      else { std::vector<float> a(dim), b(dim); raise(quant, ql, dim, a.data()); raise(quant, kv.second.data(), dim, b.data()); sc = dist(metric, order, a.data(), b.data(), dim); }
      tk.add({sc, kv.first});
    }
  };
  auto scan = [&](size_t qi, uint64_t lo, uint64_t hi, TopK& tk, float* bx, float* by) {
    if (shape == 2) {  // lo/hi select whole maps: [0,n) = all sixteen, the t-th sixteenth = map t
      if (lo == 0 && hi == n) { for (size_t sh = 0; sh < 16; sh++) scan_shard(qi, sh, tk); }
      else scan_shard(qi, (size_t)((lo * 16 + n / 2) / (n ? n : 1)), tk);
      return;
    }
    const float* qd = &qn[qi * dim]; const uint8_t* ql = &qlow[qi * rb];
    for (uint64_t r = lo; r < hi; r++) {
      const uint8_t* row = rows + r * rb; float sc;
      if (quant == Q_NONE) sc = dist(metric, order, qd, (const float*)row, dim);
      else if (shape == 0) { raise(quant, ql, dim, bx); raise(quant, row, dim, by); sc = dist(metric, order, bx, by, dim); }
      else { raise(quant, row, dim, by); sc = dist(metric, order, qd, by, dim); }
      tk.add({sc, r});
    }
  };
  auto emit = [&](size_t qi, TopK& tk) {
    std::sort(tk.h.begin(), tk.h.end(), scored_less);
    out_counts[qi] = (int32_t)tk.h.size();
    for (size_t j = 0; j < tk.h.size(); j++) { out_slots[qi * k + j] = tk.h[j].tie; out_scores[qi * k + j] = tk.h[j].score; }
  };
  double w;
  if (split == 1) {
    std::atomic<size_t> next{0};
    w = run_threads(n_threads, pin, [&](int) {
      std::vector<float> bx(dim), by(dim);
      for (;;) {
        size_t qi = next.fetch_add(1);
        if (qi >= nq) break;
        TopK tk{k, nearest != 0, {}};
        scan(qi, 0, n, tk, bx.data(), by.data());
        emit(qi, tk);
      }
    });
  } else {
    std::vector<TopK> local((size_t)split, TopK{k, nearest != 0, {}});
    pthread_barrier_t bar; pthread_barrier_init(&bar, nullptr, (unsigned)split);
    w = run_threads(split, pin, [&](int t) {
      std::vector<float> bx(dim), by(dim);
      for (size_t qi = 0; qi < nq; qi++) {
        local[(size_t)t].h.clear();
        scan(qi, n * (uint64_t)t / (uint64_t)split, n * (uint64_t)(t + 1) / (uint64_t)split, local[(size_t)t], bx.data(), by.data());
        pthread_barrier_wait(&bar);
        if (t == 0) { TopK g{k, nearest != 0, {}}; for (auto& l : local) for (auto& e : l.h) g.add(e); emit(qi, g); }
        pthread_barrier_wait(&bar);
      }
    });
    pthread_barrier_destroy(&bar);
  }
  if (wall_s) *wall_s = w;
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Product quantiser: codebook LUT + ADC scan (BASELINE.json north_star "PQ-codebook kernels").
//
// PARITY UNPINNED — THIS SECTION IS A DEFINITION, not a restatement of reference code.  What the reference holds:
//   * the parameters — pkg/models/hnsw_common.go:20-33: NumCentroids in [2, 256] (one uint8 code per sub-vector),
//     NumSubVectors >= 2 (the paper's m), TriggerThreshold = size of the training sample;
//   * the arithmetic — pkg/distancepq: euclideanDistance = SQUARED L2 (distance.go:30-32), cosineDistance = 1 - dot
//     (:40-42), dotProductDistance = -dot (:36-38), over asm.SquaredEuclideanDistance / asm.Dot (asm/euclidean.s:7-65,
//     asm/dot.s:7-55: FMA, 4 x 8-lane accumulators, scalar-FMA tail — pq_l2sq / pq_dot above), selected at init on AVX2+FMA hosts
//     (distance_amd64.go:28-36);
//   * the call shape — playground/hnswpq_verification.go:69-73, 90-105, 154, 190-199 (NumSubVectors 32, NumCentroids 256, train,
//     Fit, then search with the float vectors dropped) — but the package it drives, pkg/hnswpq, is NOT in the tree, and nothing else
//     imports pkg/distancepq (SURVEY §0 finding 9).
// So the scan is defined from the distancepq arithmetic in the form every product quantiser over such kernels takes (Jegou et al.,
// the paper the parameter comments cite):
//   codebooks [m][C][dsub] f32, dim = m * dsub; a stored vector is m uint8 codes;
//   Encode: code[j] = argmin_c SquaredEuclideanDistance(x_j, centroid[j][c]), scanning c upwards from minDist = MaxFloat32 with a
//           strict `<` (ties -> the lowest c; a NaN distance is never chosen);
//   LUT   : lut[j][c] = distFn(q_j, centroid[j][c]), distFn = the store's distancepq function (metric 0 cosineDistance,
//           1 euclideanDistance, 2 dotProductDistance);
//   score : dist = 0; for j = 0..m-1: dist += lut[j][code[j]]           (f32, in j order: "var dist float32; dist += ...")
//   top-k : the k smallest in the canonical (score bits, id) order used everywhere in this oracle;
//   Train : Lloyd iterations with everything deterministic (centroid c of sub-space j starts as sub-vector j of training vector c;
//           assignment = Encode; update = f32 sum in training-index order / float32(count); an empty cluster keeps its centroid).
// ------------------------------------------------------------------------------------------------
enum { PQ_COSINE = 0, PQ_EUCLIDEAN = 1, PQ_DOT = 2 };
static inline float pq_fn(int metric, const float* x, const float* y, size_t len) {
  if (metric == PQ_EUCLIDEAN) return pq_l2sq(x, y, len);          // euclideanDistance (squared)
  if (metric == PQ_COSINE) return 1 - pq_dot(x, y, len);           // cosineDistance
  return -pq_dot(x, y, len);                                       // dotProductDistance
}
static void pq_lut(int metric, const float* cb, int m, int C, int dsub, const float* q, float* lut /* [m][C] */) {
  for (int j = 0; j < m; j++)
    for (int c = 0; c < C; c++) lut[(size_t)j * C + c] = pq_fn(metric, q + (size_t)j * dsub, cb + ((size_t)j * C + c) * dsub, dsub);
}
static void pq_encode(const float* cb, int m, int C, int dsub, const float* x, uint8_t* code) {
  for (int j = 0; j < m; j++) {
    float minDist = 3.40282346638528859811704183484516925440e+38f;  // math.MaxFloat32
    int best = 0;
    for (int c = 0; c < C; c++) {
      const float d = pq_l2sq(x + (size_t)j * dsub, cb + ((size_t)j * C + c) * dsub, dsub);
      if (d < minDist) { minDist = d; best = c; }
    }
    code[j] = (uint8_t)best;
  }
}
static inline float pq_adc(const float* lut, int m, int C, const uint8_t* code) {
  float dist = 0;
  for (int j = 0; j < m; j++) dist += lut[(size_t)j * C + code[j]];
  return dist;
}
// The table distance of the product-quantised WALK (a definition of ours, like the walk itself — "Product-quantised HNSW" below): the code row is
// P = ceil(m / 16) 16-byte pieces; S_lo sums the entries of the first ceil(P / 2) pieces (j < JS = 16 ceil(P / 2)) in j order from +0.0, S_hi the
// entries of the rest (JS <= j < m) in j order from +0.0; d = S_lo + S_hi, one f32 add.  (Round 6: on the GPU a neighbour is owned by a LANE PAIR, and
// each lane now sums half the row; rounds 4-5 had one lane sum the whole row in j order while its partner idled — a different, equally arbitrary order.)
static inline float pq_adc_walk(const float* lut, int m, int C, const uint8_t* code) {
  const int P = (m + 15) / 16, JS = 16 * ((P + 1) / 2);
  float lo = 0.f, hi = 0.f;
  for (int j = 0; j < m && j < JS; j++) lo += lut[(size_t)j * C + code[j]];
  for (int j = JS; j < m; j++) hi += lut[(size_t)j * C + code[j]];
  return lo + hi;
}

extern "C" {

void orc_pq_lut(int metric, const float* codebooks, int m, int C, int dsub, const float* query, float* out) {
  pq_lut(metric, codebooks, m, C, dsub, query, out);
}
void orc_pq_encode(const float* codebooks, int m, int C, int dsub, const float* vecs, size_t n, uint8_t* out_codes) {
  for (size_t i = 0; i < n; i++) pq_encode(codebooks, m, C, dsub, vecs + i * (size_t)m * dsub, out_codes + i * (size_t)m);
}
void orc_pq_adc(const float* lut, int m, int C, const uint8_t* codes, size_t n, float* out) {
  for (size_t i = 0; i < n; i++) out[i] = pq_adc(lut, m, C, codes + i * (size_t)m);
}
// Lloyd iterations as defined above; codebooks is in/out only in the sense that it is fully overwritten.  n >= C.
int orc_pq_train(float* codebooks, int m, int C, int dsub, const float* vecs, size_t n, int iters) {
  if ((size_t)C > n) return -1;
  const size_t dim = (size_t)m * dsub;
  for (int j = 0; j < m; j++)
    for (int c = 0; c < C; c++) std::memcpy(codebooks + ((size_t)j * C + c) * dsub, vecs + (size_t)c * dim + (size_t)j * dsub, (size_t)dsub * 4);
  std::vector<uint8_t> codes(n * (size_t)m);
  for (int it = 0; it < iters; it++) {
    for (size_t i = 0; i < n; i++) pq_encode(codebooks, m, C, dsub, vecs + i * dim, &codes[i * m]);
    for (int j = 0; j < m; j++)
      for (int c = 0; c < C; c++) {
        uint32_t cnt = 0;
        for (size_t i = 0; i < n; i++) cnt += codes[i * m + j] == c;
        if (!cnt) continue;
        for (int e = 0; e < dsub; e++) {
          float s = 0;
          for (size_t i = 0; i < n; i++) if (codes[i * m + j] == c) s += vecs[i * dim + (size_t)j * dsub + e];
          codebooks[((size_t)j * C + c) * dsub + e] = s / (float)cnt;
        }
      }
  }
  return 0;
}
// The ADC search over contiguous codes [n][m] (row-major) on n_threads pinned threads, one query per thread: per query the LUT, then
// one pass over the codes with the bounded queue in the canonical (score, id) order.  ids == NULL: id = slot.
int orc_pq_search_mt(int metric, const float* codebooks, int m, int C, int dsub, const uint8_t* codes, const uint64_t* ids, uint64_t n,
                     const float* queries, size_t nq, int k, int n_threads, int pin, uint64_t* out_ids, float* out_scores,
                     int32_t* out_counts, double* wall_s) {
  if (n_threads < 1) n_threads = 1;
  const size_t dim = (size_t)m * dsub;
  std::atomic<size_t> next{0};
  double w = run_threads(n_threads, pin, [&](int) {
    std::vector<float> lut((size_t)m * C);
    for (;;) {
      size_t qi = next.fetch_add(1);
      if (qi >= nq) break;
      pq_lut(metric, codebooks, m, C, dsub, queries + qi * dim, lut.data());
      TopK tk{k, true, {}};
      for (uint64_t r = 0; r < n; r++) tk.add({pq_adc(lut.data(), m, C, codes + r * (size_t)m), ids ? ids[r] : r});
      std::sort(tk.h.begin(), tk.h.end(), scored_less);
      out_counts[qi] = (int32_t)tk.h.size();
      for (size_t j = 0; j < tk.h.size(); j++) { out_ids[qi * k + j] = tk.h[j].tie; out_scores[qi * k + j] = tk.h[j].score; }
    }
  });
  if (wall_s) *wall_s = w;
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Product-quantised HNSW — a DEFINITION (the reference's pkg/hnswpq, driven by playground/hnswpq_verification.go:69-105, is not in its
// tree), assembled from restatements that ARE tied to reference code: Hnsw.Search (core/vectorindex/hnsw.go:243-278) in the canonical
// closed form of csr_search above, and the quantiser's Encode / table / score of "Product quantiser" above
// (pkg/distancepq/distance.go:30-42):
//   codes    code_v = Encode(stored row of v as the index's distance sees it: normalised for cosine, lowered and raised for 2-byte rows)
//   d(q, v)  = pq_adc_walk(lut16(q'), code_v) (two half-row sums added: see pq_adc_walk), q' = the query as the index's distance sees it, lut16 = the quantiser's table (its distancepq
//            function) with every entry rounded to binary16 (round to nearest even — the f16 codec's own rounding) and read back as f32:
//            d only ranks, the answers carry exact distances, and a 2-byte table doubles the GPU kernel's resident traversals.  Before the
//            rounding every entry is multiplied by 2^-k, k = the smallest integer >= 0 with (largest entry) * 2^-k <= 32768 (exact; k = 0 for
//            unit-scale data): un-normalised Euclidean data would otherwise round to +Inf and every vertex would be equally far (round 6)
//   walk     csr_search with d in place of Distance(): entrypoint (hnsw.go:253), greedyClosestNeighbor per upper level (:320-343),
//            searchLevel(ef) on level 0 (:345-389) — admission rule, canonical neighbour order and (d, slot) ties unchanged
//   re-rank  r = min(max(rerank, k), |result set|) (rerank = 0: the whole set): the r nearest by d are re-scored with the index's
//            exact distance; the k smallest by (exact score bits, slot) are returned with the exact scores.
// Counters: st[0] table-distance evaluations (level 0: those that passed the bound and the visited test — see BOUNDED VISITING), st[1] expansions,
// st[2] greedy hops, st[3] exact evaluations.
// ------------------------------------------------------------------------------------------------
static int csr_search_pq(const CsrGraph& g, const uint8_t* codes, const float* cb, int m, int C, int dsub, int pq_metric, const float* query, int k, int ef,
                         int rerank, int32_t* out_slots, float* out_scores, uint64_t* st) {
  std::vector<float> qn(g.dim), rowbuf(g.dim), lut((size_t)m * C);
  const float* q = query;
  if (g.metric == METRIC_COS) { normalize(query, qn.data(), g.dim); q = qn.data(); }
  if (g.quant != Q_NONE) {
    std::vector<uint8_t> qlow((size_t)g.dim * quant_bytes(g.quant)); lower(g.quant, q, g.dim, qlow.data());
    raise(g.quant, qlow.data(), g.dim, qn.data()); q = qn.data();
  }
  if (g.entry < 0) return 0;
  pq_lut(pq_metric, cb, m, C, dsub, q, lut.data());
  {  // table scale: k = the smallest integer >= 0 with M * 2^-k <= 32768, M = the largest entry (k = 0 for unit-scale data); entries * 2^-k are exact
    float M = 0.f;
    for (float v : lut) if (v > M) M = v;
    float sc = 1.0f;
    if (M == M && M < std::numeric_limits<float>::infinity()) while (M * sc > 32768.0f) sc *= 0.5f;
    for (float& v : lut) v = v * sc;
  }
  for (float& v : lut) v = u2f(f16bits_to_f32bits(f32bits_to_f16bits(f2u(v))));   // the walk's table entries are binary16 (round to nearest even)
  uint64_t n_dist = 0, n_exp = 0, n_hops = 0, n_exact = 0;
  auto D = [&](uint32_t s) { n_dist++; return pq_adc_walk(lut.data(), m, C, codes + (size_t)s * m); };
  const size_t rb = (size_t)g.dim * quant_bytes(g.quant);
  auto X = [&](uint32_t s) {
    n_exact++;
    if (g.quant == Q_NONE) return dist(g.metric, g.order, q, (const float*)(g.rows + (size_t)s * rb), g.dim);
    raise(g.quant, g.rows + (size_t)s * rb, g.dim, rowbuf.data());
    return dist(g.metric, g.order, q, rowbuf.data(), g.dim);
  };
  uint32_t ep = (uint32_t)g.entry; float minD = D(ep);
  for (int l = g.entry_level; l > 0; l--) {
    for (;;) {
      int64_t closest = -1; uint32_t w; const uint32_t* row = csr_row(g, ep, l, w);
      for (uint32_t j = 0; j < w && row[j] != 0xffffffffu; j++) {
        if (csr_deleted(g, row[j])) continue;
        float d = D(row[j]);
        if (d < minD) { minD = d; closest = row[j]; }
      }
      n_hops++;
      if (closest < 0) break;
      ep = (uint32_t)closest;
    }
  }
  std::vector<RItem> res; res.reserve((size_t)ef + g.w0 + 1); res.push_back({D(ep), (int32_t)ep, false});
  VisitedTable visited; visited.reset((size_t)ef * g.w0); visited.insert(ep);
  std::vector<RItem> adm;
  for (;;) {
    int ci = -1;
    for (int i = 0; i < (int)res.size(); i++) if (!res[i].expanded) { ci = i; break; }
    if (ci < 0) break;
    res[ci].expanded = true;
    float lowerBound = res.back().d; int free_slots = ef - (int)res.size(); uint32_t c = (uint32_t)res[ci].slot;
    n_exp++; adm.clear();
    uint32_t w; const uint32_t* row = csr_row(g, c, 0, w);
    // BOUNDED VISITING (round 6; part of this definition, not of the reference): a table distance costs a few lookups, a visited test a memory request.  Once the
    // result set is full at a pop it stays full and its worst member only ever improves, so a neighbour with d >= lowerBound can never be admitted, now or at
    // any later encounter: it is skipped BEFORE the visited test — neither marked nor counted.  The sequence of result sets (hence ids, scores, n_exp) is
    // exactly that of the unbounded walk; st[0] counts the evaluations that passed the bound and the visited test (every fresh one while the set fills up).
    const bool full_at_pop = free_slots == 0;
    for (uint32_t j = 0; j < w && row[j] != 0xffffffffu; j++) {
      uint32_t nb = row[j];
      if (csr_deleted(g, nb)) continue;
      if (full_at_pop) {
        float d = pq_adc_walk(lut.data(), m, C, codes + (size_t)nb * m);   // not counted
        if (!(d < lowerBound)) continue;
        if (!visited.insert(nb)) continue;
        n_dist++;
        adm.push_back({d, (int32_t)nb, false});
        continue;
      }
      if (!visited.insert(nb)) continue;
      float d = D(nb);
      if (free_slots > 0) { adm.push_back({d, (int32_t)nb, false}); free_slots--; }
      else if (d < lowerBound) adm.push_back({d, (int32_t)nb, false});
    }
    for (auto& a : adm) res.insert(std::upper_bound(res.begin(), res.end(), a, ritem_less), a);
    if ((int)res.size() > ef) res.resize(ef);
  }
  int r = rerank == 0 ? (int)res.size() : std::max(rerank, k);
  r = std::min(r, (int)res.size());
  std::vector<RItem> ex((size_t)r);
  for (int i = 0; i < r; i++) ex[(size_t)i] = {X((uint32_t)res[(size_t)i].slot), res[(size_t)i].slot, false};
  std::sort(ex.begin(), ex.end(), ritem_less);
  const int n = std::min(k, r);
  for (int i = 0; i < n; i++) { out_slots[i] = ex[(size_t)i].slot; out_scores[i] = ex[(size_t)i].d; }
  if (st) { st[0] += n_dist; st[1] += n_exp; st[2] += n_hops; st[3] += n_exact; }
  return n;
}

extern "C" {

// codes: [n][m] row-major (slot order); codebooks [m][C][dsub]; stats4 summed over the queries; n_threads native threads (one query each)
int orc_csr_search_pq_mt(const void* rows, int quant, const uint32_t* adj0, const uint32_t* upper_off, const uint32_t* adjU,
                         const uint32_t* del_bits, uint32_t w0, uint32_t wu, uint32_t dim, int metric, int order, int32_t entry,
                         int32_t entry_level, const uint8_t* codes, const float* codebooks, int m, int C, int pq_metric, const float* queries,
                         size_t nq, int k, int ef, int rerank, int32_t* out_slots, float* out_scores, int32_t* out_counts, uint64_t* stats4,
                         int n_threads, int pin, double* wall_s) {
  CsrGraph g{(const uint8_t*)rows, adj0, upper_off, adjU, del_bits, w0, wu, dim, metric, order, entry, entry_level, quant};
  if (n_threads < 1) n_threads = 1;
  if (m <= 0 || dim % (uint32_t)m) return -1;
  const int dsub = (int)(dim / (uint32_t)m);
  std::atomic<size_t> next{0};
  std::vector<uint64_t> st((size_t)n_threads * 4, 0);
  double w = run_threads(n_threads, pin, [&](int t) {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= nq) break;
      out_counts[i] = csr_search_pq(g, codes, codebooks, m, C, dsub, pq_metric, queries + i * dim, k, ef, rerank, out_slots + i * k, out_scores + i * k, &st[(size_t)t * 4]);
    }
  });
  if (stats4) for (int t = 0; t < n_threads; t++) for (int j = 0; j < 4; j++) stats4[j] += st[(size_t)t * 4 + j];
  if (wall_s) *wall_s = w;
  return 0;
}

}  // extern "C"
