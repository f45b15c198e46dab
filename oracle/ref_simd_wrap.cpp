// ref_simd_wrap.cpp — build recipe glue (ours) for the REFERENCE's own SIMD kernels.
// TEST INFRASTRUCTURE ONLY.  Compiles /root/reference/pkg/distance/simd/cpp/{avx,sse}.cpp from where
// they lie (paths come from the Makefile; nothing is copied into this repository) and exposes them
// with C linkage so tests can pin the oracle's summation orders against the real thing.
//
// Two source-level incompatibilities of those files with Linux g++ are neutralised by macros that
// are active only while the reference text is being parsed:
//   * `inline float abs(float)` collides with <cstdlib>'s std::abs      -> renamed
//   * `_mm256_load_ps/_mm_load_ps` need 32/16-byte alignment, whereas the shipped Go assembly
//     (pkg/distance/simd/avx/AVX_amd64.s) uses unaligned vmovups        -> mapped to loadu
#include <immintrin.h>
#include <cstddef>

#define abs coltt_ref_abs
#define _mm256_load_ps _mm256_loadu_ps
#define _mm_load_ps _mm_loadu_ps
namespace ref_avx {
#include REF_AVX_CPP
}
namespace ref_sse {
#include REF_SSE_CPP
}
#undef abs
#undef _mm256_load_ps
#undef _mm_load_ps

extern "C" {
// order: 0 = avx.cpp, 1 = sse.cpp
void ref_l2sq(int order, size_t len, const float* a, const float* b, float* result) {
  if (order == 0) ref_avx::euclidean_distance_squared(len, (float*)a, (float*)b, result);
  else ref_sse::euclidean_distance_squared(len, (float*)a, (float*)b, result);
}
void ref_manhattan(int order, size_t len, const float* a, const float* b, float* result) {
  if (order == 0) ref_avx::manhattan_distance(len, (float*)a, (float*)b, result);
  else ref_sse::manhattan_distance(len, (float*)a, (float*)b, result);
}
void ref_cos_dot_norm(int order, size_t len, const float* a, const float* b, float* dot, float* norm_sq) {
  if (order == 0) ref_avx::cosine_similarity_dot_norm(len, (float*)a, (float*)b, dot, norm_sq);
  else ref_sse::cosine_similarity_dot_norm(len, (float*)a, (float*)b, dot, norm_sq);
}
}
