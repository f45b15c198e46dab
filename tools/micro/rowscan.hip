// rowscan.hip — the library's OWN exact-order distance code (coltt_amd/csrc/exact.hpp: pair_distance, one lane pair per row, the
// AVX summation order of pkg/distance/simd/cpp/avx.cpp) over RANDOM rows of a large table, without the graph walk around it: what
// the row phase of the HNSW kernels can deliver by itself, next to tools/micro/gather.hip (loads only).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I coltt_amd/csrc -o rowscan tools/micro/rowscan.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "exact.hpp"

using namespace coltt::dev;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <int QUANT, int U>
__global__ __launch_bounds__(64) void rowscan_kernel(const uint8_t* __restrict__ table, uint64_t nrows, uint32_t stride, int dim, uint32_t passes,
                                                     uint32_t active_pairs, uint32_t* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) float qs[];
  const int lane = threadIdx.x, half = lane & 1, p = lane >> 1;
  for (int e = lane; e < dim; e += 64) qs[e] = 0.01f * (float)((e * 37 + blockIdx.x) % 101) - 0.5f;
  __syncthreads();
  uint32_t acc = 0;
  for (uint32_t it = 0; it < passes; it++) {
    const uint64_t r = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)it << 8) ^ (uint64_t)p) % nrows;
    float d = 0.f;
    if ((uint32_t)p < active_pairs) d = pair_distance<M_COS, QUANT, U>(table + r * (uint64_t)stride, qs, dim, 1.0f, 1.0f, half);
    acc ^= __float_as_uint(d);
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = 1;
}

template <int QUANT, int U>
void run(const uint8_t* table, uint64_t bytes, int dim, int w, uint32_t active, uint32_t* sink, double gib) {
  const uint32_t stride = (uint32_t)dim * (QUANT == Q_NONE ? 4 : 2);
  const uint64_t nrows = bytes / stride;
  const uint32_t grid = 256u * (uint32_t)w;
  const double bytes_per_pass = (double)grid * active * stride;
  uint32_t passes = (uint32_t)(30e9 / bytes_per_pass);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto kern = rowscan_kernel<QUANT, U>;
  const size_t lds = (size_t)dim * 4;
  kern<<<grid, 64, lds>>>(table, nrows, stride, dim, passes / 4, active, sink);
  CK(hipDeviceSynchronize());
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    kern<<<grid, 64, lds>>>(table, nrows, stride, dim, passes, active, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = bytes_per_pass * passes / (ms * 1e-3) / 1e9;
    if (gbs > best) best = gbs;
  }
  printf("{\"kernel\": \"pair_distance (exact.hpp)\", \"table_GiB\": %.1f, \"rows\": \"%s\", \"dim\": %d, \"burst_U\": %d, \"waves_per_cu\": %d, \"active_pairs\": %u, "
         "\"GBps\": %.0f, \"frac_of_8TBs\": %.3f}\n", gib, QUANT == Q_NONE ? "f32" : "f16", dim, U, w, active, best, best / 8000.0);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 15.0;
  const int which = argc > 2 ? atoi(argv[2]) : 3;   // 1 f16, 2 f32, 3 both
  const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
  uint8_t* table; uint32_t* sink;
  CK(hipMalloc(&table, bytes)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(table, 0x3c, bytes)); CK(hipMemset(sink, 0, 1 << 20));   // 0x3c3c = 1.06 as binary16, a small normal as f32
  CK(hipDeviceSynchronize());
  const int dim = 768;
  for (uint32_t act : {32u, 20u}) {
    for (int w : {4, 6, 8, 12}) {
      if (which & 1) {
        run<Q_F16, 12>(table, bytes, dim, w, act, sink, gib);
        run<Q_F16, 16>(table, bytes, dim, w, act, sink, gib);
        run<Q_F16, 24>(table, bytes, dim, w, act, sink, gib);
        run<Q_F16, 32>(table, bytes, dim, w, act, sink, gib);
      }
      if (which & 2) {
        run<Q_NONE, 8>(table, bytes, dim, w, act, sink, gib);
        run<Q_NONE, 12>(table, bytes, dim, w, act, sink, gib);
        run<Q_NONE, 16>(table, bytes, dim, w, act, sink, gib);
        run<Q_NONE, 24>(table, bytes, dim, w, act, sink, gib);
      }
    }
  }
  return 0;
}
