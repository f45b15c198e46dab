// gather.hip — what the memory system of an MI355X delivers for RANDOM fixed-size row reads (the access pattern of the HNSW walk:
// core/vectorindex/hnsw.go:345-389 evaluates Distance(query, neighbour.vector) for neighbours scattered over the whole collection),
// as a function of HOW a wave covers a row.  No arithmetic, no dependent chain, no visited set: the ceiling the walk kernels are
// measured against.
//
//   G lanes per row: a wave-wide 16-byte load touches 64/G rows x (16*G) contiguous bytes.
//     G = 2  the library's lane-pair mapping (exact.hpp): 32 rows x 32 B per instruction — a 128-byte line is completed by 4 instructions
//     G = 8  8 rows x 128 B: one whole line per row per instruction
//     G = 64 one row, 1 KiB per instruction
//   U = 16-byte loads in flight per lane, W = resident waves per CU (64-thread workgroups), rows drawn by a hash (uniform, independent).
//
// build: hipcc --offload-arch=gfx950 -O3 -o gather tools/micro/gather.hip ; run: ./gather [GiB of table] > gather.jsonl
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <int G, int U, bool NT>
__global__ __launch_bounds__(64) void gather_kernel(const uint8_t* __restrict__ table, uint64_t nrows, uint32_t row_bytes, uint32_t passes,
                                                    uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x;
  const int sub = lane % G, rl = lane / G;          // position inside the row group / which of the 64/G rows of this pass
  const uint32_t steps = (row_bytes + 16 * G - 1) / (16 * G);
  u32x4 acc = {0, 0, 0, 0};
  for (uint32_t p = 0; p < passes; p++) {
    const uint64_t r = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)p << 8) ^ (uint64_t)rl) % nrows;
    const uint8_t* row = table + r * (uint64_t)row_bytes;
    for (uint32_t t0 = 0; t0 < steps; t0 += U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t off = ((t0 + u) * G + sub) * 16u;
        if (t0 + u < steps && off < row_bytes) {
          if constexpr (NT) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + off));
          else v[u] = *reinterpret_cast<const u32x4*>(row + off);
        } else v[u] = u32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc ^= v[u];
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;  // keeps the loads alive
}

template <int G, int U, bool NT>
double run(const uint8_t* table, uint64_t nrows, uint32_t row_bytes, int waves_per_cu, uint32_t* sink, double target_gb) {
  const uint32_t grid = 256u * (uint32_t)waves_per_cu;
  const double bytes_per_pass = (double)grid * (64 / G) * row_bytes;
  uint32_t passes = (uint32_t)(target_gb * 1e9 / bytes_per_pass);
  if (passes < 4) passes = 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gather_kernel<G, U, NT><<<grid, 64>>>(table, nrows, row_bytes, passes / 4, sink);  // warm-up
  CK(hipDeviceSynchronize());
  double best = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    gather_kernel<G, U, NT><<<grid, 64>>>(table, nrows, row_bytes, passes, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double gbs = bytes_per_pass * passes / (ms * 1e-3) / 1e9;
    if (gbs > best) best = gbs;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return best;
}

template <int G, int U>
void both(const uint8_t* table, uint64_t nrows, uint32_t row_bytes, int w, uint32_t* sink, double gib) {
  const double a = run<G, U, false>(table, nrows, row_bytes, w, sink, 40.0);
  const double b = run<G, U, true>(table, nrows, row_bytes, w, sink, 40.0);
  printf("{\"table_GiB\": %.1f, \"row_bytes\": %u, \"lanes_per_row\": %d, \"loads_in_flight_per_lane\": %d, \"waves_per_cu\": %d, "
         "\"KB_in_flight_per_cu\": %.0f, \"GBps\": %.0f, \"GBps_nontemporal\": %.0f, \"frac_of_8TBs\": %.3f}\n",
         gib, row_bytes, G, U, w, (double)w * 64 * U * 16 / 1024.0, a, b, (a > b ? a : b) / 8000.0);
  fflush(stdout);
}

// Dependent-load latency: ONE wave, every step reads 64/G fresh random rows (16 B per lane, G lanes per row) whose indices depend on
// the previous step's data — the round trip (DRAM + a cold translation per row) a single-query HNSW walk pays per dependent step.
template <int G>
__global__ __launch_bounds__(64) void chase_kernel(const uint8_t* __restrict__ table, uint64_t nrows, uint32_t row_bytes, uint32_t steps, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x, sub = lane % G, rl = lane / G;
  uint32_t prev = 0;
  for (uint32_t s = 0; s < steps; s++) {
    const uint64_t r = mix(((uint64_t)prev << 20) ^ ((uint64_t)s << 8) ^ (uint64_t)rl) % nrows;
    const u32x4 v = *reinterpret_cast<const u32x4*>(table + r * (uint64_t)row_bytes + (uint32_t)sub * 16u);
    uint32_t x = v.x ^ v.y ^ v.z ^ v.w;
    for (int m = 32; m >= 1; m >>= 1) x ^= (uint32_t)__shfl_xor((int)x, m, 64);   // every lane's data feeds the next address
    prev = x + s;
  }
  if (prev == 0x12345678u) sink[0] = 1;
}
template <int G> void chase(const uint8_t* table, uint64_t nrows, uint32_t rb, uint32_t* sink, double gib) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const uint32_t steps = 4000;
  chase_kernel<G><<<1, 64>>>(table, nrows, rb, 200, sink); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); chase_kernel<G><<<1, 64>>>(table, nrows, rb, steps, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("{\"table_GiB\": %.1f, \"row_bytes\": %u, \"dependent_step\": \"%d random rows x %d B, one wave\", \"ns_per_step\": %.0f}\n", gib, rb, 64 / G, 16 * G, ms * 1e6 / steps);
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

// also callable from a process that already holds a HIP runtime (python -c "import torch, ctypes; ctypes.CDLL('./gather.so').gather_run(...)")
extern "C" int gather_run(double gib, int quick);
int main(int argc, char** argv) { return gather_run(argc > 1 ? atof(argv[1]) : 15.0, argc > 2 ? atoi(argv[2]) : 0); }

extern "C" int gather_run(double gib, int quick) {
  const uint64_t bytes = (uint64_t)(gib * (1ull << 30));
  uint8_t* table; uint32_t* sink;
  CK(hipMalloc(&table, bytes)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(table, 1, bytes)); CK(hipMemset(sink, 0, 1 << 20));
  CK(hipDeviceSynchronize());
  const uint32_t rbs[2] = {1536, 3072};
  if (quick == 2) {   // latency probe only
    for (uint32_t rb : rbs) { chase<2>(table, bytes / rb, rb, sink, gib); chase<8>(table, bytes / rb, rb, sink, gib); chase<64>(table, bytes / rb, rb, sink, gib); }
    CK(hipFree(table)); CK(hipFree(sink));
    return 0;
  }
  for (uint32_t rb : rbs) {
    const uint64_t nrows = bytes / rb;
    const int ws[3] = {4, 8, 16};
    for (int w : ws) {
      if (quick) {  // the two shapes that matter, one occupancy
        if (w != 8) continue;
        both<2, 24>(table, nrows, rb, w, sink, gib);
        both<4, 12>(table, nrows, rb, w, sink, gib);
        both<8, 12>(table, nrows, rb, w, sink, gib);
        continue;
      }
      both<2, 12>(table, nrows, rb, w, sink, gib);
      both<2, 24>(table, nrows, rb, w, sink, gib);
      both<2, 48>(table, nrows, rb, w, sink, gib);
      both<4, 12>(table, nrows, rb, w, sink, gib);
      both<4, 24>(table, nrows, rb, w, sink, gib);
      both<8, 12>(table, nrows, rb, w, sink, gib);
      both<8, 24>(table, nrows, rb, w, sink, gib);
      both<16, 12>(table, nrows, rb, w, sink, gib);
      both<64, 2>(table, nrows, rb, w, sink, gib);
      both<64, 3>(table, nrows, rb, w, sink, gib);
    }
  }
  CK(hipFree(table)); CK(hipFree(sink));
  return 0;
}
