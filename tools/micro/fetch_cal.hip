// fetch_cal.hip — what rocprofv3's FETCH_SIZE reports for the two access patterns of the product-quantised walk (hnsw_pq.hpp, round 6), on known
// byte counts (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern"):
//   block_kernel   the neighbourhood blocks: a wave reads random 2 KiB blocks, lane pair p the 64 bytes at p * 64: each lane its half as two 16-byte loads
//                  (exactly AdcEval<.., NBR>::prefetch_at) — known bytes = blocks x 2048
//   probe_kernel   the visited byte map: every EVEN lane loads ONE aligned 32-bit word at a random address of a large table (the probe of
//                  search_level2's EARLY path) — algorithmic bytes = loads x 4; what HBM moves is a sector per load
//   stream_kernel  the yardstick: a plain 16 B/lane streaming read — known bytes = the table's
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_cal tools/micro/fetch_cal.hip
// run (counters in their OWN pass, no trace domains): cd /tmp && rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/cal -o c -- ./fetch_cal 16
//   then tools/pmc_traffic.py --fetch-cal /tmp/cal/*counter_collection.csv (divides each kernel's known bytes by its reported KiB)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__global__ __launch_bounds__(64) void block_kernel(const uint8_t* __restrict__ table, uint64_t nblocks, uint32_t passes, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x, p = lane >> 1;
  u32x4 acc = {0, 0, 0, 0};
  for (uint32_t i = 0; i < passes; i++) {
    const uint64_t b = mix(((uint64_t)blockIdx.x << 32) ^ i) % nblocks;   // wave-uniform: one block per pass
    {   // (round 6, call N: BOTH lanes of a pair load — each its half of the 64-byte row, two 16-byte loads)
      const u32x4* r = reinterpret_cast<const u32x4*>(table + b * 2048ull + (uint64_t)p * 64ull + (uint64_t)(lane & 1) * 32ull);
      const u32x4 a0 = r[0], a1 = r[1];
      acc ^= a0 ^ a1;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}
__global__ __launch_bounds__(64) void probe_kernel(const uint8_t* __restrict__ table, uint64_t nwords, uint32_t passes, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x;
  uint32_t acc = 0;
  for (uint32_t i = 0; i < passes; i++) {
    const uint64_t w = mix(((uint64_t)blockIdx.x << 40) ^ ((uint64_t)i << 8) ^ (uint64_t)lane) % nwords;
    if ((lane & 1) == 0) acc ^= __hip_atomic_load(reinterpret_cast<const uint32_t*>(table) + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (acc == 0x12345678u) sink[blockIdx.x] = 1;
}
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ table, uint64_t n16, uint32_t* __restrict__ sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) acc ^= table[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = 1;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~2047ull;
  uint8_t* table; uint32_t* sink;
  CK(hipMalloc(&table, bytes)); CK(hipMalloc(&sink, 1 << 20));
  CK(hipMemset(table, 1, bytes)); CK(hipMemset(sink, 0, 1 << 20));
  const uint32_t grid = 2560, passes = 4000;   // the walk's 10 waves per CU
  for (int rep = 0; rep < 3; rep++) {
    block_kernel<<<grid, 64>>>(table, bytes / 2048, passes, sink);
    probe_kernel<<<grid, 64>>>(table, bytes / 4, passes, sink);
    stream_kernel<<<4096, 256>>>(reinterpret_cast<const u32x4*>(table), bytes / 16, sink);
  }
  CK(hipDeviceSynchronize());
  printf("{\"table_GiB\": %.1f, \"block_kernel_known_bytes_per_launch\": %.0f, \"probe_kernel_loads_per_launch\": %.0f, \"stream_kernel_known_bytes_per_launch\": %.0f}\n",
         gib, (double)grid * passes * 2048.0, (double)grid * passes * 32.0, (double)bytes);
  return 0;
}
