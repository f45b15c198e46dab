// ldspath.hip — per-CU throughput of the ways a workgroup can bring a tile into LDS on gfx950 (MI355X), one persistent
// 512-thread workgroup per CU: (a) LDS-DMA 16 B/lane, (b) LDS-DMA 4 B/lane, (c) global_load_dwordx4 -> ds_write_b128,
// (d) global_load_dwordx4 only, and (e) what a concurrent ds_read_b128 stream loses while (a) or (c) runs.
// Build: hipcc --offload-arch=gfx950 -O3 -o coltt_amd/variants/ldspath tools/micro/ldspath.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void dma16(uint32_t voff, const void* sbase, uint32_t lds) {
  lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);
  const uint64_t sb = (uint64_t)(uintptr_t)sbase;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32));
  const uint64_t sbu = ((uint64_t)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbu), "s"(lds) : "memory");
}
__device__ __forceinline__ void dma4(uint32_t voff, const void* sbase, uint32_t lds) {
  lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds);
  const uint64_t sb = (uint64_t)(uintptr_t)sbase;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)sb), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(sb >> 32));
  const uint64_t sbu = ((uint64_t)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(sbu), "s"(lds) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8)); }

// MODE 0: DMA16, 1: DMA4, 2: load->ds_write, 3: load only.  LOADERS = number of waves (of 8) that move data; the others run a
// ds_read_b128 loop when READERS is set (their throughput is reported separately).  Every loader wave keeps DEPTH KB in flight.
template <int MODE, int LOADERS, bool READERS>
__global__ __launch_bounds__(512, 2) void k(const uint8_t* __restrict__ src, size_t span, int iters, unsigned long long* out, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  constexpr int DEPTH = 16;   // 1 KB instructions in flight per loader wave
  const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  unsigned long long bytes = 0, rbytes = 0;
  uint32_t acc = 0;
  if (wave < LOADERS) {
    // each loader wave streams its own region: wg * LOADERS + wave picks a 1/(grid*LOADERS) slice of [0, span)
    // big span: each loader wave streams its own slice; small span (L2-resident): every wave walks the whole span from its own start
    const bool shared_span = span <= ((size_t)64 << 20);
    const size_t slice = shared_span ? span : (span / ((size_t)gridDim.x * (LOADERS ? LOADERS : 1)) & ~(size_t)1023);
    const uint8_t* base = shared_span ? src : src + ((size_t)blockIdx.x * LOADERS + wave) * slice;
    const uint32_t ring = lds0 + wave * DEPTH * 1024;
    uint8_t* ringp = smem + wave * DEPTH * 1024;
    size_t off = shared_span ? (((size_t)blockIdx.x * LOADERS + wave) * 65536) % span : 0;
    if (MODE <= 1) {
      for (int i = 0; i < DEPTH; i++) {
        if (MODE == 0) dma16(lane * 16, base + off, ring + i * 1024); else { for (int j = 0; j < 4; j++) dma4(lane * 4, base + off + j * 256, ring + i * 1024 + j * 256); }
        off += 1024; if (off >= slice) off = 0;
      }
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
          if (MODE == 0) { wait_vm<DEPTH - 1>(); dma16(lane * 16, base + off, ring + i * 1024); }
          else { wait_vm<4 * DEPTH - 4>(); for (int j = 0; j < 4; j++) dma4(lane * 4, base + off + j * 256, ring + i * 1024 + j * 256); }
          off += 1024; if (off >= slice) off = 0;
          bytes += 1024;
        }
      }
      wait_vm<0>();
    } else {
      u32x4 r[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; i++) { r[i] = *reinterpret_cast<const u32x4*>(base + off + lane * 16); off += 1024; if (off >= slice) off = 0; }
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
          wait_vm<DEPTH - 1>();
          if (MODE == 2) *reinterpret_cast<u32x4*>(ringp + i * 1024 + lane * 16) = r[i]; else acc += r[i].x ^ r[i].w;
          r[i] = *reinterpret_cast<const u32x4*>(base + off + lane * 16);
          asm volatile("" : "+v"(r[i]) : : "memory");
          off += 1024; if (off >= slice) off = 0;
          bytes += 1024;
        }
      }
#pragma unroll
      for (int i = 0; i < DEPTH; i++) acc += r[i].y;
    }
  } else if (READERS) {
    // conflict-free ds_read_b128 stream over the first 64 KB while the loaders run (fixed amount of work per iteration)
    const uint8_t* p = smem + 128 * 1024 + lane * 16;
    for (int it = 0; it < iters * 4; it++) {
#pragma unroll
      for (int i = 0; i < 16; i++) { const u32x4 v = *reinterpret_cast<const u32x4*>(p + i * 1024); acc += v.x + v.z; }
      rbytes += 16 * 1024;
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (acc == 0x12345678u) sink[0] = acc;
  if (lane == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * 8 + wave) * 4;
    o[0] = bytes; o[1] = rbytes; o[2] = t1 - t0; o[3] = w1 - w0;
  }
}

template <int MODE, int LOADERS, bool READERS>
int run(const char* name, const uint8_t* src, size_t span, int iters, unsigned long long* d_out, uint32_t* d_sink) {
  auto kern = k<MODE, LOADERS, READERS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; rep++) {
    CK(hipMemset(d_out, 0, 256 * 8 * 4 * 8));
    CK(hipEventRecord(e0));
    kern<<<256, 512, 160 * 1024>>>(src, span, iters, d_out, d_sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(256 * 8 * 4);
  CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
  double bytes = 0, rbytes = 0, clk = 0, wall = 0; int nl = 0, nr = 0;
  double lclk = 0, rclk = 0;
  for (int w = 0; w < 256 * 8; w++) {
    bytes += (double)h[w * 4]; rbytes += (double)h[w * 4 + 1];
    if (h[w * 4]) { lclk += (double)h[w * 4 + 2]; nl++; clk += (double)h[w * 4 + 2]; wall += (double)h[w * 4 + 3]; }
    if (h[w * 4 + 1]) { rclk += (double)h[w * 4 + 2]; nr++; }
  }
  const double mhz = wall > 0 ? clk / wall * 100.0 : 0;
  printf("%-44s span %7.1f MB: %7.3f ms  %6.2f TB/s  load %5.1f B/clk/CU (%4.0f MHz)", name, span / 1048576.0, ms, bytes / ms / 1e9, nl ? bytes / 256.0 / (lclk / nl) : 0.0, mhz);
  if (nr) printf("  | ds_read %5.1f B/clk/CU", rbytes / 256.0 / (rclk / nr));
  printf("\n");
  return 0;
}

int main() {
  const size_t big = (size_t)8 << 30, small = (size_t)2 << 20;   // HBM-resident stream vs L2-resident (2 MB per ... whole chip re-reads it)
  uint8_t* src; CK(hipMalloc(&src, big)); CK(hipMemset(src, 1, big));
  unsigned long long* d_out; CK(hipMalloc(&d_out, 256 * 8 * 4 * 8)); uint32_t* d_sink; CK(hipMalloc(&d_sink, 64));
  for (int pass = 0; pass < 2; pass++) {
    const size_t span = pass == 0 ? small : big;
    const int iters = 200;
    printf("---- %s ----\n", pass == 0 ? "L2-resident source" : "HBM-resident source");
    if (run<0, 8, false>("LDS-DMA dwordx4, 8 loader waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<0, 4, false>("LDS-DMA dwordx4, 4 loader waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<0, 2, false>("LDS-DMA dwordx4, 2 loader waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<1, 8, false>("LDS-DMA dword,   8 loader waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<2, 8, false>("global_load x4 -> ds_write_b128, 8 waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<2, 4, false>("global_load x4 -> ds_write_b128, 4 waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<3, 8, false>("global_load x4 only, 8 waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<3, 4, false>("global_load x4 only, 4 waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<0, 4, true>("LDS-DMA dwordx4, 4 loaders + 4 ds_read waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<2, 4, true>("load->ds_write,  4 loaders + 4 ds_read waves", src, span, iters, d_out, d_sink)) return 1;
    if (run<3, 4, true>("load only,       4 loaders + 4 ds_read waves", src, span, iters, d_out, d_sink)) return 1;
  }
  // ds_read alone
  printf("---- reference ----\n");
  if (run<3, 0, true>("ds_read_b128 alone, 8 waves", src, small, 200, d_out, d_sink)) return 1;
  return 0;
}
