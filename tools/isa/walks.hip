// tools/isa/walks.hip — scratch translation unit for ISA inspection: the walk kernels whose expansion chain the round's instruction work is about.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-gpu-flush-denormals-to-zero --cuda-device-only -S \
//         -I coltt_amd/csrc -I include tools/isa/walks.hip -o /tmp/walks.s && python tools/isa_loops.py /tmp/walks.s
// Not part of the library (coltt_amd/build.py compiles coltt_amd/csrc/*.hip only).
#include "hnsw_kernels.hpp"
namespace coltt {
namespace kern {
#ifndef WALKS_SET
#define WALKS_SET 15
#endif
#if WALKS_SET & 1   // the headline: 10 M x 768 f32, ef 128 (LDS-visited, eight-lane core)
template __global__ void hnsw_search2_kernel<0, 0, 1, 4, 1, false, true, false>(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
    uint32_t*, uint64_t*, float*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
#endif
#if WALKS_SET & 2   // the operating point: 10 M x 768 f16, ef 1024 (HBM-visited, delta set, Bloom filter, eight-lane core)
template __global__ void hnsw_search2_kernel<0, 1, 2, 7, 0, false, true, false>(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
    uint32_t*, uint64_t*, float*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
#endif
#if WALKS_SET & 4   // the walk over product-quantiser codes (64 x 32 quantiser: LS = 5, NP = 4; 32 x 256: LS = 8, NP = 2), gathered code rows | neighbourhood blocks
#define PQK(LS, NP, NBR) template __global__ void hnsw_pq_search_kernel<2, 0, LS, NP, NBR>(GraphView, int32_t, int32_t, const unsigned short*, const uint8_t*, const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, \
    uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
PQK(5, 0, false)
PQK(5, 4, false)
PQK(5, 4, true)
PQK(8, 2, true)
#undef PQK
#endif
#if WALKS_SET & 8   // single-query latency: 768 x f32 (24 lines), 768 x f16 (12 lines), natural-order rows
#define LATK(Q, TP, SEQ) template __global__ void hnsw_search_lat_kernel<0, Q, TP, SEQ>(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, \
    uint32_t*, uint64_t*, float*, uint32_t*, unsigned long long*, unsigned long long*, int);
LATK(0, 24, false)
LATK(1, 12, false)
LATK(0, -1, false)
LATK(0, 0, false)
LATK(0, -1, true)
#undef LATK
#endif
#if WALKS_SET & 16  // the non-temporal twins of the headline and the operating-point walks (collections far larger than the caches)
template __global__ void hnsw_search2_kernel<0, 0, 1, 4, 1, false, true, false, true>(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
    uint32_t*, uint64_t*, float*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
template __global__ void hnsw_search2_kernel<0, 1, 2, 7, 0, false, true, false, true>(GraphView, int32_t, int32_t, const float*, const float*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t,
    uint32_t*, uint64_t*, float*, uint32_t*, unsigned long long*, uint8_t*, size_t, uint32_t*);
#endif
}  // namespace kern
}  // namespace coltt
