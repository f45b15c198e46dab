// pq.hip — product-quantised store on the GPU: codebooks, Encode, the per-query distance table and the ADC scan
// (BASELINE.json north_star: "PQ-codebook kernels ... PQ codebooks staged in LDS").
//
// What the reference holds for this row (the package that drove it, pkg/hnswpq — imported by playground/hnswpq_verification.go:29 —
// is not in the tree; parity for everything above the distancepq leaf functions is therefore a DEFINITION, written down in
// oracle/coltt_oracle.cpp "Product quantiser" and repeated here):
//   parameters  pkg/models/hnsw_common.go:20-33  NumCentroids in [2,256] (one uint8 code per sub-vector), NumSubVectors >= 2;
//   arithmetic  pkg/distancepq/distance.go:30-42 euclideanDistance (SQUARED L2), cosineDistance = 1 - dot, dotProductDistance = -dot,
//               over asm.SquaredEuclideanDistance / asm.Dot (asm/euclidean.s:7-65, asm/dot.s:7-55; FMA, 4 x 8-lane accumulators);
//   call shape  playground/hnswpq_verification.go:69-73,90-105,154,190-199 (m = 32, 256 centroids, train, Fit, search on codes only).
// Definition:  Encode   code[j] = argmin_c SquaredEuclideanDistance(x_j, centroid[j][c]), strict `<` scanning c upwards from MaxFloat32;
//              LUT      lut[j][c] = distFn(q_j, centroid[j][c])           (distFn = the store's distancepq function)
//              score    dist = 0; for j in 0..m-1: dist += lut[j][code[j]] (f32, in j order)
//              top-k    the k smallest by (score bits, id) — the canonical order of every store here.
//
// HBM layout: codes are TILE-INTERLEAVED so that a wave streams them with fully coalesced 16-byte loads while every lane owns one
// row: a tile is 64 rows; a row's m codes (padded to mp = a multiple of 4) are cut into T pieces of PB = 16 / 8 / 4 bytes (the
// largest that divides mp); piece t of row r lives at ((r / 64) * T + t) * 64 * PB + (r % 64) * PB.  One wave instruction reads
// piece t of 64 consecutive rows = 64 * PB contiguous bytes.  ids [cap] u64 (absent in dense-id mode).
// The scan (pq_scan_kernel): the query's table is staged in LDS ([mp][256] f32, rows j >= m are +0.0: dist + 0.0 keeps dist's bits
// because dist is never -0), every lane walks its row's codes in j order — one ds_read_b32 per code — and rows under the running
// threshold are appended to the candidate list the shared selection (select.hpp) reduces.  Four queries per pass when their
// tables fit together ([mp][256][4] f32, ds_read_b128 fetches the four values of a code at once).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.hpp"
#include "exact.hpp"
#include "select.hpp"

using namespace coltt;
using namespace coltt::dev;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int PQ_LDS_MAX = 152 * 1024;   // LDS a scan workgroup may take (of 160 KiB per CU)

// ---- pkg/distancepq: asm.Dot / asm.SquaredEuclideanDistance by ONE thread (dot.s:7-55, euclidean.s:7-65): acc[8 r + j] is lane j of
// accumulator Y_r; the tail is the scalar-FMA chain in lane 0 of X4 / X1; reduce ((Y0+Y1)+Y2)+Y3 -> lo128 + hi128 -> + {tail,0,0,0}
// -> hadd, hadd.  KIND 0 cosineDistance (1 - dot), 1 euclideanDistance (squared), 2 dotProductDistance (-dot) (distance.go:30-42).
template <int KIND, int LEN = 0>   // LEN > 0: compile-time length (x may be a register array: every loop unrolls)
__device__ __forceinline__ float pq_dist(const float* __restrict__ x, const float* __restrict__ y, int len_rt) {
  const int len = LEN > 0 ? LEN : len_rt;
  float acc[32];
#pragma unroll
  for (int u = 0; u < 32; u++) acc[u] = 0.f;
#define PQ_BLOCK_(I_)                                                                                   \
  _Pragma("unroll") for (int u = 0; u < 32; u++) {                                                      \
    const float xv = x[(I_) + u], yv = y[(I_) + u];                                                     \
    if constexpr (KIND == 1) { const float d = xv - yv; acc[u] = __builtin_fmaf(d, d, acc[u]); }        \
    else acc[u] = __builtin_fmaf(xv, yv, acc[u]);                                                       \
  }
  int i = 0;
  if constexpr (LEN > 0) {
#pragma unroll
    for (int b = 0; b < LEN / 32; b++) { PQ_BLOCK_(b * 32) }
  } else {
    for (; len - i >= 32; i += 32) { PQ_BLOCK_(i) }
  }
#undef PQ_BLOCK_
  float tail = 0.f;
  if constexpr (LEN > 0) {
#pragma unroll
    for (int e = (LEN / 32) * 32; e < LEN; e++) {
      const float xv = x[e], yv = y[e];
      if constexpr (KIND == 1) { const float d = xv - yv; tail = __builtin_fmaf(d, d, tail); }
      else tail = __builtin_fmaf(xv, yv, tail);
    }
  } else {
    for (; i < len; i++) {
      const float xv = x[i], yv = y[i];
      if constexpr (KIND == 1) { const float d = xv - yv; tail = __builtin_fmaf(d, d, tail); }
      else tail = __builtin_fmaf(xv, yv, tail);
    }
  }
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; j++) s[j] = ((acc[j] + acc[8 + j]) + acc[16 + j]) + acc[24 + j];
  float t0 = s[0] + s[4], t1 = s[1] + s[5], t2 = s[2] + s[6], t3 = s[3] + s[7];   // VEXTRACTF128 + VADDPS
  t0 = t0 + tail; t1 = t1 + 0.0f; t2 = t2 + 0.0f; t3 = t3 + 0.0f;                   // VADDPS X0, {tail,0,0,0}
  const float h0 = t0 + t1, h1 = t2 + t3;                                             // VHADDPS
  const float r = h0 + h1;                                                            // VHADDPS
  if constexpr (KIND == 0) return 1.0f - r;
  else if constexpr (KIND == 2) return -r;
  else return r;
}

// ---- the per-query distance table: lut[q][j][c], c < 256 (entries c >= C and rows j >= m are +0.0 / never read)
template <int KIND>
__global__ __launch_bounds__(256) void pq_lut_kernel(const float* __restrict__ cb, int m, int C, int dsub, const float* __restrict__ queries,
                                                     int mp, float* __restrict__ lut, uint32_t* __restrict__ cnt, uint32_t* __restrict__ thr,
                                                     uint32_t* __restrict__ ovf) {
  const int j = blockIdx.x, q = blockIdx.y, c = threadIdx.x;
  if (cnt && j == 0 && c == 0) { cnt[q] = 0; thr[q] = 0xffffffffu; if (q == 0) *ovf = 0; }   // the search's group state (one launch less)
  float v = 0.f;
  if (j < m && c < C) v = pq_dist<KIND>(queries + ((size_t)q * m + j) * dsub, cb + ((size_t)j * C + c) * dsub, dsub);
  lut[((size_t)q * mp + j) * 256 + c] = v;
}

// the product-quantised HNSW's table (hnsw_pq.hpp): the same entries, rounded to binary16, rows as long as the centroid count rounded up to a power
// of two (1 << shift, 16 .. 256) — what the walk kernel copies into its LDS as it is: 4 KiB per query for 64 x 32 instead of 64 KiB of f32
// Table scale (round 6, ADVICE r5: un-normalised Euclidean data — SIFT-like values 0..255 — puts most entries above binary16's 65504, i.e. at +Inf, and the
// walk collapses).  Part of the DEFINITION (oracle/coltt_oracle.cpp: pq_table_scale): k = the smallest integer >= 0 with M * 2^-k <= 32768, M = the
// largest entry of the query's f32 table; every entry is multiplied by 2^-k (exact) before the binary16 rounding.  k = 0 for every table whose entries stay
// under 32768 — every table of rounds 4-5.  A power-of-two scale keeps the ranking; the answers carry exact distances.
// qmax[q] = the bits of max(entry, 0) over the query's table (non-negative floats order as their bits; zeroed by the caller)
template <int KIND>
__global__ __launch_bounds__(256) void pq_lut16_max_kernel(const float* __restrict__ cb, int m, int C, int dsub, const float* __restrict__ queries,
                                                           int mp, int shift, uint32_t* __restrict__ qmax) {
  const int per = 256 >> shift;
  const int j = blockIdx.x * per + ((int)threadIdx.x >> shift), c = (int)threadIdx.x & ((1 << shift) - 1), q = blockIdx.y;
  float v = 0.f;
  if (j < m && c < C) v = pq_dist<KIND>(queries + ((size_t)q * m + j) * dsub, cb + ((size_t)j * C + c) * dsub, dsub);
  uint32_t b = v > 0.f ? __float_as_uint(v) : 0u;   // (NaN compares false: 0)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)b, o, 64); b = t > b ? t : b; }
  if ((threadIdx.x & 63) == 0 && b) atomicMax(&qmax[q], b);
}
__device__ __forceinline__ float pq_table_scale(uint32_t max_bits) {
  const float M = __uint_as_float(max_bits);
  float s = 1.0f;
  if (M == M && M < __uint_as_float(0x7f800000u)) { while (M * s > 32768.0f) s *= 0.5f; }   // at most 113 halvings; products by a power of two are exact
  return s;
}
template <int KIND>
__global__ __launch_bounds__(256) void pq_lut16_kernel(const float* __restrict__ cb, int m, int C, int dsub, const float* __restrict__ queries,
                                                       int mp, int shift, const uint32_t* __restrict__ qmax, unsigned short* __restrict__ lut) {
  const int per = 256 >> shift;                                  // table rows per workgroup
  const int j = blockIdx.x * per + ((int)threadIdx.x >> shift), c = (int)threadIdx.x & ((1 << shift) - 1), q = blockIdx.y;
  if (j >= mp) return;
  float v = 0.f;
  if (j < m && c < C) v = pq_dist<KIND>(queries + ((size_t)q * m + j) * dsub, cb + ((size_t)j * C + c) * dsub, dsub);
  v = v * pq_table_scale(qmax[q]);
  lut[(((size_t)q * mp + j) << shift) + c] = (unsigned short)f32bits_to_f16bits(__float_as_uint(v));
}
// max over the centroids of ||centroid||^2 (bits), for coltt_hnsw_pq_attach's range check of cosineDistance tables
__global__ void pq_centroid_norm_max_kernel(const float* __restrict__ cb, uint32_t n_centroids, int dsub, uint32_t* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (i < n_centroids) for (int e = 0; e < dsub; e++) { const float x = cb[(size_t)i * dsub + e]; s = __builtin_fmaf(x, x, s); }
  if (i < n_centroids) atomicMax(out, s == s ? __float_as_uint(s) : 0x7fc00000u);
}

// ---- Encode.  One thread per (row, sub-space): the sub-vector sits in registers (DS compile-time) or is re-read (DS == 0), the
// sub-space's centroids are broadcast from LDS when they fit (lds_cb) or read through the caches.
__device__ __forceinline__ size_t code_offset(uint64_t row, int j, int T, int PB) {
  return (((row >> 6) * (uint64_t)T + (uint64_t)(j / PB)) * 64 + (row & 63)) * PB + (j % PB);
}
template <int DS>
__global__ __launch_bounds__(256) void pq_encode_kernel(const float* __restrict__ cb, int m, int C, int dsub, const float* __restrict__ vecs,
                                                        uint64_t n, const uint32_t* __restrict__ slots, uint64_t slot_base, int lds_cb,
                                                        uint8_t* __restrict__ codes, int T, int PB) {
  extern __shared__ __attribute__((aligned(16))) float s_cb[];
  const int j = blockIdx.y;
  const float* cbj = cb + (size_t)j * C * dsub;
  if (lds_cb) {
    for (int i = threadIdx.x; i < C * dsub; i += blockDim.x) s_cb[i] = cbj[i];
    __syncthreads();
    cbj = s_cb;
  }
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* x = vecs + (i * (uint64_t)m + j) * dsub;
  float minDist = 3.40282346638528859811704183484516925440e+38f;   // math.MaxFloat32
  int best = 0;
  if constexpr (DS > 0) {
    float xr[DS];
#pragma unroll
    for (int e = 0; e < DS; e++) xr[e] = x[e];
    for (int c = 0; c < C; c++) {
      const float d = pq_dist<1, DS>(xr, cbj + (size_t)c * DS, DS);
      if (d < minDist) { minDist = d; best = c; }
    }
  } else {
    for (int c = 0; c < C; c++) {
      const float d = pq_dist<1>(x, cbj + (size_t)c * dsub, dsub);
      if (d < minDist) { minDist = d; best = c; }
    }
  }
  const uint64_t row = slots ? slots[i] : slot_base + i;
  codes[code_offset(row, j, T, PB)] = (uint8_t)best;
}

// row-major codes [n][m] -> the interleaved layout (coltt_pq_upsert_codes) and back (coltt_pq_fetch_codes); also counts codes >= C
__global__ void pq_place_codes_kernel(const uint8_t* __restrict__ src, uint64_t n, int m, int C, const uint32_t* __restrict__ slots, uint64_t slot_base,
                                      uint8_t* __restrict__ codes, int T, int PB, uint32_t* __restrict__ bad) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (uint64_t)m) return;
  const uint64_t i = t / m; const int j = (int)(t - i * m);
  const uint8_t c = src[t];
  if ((int)c >= C) { atomicAdd(bad, 1u); return; }
  codes[code_offset(slots ? slots[i] : slot_base + i, j, T, PB)] = c;
}
__global__ void pq_fetch_codes_kernel(const uint8_t* __restrict__ codes, int T, int PB, uint64_t first, uint64_t n, int m, uint8_t* __restrict__ out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (uint64_t)m) return;
  const uint64_t i = t / m; const int j = (int)(t - i * m);
  out[t] = codes[code_offset(first + i, j, T, PB)];
}
__global__ void pq_move_row_kernel(uint8_t* __restrict__ codes, int T, int PB, int mp, uint64_t dst, uint64_t src) {
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (j < mp) codes[code_offset(dst, j, T, PB)] = codes[code_offset(src, j, T, PB)];
}

// ---- Train (Lloyd; deterministic): init = the first C training vectors, assignment = Encode, update = f32 sum in training-index
// order / float32(count), an empty cluster keeps its centroid.
__global__ void pq_train_init_kernel(const float* __restrict__ vecs, int m, int C, int dsub, float* __restrict__ cb) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)m * C * dsub) return;
  const int e = (int)(t % dsub); const int c = (int)((t / dsub) % C); const int j = (int)(t / ((uint64_t)dsub * C));
  cb[t] = vecs[((size_t)c * m + j) * dsub + e];
}
__global__ void pq_train_update_kernel(const float* __restrict__ vecs, uint64_t n, int m, int C, int dsub, const uint8_t* __restrict__ codes,
                                       int T, int PB, float* __restrict__ cb) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)m * C * dsub) return;
  const int e = (int)(t % dsub); const int c = (int)((t / dsub) % C); const int j = (int)(t / ((uint64_t)dsub * C));
  float s = 0.f; uint32_t cnt = 0;
  for (uint64_t i = 0; i < n; i++) {
    if (codes[code_offset(i, j, T, PB)] != (uint8_t)c) continue;
    s += vecs[(i * (uint64_t)m + j) * dsub + e];
    cnt++;
  }
  if (cnt) cb[t] = div_rn(s, (float)cnt);
}

// ---- the ADC scan -------------------------------------------------------------------------------------------------------------
template <int W> struct PqRaw;
template <> struct PqRaw<1> { typedef uint32_t type; };
template <> struct PqRaw<2> { typedef u32x2e type; };
template <> struct PqRaw<4> { typedef u32x4e type; };
template <int W> __device__ __forceinline__ uint32_t pq_word(const typename PqRaw<W>::type& r, int i) {
  if constexpr (W == 1) return r;
  else if constexpr (W == 2) return i == 0 ? r.x : r.y;
  else return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}

constexpr int PQ_RING = 8;   // pieces per lane in flight (16-byte pieces: 128 B per lane, 8 KB per wave)

// W = dwords per piece, QB = queries per pass (1: lut [mp][256]; 4: lut [mp][256][4]).  grid.x = persistent workgroups over the tiles
// of [begin, end), grid.y = query groups.  begin is a multiple of 64.
template <int W, int QB>
__global__ __launch_bounds__(1024) void pq_scan_kernel(const uint8_t* __restrict__ codes, int T, int mp, const float* __restrict__ lut_g, int nq,
                                                       uint64_t begin, uint64_t end, const uint32_t* __restrict__ thr,
                                                       unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  extern __shared__ __attribute__((aligned(16))) float lut[];
  typedef typename PqRaw<W>::type raw_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int q0 = (int)blockIdx.y * QB;
  const size_t lq = (size_t)mp * 256;
  if constexpr (QB == 1) {
    const f32x4* src = reinterpret_cast<const f32x4*>(lut_g + (size_t)q0 * lq);
    f32x4* dst = reinterpret_cast<f32x4*>(lut);
    for (int i = tid; i < mp * 64; i += blockDim.x) dst[i] = src[i];
  } else if constexpr (QB == 2) {   // [mp][256][2]: one ds_read_b64 fetches both queries' values of a code
    for (int i = tid; i < mp * 256; i += blockDim.x) {
      f32x2 v;
      v.x = lut_g[(size_t)q0 * lq + i];
      v.y = q0 + 1 < nq ? lut_g[(size_t)(q0 + 1) * lq + i] : 0.f;
      reinterpret_cast<f32x2*>(lut)[i] = v;
    }
  } else {
    for (int i = tid; i < mp * 256; i += blockDim.x) {
      f32x4 v;
      v.x = lut_g[(size_t)q0 * lq + i];
      v.y = q0 + 1 < nq ? lut_g[(size_t)(q0 + 1) * lq + i] : 0.f;
      v.z = q0 + 2 < nq ? lut_g[(size_t)(q0 + 2) * lq + i] : 0.f;
      v.w = q0 + 3 < nq ? lut_g[(size_t)(q0 + 3) * lq + i] : 0.f;
      reinterpret_cast<f32x4*>(lut)[i] = v;
    }
  }
  uint32_t th[QB]; bool qv[QB];
#pragma unroll
  for (int q = 0; q < QB; q++) { qv[q] = q0 + q < nq; th[q] = qv[q] ? thr[q0 + q] : 0u; }
  __syncthreads();

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpb = (int)(blockDim.x >> 6);
  const uint64_t tile_end = (end + 63) >> 6;
  const uint64_t tstride = (uint64_t)gridDim.x * wpb;
  uint64_t ptile = (begin >> 6) + (uint64_t)blockIdx.x * wpb + wave, ctile = ptile;
  int pt = 0, ct = 0;
  if (ctile >= tile_end) return;
  constexpr int PBYTES = 4 * W;
  // a piece past this wave's last tile is fetched from the segment's last tile (valid memory) and never scored
#define PQ_LOAD(dst)                                                                                                      \
  {                                                                                                                       \
    const uint64_t lt_ = ptile < tile_end ? ptile : tile_end - 1;                                                         \
    dst = *reinterpret_cast<const raw_t*>(codes + ((lt_ * (uint64_t)T + (uint64_t)pt) * 64 + (uint64_t)lane) * PBYTES);   \
    if (++pt == T) { pt = 0; ptile += tstride; }                                                                          \
  }
  // Double-buffered walk over the wave's flattened (tile, piece) sequence: the R pieces of the NEXT round are requested before the
  // current round is scored, so their HBM latency hides under R x 4W table lookups (a ring refilled piece by piece made hipcc
  // rotate the ring registers at the back edge behind an s_waitcnt vmcnt(0), with the last load issued just before it).
  constexpr int R = W == 4 ? PQ_RING : 2 * PQ_RING;
  raw_t cur[R], nxt[R];
#pragma unroll
  for (int u = 0; u < R; u++) PQ_LOAD(cur[u])
  typedef typename std::conditional<QB == 1, float, typename std::conditional<QB == 2, f32x2, f32x4>::type>::type acc_t;
  acc_t acc;
  if constexpr (QB == 1) acc = 0.f; else if constexpr (QB == 2) acc = f32x2{0.f, 0.f}; else acc = f32x4{0.f, 0.f, 0.f, 0.f};
  while (ctile < tile_end) {
#pragma unroll
    for (int u = 0; u < R; u++) PQ_LOAD(nxt[u])
#pragma unroll
    for (int u = 0; u < R; u++) {
      const raw_t raw = cur[u];
      // codes of piece ct: j = ct * 4W + 4 d + b, in j order
      const float* lp = lut + (size_t)ct * (PBYTES * 256 * QB);
#pragma unroll
      for (int d = 0; d < W; d++) {
        const uint32_t wd = pq_word<W>(raw, d);
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const uint32_t code = (wd >> (8 * b)) & 0xffu;
          if constexpr (QB == 1) acc = acc + lp[(d * 4 + b) * 256 + code];
          else if constexpr (QB == 2) acc = acc + *reinterpret_cast<const f32x2*>(lp + ((size_t)(d * 4 + b) * 256 + code) * 2);
          else acc = acc + *reinterpret_cast<const f32x4*>(lp + ((size_t)(d * 4 + b) * 256 + code) * 4);
        }
      }
      if (++ct == T) {
        const uint64_t row = (ctile << 6) + (uint64_t)lane;
        const bool valid = ctile < tile_end && row >= begin && row < end;
#pragma unroll
        for (int q = 0; q < QB; q++) {
          float sc;
          if constexpr (QB == 1) sc = acc; else if constexpr (QB == 2) sc = q == 0 ? acc.x : acc.y; else sc = q == 0 ? acc.x : (q == 1 ? acc.y : (q == 2 ? acc.z : acc.w));
          const uint32_t key = score_key(sc);
          if (valid && qv[q] && key <= th[q]) {
            const uint32_t idx = atomicAdd(&cnt[q0 + q], 1u);
            if (idx < cap) cand[(size_t)(q0 + q) * cap + idx] = ((unsigned long long)key << 32) | (uint32_t)row;
          }
        }
        if constexpr (QB == 1) acc = 0.f; else if constexpr (QB == 2) acc = f32x2{0.f, 0.f}; else acc = f32x4{0.f, 0.f, 0.f, 0.f};
        ct = 0; ctile += tstride;
      }
    }
#pragma unroll
    for (int u = 0; u < R; u++) cur[u] = nxt[u];
  }
#undef PQ_LOAD
}

// ---- ONE query, ONE scan launch (round 6; VERDICT r5 #6: a single-query search was a chain of seven launches — table, then three scan + select
// pairs, each scan behind the threshold the previous selection published — 0.27 ms around a 0.18 ms scan).  Here every WAVE keeps its own
// self-tightening list: the rows whose key <= the wave's threshold go into a 128-entry LDS list; past 64 entries the list is cut back to the
// entries <= its k-th smallest key (ties stay) and that key becomes the threshold — after the first tile a wave admits a row with probability
// ~k / rows seen, so a wave of 2 400 rows compresses two or three times.  At the end the workgroup merges its waves' lists, keeps the entries
// <= ITS k-th smallest key and appends them to the query's candidate list: ~k per workgroup, a few thousand in all, and the ordinary selection
// (select.hpp: (score, id) order, ids looked up there) runs once.  Every level keeps ALL keys <= a bound that is >= the collection's k-th
// smallest key, so the candidate list is a superset of the answer: same answers as the segment chain, bit for bit.  Mass ties that do not fit a
// wave's list raise the overflow flag and the caller re-runs the bounded-segment chain, as before.
constexpr uint32_t PQ1_KMAX = 64;     // largest k the one-launch scan serves (a wave's list keeps <= 64 survivors)
constexpr uint32_t PQ1_WL = 128;      // entries of a wave's list
static __device__ __forceinline__ void pq_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// cut the wave's list (nl >= k entries, distinct (key << 32 | row) values) back to the entries whose key <= its k-th smallest key
static __device__ __forceinline__ uint32_t pq1_compress(unsigned long long* __restrict__ wl, uint32_t nl, uint32_t k, uint32_t& thw, int lane, uint32_t* __restrict__ ovf) {
  const bool h0 = (uint32_t)lane < nl, h1 = 64u + (uint32_t)lane < nl;
  const unsigned long long e0 = h0 ? wl[lane] : ~0ull, e1 = h1 ? wl[64 + lane] : ~0ull;
  uint32_t r0 = 0, r1 = 0;
  for (uint32_t j = 0; j < nl; j++) { const unsigned long long ej = wl[j]; r0 += ej < e0 ? 1u : 0u; r1 += ej < e1 ? 1u : 0u; }
  const unsigned long long m0 = __ballot(h0 && r0 == k - 1), m1 = __ballot(h1 && r1 == k - 1);   // exactly one entry has rank k - 1
  const uint32_t kth = m0 ? (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e0 >> 32), __builtin_ctzll(m0))
                          : (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e1 >> 32), __builtin_ctzll(m1 | (1ull << 63)));
  const bool k0 = h0 && (uint32_t)(e0 >> 32) <= kth, k1 = h1 && (uint32_t)(e1 >> 32) <= kth;
  const unsigned long long b0 = __ballot(k0), b1 = __ballot(k1), lt = (1ull << lane) - 1ull;
  const uint32_t n0 = (uint32_t)__popcll(b0), p1 = n0 + (uint32_t)__popcll(b1 & lt), tot = n0 + (uint32_t)__popcll(b1);
  pq_wave_sync();   // every lane holds its entries before the list is rewritten
  if (k0) wl[__popcll(b0 & lt)] = e0;
  if (k1 && p1 < PQ1_WL) wl[p1] = e1;
  pq_wave_sync();
  thw = kth;
  if (tot > 64u) { if (lane == 0) atomicOr(ovf, 1u); return 64u; }   // more ties than a list holds: the caller falls back to bounded segments
  return tot;
}

template <int W>
__global__ __launch_bounds__(1024) void pq_scan1_kernel(const uint8_t* __restrict__ codes, int T, int mp, const float* __restrict__ lut_g, uint64_t end, uint32_t k,
                                                        unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap, uint32_t* __restrict__ ovf) {
  extern __shared__ __attribute__((aligned(16))) float lut[];
  __shared__ uint32_t s_wn[16], s_kth;
  typedef typename PqRaw<W>::type raw_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wpb = (int)(blockDim.x >> 6);
  unsigned long long* const wl = reinterpret_cast<unsigned long long*>(lut + (size_t)mp * 256) + (size_t)wave * PQ1_WL;
  unsigned long long* const bl = reinterpret_cast<unsigned long long*>(lut + (size_t)mp * 256) + (size_t)wpb * PQ1_WL;   // [wpb * 64] the workgroup's merge list
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(lut_g);
    f32x4* dst = reinterpret_cast<f32x4*>(lut);
    for (int i = tid; i < mp * 64; i += blockDim.x) dst[i] = src[i];
  }
  if (tid == 0) s_kth = 0xffffffffu;
  __syncthreads();
  const uint64_t tile_end = (end + 63) >> 6;
  const uint64_t tstride = (uint64_t)gridDim.x * wpb;
  uint64_t ptile = (uint64_t)blockIdx.x * wpb + wave, ctile = ptile;
  int pt = 0, ct = 0;
  uint32_t nl = 0, thw = 0xffffffffu;
  const unsigned long long ltm = (1ull << lane) - 1ull;
  constexpr int PBYTES = 4 * W;
#define PQ_LOAD(dst)                                                                                                      \
  {                                                                                                                       \
    const uint64_t lt_ = ptile < tile_end ? ptile : tile_end - 1;                                                         \
    dst = *reinterpret_cast<const raw_t*>(codes + ((lt_ * (uint64_t)T + (uint64_t)pt) * 64 + (uint64_t)lane) * PBYTES);   \
    if (++pt == T) { pt = 0; ptile += tstride; }                                                                          \
  }
  constexpr int R = W == 4 ? PQ_RING : 2 * PQ_RING;
  if (ctile < tile_end) {
    raw_t cur[R], nxt[R];
#pragma unroll
    for (int u = 0; u < R; u++) PQ_LOAD(cur[u])
    float acc = 0.f;
    while (ctile < tile_end) {
#pragma unroll
      for (int u = 0; u < R; u++) PQ_LOAD(nxt[u])
#pragma unroll
      for (int u = 0; u < R; u++) {
        const raw_t raw = cur[u];
        const float* lp = lut + (size_t)ct * (PBYTES * 256);
#pragma unroll
        for (int d = 0; d < W; d++) {
          const uint32_t wd = pq_word<W>(raw, d);
#pragma unroll
          for (int b = 0; b < 4; b++) acc = acc + lp[(d * 4 + b) * 256 + ((wd >> (8 * b)) & 0xffu)];
        }
        if (++ct == T) {
          const uint64_t row = (ctile << 6) + (uint64_t)lane;
          const uint32_t key = score_key(acc);
          const bool pass = ctile < tile_end && row < end && key <= thw;
          const unsigned long long pm = __ballot(pass);
          if (pm) {
            if (pass) wl[nl + (uint32_t)__popcll(pm & ltm)] = ((unsigned long long)key << 32) | (uint32_t)row;   // nl <= 64 here: fits
            nl += (uint32_t)__popcll(pm);
            pq_wave_sync();
            if (nl > 64u) nl = pq1_compress(wl, nl, k, thw, lane, ovf);
          }
          acc = 0.f; ct = 0; ctile += tstride;
        }
      }
#pragma unroll
      for (int u = 0; u < R; u++) cur[u] = nxt[u];
    }
    if (nl > k) nl = pq1_compress(wl, nl, k, thw, lane, ovf);
  }
#undef PQ_LOAD
  // ---- the workgroup's merge: its waves' survivors, cut back to the entries <= the workgroup's k-th smallest key
  if (lane == 0) s_wn[wave] = nl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (int w = 0; w < wpb; w++) { const uint32_t c = s_wn[w]; base += w < wave ? c : 0u; tot += c; }
  for (uint32_t i = lane; i < nl; i += 64) bl[base + i] = wl[i];
  __syncthreads();
  if (tot >= k) {
    for (uint32_t i = tid; i < tot; i += blockDim.x) {
      const unsigned long long e = bl[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < tot; j++) rank += bl[j] < e ? 1u : 0u;
      if (rank == k - 1) s_kth = (uint32_t)(e >> 32);
    }
    __syncthreads();
  }
  const uint32_t kth = s_kth;
  for (uint32_t i = tid; i < tot; i += blockDim.x) {
    const unsigned long long e = bl[i];
    if ((uint32_t)(e >> 32) > kth) continue;
    const uint32_t idx = atomicAdd(&cnt[0], 1u);
    if (idx < cap) cand[idx] = e;
  }
}

// tables too large for LDS (mp * 1 KiB > PQ_LDS_MAX: more than 152 sub-vectors): the same walk with the table read through the
// caches — one thread per row, one query per launch row of the grid.  Exact; slow; rare.
__global__ __launch_bounds__(256) void pq_scan_global_kernel(const uint8_t* __restrict__ codes, int T, int PB, int m, int mp, const float* __restrict__ lut_g,
                                                             uint64_t begin, uint64_t end, const uint32_t* __restrict__ thr,
                                                             unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  const int q = blockIdx.y;
  const float* lut = lut_g + (size_t)q * mp * 256;
  const uint32_t th = thr[q];
  for (uint64_t row = begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; row < end; row += (uint64_t)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int j = 0; j < m; j++) acc = acc + lut[(size_t)j * 256 + codes[code_offset(row, j, T, PB)]];
    const uint32_t key = score_key(acc);
    if (key <= th) {
      const uint32_t idx = atomicAdd(&cnt[q], 1u);
      if (idx < cap) cand[(size_t)q * cap + idx] = ((unsigned long long)key << 32) | (uint32_t)row;
    }
  }
}

__global__ void pq_init_kernel(uint32_t* cnt, uint32_t* thr, uint32_t* ovf, int nq) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < nq) { cnt[q] = 0; thr[q] = 0xffffffffu; }
  if (q == 0) *ovf = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
struct PCtx {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evs0 = nullptr, evs1 = nullptr;   // whole search; the largest scan launch alone
  DevBuf w_q, w_lut, w_cand, w_cnt, w_out_ids, w_out_sc, w_out_cnt;
  int init() {
    COLTT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    COLTT_HIP(hipEventCreate(&ev0)); COLTT_HIP(hipEventCreate(&ev1));
    COLTT_HIP(hipEventCreate(&evs0)); COLTT_HIP(hipEventCreate(&evs1));
    return COLTT_OK;
  }
  ~PCtx() {
    for (hipEvent_t e : {ev0, ev1, evs0, evs1}) if (e) (void)hipEventDestroy(e);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

struct Pq : Object {
  uint32_t dim = 0, m = 0, C = 0, dsub = 0; int metric = 0;
  int mp = 0, PB = 0, T = 0;          // codes per row padded to 4; piece bytes; pieces per row
  bool trained = false;
  uint64_t n = 0, cap = 0;            // cap is a multiple of 64 (whole tiles)
  DevBuf cb, codes, ids;
  bool dense = true; uint64_t dense_base = 0;
  std::unordered_map<uint64_t, uint32_t> id2slot; std::vector<uint64_t> h_ids;
  hipStream_t stream = nullptr;
  DevBuf w_raw, w_slots, w_bad;
  std::atomic<float> last_ms{0.f}, last_scan_ms{0.f};
  std::atomic<uint64_t> last_scan_rows{0};
  CtxPool<PCtx> pool;
  ~Pq() override { (void)hipSetDevice(device); if (stream) (void)hipStreamDestroy(stream); }
  size_t tile_bytes() const { return (size_t)64 * mp; }
  int reserve(uint64_t rows) {
    if (rows <= cap) return COLTT_OK;
    uint64_t nc = std::max<uint64_t>({rows, cap + cap / 2, 1024});
    nc = (nc + 63) & ~63ull;
    const size_t old = codes.cap;
    COLTT_TRY(codes.reserve(nc / 64 * tile_bytes(), true, stream));
    if (codes.cap > old) {   // padding codes (j >= m) and rows never written read as code 0
      COLTT_HIP(hipMemsetAsync(codes.as<uint8_t>() + old, 0, codes.cap - old, stream));
      COLTT_HIP(hipStreamSynchronize(stream));
    }
    if (!dense) COLTT_TRY(ids.reserve(nc * 8, true, stream));
    cap = nc;
    return COLTT_OK;
  }
  int undense() {
    if (!dense) return COLTT_OK;
    h_ids.resize(n); id2slot.reserve(n * 2);
    for (uint64_t s = 0; s < n; s++) { h_ids[s] = dense_base + s; id2slot[dense_base + s] = (uint32_t)s; }
    dense = false;
    COLTT_TRY(ids.reserve(std::max<uint64_t>(cap, 1024) * 8, false, stream));
    if (n) COLTT_HIP(hipMemcpyAsync(ids.p, h_ids.data(), n * 8, hipMemcpyHostToDevice, stream));
    COLTT_HIP(hipStreamSynchronize(stream));
    return COLTT_OK;
  }
};

struct PqPlan { std::vector<uint32_t> slots; std::vector<uint64_t> new_ids; uint64_t nn = 0; };
int plan_upsert(const Pq* p, const uint64_t* ids, uint64_t first_id, size_t n, PqPlan& pl) {
  pl.slots.resize(n); pl.nn = p->n;
  std::unordered_map<uint64_t, uint32_t> fresh;
  for (size_t i = 0; i < n; i++) {
    const uint64_t id = ids ? ids[i] : first_id + i;
    auto it = p->id2slot.find(id);
    if (it != p->id2slot.end()) { pl.slots[i] = it->second; continue; }
    auto r = fresh.emplace(id, (uint32_t)pl.nn);
    if (r.second) { pl.new_ids.push_back(id); pl.nn++; }
    pl.slots[i] = r.first->second;
  }
  if (pl.nn > 0xffffffffull) return fail(COLTT_E_UNSUPPORTED, "pq upsert: more than 2^32-1 rows in one store");
  return COLTT_OK;
}
// a repeated id inside one batch: the last occurrence wins (the others would race on the same row)
void last_wins(const PqPlan& pl, size_t n, std::vector<size_t>& keep) {
  std::unordered_map<uint32_t, size_t> last;
  for (size_t i = 0; i < n; i++) last[pl.slots[i]] = i;
  keep.clear();
  if (last.size() == n) return;
  for (size_t i = 0; i < n; i++) if (last[pl.slots[i]] == i) keep.push_back(i);
}
void commit_upsert(Pq* p, const PqPlan& pl) {
  for (size_t j = 0; j < pl.new_ids.size(); j++) { p->id2slot[pl.new_ids[j]] = (uint32_t)(p->n + j); p->h_ids.push_back(pl.new_ids[j]); }
  p->n = pl.nn;
}

int launch_encode(Pq* p, hipStream_t s, const float* d_vecs, uint64_t n, const uint32_t* d_slots, uint64_t slot_base, uint8_t* codes) {
  if (n == 0) return COLTT_OK;
  const size_t cb_bytes = (size_t)p->C * p->dsub * 4;
  const int lds_cb = cb_bytes <= 64 * 1024;
  const size_t lds = lds_cb ? cb_bytes : 0;
  dim3 grid(ceil_div(n, 256), p->m);
#define PQ_ENC(DS)                                                                                                               \
  {                                                                                                                              \
    auto kern = pq_encode_kernel<DS>;                                                                                            \
    if (lds > 48 * 1024) COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    kern<<<grid, 256, lds, s>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_vecs, n, d_slots, slot_base, lds_cb, codes, p->T, p->PB); \
  }
  switch (p->dsub) {
    case 4: PQ_ENC(4) break;
    case 8: PQ_ENC(8) break;
    case 12: PQ_ENC(12) break;
    case 16: PQ_ENC(16) break;
    case 24: PQ_ENC(24) break;
    case 32: PQ_ENC(32) break;
    default: PQ_ENC(0) break;
  }
#undef PQ_ENC
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

template <int W>
int launch_scan_w(Pq* p, PCtx* c, int QBq, uint64_t b, uint64_t e, int nq, const float* lut, const uint32_t* thr, unsigned long long* cand, uint32_t* cnt, uint32_t cap) {
  const size_t lds = (size_t)p->mp * 1024 * QBq;
  const int threads = lds > 80 * 1024 ? 1024 : (lds > 40 * 1024 ? 512 : 256);
  const int wpb = threads / 64;
  const int blocks_per_cu = std::max<int>(1, std::min<int>(2048 / threads, (int)((160 * 1024) / std::max<size_t>(lds, 1))));
  const uint64_t tiles = ((e + 63) >> 6) - (b >> 6);
  const uint32_t gx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((tiles + wpb - 1) / wpb, (uint64_t)256 * blocks_per_cu));
  dim3 grid(gx, (nq + QBq - 1) / QBq);
  if (QBq == 1) {
    auto kern = pq_scan_kernel<W, 1>;
    COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 1024)));
    kern<<<grid, threads, lds, c->stream>>>(p->codes.as<uint8_t>(), p->T, p->mp, lut, nq, b, e, thr, cand, cnt, cap);
  } else if (QBq == 2) {
    auto kern = pq_scan_kernel<W, 2>;
    COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 1024)));
    kern<<<grid, threads, lds, c->stream>>>(p->codes.as<uint8_t>(), p->T, p->mp, lut, nq, b, e, thr, cand, cnt, cap);
  } else {
    auto kern = pq_scan_kernel<W, 4>;
    COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max<size_t>(lds, 1024)));
    kern<<<grid, threads, lds, c->stream>>>(p->codes.as<uint8_t>(), p->T, p->mp, lut, nq, b, e, thr, cand, cnt, cap);
  }
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

int launch_scan(Pq* p, PCtx* c, uint64_t b, uint64_t e, int nq, const float* lut, const uint32_t* thr, unsigned long long* cand, uint32_t* cnt, uint32_t cap) {
  const size_t one = (size_t)p->mp * 1024;
  if (one > (size_t)PQ_LDS_MAX) {
    dim3 grid((uint32_t)std::min<uint64_t>(ceil_div(e - b, 256), 256 * 8), nq);
    pq_scan_global_kernel<<<grid, 256, 0, c->stream>>>(p->codes.as<uint8_t>(), p->T, p->PB, (int)p->m, p->mp, lut, b, e, thr, cand, cnt, cap);
    COLTT_HIP(hipGetLastError());
    return COLTT_OK;
  }
  // queries per pass: four tables when they fit together (mp <= 38), two (mp <= 76: [mp][256][2], ds_read_b64 — the codes of such stores used to be
  // streamed once per query in a batch), else one
  const int QBq = nq < 2 ? 1 : (one * 4 <= (size_t)PQ_LDS_MAX ? 4 : (one * 2 <= (size_t)PQ_LDS_MAX ? 2 : 1));
  if (p->PB == 16) return launch_scan_w<4>(p, c, QBq, b, e, nq, lut, thr, cand, cnt, cap);
  if (p->PB == 8) return launch_scan_w<2>(p, c, QBq, b, e, nq, lut, thr, cand, cnt, cap);
  return launch_scan_w<1>(p, c, QBq, b, e, nq, lut, thr, cand, cnt, cap);
}

// the one-launch scan of a single query (pq_scan1_kernel): false = this store / k does not take it
bool scan1_geometry(const Pq* p, uint32_t k, int* threads_out, size_t* lds_out) {
  const char* e = getenv("COLTT_PQ_ONE");   // measurement knob (read per call: tests toggle it in-process): 0 = the segment chain
  const bool off = e && e[0] == '0';
  if (off || k > PQ1_KMAX) return false;
  const size_t table = (size_t)p->mp * 1024;
  const int threads = table > 80 * 1024 ? 1024 : (table > 40 * 1024 ? 512 : 256);
  const size_t lds = table + (size_t)(threads / 64) * (PQ1_WL + 64) * 8;
  if (lds > (size_t)PQ_LDS_MAX) return false;
  *threads_out = threads; *lds_out = lds;
  return true;
}
template <int W>
int launch_scan1_w(Pq* p, PCtx* c, int threads, size_t lds, uint64_t total, uint32_t k, const float* lut, unsigned long long* cand, uint32_t* cnt, uint32_t cap, uint32_t* ovf) {
  const int wpb = threads / 64;
  const int blocks_per_cu = std::max<int>(1, std::min<int>(2048 / threads, (int)((160 * 1024) / lds)));
  const uint64_t tiles = (total + 63) >> 6;
  const uint32_t gx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((tiles + wpb - 1) / wpb, (uint64_t)256 * blocks_per_cu));
  auto kern = pq_scan1_kernel<W>;
  COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<gx, threads, lds, c->stream>>>(p->codes.as<uint8_t>(), p->T, p->mp, lut, total, k, cand, cnt, cap, ovf);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// one group of <= PQ_GROUP queries: tables, candidate lists and the selection's state are sized by the group, not by the call
constexpr size_t PQ_GROUP = 256;
int pq_search_group(Pq* p, PCtx* c, const float* queries, bool q_on_device, size_t nq, uint32_t k, uint64_t* out_ids, float* out_scores,
                    uint32_t* out_counts, bool out_on_device, float* acc_ms) {
  const float* d_q = queries;
  if (!q_on_device) {
    COLTT_TRY(c->w_q.reserve(nq * p->dim * 4));
    COLTT_HIP(hipMemcpyAsync(c->w_q.p, queries, nq * p->dim * 4, hipMemcpyHostToDevice, c->stream));
    d_q = c->w_q.as<float>();
  }
  uint64_t* d_oi = out_ids; float* d_os = out_scores; uint32_t* d_oc = out_counts;
  if (!out_on_device) {
    COLTT_TRY(c->w_out_ids.reserve(nq * k * 8)); COLTT_TRY(c->w_out_sc.reserve(nq * k * 4)); COLTT_TRY(c->w_out_cnt.reserve(nq * 4));
    d_oi = c->w_out_ids.as<uint64_t>(); d_os = c->w_out_sc.as<float>(); d_oc = c->w_out_cnt.as<uint32_t>();
  }
  const uint32_t cap = std::max<uint32_t>(65536u, 8u * k);
  COLTT_TRY(c->w_lut.reserve(nq * (size_t)p->mp * 256 * 4));
  COLTT_TRY(c->w_cand.reserve(nq * (size_t)cap * 8));
  COLTT_TRY(c->w_cnt.reserve((2 * nq + 4) * 4));
  uint32_t* cnt = c->w_cnt.as<uint32_t>(); uint32_t* thr = cnt + nq; uint32_t* ovf = thr + nq;
  unsigned long long* cand = c->w_cand.as<unsigned long long>();
  float* lut = c->w_lut.as<float>();
  const uint64_t* ids = p->dense ? nullptr : p->ids.as<uint64_t>();
  const uint64_t total = p->n;
  COLTT_HIP(hipEventRecord(c->ev0, c->stream));
  {
    dim3 grid(p->mp, (uint32_t)nq);
    if (p->metric == COLTT_PQ_COSINE) pq_lut_kernel<0><<<grid, 256, 0, c->stream>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_q, p->mp, lut, cnt, thr, ovf);
    else if (p->metric == COLTT_PQ_EUCLIDEAN) pq_lut_kernel<1><<<grid, 256, 0, c->stream>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_q, p->mp, lut, cnt, thr, ovf);
    else pq_lut_kernel<2><<<grid, 256, 0, c->stream>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_q, p->mp, lut, cnt, thr, ovf);
  }
  bool timed_scan = false;
  auto scan = [&](uint64_t b, uint64_t e, bool big) -> int {
    if (big) COLTT_HIP(hipEventRecord(c->evs0, c->stream));
    COLTT_TRY(launch_scan(p, c, b, e, (int)nq, lut, thr, cand, cnt, cap));
    if (big) { COLTT_HIP(hipEventRecord(c->evs1, c->stream)); timed_scan = true; p->last_scan_rows.store(e - b); }
    flat_select_kernel<<<(uint32_t)nq, 256, 0, c->stream>>>(cand, cnt, thr, cap, k, 1, ids, p->dense_base, ovf, d_oi, d_os, d_oc);
    return COLTT_OK;
  };
  int t1 = 0; size_t lds1 = 0;
  if (total == 0) {
    flat_select_kernel<<<(uint32_t)nq, 256, 0, c->stream>>>(cand, cnt, thr, cap, k, 1, ids, p->dense_base, ovf, d_oi, d_os, d_oc);
  } else if (nq == 1 && total > 65536 && scan1_geometry(p, k, &t1, &lds1)) {
    // ONE query over a large store: table, ONE scan launch (per-wave self-tightening lists, pq_scan1_kernel), ONE selection
    COLTT_HIP(hipEventRecord(c->evs0, c->stream));
    if (p->PB == 16) COLTT_TRY(launch_scan1_w<4>(p, c, t1, lds1, total, k, lut, cand, cnt, cap, ovf));
    else if (p->PB == 8) COLTT_TRY(launch_scan1_w<2>(p, c, t1, lds1, total, k, lut, cand, cnt, cap, ovf));
    else COLTT_TRY(launch_scan1_w<1>(p, c, t1, lds1, total, k, lut, cand, cnt, cap, ovf));
    COLTT_HIP(hipEventRecord(c->evs1, c->stream)); timed_scan = true; p->last_scan_rows.store(total);
    flat_select_kernel<<<1, 256, 0, c->stream>>>(cand, cnt, thr, cap, k, 1, ids, p->dense_base, ovf, d_oi, d_os, d_oc);
  } else {
    // an unfiltered first segment of 4 Ki rows (one radix selection over 4 Ki candidates), then segments 64x what has been seen, each
    // behind the threshold the selection published from everything before it: a row passes with probability ~k / seen, so every
    // later list is a few hundred entries (the selection's short path) and 10 M rows take three scan + select pairs, not four
    // (r04a: 0.32 ms per single-query search of which the dominant scan was 0.18 — the rest was launches)
    uint64_t s0 = std::min<uint64_t>({total, (uint64_t)cap, (std::max<uint64_t>(4096, 4ull * k) + 63) & ~63ull});
    for (uint64_t b = 0, e = s0; b < total; b = e, e = std::min<uint64_t>(total, e * 64)) COLTT_TRY(scan(b, e, e == total));
  }
  // One host round trip per call: the overflow flag travels with the answers; the events bracket the device chain only.
  auto fetch = [&](uint32_t* h_ovf) -> int {
    COLTT_HIP(hipEventRecord(c->ev1, c->stream));
    COLTT_HIP(hipGetLastError());
    COLTT_HIP(hipMemcpyAsync(h_ovf, ovf, 4, hipMemcpyDeviceToHost, c->stream));
    if (!out_on_device) {
      COLTT_HIP(hipMemcpyAsync(out_ids, d_oi, nq * k * 8, hipMemcpyDeviceToHost, c->stream));
      COLTT_HIP(hipMemcpyAsync(out_scores, d_os, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
      COLTT_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, c->stream));
    }
    COLTT_HIP(hipStreamSynchronize(c->stream));
    return COLTT_OK;
  };
  uint32_t h_ovf = 0;
  COLTT_TRY(fetch(&h_ovf));
  if (h_ovf && total) {   // mass ties / adversarial order: segments that cannot overflow (the list holds <= k + segment)
    pq_init_kernel<<<ceil_div(nq, 256), 256, 0, c->stream>>>(cnt, thr, ovf, (int)nq);
    const uint64_t seg = (cap - std::min<uint32_t>(k, cap / 2)) & ~63ull;
    for (uint64_t b = 0; b < total; b += seg) COLTT_TRY(scan(b, std::min<uint64_t>(total, b + seg), false));
    COLTT_TRY(fetch(&h_ovf));
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  if (acc_ms) *acc_ms += ms;   // (summed by the call, stored once: concurrent searches share the object under a read lock)
  if (timed_scan) { float sms = 0.f; (void)hipEventElapsedTime(&sms, c->evs0, c->evs1); p->last_scan_ms.store(sms); }
  return COLTT_OK;
}

// A call of any size runs group by group (as FLAT does, flat.hip): a 10 000-query call used to reserve 512 KiB of candidate list and mp KiB
// of table PER QUERY of the whole batch in a pooled context that never shrinks; now the workspaces top out at one group's.
int pq_search_common(Pq* p, PCtx* c, const float* queries, bool q_on_device, size_t nq, uint32_t k, uint64_t* out_ids, float* out_scores,
                     uint32_t* out_counts, bool out_on_device) {
  if (k == 0 || k > K_MAX) return fail(COLTT_E_UNSUPPORTED, "pq search: k=%u outside [1,%u]", k, K_MAX);
  if (nq == 0) return COLTT_OK;
  if (!p->trained) return fail(COLTT_E_INVALID, "pq search: the quantiser has no codebooks yet (coltt_pq_set_codebooks / coltt_pq_train)");
  float total_ms = 0.f;
  for (size_t q0 = 0; q0 < nq; q0 += PQ_GROUP) {
    const size_t gn = std::min(PQ_GROUP, nq - q0);
    COLTT_TRY(pq_search_group(p, c, queries + q0 * p->dim, q_on_device, gn, k, out_ids + q0 * k, out_scores + q0 * k, out_counts + q0, out_on_device, &total_ms));
  }
  p->last_ms.store(total_ms);
  return COLTT_OK;
}

}  // namespace

// ---- the quantiser's kernels for an index that carries codes of its own (hnsw.hip: product-quantised HNSW) -----------------------------
// a snapshot of a trained quantiser: its shape and a copy of its codebooks on the CURRENT device
int coltt::pq_snapshot(coltt_handle_t h, PqShape* shape, DevBuf* cb_out, hipStream_t s) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq snapshot: unknown quantiser handle");
  ReadLock g(p->rw);
  if (!p->trained) return fail(COLTT_E_INVALID, "pq snapshot: the quantiser has no codebooks yet (coltt_pq_set_codebooks / coltt_pq_train)");
  shape->dim = p->dim; shape->m = p->m; shape->C = p->C; shape->dsub = p->dsub; shape->metric = p->metric;
  const size_t bytes = (size_t)p->m * p->C * p->dsub * 4;
  COLTT_TRY(cb_out->reserve(bytes));
  COLTT_HIP(hipMemcpyAsync(cb_out->p, p->cb.p, bytes, hipMemcpyDefault, s));
  COLTT_HIP(hipStreamSynchronize(s));
  return COLTT_OK;
}

// Encode of n f32 vectors [n][dim] into ROW-MAJOR codes [n][row_bytes] (row_bytes >= m; the bytes j >= m are left as they are)
int coltt::pq_encode_rowmajor(hipStream_t s, const float* d_cb, const PqShape& sh, const float* d_vecs, uint64_t n, uint8_t* d_codes, uint32_t row_bytes) {
  if (n == 0) return COLTT_OK;
  const size_t cb_bytes = (size_t)sh.C * sh.dsub * 4;
  const int lds_cb = cb_bytes <= 64 * 1024;
  const size_t lds = lds_cb ? cb_bytes : 0;
  dim3 grid(ceil_div(n, 256), sh.m);
  // code_offset(row, j, T = 1, PB = row_bytes) = row * row_bytes + j: the tile-interleaved address degenerates to row-major
#define PQ_ENC(DS)                                                                                                               \
  {                                                                                                                              \
    auto kern = pq_encode_kernel<DS>;                                                                                            \
    if (lds > 48 * 1024) COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    kern<<<grid, 256, lds, s>>>(d_cb, (int)sh.m, (int)sh.C, (int)sh.dsub, d_vecs, n, nullptr, 0, lds_cb, d_codes, 1, (int)row_bytes);    \
  }
  switch (sh.dsub) {
    case 4: PQ_ENC(4) break;
    case 8: PQ_ENC(8) break;
    case 12: PQ_ENC(12) break;
    case 16: PQ_ENC(16) break;
    case 24: PQ_ENC(24) break;
    case 32: PQ_ENC(32) break;
    default: PQ_ENC(0) break;
  }
#undef PQ_ENC
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// distance tables of nq queries for the product-quantised HNSW: d_lut [nq][mp][1 << shift] binary16
// d_qmax: nq words of scratch (the per-query table maxima the scale is derived from)
int coltt::pq_lut16_batch(hipStream_t s, const float* d_cb, const PqShape& sh, const float* d_queries, size_t nq, uint32_t mp, uint32_t shift, uint32_t* d_qmax, unsigned short* d_lut) {
  if (nq == 0) return COLTT_OK;
  if (shift < 4 || shift > 8 || (1u << shift) < sh.C) return fail(COLTT_E_INVALID, "pq_lut16: %u centroids do not fit rows of %u entries", sh.C, 1u << shift);
  dim3 grid(ceil_div(mp, 256u >> shift), (uint32_t)nq);
  COLTT_HIP(hipMemsetAsync(d_qmax, 0, nq * 4, s));
#define COLTT_L16(K)                                                                                                                        \
  do {                                                                                                                                      \
    pq_lut16_max_kernel<K><<<grid, 256, 0, s>>>(d_cb, (int)sh.m, (int)sh.C, (int)sh.dsub, d_queries, (int)mp, (int)shift, d_qmax);          \
    pq_lut16_kernel<K><<<grid, 256, 0, s>>>(d_cb, (int)sh.m, (int)sh.C, (int)sh.dsub, d_queries, (int)mp, (int)shift, d_qmax, d_lut);        \
  } while (0)
  if (sh.metric == COLTT_PQ_COSINE) COLTT_L16(0);
  else if (sh.metric == COLTT_PQ_EUCLIDEAN) COLTT_L16(1);
  else COLTT_L16(2);
#undef COLTT_L16
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}
// the largest ||centroid||^2 of a codebook on the device (synchronises the stream)
int coltt::pq_centroid_norm_max(hipStream_t s, const float* d_cb, const PqShape& sh, uint32_t* d_scratch, float* out) {
  COLTT_HIP(hipMemsetAsync(d_scratch, 0, 4, s));
  const uint32_t nc = sh.m * sh.C;
  pq_centroid_norm_max_kernel<<<ceil_div(nc, 256u), 256, 0, s>>>(d_cb, nc, (int)sh.dsub, d_scratch);
  COLTT_HIP(hipGetLastError());
  uint32_t bits = 0;
  COLTT_HIP(hipMemcpyAsync(&bits, d_scratch, 4, hipMemcpyDeviceToHost, s));
  COLTT_HIP(hipStreamSynchronize(s));
  std::memcpy(out, &bits, 4);
  return COLTT_OK;
}

extern "C" {

int coltt_pq_create(uint32_t dim, int metric, uint32_t num_subvectors, uint32_t num_centroids, coltt_handle_t* out) {
  if (!out) return fail(COLTT_E_INVALID, "pq_create: out is NULL");
  if (metric != COLTT_PQ_COSINE && metric != COLTT_PQ_EUCLIDEAN && metric != COLTT_PQ_DOT) return fail(COLTT_E_INVALID, "pq_create: bad metric %d", metric);
  if (num_centroids < 2 || num_centroids > 256) return fail(COLTT_E_INVALID, "pq_create: numCentroids %u outside [2,256]", num_centroids);   // hnsw_common.go:25
  if (num_subvectors < 2) return fail(COLTT_E_INVALID, "pq_create: numSubVectors %u < 2", num_subvectors);                                // hnsw_common.go:28
  if (dim == 0 || dim % num_subvectors) return fail(COLTT_E_INVALID, "pq_create: dim %u is not a multiple of numSubVectors %u", dim, num_subvectors);
  if (num_subvectors > 4096) return fail(COLTT_E_UNSUPPORTED, "pq_create: numSubVectors %u > 4096", num_subvectors);
  COLTT_DEVICE(-1);
  auto p = std::make_shared<Pq>();
  p->dim = dim; p->metric = metric; p->m = num_subvectors; p->C = num_centroids; p->dsub = dim / num_subvectors;
  p->mp = (int)((num_subvectors + 3) & ~3u);
  p->PB = p->mp % 16 == 0 ? 16 : (p->mp % 8 == 0 ? 8 : 4);
  p->T = p->mp / p->PB;
  p->device = coltt_dev_scope_.device();
  COLTT_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  COLTT_TRY(p->cb.reserve((size_t)p->m * p->C * p->dsub * 4));
  COLTT_TRY(p->w_bad.reserve(16));
  *out = Registry::get().add(p);
  return COLTT_OK;
}

int coltt_pq_destroy(coltt_handle_t h) {
  if (!Registry::get().erase(h)) return fail(COLTT_E_NOT_FOUND, "pq_destroy: unknown handle");
  return COLTT_OK;
}

int coltt_pq_set_codebooks(coltt_handle_t h, const float* codebooks) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_set_codebooks: unknown handle");
  if (!codebooks) return fail(COLTT_E_INVALID, "pq_set_codebooks: NULL codebooks");
  WriteLock g(p->rw);
  COLTT_DEVICE(p->device);
  if (p->n) return fail(COLTT_E_INVALID, "pq_set_codebooks: the store holds %llu encoded rows — their codes belong to the current codebooks", (unsigned long long)p->n);
  COLTT_HIP(hipMemcpyAsync(p->cb.p, codebooks, (size_t)p->m * p->C * p->dsub * 4, hipMemcpyHostToDevice, p->stream));
  COLTT_HIP(hipStreamSynchronize(p->stream));
  p->trained = true;
  return COLTT_OK;
}

int coltt_pq_get_codebooks(coltt_handle_t h, float* out) {
  auto p = lookup<Pq>(h);
  if (!p || !out) return fail(COLTT_E_NOT_FOUND, "pq_get_codebooks: unknown handle");
  ReadLock g(p->rw);
  COLTT_DEVICE(p->device);
  if (!p->trained) return fail(COLTT_E_INVALID, "pq_get_codebooks: no codebooks yet");
  COLTT_HIP(hipMemcpy(out, p->cb.p, (size_t)p->m * p->C * p->dsub * 4, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_pq_train(coltt_handle_t h, const float* vecs, size_t n, uint32_t iterations) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_train: unknown handle");
  if (!vecs) return fail(COLTT_E_INVALID, "pq_train: NULL vectors");
  if (n < p->C) return fail(COLTT_E_INVALID, "pq_train: %zu training vectors for %u centroids", n, p->C);
  WriteLock g(p->rw);
  COLTT_DEVICE(p->device);
  if (p->n) return fail(COLTT_E_INVALID, "pq_train: the store holds %llu encoded rows — their codes belong to the current codebooks", (unsigned long long)p->n);
  DevBuf d_v, d_codes;
  COLTT_TRY(d_v.reserve(n * p->dim * 4));
  const uint64_t rows = (n + 63) & ~63ull;
  COLTT_TRY(d_codes.reserve(rows / 64 * p->tile_bytes()));
  COLTT_HIP(hipMemcpyAsync(d_v.p, vecs, n * p->dim * 4, hipMemcpyHostToDevice, p->stream));
  COLTT_HIP(hipMemsetAsync(d_codes.p, 0, d_codes.cap, p->stream));
  const uint64_t cbn = (uint64_t)p->m * p->C * p->dsub;
  pq_train_init_kernel<<<ceil_div(cbn, 256), 256, 0, p->stream>>>(d_v.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, p->cb.as<float>());
  for (uint32_t it = 0; it < iterations; it++) {
    COLTT_TRY(launch_encode(p.get(), p->stream, d_v.as<float>(), n, nullptr, 0, d_codes.as<uint8_t>()));
    pq_train_update_kernel<<<ceil_div(cbn, 256), 256, 0, p->stream>>>(d_v.as<float>(), n, (int)p->m, (int)p->C, (int)p->dsub, d_codes.as<uint8_t>(),
                                                                       p->T, p->PB, p->cb.as<float>());
  }
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipStreamSynchronize(p->stream));
  p->trained = true;
  return COLTT_OK;
}

int coltt_pq_encode(coltt_handle_t h, const float* vecs, size_t n, uint8_t* out_codes) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_encode: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!vecs || !out_codes) return fail(COLTT_E_INVALID, "pq_encode: NULL buffer");
  ReadLock g(p->rw);
  COLTT_DEVICE(p->device);
  if (!p->trained) return fail(COLTT_E_INVALID, "pq_encode: no codebooks yet");
  hipStream_t s = nullptr;   // a stateless call on the default stream (blocking copies around it)
  DevBuf d_v, d_c, d_o;
  COLTT_TRY(d_v.reserve(n * p->dim * 4));
  const uint64_t rows = (n + 63) & ~63ull;
  COLTT_TRY(d_c.reserve(rows / 64 * p->tile_bytes())); COLTT_TRY(d_o.reserve(n * p->m));
  COLTT_HIP(hipMemcpy(d_v.p, vecs, n * p->dim * 4, hipMemcpyHostToDevice));
  COLTT_HIP(hipMemset(d_c.p, 0, d_c.cap));
  COLTT_TRY(launch_encode(p.get(), s, d_v.as<float>(), n, nullptr, 0, d_c.as<uint8_t>()));
  pq_fetch_codes_kernel<<<ceil_div(n * p->m, 256), 256, 0, s>>>(d_c.as<uint8_t>(), p->T, p->PB, 0, n, (int)p->m, d_o.as<uint8_t>());
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy(out_codes, d_o.p, n * p->m, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

static int pq_upsert_impl(Pq* p, const uint64_t* ids, uint64_t first_id, const float* vecs, bool on_device, const uint8_t* codes, size_t n) {
  if (!p->trained) return fail(COLTT_E_INVALID, "pq upsert: no codebooks yet (coltt_pq_set_codebooks / coltt_pq_train)");
  if (codes)   // codes >= NumCentroids are refused before anything is touched
    for (size_t i = 0; i < n * p->m; i++) if (codes[i] >= p->C) return fail(COLTT_E_INVALID, "pq upsert_codes: code %u >= numCentroids %u", (unsigned)codes[i], p->C);
  // append-only dense fast path: ids first_id, first_id + 1, ... continuing the store
  const bool dense_ok = !ids && p->dense && (p->n == 0 || first_id == p->dense_base + p->n);
  PqPlan pl; std::vector<size_t> keep;
  const uint32_t* d_slots = nullptr; uint64_t slot_base = p->n; size_t m_rows = n;
  const float* src_v = vecs; const uint8_t* src_c = codes;
  std::vector<float> v2; std::vector<uint8_t> c2; std::vector<uint32_t> s2;
  if (dense_ok) {
    if (p->n + n > 0xffffffffull) return fail(COLTT_E_UNSUPPORTED, "pq upsert: more than 2^32-1 rows in one store");
    COLTT_TRY(p->reserve(p->n + n));
  } else {
    COLTT_TRY(p->undense());
    COLTT_TRY(plan_upsert(p, ids, first_id, n, pl));
    COLTT_TRY(p->reserve(pl.nn));
    last_wins(pl, n, keep);
    const uint32_t* sl = pl.slots.data();
    if (!keep.empty()) {
      if (on_device) return fail(COLTT_E_UNSUPPORTED, "pq upsert_device: an id is repeated inside the batch");
      s2.reserve(keep.size());
      for (size_t i : keep) {
        s2.push_back(pl.slots[i]);
        if (vecs) v2.insert(v2.end(), vecs + i * p->dim, vecs + (i + 1) * p->dim);
        else c2.insert(c2.end(), codes + i * p->m, codes + (i + 1) * p->m);
      }
      sl = s2.data(); m_rows = s2.size(); src_v = vecs ? v2.data() : nullptr; src_c = codes ? c2.data() : nullptr;
    }
    COLTT_TRY(p->w_slots.reserve(m_rows * 4));
    COLTT_HIP(hipMemcpyAsync(p->w_slots.p, sl, m_rows * 4, hipMemcpyHostToDevice, p->stream));
    d_slots = p->w_slots.as<uint32_t>(); slot_base = 0;
  }
  if (vecs) {
    const float* d_v = src_v;
    if (!on_device) {
      COLTT_TRY(p->w_raw.reserve(m_rows * p->dim * 4));
      COLTT_HIP(hipMemcpyAsync(p->w_raw.p, src_v, m_rows * p->dim * 4, hipMemcpyHostToDevice, p->stream));
      d_v = p->w_raw.as<float>();
    }
    COLTT_TRY(launch_encode(p, p->stream, d_v, m_rows, d_slots, slot_base, p->codes.as<uint8_t>()));
  } else {
    COLTT_TRY(p->w_raw.reserve(m_rows * p->m));
    COLTT_HIP(hipMemcpyAsync(p->w_raw.p, src_c, m_rows * p->m, hipMemcpyHostToDevice, p->stream));
    COLTT_HIP(hipMemsetAsync(p->w_bad.p, 0, 4, p->stream));
    pq_place_codes_kernel<<<ceil_div(m_rows * p->m, 256), 256, 0, p->stream>>>(p->w_raw.as<uint8_t>(), m_rows, (int)p->m, (int)p->C, d_slots, slot_base,
                                                                              p->codes.as<uint8_t>(), p->T, p->PB, p->w_bad.as<uint32_t>());
    COLTT_HIP(hipGetLastError());
  }
  if (!dense_ok && !pl.new_ids.empty())
    COLTT_HIP(hipMemcpyAsync(p->ids.as<uint64_t>() + p->n, pl.new_ids.data(), pl.new_ids.size() * 8, hipMemcpyHostToDevice, p->stream));
  COLTT_HIP(hipStreamSynchronize(p->stream));
  if (dense_ok) { if (p->n == 0) p->dense_base = first_id; p->n += n; }
  else commit_upsert(p, pl);
  return COLTT_OK;
}

int coltt_pq_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_upsert: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!ids || !vecs) return fail(COLTT_E_INVALID, "pq_upsert: NULL input");
  WriteLock g(p->rw);
  COLTT_DEVICE(p->device);
  return pq_upsert_impl(p.get(), ids, 0, vecs, false, nullptr, n);
}

int coltt_pq_upsert_device(coltt_handle_t h, const uint64_t* ids, uint64_t first_id, const float* d_vecs, size_t n) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_upsert_device: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!d_vecs) return fail(COLTT_E_INVALID, "pq_upsert_device: NULL vectors");
  WriteLock g(p->rw);
  COLTT_DEVICE(p->device);
  return pq_upsert_impl(p.get(), ids, first_id, d_vecs, true, nullptr, n);
}

int coltt_pq_upsert_codes(coltt_handle_t h, const uint64_t* ids, const uint8_t* codes, size_t n) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_upsert_codes: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!ids || !codes) return fail(COLTT_E_INVALID, "pq_upsert_codes: NULL input");
  WriteLock g(p->rw);
  COLTT_DEVICE(p->device);
  return pq_upsert_impl(p.get(), ids, 0, nullptr, false, codes, n);
}

int coltt_pq_remove(coltt_handle_t h, const uint64_t* ids, size_t n) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_remove: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!ids) return fail(COLTT_E_INVALID, "pq_remove: NULL ids");
  WriteLock g(p->rw);
  COLTT_DEVICE(p->device);
  COLTT_TRY(p->undense());
  for (size_t i = 0; i < n; i++) {
    auto it = p->id2slot.find(ids[i]);
    if (it == p->id2slot.end()) continue;   // delete() of a missing key is a no-op in Go
    const uint32_t s = it->second; const uint64_t last = p->n - 1;
    p->id2slot.erase(it);
    if (s != last) {   // the last row moves into the hole: scans cover a dense prefix
      pq_move_row_kernel<<<(p->mp + 255) / 256, 256, 0, p->stream>>>(p->codes.as<uint8_t>(), p->T, p->PB, p->mp, s, last);
      const uint64_t moved = p->h_ids[last];
      p->h_ids[s] = moved; p->id2slot[moved] = s;
      COLTT_HIP(hipMemcpyAsync(p->ids.as<uint64_t>() + s, p->ids.as<uint64_t>() + last, 8, hipMemcpyDeviceToDevice, p->stream));
    }
    p->h_ids.pop_back();
    p->n--;
  }
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipStreamSynchronize(p->stream));
  return COLTT_OK;
}

int coltt_pq_len(coltt_handle_t h, uint64_t* out) {
  auto p = lookup<Pq>(h);
  if (!p || !out) return fail(COLTT_E_NOT_FOUND, "pq_len: unknown handle");
  ReadLock g(p->rw);
  *out = p->n;
  return COLTT_OK;
}

int coltt_pq_fetch_codes(coltt_handle_t h, uint64_t first_slot, uint64_t n, uint8_t* out_codes, uint64_t* out_ids) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_fetch_codes: unknown handle");
  if (n == 0) return COLTT_OK;
  ReadLock g(p->rw);
  COLTT_DEVICE(p->device);
  if (first_slot + n > p->n) return fail(COLTT_E_INVALID, "pq_fetch_codes: range outside [0,%llu)", (unsigned long long)p->n);
  if (out_codes) {
    DevBuf d_o;
    COLTT_TRY(d_o.reserve(n * p->m));
    pq_fetch_codes_kernel<<<ceil_div(n * p->m, 256), 256>>>(p->codes.as<uint8_t>(), p->T, p->PB, first_slot, n, (int)p->m, d_o.as<uint8_t>());
    COLTT_HIP(hipGetLastError());
    COLTT_HIP(hipMemcpy(out_codes, d_o.p, n * p->m, hipMemcpyDeviceToHost));
  }
  if (out_ids) for (uint64_t i = 0; i < n; i++) out_ids[i] = p->dense ? p->dense_base + first_slot + i : p->h_ids[first_slot + i];
  return COLTT_OK;
}

int coltt_pq_lut(coltt_handle_t h, const float* query, float* out_lut) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_lut: unknown handle");
  if (!query || !out_lut) return fail(COLTT_E_INVALID, "pq_lut: NULL buffer");
  ReadLock g(p->rw);
  COLTT_DEVICE(p->device);
  if (!p->trained) return fail(COLTT_E_INVALID, "pq_lut: no codebooks yet");
  DevBuf d_q, d_l;
  COLTT_TRY(d_q.reserve((size_t)p->dim * 4)); COLTT_TRY(d_l.reserve((size_t)p->mp * 256 * 4));
  COLTT_HIP(hipMemcpy(d_q.p, query, (size_t)p->dim * 4, hipMemcpyHostToDevice));
  dim3 grid(p->mp, 1);
  if (p->metric == COLTT_PQ_COSINE) pq_lut_kernel<0><<<grid, 256>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_q.as<float>(), p->mp, d_l.as<float>(), nullptr, nullptr, nullptr);
  else if (p->metric == COLTT_PQ_EUCLIDEAN) pq_lut_kernel<1><<<grid, 256>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_q.as<float>(), p->mp, d_l.as<float>(), nullptr, nullptr, nullptr);
  else pq_lut_kernel<2><<<grid, 256>>>(p->cb.as<float>(), (int)p->m, (int)p->C, (int)p->dsub, d_q.as<float>(), p->mp, d_l.as<float>(), nullptr, nullptr, nullptr);
  COLTT_HIP(hipGetLastError());
  COLTT_HIP(hipMemcpy2D(out_lut, (size_t)p->C * 4, d_l.p, 256 * 4, (size_t)p->C * 4, p->m, hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_pq_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_search: unknown handle");
  if (nq && (!queries || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "pq_search: NULL buffer");
  ReadLock g(p->rw);
  COLTT_DEVICE(p->device);
  CtxLease<PCtx> ctx(p->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return pq_search_common(p.get(), ctx.c, queries, false, nq, k, out_ids, out_scores, out_counts, false);
}

int coltt_pq_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, uint64_t* d_out_ids, float* d_out_scores,
                           uint32_t* d_out_counts) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_search_device: unknown handle");
  if (nq && (!d_queries || !d_out_ids || !d_out_scores || !d_out_counts)) return fail(COLTT_E_INVALID, "pq_search_device: NULL buffer");
  ReadLock g(p->rw);
  COLTT_DEVICE(p->device);
  CtxLease<PCtx> ctx(p->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return pq_search_common(p.get(), ctx.c, d_queries, true, nq, k, d_out_ids, d_out_scores, d_out_counts, true);
}

int coltt_pq_last_kernel_ms(coltt_handle_t h, float* out_search_ms, float* out_scan_ms, uint64_t* out_scan_rows) {
  auto p = lookup<Pq>(h);
  if (!p) return fail(COLTT_E_NOT_FOUND, "pq_last_kernel_ms: unknown handle");
  if (out_search_ms) *out_search_ms = p->last_ms.load();
  if (out_scan_ms) *out_scan_ms = p->last_scan_ms.load();
  if (out_scan_rows) *out_scan_rows = p->last_scan_rows.load();
  return COLTT_OK;
}

}  // extern "C"
