// rows8.hpp — "eight lanes per row": the LINE-TRANSPOSED layout of an index's stored vectors and the distance core that walks it.  (Rounds 3-4 kept a
// second, transposed COPY; since round 5 `rows` itself is stored in this layout for the shapes below — the index's ONE row array — and every host
// read-back un-permutes through r8_index.  "rows8" below = the row array in that layout.)
//
// Why (profiles/r03_gather_*.jsonl, r03_pmc_walk_utcl*.csv): with a row owned by a lane PAIR (exact.hpp) one load
// instruction of a wave touches 32 rows x 32 bytes — 32 different cache lines and 32 translations, and a line is complete only after
// four consecutive instructions.  On a 10 M x 768 index that is a UTCL1 miss rate of 9-17 % and an L2 TLB that is busy 95-99 % of the
// kernel.  Eight lanes per row make an instruction touch 8 rows x ONE WHOLE 128-byte line (6 % UTCL1 misses in tools/micro/gather.hip).
//
// The reference's AVX order (avx.cpp:15-32, 51-75) puts element i into partial sum i mod 8, partial sums growing in increasing i.
// Lane j of an 8-lane group therefore has to own residue j — every 8th element — while a coalesced 16-byte load hands a lane 8 (f16)
// or 4 (f32) CONSECUTIVE elements.  Instead of transposing through LDS per evaluation (hnsw_lat.hpp) the transposition is done ONCE, at
// ingest: rows8 holds every row with each 128-byte line rewritten so that its 16-byte chunk r carries residue r's S = 16 / elem_bytes
// consecutive AVX steps:
//        rows8 element (l * 8 + r) * S + t   =   row element 8 * (S * l + t) + r            l = line, r = residue, t = step in the line
// Lane r of the group loads chunk r of every line and adds its products in increasing step order: the same values in the same order
// as the pair-owned walk, hence the same bits.  The query is staged in LDS with the same permutation (qp), so lane r reads its S
// query elements of a line with one (f32 rows) or two (2-byte rows) ds_read_b128.
//
// The layout is an internal one: writers transpose at ingest (through a natural-order staging block), Commit / Get / fetch and the quantiser's Encode
// un-permute element indexes, no stream format ever sees it.  It exists for f32 / 2-byte rows whose byte length is a multiple of 128 (dim % 32 == 0 /
// dim % 64 == 0: 128, 256, 512, 768, 1024, 1536 ...; dim >= 256 by default); other shapes keep natural-order rows and the pair-owned walk.
#pragma once
#include <type_traits>
#include "exact.hpp"

namespace coltt {
namespace dev {

template <int QUANT> __device__ __host__ __forceinline__ constexpr int rows8_steps() { return QUANT == Q_NONE ? 4 : 8; }   // AVX steps per 128-byte line

// n natural-order rows at `rows` -> their line-transposed form at `rows8` (same stride; row i of the source is row i of the destination).
// One thread per 16-byte destination chunk.
template <int QUANT>
__global__ __launch_bounds__(256) void rows8_permute_kernel(const uint8_t* __restrict__ rows, uint8_t* __restrict__ rows8, size_t stride, int dim, uint64_t n) {
  constexpr int S = rows8_steps<QUANT>();
  constexpr int EB = 16 / S;                    // bytes per element (4 | 2)
  const int chunks = dim * EB / 16;             // 16-byte chunks per row (dim * EB is a multiple of 128)
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * (uint64_t)chunks) return;
  const uint64_t row = t / chunks;
  const int c = (int)(t % chunks), l = c >> 3, r = c & 7;
  const uint8_t* src = rows + row * stride;
  uint8_t* dst = rows8 + row * stride + (size_t)c * 16;
  if constexpr (EB == 4) {
    f32x4 v;
#pragma unroll
    for (int s = 0; s < 4; s++) v[s] = *reinterpret_cast<const float*>(src + (size_t)(8 * (4 * l + s) + r) * 4);
    *reinterpret_cast<f32x4*>(dst) = v;
  } else {
    unsigned short h[8];
#pragma unroll
    for (int s = 0; s < 8; s++) h[s] = *reinterpret_cast<const unsigned short*>(src + (size_t)(8 * (8 * l + s) + r) * 2);
    u32x4e v = {(uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16), (uint32_t)h[4] | ((uint32_t)h[5] << 16),
                (uint32_t)h[6] | ((uint32_t)h[7] << 16)};
    *reinterpret_cast<u32x4e*>(dst) = v;
  }
}

// index of query element e in the permuted LDS copy qp
template <int QUANT> __device__ __forceinline__ int rows8_qindex(int e) {
  constexpr int S = rows8_steps<QUANT>();
  const int step = e >> 3, r = e & 7;
  return ((step / S) * 8 + r) * S + (step % S);
}

// 8-lane sum of one accumulator per lane in the reference's tree ((l0+l1)+(l2+l3))+((l4+l5)+(l6+l7)); result in all 8 lanes.
// DPP only (quad_perm, row_half_mirror): every lane of the wave must be active.
__device__ __forceinline__ float group8_hsum(float a) {
  const float b = a + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
  const float c = b + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, b), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
  return c + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, c), 0x141, 0xf, 0xf, true));            // row_half_mirror: lane i <-> 7 - i
}

// Distance(query, row) of ROWS rows per 8-lane group at once (ROWS x U..2U 16-byte loads per lane in flight); rj = this lane's
// residue (lane & 7).  row8[i] -> the rows8 copy of row i, qp -> the permuted query in LDS, nl = 128-byte lines per row.
// live[i] == false: this group has no i-th row in this pass — row8[i] then points at some live row, the result is dropped by the caller.
#ifndef COLTT_G8_PEEL   // A/B knob: see the burst loop below
#define COLTT_G8_PEEL 0
#endif
template <int METRIC, int QUANT, int ROWS, int U, bool ONEBURST = false, bool NT = false>
__device__ __forceinline__ void group8_distance(const uint8_t* const (&row8)[ROWS], const bool (&live)[ROWS], const float* __restrict__ qp, int nl,
                                                float qnorm, const float (&rnorm)[ROWS], int rj, float (&out)[ROWS]) {
  constexpr int S = rows8_steps<QUANT>();
  float acc[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; i++) acc[i] = 0.f;
  const float* qb = qp + rj * S;
  struct Raw { u32x4e v[ROWS]; };
  // (the loads are NOT predicated on `live`: wrapping each of them in its own EXEC save / restore broke the back-to-back issue of a burst —
  //  10 M x 768 f32, ef 128: 19.9 -> 27.3 ms per 10 k queries, profiles/r04d_ev8_variants.md; an idle group re-reads a live row instead)
#define COLTT_G8_LD(L, DST) { _Pragma("unroll") for (int i = 0; i < ROWS; i++) DST.v[i] = row_ld<NT>(reinterpret_cast<const u32x4e*>(row8[i] + (size_t)(L) * 128 + rj * 16)); }
#define COLTT_G8_CS(RAW, L)                                                                                                   \
  {                                                                                                                           \
    if constexpr (QUANT == Q_NONE) {                                                                                          \
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + (L) * 32);                                                        \
      _Pragma("unroll") for (int i = 0; i < ROWS; i++) {                                                                      \
        const f32x4 x = __builtin_bit_cast(f32x4, RAW.v[i]);                                                                  \
        _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                                       \
          if constexpr (METRIC == M_COS) { const float p = q0[s] * x[s]; acc[i] = acc[i] + p; }                               \
          else { const float d = q0[s] - x[s]; const float p = d * d; acc[i] = acc[i] + p; }                                  \
        }                                                                                                                     \
      }                                                                                                                       \
    } else {                                                                                                                  \
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + (L) * 64), q1 = *reinterpret_cast<const f32x4*>(qb + (L) * 64 + 4); \
      _Pragma("unroll") for (int i = 0; i < ROWS; i++) {                                                                      \
        const u32x2e lo = {RAW.v[i].x, RAW.v[i].y}, hi = {RAW.v[i].z, RAW.v[i].w};                                            \
        const f32x4 x0 = __builtin_convertvector(__builtin_bit_cast(f16x4, lo), f32x4);                                       \
        const f32x4 x1 = __builtin_convertvector(__builtin_bit_cast(f16x4, hi), f32x4);                                       \
        _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                                       \
          if constexpr (METRIC == M_COS) { const float p = q0[s] * x0[s]; acc[i] = acc[i] + p; }                              \
          else { const float d = q0[s] - x0[s]; const float p = d * d; acc[i] = acc[i] + p; }                                 \
        }                                                                                                                     \
        _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                                       \
          if constexpr (METRIC == M_COS) { const float p = q1[s] * x1[s]; acc[i] = acc[i] + p; }                              \
          else { const float d = q1[s] - x1[s]; const float p = d * d; acc[i] = acc[i] + p; }                                 \
        }                                                                                                                     \
      }                                                                                                                       \
    }                                                                                                                         \
  }
  // bursts of UU lines, the next burst requested before the current one is consumed (2 x ROWS x UU 16-byte loads per lane in flight at most)
  auto bursts = [&](auto uc) {
    constexpr int UU = decltype(uc)::value;
    const int nb = nl / UU;
    Raw cur[UU], nxt[UU];
    if (nb > 0) {
#pragma unroll
      for (int u = 0; u < UU; u++) COLTT_G8_LD(u, cur[u])
    }
#if COLTT_G8_PEEL
    // the last burst peeled off: inside the loop the next burst is requested UNCONDITIONALLY (a branch around loads in flight makes the compiler wait for all of
    // them — vmcnt(0) — in front of the first use of the current burst: see group8_stream)
    for (int b = 0; b + 1 < nb; b++) {
#pragma unroll
      for (int u = 0; u < UU; u++) COLTT_G8_LD((b + 1) * UU + u, nxt[u])
#pragma unroll
      for (int u = 0; u < UU; u++) COLTT_G8_CS(cur[u], b * UU + u)
#pragma unroll
      for (int u = 0; u < UU; u++) cur[u] = nxt[u];
    }
    if (nb > 0) {
#pragma unroll
      for (int u = 0; u < UU; u++) COLTT_G8_CS(cur[u], (nb - 1) * UU + u)
    }
#else
    for (int b = 0; b < nb; b++) {
      if (b + 1 < nb) {
#pragma unroll
        for (int u = 0; u < UU; u++) COLTT_G8_LD((b + 1) * UU + u, nxt[u])
      }
#pragma unroll
      for (int u = 0; u < UU; u++) COLTT_G8_CS(cur[u], b * UU + u)
#pragma unroll
      for (int u = 0; u < UU; u++) cur[u] = nxt[u];
    }
#endif
    // lines beyond whole bursts (nl % UU; the whole row when nl < UU — 128-d f32 rows are 4 lines): ONE predicated burst, every load
    // in flight before the first is consumed (a line-by-line loop here made a short row cost nl dependent round trips: 1 M x 128 f32,
    // one query, 0.111 -> 0.191 ms in the first bench run of this core)
    const int l0 = nb * UU;
    if (l0 < nl) {
#pragma unroll
      for (int u = 0; u < UU; u++) if (l0 + u < nl) COLTT_G8_LD(l0 + u, cur[u])
#pragma unroll
      for (int u = 0; u < UU; u++) if (l0 + u < nl) COLTT_G8_CS(cur[u], l0 + u)
    }
  };
  if constexpr (ONEBURST) {
    // rows of exactly U lines (768 x 2-byte with U = 12): the WHOLE row of every one of the ROWS rows in flight at once, no second buffer — the registers the
    // double-buffered form spends on `nxt` carry a second row instead (ROWS = 2: 16 rows per pass at the register cost of 8).  Any other length: bursts of U / 2.
    if (nl == U) {
      Raw cur[U];
#pragma unroll
      for (int u = 0; u < U; u++) COLTT_G8_LD(u, cur[u])
#pragma unroll
      for (int u = 0; u < U; u++) COLTT_G8_CS(cur[u], u)
    } else bursts(std::integral_constant<int, (U / 2 > 0 ? U / 2 : 1)>());
  } else bursts(std::integral_constant<int, U>());
#undef COLTT_G8_LD
#undef COLTT_G8_CS
#pragma unroll
  for (int i = 0; i < ROWS; i++) {
    const float s = group8_hsum(acc[i]);
    if constexpr (METRIC == M_COS) out[i] = cos_epilogue(s, qnorm, rnorm[i]);
    else out[i] = go_sqrt(s);
  }
}

// ONE row per lane group, the rows of ALL passes as one stream of bursts (COLTT_G8_STREAM): group8_distance leaves a bubble at every pass boundary — the last
// burst of a row is consumed with nothing in flight, and the first burst of the next pass's row is requested only after the distance has been reduced and
// handed over.  Here the next burst is ALWAYS in flight while the current one is consumed, across row boundaries too (its row pointer comes from the compacted
// list in LDS).  Ping-pong buffers: no register moves.  Same lines, same order per row, same reduction: same bits.  Requires nl % U == 0, nl >= U.
// s_nb / s_nr: the compacted fresh neighbours (slot, norm) of the chunk; s_d: their distances, written by lane 0 of the group that evaluated the row.
template <int METRIC, int QUANT, int U, bool NT, bool ADJN>
__device__ __forceinline__ void group8_stream(const uint8_t* __restrict__ rows8, size_t stride, const float* __restrict__ norms, const uint32_t* s_nb, const float* s_nr,
                                              float* s_d, uint32_t nf, int grp, int rj, const float* __restrict__ qp, int nl, float qnorm) {
  constexpr int S = rows8_steps<QUANT>();
  const float* qb = qp + rj * S;
  const int nbur = nl / U;
  const uint32_t total = ((nf + 7u) >> 3) * (uint32_t)nbur;
  u32x4e A[U], B[U];
  auto row_of = [&](uint32_t pass) -> const uint8_t* {
    const uint32_t idx = pass * 8u + (uint32_t)grp;
    return rows8 + (size_t)s_nb[idx < nf ? idx : 0u] * stride + rj * 16;   // an idle group re-reads a live row (see group8_distance)
  };
  const uint8_t* rp = row_of(0);
#pragma unroll
  for (int u = 0; u < U; u++) A[u] = row_ld<NT>(reinterpret_cast<const u32x4e*>(rp + (size_t)u * 128));
  float acc = 0.f;
  uint32_t pass = 0; int b = 0;
  // LOADNEXT is a literal: the step that has a successor issues its loads UNCONDITIONALLY (a branch around loads in flight makes the compiler's wait-count
  // insertion give up at the merge and wait for everything — vmcnt(0) — in front of the first use: no overlap at all; seen in the first build of this function)
#define COLTT_G8S_STEP(CUR, NXT, LOADNEXT)                                                                                   \
  {                                                                                                                         \
    int nb_ = b + 1; uint32_t np_ = pass; const uint8_t* nrp = rp;                                                          \
    if (nb_ == nbur) { nb_ = 0; np_ = pass + 1u; }                                                                          \
    if (LOADNEXT) {                                                                                                         \
      if (nb_ == 0) nrp = row_of(np_);                                                                                      \
      _Pragma("unroll") for (int u = 0; u < U; u++) NXT[u] = row_ld<NT>(reinterpret_cast<const u32x4e*>(nrp + (size_t)(nb_ * U + u) * 128)); \
    }                                                                                                                       \
    _Pragma("unroll") for (int u = 0; u < U; u++) {                                                                         \
      const int L = b * U + u;                                                                                              \
      if constexpr (QUANT == Q_NONE) {                                                                                      \
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + L * 32);                                                      \
        const f32x4 x = __builtin_bit_cast(f32x4, CUR[u]);                                                                  \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++) {                                                                  \
          if constexpr (METRIC == M_COS) { const float p = q0[s_] * x[s_]; acc = acc + p; }                                 \
          else { const float d = q0[s_] - x[s_]; const float p = d * d; acc = acc + p; }                                    \
        }                                                                                                                   \
      } else {                                                                                                              \
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(qb + L * 64), q1 = *reinterpret_cast<const f32x4*>(qb + L * 64 + 4); \
        const u32x2e lo = {CUR[u].x, CUR[u].y}, hi = {CUR[u].z, CUR[u].w};                                                  \
        const f32x4 x0 = __builtin_convertvector(__builtin_bit_cast(f16x4, lo), f32x4);                                     \
        const f32x4 x1 = __builtin_convertvector(__builtin_bit_cast(f16x4, hi), f32x4);                                     \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++) {                                                                  \
          if constexpr (METRIC == M_COS) { const float p = q0[s_] * x0[s_]; acc = acc + p; }                                \
          else { const float d = q0[s_] - x0[s_]; const float p = d * d; acc = acc + p; }                                   \
        }                                                                                                                   \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; s_++) {                                                                  \
          if constexpr (METRIC == M_COS) { const float p = q1[s_] * x1[s_]; acc = acc + p; }                                \
          else { const float d = q1[s_] - x1[s_]; const float p = d * d; acc = acc + p; }                                   \
        }                                                                                                                   \
      }                                                                                                                     \
    }                                                                                                                       \
    if (b == nbur - 1) {   /* wave-uniform: every group is at the same burst of its row */                                   \
      const float sum = group8_hsum(acc);                                                                                   \
      const uint32_t idx = pass * 8u + (uint32_t)grp;                                                                       \
      const bool live = idx < nf;                                                                                           \
      float out;                                                                                                            \
      if constexpr (METRIC == M_COS) {                                                                                      \
        float rn;                                                                                                           \
        if constexpr (ADJN) rn = s_nr[live ? idx : 0u]; else rn = norms[s_nb[live ? idx : 0u]];                             \
        out = cos_epilogue(sum, qnorm, rn);                                                                                 \
      } else out = go_sqrt(sum);                                                                                            \
      if (rj == 0 && live) s_d[idx] = out;                                                                                  \
      acc = 0.f;                                                                                                            \
    }                                                                                                                       \
    b = nb_; pass = np_; rp = nrp;                                                                                          \
  }
  uint32_t t = 0;
  for (; t + 3u <= total; t += 2) {   // both steps have a successor
    COLTT_G8S_STEP(A, B, 1)
    COLTT_G8S_STEP(B, A, 1)
  }
  if (total - t == 2u) { COLTT_G8S_STEP(A, B, 1) COLTT_G8S_STEP(B, A, 0) }
  else if (total - t == 1u) { COLTT_G8S_STEP(A, B, 0) }
#undef COLTT_G8S_STEP
}

}  // namespace dev
}  // namespace coltt
