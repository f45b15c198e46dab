// flat.hip — edge FLAT store on the GPU: replaces {none,f16,f8,bf16}VecSpace's vector half
// (edge/none_vectorstore.go:66-253, edge/f16_vectorstore.go:67-262 and the f8/bf16 twins).
//
// HBM layout (one store = one collection shard on one GPU):
//   rows   [cap][row_stride]  stored ("lowered") vectors, dense in slot order, row_stride = dim*s rounded to 16 B
//   norms  [cap] f32          ||row||^2 in AVX order (the reference recomputes it per pair; same bits)
//   ids    [cap] u64          slot -> id           (absent in dense-id mode: id = dense_base + slot)
// Removal swaps the last row into the hole, so a scan is always over the dense prefix [0, n).
//
// Kernels: prep_rows (Normalize + Lower), row_norms, prep_queries, flat_scan (exact-order distances for a
// tile of QB queries per row read, threshold-filtered candidate emission), flat_select (radix-select +
// rank-sort of the candidates by the canonical (score, id) order).
#include <algorithm>
#include <atomic>

#include "common.hpp"
#include "exact.hpp"
#include "prep.hpp"
#include "flat_mfma.hpp"
#include "select.hpp"

using namespace coltt;
using namespace coltt::dev;

namespace {

constexpr uint64_t ROW_SLACK = 512;  // see Flat::reserve
constexpr int QB = 16;            // queries per row read in the exact scan (dim <= 1024; 4 above that: LDS query tile)

// ---------------------------------------------------------------------------------------------------
// Exact-order scan: the hot loop of VertexSearch (none_vectorstore.go:136-147) for QB queries at once.
// Each wave owns 32 rows per iteration (lane pair per row); every row chunk read from HBM is used for
// all QB queries (queries broadcast from LDS).  Survivors of the per-query threshold are appended to a
// candidate list as (score_key << 32 | slot).
// ---------------------------------------------------------------------------------------------------
// exact-order score keys of ONE row (a lane pair: `half` takes elements 4*half .. 4*half+3 of every 8) against the QB queries of
// the LDS tile qs[QB][dimp]; both lanes of the pair return the keys
template <int METRIC, int QUANT, int QB, int U = 4>   // U: 16-byte loads per lane in flight (x2 with the prefetched next batch)
__device__ __forceinline__ void flat_eval_row(const uint8_t* __restrict__ row, float rn, const float* __restrict__ qs, int dimp, int dim,
                                              int half, const float (&qn)[QB], uint32_t (&sk)[QB]) {
  const int n8 = dim >> 3;
  f32x4 acc[QB];
#pragma unroll
  for (int q = 0; q < QB; q++) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nb = n8 / U;
  typename Raw4<QUANT>::type cur[U], nxt[U];  // raw bits stay in the pipeline registers; decoded at use
  if (nb > 0) {
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = load_raw4<QUANT>(row, 8 * u + 4 * half);
  }
  for (int b = 0; b < nb; b++) {
    if (b + 1 < nb) {
#pragma unroll
      for (int u = 0; u < U; u++) nxt[u] = load_raw4<QUANT>(row, 8 * ((b + 1) * U + u) + 4 * half);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const float* qp = qs + 8 * (b * U + u) + 4 * half;
      const f32x4 rc = decode4<QUANT>(cur[u]);
#pragma unroll
      for (int q = 0; q < QB; q++) {
        f32x4 qq = *reinterpret_cast<const f32x4*>(qp + q * dimp);
        if constexpr (METRIC == M_COS) { f32x4 pr = qq * rc; acc[q] = acc[q] + pr; }
        else { f32x4 d = qq - rc; f32x4 pr = d * d; acc[q] = acc[q] + pr; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = nxt[u];
  }
  for (int t = nb * U; t < n8; t++) {
    f32x4 r = load4<QUANT>(row, 8 * t + 4 * half);
    const float* qp = qs + 8 * t + 4 * half;
#pragma unroll
    for (int q = 0; q < QB; q++) {
      f32x4 qq = *reinterpret_cast<const f32x4*>(qp + q * dimp);
      if constexpr (METRIC == M_COS) { f32x4 pr = qq * r; acc[q] = acc[q] + pr; }
      else { f32x4 d = qq - r; f32x4 pr = d * d; acc[q] = acc[q] + pr; }
    }
  }
#pragma unroll
  for (int q = 0; q < QB; q++) {
    float s = pair_hsum(acc[q], half);
    for (int e = n8 * 8; e < dim; e++) {
      float r = load1<QUANT>(row, e);
      if constexpr (METRIC == M_COS) s += qs[q * dimp + e] * r;
      else { float d = qs[q * dimp + e] - r; s += d * d; }
    }
    float score;
    if constexpr (METRIC == M_COS) score = cos_epilogue(s, qn[q], rn);
    else score = go_sqrt(s);
    sk[q] = score_key(score);
  }
}

template <int METRIC, int QUANT, bool GATHER, int QB>
__global__ __launch_bounds__(256) void flat_scan_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms,
    const uint32_t* __restrict__ gather, uint64_t begin, uint64_t end, const float* __restrict__ q_eff,
    const float* __restrict__ qnorms, int nq_grp, int dim, const uint32_t* __restrict__ thr, int nearest,
    unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t cap) {
  // [QB][dimp]: the query rows are read with ds_read_b128, so each starts on a 16-byte boundary (dim % 4 != 0 is legal)
  extern __shared__ __attribute__((aligned(16))) float qs[];
  const int dimp = (dim + 3) & ~3;
  for (int i = threadIdx.x; i < QB * dimp; i += blockDim.x) {
    const int q = i / dimp, e = i - q * dimp;
    qs[i] = (q < nq_grp && e < dim) ? q_eff[(size_t)q * dim + e] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane & 1, p = lane >> 1;
  float qn[QB];
  uint32_t th[QB];
#pragma unroll
  for (int q = 0; q < QB; q++) { qn[q] = q < nq_grp ? qnorms[q] : 0.f; th[q] = q < nq_grp ? thr[q] : 0u; }
  const uint64_t ngroups = (end - begin + 31) / 32;
  for (uint64_t g = (uint64_t)blockIdx.x * 4 + wave; g < ngroups; g += (uint64_t)gridDim.x * 4) {
    uint64_t pos = begin + g * 32 + p;
    bool valid = pos < end;
    uint32_t slot = GATHER ? gather[valid ? pos : begin] : (uint32_t)(valid ? pos : begin);
    const uint8_t* row = rows + (size_t)slot * stride;
    float rn = 0.f;
    if constexpr (METRIC == M_COS) rn = norms[slot];
    uint32_t sk[QB];
    flat_eval_row<METRIC, QUANT, QB>(row, rn, qs, dimp, dim, half, qn, sk);
#pragma unroll
    for (int q = 0; q < QB; q++) {
      bool pass = valid && half == 0 && q < nq_grp && (nearest ? sk[q] <= th[q] : sk[q] >= th[q]);
      if (pass) {
        uint32_t idx = atomicAdd(&cnt[q], 1u);
        if (idx < cap) cand[(size_t)q * cap + idx] = ((unsigned long long)sk[q] << 32) | slot;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Small batches (<= 4 queries, k <= 64) in ONE launch: the reference serves one query per RPC (edge/edge_search.go -> VertexSearch,
// none_vectorstore.go:104-180), and a chain of scan + select launches over a few hundred thousand rows is all launch gaps.
//   every wave   scans its 32-row groups in exact order and keeps its own k best per query as a sorted 64-lane register array
//                (ascending (key', id'): key' = score key ^ flip, id' = id ^ idflip — the canonical (score, id) order, best first);
//   every block  merges its four waves' lists (LDS rank sort of <= 256 entries), writes its k best records to HBM and lowers
//                bucket[q][block % k] to its BEST key': the k buckets end up holding k different rows, so their maximum bounds the
//                collection's k-th best from above — closely (about its k ln k-th best);
//   the LAST block to finish (ticket counter) gathers the records with key' <= max bucket[q][.] — a few dozen — into the candidate
//                list and runs the ordinary selection (select.hpp) on them; it leaves the counter and the buckets reset.
// Same (score, id) total order and same exact-order score bits as flat_scan_kernel + flat_select_kernel: identical answers.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t ONE_KMAX = 64;
constexpr uint32_t ONE_MAX_BLOCKS = 2048;
constexpr int ONE_QMAX = 4;
typedef unsigned long long OneRec;   // key' << 32 | slot (the id' that ordered it inside the block is recomputed by the selection)
struct OneState { uint32_t done, pad[3]; uint32_t bucket[ONE_QMAX * ONE_KMAX]; };
constexpr size_t ONE_STATE_BYTES = 2048;
static_assert(sizeof(OneState) <= ONE_STATE_BYTES && ONE_QMAX * ONE_KMAX == 256, "the last block resets one bucket per thread");

static __device__ __forceinline__ void one_wave_sync() {   // LDS written by some lanes of this wave, read by others
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int METRIC, int QUANT, bool GATHER, int QB>
__global__ __launch_bounds__(256) void flat_one_kernel(
    const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, const uint32_t* __restrict__ gather, uint64_t total,
    const float* __restrict__ q_eff, const float* __restrict__ qnorms, int nq_grp, int dim, uint32_t k, int nearest,
    const uint64_t* __restrict__ ids, uint64_t dense_base, OneRec* __restrict__ recs, OneState* __restrict__ st,
    unsigned long long* __restrict__ cand, uint32_t* __restrict__ cnt, uint32_t* __restrict__ thr, uint32_t cap, uint32_t* __restrict__ ovf,
    uint64_t* __restrict__ out_ids, float* __restrict__ out_scores, uint32_t* __restrict__ out_counts) {
  extern __shared__ __attribute__((aligned(16))) float qs[];
  __shared__ __attribute__((aligned(16))) uint32_t m_key[4 * ONE_KMAX];
  __shared__ __attribute__((aligned(16))) uint64_t m_id[4 * ONE_KMAX];
  __shared__ uint32_t m_slot[4 * ONE_KMAX];
  __shared__ uint32_t s_wn[4], s_last, s_n, s_bmax;
  constexpr int U = QB == 1 ? 8 : 4;
  const int dimp = (dim + 3) & ~3;
  for (int i = threadIdx.x; i < QB * dimp; i += blockDim.x) {
    const int q = i / dimp, e = i - q * dimp;
    qs[i] = (q < nq_grp && e < dim) ? q_eff[(size_t)q * dim + e] : 0.f;
  }
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane & 1, p = lane >> 1;
  const uint32_t flip = nearest ? 0u : 0xffffffffu;
  const uint64_t idflip = nearest ? 0ull : ~0ull;
  uint32_t* const w_key = m_key + wave * 64; uint64_t* const w_id = m_id + wave * 64; uint32_t* const w_slot = m_slot + wave * 64;   // this wave's merge scratch
  float qn[QB];
  uint32_t bk[QB], bs[QB], bn[QB], tk[QB];
  uint64_t bi[QB], ti[QB];
#pragma unroll
  for (int q = 0; q < QB; q++) { qn[q] = q < nq_grp ? qnorms[q] : 0.f; bk[q] = 0xffffffffu; bs[q] = 0xffffffffu; bi[q] = ~0ull; bn[q] = 0; tk[q] = 0xffffffffu; ti[q] = ~0ull; }
  const uint64_t ngroups = (total + 31) / 32;
  for (uint64_t g = (uint64_t)blockIdx.x * 4 + wave; g < ngroups; g += (uint64_t)gridDim.x * 4) {
    const uint64_t pos = g * 32 + p;
    const bool valid = pos < total;
    const uint32_t slot = GATHER ? gather[valid ? pos : 0] : (uint32_t)(valid ? pos : 0);
    const uint8_t* row = rows + (size_t)slot * stride;
    float rn = 0.f;
    if constexpr (METRIC == M_COS) rn = norms[slot];
    const uint64_t rid = (ids ? ids[slot] : dense_base + slot) ^ idflip;
    uint32_t sk[QB];
    flat_eval_row<METRIC, QUANT, QB, U>(row, rn, qs, dimp, dim, half, qn, sk);
#pragma unroll
    for (int q = 0; q < QB; q++) {
      const uint32_t kq = sk[q] ^ flip;
      const bool pass = valid && half == 0 && q < nq_grp && (bn[q] < k || kq < tk[q] || (kq == tk[q] && rid < ti[q]));
      const unsigned long long m = __ballot(pass);
      if (!m) continue;
      // Parallel merge of the passing rows into the sorted list: one uniform loop over the candidates gives every existing entry
      // the number of candidates in front of it and every candidate its rank among the candidates plus the number of existing
      // entries in front of it — final positions, a permutation — then one scatter through the wave's LDS scratch.
      const bool have = (uint32_t)lane < bn[q];
      uint32_t e_shift = 0, c_rank = 0, c_below = 0;
      for (unsigned long long mm = m; mm; mm &= mm - 1) {
        const int j = __builtin_ctzll(mm);
        const uint32_t kc = (uint32_t)__builtin_amdgcn_readlane((int)kq, j);
        bool c_lt_e = kc < bk[q], c_lt_c = kc < kq;   // candidate j in front of this lane's entry / of this lane's candidate
        if (__ballot((have && kc == bk[q]) || (pass && kc == kq && lane != j))) {   // equal keys (rare): the ids decide
          const uint64_t ic = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rid >> 32), j) << 32) |
                              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rid, j);
          c_lt_e = c_lt_e || (kc == bk[q] && ic < bi[q]);
          c_lt_c = c_lt_c || (kc == kq && ic < rid);
        }
        e_shift += (have && c_lt_e) ? 1u : 0u;
        const uint32_t below = (uint32_t)__builtin_popcountll(__ballot(have && !c_lt_e));   // (ids are unique: never equal)
        if (lane == j) c_below = below;
        c_rank += (pass && c_lt_c) ? 1u : 0u;
      }
      const uint32_t pe = (uint32_t)lane + e_shift, pc = c_rank + c_below;
      if (have && pe < 64u) { w_key[pe] = bk[q]; w_id[pe] = bi[q]; w_slot[pe] = bs[q]; }
      if (pass && pc < 64u) { w_key[pc] = kq; w_id[pc] = rid; w_slot[pc] = slot; }
      one_wave_sync();
      const uint32_t nn = bn[q] + (uint32_t)__builtin_popcountll(m);
      bn[q] = nn < 64u ? nn : 64u;
      if ((uint32_t)lane < bn[q]) { bk[q] = w_key[lane]; bi[q] = w_id[lane]; bs[q] = w_slot[lane]; }
      if (bn[q] >= k) { tk[q] = w_key[k - 1]; ti[q] = w_id[k - 1]; }
      one_wave_sync();
    }
  }
  // ---- block merge: the four waves' lists -> the block's k best, in order
#pragma unroll
  for (int q = 0; q < QB; q++) {
    if (q >= nq_grp) break;
    __syncthreads();
    const uint32_t mine = bn[q] < k ? bn[q] : k;
    if ((uint32_t)lane < k) {
      const uint32_t i = (uint32_t)wave * k + lane;
      const bool have = (uint32_t)lane < mine;
      m_key[i] = have ? bk[q] : 0xffffffffu; m_id[i] = have ? bi[q] : ~0ull; m_slot[i] = have ? bs[q] : 0xffffffffu;
    }
    if (lane == 0) s_wn[wave] = mine;
    __syncthreads();
    const uint32_t all = s_wn[0] + s_wn[1] + s_wn[2] + s_wn[3];
    // Records travel between blocks (and XCDs: one L2 each) as agent-scope atomic stores / loads — written through to the coherence
    // point, so no wave needs a release fence at agent scope (an L2 write-back per wave: 8192 of them cost more than the scan of a
    // million rows).
    OneRec* out = recs + ((size_t)q * gridDim.x + blockIdx.x) * k;   // [q][block][k]: one query's records are contiguous
    if ((uint32_t)tid < 4 * k && m_slot[tid] != 0xffffffffu) {
      const uint32_t ki = m_key[tid]; const uint64_t ii = m_id[tid];
      const uint32_t rank = sel_rank(m_key, m_id, 4 * k, ki, ii);
      if (rank < k) {
        __hip_atomic_store(out + rank, ((unsigned long long)ki << 32) | m_slot[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (rank == 0) atomicMin(&st->bucket[q * ONE_KMAX + blockIdx.x % k], ki);
      }
    }
    if ((uint32_t)tid < k && (uint32_t)tid >= all)   // fewer than k rows here
      __hip_atomic_store(out + tid, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- the last block to arrive finishes the search
  // Every wave waits for ITS OWN record stores and bucket atomics to be acknowledged before the barrier that precedes the ticket:
  // they are agent-scope (sc1, written through to the coherence point), so once vmcnt reaches 0 they are visible to any block on
  // any XCD that later observes the ticket.  (A workgroup-scope release fence is NOT enough: on gfx9 it waits for lgkmcnt only, and
  // the ticket could reach L2 before the stores of the same block — ADVICE r3.  An agent-scope release fence would add an L2
  // write-back per wave, which is what this design avoids.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (tid == 0) s_last = atomicAdd(&st->done, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const uint32_t per_q = gridDim.x * k;
  for (int q = 0; q < nq_grp; q++) {
    if (tid == 0) { s_n = 0; s_bmax = 0; }
    __syncthreads();
    if ((uint32_t)tid < k) atomicMax(&s_bmax, __hip_atomic_load(&st->bucket[q * ONE_KMAX + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __syncthreads();
    const uint32_t bound = s_bmax;   // an empty bucket (fewer than k blocks) is 0xffffffff: every record passes, and there are < k * k of them
    OneRec* src = recs + (size_t)q * per_q;
    constexpr int B = 16;   // records per thread in flight: the loop is all L2 / fabric latency
    for (uint32_t i0 = tid; i0 < per_q; i0 += 256 * B) {
      unsigned long long e[B];
#pragma unroll
      for (int u = 0; u < B; u++) {
        const uint32_t i = i0 + (uint32_t)u * 256u;
        e[u] = i < per_q ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ~0ull;
      }
#pragma unroll
      for (int u = 0; u < B; u++) {
        const uint32_t ek = (uint32_t)(e[u] >> 32), es = (uint32_t)e[u];
        if (es == 0xffffffffu || ek > bound) continue;
        const uint32_t j = atomicAdd(&s_n, 1u);
        if (j < cap) cand[(size_t)q * cap + j] = ((unsigned long long)(ek ^ flip) << 32) | es;
      }
    }
    __syncthreads();
    if (tid == 0) cnt[q] = s_n;
    __syncthreads();
    flat_select_block<SELECT_SMALL>(q, cand, cnt, thr, cap, k, nearest, ids, dense_base, ovf, out_ids, out_scores, out_counts);
    __syncthreads();
  }
  if (tid == 0) { st->done = 0; }
  st->bucket[tid] = 0xffffffffu;
}

// "f8" rows through the matrix cores.  The reference's Float8 decodes to eight values (float8.go:233-266; exact.hpp:
// f8bits_to_f32bits): 0, 2^-24, 2^-23, 1.5 * 2^-23, each possibly with bit 15 of the f32 pattern set (+2^-8, +2^-8, +2^-8 / 1.5 relative;
// a code whose low two bits are 0 decodes to 0 or to the denormal 2^-134).  Scaled by 2^24 every one of them is exactly a binary16 number
// (0, 1, 2, 3, 1 + 2^-8, 2 + 2^-7, 3 + 2^-7; the denormal becomes 0: a relative 2^-110 of any product) — so a DERIVED copy of the rows,
// rows16 = decode(code) * 2^24 as binary16, with norms16 = ||row||^2 * 2^48, feeds the unchanged 2-byte candidate GEMM: products
// exact, f32 accumulation, the cosine value scale-free — the same candidate margin as for binary16 codes — and the survivors are re-scored
// from the 1-byte rows in the reference's order.  Cosine only: the Euclidean candidate value is not scale-free.
__device__ __forceinline__ unsigned short f8_to_scaled_f16bits(uint32_t code) {
  const uint32_t m = code & 3u, sgn = (code >> 7) & 1u;
  if (m == 0) return 0;
  const uint32_t base = m == 1 ? 0x3C00u : (m == 2 ? 0x4000u : 0x4200u);   // 1.0, 2.0, 3.0
  // bit 15 of the f32 pattern = 2^-8 of the mantissa scale: 1 + 2^-8 (4 ulp of binary16 at 1), 2 * (1 + 2^-8), 2 * (1.5 + 2^-8)
  return (unsigned short)(base + (sgn ? 4u : 0u));
}
__global__ void f8_expand_kernel(const uint8_t* __restrict__ rows, size_t stride, const float* __restrict__ norms, const uint32_t* __restrict__ slots,
                                 uint64_t slot_base, uint64_t n, int dim, uint8_t* __restrict__ rows16, size_t stride16, float* __restrict__ norms16) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = (int)(stride16 / 2);   // binary16 elements per destination row incl. its zero padding
  if (t >= n * (uint64_t)per) return;
  const uint64_t i = t / per; const int e = (int)(t - i * per);
  const uint64_t slot = slots ? slots[i] : slot_base + i;
  reinterpret_cast<unsigned short*>(rows16 + slot * stride16)[e] = e < dim ? f8_to_scaled_f16bits(rows[slot * stride + e]) : (unsigned short)0;
  if (e == 0) norms16[slot] = norms[slot] * 281474976710656.0f;   // 2^48: an exponent shift
}

// stored codes of the edge .vertex stream (big-endian f32 / u16, raw u8) at arbitrary byte offsets -> rows
template <int QUANT>
__global__ void be_codes_kernel(const uint8_t* __restrict__ chunk, const uint64_t* __restrict__ offs, uint64_t m, int dim,
                                uint8_t* __restrict__ rows, size_t stride, uint64_t slot_base) {
  uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= m * (uint64_t)dim) return;
  uint64_t i = t / dim; int e = (int)(t - i * dim);
  constexpr int EB = QUANT == Q_NONE ? 4 : (QUANT == Q_F8 ? 1 : 2);
  const uint8_t* s = chunk + offs[i] + (size_t)e * EB;
  uint8_t* d = rows + (slot_base + i) * stride + (size_t)e * EB;
  if (EB == 4) { d[0] = s[3]; d[1] = s[2]; d[2] = s[1]; d[3] = s[0]; }
  else if (EB == 2) { d[0] = s[1]; d[1] = s[0]; }
  else d[0] = s[0];
  if (e == 0) for (size_t b = (size_t)dim * EB; b < stride; b++) rows[(slot_base + i) * stride + b] = 0;
}
struct FBER {
  const uint8_t* p; uint64_t n, i = 0; bool ok = true;
  bool need(uint64_t k) { if (i + k > n) { ok = false; return false; } return true; }
  uint8_t u8() { if (!need(1)) return 0; return p[i++]; }
  uint16_t u16() { if (!need(2)) return 0; uint16_t v = (uint16_t)((p[i] << 8) | p[i + 1]); i += 2; return v; }
  uint32_t u32() { if (!need(4)) return 0; uint32_t v = 0; for (int k = 0; k < 4; k++) v = (v << 8) | p[i + k]; i += 4; return v; }
  uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int k = 0; k < 8; k++) v = (v << 8) | p[i + k]; i += 8; return v; }
};
struct FBEW {
  uint8_t* p; uint64_t cap, n = 0;
  void put(const void* s, size_t k) { if (p && n + k <= cap) std::memcpy(p + n, s, k); n += k; }
  void u32(uint32_t v) { uint8_t b[4]; for (int i = 0; i < 4; i++) b[i] = (uint8_t)(v >> (8 * (3 - i))); put(b, 4); }
  void u64(uint64_t v) { uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * (7 - i))); put(b, 8); }
};

// ---------------------------------------------------------------------------------------------------
// Per-call search context (stream, events, workspaces): searches hold the store's lock shared and run concurrently.
struct FCtx {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  DevBuf w_qraw, w_qeff, w_qn, w_qn16, w_cand, w_cand2, w_q16, w_cnt, w_out_ids, w_out_sc, w_out_cnt, w_gather;
  DevBuf w_pack;           // small host-buffer calls: ids | scores | counts in one block (one D2H; see PinnedBuf)
  PinnedBuf h_in, h_out;
  DevBuf w_one;            // flat_one_kernel: OneState | per-block records (reserved once, at its maximum)
  bool one_ready = false;  // ... and its state words initialised (the kernel leaves them reset)
  int init() {  // the caller has selected the store's device
    COLTT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    COLTT_HIP(hipEventCreate(&ev0));
    COLTT_HIP(hipEventCreate(&ev1));
    return COLTT_OK;
  }
  ~FCtx() {
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

struct Flat : Object {
  uint32_t dim = 0; int metric = 0, quant = 0; size_t stride = 0;
  uint64_t n = 0, cap = 0;
  DevBuf rows, norms, ids;
  DevBuf rows16, norms16; size_t stride16 = 0; bool f8x = false;   // f8 + cosine: the binary16 copy the candidate GEMM reads (f8_expand_kernel)
  bool dense = true; uint64_t dense_base = 0;     // id = dense_base + slot, no map
  std::unordered_map<uint64_t, uint32_t> id2slot;  // !dense
  std::vector<uint64_t> h_ids;                     // !dense : slot -> id (host mirror)
  hipStream_t stream = nullptr;      // mutations (exclusive lock); searches use their context's stream
  std::atomic<float> last_ms{0.f};   // kernel time of the most recently finished search call
  CtxPool<FCtx> pool;
  DevBuf w_raw, w_slots;             // ingest staging
  DevBuf d_maxn; uint32_t norm_bits[2] = {0u, 0u};  // running bounds of ||row||^2 over everything ever stored: [0] max (bits), [1] max of the complemented bits = min
  float max_norm() const { return __builtin_bit_cast(float, norm_bits[0]); }   // scales the Euclidean matrix-core margin
  float min_norm() const { return __builtin_bit_cast(float, ~norm_bits[1]); }   // NaN bits while the store is empty
  std::atomic<uint64_t> mfma_groups{0}, mfma_fallbacks{0};  // groups served by the MFMA path / sent back to the exact path
  std::atomic<uint64_t> one_groups{0};                      // searches served by the one-launch kernel (<= 4 queries)
  ~Flat() override {
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamDestroy(stream);
  }
  int reserve(uint64_t rows_needed) {
    if (rows_needed <= cap) return COLTT_OK;
    uint64_t ncap = std::max<uint64_t>(rows_needed, cap + cap / 2);
    ncap = std::max<uint64_t>(ncap, 1024);
    // ROW_SLACK rows (and norms) behind the capacity: the matrix-core scan fetches whole 256/384-row tiles without clamping
    // the last one (flat_mfma.hpp); what it reads there is never scored
    // ... and everything behind the stored rows is ZERO: for dim % 32 != 0 the last K step of a row reads into its (zeroed)
    // padding and the head of the next row, against zero query columns — finite garbage is harmless there, NaN bits are not
    const size_t old_bytes = rows.cap;
    COLTT_TRY(rows.reserve((ncap + ROW_SLACK) * stride, true, stream));
    if (rows.cap > old_bytes) {
      COLTT_HIP(hipMemsetAsync(rows.as<uint8_t>() + old_bytes, 0, rows.cap - old_bytes, stream));
      COLTT_HIP(hipStreamSynchronize(stream));
    }
    COLTT_TRY(norms.reserve((ncap + 2 * ROW_SLACK) * 4, true, stream));
    if (f8x) {
      const size_t old16 = rows16.cap;
      COLTT_TRY(rows16.reserve((ncap + ROW_SLACK) * stride16, true, stream));
      if (rows16.cap > old16) {
        COLTT_HIP(hipMemsetAsync(rows16.as<uint8_t>() + old16, 0, rows16.cap - old16, stream));
        COLTT_HIP(hipStreamSynchronize(stream));
      }
      COLTT_TRY(norms16.reserve((ncap + 2 * ROW_SLACK) * 4, true, stream));
    }
    if (!dense) COLTT_TRY(ids.reserve(ncap * 8, true, stream));
    cap = ncap;
    return COLTT_OK;
  }
  int undense() {  // switch from "id = base + slot" to an explicit id table
    if (!dense) return COLTT_OK;
    h_ids.resize(n);
    id2slot.reserve(n * 2);
    for (uint64_t s = 0; s < n; s++) { h_ids[s] = dense_base + s; id2slot[dense_base + s] = (uint32_t)s; }
    dense = false;
    COLTT_TRY(ids.reserve(std::max<uint64_t>(cap, 1024) * 8, false, stream));
    if (n) COLTT_HIP(hipMemcpyAsync(ids.p, h_ids.data(), n * 8, hipMemcpyHostToDevice, stream));
    COLTT_HIP(hipStreamSynchronize(stream));
    return COLTT_OK;
  }
};

template <int QUANT>
int launch_prep_rows(Flat* f, const float* d_raw, uint64_t n, const uint32_t* d_slots, uint64_t slot_base) {
  if (n == 0) return COLTT_OK;
  dev::launch_prep_rows<QUANT>(f->stream, d_raw, n, (int)f->dim, f->metric == COLTT_COSINE, d_slots, slot_base,
                               f->rows.as<uint8_t>(), f->stride);
  row_norms_kernel<QUANT><<<ceil_div(n * 2, 256), 256, 0, f->stream>>>(f->rows.as<uint8_t>(), f->stride, d_slots,
                                                                       slot_base, n, (int)f->dim, f->norms.as<float>(), f->d_maxn.as<uint32_t>());
  COLTT_HIP(hipMemcpyAsync(f->norm_bits, f->d_maxn.p, 8, hipMemcpyDeviceToHost, f->stream));  // complete at the caller's stream sync
  if constexpr (QUANT == Q_F8) {
    if (f->f8x) f8_expand_kernel<<<ceil_div(n * (f->stride16 / 2), 256), 256, 0, f->stream>>>(f->rows.as<uint8_t>(), f->stride, f->norms.as<float>(), d_slots, slot_base, n,
                                                                                       (int)f->dim, f->rows16.as<uint8_t>(), f->stride16, f->norms16.as<float>());
  }
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}
int prep_rows(Flat* f, const float* d_raw, uint64_t n, const uint32_t* d_slots, uint64_t slot_base) {
#define COLTT_PR(Q) return launch_prep_rows<Q>(f, d_raw, n, d_slots, slot_base)
  COLTT_DISPATCH_QUANT(f->quant, COLTT_PR)
#undef COLTT_PR
  return COLTT_OK;
}

inline int scan_qb(const Flat* f) { return f->dim <= 1024 ? QB : 4; }

template <int METRIC, int QUANT, bool GATHER, int QBT>
void launch_scan_q(Flat* f, FCtx* c, const uint32_t* gather, uint64_t begin, uint64_t end, const float* q_eff, const float* qn,
                   int nq_grp, const uint32_t* thr, int nearest, unsigned long long* cand, uint32_t* cnt, uint32_t cap) {
  uint64_t groups = (end - begin + 31) / 32;
  uint32_t grid = (uint32_t)std::min<uint64_t>((groups + 3) / 4, 256 * 8);
  size_t lds = (size_t)QBT * ((f->dim + 3) & ~3u) * 4;
  auto kern = flat_scan_kernel<METRIC, QUANT, GATHER, QBT>;
  if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  kern<<<grid, 256, lds, c->stream>>>(
      f->rows.as<uint8_t>(), f->stride, f->norms.as<float>(), gather, begin, end, q_eff, qn, nq_grp, (int)f->dim, thr,
      nearest, cand, cnt, cap);
}
template <int METRIC, int QUANT, bool GATHER>
void launch_scan(Flat* f, FCtx* c, const uint32_t* gather, uint64_t begin, uint64_t end, const float* q_eff, const float* qn,
                 int nq_grp, const uint32_t* thr, int nearest, unsigned long long* cand, uint32_t* cnt, uint32_t cap) {
  // the kernel evaluates QBT query slots for every row whether they are filled or not: a single-query call (the reference's RPC
  // shape, and every filtered search) takes the 1-slot instance, up to four queries the 4-slot one
  if (nq_grp <= 1) launch_scan_q<METRIC, QUANT, GATHER, 1>(f, c, gather, begin, end, q_eff, qn, nq_grp, thr, nearest, cand, cnt, cap);
  else if (nq_grp <= 4 || scan_qb(f) != QB) launch_scan_q<METRIC, QUANT, GATHER, 4>(f, c, gather, begin, end, q_eff, qn, nq_grp, thr, nearest, cand, cnt, cap);
  else launch_scan_q<METRIC, QUANT, GATHER, QB>(f, c, gather, begin, end, q_eff, qn, nq_grp, thr, nearest, cand, cnt, cap);
}
template <bool GATHER>
int scan_dispatch(Flat* f, FCtx* c, const uint32_t* gather, uint64_t begin, uint64_t end, const float* q_eff, const float* qn,
                   int nq_grp, const uint32_t* thr, int nearest, unsigned long long* cand, uint32_t* cnt, uint32_t cap) {
#define COLTT_SCAN_ARGS f, c, gather, begin, end, q_eff, qn, nq_grp, thr, nearest, cand, cnt, cap
#define COLTT_SCAN(Q) do { if (f->metric == COLTT_COSINE) launch_scan<M_COS, Q, GATHER>(COLTT_SCAN_ARGS); else launch_scan<M_L2, Q, GATHER>(COLTT_SCAN_ARGS); } while (0)
  COLTT_DISPATCH_QUANT(f->quant, COLTT_SCAN)
#undef COLTT_SCAN
#undef COLTT_SCAN_ARGS
  return COLTT_OK;
}

int prep_queries(Flat* f, FCtx* c, const float* d_qraw, size_t nq) {
  COLTT_TRY(c->w_qeff.reserve(nq * f->dim * 4));
  COLTT_TRY(c->w_qn.reserve(nq * 4));
  int norm = f->metric == COLTT_COSINE;
#define COLTT_PQ(Q) launch_prep_queries<Q>(c->stream, d_qraw, nq, (int)f->dim, norm, c->w_qeff.as<float>(), c->w_qn.as<float>())
  COLTT_DISPATCH_QUANT(f->quant, COLTT_PQ)
#undef COLTT_PQ
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// One group of <= QB prepared queries [q0, q0+g) over positions [0, total) (rows, or entries of the gather list), exact order.
int search_group_exact(Flat* f, FCtx* c, size_t q0, int g, uint32_t k, int nearest, const uint32_t* d_gather, uint64_t total,
                       uint64_t* d_out_ids, float* d_out_sc, uint32_t* d_out_cnt, uint32_t cap) {
  uint32_t* cnt = c->w_cnt.as<uint32_t>();
  uint32_t* thr = cnt + 256;
  uint32_t* ovf = cnt + 512;
  unsigned long long* cand = c->w_cand.as<unsigned long long>();
  const uint64_t* ids = f->dense ? nullptr : f->ids.as<uint64_t>();
  const float* qe = c->w_qeff.as<float>() + q0 * f->dim;
  const float* qn = c->w_qn.as<float>() + q0;
  uint64_t* oi = d_out_ids + q0 * k; float* os = d_out_sc + q0 * k; uint32_t* oc = d_out_cnt + q0;
  auto scan = [&](uint64_t b, uint64_t e) -> int {
    if (d_gather) COLTT_TRY(scan_dispatch<true>(f, c, d_gather, b, e, qe, qn, g, thr, nearest, cand, cnt, cap));
    else COLTT_TRY(scan_dispatch<false>(f, c, nullptr, b, e, qe, qn, g, thr, nearest, cand, cnt, cap));
    flat_select_kernel<<<g, 256, 0, c->stream>>>(cand, cnt, thr, cap, k, nearest, ids, f->dense_base, ovf, oi, os, oc);
    return COLTT_OK;
  };
  init_group_kernel<<<1, 256, 0, c->stream>>>(cnt, thr, ovf, nearest);
  if (total == 0) { flat_select_kernel<<<g, 256, 0, c->stream>>>(cand, cnt, thr, cap, k, nearest, ids, f->dense_base, ovf, oi, os, oc); return COLTT_OK; }
  // optimistic: a small unfiltered first segment, then segments 32x what has been seen, each behind the threshold picked from
  // everything before it — an element passes with probability ~k / seen, so every candidate list stays short (<= 512: the select's
  // rank-sort path, no radix passes).  One unfiltered segment of `cap` rows followed by the rest cost a 65 536-candidate radix
  // select per query group: 0.32 ms for ONE query over 100 k x 128 rows, most of it selection.
  uint64_t s0 = std::min<uint64_t>({total, (uint64_t)cap, std::max<uint64_t>(512, 4ull * k)});
  for (uint64_t b = 0, e = s0; b < total; b = e, e = std::min<uint64_t>(total, e * 32)) COLTT_TRY(scan(b, e));
  uint32_t h_ovf = 0;
  COLTT_HIP(hipMemcpyAsync(&h_ovf, ovf, 4, hipMemcpyDeviceToHost, c->stream));
  COLTT_HIP(hipStreamSynchronize(c->stream));
  if (h_ovf) {  // adversarial order: redo with segments that cannot overflow (list holds <= k + segment)
    init_group_kernel<<<1, 256, 0, c->stream>>>(cnt, thr, ovf, nearest);
    uint64_t seg = cap - std::min<uint32_t>(k, cap / 2);
    for (uint64_t b = 0; b < total; b += seg) COLTT_TRY(scan(b, std::min<uint64_t>(total, b + seg)));
  }
  return COLTT_OK;
}

// <= 4 prepared queries [q0, q0+g) over positions [0, total), k <= 64: one launch (flat_one_kernel).
// COLTT_FLAT_ONE=0 sends small batches through the scan + select chain instead (measurement and test knob).
bool flat_one_enabled() { return policy().flat_one; }

template <int METRIC, int QUANT, bool GATHER, int QBT>
int launch_one(Flat* f, FCtx* c, const uint32_t* gather, uint64_t total, const float* qe, const float* qn, int g, uint32_t k, int nearest,
               uint64_t* oi, float* os, uint32_t* oc, uint32_t cap) {
  const uint64_t groups = (total + 31) / 32;
  // two 32-row groups per wave where the collection allows: 8 groups per block; at most 8 blocks per CU
  const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((groups + 7) / 8, ONE_MAX_BLOCKS));
  const size_t lds = (size_t)QBT * ((f->dim + 3) & ~3u) * 4;
  auto kern = flat_one_kernel<METRIC, QUANT, GATHER, QBT>;
  if (lds > 40 * 1024) COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  uint8_t* base = c->w_one.as<uint8_t>();
  OneState* st = reinterpret_cast<OneState*>(base);
  OneRec* recs = reinterpret_cast<OneRec*>(base + ONE_STATE_BYTES);
  uint32_t* cnt = c->w_cnt.as<uint32_t>();
  kern<<<grid, 256, lds, c->stream>>>(f->rows.as<uint8_t>(), f->stride, f->norms.as<float>(), gather, total, qe, qn, g, (int)f->dim, k, nearest,
                                      f->dense ? nullptr : f->ids.as<uint64_t>(), f->dense_base, recs, st,
                                      c->w_cand.as<unsigned long long>(), cnt, cnt + 256, cap, cnt + 512, oi, os, oc);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

int search_group_one(Flat* f, FCtx* c, size_t q0, int g, uint32_t k, int nearest, const uint32_t* d_gather, uint64_t total,
                     uint64_t* d_out_ids, float* d_out_sc, uint32_t* d_out_cnt, uint32_t cap) {
  if (!c->one_ready) {
    COLTT_TRY(c->w_one.reserve(ONE_STATE_BYTES + (size_t)ONE_MAX_BLOCKS * ONE_QMAX * ONE_KMAX * sizeof(OneRec)));
    COLTT_HIP(hipMemsetAsync(c->w_one.p, 0xff, ONE_STATE_BYTES, c->stream));   // bucket[] = "no bound yet"
    COLTT_HIP(hipMemsetAsync(c->w_one.p, 0, 16, c->stream));       // done = 0
    c->one_ready = true;
  }
  const float* qe = c->w_qeff.as<float>() + q0 * f->dim;
  const float* qn = c->w_qn.as<float>() + q0;
  uint64_t* oi = d_out_ids + q0 * k; float* os = d_out_sc + q0 * k; uint32_t* oc = d_out_cnt + q0;
#define COLTT_ONE_ARGS f, c, d_gather, total, qe, qn, g, k, nearest, oi, os, oc, cap
#define COLTT_ONE_Q(M, Q) (d_gather ? (g <= 1 ? launch_one<M, Q, true, 1>(COLTT_ONE_ARGS) : launch_one<M, Q, true, ONE_QMAX>(COLTT_ONE_ARGS)) \
                                    : (g <= 1 ? launch_one<M, Q, false, 1>(COLTT_ONE_ARGS) : launch_one<M, Q, false, ONE_QMAX>(COLTT_ONE_ARGS)))
  int rc = COLTT_OK;
#define COLTT_ONE(Q) rc = f->metric == COLTT_COSINE ? COLTT_ONE_Q(M_COS, Q) : COLTT_ONE_Q(M_L2, Q)
  COLTT_DISPATCH_QUANT(f->quant, COLTT_ONE)
#undef COLTT_ONE
#undef COLTT_ONE_Q
#undef COLTT_ONE_ARGS
  return rc;
}

// The FLAT matrix-core kernel is flat_mfma.hpp (split LDS-DMA rings) — the third generation; the superseded ones (rounds 1-2, and the round-3
// experiment that lost) were A/B material until round 5 and live in the history only.

// what the candidate GEMM streams: the stored rows, or — "f8" stores — their derived binary16 copy (f8_expand_kernel)
struct RowSrc { const uint8_t* rows; size_t stride; const float* norms; };
inline RowSrc mfma_rows(const Flat* f) {
  if (f->quant == COLTT_Q_F8) return RowSrc{f->rows16.as<uint8_t>(), f->stride16, f->norms16.as<float>()};
  return RowSrc{f->rows.as<uint8_t>(), f->stride, f->norms.as<float>()};
}

template <int BN, bool AF32>
int launch_mfma_scan_t(Flat* f, FCtx* c, uint64_t b, uint64_t e, const _Float16* q16, const float* qn, int g, const uint32_t* thr, int nearest,
                       unsigned long long* cand, uint32_t* cnt, uint32_t cap, bool seed, int kdim, const uint32_t* d_gather) {
  const RowSrc src = mfma_rows(f);
  if (d_gather) {   // rows[gather[pos]] for pos in [b, e): FilterableVertexSearch through the matrix cores (flat_mfma.hpp, GATHER)
    auto kern = f->metric == COLTT_COSINE ? (seed ? flat_mfma3_kernel<BN, AF32, true, M2_BM, M_COS, true> : flat_mfma3_kernel<BN, AF32, false, M2_BM, M_COS, true>)
                                          : (seed ? flat_mfma3_kernel<BN, AF32, true, M2_BM, M_L2, true> : flat_mfma3_kernel<BN, AF32, false, M2_BM, M_L2, true>);
    const size_t lds = M3Geom<BN, AF32, M2_BM, true>::LDS;
    COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t tiles = (e - b + M2_BM - 1) / M2_BM;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(tiles, 256);
    kern<<<grid, M2_NT, lds, c->stream>>>(src.rows, src.stride, src.norms, b, e, q16, qn, g, kdim, thr,
                                          nearest, cand, cnt, cap, d_gather);
    COLTT_HIP(hipGetLastError());
    return COLTT_OK;
  }
  {
#ifdef COLTT_M3_BM
    constexpr int BM = (!AF32 && BN == 256) ? COLTT_M3_BM : M2_BM;
#else
    constexpr int BM = M2_BM;
#endif
    auto kern = f->metric == COLTT_COSINE ? (seed ? flat_mfma3_kernel<BN, AF32, true, BM, M_COS> : flat_mfma3_kernel<BN, AF32, false, BM, M_COS>)
                                          : (seed ? flat_mfma3_kernel<BN, AF32, true, BM, M_L2> : flat_mfma3_kernel<BN, AF32, false, BM, M_L2>);
    const size_t lds = M3Geom<BN, AF32, BM>::LDS;
    COLTT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint64_t tiles = (e - b + BM - 1) / BM;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(tiles, 256);
    kern<<<grid, M2_NT, lds, c->stream>>>(src.rows, src.stride, src.norms, b, e, q16, qn, g, kdim, thr,
                                          nearest, cand, cnt, cap, nullptr);
    COLTT_HIP(hipGetLastError());
    return COLTT_OK;
  }
}

template <int BN>
int launch_mfma_scan(Flat* f, FCtx* c, uint64_t b, uint64_t e, const _Float16* q16, const float* qn, int g, const uint32_t* thr, int nearest,
                     unsigned long long* cand, uint32_t* cnt, uint32_t cap, bool seed, int kdim, const uint32_t* d_gather) {
  if (f->quant == COLTT_Q_NONE) return launch_mfma_scan_t<BN, true>(f, c, b, e, q16, qn, g, thr, nearest, cand, cnt, cap, seed, kdim, d_gather);
  return launch_mfma_scan_t<BN, false>(f, c, b, e, q16, qn, g, thr, nearest, cand, cnt, cap, seed, kdim, d_gather);
}

// relative error bound of the Euclidean candidate value s~^2 against the exact s^2, in units of (||q||^2 + ||r||^2): D * 2^-24 from
// the f32 accumulation of exact f16 products, 2^-10 more when f32 rows (and queries) are rounded to binary16 on their way into
// LDS, and 3 * D * 2^-24 for the f32 rounding inside the two norms and the reference's own difference-form sum
float l2_eps(const Flat* f) {
  const float acc = (float)f->dim * 5.9604645e-8f;
  return 4.0f * acc + (f->quant == COLTT_Q_NONE ? 9.8e-4f : 0.f);
}

// K of the candidate GEMM: dim padded to whole 32-column steps, and to at least FOUR steps — the kernel's raw-norm parity buffers
// assume >= 4 K steps per tile.  Short rows (dim < 128: 64- and 96-d collections) therefore run with K = 128: a row's K range ends in
// the rows stored behind it (finite bits, multiplied by zero query columns), exactly like the overhang of a dim % 32 != 0 row.
inline int mfma_kdim(const Flat* f) { return std::max<int>(4 * MF_BK, (int)((f->dim + MF_BK - 1) / MF_BK * MF_BK)); }

// One group of <= 256 prepared queries through the matrix cores (cosine, 2-byte codes, dim % 64 == 0), then exact re-score.
int search_group_mfma(Flat* f, FCtx* c, size_t q0, int g, uint32_t k, int nearest, const uint32_t* d_gather, uint64_t total, uint64_t* d_out_ids, float* d_out_sc,
                      uint32_t* d_out_cnt, uint32_t cap, uint32_t* ovf) {
  uint32_t* cnt = c->w_cnt.as<uint32_t>();
  uint32_t* thr = cnt + 256;
  unsigned long long* cur = c->w_cand.as<unsigned long long>();
  unsigned long long* oth = c->w_cand2.as<unsigned long long>();
  const uint64_t* ids = f->dense ? nullptr : f->ids.as<uint64_t>();
  const float* qe = c->w_qeff.as<float>() + q0 * f->dim;
  const float* qn = c->w_qn.as<float>() + q0;
  _Float16* q16 = c->w_q16.as<_Float16>();
  const int BN = g <= 64 ? 64 : (g <= 128 ? 128 : 256);
  const int dimp = mfma_kdim(f);   // K padded to whole steps (and to at least four of them) with zero query columns
  // "f8" stores: the query (already lowered + decoded: the same eight values) goes in scaled by 2^24, its norm by 2^48 (see f8_expand_kernel)
  const bool f8 = f->quant == COLTT_Q_F8;
  if (f8) COLTT_TRY(c->w_qn16.reserve(256 * 4));
  mfma_prep_queries_kernel<<<ceil_div((uint64_t)BN * dimp, 256), 256, 0, c->stream>>>(qe, g, BN, (int)f->dim, dimp, q16, cnt, thr, ovf, nearest,
                                                                                      f8 ? 16777216.0f : 1.0f, f8 ? qn : nullptr, f8 ? c->w_qn16.as<float>() : nullptr);
  const float* qn_exact = qn;                 // the exact re-score keeps the unscaled norms
  if (f8) qn = c->w_qn16.as<float>();
  auto scan = [&](uint64_t b, uint64_t e) -> int {
    const bool seed = b == 0 && e - b <= cap;
    if (BN == 64) COLTT_TRY(launch_mfma_scan<64>(f, c, b, e, q16, qn, g, thr, nearest, cur, cnt, cap, seed, dimp, d_gather));
    else if (BN == 128) COLTT_TRY(launch_mfma_scan<128>(f, c, b, e, q16, qn, g, thr, nearest, cur, cnt, cap, seed, dimp, d_gather));
    else COLTT_TRY(launch_mfma_scan<256>(f, c, b, e, q16, qn, g, thr, nearest, cur, cnt, cap, seed, dimp, d_gather));
    if (f->metric == COLTT_COSINE)
      flat_pick_kernel<<<g, 256, 0, c->stream>>>(cur, oth, cnt, thr, cap, k, nearest, f->quant == COLTT_Q_NONE ? MF_MARGIN_F32 : MF_MARGIN, ovf);
    else  // Euclidean: |s~^2 - s^2| <= eps * (||q||^2 + ||r||^2), eps = dot error (+ f16 rounding of f32 rows) + f32 rounding of the norms / the exact sum
      flat_pick_kernel<<<g, 256, 0, c->stream>>>(cur, oth, cnt, thr, cap, k, nearest, 2.0f * l2_eps(f), ovf, qn, f->max_norm());
    std::swap(cur, oth);
    return COLTT_OK;
  };
  // First segment unfiltered (it seeds the threshold); every later segment runs behind the bound picked from all rows
  // before it and is 8x what has been seen, so an element passes with probability ~k/seen: ~7k survivors per segment and
  // query, and the epilogue's element path (mf_emit_block) is practically never taken.  (One unfiltered seed followed by
  // ONE big segment left p = k/8192 for the whole scan: 72 % of the wave-blocks took the element path, 12k survivors/query.)
  // (the seed is small: all of its s0 x g scores are appended through atomics — 8192 rows x 256 queries took 0.7 ms)
  static const uint64_t seed_rows = [] { const char* e = getenv("COLTT_MFMA_SEED"); long v = e && *e ? atol(e) : 1024; return (uint64_t)(v < 256 ? 256 : v); }();
  // (batches up to 64: a 4 Ki-row seed — 1 M x 768 f32 over 100 k gathered rows, batch 16: 0.201 -> 0.188 ms, batch 64 0.231 -> 0.217; all
  //  1 M rows 0.758 -> 0.738 / 0.809 -> 0.785; above, the seed's g x s0 appended scores cost more than the tighter threshold gains)
  static const bool seed_set = [] { const char* e = getenv("COLTT_MFMA_SEED"); return e && *e; }();
  const uint64_t seed_now = seed_set ? seed_rows : (g <= 64 ? 4096 : 1024);
  uint64_t s0 = std::min<uint64_t>({total, (uint64_t)cap, std::max<uint64_t>(seed_now, 16ull * k)});
  static const uint64_t grow = [] { const char* e = getenv("COLTT_MFMA_GROW"); long v = e && *e ? atol(e) : 16; return (uint64_t)(v < 2 ? 2 : v); }();
  for (uint64_t b = 0, e = s0; b < total; b = e, e = std::min<uint64_t>(total, e * grow)) COLTT_TRY(scan(b, e));
  // exact re-score of the survivors + the ordinary select, queued behind the scans with no host round trip in between; the
  // overflow flag is read once at the end (a group whose candidate list overflowed is re-run in exact mode by the caller)
  {
    dim3 grid(16, g);
#define COLTT_RS(M, Q) flat_rescore_kernel<M, Q><<<grid, 64, 0, c->stream>>>(f->rows.as<uint8_t>(), f->stride, f->norms.as<float>(), qe, qn_exact, (int)f->dim, cur, cnt, cap)
    if (f->metric == COLTT_COSINE) { if (f->quant == COLTT_Q_NONE) COLTT_RS(M_COS, Q_NONE); else if (f8) COLTT_RS(M_COS, Q_F8); else COLTT_RS(M_COS, Q_F16); }
    else { if (f->quant == COLTT_Q_NONE) COLTT_RS(M_L2, Q_NONE); else COLTT_RS(M_L2, Q_F16); }
#undef COLTT_RS
  }
  flat_select_kernel<<<g, 256, 0, c->stream>>>(cur, cnt, thr, cap, k, nearest, ids, f->dense_base, ovf, d_out_ids + q0 * k, d_out_sc + q0 * k, d_out_cnt + q0);
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

// Search nq prepared queries over positions [0, total) (rows, or entries of the gather list).
int search_prepared(Flat* f, FCtx* c, size_t nq, uint32_t k, int select, int mode, const uint32_t* d_gather, uint64_t total,
                    uint64_t* d_out_ids, float* d_out_sc, uint32_t* d_out_cnt) {
  const int nearest = select == COLTT_SELECT_NEAREST;
  const uint32_t cap = std::max<uint32_t>(65536u, 8u * k);
  // matrix-core candidates: cosine with every kernel generation; Euclidean (s~^2 = ||q||^2 + ||r||^2 - 2 dot, exact re-score) with
  // the third-generation kernel, as long as every stored norm is finite (max_norm bounds the margin)
  // (f32 rows are rounded to binary16 for candidate generation: ||row||^2 <= 4e9 keeps every element inside its range)
  const float max_norm = f->max_norm(), min_norm = f->min_norm();
  const bool l2_ok = f->metric == COLTT_EUCLIDEAN && max_norm == max_norm &&
                     max_norm <= (f->quant == COLTT_Q_NONE ? 4.0e9f : 3.0e38f);
  // Cosine: the candidate margin (flat_mfma.hpp: MF_MARGIN / MF_MARGIN_F32) is proved for rows of norm ~1 — f32 rows are rounded to
  // binary16 on their way into the fragments, which is a RELATIVE perturbation only while the elements stay in binary16's normal
  // range.  Upserts normalise (none_vectorstore.go:63-65), but a loaded stream is stored as it is (vectorstore load paths do not
  // re-normalise), so a store that ever held a row with ||row||^2 outside [1/4, 4] answers through the exact scan.
  // ("f8" rows: their binary16 copy holds exact values and the candidate value is scale-free — any finite, non-zero-range store qualifies)
  const bool cos_ok = f->metric == COLTT_COSINE && (f->quant == COLTT_Q_F8 ? (f->f8x && max_norm == max_norm && max_norm < 3.0e38f) : (min_norm >= 0.25f && max_norm <= 4.0f));
  // K need not be a multiple of the 32-column step with the DMA kernels: the query tile is zero-padded and the rows' overhang
  // (padding, head of the next row — all finite as long as no stored norm ever was non-finite; Flat::reserve zeroes the rest)
  // multiplies zeros.  dim >= 128: the raw-norm parity buffers assume >= 4 K steps per tile.
  const bool finite_rows = max_norm == max_norm && max_norm < 3.0e38f;
  const bool k_ok = (f->dim % MF_BK == 0 && f->dim >= 4 * MF_BK) || (finite_rows);
  // (a filtered search — d_gather: positions of a slot list — takes the matrix cores too, through the kernel's gather mode; the
  //  superseded experiment generations have none)
  const bool mfma = mode == COLTT_MODE_MFMA && (cos_ok || l2_ok) &&
                    (f->quant == COLTT_Q_NONE || f->quant == COLTT_Q_F16 || f->quant == COLTT_Q_BF16 || (f->quant == COLTT_Q_F8 && f->f8x && f->metric == COLTT_COSINE)) &&
                    k_ok && f->dim >= 8 && f->dim <= 4096 && total > 0;
  // small batches: the whole search in one launch, whatever the mode asked for (exact-order scores either way)
  // (2-4 queries: while the scan is short — its four-query tile streams at about half the one-query rate, and past ~400 MB the chain's
  //  launch gaps no longer matter: 1 M x 128 f32 x 4 queries 286 us here, 212 us through the chain; 100 k x 768: 123 vs 185)
  const bool one = k <= ONE_KMAX && total > 0 && flat_one_enabled() &&
                   (nq == 1 || (nq <= (size_t)ONE_QMAX && total * (uint64_t)f->stride <= (400ull << 20)));
  if (one) {
    // the last block may have to keep EVERY block record (mass ties: a zero cosine query, a store of duplicates — all keys equal the
    // bound): the list is sized for grid x k records per query, so it cannot overflow and the (score, id) winners are exact
    const uint32_t cap1 = std::max<uint32_t>(cap, ONE_MAX_BLOCKS * k);
    COLTT_TRY(c->w_cand.reserve(std::max<size_t>((size_t)QB * cap, (size_t)ONE_QMAX * cap1) * 8));
    COLTT_TRY(c->w_cnt.reserve(4096 + 4));
    COLTT_HIP(hipEventRecord(c->ev0, c->stream));
    COLTT_TRY(search_group_one(f, c, 0, (int)nq, k, nearest, d_gather, total, d_out_ids, d_out_sc, d_out_cnt, cap1));
    COLTT_HIP(hipEventRecord(c->ev1, c->stream));
    f->one_groups.fetch_add(1);
    return COLTT_OK;
  }
  const size_t gq = mfma ? 256 : (size_t)scan_qb(f);
  COLTT_TRY(c->w_cand.reserve((size_t)std::max<size_t>(gq, QB) * cap * 8));
  if (mfma) { COLTT_TRY(c->w_cand2.reserve((size_t)gq * cap * 8)); COLTT_TRY(c->w_q16.reserve((size_t)256 * mfma_kdim(f) * 2)); }
  const size_t n_groups = (nq + gq - 1) / gq;
  COLTT_TRY(c->w_cnt.reserve(4096 + n_groups * 4));
  uint32_t* d_ovf = c->w_cnt.as<uint32_t>() + 1024;   // one overflow flag per matrix-core group, checked once after the last group
  COLTT_HIP(hipEventRecord(c->ev0, c->stream));
  if (mfma) {
    for (size_t q0 = 0, gi = 0; q0 < nq; q0 += gq, gi++) {
      int g = (int)std::min<size_t>(gq, nq - q0);
      COLTT_TRY(search_group_mfma(f, c, q0, g, k, nearest, d_gather, total, d_out_ids, d_out_sc, d_out_cnt, cap, d_ovf + gi));
      f->mfma_groups.fetch_add(1);
    }
    COLTT_HIP(hipEventRecord(c->ev1, c->stream));
    std::vector<uint32_t> h_ovf(n_groups);
    COLTT_HIP(hipMemcpyAsync(h_ovf.data(), d_ovf, n_groups * 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipStreamSynchronize(c->stream));
    bool any = false;
    for (size_t gi = 0; gi < n_groups; gi++) {
      if (!h_ovf[gi]) continue;  // candidate list overflowed (adversarial data): that group is re-run in exact mode
      any = true;
      f->mfma_fallbacks.fetch_add(1);
      const size_t q0 = gi * gq; const int g = (int)std::min<size_t>(gq, nq - q0);
      for (size_t s = 0; s < (size_t)g; s += scan_qb(f))
        COLTT_TRY(search_group_exact(f, c, q0 + s, (int)std::min<size_t>(scan_qb(f), g - s), k, nearest, d_gather, total, d_out_ids, d_out_sc, d_out_cnt, cap));
    }
    if (any) COLTT_HIP(hipEventRecord(c->ev1, c->stream));
    COLTT_HIP(hipGetLastError());
    return COLTT_OK;
  }
  for (size_t q0 = 0; q0 < nq; q0 += gq) {
    int g = (int)std::min<size_t>(gq, nq - q0);
    COLTT_TRY(search_group_exact(f, c, q0, g, k, nearest, d_gather, total, d_out_ids, d_out_sc, d_out_cnt, cap));
  }
  COLTT_HIP(hipEventRecord(c->ev1, c->stream));
  COLTT_HIP(hipGetLastError());
  return COLTT_OK;
}

int flat_search_common(Flat* f, FCtx* c, const float* queries, bool q_on_device, size_t nq, uint32_t k, int select, int mode,
                       const uint32_t* d_gather, uint64_t total, uint64_t* out_ids, float* out_scores,
                       uint32_t* out_counts, bool out_on_device) {
  if (k == 0 || k > K_MAX) return fail(COLTT_E_UNSUPPORTED, "flat search: k=%u outside [1,%u]", k, K_MAX);
  if (select != COLTT_SELECT_REFERENCE && select != COLTT_SELECT_NEAREST) return fail(COLTT_E_INVALID, "flat search: bad select %d", select);
  if (mode != COLTT_MODE_EXACT && mode != COLTT_MODE_MFMA) return fail(COLTT_E_INVALID, "flat search: bad mode %d", mode);
  if (nq == 0) return COLTT_OK;
  const float* d_q = queries;
  const size_t pack_bytes = nq * k * 12 + nq * 4;
  const bool small_q = !q_on_device && nq * f->dim * 4 <= SMALL_CALL_BYTES && small_call_staging();
  const bool packed = !out_on_device && pack_bytes <= SMALL_CALL_BYTES && small_call_staging();
  if (!q_on_device) {
    COLTT_TRY(c->w_qraw.reserve(nq * f->dim * 4));
    const void* src = queries;
    if (small_q) { COLTT_TRY(c->h_in.reserve(SMALL_CALL_BYTES)); std::memcpy(c->h_in.p, queries, nq * f->dim * 4); src = c->h_in.p; }
    COLTT_HIP(hipMemcpyAsync(c->w_qraw.p, src, nq * f->dim * 4, hipMemcpyHostToDevice, c->stream));
    d_q = c->w_qraw.as<float>();
  }
  COLTT_TRY(prep_queries(f, c, d_q, nq));
  uint64_t* d_oi = out_ids; float* d_os = out_scores; uint32_t* d_oc = out_counts;
  if (packed) {
    COLTT_TRY(c->w_pack.reserve(SMALL_CALL_BYTES));
    COLTT_TRY(c->h_out.reserve(SMALL_CALL_BYTES));
    uint8_t* b = c->w_pack.as<uint8_t>();
    d_oi = reinterpret_cast<uint64_t*>(b); d_os = reinterpret_cast<float*>(b + nq * k * 8); d_oc = reinterpret_cast<uint32_t*>(b + nq * k * 12);
  } else if (!out_on_device) {
    COLTT_TRY(c->w_out_ids.reserve(nq * k * 8));
    COLTT_TRY(c->w_out_sc.reserve(nq * k * 4));
    COLTT_TRY(c->w_out_cnt.reserve(nq * 4));
    d_oi = c->w_out_ids.as<uint64_t>(); d_os = c->w_out_sc.as<float>(); d_oc = c->w_out_cnt.as<uint32_t>();
  }
  COLTT_TRY(search_prepared(f, c, nq, k, select, mode, d_gather, total, d_oi, d_os, d_oc));
  if (packed) COLTT_HIP(hipMemcpyAsync(c->h_out.p, c->w_pack.p, pack_bytes, hipMemcpyDeviceToHost, c->stream));
  else if (!out_on_device) {
    COLTT_HIP(hipMemcpyAsync(out_ids, d_oi, nq * k * 8, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipMemcpyAsync(out_scores, d_os, nq * k * 4, hipMemcpyDeviceToHost, c->stream));
    COLTT_HIP(hipMemcpyAsync(out_counts, d_oc, nq * 4, hipMemcpyDeviceToHost, c->stream));
  }
  COLTT_HIP(hipStreamSynchronize(c->stream));
  if (packed) {
    const uint8_t* hb = c->h_out.as<uint8_t>();
    std::memcpy(out_ids, hb, nq * k * 8);
    std::memcpy(out_scores, hb + nq * k * 8, nq * k * 4);
    std::memcpy(out_counts, hb + nq * k * 12, nq * 4);
  }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  f->last_ms.store(ms);
  return COLTT_OK;
}

// Slot plan of an upsert, computed WITHOUT touching the store: an existing id keeps its slot, new ids get n, n+1, ...
// (a repeated new id gets one slot).  The store's maps, id table and row count change only in commit_upsert, after the
// device work has succeeded — a failed upsert (NOMEM on growth, device error) leaves the store exactly as it was.
struct UpsertPlan { std::vector<uint32_t> slots; std::vector<uint64_t> new_ids; uint64_t nn = 0; };
int plan_upsert(const Flat* f, const uint64_t* ids, uint64_t first_id, size_t n, UpsertPlan& p) {
  p.slots.resize(n); p.nn = f->n;
  std::unordered_map<uint64_t, uint32_t> fresh;
  for (size_t i = 0; i < n; i++) {
    const uint64_t id = ids ? ids[i] : first_id + i;
    auto it = f->id2slot.find(id);
    if (it != f->id2slot.end()) { p.slots[i] = it->second; continue; }
    auto r = fresh.emplace(id, (uint32_t)p.nn);
    if (r.second) { p.new_ids.push_back(id); p.nn++; }
    p.slots[i] = r.first->second;
  }
  if (p.nn > 0xffffffffull) return fail(COLTT_E_UNSUPPORTED, "flat upsert: more than 2^32-1 rows in one store");
  return COLTT_OK;
}
void commit_upsert(Flat* f, const UpsertPlan& p) {
  for (size_t j = 0; j < p.new_ids.size(); j++) { f->id2slot[p.new_ids[j]] = (uint32_t)(f->n + j); f->h_ids.push_back(p.new_ids[j]); }
  f->n = p.nn;
}

}  // namespace

extern "C" {

int coltt_flat_create(uint32_t dim, int metric, int quant, coltt_handle_t* out) {
  return coltt::flat_create_on(-1 /* the process default device, selected after the arguments have been validated */, dim, metric, quant, out);
}

}  // extern "C"

// a store on an explicit device (collection groups place one shard per GPU)
int coltt::flat_create_on(int device, uint32_t dim, int metric, int quant, coltt_handle_t* out) {
  if (!out) return fail(COLTT_E_INVALID, "flat_create: out is NULL");
  if (dim == 0 || dim > 8192) return fail(COLTT_E_INVALID, "flat_create: dim %u outside [1,8192]", dim);
  if (metric != COLTT_COSINE && metric != COLTT_EUCLIDEAN) return fail(COLTT_E_INVALID, "flat_create: bad metric %d", metric);
  if (quant < COLTT_Q_NONE || quant > COLTT_Q_BF16) return fail(COLTT_E_UNSUPPORTED, "not support quantization type");  // vectorstore.go:79
  COLTT_DEVICE(device); device = coltt_dev_scope_.device();
  auto f = std::make_shared<Flat>();
  f->dim = dim; f->metric = metric; f->quant = quant;
  f->stride = ((size_t)dim * quant_bytes(quant) + 15) & ~(size_t)15;
  f->stride16 = ((size_t)dim * 2 + 15) & ~(size_t)15;
  f->f8x = quant == COLTT_Q_F8 && metric == COLTT_COSINE && policy().f8_mfma;
  f->device = device;
  COLTT_HIP(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
  COLTT_TRY(f->d_maxn.reserve(16));
  COLTT_HIP(hipMemsetAsync(f->d_maxn.p, 0, 16, f->stream));
  *out = Registry::get().add(f);
  return COLTT_OK;
}

extern "C" {

int coltt_flat_destroy(coltt_handle_t h) {
  if (!Registry::get().erase(h)) return fail(COLTT_E_NOT_FOUND, "flat_destroy: unknown handle");
  return COLTT_OK;
}

int coltt_flat_reserve(coltt_handle_t h, uint64_t n_rows) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_reserve: unknown handle");
  WriteLock g(f->rw);
  COLTT_DEVICE(f->device);
  return f->reserve(n_rows);
}

int coltt_flat_upsert(coltt_handle_t h, const uint64_t* ids, const float* vecs, size_t n) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_upsert: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!ids || !vecs) return fail(COLTT_E_INVALID, "flat_upsert: NULL input");
  WriteLock g(f->rw);
  COLTT_DEVICE(f->device);
  COLTT_TRY(f->undense());
  UpsertPlan pl;
  COLTT_TRY(plan_upsert(f.get(), ids, 0, n, pl));
  COLTT_TRY(f->reserve(pl.nn));
  // a repeated id inside the batch: last one wins — upload only the winning rows
  const float* src = vecs; const uint32_t* sl = pl.slots.data(); size_t m = n;
  std::vector<uint32_t> s2; std::vector<float> v2;
  {
    std::unordered_map<uint32_t, size_t> last;
    for (size_t i = 0; i < n; i++) last[pl.slots[i]] = i;
    if (last.size() != n) {
      s2.reserve(last.size()); v2.reserve(last.size() * f->dim);
      for (size_t i = 0; i < n; i++) if (last[pl.slots[i]] == i) { s2.push_back(pl.slots[i]); v2.insert(v2.end(), vecs + i * f->dim, vecs + (i + 1) * f->dim); }
      src = v2.data(); sl = s2.data(); m = s2.size();
    }
  }
  COLTT_TRY(f->w_raw.reserve(m * f->dim * 4));
  COLTT_TRY(f->w_slots.reserve(m * 4));
  COLTT_HIP(hipMemcpyAsync(f->w_raw.p, src, m * f->dim * 4, hipMemcpyHostToDevice, f->stream));
  COLTT_HIP(hipMemcpyAsync(f->w_slots.p, sl, m * 4, hipMemcpyHostToDevice, f->stream));
  COLTT_TRY(prep_rows(f.get(), f->w_raw.as<float>(), m, f->w_slots.as<uint32_t>(), 0));
  if (!pl.new_ids.empty()) COLTT_HIP(hipMemcpyAsync(f->ids.as<uint64_t>() + f->n, pl.new_ids.data(), pl.new_ids.size() * 8, hipMemcpyHostToDevice, f->stream));
  COLTT_HIP(hipStreamSynchronize(f->stream));
  commit_upsert(f.get(), pl);
  return COLTT_OK;
}

int coltt_flat_upsert_device(coltt_handle_t h, const uint64_t* ids, uint64_t first_id, const float* d_vecs, size_t n) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_upsert_device: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!d_vecs) return fail(COLTT_E_INVALID, "flat_upsert_device: NULL vectors");
  WriteLock g(f->rw);
  COLTT_DEVICE(f->device);
  if (!ids && f->dense && (f->n == 0 || first_id == f->dense_base + f->n)) {  // append-only dense fast path
    if (f->n + n > 0xffffffffull) return fail(COLTT_E_UNSUPPORTED, "flat upsert: more than 2^32-1 rows in one store");
    COLTT_TRY(f->reserve(f->n + n));
    COLTT_TRY(prep_rows(f.get(), d_vecs, n, nullptr, f->n));
    COLTT_HIP(hipStreamSynchronize(f->stream));
    if (f->n == 0) f->dense_base = first_id;
    f->n += n;
    return COLTT_OK;
  }
  COLTT_TRY(f->undense());
  UpsertPlan pl;
  COLTT_TRY(plan_upsert(f.get(), ids, first_id, n, pl));
  COLTT_TRY(f->reserve(pl.nn));
  COLTT_TRY(f->w_slots.reserve(n * 4));
  COLTT_HIP(hipMemcpyAsync(f->w_slots.p, pl.slots.data(), n * 4, hipMemcpyHostToDevice, f->stream));
  COLTT_TRY(prep_rows(f.get(), d_vecs, n, f->w_slots.as<uint32_t>(), 0));
  if (!pl.new_ids.empty()) COLTT_HIP(hipMemcpyAsync(f->ids.as<uint64_t>() + f->n, pl.new_ids.data(), pl.new_ids.size() * 8, hipMemcpyHostToDevice, f->stream));
  COLTT_HIP(hipStreamSynchronize(f->stream));
  commit_upsert(f.get(), pl);
  return COLTT_OK;
}

int coltt_flat_remove(coltt_handle_t h, const uint64_t* ids, size_t n) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_remove: unknown handle");
  if (n == 0) return COLTT_OK;
  if (!ids) return fail(COLTT_E_INVALID, "flat_remove: NULL ids");
  WriteLock g(f->rw);
  COLTT_DEVICE(f->device);
  COLTT_TRY(f->undense());
  for (size_t i = 0; i < n; i++) {
    auto it = f->id2slot.find(ids[i]);
    if (it == f->id2slot.end()) continue;  // delete() of a missing key is a no-op in Go
    uint32_t s = it->second;
    uint64_t last = f->n - 1;
    f->id2slot.erase(it);
    if (s != last) {  // move the last row into the hole
      uint8_t* R = f->rows.as<uint8_t>();
      COLTT_HIP(hipMemcpyAsync(R + (size_t)s * f->stride, R + (size_t)last * f->stride, f->stride, hipMemcpyDeviceToDevice, f->stream));
      COLTT_HIP(hipMemcpyAsync(f->norms.as<float>() + s, f->norms.as<float>() + last, 4, hipMemcpyDeviceToDevice, f->stream));
      if (f->f8x) {
        uint8_t* R16 = f->rows16.as<uint8_t>();
        COLTT_HIP(hipMemcpyAsync(R16 + (size_t)s * f->stride16, R16 + (size_t)last * f->stride16, f->stride16, hipMemcpyDeviceToDevice, f->stream));
        COLTT_HIP(hipMemcpyAsync(f->norms16.as<float>() + s, f->norms16.as<float>() + last, 4, hipMemcpyDeviceToDevice, f->stream));
      }
      uint64_t moved = f->h_ids[last];
      f->h_ids[s] = moved;
      f->id2slot[moved] = s;
      COLTT_HIP(hipMemcpyAsync(f->ids.as<uint64_t>() + s, f->ids.as<uint64_t>() + last, 8, hipMemcpyDeviceToDevice, f->stream));
    }
    f->h_ids.pop_back();
    f->n--;
  }
  COLTT_HIP(hipStreamSynchronize(f->stream));
  return COLTT_OK;
}

int coltt_flat_len(coltt_handle_t h, uint64_t* out) {
  auto f = lookup<Flat>(h);
  if (!f || !out) return fail(COLTT_E_NOT_FOUND, "flat_len: unknown handle");
  ReadLock g(f->rw);
  *out = f->n;
  return COLTT_OK;
}

int coltt_flat_get(coltt_handle_t h, uint64_t id, void* out_row) {
  auto f = lookup<Flat>(h);
  if (!f || !out_row) return fail(COLTT_E_NOT_FOUND, "flat_get: unknown handle");
  ReadLock g(f->rw);
  COLTT_DEVICE(f->device);
  uint64_t slot;
  if (f->dense) { if (id < f->dense_base || id >= f->dense_base + f->n) return fail(COLTT_E_NOT_FOUND, "NodeID: %llu is not found", (unsigned long long)id); slot = id - f->dense_base; }
  else { auto it = f->id2slot.find(id); if (it == f->id2slot.end()) return fail(COLTT_E_NOT_FOUND, "NodeID: %llu is not found", (unsigned long long)id); slot = it->second; }
  COLTT_HIP(hipMemcpy(out_row, f->rows.as<uint8_t>() + slot * f->stride, (size_t)f->dim * quant_bytes(f->quant), hipMemcpyDeviceToHost));
  return COLTT_OK;
}

int coltt_flat_fetch_rows(coltt_handle_t h, uint64_t first_slot, uint64_t n, void* out_rows, uint64_t* out_ids) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_fetch_rows: unknown handle");
  if (n == 0) return COLTT_OK;
  ReadLock g(f->rw);
  COLTT_DEVICE(f->device);
  if (first_slot + n > f->n) return fail(COLTT_E_INVALID, "flat_fetch_rows: range outside [0,%llu)", (unsigned long long)f->n);
  const size_t rb = (size_t)f->dim * quant_bytes(f->quant);
  if (out_rows) COLTT_HIP(hipMemcpy2D(out_rows, rb, f->rows.as<uint8_t>() + first_slot * f->stride, f->stride, rb, n, hipMemcpyDeviceToHost));
  if (out_ids) for (uint64_t i = 0; i < n; i++) out_ids[i] = f->dense ? f->dense_base + first_slot + i : f->h_ids[first_slot + i];
  return COLTT_OK;
}

int coltt_flat_search(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select, int mode,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_counts) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_search: unknown handle");
  if (nq && (!queries || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "flat_search: NULL buffer");
  ReadLock g(f->rw);
  COLTT_DEVICE(f->device);
  CtxLease<FCtx> ctx(f->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return flat_search_common(f.get(), ctx.c, queries, false, nq, k, select, mode, nullptr, f->n, out_ids, out_scores, out_counts, false);
}

int coltt_flat_search_device(coltt_handle_t h, const float* d_queries, size_t nq, uint32_t k, int select, int mode,
                             uint64_t* d_out_ids, float* d_out_scores, uint32_t* d_out_counts) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_search_device: unknown handle");
  if (nq && (!d_queries || !d_out_ids || !d_out_scores || !d_out_counts)) return fail(COLTT_E_INVALID, "flat_search_device: NULL buffer");
  ReadLock g(f->rw);
  COLTT_DEVICE(f->device);
  CtxLease<FCtx> ctx(f->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  return flat_search_common(f.get(), ctx.c, d_queries, true, nq, k, select, mode, nullptr, f->n, d_out_ids, d_out_scores, d_out_counts, true);
}

int coltt_flat_search_ids(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select,
                          const uint64_t* cand_ids, size_t n_cand, uint64_t* out_ids, float* out_scores,
                          uint32_t* out_counts) {
  return coltt_flat_search_ids_mode(h, queries, nq, k, select, COLTT_MODE_EXACT, cand_ids, n_cand, out_ids, out_scores, out_counts);
}

int coltt_flat_search_ids_mode(coltt_handle_t h, const float* queries, size_t nq, uint32_t k, int select, int mode,
                               const uint64_t* cand_ids, size_t n_cand, uint64_t* out_ids, float* out_scores,
                               uint32_t* out_counts) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_search_ids: unknown handle");
  if (nq && (!queries || !out_ids || !out_scores || !out_counts)) return fail(COLTT_E_INVALID, "flat_search_ids: NULL buffer");
  if (n_cand && !cand_ids) return fail(COLTT_E_INVALID, "flat_search_ids: NULL candidates");
  ReadLock g(f->rw);
  COLTT_DEVICE(f->device);
  // id -> slot; ids that are not stored are skipped (none_vectorstore.go:201 `if node, ok := ...; ok`);
  // a repeated candidate id is scored once (roaring64 ToArray yields a set, pkg/inverted/search.go:113-119)
  std::vector<uint32_t> slots;
  slots.reserve(n_cand);
  for (size_t i = 0; i < n_cand; i++) {
    uint64_t id = cand_ids[i];
    if (f->dense) { if (id >= f->dense_base && id < f->dense_base + f->n) slots.push_back((uint32_t)(id - f->dense_base)); }
    else { auto it = f->id2slot.find(id); if (it != f->id2slot.end()) slots.push_back(it->second); }
  }
  std::sort(slots.begin(), slots.end());
  slots.erase(std::unique(slots.begin(), slots.end()), slots.end());
  CtxLease<FCtx> ctx(f->pool);
  if (!ctx.c) return COLTT_E_DEVICE;
  COLTT_TRY(ctx.c->w_gather.reserve(std::max<size_t>(slots.size(), 1) * 4));
  if (!slots.empty()) COLTT_HIP(hipMemcpyAsync(ctx.c->w_gather.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, ctx.c->stream));
  return flat_search_common(f.get(), ctx.c, queries, false, nq, k, select, mode, ctx.c->w_gather.as<uint32_t>(), slots.size(),
                            out_ids, out_scores, out_counts, false);
}


/* LoadVertex (edge/none_vectorstore.go:425-516 and the f16/f8/bf16 twins) */
int coltt_flat_load_vertex(coltt_handle_t h, const uint8_t* buf, uint64_t len, uint64_t* out_n, uint64_t* out_ids,
                           uint64_t* out_meta_off, uint32_t* out_meta_len, uint64_t cap_n) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_load_vertex: unknown handle");
  if (!buf && len) return fail(COLTT_E_INVALID, "flat_load_vertex: NULL buffer");
  WriteLock g(f->rw);
  COLTT_DEVICE(f->device);
  FBER r{buf, len};
  const size_t eb = quant_bytes(f->quant);
  std::vector<uint64_t> ids, voff, moff; std::vector<uint32_t> mlen;
  for (int s = 0; s < 16 && r.ok; s++) {
    uint64_t cnt = r.u64();
    for (uint64_t i = 0; i < cnt && r.ok; i++) {
      ids.push_back(r.u64());
      uint32_t vl = r.u32();
      if (r.ok && vl != f->dim) return fail(COLTT_E_INVALID, "Dim Length UnmatchdError: expect dimension: [%u], but got [%u]", f->dim, vl);
      voff.push_back(r.i); r.need((uint64_t)vl * eb); r.i += (uint64_t)vl * eb;
      uint64_t m0 = r.i; uint32_t mc = r.u32();
      for (uint32_t m = 0; m < mc && r.ok; m++) {
        uint16_t kl = r.u16(); r.need(kl); r.i += kl; uint8_t tag = r.u8();
        if (tag == 0 || tag == 2) { r.need(8); r.i += 8; } else if (tag == 1) { uint16_t sl = r.u16(); r.need(sl); r.i += sl; }
        else if (tag == 3) { r.need(1); r.i += 1; } else return fail(COLTT_E_INVALID, "unsupported metadata type tag: %d", (int)tag);
      }
      moff.push_back(m0); mlen.push_back((uint32_t)(r.i - m0));
    }
  }
  if (!r.ok) return fail(COLTT_E_INVALID, "flat_load_vertex: truncated stream");
  const uint64_t n = ids.size();
  if (n > 0xffffffffull) return fail(COLTT_E_UNSUPPORTED, "flat_load_vertex: more than 2^32-1 rows");
  // the whole stream is parsed and validated before the store is touched; a device failure while the rows are being
  // replaced leaves an EMPTY store (LoadVertex swaps complete shard maps in, none_vectorstore.go:510-515)
  std::unordered_map<uint64_t, uint32_t> new_map;
  new_map.reserve(n * 2);
  for (uint64_t i = 0; i < n; i++) if (!new_map.emplace(ids[i], (uint32_t)i).second) return fail(COLTT_E_INVALID, "flat_load_vertex: duplicate key in stream");
  COLTT_TRY(f->reserve(std::max<uint64_t>(n, 1)));
  COLTT_TRY(f->ids.reserve(std::max<uint64_t>(f->cap, 1024) * 8, false, f->stream));
  f->n = 0; f->id2slot.clear(); f->h_ids.clear(); f->dense = false;
  auto upload = [&]() -> int {
    COLTT_HIP(hipMemcpyAsync(f->ids.p, ids.data(), n * 8, hipMemcpyHostToDevice, f->stream));
    const uint64_t rows_per = std::max<uint64_t>(1, (128ull << 20) / ((uint64_t)f->dim * eb));
    DevBuf d_chunk, d_offs;
    for (uint64_t b = 0; b < n; b += rows_per) {
      uint64_t m = std::min<uint64_t>(rows_per, n - b);
      uint64_t lo = voff[b], hi = voff[b + m - 1] + (uint64_t)f->dim * eb;
      std::vector<uint64_t> rel(m);
      for (uint64_t i = 0; i < m; i++) rel[i] = voff[b + i] - lo;
      COLTT_TRY(d_chunk.reserve(hi - lo)); COLTT_TRY(d_offs.reserve(m * 8));
      COLTT_HIP(hipMemcpyAsync(d_chunk.p, buf + lo, hi - lo, hipMemcpyHostToDevice, f->stream));
      COLTT_HIP(hipMemcpyAsync(d_offs.p, rel.data(), m * 8, hipMemcpyHostToDevice, f->stream));
      uint32_t grid = ceil_div(m * f->dim, 256);
      uint8_t* R = f->rows.as<uint8_t>();
#define COLTT_BE(Q) do { be_codes_kernel<Q><<<grid, 256, 0, f->stream>>>(d_chunk.as<uint8_t>(), d_offs.as<uint64_t>(), m, (int)f->dim, R, f->stride, b); \
        row_norms_kernel<Q><<<ceil_div(m * 2, 256), 256, 0, f->stream>>>(R, f->stride, nullptr, b, m, (int)f->dim, f->norms.as<float>(), f->d_maxn.as<uint32_t>()); \
        (void)hipMemcpyAsync(f->norm_bits, f->d_maxn.p, 8, hipMemcpyDeviceToHost, f->stream); } while (0)
      COLTT_DISPATCH_QUANT(f->quant, COLTT_BE)
#undef COLTT_BE
      if (f->f8x) f8_expand_kernel<<<ceil_div(m * (f->stride16 / 2), 256), 256, 0, f->stream>>>(R, f->stride, f->norms.as<float>(), nullptr, b, m, (int)f->dim,
                                                                                         f->rows16.as<uint8_t>(), f->stride16, f->norms16.as<float>());
      COLTT_HIP(hipGetLastError());
      COLTT_HIP(hipStreamSynchronize(f->stream));
    }
    return COLTT_OK;
  };
  if (n) COLTT_TRY(upload());
  f->id2slot = std::move(new_map);
  f->h_ids = ids;
  f->n = n;
  if (out_n) *out_n = n;
  for (uint64_t i = 0; i < n && i < cap_n; i++) {
    if (out_ids) out_ids[i] = ids[i];
    if (out_meta_off) out_meta_off[i] = moff[i];
    if (out_meta_len) out_meta_len[i] = mlen[i];
  }
  return COLTT_OK;
}

/* SaveVertex (edge/none_vectorstore.go:308-423 and twins).  meta_blobs[slot] = {u32 metaCount, typed pairs} or NULL. */
int coltt_flat_save_vertex(coltt_handle_t h, const uint64_t* meta_ids, const uint8_t* const* meta_blobs, const uint32_t* meta_lens,
                           uint64_t n_meta, uint8_t* out, uint64_t cap, uint64_t* out_len) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_save_vertex: unknown handle");
  if (!out_len) return fail(COLTT_E_INVALID, "flat_save_vertex: out_len is NULL");
  ReadLock g(f->rw);
  COLTT_DEVICE(f->device);
  const size_t eb = quant_bytes(f->quant);
  std::unordered_map<uint64_t, uint64_t> meta_of;
  for (uint64_t i = 0; i < n_meta; i++) if (meta_ids && meta_blobs && meta_blobs[i] && meta_lens[i] >= 4) meta_of[meta_ids[i]] = i;
  auto id_of = [&](uint64_t s) { return f->dense ? f->dense_base + s : f->h_ids[s]; };
  // canonical order: shard 0..15 (FNV ShardVertex), ascending id inside (Go iterates its maps in random order)
  std::vector<std::vector<std::pair<uint64_t, uint32_t>>> shards(16);
  for (uint64_t s = 0; s < f->n; s++) shards[shard_vertex(id_of(s), 16)].push_back({id_of(s), (uint32_t)s});
  for (auto& sh : shards) std::sort(sh.begin(), sh.end());
  std::vector<uint8_t> all;
  if (out && f->n) { all.resize(f->n * (size_t)f->dim * eb);
    COLTT_HIP(hipMemcpy2D(all.data(), (size_t)f->dim * eb, f->rows.p, f->stride, (size_t)f->dim * eb, f->n, hipMemcpyDeviceToHost)); }
  FBEW w{out, out ? cap : 0};
  for (auto& sh : shards) {
    w.u64(sh.size());
    for (auto& e : sh) {
      w.u64(e.first); w.u32(f->dim);
      if (out) {
        const uint8_t* p = all.data() + (size_t)e.second * f->dim * eb;
        for (uint32_t k = 0; k < f->dim; k++) { uint8_t b[4]; for (size_t j = 0; j < eb; j++) b[j] = p[k * eb + (eb - 1 - j)]; w.put(b, eb); }
      } else w.n += (uint64_t)f->dim * eb;
      auto it = meta_of.find(e.first);
      if (it != meta_of.end()) w.put(meta_blobs[it->second], meta_lens[it->second]); else w.u32(0);
    }
  }
  *out_len = w.n;
  if (out && w.n > cap) return fail(COLTT_E_INVALID, "flat_save_vertex: buffer too small");
  return COLTT_OK;
}

int coltt_flat_stats(coltt_handle_t h, uint64_t* mfma_groups, uint64_t* mfma_fallbacks) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_stats: unknown handle");
  if (mfma_groups) *mfma_groups = f->mfma_groups.load();
  if (mfma_fallbacks) *mfma_fallbacks = f->mfma_fallbacks.load();
  return COLTT_OK;
}

int coltt_flat_norm_bounds(coltt_handle_t h, float* out_min, float* out_max, int32_t* out_cosine_matrix_core_open) {
  auto f = lookup<Flat>(h);
  if (!f) return fail(COLTT_E_NOT_FOUND, "flat_norm_bounds: unknown handle");
  ReadLock g(f->rw);
  const float mx = f->max_norm(), mn = f->min_norm();
  if (out_min) *out_min = mn;
  if (out_max) *out_max = mx;
  if (out_cosine_matrix_core_open)
    *out_cosine_matrix_core_open = f->metric == COLTT_COSINE &&
      (f->quant == COLTT_Q_F8 ? (f->f8x && mx == mx && mx < 3.0e38f) : (mn >= 0.25f && mx <= 4.0f)) ? 1 : 0;
  return COLTT_OK;
}

int coltt_flat_one_launch_searches(coltt_handle_t h, uint64_t* out) {
  auto f = lookup<Flat>(h);
  if (!f || !out) return fail(COLTT_E_NOT_FOUND, "flat_one_launch_searches: unknown handle");
  *out = f->one_groups.load();
  return COLTT_OK;
}

int coltt_last_kernel_ms_flat(coltt_handle_t h, float* out_ms) {
  auto f = lookup<Flat>(h);
  if (!f || !out_ms) return fail(COLTT_E_NOT_FOUND, "last_kernel_ms: unknown handle");
  *out_ms = f->last_ms.load();
  return COLTT_OK;
}

}  // extern "C"
