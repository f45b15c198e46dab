// hnsw_walk2.hpp — Hnsw.searchLevel at level 0 for LARGE ef (the recall >= 0.98 operating points: ef 256 ... 4096).
//
// Same closed form of core/vectorindex/hnsw.go:345-389 as hnsw_dev.hpp:search_level (stale lowerBound per popped candidate, canonical
// neighbour order, ties by (distance, slot), the ef smallest kept) and the same exact-order distances — ids, score bits and the
// traversal counters stay equal to the oracle's — but the three things that made the large-ef walk slow are restructured:
//
//  W2_DELTA  result set = a SORTED main array in LDS + a SORTED delta of <= 64 entries held in registers (one per lane, ascending).
//            An admission is a one-lane shift of the delta's tail (DPP wave_shr:1) behind one 64-bit compare; the largest member is
//            the larger of the main array's tail and the delta's last lane (two v_readlane), the smallest unexpanded member of the
//            delta is the first set bit of one ballot — no cross-lane reductions anywhere on the expansion chain (round 6; rounds 4-5
//            kept the delta UNSORTED and paid a 6-step DPP arg-max per admission batch and per eviction, a DPP arg-min or two per pop
//            and a rank loop per flush).  The delta is merged into the main array only when it is full — one pass over the array per
//            64 admissions instead of a binary search + partial shift per admission.  The set is the same set at every step, so pop
//            order, lowerBound and the final order are unchanged.
//  W2_BLOOM  a blocked Bloom filter in LDS (2 bits in one 32-bit word, one ds_or_rtn per test) in front of the HBM byte-per-slot
//            visited map: a negative answer is definite ("never tested in this traversal"), so only positives pay the dependent
//            HBM probe; every fresh vertex is still recorded in the byte map, which stays the exact set.
//  W2_ADJN   the neighbours' ||row||^2 (cosine) come with the adjacency row (GraphView::adj0_n) instead of one 4-byte gather per
//            evaluation — a gather that costs a whole cache line of HBM traffic each.
//
// The wave must be the only one in its workgroup (wave_sync is a wave-level LDS fence).  VISG (HBM byte map) only.
#pragma once
#include <type_traits>
#include "hnsw_dev.hpp"
#include "rows8.hpp"

namespace coltt {
namespace dev {

enum { W2_BLOOM = 1, W2_DELTA = 2, W2_ADJN = 4 };

// ---- wave64 reductions through DPP (no LDS crossbar): quad_perm, row_shr, row_bcast — the result lands in lane 63 -------------
// One reduction step = ONE v_min_u32_dpp / v_max_u32_dpp: the DPP move is given the operation's identity as its `old` value (lanes without a source,
// rows outside ROW_MASK), which is the form the compiler's DPP combine folds into the arithmetic instruction (old == the value itself compiled to
// v_mov + v_mov_dpp + v_min: three issue slots and a longer hazard stall per step, six steps per reduction, five to eight reductions per expansion).
template <int CTRL, int ROW_MASK> __device__ __forceinline__ uint32_t dpp_or(uint32_t v, uint32_t ident) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)ident, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = umin32(dpp_or<0xB1, 0xf>(v, 0xffffffffu), v);    // quad_perm [1,0,3,2]
  v = umin32(dpp_or<0x4E, 0xf>(v, 0xffffffffu), v);    // quad_perm [2,3,0,1]
  v = umin32(dpp_or<0x114, 0xf>(v, 0xffffffffu), v);   // row_shr:4
  v = umin32(dpp_or<0x118, 0xf>(v, 0xffffffffu), v);   // row_shr:8   -> lane 15 of every row holds the row's minimum
  v = umin32(dpp_or<0x142, 0xa>(v, 0xffffffffu), v);   // row_bcast:15 into rows 1 and 3
  v = umin32(dpp_or<0x143, 0xc>(v, 0xffffffffu), v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's minimum
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = umax32(dpp_or<0xB1, 0xf>(v, 0u), v);
  v = umax32(dpp_or<0x4E, 0xf>(v, 0u), v);
  v = umax32(dpp_or<0x114, 0xf>(v, 0u), v);
  v = umax32(dpp_or<0x118, 0xf>(v, 0u), v);
  v = umax32(dpp_or<0x142, 0xa>(v, 0u), v);
  v = umax32(dpp_or<0x143, 0xc>(v, 0u), v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// lane holding the smallest / largest 64-bit key (hi:lo) among the lanes with `valid` (wave-uniform; -1 when there is none).
// Keys are distinct.  The common case — one lane holds the extreme distance — costs one 32-bit reduction and a ballot.
__device__ __forceinline__ int wave_argmin_key(bool valid, uint32_t hi, uint32_t lo, unsigned long long& key) {
  if (!__ballot(valid)) { key = ~0ull; return -1; }
  const uint32_t mh = wave_min_u32(valid ? hi : 0xffffffffu);
  const bool tie = valid && hi == mh;
  const unsigned long long t = __ballot(tie);
  int L; uint32_t ml;
  if (__popcll(t) == 1) { L = __builtin_ctzll(t); ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, L); }
  else { ml = wave_min_u32(tie ? lo : 0xffffffffu); L = __builtin_ctzll(__ballot(tie && lo == ml)); }
  key = ((unsigned long long)mh << 32) | ml;
  return L;
}
__device__ __forceinline__ int wave_argmax_key(bool valid, uint32_t hi, uint32_t lo, unsigned long long& key) {
  if (!__ballot(valid)) { key = 0ull; return -1; }
  const uint32_t mh = wave_max_u32(valid ? hi : 0u);
  const bool tie = valid && hi == mh;
  const unsigned long long t = __ballot(tie);
  int L; uint32_t ml;
  if (__popcll(t) == 1) { L = __builtin_ctzll(t); ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, L); }
  else { ml = wave_max_u32(tie ? lo : 0u); L = __builtin_ctzll(__ballot(tie && lo == ml)); }
  key = ((unsigned long long)mh << 32) | ml;
  return L;
}

template <int METRIC, int QUANT, int PROFILE, bool R8 = false>
__device__ __forceinline__ float eval_pair_n(const GraphView& g, const WaveCtx& w, uint32_t slot, int half, float rn) {
  constexpr int U = burst_depth<QUANT, PROFILE>();
  if constexpr (R8) return pair_distance_r8<METRIC, QUANT, U / (QUANT == Q_NONE ? 4 : 8)>(g.rows + (size_t)slot * g.stride, w.qs, g.dim, w.qnorm, rn, half);
  else return pair_distance<METRIC, QUANT, U>(g.rows + (size_t)slot * g.stride, w.qs, g.dim, w.qnorm, rn, half);
}

// a < b for two WAVE-UNIFORM 64-bit keys (d bits << 32 | slot << 1 | expanded), on the scalar unit: the distance words decide unless they are equal
// (hipcc compares uniform 64-bit integers on the VALU — two v_mov, v_cmp_lt_u64, then s_and / s_cselect behind the VALU's latency — six to eight times
// per expansion of the large-ef walks)
// MEASURED (GPU call H, profiles/r06h_setcache_keylt_ab.md): with the scalar form the operating-point row walk lost 11 % (126.1 against 141.6 k queries/s on one
// box) — the select chains lengthen the scalar dependence of every pop, and the register allocation of the eight-lane walk moved with them.  The
// compiler's own form is the default; -DCOLTT_KEY_SCALAR=1 is the A/B partner.
#ifndef COLTT_KEY_SCALAR
#define COLTT_KEY_SCALAR 0
#endif
__device__ __forceinline__ bool key_lt(unsigned long long a, unsigned long long b) {
#if COLTT_KEY_SCALAR
  const uint32_t ah = (uint32_t)(a >> 32), bh = (uint32_t)(b >> 32);
  return ah != bh ? ah < bh : (uint32_t)a < (uint32_t)b;
#else
  return a < b;
#endif
}
__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b) { return key_lt(b, a) ? b : a; }

// The delta: one (hi, lo) key per lane, lanes [0, n) valid and ASCENDING by key (bit 0 of lo = expanded; slots are distinct, so bit 0 never
// decides an order); lanes >= n hold stale values.  n is wave-uniform.  Every member function is called by all 64 lanes.
struct Delta {
  uint32_t hi, lo;
  uint32_t n;
  __device__ __forceinline__ void clear() { hi = lo = 0xffffffffu; n = 0; }
  // the largest key, bit 0 cleared (0 when empty)
  __device__ __forceinline__ unsigned long long max_key() const {
    const int L = (int)n - 1;   // n == 0: lane 63's stale value, discarded
    const uint32_t mh = (uint32_t)__builtin_amdgcn_readlane((int)hi, L), ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, L);
    return n ? ((((unsigned long long)mh) << 32) | (ml & ~1u)) : 0ull;
  }
  // ballot of the unexpanded members (ascending: the first set bit is the smallest)
  __device__ __forceinline__ unsigned long long unexpanded(int lane) const {
    const unsigned long long live = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    (void)lane;
    return __ballot(!(lo & 1u)) & live;
  }
  __device__ __forceinline__ unsigned long long key_at(int L) const {
    const uint32_t mh = (uint32_t)__builtin_amdgcn_readlane((int)hi, L), ml = (uint32_t)__builtin_amdgcn_readlane((int)lo, L);
    return (((unsigned long long)mh) << 32) | ml;
  }
  // insert the wave-uniform key (vh, vl) (n < 64): members behind its position move up one lane
  __device__ __forceinline__ void insert(uint32_t vh, uint32_t vl, int lane) {
    const unsigned long long live = (1ull << n) - 1ull;
    const unsigned long long mine = (((unsigned long long)hi) << 32) | lo, key = (((unsigned long long)vh) << 32) | vl;
    const uint32_t pos = (uint32_t)__popcll(__ballot(mine < key) & live);   // a prefix of the live lanes
    const uint32_t sh = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, 0x138, 0xf, 0xf, false);   // wave_shr:1 — lane i takes lane i - 1
    const uint32_t sl = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, 0x138, 0xf, 0xf, false);
    if ((uint32_t)lane > pos) { hi = sh; lo = sl; }
    if ((uint32_t)lane == pos) { hi = vh; lo = vl; }
    n++;
  }
};

// Merge the delta into the sorted main array res[0, len), in place (len + delta.n <= ef_pad).  scan_lo: see search_level2.
__device__ __forceinline__ void delta_flush(unsigned long long* res, uint32_t& len, Delta& dl, uint32_t& scan_lo, int lane) {
  if (dl.n == 0) return;
  const bool valid = (uint32_t)lane < dl.n;
  const unsigned long long key = ((unsigned long long)dl.hi << 32) | dl.lo;
  uint32_t pos = 0xffffffffu;
  if (valid) {  // insertion point in the main array (all delta keys at once; keys are distinct, bit 0 never decides an order)
    uint32_t lo = 0, hi = len;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (res[mid] < key) lo = mid + 1; else hi = mid; }
    pos = lo;
  }
  const uint32_t r = (uint32_t)lane;  // rank among the delta keys: the delta is sorted
  const uint32_t minpos = (uint32_t)__builtin_amdgcn_readfirstlane((int)pos);   // lane 0 holds the smallest key, hence the smallest insertion point
  scan_lo = minpos < scan_lo ? minpos : scan_lo;  // flushed keys may be unexpanded
  wave_sync();
  if (len) {
    // member i moves up by #{delta keys whose insertion point is <= i}; chunks in descending order, targets lie <= 64 above, i.e.
    // in territory that has already been moved; members before the smallest insertion point stay
    for (int base = (int)((len - 1) & ~63u); base >= (int)(minpos & ~63u); base -= 64) {
      const uint32_t i = (uint32_t)base + lane;
      const unsigned long long e = i < len ? res[i] : ~0ull;
      uint32_t shift = __popcll(__ballot(valid && pos < (uint32_t)base));
      unsigned long long im = __ballot(valid && pos >= (uint32_t)base && pos < (uint32_t)base + 64u);
      while (im) {
        const int j = __builtin_ctzll(im); im &= im - 1;
        const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)pos, j);
        shift += (pj <= i) ? 1u : 0u;
      }
      wave_sync();  // every lane holds its member before any lane overwrites one
      if (i < len && shift) res[i + shift] = e;
    }
  }
  wave_sync();
  if (valid) res[pos + r] = key;
  len += dl.n;
  dl.clear();
  wave_sync();
}

#ifndef COLTT_EVICT_BULK   // A/B knob: evictions of more than two members by one merge-path step (below)
#define COLTT_EVICT_BULK 1
#endif
// Drop the `e` largest members of main ∪ delta ("keep the ef smallest").
// The main array's TAIL held in registers across expansions (SETCACHE walks): lane t holds res[tb + t] as it was when the window was loaded.  Nothing but
// a flush adds to or reorders the main array, evictions only shorten it from the end, and the `expanded` bit a pop sets later is masked wherever the
// window is read — so the window stays usable until a flush (invalidate) or until more than its 64 entries have been evicted (reload).
struct TailWin {
  unsigned long long e; uint32_t tb; bool valid;
  __device__ __forceinline__ void load(const unsigned long long* res, uint32_t len, int lane) {
    tb = len > 64u ? len - 64u : 0u;
    e = tb + (uint32_t)lane < len ? res[tb + lane] : 0ull;
    valid = true;
  }
  // the largest main member with the expanded bit cleared (0 when the array is empty)
  __device__ __forceinline__ unsigned long long last(const unsigned long long* res, uint32_t len, int lane) {
    if (!len) return 0ull;
    if (!valid || len <= tb) load(res, len, lane);   // wave-uniform
    return readlane_u64(e, (int)(len - 1u - tb)) & ~1ull;
  }
};
__device__ __forceinline__ void evict_largest(const unsigned long long* res, uint32_t& len, Delta& dl, uint32_t e, int lane, TailWin* tw = nullptr) {
  // the main array's tail in registers: lane t holds res[tb + t]; at most 32 members leave per call
  if (tw) {
    for (uint32_t t = 0; t < e; t++) {
      const unsigned long long mt = tw->last(res, len, lane);
      if (dl.n && (!len || key_lt(mt, dl.max_key()))) dl.n--;   // the largest member is the delta's last lane
      else len--;
    }
    return;
  }
#if COLTT_EVICT_BULK
  if (e > 2u && e <= 64u) {
    // MERGE PATH (round 6): the e largest of main ∪ delta in one step instead of e dependent rounds of four v_readlane + a 64-bit compare.  Both are sorted; with
    // D_i = the delta's i-th largest key and M_j = the main array's j-th largest (expanded bits cleared), D_i is among the e largest of the union iff fewer than
    // e - i main members exceed it, i.e. iff M_{e-1-i} < D_i (or the array has no such member) — true for a prefix of i: their count leaves the delta, the rest
    // of e leaves the array.  Lane i tests D_i (ds_bpermute from lane n - 1 - i) against M_{e-1-i} (one LDS read).  Keys are distinct.
    const uint32_t i = (uint32_t)lane;
    const bool in = i < e && i < dl.n;
    const int src = (int)((dl.n - 1u - i) & 63u) << 2;
    const uint32_t dh = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dl.hi), dlo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dl.lo) & ~1u;
    const uint32_t j = e - 1u - i;                       // (meaningful for i < e)
    const bool mj = in && j < len;
    const unsigned long long M = mj ? (res[len - 1u - j] & ~1ull) : 0ull;
    const unsigned long long D = (((unsigned long long)dh) << 32) | dlo;
    const uint32_t x = (uint32_t)__popcll(__ballot(in && (!mj || D > M)));
    dl.n -= x; len -= e - x;
    return;
  }
#endif
  const uint32_t tb = len > 64u ? len - 64u : 0u;
  const unsigned long long treg = tb + (uint32_t)lane < len ? res[tb + lane] : 0ull;
  for (uint32_t t = 0; t < e; t++) {
    const unsigned long long mt = len ? (readlane_u64(treg, (int)(len - 1u - tb)) & ~1ull) : 0ull;
    if (dl.n && (!len || dl.max_key() > mt)) dl.n--;   // the largest member is the delta's last lane
    else len--;
  }
}

#ifndef COLTT_PQ_BYSET
#define COLTT_PQ_BYSET 0
#endif
// Is the lane's key (khi = distance bits, klo = slot << 1) a CURRENT member of main ∪ delta?  (COLTT_PQ_BYSET, see search_level2; EVERY lane must call — never
// behind a per-lane `want && ...` short circuit: the delta's keys sit one per lane, an inactive lane's key is not compared — that bug cost three GPU calls.)  Called by every lane of the wave; the lanes with `want` test.
// The main array is sorted by key and a member's key differs from the probe in the `expanded` bit at most: a binary search; the delta's <= 64 keys sit one
// per lane: one broadcast + compare per lane that the array did not settle.  LDS and registers only — no memory request.
__device__ __forceinline__ bool set_member(const unsigned long long* res, uint32_t len, const Delta& dl, bool want, uint32_t khi, uint32_t klo, uint32_t rej_slot, int lane) {
  // lower bound of the key in the main array, the SAME number of steps in every lane (a lane without a key searches for ~0): no divergent loop in front of
  // the cross-lane operations below
  const unsigned long long key = want ? ((((unsigned long long)khi) << 32) | klo) : ~0ull;
  uint32_t base = 0;
  for (uint32_t sz = len; sz > 1u;) {   // wave-uniform trip count
    const uint32_t h = sz >> 1;
    base = res[base + h - 1u] < key ? base + h : base;
    sz -= h;
  }
  bool member = false;
  if (len) {
    const unsigned long long e0 = res[base];
    const uint32_t pos = base + (e0 < key ? 1u : 0u);
    const unsigned long long e = pos < len ? res[pos] : ~0ull;
    member = want && pos < len && ((e ^ key) >> 1) == 0ull;
  }
  // the delta's keys (one per lane) and the vertices the walk REJECTED in the expansion that filled the set (rej_slot, one per lane or NBR_NONE): they were
  // tested against a lowerBound sampled before that expansion's unconditional admissions raised the worst member, so they may lie under a later bound
  // without being members — the only visited non-members that can
  unsigned long long cand = __ballot(want && !member);
  while (cand) {
    const int j = __builtin_ctzll(cand); cand &= cand - 1;
    const uint32_t kh = (uint32_t)__builtin_amdgcn_readlane((int)khi, j), kl = (uint32_t)__builtin_amdgcn_readlane((int)klo, j);
    const unsigned long long hit = __ballot(((uint32_t)lane < dl.n && dl.hi == kh && ((dl.lo ^ kl) >> 1) == 0u) || rej_slot == (kl >> 1));
    if (hit && lane == j) member = true;
  }
  return member;
}

// Where the distances of a chunk come from.  The throughput kernels evaluate them in place (one lane pair per row, exact.hpp); the
// latency kernel (hnsw_lat.hpp) hands the chunk to all four waves of its workgroup.  Called by every lane of the walking wave.
template <int METRIC, int QUANT, int PROFILE, bool ADJN, bool R8 = false> struct PairEval {
  static constexpr bool CHUNK_ADJ = false;   // the evaluator does not bring the neighbours' adjacency rows along
  static constexpr bool SPEC = false;        // no speculation on the next expansion's inputs (see AdcEval, hnsw_pq.hpp)
  static constexpr bool RADJ = false;        // the runner-up's adjacency row is not requested at pop time
  static constexpr bool ROWPF = false;       // no per-neighbour input addressed by (candidate, position) (see AdcEval<.., NBR>, hnsw_pq.hpp)
  static constexpr bool SETCACHE = false;    // the head / tail windows of the main array are not cached in registers (register budget of the row walks)
  static constexpr bool EARLY = false;       // distances are computed for the fresh neighbours only, after the visited test
  static constexpr bool BOUNDED = false;     // the reference's order: visited test first, every fresh neighbour evaluated (hnsw.go:366-373)
  static constexpr bool SPLIT = false;       // no evaluation pass in front of the visited probe's answer (see Group8Eval)
  __device__ __forceinline__ uint32_t chunk_adj(int, int) const { return NBR_NONE; }
  __device__ __forceinline__ void prefetch(uint32_t, bool, int) const {}   // nothing worth requesting before the visited test (a row is 1.5-3 KB)
  __device__ __forceinline__ float operator()(const GraphView& g, const WaveCtx& w, uint32_t nb, bool fresh, float nrm, int half, int /*lane*/) const {
    if (!fresh) return 0.f;
    if constexpr (ADJN) return eval_pair_n<METRIC, QUANT, PROFILE, R8>(g, w, nb, half, nrm);
    else return eval_pair<METRIC, QUANT, PROFILE, R8>(g, w, nb, half);
  }
};
// Eight lanes per row over the line-transposed copy (rows8.hpp).  The fresh neighbours of the chunk are compacted through LDS (slot
// and norm at their rank among the fresh ones), every 8-lane group evaluates ROWS of them per pass — 8 x ROWS rows per pass, whole
// 128-byte lines per load instruction — and the distances travel back through LDS to the lane pairs that own the neighbours in the
// walk (admission, ranks and keys stay where they were).  Same values in the same order as PairEval: same bits.
// Rows per lane group in flight x burst depth (128-byte lines), per kernel family — measured on the two 10 M x 768 shapes, every
// variant in one process against the pair-owned walk (profiles/r04d_ev8_variants.md, r04e_*): two rows x 6 lines for f32 rows and for
// the LDS-visited 2-byte kernel (one wave per SIMD), ONE row with all of its 12 lines in flight for the HBM-visited 2-byte kernels (two
// waves per SIMD; the deeper two-row bursts lose there: 2 x 12 lines -35 % on f32, -11 % on 2-byte rows).
#ifndef COLTT_G8_ROWS
#define COLTT_G8_ROWS 2
#endif
#ifndef COLTT_G8_U
#define COLTT_G8_U 6
#endif
#ifndef COLTT_G8_ROWS_NT32
#define COLTT_G8_ROWS_NT32 1
#endif
#ifndef COLTT_G8_U_NT32
#define COLTT_G8_U_NT32 12
#endif
#ifndef COLTT_G8_ROWS_H16
#define COLTT_G8_ROWS_H16 1
#endif
#ifndef COLTT_G8_U_H16
#define COLTT_G8_U_H16 12
#endif
// SPLIT (round 6, third session; A/B knob, exact, measured SLOWER, off): behind the Bloom filter a listed neighbour is either DEFINITELY fresh (a negative
// answer) or needs the dependent probe of the HBM byte map, and the walk waits for that probe before it requests a single row.  With the knob the probe is
// issued, the first full pass of definitely-fresh rows (8 x ROWS of them: two thirds of a listed row are Bloom negatives) is requested and evaluated under it,
// and the probe's answer is looked at only then; the remaining fresh neighbours follow in the usual passes.  Every neighbour's distance is the same function of
// the same row whichever pass computes it and admission sees all of them at once: ids, score bits and counters are unchanged (GPU call AE: same answers'
// hash from both libraries).  But the walk under load is bound by the memory system's row gather, not by one wave's dependent chain: 10 M x 768 f16, ef 1 024,
// same box: 79.5 ms per 10 000 queries without, 80.9 ms with (ef 256 and f32 rows: +-0) — profiles/r06ae_split_pass_ab.md.
#ifndef COLTT_G8_SPLIT
#define COLTT_G8_SPLIT 0
#endif
// ONEBURST (HBM-visited 2-byte kernels): rows of exactly U lines are read in one burst without a second buffer (rows8.hpp), and the registers that frees
// carry a second row per lane group — 16 rows per pass instead of 8, i.e. one dependent round trip less per expansion of ~21 fresh neighbours.
#ifndef COLTT_G8_ONEBURST_H16
#define COLTT_G8_ONEBURST_H16 0
#endif
#ifndef COLTT_G8_STREAM
#define COLTT_G8_STREAM 1
#endif
#ifndef COLTT_G8_ONEBURST   // the same for the other eight-lane kernels (f32 rows; 2-byte rows behind the LDS hash): rows of exactly COLTT_G8_U lines in one burst
#define COLTT_G8_ONEBURST 0
#endif
template <int METRIC, int QUANT, bool ADJN, bool HBM16 = false, bool NT = false> struct Group8Eval {   // NT: exact.hpp row_ld
  static constexpr bool ONEB = HBM16 ? COLTT_G8_ONEBURST_H16 != 0 : COLTT_G8_ONEBURST != 0;
  // f32 rows read with the non-temporal hint (collections far larger than the caches): ONE row per lane group with 12 lines in flight instead of two rows x 6 —
  // 10 M x 768 f32, hint on, same process: ef 128 19.39 -> 19.09 ms per 10 000 queries, ef 256 40.77 -> 40.31, ef 512 78.05 -> 77.46 (call AQ; +2.6 / +3 % on the
  // boxes of calls AO / AP); without the hint, and for 2-byte rows behind the LDS hash with or without it (9.7 -> 11.6 ms), two rows x 6 stay better.
  static constexpr bool F32NT = NT && QUANT == Q_NONE && !HBM16;
  static constexpr int G8R = (ONEB && HBM16) ? 2 : (HBM16 ? COLTT_G8_ROWS_H16 : (F32NT ? COLTT_G8_ROWS_NT32 : COLTT_G8_ROWS)),
                       G8U = HBM16 ? COLTT_G8_U_H16 : (F32NT ? COLTT_G8_U_NT32 : COLTT_G8_U);
  // COLTT_G8_STREAM (bit 0: the f32 non-temporal twins, bit 1: the HBM-visited 2-byte walk): rows of a whole number of bursts are evaluated as one stream of
  // bursts over all passes of the chunk — no bubble at the pass boundaries (rows8.hpp: group8_stream)
  static constexpr bool STREAM = G8R == 1 && !ONEB && ((F32NT && (COLTT_G8_STREAM & 1)) || (HBM16 && (COLTT_G8_STREAM & 2)) || (!HBM16 && (COLTT_G8_STREAM & 4)));   // bit 2: every other eight-lane kernel built with one row per group (A/B)
  static constexpr bool CHUNK_ADJ = false;
  static constexpr bool SPEC = false;
  static constexpr bool RADJ = false;
  static constexpr bool ROWPF = false;
  static constexpr bool EARLY = false;
  static constexpr bool BOUNDED = false;
  static constexpr bool SETCACHE = false;
  static constexpr bool SPLIT = COLTT_G8_SPLIT != 0;
  static constexpr int PASS_ROWS = 8 * G8R;   // rows one pass evaluates
  __device__ __forceinline__ uint32_t chunk_adj(int, int) const { return NBR_NONE; }
  __device__ __forceinline__ void prefetch(uint32_t, bool, int) const {}
  // distances of the chunk's `fresh` neighbours (one per lane pair, held by both lanes); the result is valid in BOTH lanes of a pair
  __device__ __forceinline__ float operator()(const GraphView& g, const WaveCtx& w, uint32_t nb, bool fresh, float nrm, int half, int lane) const {
    constexpr int ROWS = G8R;
    const bool mine = fresh && half == 0;
    const unsigned long long E = __ballot(mine);
    const uint32_t nf = (uint32_t)__popcll(E);
    const uint32_t rank = (uint32_t)__popcll(E & ((1ull << lane) - 1ull));
    uint32_t* const s_nb = w.scr; float* const s_nr = reinterpret_cast<float*>(w.scr + 32); float* const s_d = reinterpret_cast<float*>(w.scr + 64);
    if (mine) { s_nb[rank] = nb; if constexpr (ADJN) s_nr[rank] = nrm; }
    wave_sync();
    const int grp = lane >> 3, rj = lane & 7;
    const int nl = (g.dim * (QUANT == Q_NONE ? 4 : 2)) >> 7;
    bool streamed = false;
    if constexpr (STREAM) {   // one row per lane group: all passes as ONE stream of bursts (rows8.hpp: group8_stream)
      if (nl >= G8U && nl % G8U == 0) {
        group8_stream<METRIC, QUANT, G8U, NT, ADJN>(g.rows8, g.stride, g.norms, s_nb, s_nr, s_d, nf, grp, rj, w.qp, nl, w.qnorm);
        streamed = true;
      }
    }
    if (!streamed)
    for (uint32_t base = 0; base < nf; base += 8 * ROWS) {          // wave-uniform
      const uint8_t* rp[ROWS]; float rn[ROWS], d[ROWS]; uint32_t idx[ROWS]; bool live[ROWS];
#pragma unroll
      for (int i = 0; i < ROWS; i++) {
        idx[i] = base + (uint32_t)(i * 8 + grp);
        live[i] = idx[i] < nf;
        const uint32_t slot = s_nb[live[i] ? idx[i] : 0];
        rp[i] = g.rows8 + (size_t)slot * g.stride;
        rn[i] = 0.f;
        if constexpr (METRIC == M_COS) { if constexpr (ADJN) rn[i] = s_nr[live[i] ? idx[i] : 0]; else rn[i] = g.norms[slot]; }
      }
      group8_distance<METRIC, QUANT, ROWS, G8U, ONEB, NT>(rp, live, w.qp, nl, w.qnorm, rn, rj, d);
#pragma unroll
      for (int i = 0; i < ROWS; i++) if (rj == 0 && live[i]) s_d[idx[i]] = d[i];
    }
    wave_sync();
    float r = mine ? s_d[rank] : 0.f;
    wave_sync();   // the next chunk rewrites the scratch
    r = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, r), 0xA0, 0xf, 0xf, true));   // even lane's value to its pair: quad_perm [0,0,2,2]
    return r;
  }
  // Distance(query, one row), the same value in every lane (entrypoint, hnsw.go:253)
  __device__ __forceinline__ float one(const GraphView& g, const WaveCtx& w, uint32_t slot, int lane) const {
    const uint8_t* rp[1] = {g.rows8 + (size_t)slot * g.stride};
    const bool live[1] = {true};
    float rn[1] = {0.f}, d[1];
    if constexpr (METRIC == M_COS) rn[0] = g.norms[slot];
    group8_distance<METRIC, QUANT, 1, G8U, ONEB, NT>(rp, live, w.qp, (g.dim * (QUANT == Q_NONE ? 4 : 2)) >> 7, w.qnorm, rn, lane & 7, d);
    return d[0];
  }
};

// greedyClosestNeighbor (hnsw.go:320-343) with the eight-lane core: hnsw_dev.hpp:greedy_level with the distances of a chunk coming
// from Group8Eval (the upper rows carry no norms: the 4-byte gather serves the handful of evaluations up here)
template <int METRIC, int QUANT, bool HBM16, bool NT = false>
__device__ __forceinline__ void greedy_level8(const GraphView& g, WaveCtx& w, uint32_t& cur, float& curd, int level, int lane_in) {
  const Group8Eval<METRIC, QUANT, false, HBM16, NT> ev;
  for (uint32_t hops = 0;; hops++) {
    const int lane = opaque_lane(lane_in);
    const int half = lane & 1, p = lane >> 1;
    if (hops > (1u << 20)) { w.err |= 4u; break; }
    uint32_t width;
    const uint32_t* row = adj_row(g, cur, level, width);
    unsigned long long best = ~0ull;
    uint32_t best_slot = NBR_NONE;
    for (uint32_t c0 = 0; c0 < width; c0 += 32) {
      const uint32_t idx = c0 + p;
      const uint32_t nb = idx < width ? row[idx] : NBR_NONE;
      const bool valid = nb != NBR_NONE && !is_deleted(g, nb);
      float d = 0.f;
      if (__ballot(valid)) d = ev(g, w, nb, valid, 0.f, half, lane);
      w.n_dist += __popcll(__ballot(valid && half == 0));
      const unsigned long long key = valid ? (((unsigned long long)__float_as_uint(d) << 32) | idx) : ~0ull;
      const unsigned long long km = wave_min_u64(key);
      if (km < best) {
        best = km;
        const int src = (int)(((uint32_t)km - c0) * 2);
        best_slot = (uint32_t)__builtin_amdgcn_readlane((int)nb, src);
      }
    }
    w.n_hops++;
    const float bd = __uint_as_float((uint32_t)(best >> 32));
    if (best != ~0ull && bd < curd) { cur = best_slot; curd = bd; }
    else break;
  }
}
// Probe of the HBM byte map.  The region is this wave's own for the whole launch; the load must not be served by a stale L1 line (the wave's own marks went
// to L2): the agent-scope atomic load (`sc1`) bypasses L1 — and so does a non-temporal load (`nt`, MI355X_MICROARCH.md), which in addition tells L2 / MALL that
// the line will not be used again: a random probe pulls in a whole line for one byte, and what it displaces are the neighbourhood blocks and adjacency rows of hub
// vertices that other traversals do re-read.  COLTT_VIS_NT: 0 = `sc1` (rounds 2-6), 1 = `nt` (A/B knob).
#ifndef COLTT_VIS_NT
#define COLTT_VIS_NT 0
#endif
template <class T> __device__ __forceinline__ T vis_probe(const T* p) {
#if COLTT_VIS_NT
  return __builtin_nontemporal_load(p);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
enum { VIS_HBM = 0, VIS_LDS = 1 };   // visited set of search_level2: HBM byte map (behind the Bloom filter) | LDS hash that is never reset (err 8)

// searchLevel (hnsw.go:345-389) on level 0.  On return res[0, len) holds the result set ascending by (d, slot).
template <int METRIC, int QUANT, int PROFILE, int OPT, int VISMODE = VIS_HBM, bool ALWAYS_PREF = false, class EVAL = PairEval<METRIC, QUANT, PROFILE, (OPT & W2_ADJN) != 0 && METRIC == M_COS>>
__device__ __forceinline__ void search_level2(const GraphView& g, WaveCtx& w, uint32_t ep, float epd, uint32_t ef, int lane_in,
                                              uint32_t& out_len, EVAL&& ev = EVAL()) {
  constexpr bool BLOOM = (OPT & W2_BLOOM) != 0 && VISMODE == VIS_HBM, DELTA = (OPT & W2_DELTA) != 0, ADJN = (OPT & W2_ADJN) != 0 && METRIC == M_COS;
#ifdef COLTT_NO_ADJ_PREFETCH
  constexpr bool PREF = ALWAYS_PREF;
#else
  constexpr bool PREF = ALWAYS_PREF || QUANT != Q_NONE;  // see hnsw_dev.hpp: in the throughput kernels the adjacency prefetch pays for 2-/1-byte rows only
#endif
  int lane = lane_in;
  unsigned long long* const res = w.res0;
  uint32_t vis_count = 1;
  if constexpr (VISMODE == VIS_HBM) {
    // (a BIT per slot instead of a byte — 1/8 of the footprint and of its address translations, paid for with one 1.25 MB wipe per traversal
    //  and an atomic OR per test — lost: 10 M x 768 f16, ef 1024: 73.7 -> 77.0 ms per 10 k queries, f32 ef 256: 47.5 -> 49.0;
    //  profiles/r04l_visbits_ab.md.  The byte map with 8-bit epochs stays.)
    if (++w.epoch > 255u) {  // 8-bit epoch wrapped: wipe the region (once per 255 traversals)
      for (size_t i = (size_t)lane * 16; i < w.vis_bytes; i += 64 * 16) *reinterpret_cast<u32x4v*>(w.visg + i) = u32x4v{0, 0, 0, 0};
      __threadfence();
      w.epoch = 1;
    }
  } else vis_clear(w, lane);
  if constexpr (BLOOM) {
    for (uint32_t i = (uint32_t)lane * 4; i < w.bloom_words; i += 256) *reinterpret_cast<u32x4v*>(w.bloom + i) = u32x4v{0, 0, 0, 0};
  }
  if (lane == 0) res[0] = ((unsigned long long)__float_as_uint(epd) << 32) | ((unsigned long long)ep << 1);
  wave_sync();
  if (lane == 0) {
    if constexpr (VISMODE == VIS_HBM) __hip_atomic_store(w.visg + ep, (uint8_t)w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else vis_insert(w.vis, w.hcap_mask, ep);
    if constexpr (BLOOM) {
      const uint32_t h = ep * 0x9E3779B1u;
      w.bloom[h >> w.bloom_shift] |= (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
    }
  }
  uint32_t len = 1;
  uint32_t scan_lo = 0;   // every main-array member before this index is expanded (pop scans start at its 64-entry chunk)
  Delta dl; dl.clear();
  unsigned long long hwin = 1ull; uint32_t hb = 0; bool hvalid = false;   // SETCACHE: the head window (pop)
  TailWin twin; twin.e = 0ull; twin.tb = 0; twin.valid = false;           // SETCACHE: the tail window (lowerBound, eviction)
  uint32_t pre_slot = NBR_NONE, pre_nb = NBR_NONE; float pre_nn = 0.f;
  // SPEC evaluators (small per-neighbour inputs: hnsw_pq.hpp): when the next candidate is predicted to be the runner-up — whose adjacency row was
  // requested at pop time and has arrived by the end of the expansion — the visited bytes and the evaluator's inputs of ITS neighbours are requested
  // one expansion ahead (spec_slot = that candidate; spec_vis = the byte-map values).  The probe is issued after every visited mark of the current
  // expansion (program order, agent-scope accesses served by L2) and nothing marks in between, so the values are exactly what the probe of the
  // next expansion would read.
  uint32_t spec_slot = NBR_NONE; uint32_t spec_vis = 0;
  uint32_t rej_slot = NBR_NONE;   // BOUNDED: the fresh neighbour this lane pair rejected in the expansion that filled the set (see set_member)
  const uint32_t width = g.mMax0;
  wave_sync();
  for (uint32_t iters = 0;; iters++) {
    if (iters > (1u << 22)) { w.err |= 2u; break; }
    lane = opaque_lane(lane_in);
    const int half = lane & 1, p = lane >> 1;
    // ---- pop: the smallest unexpanded member of main ∪ delta (cj = the unexpanded main member after the first one).  The keys come out of
    // the registers the scan loaded its chunk into (v_readlane), not out of a second, dependent LDS read.
    typedef typename std::remove_reference<EVAL>::type eval_t;
    constexpr bool CACHE = DELTA && eval_t::SETCACHE;
    int ci = -1, cj = -1;
    unsigned long long kci = ~0ull, kcj = ~0ull;
    if constexpr (CACHE) {
      // SETCACHE: the 64-entry chunk that holds the first unexpanded main member stays in registers from pop to pop (hwin, base hb): only a flush changes
      // the array's members (it invalidates the window), a pop's `expanded` bit is set in the register too, evictions are a compare against len
      for (;;) {
        const uint32_t want = scan_lo & ~63u;
        if (want >= len) break;
        if (!hvalid || hb != want) { hb = want; hwin = hb + (uint32_t)lane < len ? res[hb + lane] : 1ull; hvalid = true; }
        unsigned long long m = __ballot(!(hwin & 1ull) && hb + (uint32_t)lane < len);
        if (m) {
          const int l0 = __builtin_ctzll(m);
          ci = (int)hb + l0; kci = readlane_u64(hwin, l0);
          m &= m - 1;
          if (m) { const int l1 = __builtin_ctzll(m); cj = (int)hb + l1; kcj = readlane_u64(hwin, l1); }
          break;
        }
        scan_lo = hb + 64u;   // every member of this chunk is expanded
      }
    } else {
      for (uint32_t base = scan_lo & ~63u; base < len; base += 64) {
        const uint32_t i = base + lane;
        const unsigned long long e = i < len ? res[i] : 1ull;
        unsigned long long m = __ballot(!(e & 1ull));
        if (m) {
          const int l0 = __builtin_ctzll(m);
          ci = (int)base + l0; kci = readlane_u64(e, l0);
          m &= m - 1;
          if (m) { const int l1 = __builtin_ctzll(m); cj = (int)base + l1; kcj = readlane_u64(e, l1); }
          break;
        }
      }
    }
    unsigned long long kd = ~0ull; int dlane = -1;
    unsigned long long d_un = 0ull;   // the delta's unexpanded members (ballot; ascending lanes = ascending keys)
    if constexpr (DELTA) {
      d_un = dl.unexpanded(lane);
      if (d_un) { dlane = __builtin_ctzll(d_un); kd = dl.key_at(dlane); }
    }
    if (ci < 0 && dlane < 0) break;
    if constexpr (VISMODE == VIS_LDS) {
      if (vis_count + 64 > (w.hcap >> 2) * 3) { w.err |= 8u; break; }   // the table would need the reset path: give up, the host re-runs the call on the one-wave kernel
    }
    const bool from_delta = key_lt(kd, kci);
    const unsigned long long ce = from_delta ? kd : kci;
    unsigned long long runner_key = ~0ull;
    if constexpr (PREF) {
      if (from_delta) {
        const unsigned long long u2 = d_un & (d_un - 1ull);   // the delta's second unexpanded member
        const unsigned long long kd2 = u2 ? dl.key_at(__builtin_ctzll(u2)) : ~0ull;
        runner_key = key_min(kci, kd2);
      } else {
        runner_key = key_min(kcj, kd);
      }
    }
    // evaluators that fetch the chunk's adjacency rows with its vectors (hnsw_lat.hpp): the only adjacency row that can be
    // needed next and is not on chip then is the runner-up's — requested now, it flies during the whole expansion
    uint32_t runner_nb = NBR_NONE;
    if constexpr (eval_t::RADJ && PREF) {
      if (runner_key != ~0ull && (uint32_t)p < g.mMax0) runner_nb = g.adj0[(size_t)((uint32_t)runner_key >> 1) * g.mMax0 + p];
      // ... and with it the evaluator's (candidate, position)-addressed inputs of the runner-up's neighbours (AdcEval<.., NBR>: their code rows): a whole
      // expansion of lead time instead of the pop's
      if constexpr (eval_t::ROWPF) { if (runner_key != ~0ull && g.mMax0 <= 32) ev.prefetch_spec((uint32_t)runner_key >> 1, (uint32_t)p, (uint32_t)p < g.mMax0, half); }
    }
    int best_src = -1;   // lane of the smallest key admitted in this expansion's (single) chunk
    COLTT_PT(w, 0)  // pop
    // lowerBound: the distance of the largest member, sampled once per pop (hnsw.go:357)
    unsigned long long worst;
    if constexpr (CACHE) worst = twin.last(res, len, lane);
    else worst = len ? (res[len - 1] & ~1ull) : 0ull;
    if constexpr (DELTA) { const unsigned long long dmx = dl.max_key(); worst = key_lt(worst, dmx) ? dmx : worst; }
    const float lower_bound = __uint_as_float((uint32_t)(worst >> 32));
    wave_sync();
    if (from_delta) { if (lane == dlane) dl.lo |= 1u; }
    else {
      if (lane == 0) res[ci] = ce | 1ull;
      if constexpr (CACHE) { if ((uint32_t)lane == (uint32_t)ci - hb) hwin |= 1ull; }
      scan_lo = (uint32_t)ci + 1;
    }
    const uint32_t cslot = (uint32_t)ce >> 1;
    uint32_t free_slots = ef - (len + dl.n);  // the set never exceeds ef
    // BOUNDED evaluators (the walk over product-quantiser codes — a definition of ours, coltt_oracle.cpp: csr_search_pq): once the result set is full at a pop
    // it stays full and its worst member only ever improves, so a neighbour whose table distance is not below lowerBound can never be admitted, now or later:
    // it is neither marked visited nor counted (its distance costs a table sum, not a row).  The sequence of result sets — hence ids, scores, n_exp — is
    // exactly that of the unbounded walk; n_dist counts the evaluations that passed the bound and were fresh.
    // What this buys is the visited MARKS: a byte store into the map is a read-modify-write in HBM, and 22 of them per expansion became ~3.4 — 412.6 -> 467.8 k
    // queries/s at 10 M x 768, ef 1 344 (profiles/r06o_*).
    // -DCOLTT_PQ_BYSET=1 (A/B knob, exact, measured SLOWER, off): THE RESULT SET AS THE VISITED SET.  A vertex under the bound that was met before was admitted (it
    // was under the bound then, too: the bound only falls) and is still a member — had it been evicted, it would be above the bound now; the one exception are the
    // neighbours REJECTED by the expansion that filled the set (rej_slot, see set_member).  So in a full set "visited" == "is a current member", and the byte map
    // need not be read or written again.  But the membership test — a binary search of 11 dependent LDS reads + the delta's registers — costs more than the probe
    // it replaces, which flies under the table sums: 467.8 -> 403.9 k queries/s, one query 2.75 -> 3.66 ms (profiles/r06s_*).
    // (Rows wider than one chunk keep the byte map in any case: the set changes between the chunks of one expansion, the oracle's visited set does not.)
    const bool full_at_pop = free_slots == 0;
    const bool by_set = COLTT_PQ_BYSET != 0 && full_at_pop && width <= 32;
    w.n_exp++;
    wave_sync();
    const uint32_t* row = g.adj0 + (size_t)cslot * width;
    const float* nrow = ADJN ? g.adj0_n + (size_t)cslot * width : nullptr;
    const bool use_pre = pre_slot == cslot;
    const uint32_t pre_now = pre_nb; const float pre_nn_now = pre_nn;
    const uint32_t spec_now = spec_slot, spec_vis_now = spec_vis;
    pre_slot = NBR_NONE; spec_slot = NBR_NONE;
    unsigned long long best_new = ~0ull;
#define COLTT_PREFETCH_NEXT2()                                                                       \
    if constexpr (PREF) {                                                                            \
      const unsigned long long nk_ = key_min(runner_key, best_new);                                  \
      if (nk_ != ~0ull) {                                                                            \
        pre_slot = (uint32_t)nk_ >> 1;                                                               \
        if (eval_t::RADJ && width <= 32 && runner_key < best_new) pre_nb = runner_nb;                \
        else if (eval_t::CHUNK_ADJ && width <= 32 && best_src >= 0) pre_nb = (uint32_t)p < width ? ev.chunk_adj(best_src >> 1, p) : NBR_NONE; \
        else pre_nb = (uint32_t)p < width ? g.adj0[(size_t)pre_slot * width + p] : NBR_NONE;        \
        if constexpr (ADJN) pre_nn = (uint32_t)p < width ? g.adj0_n[(size_t)pre_slot * width + p] : 0.f; \
        if constexpr (eval_t::ROWPF) {                                                                 \
          if (eval_t::RADJ && width <= 32 && runner_key < best_new) ev.take_spec();   /* requested at pop time: the runner-up it is */ \
          else ev.prefetch_at(pre_slot, (uint32_t)p, (uint32_t)p < width, half);       /* the next candidate's neighbours' inputs fly with its adjacency row */ \
        }                                                                                              \
        if constexpr (eval_t::SPEC) {                                                                \
          spec_slot = NBR_NONE;                                                                      \
          if (width <= 32 && runner_key < best_new) {   /* pre_nb is in registers (requested at pop time) */ \
            spec_slot = pre_slot;                                                                    \
            const bool sv_ = pre_nb != NBR_NONE;                                                     \
            if constexpr (VISMODE == VIS_HBM) { spec_vis = 0; if (sv_ && half == 0) spec_vis = __hip_atomic_load(w.visg + pre_nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } \
            ev.prefetch(pre_nb, sv_, half);                                                          \
          }                                                                                          \
        }                                                                                            \
      }                                                                                              \
    }
    for (uint32_t c0 = 0; c0 < width; c0 += 32) {
      float bounded_d = 0.f;   // BOUNDED, not EARLY: the chunk's table distances, computed before the visited test
      (void)bounded_d;
      const uint32_t idx = c0 + p;
      const bool pre_hit = use_pre && c0 == 0;
      const uint32_t nb = idx < width ? (pre_hit ? pre_now : row[idx]) : NBR_NONE;
      float nrm = 0.f;
      if constexpr (ADJN) nrm = idx < width ? (pre_hit ? pre_nn_now : nrow[idx]) : 0.f;
      const bool valid = nb != NBR_NONE && !is_deleted(g, nb);
      bool spec_hit = false;
      if constexpr (eval_t::SPEC) spec_hit = pre_hit && spec_now == cslot;   // this expansion's inputs were requested during the previous one
      if constexpr (eval_t::ROWPF) { if (!pre_hit) ev.prefetch_at(cslot, idx, idx < width, half); }   // (requested with the adjacency row when that was prefetched)
      else { if (!spec_hit) ev.prefetch(nb, valid, half); }   // evaluators whose per-neighbour input is small (hnsw_pq.hpp: a 32-128 byte code row) request it NOW, under the visited test
#ifdef COLTT_PHASE_TIMING
      if (__ballot(valid) == 0xdeadbeefcafeull) w.err |= 64u;  // forces the adjacency values to have arrived
#endif
      COLTT_PT(w, 1)  // adjacency row
      int fresh_i = 0;
      bool split_done = false; float split_d = 0.f;   // SPLIT evaluators: this pair's row went through the pass in front of the probe's answer
      if constexpr (eval_t::EARLY && VISMODE == VIS_HBM && !BLOOM) {
        // The probe of the byte map is ISSUED, the evaluator computes the distances of all listed neighbours out of inputs that are already on chip
        // (AdcEval<.., NBR>: the code rows came with the adjacency row), and only then is the probe's answer looked at: its round trip runs under
        // ~200 issue slots of table lookups instead of in front of them.  Test-and-set as below.
        static_assert(!eval_t::SPEC, "EARLY evaluators take no speculative visited bytes");
        const bool probe = valid && half == 0;
        if (eval_t::BOUNDED && by_set) {
          // full set: the table sums, the bound, then membership in the result set (no probe, no mark)
          ev.early(valid, half);
          const bool want = probe && ev.pre_d < lower_bound;
          if (__ballot(want)) {   // (every lane calls: the delta's keys sit one per lane)
            const bool member = set_member(res, len, dl, want, __float_as_uint(ev.pre_d), nb << 1, rej_slot, lane);
            fresh_i = (want && !member) ? 1 : 0;
          }
        } else {
        // The aligned 32-bit word around the byte is loaded (no zero-extension for the compiler to place — with its s_waitcnt vmcnt — right behind the
        // load) and the byte is taken out of it after early().  The region is this wave's own and its size a multiple of 16; agent scope: served by L2.
        // EVERY lane loads (the others word 0 of the region: one more address in the same request): a load under a divergent branch would leave the
        // compiler two paths with different numbers of loads in flight, and it then waits for all of them (vmcnt(0)) in front of the sums.
        uint32_t vw = vis_probe(reinterpret_cast<const uint32_t*>(w.visg + (probe ? (nb & ~3u) : 0u)));
        ev.early(valid, half);
        ev.after_early(vw);   // the probe's value is not looked at (no s_waitcnt vmcnt for it) before the table sums are there
        const uint32_t v = (vw >> ((nb & 3u) * 8u)) & 0xffu;
        fresh_i = probe && v != (w.epoch & 0xffu) ? 1 : 0;
        if constexpr (eval_t::BOUNDED) { if (full_at_pop && !(ev.pre_d < lower_bound)) fresh_i = 0; }   // (wide rows) the bound: not marked, not counted
        if (fresh_i) __hip_atomic_store(w.visg + nb, (uint8_t)w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else if constexpr (eval_t::SPLIT && BLOOM && VISMODE == VIS_HBM && !eval_t::EARLY && !eval_t::BOUNDED && !eval_t::SPEC) {
        // Bloom filter, then the byte-map probe of the positives ISSUED and the first full pass of definitely-fresh rows evaluated under it (see COLTT_G8_SPLIT)
        const bool want = valid && half == 0;
        bool maybe = false;
        if (want) {
          const uint32_t h = nb * 0x9E3779B1u;
          const uint32_t bits = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
          const uint32_t old = atomicOr(&w.bloom[h >> w.bloom_shift], bits);
          maybe = (old & bits) == bits;
        }
        const bool sure = want && !maybe;
        const unsigned long long S = __ballot(sure);
        // every lane loads (the others byte 0 of the region: one more address in the same request) — no divergent branch around a load in flight
        const uint8_t pv = vis_probe(w.visg + (maybe ? nb : 0u));
        if ((uint32_t)__popcll(S) >= (uint32_t)eval_t::PASS_ROWS && __ballot(maybe)) {   // wave-uniform: a full pass of negatives and a probe to hide
          const uint32_t srank = __builtin_amdgcn_mbcnt_hi((uint32_t)(S >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)S, 0u));
          int first_i = (sure && srank < (uint32_t)eval_t::PASS_ROWS) ? 1 : 0;
          first_i = __builtin_amdgcn_mov_dpp(first_i, 0xA0, 0xf, 0xf, true);   // to the odd lane of the pair
          split_done = first_i != 0;
          split_d = ev(g, w, nb, split_done, nrm, half, lane);
        }
        fresh_i = (sure || (maybe && pv != (uint8_t)w.epoch)) ? 1 : 0;
        if (fresh_i) __hip_atomic_store(w.visg + nb, (uint8_t)w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        if constexpr (eval_t::EARLY) ev.early(valid, half);
        bool want = valid && half == 0;
        if constexpr (eval_t::BOUNDED) {   // full set (see full_at_pop): the table sums of ALL listed neighbours, the bound, then membership in the result set
          if (full_at_pop) {
            bounded_d = ev.eval_now(valid, half); want = want && bounded_d < lower_bound;
            if (by_set && __ballot(want)) {   // (every lane calls: the delta's keys sit one per lane)
              const bool member = set_member(res, len, dl, want, __float_as_uint(bounded_d), nb << 1, rej_slot, lane);
              fresh_i = (want && !member) ? 1 : 0;
            }
          }
        }
        if (want && !(eval_t::BOUNDED && by_set)) {
          // Test-and-set.  No two lanes hold the same slot (a row lists a neighbour once), so load + store on the byte map is
          // race-free; agent-scope atomics are served by L2, never by a stale L1 line.
          bool maybe = true;
          if constexpr (BLOOM) {
            const uint32_t h = nb * 0x9E3779B1u;
            const uint32_t bits = (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
            const uint32_t old = atomicOr(&w.bloom[h >> w.bloom_shift], bits);
            maybe = (old & bits) == bits;
          }
          if constexpr (VISMODE == VIS_LDS) fresh_i = vis_insert(w.vis, w.hcap_mask, nb) ? 1 : 0;
          else {
            if (maybe) {
              const uint8_t v = spec_hit ? (uint8_t)spec_vis_now : vis_probe(w.visg + nb);
              fresh_i = v != (uint8_t)w.epoch ? 1 : 0;
            } else fresh_i = 1;
            if (fresh_i) __hip_atomic_store(w.visg + nb, (uint8_t)w.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      fresh_i = __builtin_amdgcn_mov_dpp(fresh_i, 0xA0, 0xf, 0xf, true);  // even lane's verdict to its pair: quad_perm [0,0,2,2]
      const bool fresh = fresh_i != 0;
      const unsigned long long E = __ballot(fresh && half == 0);
      const uint32_t nfresh = __popcll(E);
      COLTT_PT(w, 2)  // visited test-and-set
      const bool last_chunk = c0 + 32 >= width;
      if (nfresh == 0) { if (last_chunk) { COLTT_PREFETCH_NEXT2() } continue; }
      w.n_dist += nfresh; vis_count += (eval_t::BOUNDED && by_set) ? 0u : nfresh;   // (vis_count: entries of the LDS hash)
      float d;
      if constexpr (eval_t::BOUNDED && !eval_t::EARLY) { if (full_at_pop) d = fresh ? bounded_d : 0.f; else d = ev(g, w, nb, fresh, nrm, half, lane); }
      else if constexpr (eval_t::SPLIT) {
        const bool rest = fresh && !split_done;
        d = split_d;
        if (__ballot(rest)) { const float d2 = ev(g, w, nb, rest, nrm, half, lane); d = split_done ? split_d : d2; }
      }
      else d = ev(g, w, nb, fresh, nrm, half, lane);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(E >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E, 0u));   // fresh neighbours in front of this lane
      const bool adm = fresh && half == 0 && (rank < free_slots || d < lower_bound);
      if constexpr (eval_t::BOUNDED) { if (!full_at_pop && fresh && half == 0 && !adm) rej_slot = nb; }   // only the expansion that fills the set rejects while filling
#ifdef COLTT_PHASE_TIMING
      if (__ballot(adm) == 0xdeadbeefcafeull) w.err |= 64u;  // forces the distances
#endif
      COLTT_PT(w, 3)  // row reads + distances
      free_slots = free_slots > nfresh ? free_slots - nfresh : 0;
      const unsigned long long A = __ballot(adm);
      const uint32_t m = __popcll(A);
      const uint32_t khi = __float_as_uint(d), klo = nb << 1;
      const unsigned long long mykey = adm ? (((unsigned long long)khi << 32) | klo) : ~0ull;
      if constexpr (DELTA) {
        if (m) {
          if constexpr (PREF) {   // the smallest admitted key: a scalar minimum over a handful of lanes, a DPP arg-min when the set is still filling up
            unsigned long long mn; int bl;
            if (m <= 4u) {
              mn = ~0ull; bl = -1;
              unsigned long long am = A;
              while (am) {
                const int j = __builtin_ctzll(am); am &= am - 1;
                const unsigned long long kj = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)khi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)klo, j);
                if (key_lt(kj, mn)) { mn = kj; bl = j; }
              }
            } else bl = wave_argmin_key(adm, khi, klo, mn);
            if (key_lt(mn, best_new)) { best_new = mn; best_src = bl; }
          }
        }
        if (last_chunk) { COLTT_PREFETCH_NEXT2() }
        if (m == 0) continue;
        if (dl.n + m > 64u) { delta_flush(res, len, dl, scan_lo, lane); hvalid = false; twin.valid = false; }
        {
          unsigned long long am = A;
          while (am) {
            const int j = __builtin_ctzll(am); am &= am - 1;
            dl.insert((uint32_t)__builtin_amdgcn_readlane((int)khi, j), (uint32_t)__builtin_amdgcn_readlane((int)klo, j), lane);
          }
        }
        const uint32_t total = len + dl.n;
        if (total > ef) { if constexpr (CACHE) evict_largest(res, len, dl, total - ef, lane, &twin); else evict_largest(res, len, dl, total - ef, lane); }
        COLTT_PT(w, 4)  // admission + eviction (+ the occasional flush)
      } else {
        uint32_t myrank = 0;  // rank of my key among the admitted ones (readlane broadcasts: no LDS round trips)
        {
          unsigned long long am = A;
          while (am) {
            const int j = __builtin_ctzll(am); am &= am - 1;
            const unsigned long long kj = readlane_u64(mykey, j);
            myrank += (kj < mykey) ? 1u : 0u;
          }
        }
        if (m) {  // the smallest admitted key is the one of rank 0
          const unsigned long long z = __ballot(adm && myrank == 0);
          const unsigned long long mn = readlane_u64(mykey, __builtin_ctzll(z));
          if (mn < best_new) { best_new = mn; best_src = (int)__builtin_ctzll(z); }
        }
        if (last_chunk) { COLTT_PREFETCH_NEXT2() }
        if (m == 0) continue;
        if (w.ef_pad <= 128u) {
          // Small result sets (two entries per lane) — the headline configuration: ONE LDS round trip fetches the whole set, one
          // uniform loop over the admitted keys gives every key its position (members in front of it, counted by ballots) and
          // every member its shift (admitted keys in front of it): no binary search (7 dependent LDS reads), no second loop.
          const unsigned long long e0 = (uint32_t)lane < len ? res[lane] : ~0ull;
          const unsigned long long e1 = 64u + (uint32_t)lane < len ? res[64 + lane] : ~0ull;
          uint32_t sh0 = 0, sh1 = 0, spos = 0;
          {
            unsigned long long am = A;
            while (am) {
              const int j = __builtin_ctzll(am); am &= am - 1;
              const unsigned long long kj = readlane_u64(mykey, j);
              const bool lt0 = e0 < kj, lt1 = e1 < kj;   // members in front of admitted key j (a fresh vertex: never equal to a member)
              const uint32_t below = (uint32_t)__popcll(__ballot(lt0)) + (uint32_t)__popcll(__ballot(lt1));
              if (lane == j) spos = below;
              sh0 += lt0 ? 0u : 1u; sh1 += lt1 ? 0u : 1u;
            }
          }
          const uint32_t minpos = (uint32_t)__builtin_amdgcn_readlane((int)spos, __builtin_ctzll(__ballot(adm && myrank == 0)));
          scan_lo = minpos < scan_lo ? minpos : scan_lo;
          { const uint32_t np = (uint32_t)lane + sh0; if ((uint32_t)lane < len && sh0 && np < ef) res[np] = e0; }
          { const uint32_t np = 64u + (uint32_t)lane + sh1; if (64u + (uint32_t)lane < len && sh1 && np < ef) res[np] = e1; }
          { const uint32_t np = spos + myrank; if (adm && np < ef) res[np] = mykey; }
          len = len + m < ef ? len + m : ef;
          wave_sync();
          COLTT_PT(w, 4)  // merge
          continue;
        }
        // in-place merge from the tail down to the chunk of the smallest new key (hnsw_dev.hpp:search_level)
        uint32_t mypos = 0xffffffffu;
        if (adm) {
          uint32_t lo = 0, hi = len;
          while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (res[mid] < mykey) lo = mid + 1; else hi = mid; }
          mypos = lo;
        }
        const uint32_t minpos = (uint32_t)__builtin_amdgcn_readlane((int)mypos, __builtin_ctzll(__ballot(adm && myrank == 0)));
        scan_lo = minpos < scan_lo ? minpos : scan_lo;
        wave_sync();
        if (len) {
          for (int base = (int)((len - 1) & ~63u); base >= (int)(minpos & ~63u); base -= 64) {
            const uint32_t i = (uint32_t)base + lane;
            const unsigned long long e = i < len ? res[i] : ~0ull;
            uint32_t shift = 0;
            unsigned long long am = A;
            while (am) {
              const int j = __builtin_ctzll(am); am &= am - 1;
              const uint32_t pj = (uint32_t)__builtin_amdgcn_readlane((int)mypos, j);
              shift += (pj <= i) ? 1u : 0u;
            }
            wave_sync();
            const uint32_t np = i + shift;
            if (i < len && shift && np < ef) res[np] = e;
          }
        }
        wave_sync();
        { const uint32_t np = mypos + myrank; if (adm && np < ef) res[np] = mykey; }
        len = len + m < ef ? len + m : ef;
        wave_sync();
        COLTT_PT(w, 4)  // merge
      }
    }
  }
#undef COLTT_PREFETCH_NEXT2
  if constexpr (DELTA) delta_flush(res, len, dl, scan_lo, lane);
  out_len = len;
}

}  // namespace dev
}  // namespace coltt
