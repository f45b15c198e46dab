// hnsw_pq.hpp — Hnsw.Search over product-quantiser codes with an exact re-rank (VERDICT r4 missing #1).
//
// The reference's only PQ call shape is a product-quantised HNSW (playground/hnswpq_verification.go:69-105:
// hnswpq.NewProductQuantizationHnsw(), m = 32, 256 centroids, train, Fit, search on the codes; UPDATE-LOG.md:190-194) whose package
// (pkg/hnswpq) is not in its tree.  What follows is therefore a DEFINITION, stated in oracle/coltt_oracle.cpp ("Product-quantised HNSW")
// and restated here, built from pieces that ARE pinned: the graph and the traversal of core/vectorindex/hnsw.go:243-278, 320-389 (canonical
// closed form, hnsw_walk2.hpp) and the quantiser of pq.hip (distancepq arithmetic, pkg/distancepq/distance.go:30-42):
//   codes     one row-major code per stored row: Encode(decode(stored row))          (pq.hip: pq_encode_kernel)
//   table     lut[j][c] = binary16(distFn(q_j, centroid[j][c])) over the query the index's distance sees (normalised / lowered): the
//             quantiser's table entry (pq.hip) ROUNDED TO BINARY16 (round to nearest even, compresshelper.Fromfloat32's rounding,
//             float16.go:276-321) — d only ranks candidates, the answers carry exact distances, and half the table bytes are twice the
//             resident traversals (the table is what bounds this kernel's occupancy, see below)
//   distance  d(q, v) = S_lo + S_hi (one f32 add), S_lo = the f32 sum, in j order from +0.0, of float32(lut[j][code_v[j]]) over j < JS, S_hi the same over
//             JS <= j < m; JS = 16 * ceil(P / 2), P = ceil(m / 16) the row's 16-byte pieces (round 6: each lane of the pair that owns a neighbour sums
//             half the code row — rounds 4-5 summed all of it in j order in ONE lane while its partner idled; a definition either way)
//   walk      Hnsw.Search with d in place of Distance(): entrypoint, greedyClosestNeighbor on the upper levels, searchLevel(ef) on
//             level 0 — same admission rule, same canonical neighbour order, ties by (d bits, slot)
//   re-rank   the r = min(max(rerank, k), len) nearest survivors by d (rerank = 0: all of them) get the index's EXACT distance
//             (reference summation order, exact.hpp); the k smallest by (exact score bits, slot) are returned with the exact scores.
// d is a sum of non-negative terms for the supported pairings (squared L2 always; 1 - dot on a cosine index, whose rows, query and
// centroid pieces have norm <= 1), so its f32 bits order as unsigned integers — the key order of the walk.
//
// One wave per query.  LDS: [the query's table | result set | visited hash (small ef; above it the byte map in HBM, no Bloom filter)].  The walk writes its survivors to HBM; the exact re-rank is
// two kernels of its own (hnsw_kernels.hpp: hnsw_pq_rerank_kernel — one wave per 32 survivors, the HBM-visited walks' burst profile — and hnsw_pq_select_kernel).
// The table is what bounds occupancy: mp16 x C' x 2 bytes per resident traversal, C' = the centroid count rounded up to a power of two
// (m = 32 x 256 centroids: 16 KiB; m = 96 x 256: 48 KiB; m = 64 x 16 — the same 256 bits per row as 32 x 256 — 2 KiB).  Measured with f32 tables
// (profiles/r05b_hnswpq_probe_10m.jsonl, 10 M x 768 f16): the walk is a chain of dependent round trips (~5 us per expansion), its
// throughput is resident traversals / latency — 157 k queries/s at 3 waves per CU (m = 32), 49 k at 1 (m = 96) — never HBM bytes.
#pragma once
#include "hnsw_walk2.hpp"

namespace coltt {
namespace dev {

// Where the walk's distances come from: the table in LDS, the code row of the neighbour (16-byte pieces, all requested before the
// first lookup).  The even lane of a pair computes, both lanes of the pair receive (the walk keeps one neighbour per lane pair).
// LS: the table's row length as a compile-time constant (log2; 0 = read lut_shift at run time).  With LS known and the table at the START of the
// workgroup's LDS, lookup j is `ds_read_u16 v, (code byte << 1) offset: j << (LS + 1)`: no per-lookup address arithmetic beyond the byte extraction
// (the run-time form keeps 128 row bases, which the compiler parks in VGPR lanes and fetches back one v_readlane + hazard nop at a time).
// NP: 16-byte pieces per code row as a compile-time constant (0 = read row_bytes at run time): the row's registers and the lookups are then straight-line
// code — the run-time form tests `piece < row_bytes / 16` in front of every piece and carries all 8 x 4 row registers through every loop of the walk
// (32 v_mov_b64 per expansion at the loop's back edge, round 5's ISA).
// NBR: the walk's code rows come from the NEIGHBOURHOOD BLOCKS (round 6) — nbrc[slot][p][row_bytes] = the code row of slot's p-th level-0 neighbour, the
// same bytes as codes[adj0[slot][p]] laid beside the adjacency row's order — addressed by (candidate, position) instead of by the neighbour's slot:
// they are requested TOGETHER with the candidate's adjacency row at the end of the previous expansion (search_level2: ROWPF), one contiguous
// mMax0 x row_bytes block (2 KiB for 32 x 64) instead of 32 gathers of 64 bytes (each a whole 128-byte line of HBM traffic) behind the adjacency row's
// own round trip; the table sums of ALL listed neighbours then run under the visited probe's round trip (EARLY) instead of behind it.  One dependent
// round trip per expansion (the probe) instead of three.  Derived data like adj0_n (hnsw.hip: sync_pq_nbr); the upper levels gather from `codes`.
// Round 6 — BOTH lanes of a pair work: the even lane owns the row's first PH = ceil(P / 2) pieces (table rows j < JS = 16 PH), the odd lane the rest; each sums its
// half in j order from +0.0 and the pair adds the two partial sums (one DPP swap + one add: the definition above).  Half the issue slots per expansion for the
// table sums, half the row registers.  So that ONE instruction stream serves both lanes, the table sits in LDS PAIR-INTERLEAVED — row j of the table at LDS row
// 2 (j mod JS) + (j div JS) — and lookup t of a lane addresses `t * 2R + (code + h * C') * 2` (R = bytes per table row, C' = its entries, h = the lane's half): the
// immediate offset is shared, and the lane's half rides in the CODE BYTE — one packed add of h * C' to every dword of the row when C' <= 128 (no carries: a code
// is < C'), one v_add per lookup for 256-entry rows.  For an odd P the odd lane owns one piece less and the interleaved table has unused rows (one piece's worth).
__host__ __device__ __forceinline__ constexpr uint32_t pq_walk_table_rows(uint32_t pieces) { return 2u * ((pieces + 1u) / 2u) * 16u; }   // LDS rows of the interleaved table
template <int LS = 0, int NP = 0, bool NBR = false> struct AdcEval {
  const uint8_t* codes; uint32_t row_bytes;   // [n][row_bytes], row_bytes = mp16 (a multiple of 16, <= 128); bytes j >= m are 0
  const unsigned short* lut;                   // LDS: [pq_walk_table_rows][1 << lut_shift] binary16, pair-interleaved; rows j >= m are +0.0 (d + 0.0 keeps d's bits: d is never -0)
  uint32_t lut_shift;                          // log2 of the table's row length = the number of centroids rounded up to a power of two (4 .. 8)
  const uint8_t* nbrc; uint32_t nbr_stride;    // NBR: [n][mMax0][row_bytes], nbr_stride = mMax0 * row_bytes
  uint32_t hsel;                               // this lane's half of its pair: 0 = the row's first PH pieces, 1 = the rest
  static constexpr int NR = NP ? (NP + 1) / 2 : 4;   // row registers (16-byte pieces) of ONE lane
  static constexpr bool BIAS = LS != 0 && LS <= 7;   // the lane's half rides in the code bytes (packed add per dword); LS 8: one add per lookup
  __device__ __forceinline__ int pieces() const { return NP ? NP : (int)(row_bytes >> 4); }
  __device__ __forceinline__ int first_piece() const { return hsel ? (pieces() + 1) >> 1 : 0; }
  __device__ __forceinline__ int my_pieces() const { const int np = pieces(), ph = (np + 1) >> 1; return hsel ? np - ph : ph; }
  // the pair's total: both lanes end up with S_lo + S_hi (f32 addition commutes bit for bit)
  static __device__ __forceinline__ float pair_total(float part) {
    const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, part), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    return part + other;
  }
  // s + float32(h), h = a binary16 in the low half of a register.  v_fma_mix_f32 computes fma(float32(h), 1.0f, s) with ONE rounding; float32(h) * 1.0f
  // is exact, so the result is the IEEE sum of the converted entry — the bits of v_cvt_f32_f16 + v_add_f32 (the definition) in one issue slot.
  static __device__ __forceinline__ float acc(float s, uint32_t h) {
    float r; const float one = 1.0f;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(one), "v"(s));
    return r;
  }
  // byte B of v, times two (the byte offset of a binary16 entry in its table row): one SDWA shift instead of v_bfe + v_lshl
  template <int B> static __device__ __forceinline__ uint32_t byte2(uint32_t v) {
    uint32_t r; const uint32_t one = 1u;
    if constexpr (B == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(one), "v"(v));
    else if constexpr (B == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(one), "v"(v));
    else if constexpr (B == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(one), "v"(v));
    else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(one), "v"(v));
    return r;
  }
  // table entry (row j, code byte B of v), as the 16 bits the LDS read returns.  LS != 0: the table starts at LDS address 0 (the kernel checks it), so
  // the read is addressed by the shifted code byte alone, the row in the instruction's offset field — through the generic pointer the compiler adds
  // the (zero) base symbol to every address: one v_add per lookup.
  // jl = the lookup's index within this lane's half; the LDS row is 2 jl + hsel (pair-interleaved table)
  template <int B> __device__ __forceinline__ uint32_t entry(uint32_t jl, uint32_t sh, uint32_t v) const {
    return *reinterpret_cast<const unsigned short*>(reinterpret_cast<const uint8_t*>(lut) + ((size_t)(2u * jl + hsel) << (sh + 1)) + byte2<B>(v));
  }
  u32x4e raw[NR];                              // this lane's half of the code row requested by prefetch() / prefetch_at() for the pair's neighbour
  static constexpr bool CHUNK_ADJ = false;
  // hnsw_walk2.hpp SPEC: visited bytes + code rows of the predicted next candidate's (the runner-up's) neighbours requested one expansion ahead.
  // Exact (tests + 246 randomised rounds with it on), but measured SLOWER — an A/B knob (-DCOLTT_PQ_SPEC=1), off in the shipped library
  // (profiles/r05m_pq_spec_ab.md)
#ifndef COLTT_PQ_SPEC
#define COLTT_PQ_SPEC 0
#endif
  static constexpr bool SPEC = COLTT_PQ_SPEC != 0 && !NBR;
  // hnsw_walk2.hpp RADJ: the runner-up's adjacency row requested at pop time (it is the next candidate unless this expansion admits a nearer
  // vertex).  On its own, without the speculation above: 1 % slower (profiles/r05s_pq_ab.md) — the exact prefetch at the end of the expansion already
  // flies under the admission and the next pop.
  // Round 6 tried the same over the neighbourhood blocks (-DCOLTT_PQ_RADJ=1: the runner-up's adjacency row AND its 2 KiB block of code rows requested at pop
  // time into a second set of row registers, taken over when the runner-up is indeed the next candidate): 2.6 % SLOWER in throughput (344.4 against
  // 353.5 k queries/s on one box, profiles/r06d_pq_radj_ab.md) and no faster for one query alone — 32 more VGPRs and 2 KiB of wasted fetch per misprediction
  // against a prefetch that the admission, the eviction and the next pop already cover.  Off in the shipped library.
#ifndef COLTT_PQ_RADJ
#define COLTT_PQ_RADJ 0
#endif
  static constexpr bool RADJ = SPEC || (NBR && COLTT_PQ_RADJ != 0);
  // A/B knob: the walk keeps the head / tail windows of its result set's main array in registers (hnsw_walk2.hpp: SETCACHE) — three LDS round trips
  // fewer per expansion, exact (154 GPU tests + 460 randomised rounds with it on), and NOT faster: 403.1 against 408.9 k queries/s (call I; separate
  // processes on one box differ by +-5 % on identical code, so "no gain" is all that can be said).  Off.
#ifndef COLTT_PQ_SETCACHE
#define COLTT_PQ_SETCACHE 0
#endif
  static constexpr bool SETCACHE = COLTT_PQ_SETCACHE != 0;
  static constexpr bool SPLIT = false;
  static constexpr bool BOUNDED = true;   // hnsw_walk2.hpp: once the set is full, a neighbour whose table distance is not below lowerBound is neither marked nor counted,
                                          // and the result set itself answers "visited?" (no byte-map probe, no mark)
  static constexpr bool ROWPF = NBR;   // per-neighbour inputs addressed by (candidate, position): requested with the candidate's adjacency row
  static constexpr bool EARLY = NBR;   // the distances of all listed neighbours are computed under the visited probe (early())
  __device__ __forceinline__ uint32_t chunk_adj(int, int) const { return NBR_NONE; }
#ifndef COLTT_PQ_NT   // A/B knob: non-temporal hint on the code rows / neighbourhood blocks — measured SLOWER (475.6 -> 440.0 k queries/s at ef 1 344, GPU call AG: hub vertices' blocks are re-read by other traversals out of L2 / MALL), off
#define COLTT_PQ_NT 0
#endif
  static __device__ __forceinline__ u32x4e pq_ld(const u32x4e* p) {
#if COLTT_PQ_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
  }
  __device__ __forceinline__ void load_from(const uint8_t* row, u32x4e (&r)[NR]) const {
    const u32x4e* p = reinterpret_cast<const u32x4e*>(row) + first_piece();
    if constexpr (NP != 0 && NP % 2 == 0) {
#pragma unroll
      for (int i = 0; i < NR; i++) r[i] = pq_ld(p + i);
    } else {
      const int cnt = my_pieces();
#pragma unroll
      for (int i = 0; i < NR; i++) if (i < cnt) r[i] = pq_ld(p + i);
    }
  }
  __device__ __forceinline__ void load(uint32_t slot, u32x4e (&r)[NR]) const { load_from(codes + (size_t)slot * row_bytes, r); }
  // Eight lookups at a time, as ONE instruction block (LS known): the eight byte extractions, then the eight LDS reads, then the eight adds — each add
  // behind `s_waitcnt lgkmcnt(7 - t)`, i.e. as soon as ITS read is back (LDS returns in order).  The adds are one dependent chain (the definition sums in
  // j order), so what can overlap is the reads' latency: left to the scheduler, hipcc 7.2 emits read / wait lgkmcnt(0) / add per lookup (64 x ~64 cycles of
  // exposed LDS latency per expansion; three reads in flight at best in round 5's form).  The block only ever lowers the outstanding-LDS count it raised
  // itself, so the compiler's own wait counts around it stay conservative-correct.
#ifndef COLTT_PQ_SUM_WAITS   // A/B knob: s_waitcnt instructions per block of eight lookups (8: one in front of every add; 2: one per four adds — measured the same to 0.3 %, r06d)
#define COLTT_PQ_SUM_WAITS 8
#endif
  // J = the first lookup's index within the lane's half; kadd: the lane's byte offset into the pair of interleaved rows (ADDK form: 256-entry rows, the half does
  // not fit beside the code in a byte) — 0 in the BIAS form, where v0 / v1 already carry code + h * C' in every byte
  template <int J> static __device__ __forceinline__ float sum8(float s, uint32_t v0, uint32_t v1, uint32_t kadd) {
    static_assert(LS != 0, "sum8 needs the table's row length at compile time");
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
    const uint32_t one = 1u; const float onef = 1.0f;
    constexpr int R = 2 << (LS + 1);   // bytes per PAIR of interleaved table rows
    if constexpr (!BIAS) {
      asm volatile(
        "v_lshlrev_b32_sdwa %1, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %2, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %3, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %4, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_lshlrev_b32_sdwa %5, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %6, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %7, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %8, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_add_u32 %1, %21, %1\n\t" "v_add_u32 %2, %21, %2\n\t" "v_add_u32 %3, %21, %3\n\t" "v_add_u32 %4, %21, %4\n\t"
        "v_add_u32 %5, %21, %5\n\t" "v_add_u32 %6, %21, %6\n\t" "v_add_u32 %7, %21, %7\n\t" "v_add_u32 %8, %21, %8\n\t"
        "ds_read_u16 %1, %1 offset:%13\n\t"
        "ds_read_u16 %2, %2 offset:%14\n\t"
        "ds_read_u16 %3, %3 offset:%15\n\t"
        "ds_read_u16 %4, %4 offset:%16\n\t"
        "ds_read_u16 %5, %5 offset:%17\n\t"
        "ds_read_u16 %6, %6 offset:%18\n\t"
        "ds_read_u16 %7, %7 offset:%19\n\t"
        "ds_read_u16 %8, %8 offset:%20\n\t"
        "s_waitcnt lgkmcnt(7)\n\t" "v_fma_mix_f32 %0, %1, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t" "v_fma_mix_f32 %0, %2, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(5)\n\t" "v_fma_mix_f32 %0, %3, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t" "v_fma_mix_f32 %0, %4, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t" "v_fma_mix_f32 %0, %5, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t" "v_fma_mix_f32 %0, %6, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t" "v_fma_mix_f32 %0, %7, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t" "v_fma_mix_f32 %0, %8, %12, %0 op_sel_hi:[1,0,0]"
        : "+v"(s), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(v0), "v"(v1), "v"(one), "v"(onef),
          "n"((J + 0) * R), "n"((J + 1) * R), "n"((J + 2) * R), "n"((J + 3) * R), "n"((J + 4) * R), "n"((J + 5) * R), "n"((J + 6) * R), "n"((J + 7) * R), "v"(kadd)
        : "memory");
      return s;
    }
    asm volatile(
        "v_lshlrev_b32_sdwa %1, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %2, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %3, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %4, %11, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "v_lshlrev_b32_sdwa %5, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_lshlrev_b32_sdwa %6, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
        "v_lshlrev_b32_sdwa %7, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
        "v_lshlrev_b32_sdwa %8, %11, %10 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
        "ds_read_u16 %1, %1 offset:%13\n\t"
        "ds_read_u16 %2, %2 offset:%14\n\t"
        "ds_read_u16 %3, %3 offset:%15\n\t"
        "ds_read_u16 %4, %4 offset:%16\n\t"
        "ds_read_u16 %5, %5 offset:%17\n\t"
        "ds_read_u16 %6, %6 offset:%18\n\t"
        "ds_read_u16 %7, %7 offset:%19\n\t"
        "ds_read_u16 %8, %8 offset:%20\n\t"
#if COLTT_PQ_SUM_WAITS == 8
        "s_waitcnt lgkmcnt(7)\n\t" "v_fma_mix_f32 %0, %1, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t" "v_fma_mix_f32 %0, %2, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(5)\n\t" "v_fma_mix_f32 %0, %3, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t" "v_fma_mix_f32 %0, %4, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t" "v_fma_mix_f32 %0, %5, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(2)\n\t" "v_fma_mix_f32 %0, %6, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(1)\n\t" "v_fma_mix_f32 %0, %7, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t" "v_fma_mix_f32 %0, %8, %12, %0 op_sel_hi:[1,0,0]"
#else   // two waits per block: 14 issue slots fewer per 16 lookups, the first add of each half behind four reads instead of one
        "s_waitcnt lgkmcnt(4)\n\t" "v_fma_mix_f32 %0, %1, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %2, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %3, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %4, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t" "v_fma_mix_f32 %0, %5, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %6, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %7, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %8, %12, %0 op_sel_hi:[1,0,0]"
#endif
        : "+v"(s), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(v0), "v"(v1), "v"(one), "v"(onef),
          "n"((J + 0) * R), "n"((J + 1) * R), "n"((J + 2) * R), "n"((J + 3) * R), "n"((J + 4) * R), "n"((J + 5) * R), "n"((J + 6) * R), "n"((J + 7) * R)
        : "memory");
    return s;
  }
  template <int I> __device__ __forceinline__ float sum_piece(float s, const u32x4e& r, uint32_t bias, uint32_t kadd) const {   // the 16 lookups of this lane's piece I
    s = sum8<I * 16>(s, r[0] + bias, r[1] + bias, kadd);
    return sum8<I * 16 + 8>(s, r[2] + bias, r[3] + bias, kadd);
  }
  // this lane's PARTIAL sum (its half of the row, in j order from +0.0); pair_total() makes the distance
  __device__ __forceinline__ float sum(const u32x4e (&r)[NR]) const {
    const int cnt = my_pieces();
    float s = 0.f;
    if constexpr (LS != 0) {
      const uint32_t bias = BIAS ? (hsel ? (uint32_t)(1u << LS) * 0x01010101u : 0u) : 0u;   // h * C' in every byte (BIAS form)
      const uint32_t kadd = BIAS ? 0u : (hsel << (LS + 1));                                 // h * R bytes (ADDK form)
      constexpr bool EVEN = NP != 0 && NP % 2 == 0;   // both lanes own NR pieces: no per-lane test
      if (EVEN || 0 < cnt) s = sum_piece<0>(s, r[0], bias, kadd);
      if constexpr (NR > 1) { if (EVEN || 1 < cnt) s = sum_piece<1>(s, r[1], bias, kadd); }
      if constexpr (NR > 2) { if (EVEN || 2 < cnt) s = sum_piece<2>(s, r[2], bias, kadd); }
      if constexpr (NR > 3) { if (EVEN || 3 < cnt) s = sum_piece<3>(s, r[3], bias, kadd); }
      return s;
    } else {
      const uint32_t sh = lut_shift;
#pragma unroll
      for (int i = 0; i < NR; i++) {
        if (i < cnt) {
#pragma unroll
          for (int wd = 0; wd < 4; wd++) {
            const uint32_t v = r[i][wd];
            const uint32_t j = (uint32_t)(i * 16 + wd * 4);
            s = acc(s, entry<0>(j, sh, v)); s = acc(s, entry<1>(j + 1, sh, v)); s = acc(s, entry<2>(j + 2, sh, v)); s = acc(s, entry<3>(j + 3, sh, v));
          }
        }
      }
      return s;
    }
  }
  // d(query, slot) in every lane of the pair (every lane of the wave when they all pass the same slot: the entrypoint)
  __device__ __forceinline__ float adc(uint32_t slot) const { u32x4e r[NR]; load(slot, r); return pair_total(sum(r)); }
  // The code row of every LISTED neighbour is requested before the walk knows which of them are fresh: 32-128 bytes each, in flight
  // under the visited test's own dependent HBM probe instead of behind it (one round trip less per expansion; rows of already visited
  // neighbours are fetched for nothing — a few KB per expansion against a dependent ~2 us).
  __device__ __forceinline__ void prefetch(uint32_t nb, bool valid, int /*half*/) { if (valid) load(nb, raw); }   // both lanes: each its half of the row
  // NBR: the code row of candidate `cand`'s neighbour at position idx of its level-0 row (in_row: idx < mMax0)
  __device__ __forceinline__ void prefetch_at(uint32_t cand, uint32_t idx, bool in_row, int /*half*/) {
    if (in_row) load_from(nbrc + (size_t)cand * nbr_stride + (size_t)idx * row_bytes, raw);
  }
  // RADJ over the neighbourhood blocks: the runner-up's block is requested at POP time into a second set of row registers; if the runner-up is indeed the
  // next candidate (no nearer vertex admitted meanwhile — the common case once the result set is full) the rows are simply taken over
  u32x4e spec_raw[NBR ? NR : 1];
  __device__ __forceinline__ void prefetch_spec(uint32_t cand, uint32_t idx, bool in_row, int /*half*/) {
    if constexpr (NBR) { if (in_row) load_from(nbrc + (size_t)cand * nbr_stride + (size_t)idx * row_bytes, spec_raw); }
  }
  __device__ __forceinline__ void take_spec() {
    if constexpr (NBR) {
#pragma unroll
      for (int i = 0; i < NR; i++) raw[i] = spec_raw[i];
    }
  }
  float pre_d;   // EARLY: the table sum of this lane pair's neighbour, computed under the visited probe (both lanes hold it)
  __device__ __forceinline__ void early(bool valid, int /*half*/) { float part = 0.f; if (valid) part = sum(raw); pre_d = pair_total(part); }
  // ties a value loaded before early() to early()'s result: the compiler may not use (hence wait for) it before the sums are computed
  __device__ __forceinline__ void after_early(uint32_t& x) { asm volatile("" : "+v"(x), "+v"(pre_d)); }
  // the table sum over the code row prefetch() requested (greedy descent: every valid neighbour is evaluated at once)
  __device__ __forceinline__ float eval_now(bool fresh, int /*half*/) const {
    float part = 0.f;
    if (fresh) part = sum(raw);
    return pair_total(part);   // both lanes of the pair
  }
  __device__ __forceinline__ float operator()(const GraphView&, const WaveCtx&, uint32_t, bool fresh, float, int half, int) const {
    if constexpr (EARLY) return fresh ? pre_d : 0.f;
    else return eval_now(fresh, half);
  }
};

// greedyClosestNeighbor (hnsw.go:320-343) with table distances: hnsw_dev.hpp:greedy_level with AdcEval in place of eval_pair
template <class EV>
__device__ __forceinline__ void greedy_level_adc(const GraphView& g, WaveCtx& w, EV& ev, uint32_t& cur, float& curd, int level, int lane_in) {
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
      ev.prefetch(nb, valid, half);
      const float d = ev.eval_now(valid, half);
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

}  // namespace dev
}  // namespace coltt
