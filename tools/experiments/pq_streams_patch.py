"""Round 5, calls Y / Z: the experiment behind profiles/r05y_pq_streams_ab.md — coltt_hnsw_pq_search as query groups pipelined over 2 / 3 HIP streams
(COLTT_PQ_STREAMS).  Bit-identical answers (a GPU test compared it with the single chain and the oracle), 12-35 % SLOWER; not in the product.  This is the
patch as it was applied to the tree of commit "INTEGRATION: COLTT_PQ_WAVES knob" (python tools/experiments/pq_streams_patch.py from the repo root)."""
def patch(p, pairs):
    s = open(p).read()
    for old, new in pairs:
        assert old in s, old[:90]
        s = s.replace(old, new, 1)
    open(p, 'w').write(s)

patch('coltt_amd/csrc/common.hpp', [('''  int pq_waves = 0;            // COLTT_PQ_WAVES: resident traversals per CU of the product-quantised walk, 0 = the default cap
''', '''  int pq_waves = 0;            // COLTT_PQ_WAVES: resident traversals per CU of the product-quantised walk, 0 = the default cap
  int pq_streams = 0;          // COLTT_PQ_STREAMS: 1 = one launch chain per call, 2 / 3 = query groups pipelined over that many streams (hnsw.hip: pq_search_once), 0 = default
''')])
patch('coltt_amd/csrc/common.hip', [('''  v = num("COLTT_PQ_WAVES", set); p.pq_waves = set ? (int)std::max<long long>(1, std::min<long long>(16, v)) : 0;
''', '''  v = num("COLTT_PQ_WAVES", set); p.pq_waves = set ? (int)std::max<long long>(1, std::min<long long>(16, v)) : 0;
  v = num("COLTT_PQ_STREAMS", set); p.pq_streams = set ? (int)std::max<long long>(1, std::min<long long>(3, v)) : 0;
''')])
patch('coltt_amd/_lib.py', [('''"COLTT_PQ_WAVES",''', '''"COLTT_PQ_WAVES", "COLTT_PQ_STREAMS",''')])

patch('coltt_amd/csrc/hnsw.hip', [
# HCtx: extra streams
('''  PinnedBuf h_in, h_out;   // small calls: see PinnedBuf
  int init() {  // the caller has selected the index's device''', '''  PinnedBuf h_in, h_out;   // small calls: see PinnedBuf
  hipStream_t xs[2] = {nullptr, nullptr};            // product-quantised search, pipelined form: the other streams ...
  hipEvent_t xe[3] = {nullptr, nullptr, nullptr};    // ... fork event, one join event per extra stream
  int ensure_extra_streams() {
    for (auto& s : xs) if (!s) COLTT_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& e : xe) if (!e) COLTT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return COLTT_OK;
  }
  int init() {  // the caller has selected the index's device'''),
('''    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

struct Hnsw : Object {''', '''    if (ev1) (void)hipEventDestroy(ev1);
    for (auto e : xe) if (e) (void)hipEventDestroy(e);
    for (auto s : xs) if (s) (void)hipStreamDestroy(s);
    if (stream) (void)hipStreamDestroy(stream);
  }
};

struct Hnsw : Object {'''),
# launch helpers take a stream
('''int launch_pq_walk(Hnsw* x, HCtx* c, const PqGeom& sg, uint32_t grid, uint32_t region_base, const float* lut, uint32_t nq, uint32_t k,
                   uint32_t rerank, uint32_t* counter, uint32_t* surv, uint32_t* surv_cnt, unsigned long long* stats) {''',
 '''int launch_pq_walk(Hnsw* x, hipStream_t st, const PqGeom& sg, uint32_t grid, uint32_t region_base, const float* lut, uint32_t nq, uint32_t k,
                   uint32_t rerank, uint32_t* counter, uint32_t* surv, uint32_t* surv_cnt, unsigned long long* stats) {'''),
('''  kern<<<grid, 64, sg.lds, c->stream>>>(x->view(), x->entry, x->entry_level, lut, x->pq_codes.as<uint8_t>(), x->pq_row, sh, nq, k, sg.ef, sg.ef_pad, rerank,''',
 '''  kern<<<grid, 64, sg.lds, st>>>(x->view(), x->entry, x->entry_level, lut, x->pq_codes.as<uint8_t>(), x->pq_row, sh, nq, k, sg.ef, sg.ef_pad, rerank,'''),
('''int launch_pq_rerank(Hnsw* x, HCtx* c, const PqGeom& sg, uint32_t q0, uint32_t nq, uint32_t k, const uint32_t* surv, const uint32_t* surv_cnt,
                     unsigned long long* keys, uint64_t* oi, float* os, uint32_t* oc) {''',
 '''int launch_pq_rerank(Hnsw* x, HCtx* c, hipStream_t st, const PqGeom& sg, uint32_t q0, uint32_t nq, uint32_t k, const uint32_t* surv, const uint32_t* surv_cnt,
                     unsigned long long* keys, uint64_t* oi, float* os, uint32_t* oc) {'''),
('''  if (x->r8) hnsw_pq_rerank_kernel<METRIC, QUANT, true><<<grid, 64, 0, c->stream>>>(g, qe, qn, surv, surv_cnt, sg.ef_pad, keys);
  else hnsw_pq_rerank_kernel<METRIC, QUANT, false><<<grid, 64, 0, c->stream>>>(g, qe, qn, surv, surv_cnt, sg.ef_pad, keys);''',
 '''  if (x->r8) hnsw_pq_rerank_kernel<METRIC, QUANT, true><<<grid, 64, 0, st>>>(g, qe, qn, surv, surv_cnt, sg.ef_pad, keys);
  else hnsw_pq_rerank_kernel<METRIC, QUANT, false><<<grid, 64, 0, st>>>(g, qe, qn, surv, surv_cnt, sg.ef_pad, keys);'''),
('''  hnsw_pq_select_kernel<<<nq, 64, lds, c->stream>>>(keys, surv_cnt, sg.ef_pad, k, g.ids, oi + (size_t)q0 * k, os + (size_t)q0 * k, oc + q0);''',
 '''  hnsw_pq_select_kernel<<<nq, 64, lds, st>>>(keys, surv_cnt, sg.ef_pad, k, g.ids, oi + (size_t)q0 * k, os + (size_t)q0 * k, oc + q0);'''),
# geometry of the call
('''  uint32_t grid = (uint32_t)std::min<size_t>(nq, (size_t)256 * sg.per_cu);
  RegionLease lease;
  if (sg.variant != 0) { acquire_regions(x, grid, lease); grid = lease.count; }
  const float* d_q = queries;''', '''  const uint32_t resident = 256u * sg.per_cu;   // traversals the device holds at once
  uint32_t grid = (uint32_t)std::min<size_t>(nq, resident);
  const size_t lut_q = (size_t)x->pq_row * 1024;
  // Pipelined form (byte-map walk, calls of several thousand queries): the queries go in groups of R — one traversal per workgroup — round-robin over
  // S streams, each stream running  table -> walk -> re-rank -> select  for its group, every stream's walks on its own R visited-map regions.  The
  // re-rank (HBM-bound: 1 408 rows of 1.5 KB per query) and the table kernel of one group run under the other streams' walks (latency-bound), and with
  // S x R > the resident traversals there are always workgroups waiting to take the slots a finishing group frees: no launch tail but the last.
  const int S_pol = policy().pq_streams > 0 ? policy().pq_streams : PQ_STREAMS_DEFAULT;
  int S = (sg.variant == 2 && S_pol > 1 && nq >= 4096 && nq * lut_q <= (1024ull << 20) && nq <= 32768) ? S_pol : 1;
  RegionLease lease;
  if (sg.variant != 0) {
    acquire_regions(x, S > 1 ? (uint32_t)std::min<size_t>(nq, (size_t)S * resident) : grid, lease);
    if (S > 1 && lease.count / (uint32_t)S < 512u) S = 1;   // not enough regions free for groups worth launching
    grid = std::min(grid, lease.count);
  }
  const float* d_q = queries;'''),
('''  const size_t lut_q = (size_t)x->pq_row * 1024;
  const size_t group = std::max<size_t>(1, std::min<size_t>({nq, (1024ull << 20) / lut_q, (size_t)32768}));
  COLTT_TRY(c->w_pack.reserve(group * lut_q));
  COLTT_TRY(c->w_surv.reserve(group * sg.ef_pad * 4)); COLTT_TRY(c->w_scnt.reserve(group * 4)); COLTT_TRY(c->w_keys.reserve(group * sg.ef_pad * 8));
  COLTT_TRY(c->w_misc.reserve(256));
  uint8_t* misc = c->w_misc.as<uint8_t>();
  uint32_t* counter = reinterpret_cast<uint32_t*>(misc);
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(misc + 16);
  COLTT_HIP(hipMemsetAsync(misc, 0, 256, c->stream));
  COLTT_HIP(hipEventRecord(c->ev0, c->stream));
  for (size_t q0 = 0; q0 < nq; q0 += group) {''', '''  const size_t group = std::max<size_t>(1, std::min<size_t>({nq, (1024ull << 20) / lut_q, (size_t)32768}));   // S > 1: group == nq
  COLTT_TRY(c->w_pack.reserve(group * lut_q));
  COLTT_TRY(c->w_surv.reserve(group * sg.ef_pad * 4)); COLTT_TRY(c->w_scnt.reserve(group * 4)); COLTT_TRY(c->w_keys.reserve(group * sg.ef_pad * 8));
  COLTT_TRY(c->w_misc.reserve(1024));
  uint8_t* misc = c->w_misc.as<uint8_t>();
  uint32_t* counter = reinterpret_cast<uint32_t*>(misc);
  unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(misc + 16);
  COLTT_HIP(hipMemsetAsync(misc, 0, 1024, c->stream));
  COLTT_HIP(hipEventRecord(c->ev0, c->stream));
  if (S > 1) {
    const uint32_t R = lease.count / (uint32_t)S;
    const size_t G = (nq + R - 1) / R;
    if (G > 128) return fail(COLTT_E_DEVICE, "hnsw_pq_search: %zu query groups", G);   // (nq <= 32 768, R >= 512: at most 64)
    COLTT_TRY(c->ensure_extra_streams());
    hipStream_t st[3] = {c->stream, c->xs[0], c->xs[1]};
    COLTT_HIP(hipEventRecord(c->xe[0], c->stream));   // fork: the prepared queries and the zeroed counters are stream 0's work
    for (int s = 1; s < S; s++) COLTT_HIP(hipStreamWaitEvent(st[s], c->xe[0], 0));
    uint32_t* gcounter = reinterpret_cast<uint32_t*>(misc + 512);   // one work counter per group
    for (size_t gi = 0; gi < G; gi++) {
      const int s = (int)(gi % (size_t)S);
      const size_t q0 = gi * R, gn = std::min<size_t>(R, nq - q0);
      float* lut = c->w_pack.as<float>() + q0 * (lut_q / 4);
      uint32_t* surv = c->w_surv.as<uint32_t>() + q0 * sg.ef_pad; uint32_t* scnt = c->w_scnt.as<uint32_t>() + q0;
      unsigned long long* keys = c->w_keys.as<unsigned long long>() + q0 * sg.ef_pad;
      COLTT_TRY(pq_lut_batch(st[s], x->pq_cb.as<float>(), x->pq_shape, c->w_qeff.as<float>() + q0 * x->dim, gn, x->pq_row, lut));
      COLTT_TRY(launch_pq_walk(x, st[s], sg, (uint32_t)gn, lease.base + (uint32_t)s * R, lut, (uint32_t)gn, k, rerank, gcounter + gi, surv, scnt, d_stats));
      int rc;
#define COLTT_LP_ARGS x, c, st[s], sg, (uint32_t)q0, (uint32_t)gn, k, surv, scnt, keys, d_oi, d_os, d_oc
#define COLTT_LP(Q) rc = x->metric == COLTT_COSINE ? launch_pq_rerank<M_COS, Q>(COLTT_LP_ARGS) : launch_pq_rerank<M_L2, Q>(COLTT_LP_ARGS)
      if (x->quant == COLTT_Q_NONE) { COLTT_LP(Q_NONE); } else { COLTT_LP(Q_F16); }
#undef COLTT_LP
#undef COLTT_LP_ARGS
      COLTT_TRY(rc);
    }
    for (int s = 1; s < S; s++) {   // join
      COLTT_HIP(hipEventRecord(c->xe[s], st[s]));
      COLTT_HIP(hipStreamWaitEvent(c->stream, c->xe[s], 0));
    }
  } else
  for (size_t q0 = 0; q0 < nq; q0 += group) {'''),
('''    COLTT_TRY(launch_pq_walk(x, c, sg, (uint32_t)std::min<size_t>(grid, gn), lease.base, c->w_pack.as<float>(), (uint32_t)gn, k, rerank, counter, c->w_surv.as<uint32_t>(),
                             c->w_scnt.as<uint32_t>(), d_stats));''', '''    COLTT_TRY(launch_pq_walk(x, c->stream, sg, (uint32_t)std::min<size_t>(grid, gn), lease.base, c->w_pack.as<float>(), (uint32_t)gn, k, rerank, counter, c->w_surv.as<uint32_t>(),
                             c->w_scnt.as<uint32_t>(), d_stats));'''),
('''#define COLTT_LP_ARGS x, c, sg, (uint32_t)q0, (uint32_t)gn, k, c->w_surv.as<uint32_t>(), c->w_scnt.as<uint32_t>(), c->w_keys.as<unsigned long long>(), d_oi, d_os, d_oc''',
 '''#define COLTT_LP_ARGS x, c, c->stream, sg, (uint32_t)q0, (uint32_t)gn, k, c->w_surv.as<uint32_t>(), c->w_scnt.as<uint32_t>(), c->w_keys.as<unsigned long long>(), d_oi, d_os, d_oc'''),
('''constexpr size_t PQ_WAVES_CAP = 12;
''', '''constexpr size_t PQ_WAVES_CAP = 12;
// streams of the pipelined product-quantised search (pq_search_once): COLTT_PQ_STREAMS overrides
constexpr int PQ_STREAMS_DEFAULT = 1;
'''),
])
print('ok')
