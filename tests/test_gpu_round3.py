"""Round-3 GPU parity cases.
* The committed golden fixtures (tests/golden/*.npz, produced by the oracle and pinned on CPU by tests/test_oracle.py) are consumed
  DIRECTLY by the HIP path: leaf kernels, FLAT searches in both modes, HNSW answers and counters.
* Matrix-core (COLTT_MODE_MFMA) searches are compared with the ORACLE — not only with the GPU's exact mode — at 768-d, large
  dims, dims that are no multiple of the 32-column step, and Euclidean (edge/none_vectorstore.go:129-180).
* The cosine matrix-core path is closed for stores that hold rows of norm far from 1 (loaded streams are not re-normalised)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results, bits

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DIST_DIMS = (1, 7, 8, 9, 31, 32, 33, 128, 768, 1536)   # tests/golden/make_golden.py: DIST_DIMS


def test_golden_kernels_on_the_gpu(gpu):
    g = np.load(os.path.join(GOLD, "kernels.npz"))
    K = gpu.kernels
    for d in DIST_DIMS:
        a = O.fill_normal(100 + d, (8, d)); b = O.fill_normal(200 + d, (8, d))
        for mname, metric in (("cos", O.COSINE), ("l2", O.L2)):
            for order in (0, 1, 2):
                if order == 0 and d % 4:   # AVX order over PACKED rows needs dim % 4 == 0; the stores pad their rows — see below
                    continue
                assert np.array_equal(bits(K.distance_pairs(metric, a, b, order)), g[f"dist_{mname}_{order}_{d}"]), (mname, order, d)
        if d % 4:   # AVX-order Euclidean distance at odd dims through a FLAT store (rows padded to 16 B, scalar tail: avx.cpp:28-31)
            f = gpu.FlatSpace(d, O.L2, gpu.Q_NONE); f.ChangedVertex(np.arange(8, dtype=np.uint64), b)
            gi, gs, gc = f.VertexSearch(a, 8, gpu.SELECT_NEAREST)
            for i in range(8):
                assert bits(gs[i][list(gi[i]).index(i)]) == g[f"dist_l2_0_{d}"][i], (d, i)
        assert np.array_equal(bits(K.normalize(a)), g[f"norm_{d}"]), d
        assert np.array_equal(bits(np.array([K.pq_float_scan(0, a[i], b[i:i + 1])[0] for i in range(8)], np.float32)), g[f"pqdot_{d}"]), d
        assert np.array_equal(bits(np.array([K.pq_float_scan(1, a[i], b[i:i + 1])[0] for i in range(8)], np.float32)), g[f"pql2_{d}"]), d
    x = g["enc_in"].view(np.float32)
    assert np.array_equal(K.quant_lower(gpu.Q_F16, x), g["f16_encode"]) and np.array_equal(K.quant_lower(gpu.Q_F8, x), g["f8_encode"])
    assert np.array_equal(bits(K.quant_raise(gpu.Q_F8, np.arange(256, dtype=np.uint8))), g["f8_lut"])
    assert np.array_equal(bits(K.quant_raise(gpu.Q_F16, g["f16_encode"])), bits(O.f16_decode(g["f16_encode"])))
    ids = (np.arange(1000, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(12345)
    assert np.array_equal(K.shard_vertex(ids, 16).astype(np.uint8), g["shard16"])
    assert np.array_equal(K.pq_bit_scan(0, g["bit_q"], g["bit_rows"]), g["hamming"])
    assert np.array_equal(bits(K.pq_bit_scan(1, g["bit_q"], g["bit_rows"])), g["jaccard"])


@pytest.mark.parametrize("mode", ["exact", "mfma"])
def test_golden_flat_on_the_gpu(gpu, mode):
    g = np.load(os.path.join(GOLD, "flat_2048x128.npz"))
    n, d = 2048, 128
    X = O.fill_normal(1, (n, d)); Q = O.fill_normal(99, (16, d))
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7919) + np.uint64(1000003)) % np.uint64(1 << 40)
    m = gpu.MODE_MFMA if mode == "mfma" else gpu.MODE_EXACT
    for metric, quant in ((0, 0), (1, 1), (0, 2), (1, 3)):
        f = gpu.FlatSpace(d, metric, quant); f.ChangedVertex(ids, X)
        for k, nearest in ((10, 0), (100, 1)):
            gi, gs, gc = f.VertexSearch(Q, k, gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE, m)
            assert (gc == k).all()
            assert np.array_equal(gi, g[f"ids_{metric}_{quant}_{k}_{nearest}"]), (metric, quant, k, nearest)
            assert np.array_equal(bits(gs), g[f"sc_{metric}_{quant}_{k}_{nearest}"]), (metric, quant, k, nearest)
        if mode == "mfma" and quant != 2:
            assert f.Stats()["mfma_groups"] > 0, (metric, quant)


def test_golden_hnsw_on_the_gpu(gpu):
    """tests/golden/hnsw.npz: graphs built by the oracle's literal Insert, answers + (n_dist, n_exp, n_hops) per query."""
    h = np.load(os.path.join(GOLD, "hnsw.npz"))
    for n, d, metric, tag in ((1000, 128, O.COSINE, "1000x128_cos"), (3000, 64, O.L2, "3000x64_l2"), (1500, 768, O.COSINE, "1500x768_cos")):
        X = O.fill_normal(40 + d, (n, d)); lv = O.levels(41 + d, n)
        ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(11)
        oh = O.Hnsw(d, metric); oh.insert_many(ids, X, lv)
        assert oh.graph_hash() == int(h[f"{tag}_graph_hash"][0])
        gh = gpu.Hnsw(d, metric); gh.BulkLoad(oh.export(with_vectors=False), X)
        Q = O.fill_normal(123, (40, d))
        for ef in (20, 128):
            for qi in range(40):   # one query per call: the fixture holds per-query counters
                gi, gs, gc, st = gh.Search(Q[qi:qi + 1], 10, ef=ef, with_stats=True)
                assert np.array_equal(gi[0, :gc[0]], h[f"{tag}_ids_{ef}"][qi]) and np.array_equal(bits(gs[0, :gc[0]]), h[f"{tag}_sc_{ef}"][qi]), (tag, ef, qi)
                assert (st["n_dist"], st["n_exp"], st["n_hops"]) == tuple(int(v) for v in h[f"{tag}_stats_{ef}"][qi]), (tag, ef, qi)


@pytest.mark.parametrize("metric,quant,n,d", [(O.COSINE, O.Q_F16, 9000, 768), (O.COSINE, O.Q_NONE, 5000, 768), (O.L2, O.Q_F16, 6000, 768),
                                              (O.L2, O.Q_NONE, 4000, 300), (O.COSINE, O.Q_BF16, 3000, 1536), (O.COSINE, O.Q_F16, 2500, 2048),
                                              (O.COSINE, O.Q_NONE, 6000, 300), (O.COSINE, O.Q_F16, 5000, 200)])
def test_mfma_mode_equals_the_oracle(gpu, metric, quant, n, d):
    """Matrix-core candidates + exact re-score against the oracle's scan over the store's own rows (ids, ranks, score bits), both
    queue directions, several batch shapes."""
    X = O.fill_normal(4000 + d + quant, (n, d)); X[100:120] = X[7]          # exact duplicates sit on the candidate margin
    ids = np.arange(n, dtype=np.uint64) + np.uint64(5)
    gf = gpu.FlatSpace(d, metric, quant); gf.ChangedVertex(ids, X)
    rows = gf.FetchRows()
    for nq, k in ((1, 10), (70, 10), (200, 33)):
        Q = np.concatenate([X[7:8], O.fill_normal(4100 + nq, (nq - 1, d))]) if nq > 1 else O.fill_normal(4101, (1, d))
        for nearest in (True, False):
            mi, ms, mc = gf.VertexSearch(Q, k, gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE, gpu.MODE_MFMA)
            sl, sc, cn, _ = O.flat_scan(rows, quant, d, metric, Q[:12], k, nearest=nearest, threads=4)
            for qi in range(min(nq, 12)):
                assert_same_results(mi[qi, :mc[qi]], ms[qi, :mc[qi]], sl[qi, :cn[qi]] + np.uint64(5), sc[qi, :cn[qi]], f"nq{nq} k{k} nearest{nearest} q{qi}")
    assert gf.Stats()["mfma_groups"] > 0


def test_cosine_mfma_is_closed_for_rows_of_odd_norm(gpu):
    """LoadVertex stores the stream's vectors as they are (none_vectorstore.go:425-516 does not re-normalise).  Rows of norm 1e-5
    or 1e4 in a cosine store leave binary16's normal range when f32 rows are rounded into the matrix-core fragments: the proof of the
    candidate margin does not cover them, so such a store answers MODE_MFMA requests through the exact scan — same bits as the oracle."""
    n, d = 4000, 128
    X = O.fill_normal(4200, (n, d))
    X[::3] *= np.float32(1e-5 / np.sqrt(d)); X[1::3] *= np.float32(1e4 / np.sqrt(d))
    ids = np.arange(n, dtype=np.uint64)
    src = O.Flat(d, O.L2, O.Q_NONE); src.upsert(ids, X)          # an L2 store keeps the rows unnormalised; the stream is metric-agnostic
    stream = src.save_vertex()
    of = O.Flat(d, O.COSINE, O.Q_NONE); assert of.load_vertex(stream) == 0
    gf = gpu.FlatSpace(d, O.COSINE, O.Q_NONE); assert gf.LoadVertex(stream) == n
    Q = O.fill_normal(4201, (40, d))
    before = gf.Stats()["mfma_groups"]
    for nearest in (True, False):
        mi, ms, mc = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE, gpu.MODE_MFMA)
        for qi in range(len(Q)):
            wi, ws = of.search(Q[qi], 10, nearest=nearest, mode=2)
            assert_same_results(mi[qi, :mc[qi]], ms[qi, :mc[qi]], wi, ws, f"nearest{nearest} q{qi}")
    assert gf.Stats()["mfma_groups"] == before, "norms outside [1/2, 2]: the matrix-core path must not have been taken"
    mn, mx, open_ = gf.NormBounds()          # the latch is visible to the caller (coltt_flat_norm_bounds, round 4)
    assert not open_ and mn < 1e-8 and mx > 1e6
    # a well-formed store next to it still takes the matrix cores
    g2 = gpu.FlatSpace(d, O.COSINE, O.Q_NONE); g2.ChangedVertex(ids, X)
    g2.VertexSearch(Q, 10, gpu.SELECT_NEAREST, gpu.MODE_MFMA)
    assert g2.Stats()["mfma_groups"] > 0
    mn, mx, open_ = g2.NormBounds()
    assert open_ and 0.99 < mn <= mx < 1.01


@pytest.mark.parametrize("metric,quant,n,d", [(O.COSINE, O.Q_NONE, 40000, 128), (O.COSINE, O.Q_F16, 30000, 768), (O.L2, O.Q_BF16, 20000, 200),
                                              (O.L2, O.Q_NONE, 12000, 768)])
def test_filtered_search_through_the_matrix_cores(gpu, metric, quant, n, d):
    """FilterableVertexSearch (edge/none_vectorstore.go:182-253) with COLTT_MODE_MFMA: the gathered rows feed the matrix-core candidate
    kernel (flat_mfma.hpp, GATHER), survivors are re-scored in exact order — ids, ranks and score bits equal exact mode and the
    oracle's scan over exactly the candidate rows.  Candidate lists: strided, random, tiny (< one tile), with unknown and removed ids
    (skipped, :201) and repeated ids (scored once)."""
    X = O.fill_normal(5000 + d + quant, (n, d)); X[300:310] = X[4]
    ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(7)
    gf = gpu.FlatSpace(d, metric, quant); gf.ChangedVertex(ids, X)
    gone = ids[100:140]; gf.RemoveVertex(gone)
    rows, row_ids = gf.FetchRows(with_ids=True)                       # the store's rows in scan order, after the removal
    slot_of = {int(i): s for s, i in enumerate(row_ids)}
    rng = np.random.default_rng(11)
    lists = {"strided": ids[::7], "random": np.sort(rng.choice(ids, n // 3, replace=False)), "tiny": ids[1000:1100],
             "dirty": np.concatenate([ids[::5], ids[:50], gone, np.uint64(10**12) + np.arange(30, dtype=np.uint64)])}
    before = gf.Stats()["mfma_groups"]
    for name, cand in lists.items():
        live = np.array(sorted({slot_of[int(c)] for c in cand if int(c) in slot_of}), np.int64)
        sub = np.ascontiguousarray(rows[live])
        for nq, k in ((1, 10), (40, 10), (130, 25)):
            Q = np.concatenate([X[4:5], O.fill_normal(5100 + nq, (nq - 1, d))]) if nq > 1 else O.fill_normal(5101, (1, d))
            for nearest in (True, False):
                sel = gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE
                ei, es, ec = gf.FilterableVertexSearch(cand, Q, k, sel, gpu.MODE_EXACT)
                mi, ms, mc = gf.FilterableVertexSearch(cand, Q, k, sel, gpu.MODE_MFMA)
                assert np.array_equal(ec, mc) and np.array_equal(ei, mi) and np.array_equal(bits(es), bits(ms)), (name, nq, k, nearest)
                sl, sc, cn, _ = O.flat_scan(sub, quant, d, metric, Q[:6], k, nearest=nearest, threads=4)
                for qi in range(min(nq, 6)):
                    assert_same_results(mi[qi, :mc[qi]], ms[qi, :mc[qi]], row_ids[live[sl[qi, :cn[qi]].astype(np.int64)]], sc[qi, :cn[qi]], f"{name} nq{nq} q{qi} near{nearest}")
    assert gf.Stats()["mfma_groups"] > before, "the filtered searches must really have gone through the matrix cores"


@pytest.mark.parametrize("metric,quant,d,n", [(O.COSINE, O.Q_NONE, 128, 70000), (O.L2, O.Q_NONE, 77, 20000), (O.COSINE, O.Q_F16, 768, 9000),
                                              (O.L2, O.Q_F8, 40, 30000), (O.COSINE, O.Q_BF16, 200, 12000), (O.L2, O.Q_F16, 12, 300000)])
def test_small_batches_in_one_launch_equal_the_oracle(gpu, monkeypatch, metric, quant, d, n):
    """<= 4 queries, k <= 64: scan, per-wave / per-block k best and the final selection in ONE kernel (flat.hip: flat_one_kernel) —
    the shape of the reference's RPC (one query per VertexSearch call, edge/none_vectorstore.go:104-180).  Ids, ranks and score bits
    equal the oracle's for both queue directions, every k, unfiltered and filtered (none_vectorstore.go:182-253), with score ties
    (duplicated rows: broken by id, and the ids are NOT in slot order) and for stores smaller than k, a wave and a block."""
    X = O.fill_normal(6000 + d + quant, (n, d)); X[500:600] = X[3]; X[n - 40:] = X[3]
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 33)
    Q = np.concatenate([X[3:4], O.fill_normal(6100 + d, (3, d))])
    calls = 0
    for m in (3, 31, 700, n):
        gf = gpu.FlatSpace(d, metric, quant); gf.ChangedVertex(ids[n - m:], X[n - m:])
        of = O.Flat(d, metric, quant); of.upsert(ids[n - m:], X[n - m:])
        rng = np.random.default_rng(m)
        cand = np.concatenate([np.unique(np.concatenate([rng.choice(ids[n - m:], max(1, m // 3), replace=False), ids[n - 2:]])),   # a roaring ToArray(): ascending, unique
                               np.uint64(10**13) + np.arange(5, dtype=np.uint64)])
        calls = 0
        for nq in (1, 2, 4):
            for k in (1, 10, 64):
                for nearest in (True, False):
                    sel = gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE
                    mode = gpu.MODE_EXACT if (nq + k) % 2 else gpu.MODE_MFMA      # the mode is irrelevant for these shapes
                    gi, gs, gc = gf.VertexSearch(Q[:nq], k, sel, mode)
                    fi, fs, fc = gf.FilterableVertexSearch(cand, Q[:nq], k, sel, mode)
                    calls += 2
                    for qi in range(nq):
                        wi, ws = of.search(Q[qi], k, nearest=nearest, mode=2)
                        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"m{m} nq{nq} k{k} near{nearest} q{qi}")
                        wi, ws = of.search(Q[qi], k, nearest=nearest, mode=2, cand=cand)
                        assert_same_results(fi[qi, :fc[qi]], fs[qi, :fc[qi]], wi, ws, f"filtered m{m} nq{nq} k{k} near{nearest} q{qi}")
        assert gf.OneLaunchSearches() == calls, (m, gf.OneLaunchSearches(), calls)
    # the chain of scan + select launches gives the same answers (COLTT_FLAT_ONE=0), and k > 64 / more than four queries still take it
    e1 = gf.VertexSearch(Q[:1], 10, gpu.SELECT_NEAREST, gpu.MODE_EXACT)
    monkeypatch.setenv("COLTT_FLAT_ONE", "0")
    e0 = gf.VertexSearch(Q[:1], 10, gpu.SELECT_NEAREST, gpu.MODE_EXACT)
    assert np.array_equal(e0[0], e1[0]) and np.array_equal(bits(e0[1]), bits(e1[1])) and gf.OneLaunchSearches() == calls + 1
    monkeypatch.delenv("COLTT_FLAT_ONE")
    gf.VertexSearch(Q[:1], 65, gpu.SELECT_NEAREST, gpu.MODE_EXACT); gf.VertexSearch(np.concatenate([Q, Q]), 10, gpu.SELECT_NEAREST, gpu.MODE_EXACT)
    assert gf.OneLaunchSearches() == calls + 1
