"""Eight lanes per row (coltt_amd/csrc/rows8.hpp): indexes whose rows are f32 / 2-byte codes of a byte length that is a multiple of 128
keep a line-transposed copy of their rows, and the level-0 distances of Hnsw.Search (core/vectorindex/hnsw.go:345-389) come from an
8-lane core over it.  Same summation order as the pair-owned walk, so ids, score bits and traversal counters must equal the oracle's
AND the pair-owned kernel's (COLTT_EV8=0), at every ef, for both walks (LDS hash, HBM map), with tombstones, after Load / BulkLoad."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results, bits

pytestmark = pytest.mark.gpu


def _build(gpu, X, lv, metric, quant, cfg=None, batch=256, ids=None):
    import torch
    n, d = X.shape
    gh = gpu.Hnsw(d, metric, cfg, quantization=quant)
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    i = 0
    while i < n:
        b = int(min(n - i, max(1, min(batch, i // 16))))
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i, ids=None if ids is None else ids[i:i + b])
        i += b
    return gh


def _oracle_check(gh, Q, quant, metric, ef, k, del_bits=None):
    g = gh.ExportRaw(); rows = gh.FetchRows()
    gi, gs, gc, st = gh.Search(Q, k, ef=ef, with_stats=True)
    sl, sc, cn, ost, _ = O.csr_search(rows, quant, g["adj0"], g["upper_off"], g["adjU"], gh.dim, metric, g["entry"], g["entry_level"], Q, k, ef,
                                      del_bits=del_bits, threads=4)
    for qi in range(len(Q)):
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"q{qi} ef{ef}")
    assert {k_: st[k_] for k_ in ost} == ost, (ef, st, ost)
    return gi, gs, gc, st


@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
@pytest.mark.parametrize("quant,d", [(O.Q_NONE, 32), (O.Q_NONE, 128), (O.Q_NONE, 768), (O.Q_F16, 64), (O.Q_F16, 256), (O.Q_BF16, 768), (O.Q_F16, 1536)])
def test_eight_lane_core_equals_the_oracle_and_the_pair_owned_walk(gpu, monkeypatch, metric, quant, d):
    """1 / 4 / 24 lines of f32 rows, 1 / 4 / 12 / 24 lines of 2-byte rows; ef below and above the LDS / HBM visited threshold.
    (COLTT_ROWS8=2: by default only dim >= 256 keeps the copy — short rows do not amortise the core's per-chunk hand-off.)"""
    monkeypatch.setenv("COLTT_ROWS8", "2")
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")          # the one-wave throughput kernels (the latency kernel has its own evaluator)
    n = 4000 if d <= 768 else 1500
    X = O.fill_normal(9000 + d, (n, d)); lv = O.levels(9001 + d, n)
    gh = _build(gpu, X, lv, metric, quant, gpu.HnswCfg.default(ef_construction=60))
    launches0, has = gh.Rows8()
    assert has and launches0 == 0                         # the builder searches the pair-owned rows
    Q = O.fill_normal(9002 + d, (70, d))
    served = 0
    for ef, k in ((16, 10), (128, 10), (129, 10), (300, 10), (1024, 100)):
        gi, gs, gc, st = _oracle_check(gh, Q, quant, metric, ef, k)
        served += 1
        assert gh.Rows8()[0] == launches0 + served, (ef, gh.Rows8())
        monkeypatch.setenv("COLTT_EV8", "0")
        pi, ps, pc, pst = gh.Search(Q, k, ef=ef, with_stats=True)
        monkeypatch.delenv("COLTT_EV8")
        assert gh.Rows8()[0] == launches0 + served         # ... and that call did not use it
        assert np.array_equal(gi, pi) and np.array_equal(bits(gs), bits(ps)) and np.array_equal(gc, pc) and st == pst, ef


def test_tombstones_load_and_bulk_load_keep_the_copy_in_step(gpu, monkeypatch):
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    n, d = 3000, 256
    X = O.fill_normal(9100, (n, d)); lv = O.levels(9101, n); Q = O.fill_normal(9102, (30, d))
    gh = _build(gpu, X, lv, O.COSINE, O.Q_NONE, gpu.HnswCfg.default(ef_construction=50))
    db = np.zeros((n + 31) // 32, np.uint32)
    for i in range(0, n, 9):
        gh.Remove(i)
        db[i >> 5] |= np.uint32(1 << (i & 31))
    _oracle_check(gh, Q, O.Q_NONE, O.COSINE, 64, 10, del_bits=db)
    _oracle_check(gh, Q, O.Q_NONE, O.COSINE, 200, 10, del_bits=db)
    assert gh.Rows8()[1]
    # single inserts after a batch: the copy follows
    extra = O.fill_normal(9103, (5, d))
    for j in range(5):
        gh.Insert(10_000 + j, extra[j], 0)
    assert gh.Rows8()[1]
    a = gh.Search(Q, 10, ef=64)
    monkeypatch.setenv("COLTT_EV8", "0"); b = gh.Search(Q, 10, ef=64); monkeypatch.delenv("COLTT_EV8")
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))
    # Commit -> Load into a fresh index, BulkLoad of an oracle graph
    stream = gh.Commit()
    g2 = gpu.Hnsw(d, O.COSINE); g2.Load(stream)
    assert g2.Rows8() == (0, True)
    c = g2.Search(Q, 10, ef=64)
    assert np.array_equal(a[0], c[0]) and np.array_equal(bits(a[1]), bits(c[1])) and g2.Rows8()[0] == 1
    oh = O.Hnsw(d, O.L2); oh.insert_many(np.arange(800, dtype=np.uint64), X[:800], lv[:800])
    g3 = gpu.Hnsw(d, O.L2); g3.BulkLoad(oh.export(with_vectors=False), X[:800])
    assert g3.Rows8() == (0, True)
    gi, gs, gc = g3.Search(Q, 10, ef=40)
    for qi in range(len(Q)):
        wi, ws = oh.search(Q[qi], 10, mode=1, ef=40)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"bulk q{qi}")
    assert g3.Rows8()[0] == 1


def test_shapes_without_a_copy_and_the_create_time_switch(gpu, monkeypatch):
    X = O.fill_normal(9200, (500, 96)); lv = O.levels(9201, 500)
    for quant, d, want in ((O.Q_F16, 96, False), (O.Q_NONE, 128, False), (O.Q_NONE, 256, True), (O.Q_F16, 320, True), (O.Q_F8, 256, False), (O.Q_NONE, 300, False)):
        gh = _build(gpu, np.ascontiguousarray(O.fill_normal(9202 + d, (500, d))), lv, O.COSINE, quant)
        assert gh.Rows8()[1] == want, (quant, d)
        gh.Search(O.fill_normal(9203, (4, d)), 5, ef=32)
        assert (gh.Rows8()[0] > 0) == want
    monkeypatch.setenv("COLTT_ROWS8", "0")
    gh = _build(gpu, O.fill_normal(9204, (500, 256)), lv, O.COSINE, O.Q_NONE)
    monkeypatch.delenv("COLTT_ROWS8")
    gh.Search(O.fill_normal(9205, (4, 256)), 5, ef=32)
    assert gh.Rows8() == (0, False)
    monkeypatch.setenv("COLTT_ROWS8", "2")            # ... and the limit lifted: a 128-d f32 index (4 lines per row) keeps the copy
    gh = _build(gpu, O.fill_normal(9206, (500, 128)), lv, O.COSINE, O.Q_NONE)
    monkeypatch.delenv("COLTT_ROWS8")
    gh.Search(O.fill_normal(9207, (4, 128)), 5, ef=32)
    assert gh.Rows8() == (1, True)


@pytest.mark.parametrize("quant,d", [(O.Q_NONE, 768), (O.Q_NONE, 256), (O.Q_NONE, 1536), (O.Q_F16, 768), (O.Q_BF16, 256)])
def test_non_temporal_twins_equal_the_oracle_and_the_default_kernels(gpu, monkeypatch, quant, d):
    """Round 6: collections far larger than the caches run the eight-lane kernels with non-temporal row loads (exact.hpp: row_ld; chosen per launch from the
    size of the row array, hnsw.hip: rows_nt).  A cache hint cannot change a value: with the hint forced on (COLTT_ROWS_NT=1) ids, score bits and counters
    equal the oracle's and those of the default kernels (COLTT_ROWS_NT=0), for the LDS-visited and the HBM-visited walk."""
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    n = 3000 if d <= 768 else 1500   # (f32 twins: one row x 12 lines per lane group — 8 lines: the tail burst alone; 24 and 48 lines: two and four bursts)
    X = O.fill_normal(9300 + d, (n, d)); lv = O.levels(9301 + d, n)
    gh = _build(gpu, X, lv, O.COSINE, quant, gpu.HnswCfg.default(ef_construction=60))
    assert gh.Rows8()[1]
    Q = O.fill_normal(9302 + d, (50, d))
    for ef, k in ((64, 10), (128, 10), (300, 10), (1024, 50)):
        monkeypatch.setenv("COLTT_ROWS_NT", "1")
        gi, gs, gc, st = _oracle_check(gh, Q, quant, O.COSINE, ef, k)
        monkeypatch.setenv("COLTT_ROWS_NT", "0")
        pi, ps, pc, pst = gh.Search(Q, k, ef=ef, with_stats=True)
        monkeypatch.delenv("COLTT_ROWS_NT")
        assert np.array_equal(gi, pi) and np.array_equal(bits(gs), bits(ps)) and np.array_equal(gc, pc) and st == pst, ef
