"""The large-ef level-0 walk (coltt_amd/csrc/hnsw_walk2.hpp: delta result set, LDS Bloom filter in front of the HBM visited
map, neighbour norms riding with the adjacency rows) against the oracle's canonical Hnsw.Search (core/vectorindex/hnsw.go:243-278,
searchLevel :345-389) over the very arrays copied out of HBM: ids, ranks, f32 score bits AND the traversal counters.
Every case runs on the round-2 kernel (COLTT_WALK2=off), on the shipped variant and on its Bloom-less twin."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["off", "7", "6"], ids=["round2-kernel", "walk2", "walk2-no-bloom"])
def walk(request, monkeypatch):
    monkeypatch.setenv("COLTT_WALK2", request.param)
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    return request.param


def _gpu_build(gpu, X, lv, metric, quant, cfg=None, batch=64, ids=None):
    import torch
    n, d = X.shape
    gh = gpu.Hnsw(d, metric, cfg, quantization=quant)
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    i = 0
    while i < n:
        b = int(min(n - i, max(1, min(batch, i // 16))))
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i,
                             ids=None if ids is None else ids[i:i + b])
        i += b
    return gh


def _check(gh, Q, quant, metric, efs, k=10, del_bits=None, threads=4, id_of=None):
    d = gh.dim
    g = gh.ExportRaw(); rows = gh.FetchRows()
    for ef in efs:
        gi, gs, gc, st = gh.Search(Q, k, ef=ef, with_stats=True)
        sl, sc, cn, ost, _ = O.csr_search(rows, quant, g["adj0"], g["upper_off"], g["adjU"], d, metric, g["entry"], g["entry_level"],
                                          Q, k, ef, del_bits=del_bits, threads=threads)
        assert st["n_visit_resets"] == 0
        for qi in range(len(Q)):
            want = sl[qi, :cn[qi]].astype(np.uint64) if id_of is None else id_of[sl[qi, :cn[qi]]]
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], want, sc[qi, :cn[qi]], f"q{qi} ef{ef}")
        assert {k_: st[k_] for k_ in ost} == ost, (ef, st, ost)


@pytest.mark.parametrize("metric,quant,n,d", [(O.COSINE, O.Q_NONE, 6000, 128), (O.L2, O.Q_NONE, 5000, 64), (O.COSINE, O.Q_F16, 6000, 77),
                                              (O.COSINE, O.Q_BF16, 2500, 768), (O.L2, O.Q_F8, 3000, 40)])
def test_large_ef_walk_equals_oracle(gpu, walk, metric, quant, n, d):
    """ef from just above the LDS/HBM threshold to the 4096 maximum: delta flushes (64 admissions), evictions from the delta and
    from the main array's tail, sets that never fill (ef > what the graph can reach), k > 64."""
    X = O.fill_normal(3000 + d, (n, d)); lv = O.levels(3001 + d, n)
    gh = _gpu_build(gpu, X, lv, metric, quant, gpu.HnswCfg.default(ef_construction=60), batch=256)
    Q = O.fill_normal(3002 + d, (40, d))
    _check(gh, Q, quant, metric, (129, 300, 1024, 4096))
    _check(gh, Q[:8], quant, metric, (700,), k=300)


@pytest.mark.parametrize("m", [4, 24, 32])
def test_row_widths_other_than_32(gpu, walk, m):
    """mMax0 = 8 (a quarter chunk), 48 and 64 (two 32-neighbour chunks per expansion: free slots and the stale lowerBound
    carry across the chunks of one expansion, hnsw.go:357,374)."""
    n, d = 4000, 48
    X = O.fill_normal(3100 + m, (n, d)); lv = O.levels(3101 + m, n, m)
    gh = _gpu_build(gpu, X, lv, O.COSINE, O.Q_NONE, gpu.HnswCfg.default(m=m, ef_construction=48), batch=128)
    assert gh.cfg.m_max0 == 2 * m
    Q = O.fill_normal(3102, (32, d))
    _check(gh, Q, O.Q_NONE, O.COSINE, (130, 500, 2000))


def test_tiny_ef_through_the_hbm_visited_walk(gpu, walk, monkeypatch):
    """COLTT_VISG=1 sends every ef through the HBM-visited kernels: ef 1, 2, 3, 7 (the main array can drain to nothing while the
    delta holds the set), 64/65 (the delta's capacity)."""
    monkeypatch.setenv("COLTT_VISG", "1")
    n, d = 3000, 24
    X = O.fill_normal(3200, (n, d)); lv = O.levels(3201, n)
    gh = _gpu_build(gpu, X, lv, O.L2, O.Q_NONE, gpu.HnswCfg.default(ef_construction=40), batch=64)
    Q = O.fill_normal(3202, (50, d))
    for k, efs in ((1, (1, 2, 3, 7)), (10, (10, 33, 64, 65, 128))):
        _check(gh, Q, O.Q_NONE, O.L2, efs, k=k)


def test_neighbour_norm_rows_follow_insert_remove_and_load(gpu, walk):
    """GraphView::adj0_n is derived data (norms[adj0[..]]): written by the builder's two kernels, compacted by Remove's unlink
    kernel, rebuilt by bulk installs.  Cosine answers after every kind of mutation still equal the oracle's."""
    n, d = 2500, 32
    X = O.fill_normal(3300, (n, d)) * np.linspace(0.5, 4.0, n, dtype=np.float32)[:, None]   # norms differ before Normalize
    lv = O.levels(3301, n); ids = np.arange(n, dtype=np.uint64)
    gh = _gpu_build(gpu, X, lv, O.COSINE, O.Q_F16, gpu.HnswCfg.default(ef_construction=40), batch=64, ids=ids)
    Q = O.fill_normal(3302, (24, d))
    _check(gh, Q, O.Q_F16, O.COSINE, (200,))
    rng = np.random.default_rng(5)
    dead = rng.choice(n, 300, replace=False)
    for i in dead:
        gh.Remove(int(ids[i]))
    db = np.zeros((n + 31) // 32, np.uint32)
    for i in dead:
        db[i >> 5] |= np.uint32(1 << (i & 31))
    _check(gh, Q, O.Q_F16, O.COSINE, (200, 600), del_bits=db)
    # single Inserts on top (batch = 1: the reference's Insert), then search again
    Y = O.fill_normal(3303, (40, d)); ly = O.levels(3304, 40)
    for j in range(40):
        gh.Insert(10**6 + j, Y[j], int(ly[j]))
    db2 = np.zeros((n + 40 + 31) // 32, np.uint32); db2[:len(db)] = db
    _check(gh, Q, O.Q_F16, O.COSINE, (200,), del_bits=db2, id_of=np.concatenate([ids, np.uint64(10**6) + np.arange(40, dtype=np.uint64)]))
    # an f32 cosine index through Commit -> Load (stored vectors are not re-normalised on load, hnsw_commit.go:217-220)
    g32 = _gpu_build(gpu, X[:1200], lv[:1200], O.COSINE, O.Q_NONE, gpu.HnswCfg.default(ef_construction=40), batch=64)
    blob = g32.Commit()
    g2 = gpu.Hnsw(d, O.COSINE)
    assert g2.Load(blob) == 1200
    _check(g2, Q, O.Q_NONE, O.COSINE, (300,), id_of=g2.Export()["ids"])   # slots follow the stream's shard order
    # bulk load of an oracle-built graph
    oh = O.Hnsw(d, O.COSINE); oh.insert_many(ids[:900], X[:900], lv[:900])
    g3 = gpu.Hnsw(d, O.COSINE); g3.BulkLoad(oh.export(with_vectors=False), X[:900])
    gi, gs, gc = g3.Search(Q, 10, ef=256)
    for qi in range(len(Q)):
        wi, ws = oh.search(Q[qi], 10, mode=1, ef=256)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"bulk q{qi}")


def test_nan_scores_and_duplicate_distances(gpu, walk):
    """zero vectors under cosine give NaN distances (ordered by their IEEE bits on both sides); duplicated rows give exact
    distance ties, broken by slot — the 32-bit distance reductions of the delta must fall through to the slot bits."""
    n, d = 3000, 8
    X = O.fill_normal(3400, (n, d)); X[7] = 0.0; X[100] = 0.0
    X[1000:2000] = X[0:1000]                       # every distance appears twice
    lv = O.levels(3401, n, 8)
    gh = _gpu_build(gpu, X, lv, O.COSINE, O.Q_NONE, gpu.HnswCfg.default(m=8, ef_construction=24), batch=128)
    Q = O.fill_normal(3402, (16, d))
    _check(gh, Q, O.Q_NONE, O.COSINE, (200, 1500))


def test_epoch_wrap_with_the_bloom_filter(gpu, walk, monkeypatch):
    """> 255 traversals by one workgroup: the byte map's epoch wraps (region wiped) while the Bloom filter is cleared per
    traversal — the two must stay consistent."""
    monkeypatch.setenv("COLTT_VISG", "1")
    n, d = 2000, 16
    X = O.fill_normal(3500, (n, d)); lv = O.levels(3501, n)
    gh = _gpu_build(gpu, X, lv, O.L2, O.Q_NONE, gpu.HnswCfg.default(ef_construction=32), batch=64)
    g = gh.ExportRaw(); rows = gh.FetchRows()
    Q = O.fill_normal(3502, (2, d))
    sl, sc, cn, ost, _ = O.csr_search(rows, O.Q_NONE, g["adj0"], g["upper_off"], g["adjU"], d, O.L2, g["entry"], g["entry_level"], Q, 10, 150)
    for rep in range(300):
        gi, gs, gc, st = gh.Search(Q, 10, ef=150, with_stats=True)
        if rep % 60 == 0 or rep == 299:
            for qi in range(2):
                assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"rep{rep} q{qi}")
            assert {k_: st[k_] for k_ in ost} == ost


@pytest.mark.parametrize("m,metric,quant,d", [(16, O.COSINE, O.Q_NONE, 96), (16, O.L2, O.Q_F16, 77), (4, O.COSINE, O.Q_NONE, 48), (24, O.L2, O.Q_NONE, 48),
                                              (32, O.COSINE, O.Q_BF16, 40), (16, O.L2, O.Q_F8, 33)])
@pytest.mark.parametrize("seq", ["0", "1"], ids=["pipelined", "sequential"])
def test_latency_kernel_equals_oracle(gpu, monkeypatch, m, metric, quant, d, seq):
    """The 256-thread latency kernel (hnsw_lat.hpp): rows of one chunk take the walk that is software-pipelined over expansions
    (search_level_lat2), wider rows (mMax0 = 48, 64) the sequential one; dims with a scalar tail, 1-/2-/4-byte rows; small ef where
    the speculatively chosen next candidate can be truncated away; single queries and batches; k > ef; counters equal the oracle's."""
    monkeypatch.setenv("COLTT_LAT_MAX_NQ", "64"); monkeypatch.setenv("COLTT_WALK2", "off"); monkeypatch.setenv("COLTT_LAT_SEQ", seq)
    n = 5000
    X = O.fill_normal(3600 + m + d, (n, d)); lv = O.levels(3601 + m, n, m)
    gh = _gpu_build(gpu, X, lv, metric, quant, gpu.HnswCfg.default(m=m, ef_construction=48), batch=128)
    Q = O.fill_normal(3602 + d, (40, d))
    g = gh.ExportRaw(); rows = gh.FetchRows()
    for nq, k, efs in ((1, 10, (10, 64, 128)), (40, 1, (1, 2, 3, 7)), (40, 10, (20, 100)), (8, 50, (50, 120))):   # Hnsw.Search walks with max(ef, k) (hnsw.go:258)
        for ef in efs:
            gi, gs, gc, st = gh.Search(Q[:nq], k, ef=ef, with_stats=True)
            sl, sc, cn, ost, _ = O.csr_search(rows, quant, g["adj0"], g["upper_off"], g["adjU"], d, metric, g["entry"], g["entry_level"], Q[:nq], k, ef, threads=4)
            for qi in range(nq):
                assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"nq{nq} k{k} ef{ef} q{qi}")
            assert {k_: st[k_] for k_ in ost} == ost, (nq, k, ef, st, ost)


@pytest.mark.parametrize("lds_walk", ["off", "4"], ids=["round2-kernel", "walk2-lds"])
@pytest.mark.parametrize("metric,quant,n,d", [(O.COSINE, O.Q_NONE, 5000, 96), (O.L2, O.Q_NONE, 4000, 20), (O.COSINE, O.Q_F16, 5000, 77),
                                              (O.COSINE, O.Q_BF16, 2500, 768), (O.L2, O.Q_F8, 3000, 40)])
def test_small_ef_walk_over_the_lds_hash_equals_oracle(gpu, monkeypatch, lds_walk, metric, quant, n, d):
    """ef <= 128 (the LDS visited hash): the walk of hnsw_walk2.hpp (delta result set, adjacency-carried norms; COLTT_WALK2_LDS,
    the default) and hnsw_dev.hpp:search_level give the oracle's ids, score bits and traversal counters."""
    monkeypatch.setenv("COLTT_WALK2_LDS", lds_walk)
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    X = O.fill_normal(3600 + d, (n, d)); lv = O.levels(3601 + d, n)
    gh = _gpu_build(gpu, X, lv, metric, quant, gpu.HnswCfg.default(ef_construction=60), batch=256)
    Q = O.fill_normal(3602 + d, (48, d))
    _check(gh, Q, quant, metric, (1, 2, 5), k=1)
    _check(gh, Q, quant, metric, (10, 63, 64, 65, 100, 128))
    _check(gh, Q[:8], quant, metric, (128,), k=128)


def test_lds_hash_overflow_reruns_on_the_resetting_kernel(gpu, monkeypatch):
    """COLTT_VISG=0 keeps ef = 4096 on the LDS hash: the walk2 kernel gives up when the table would need a reset (err 8) and the
    call is served by search_level's reset-and-reseed path — same answers as the oracle, resets reported."""
    monkeypatch.setenv("COLTT_VISG", "0")
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    n, d = 40000, 8
    X = O.fill_normal(3700, (n, d)); lv = O.levels(3701, n)
    gh = _gpu_build(gpu, X, lv, O.L2, O.Q_NONE, gpu.HnswCfg.default(ef_construction=40), batch=2048)
    g = gh.ExportRaw(); rows = gh.FetchRows()
    Q = O.fill_normal(3702, (6, d))
    for pol in ("4", "off"):
        monkeypatch.setenv("COLTT_WALK2_LDS", pol)
        gi, gs, gc, st = gh.Search(Q, 10, ef=4096, with_stats=True)
        sl, sc, cn, ost, _ = O.csr_search(rows, O.Q_NONE, g["adj0"], g["upper_off"], g["adjU"], d, O.L2, g["entry"], g["entry_level"], Q, 10, 4096, threads=4)
        for qi in range(len(Q)):
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"{pol} q{qi}")
        assert st["n_visit_resets"] > 0, (pol, st)


@pytest.mark.parametrize("m", [24, 32])
def test_two_chunk_rows_at_small_ef(gpu, monkeypatch, m):
    """mMax0 = 48 / 64 with ef <= 128: two 32-neighbour chunks per expansion go through the small-set merge of hnsw_walk2.hpp (the
    free slots and the stale lowerBound carry from the first chunk to the second, hnsw.go:357,374) — both kernels, oracle's counters."""
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    n, d = 3000, 32
    X = O.fill_normal(3800 + m, (n, d)); lv = O.levels(3801 + m, n, m)
    gh = _gpu_build(gpu, X, lv, O.COSINE, O.Q_NONE, gpu.HnswCfg.default(m=m, ef_construction=48), batch=128)
    assert gh.cfg.m_max0 == 2 * m
    Q = O.fill_normal(3802, (32, d))
    for pol in ("4", "off"):
        monkeypatch.setenv("COLTT_WALK2_LDS", pol)
        _check(gh, Q, O.Q_NONE, O.COSINE, (3,), k=1)          # Hnsw.Search walks with max(ef, k) (hnsw.go:258)
        _check(gh, Q, O.Q_NONE, O.COSINE, (10, 64, 128))
