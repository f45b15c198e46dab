"""HNSW search on the GPU vs the oracle on the oracle-built graph: identical ids, ranks, score bits AND identical
traversal counters."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results

pytestmark = pytest.mark.gpu


def oracle_index(n, d, metric, seed, cfg=None, remove=0):
    X = O.fill_normal(seed, (n, d)); lv = O.levels(seed + 1, n)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(11)
    h = O.Hnsw(d, metric, cfg or O.default_cfg())
    h.insert_many(ids, X, lv)
    rng = np.random.default_rng(seed)
    for i in rng.choice(n, remove, replace=False):
        assert h.remove(ids[i]) == 0
    return X, ids, h


@pytest.fixture(params=["0", "128"], ids=["one-wave-per-query", "four-waves-per-query"])
def mw(request, monkeypatch):
    """COLTT_MW_MAX_NQ: batches up to that size take the multi-wave (latency) kernel; 0 = always one wave per query.  Both kernels
    must give the oracle's ids, score bits and traversal counters."""
    monkeypatch.setenv("COLTT_MW_MAX_NQ", request.param)
    return request.param


@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
@pytest.mark.parametrize("n,d", [(1000, 128), (3000, 64), (1500, 768)])
def test_hnsw_search_parity(gpu, mw, metric, n, d):
    X, ids, oh = oracle_index(n, d, metric, seed=40 + d)
    g = oh.export(with_vectors=False)
    gh = gpu.Hnsw(d, metric)
    gh.BulkLoad(g, X)
    assert gh.Len() == n
    Q = O.fill_normal(123, (40, d))
    for ef in (20, 128):
        gi, gs, gc, st = gh.Search(Q, 10, ef=ef, with_stats=True)
        tot = {"n_dist": 0, "n_exp": 0, "n_hops": 0}
        for qi in range(len(Q)):
            wi, ws, s = oh.search(Q[qi], 10, mode=1, ef=ef, with_stats=True)
            li, ls = oh.search(Q[qi], 10, mode=0, ef=ef)  # literal Go-heap restatement agrees with the canonical form
            assert_same_results(wi, ws, li, ls, "oracle literal vs canonical")
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi} ef{ef}")
            for k in tot: tot[k] += s[k]
        assert st["n_visit_resets"] == 0
        assert {k: st[k] for k in tot} == tot, (st, tot)


def test_hnsw_with_removed_vertices_and_small_k(gpu, mw):
    n, d = 1200, 32
    X, ids, oh = oracle_index(n, d, O.COSINE, seed=7, remove=200)
    gh = gpu.Hnsw(d, O.COSINE)
    gh.BulkLoad(oh.export(with_vectors=False), X)
    assert gh.Len() == len(oh) == n - 200
    Q = O.fill_normal(5, (30, d))
    for k, ef in ((1, 20), (10, 64), (50, 20)):  # ef = max(ef, k)
        gi, gs, gc = gh.Search(Q, k, ef=ef)
        for qi in range(len(Q)):
            wi, ws = oh.search(Q[qi], k, mode=1, ef=ef)
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi} k{k}")


def test_hnsw_empty_and_tiny(gpu):
    d = 16
    gh = gpu.Hnsw(d, O.L2)
    _, _, c = gh.Search(np.zeros((3, d), np.float32), 5)
    assert (c == 0).all()  # empty index => empty result (hnsw.go:249-251)
    X, ids, oh = oracle_index(3, d, O.L2, seed=1)
    gh.BulkLoad(oh.export(with_vectors=False), X)
    gi, gs, gc = gh.Search(X, 5)
    for qi in range(3):
        wi, ws = oh.search(X[qi], 5, mode=1)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws)


@pytest.mark.parametrize("quant,d", [(O.Q_F16, 64), (O.Q_F8, 64), (O.Q_F16, 24), (O.Q_F16, 77), (O.Q_F16, 784), (O.Q_F8, 77)])
def test_hnsw_quantised_rows(gpu, mw, quant, d):
    """2-/1-byte stored codes (BASELINE configs[4]): distances as the edge quantised stores compute them —
    decode(query') vs decode(row).  Oracle: an f32 index over the decoded vectors gives the same arithmetic.
    Dims cover the wide 16-byte walk of 2-byte rows: even / odd 8-element group counts (64 / 24), odd + scalar tail (77),
    full bursts + remainder (784 = 49 group pairs)."""
    n = 800
    X = O.fill_normal(3, (n, d)); lv = O.levels(4, n); ids = np.arange(n, dtype=np.uint64)
    Xs = O.f16_decode(O.lower(quant, X)) if quant != O.Q_F8 else O.f8_decode(O.lower(quant, X))
    oh = O.Hnsw(d, O.L2); oh.insert_many(ids, Xs, lv)
    gh = gpu.Hnsw(d, O.L2, quantization=quant); gh.BulkLoad(oh.export(with_vectors=False), X)
    Q = O.fill_normal(8, (20, d))
    Qs = O.f16_decode(O.lower(quant, Q)) if quant != O.Q_F8 else O.f8_decode(O.lower(quant, Q))
    gi, gs, gc = gh.Search(Q, 10, ef=64)
    for qi in range(20):
        wi, ws = oh.search(Qs[qi], 10, mode=1, ef=64)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws)


# ---------------------------------------------------------------------------------------------- builder
def _graph_equal(a, b):
    for k in ("levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["nbr_dist"].view(np.uint32), b["nbr_dist"].view(np.uint32)), "edge distances"
    assert a["entry"] == b["entry"]


@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
def test_hnsw_gpu_build_sequential_equals_reference_insert(gpu, metric):
    """batch == 1: the GPU builder is Hnsw.Insert; the graph (edges AND stored distances) equals the oracle's."""
    import torch
    n, d = 700, 48
    X = O.fill_normal(61, (n, d)); lv = O.levels(62, n); ids = np.arange(n, dtype=np.uint64) + np.uint64(500)
    oh = O.Hnsw(d, metric); oh.insert_many(ids, X, lv)          # literal restatement (Go heaps)
    gh = gpu.Hnsw(d, metric)
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    gh.InsertBatchDevice(xd.data_ptr(), n, lv, batch=1, ids=ids)
    go = gh.Export(); oo = oh.export(with_vectors=False)
    assert np.array_equal(go["ids"], oo["ids"])
    _graph_equal(go, oo)
    # single host-side Insert + duplicate id
    v = O.fill_normal(63, d)
    gh.Insert(10**9, v, 1); assert oh.insert(10**9, v, 1) == 0
    with pytest.raises(gpu.ColttError) as e:
        gh.Insert(10**9, v, 0)
    assert e.value.code == -2  # ItemAlreadyExistsError
    _graph_equal(gh.Export(), oh.export(with_vectors=False))


def test_hnsw_gpu_build_batched(gpu):
    """batch > 1: equals the oracle's batched-insert restatement; recall is checked against exact search."""
    import torch
    n, d = 4000, 64
    X = O.fill_normal(71, (n, d)); lv = O.levels(72, n); ids = np.arange(n, dtype=np.uint64)
    sched = lambda i: max(1, min(256, i // 16))
    oh = O.Hnsw(d, O.L2, O.default_cfg(efConstruction=64)); oh.insert_batched(ids, X, lv, 0, schedule=sched)
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(ef_construction=64))
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    i = 0
    while i < n:
        b = min(sched(i), n - i)
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i)
        i += b
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    Q = O.fill_normal(73, (64, d))
    gi, gs, gc = gh.Search(Q, 10, ef=128)
    fl = gpu.FlatSpace(d, O.L2); fl.ChangedVertex(ids, X)
    ti, ts, tc = fl.VertexSearch(Q, 10, gpu.SELECT_NEAREST)
    rec = np.mean([len(set(gi[q]) & set(ti[q])) / 10 for q in range(len(Q))])
    assert rec > 0.9, rec


def test_hnsw_gpu_build_bench_schedule_100k(gpu):
    """The builder exactly as bench.py drives it — cosine, batches growing to 1/32 of the graph (3 125 vertices here, the 16 384
    cap is the same code path) — on 100 000 x 32: the whole graph (levels, every edge list, stored edge distances, entrypoint)
    equals the oracle's batched-insert restatement bit for bit.  (The 10 M bench graph itself is validated by search parity on
    the exported arrays; this is build parity at the largest size the CPU oracle builds in about half a minute.)"""
    import torch
    n, d = 100_000, 32
    X = O.fill_normal(171, (n, d)); lv = O.levels(172, n); ids = np.arange(n, dtype=np.uint64)
    sched = lambda i: max(1, min(16384, i // 32))
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(efConstruction=64)); oh.insert_batched(ids, X, lv, 0, schedule=sched)
    gh = gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(ef_construction=64))
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    i = 0
    while i < n:
        b = min(sched(i), n - i)
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i)
        i += b
    _graph_equal(gh.Export(), oh.export(with_vectors=False))


def test_hnsw_remove_parity(gpu):
    import torch
    n, d = 500, 32
    X = O.fill_normal(81, (n, d)); lv = O.levels(82, n); ids = np.arange(n, dtype=np.uint64)
    oh = O.Hnsw(d, O.COSINE); oh.insert_many(ids, X, lv)
    gh = gpu.Hnsw(d, O.COSINE); gh.BulkLoad(oh.export(with_vectors=False), X)
    rng = np.random.default_rng(5)
    victims = list(rng.choice(n, 120, replace=False))
    ent = oh.export(with_vectors=False)["entry"]
    if ent not in victims: victims.insert(3, ent)  # removing the entrypoint re-elects one (hnsw.go:197-217)
    for v in victims:
        assert oh.remove(ids[v]) == 0
        gh.Remove(ids[v])
    with pytest.raises(gpu.ColttError) as e:
        gh.Remove(ids[victims[0]])
    assert e.value.code == -3  # ItemNotFoundError
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    assert gh.Len() == len(oh)
    Q = O.fill_normal(83, (20, d))
    gi, gs, gc = gh.Search(Q, 10, ef=50)
    for qi in range(20):
        wi, ws = oh.search(Q[qi], 10, mode=1, ef=50)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws)
    # inserts after removals keep matching
    Y = O.fill_normal(84, (50, d)); ly = O.levels(85, 50)
    yd = torch.from_numpy(Y).cuda(); torch.cuda.synchronize()
    nid = np.arange(50, dtype=np.uint64) + np.uint64(10000)
    for i in range(50): assert oh.insert(nid[i], Y[i], ly[i]) == 0
    gh.InsertBatchDevice(yd.data_ptr(), 50, ly, batch=1, ids=nid)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))


def test_concurrent_callers_thread_safety(gpu):
    """cgo calls arrive on arbitrary OS threads: hammer one HNSW handle and one FLAT handle from 8 threads at once and
    require every answer to equal the single-threaded answer."""
    from concurrent.futures import ThreadPoolExecutor
    n, d = 1500, 48
    X, ids, oh = oracle_index(n, d, O.L2, seed=91)
    gh = gpu.Hnsw(d, O.L2); gh.BulkLoad(oh.export(with_vectors=False), X)
    gf = gpu.FlatSpace(d, O.L2); gf.ChangedVertex(ids, X)
    Q = O.fill_normal(92, (64, d))
    want_h = gh.Search(Q, 10, ef=40); want_f = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST)

    def work(t):
        for it in range(6):
            lo = (t * 8) % 64
            a = gh.Search(Q[lo:lo + 8], 10, ef=40); b = gf.VertexSearch(Q[lo:lo + 8], 10, gpu.SELECT_NEAREST)
            assert np.array_equal(a[0], want_h[0][lo:lo + 8]) and np.array_equal(a[1].view(np.uint32), want_h[1][lo:lo + 8].view(np.uint32))
            assert np.array_equal(b[0], want_f[0][lo:lo + 8]) and np.array_equal(b[1].view(np.uint32), want_f[1][lo:lo + 8].view(np.uint32))
        return True
    with ThreadPoolExecutor(8) as ex:
        assert all(ex.map(work, range(8)))


def test_hnsw_commit_load_streams(gpu):
    """Hnsw.Commit / Hnsw.Load (hnsw_commit.go:69-278): the reference's own Commit->Load invariant (hnsw_commit_test.go:127-181:
    structure, vectors and edge distances survive the round trip), here ACROSS implementations: oracle stream -> GPU index,
    GPU stream -> oracle index, with searches bit-identical on both sides."""
    import torch
    n, d = 900, 40
    X = O.fill_normal(101, (n, d)); lv = O.levels(102, n); ids = np.arange(n, dtype=np.uint64) * np.uint64(11) + np.uint64(5)
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(ef=33)); oh.insert_many(ids, X, lv)
    rng = np.random.default_rng(7)
    for i in rng.choice(n, 150, replace=False): oh.remove(ids[i])      # 20 % removals, as the reference test does
    stream = oh.commit(header=True)
    gh = gpu.Hnsw(d, O.COSINE)
    assert gh.Load(stream, header=True) == len(oh) == n - 150
    assert gh.cfg.ef == 33 and gh.Len() == n - 150
    Q = O.fill_normal(103, (25, d))
    gi, gs, gc = gh.Search(Q, 10, ef=60)
    o2 = O.Hnsw(d, O.COSINE); assert o2.load_stream(stream) == 0          # same slot order as the GPU (stream order)
    for qi in range(25):
        wi, ws = o2.search(Q[qi], 10, mode=1, ef=60)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi}")
    # GPU -> stream -> oracle, and the stream is a fixed point of Commit(Load(.))
    s2 = gh.Commit(header=True)
    o3 = O.Hnsw(d, O.L2); assert o3.load_stream(s2) == 0
    assert o3.graph_hash() == o2.graph_hash()
    assert o2.commit(header=True) == s2
    # a GPU-built index commits to a stream the oracle loads into the same graph
    g2 = gpu.Hnsw(d, O.L2); xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    g2.InsertBatchDevice(xd.data_ptr(), n, lv, batch=1, ids=ids)
    o4 = O.Hnsw(d, O.L2); o4.insert_many(ids, X, lv)
    o5 = O.Hnsw(d, O.L2); assert o5.load_stream(g2.Commit()) == 0
    o6 = O.Hnsw(d, O.L2); assert o6.load_stream(o4.commit()) == 0
    assert o5.graph_hash() == o6.graph_hash()
    # malformed input is an error, not a crash
    with pytest.raises(gpu.ColttError):
        gh.Load(stream[:len(stream) // 2])
    with pytest.raises(gpu.ColttError):
        gpu.Hnsw(d + 8, O.COSINE).Load(stream)
    e = gpu.Hnsw(d, O.COSINE); assert e.Load(e.Commit()) == 0             # empty index round trip


def test_hnsw_ragged_dim_large_ef_and_k_above_ef(gpu, mw):
    """dim not a multiple of 8 (scalar tail of the AVX kernels), ef = 1024 (largest LDS geometry), k > cfg.ef (ef = max(ef, k))."""
    n, d = 2500, 20
    X, ids, oh = oracle_index(n, d, O.COSINE, seed=111)
    gh = gpu.Hnsw(d, O.COSINE); gh.BulkLoad(oh.export(with_vectors=False), X)
    Q = O.fill_normal(112, (12, d))
    for k, ef in ((10, 1024), (300, 20), (5, 7)):
        gi, gs, gc, st = gh.Search(Q, k, ef=ef, with_stats=True)
        assert st["n_visit_resets"] == 0
        for qi in range(len(Q)):
            wi, ws = oh.search(Q[qi], k, mode=1, ef=ef)
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi} k{k} ef{ef}")


def test_hnsw_visited_set_reset_keeps_results_exact(gpu, monkeypatch):
    """ef = 4096 (the maximum) on a small dense graph with the LDS visited set forced (COLTT_VISG=0; above ef 128 the default is the HBM byte
    map): the bounded hash (16384 slots beside the 32 KB result set) is reset-and-reseeded; results stay exact."""
    monkeypatch.setenv("COLTT_VISG", "0")
    n, d = 60000, 8
    X = O.fill_normal(121, (n, d)); lv = O.levels(122, n); ids = np.arange(n, dtype=np.uint64)
    import torch
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(ef_construction=40))
    i = 0
    while i < n:
        b = max(1, min(2048, i // 16)); b = min(b, n - i)
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i); i += b
    g = gh.Export(); g["vectors"] = X
    oh = O.Hnsw(d, O.L2, O.default_cfg(efConstruction=40)); oh.load(g)
    Q = O.fill_normal(123, (6, d))
    gi, gs, gc, st = gh.Search(Q, 10, ef=4096, with_stats=True)
    for qi in range(len(Q)):
        wi, ws = oh.search(Q[qi], 10, mode=1, ef=4096)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi}")
    assert st["n_visit_resets"] > 0, st


def test_hnsw_hbm_visited_set_epoch_wrap(gpu, monkeypatch):
    """HBM byte-per-slot visited set forced at small ef (COLTT_VISG=1): the batched builder's graph, the answers and the
    traversal counters equal the oracle's, and keep doing so after a workgroup's 8-bit epoch has wrapped (> 255 traversals by
    the same workgroup => its region is wiped)."""
    import torch
    monkeypatch.setenv("COLTT_VISG", "1")
    n, d = 3000, 24
    X = O.fill_normal(131, (n, d)); lv = O.levels(132, n); ids = np.arange(n, dtype=np.uint64)
    sched = lambda i: max(1, min(8, i // 16))       # few workgroups, many traversals each: epochs wrap during the build
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(efConstruction=32)); oh.insert_batched(ids, X, lv, 0, schedule=sched)
    gh = gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(ef_construction=32))
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    i = 0
    while i < n:
        b = min(sched(i), n - i)
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i)
        i += b
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    Q = O.fill_normal(133, (2, d))
    for rep in range(300):                          # 2 workgroups x 300 launches on top of the build's epochs
        gi, gs, gc, st = gh.Search(Q, 10, ef=40, with_stats=True)
        if rep % 50 == 0 or rep == 299:
            for qi in range(len(Q)):
                wi, ws = oh.search(Q[qi], 10, mode=1, ef=40)
                assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"rep{rep} q{qi}")
            assert st["n_visit_resets"] == 0


def test_hnsw_random_level_matches_reference_formula(gpu):
    """coltt_hnsw_random_level == gomath.Floor(-gomath.Log(u) * levelMultiplier) (hnsw.go:280-282) for given draws."""
    gh = gpu.Hnsw(8, O.L2)
    mult = gh.cfg.level_multiplier
    rng = np.random.default_rng(7)
    us = np.concatenate([rng.random(2000, dtype=np.float32), np.float32([1e-30, 1e-7, 0.0624, 0.0625, 0.0626, 0.25, 0.999999])])
    for u in us:
        if not (0.0 < u < 1.0):
            continue
        assert gh.RandomLevel(float(u)) == O.level_from_u(float(u), mult), u
    for bad in (0.0, 1.0, -0.5, float("nan")):
        with pytest.raises(gpu.ColttError) as e:
            gh.RandomLevel(bad)
        assert e.value.code == -1
