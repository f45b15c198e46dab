"""COLTT_HNSW_DIVERSE (algo 2) — an OPT-IN neighbour selection the reference does not have (its `selectNeighborsHeuristic`,
core/vectorindex/hnsw.go:399-447, keeps the k nearest without a diversity test).  The definition is the oracle's `select_diverse`
(oracle/coltt_oracle.cpp; second statement: oracle/pyref.py); the GPU builder must produce THAT graph bit for bit: levels, every edge
list, the stored edge distances, the entrypoint — sequentially (batch = 1), in batches, after Removes, for line-transposed rows and for
2-byte rows.  The default modes (0 / 1) are untouched: their tests live in test_gpu_hnsw.py."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu

DIVERSE = 2


def _graph_equal(a, b):
    for k in ("levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["nbr_dist"].view(np.uint32), b["nbr_dist"].view(np.uint32)), "edge distances"
    assert a["entry"] == b["entry"]


def _gpu_build(gpu, gh, X, lv, sched, ids=None, first_id=0):
    import torch
    n, d = X.shape
    xd = torch.from_numpy(np.ascontiguousarray(X)).cuda(); torch.cuda.synchronize()
    i = 0
    while i < n:
        b = min(sched(i), n - i)
        if ids is None:
            gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=first_id + i)
        else:
            gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, ids=ids[i:i + b])
        i += b


@pytest.mark.parametrize("keep", [0, 1])
@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
def test_diverse_sequential_build_equals_the_definition(gpu, metric, keep):
    n, d = 700, 48
    X = O.fill_normal(9061, (n, d)); lv = O.levels(9062, n); ids = np.arange(n, dtype=np.uint64) + np.uint64(500)
    oh = O.Hnsw(d, metric, O.default_cfg(algo=DIVERSE, keepPruned=keep)); oh.insert_many(ids, X, lv)
    gh = gpu.Hnsw(d, metric, gpu.HnswCfg.default(algo=DIVERSE, keep_pruned=keep))
    _gpu_build(gpu, gh, X, lv, lambda i: 1, ids=ids)
    go = gh.Export(); oo = oh.export(with_vectors=False)
    assert np.array_equal(go["ids"], oo["ids"])
    _graph_equal(go, oo)
    # the selection really differs from the k-nearest graph of the default modes (otherwise this file tests nothing)
    ref = O.Hnsw(d, metric, O.default_cfg(), canonical_build=True); ref.insert_many(ids, X, lv)
    assert not np.array_equal(ref.export(with_vectors=False)["nbr"], oo["nbr"])


@pytest.mark.parametrize("keep,m,mmax0", [(0, 16, -1), (1, 16, -1), (0, 4, 8), (1, 5, 7)])
def test_diverse_batched_build_equals_the_definition(gpu, keep, m, mmax0):
    """batch > 1: every vertex of a batch searches the graph as it was before the batch; a row receives ALL the batch's links and
    is pruned ONCE if it overflows (the oracle's hnsw_insert_batch states the rule).  Narrow rows (m 4 / mMax0 8) prune all the time."""
    n, d = 4000, 64
    X = O.fill_normal(9071, (n, d)); lv = O.levels(9072, n, m=m); ids = np.arange(n, dtype=np.uint64)
    sched = lambda i: max(1, min(256, i // 16))
    oh = O.Hnsw(d, O.L2, O.default_cfg(m=m, mMax0=mmax0, efConstruction=64, algo=DIVERSE, keepPruned=keep))
    oh.insert_batched(ids, X, lv, 0, schedule=sched)
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(m=m, m_max0=mmax0, ef_construction=64, algo=DIVERSE, keep_pruned=keep))
    _gpu_build(gpu, gh, X, lv, sched)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    Q = O.fill_normal(9073, (64, d))
    gi, gs, gc = gh.Search(Q, 10, ef=128)
    fl = gpu.FlatSpace(d, O.L2); fl.ChangedVertex(ids, X)
    ti, ts, tc = fl.VertexSearch(Q, 10, gpu.SELECT_NEAREST)
    rec = np.mean([len(set(gi[q]) & set(ti[q])) / 10 for q in range(len(Q))])
    assert rec > (0.9 if m >= 16 else 0.5), rec


def test_diverse_line_transposed_rows(gpu):
    """768 x f32 rows are stored line-transposed (rows8.hpp): the candidate under test is decoded back into natural order."""
    n, d = 1500, 768
    X = O.fill_normal(9081, (n, d)); lv = O.levels(9082, n); ids = np.arange(n, dtype=np.uint64)
    sched = lambda i: max(1, min(128, i // 16))
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(efConstruction=48, algo=DIVERSE, keepPruned=0)); oh.insert_batched(ids, X, lv, 0, schedule=sched)
    gh = gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(ef_construction=48, algo=DIVERSE, keep_pruned=0))
    _gpu_build(gpu, gh, X, lv, sched)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))


@pytest.mark.parametrize("d", [256, 72])
def test_diverse_two_byte_rows(gpu, d):
    """binary16 rows (Euclidean: stored = binary16(x), so the oracle fed the decoded rows holds the same vectors); d 256 is a
    line-transposed shape, d 72 a natural-order one with a ragged tail."""
    n = 2500
    X = O.fill_normal(9091, (n, d)); lv = O.levels(9092, n); ids = np.arange(n, dtype=np.uint64)
    X16 = O.f16_decode(O.f16_encode(X)).reshape(n, d)
    sched = lambda i: max(1, min(200, i // 16))
    oh = O.Hnsw(d, O.L2, O.default_cfg(efConstruction=64, algo=DIVERSE, keepPruned=0)); oh.insert_batched(ids, X16, lv, 0, schedule=sched)
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(ef_construction=64, algo=DIVERSE, keep_pruned=0), quantization=gpu.Q_F16)
    _gpu_build(gpu, gh, X, lv, sched)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))


def test_diverse_remove_then_insert(gpu):
    """Remove's re-prune only drops tombstones (the selection runs on overflow only); a later Insert whose link overflows a row with
    tombstoned entries drops them first and selects among the live ones."""
    n, d = 600, 32
    X = O.fill_normal(9101, (n, d)); lv = O.levels(9102, n); ids = np.arange(n, dtype=np.uint64)
    cfg = dict(m=6, mMax0=10, efConstruction=40, algo=DIVERSE, keepPruned=0)
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(**cfg)); oh.insert_many(ids, X, lv)
    gh = gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(m=6, m_max0=10, ef_construction=40, algo=DIVERSE, keep_pruned=0))
    _gpu_build(gpu, gh, X, lv, lambda i: 1)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    rng = np.random.default_rng(9103)
    for v in rng.choice(n, 150, replace=False):
        assert oh.remove(ids[v]) == 0
        gh.Remove(ids[v])
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    Y = O.fill_normal(9104, (120, d)); ly = O.levels(9105, 120); nid = np.arange(120, dtype=np.uint64) + np.uint64(10_000)
    for i in range(120): assert oh.insert(nid[i], Y[i], ly[i]) == 0
    _gpu_build(gpu, gh, Y, ly, lambda i: 1, ids=nid)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))
    # the search over the diverse graph is the unchanged Hnsw.Search: ids, score bits, counters == the oracle's walk of the same graph
    Q = O.fill_normal(9106, (24, d))
    gi, gs, gc, st = gh.Search(Q, 10, ef=64, with_stats=True)
    tot = {"n_dist": 0, "n_exp": 0, "n_hops": 0}
    for qi in range(len(Q)):
        wi, ws, s = oh.search(Q[qi], 10, mode=1, ef=64, with_stats=True)
        assert np.array_equal(gi[qi, :gc[qi]], wi) and np.array_equal(gs[qi, :gc[qi]].view(np.uint32), ws.view(np.uint32)), qi
        for k in tot: tot[k] += s[k]
    assert {k: st[k] for k in tot} == tot


def test_diverse_commit_load_and_config(gpu):
    n, d = 400, 24
    X = O.fill_normal(9111, (n, d)); lv = O.levels(9112, n)
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(algo=DIVERSE, keep_pruned=0, ef_construction=32))
    _gpu_build(gpu, gh, X, lv, lambda i: max(1, min(64, i // 8)))
    assert gh.cfg.algo == DIVERSE
    blob = gh.Commit()
    g2 = gpu.Hnsw(d, O.L2); g2.Load(blob)
    assert g2.Config().algo == DIVERSE   # (the reference's stream carries no keepPruned / extendCandidates: hnsw_config.go:179-203)
    assert g2.Len() == n
    Q = O.fill_normal(9113, (16, d))                         # (slots are renumbered in stream order: compare what a caller sees)
    a = gh.Search(Q, 10, ef=64); b = g2.Search(Q, 10, ef=64)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2])
    c1 = g2.Commit(); g3 = gpu.Hnsw(d, O.L2); g3.Load(c1)
    assert g3.Commit() == c1 and g3.Config().algo == DIVERSE   # the stream is a fixed point of Commit(Load(.))
    with pytest.raises(gpu.ColttError):
        gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(algo=DIVERSE, extend_candidates=1))
    with pytest.raises(gpu.ColttError):
        gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(algo=3))


def test_diverse_batch_that_would_overflow_a_row_is_retried_in_halves(gpu):
    """One vertex in the graph, 1 100 new ones in ONE batch: all of them link to it — more candidates than the link kernel holds (1 024).  The builder finds
    that BEFORE anything is applied and retries the batch in halves (a smaller batch is another legal schedule of the same Inserts): the graph equals the
    oracle's for the schedule 1, 550, 550 — never a truncated candidate list, never a half-applied batch."""
    d = 16
    X = O.fill_normal(9121, (1101, d)); lv = np.zeros(1101, np.int32); ids = np.arange(1101, dtype=np.uint64)
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(algo=DIVERSE, keep_pruned=0))
    _gpu_build(gpu, gh, X[:1], lv[:1], lambda i: 1)
    _gpu_build(gpu, gh, X[1:], lv[1:], lambda i: 1100, first_id=1)
    assert gh.Len() == 1101
    oh = O.Hnsw(d, O.L2, O.default_cfg(algo=DIVERSE, keepPruned=0))
    oh.insert_batched(ids, X, lv, 0, schedule=lambda i: 1 if i == 0 else 550)
    _graph_equal(gh.Export(), oh.export(with_vectors=False))


@pytest.mark.parametrize("ci", [0, 1])
def test_diverse_graph_equals_the_committed_golden_vectors(gpu, ci):
    """tests/golden/round6_definitions.npz (written by oracle/pyref.py: DiverseHnsw, pure Python): inserts with interleaved Removes, cosine + narrow rows and
    Euclidean + keepPruned — the GPU builder (batch = 1) must end at the committed graph: levels, tombstones, edge lists, stored edge distances, entrypoint."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round6_definitions.npz"))
    g = lambda k: z[f"d{ci}_{k}"]
    d, metric, m, mmax0, efc, keep = (int(v) for v in g("cfg"))
    gh = gpu.Hnsw(d, O.COSINE if metric == 0 else O.L2, gpu.HnswCfg.default(m=m, m_max0=mmax0, ef_construction=efc, algo=DIVERSE, keep_pruned=keep))
    X, ids, lv = g("X"), g("ids"), g("levels")
    rem = {int(a): int(b) for a, b in g("removes")}
    for i in range(len(X)):
        gh.Insert(ids[i], X[i], lv[i])
        if i in rem:
            gh.Remove(ids[rem[i]])
    e = gh.Export()
    for k in ("levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(e[k], g("g_" + k)), k
    assert np.array_equal(e["nbr_dist"].view(np.uint32), g("g_nbr_dist").view(np.uint32)) and e["entry"] == int(g("g_entry"))
