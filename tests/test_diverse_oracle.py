"""COLTT_HNSW_DIVERSE (algo 2) is a DEFINITION, not reference behaviour (the reference's `selectNeighborsHeuristic`, hnsw.go:399-447, has
no diversity test).  Double entry: the C++ oracle (`select_diverse`, canonical forms) and an independent pure-Python statement
(`oracle/pyref.py: DiverseHnsw`, Go-heap searchLevel) must build the same graph — and the mode must leave algo 0 / 1 alone."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref as P


def _same_graph(a, b):
    for k in ("levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["nbr_dist"].view(np.uint32), b["nbr_dist"].view(np.uint32))
    assert a["entry"] == b["entry"]


@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
@pytest.mark.parametrize("keep,m,mmax0,efc", [(0, 16, -1, 40), (1, 16, -1, 40), (0, 4, 6, 24), (1, 5, 7, 30)])
def test_cpp_definition_equals_python_definition(metric, keep, m, mmax0, efc):
    n, d = 260, 12
    X = O.fill_normal(9200 + m, (n, d)); lv = O.levels(9201, n, m=m); ids = np.arange(n, dtype=np.uint64) + np.uint64(7)
    oh = O.Hnsw(d, metric, O.default_cfg(m=m, mMax0=mmax0, efConstruction=efc, algo=2, keepPruned=keep))
    ph = P.DiverseHnsw(d, P.Hnsw.COSINE if metric == O.COSINE else P.Hnsw.L2, keep_pruned=bool(keep), m=m, m_max0=mmax0, ef_construction=efc)
    rng = np.random.default_rng(9202)
    for i in range(n):
        assert oh.insert(ids[i], X[i], lv[i]) == 0
        assert ph.insert(int(ids[i]), X[i], int(lv[i])) is None
        if i > 50 and rng.random() < 0.15:      # Removes in between: the re-prune drops tombstones only
            v = int(rng.integers(0, i))
            a = oh.remove(ids[v]); b = ph.remove(int(ids[v]))
            assert (a == 0) == (b is None)
    _same_graph(oh.export(with_vectors=False), ph.export())
    Q = O.fill_normal(9203, (10, d))
    for q in Q:
        wi, ws = oh.search(q, 5, mode=1, ef=30)
        pr = ph.search(q, 5, ef_override=30)
        assert [int(i) for i in wi] == [i for i, _ in pr]
        assert np.array_equal(ws.view(np.uint32), np.array([s for _, s in pr], np.float32).view(np.uint32))


def test_batch_of_one_is_the_sequential_insert_and_batches_differ_only_by_the_frozen_graph():
    n, d = 500, 16
    X = O.fill_normal(9210, (n, d)); lv = O.levels(9211, n); ids = np.arange(n, dtype=np.uint64)
    cfg = lambda: O.default_cfg(efConstruction=32, algo=2, keepPruned=0)
    a = O.Hnsw(d, O.L2, cfg()); a.insert_many(ids, X, lv)
    b = O.Hnsw(d, O.L2, cfg()); b.insert_batched(ids, X, lv, 1)
    _same_graph(a.export(with_vectors=False), b.export(with_vectors=False))
    c = O.Hnsw(d, O.L2, cfg()); c.insert_batched(ids, X, lv, 0, schedule=lambda i: max(1, min(64, i // 8)))
    assert np.array_equal(c.export(with_vectors=False)["levels"], a.export(with_vectors=False)["levels"])


def test_the_mode_changes_the_graph_and_keeps_rows_within_their_width():
    n, d = 600, 8    # low dimension: the diversity test rejects a lot
    X = O.fill_normal(9220, (n, d)); lv = O.levels(9221, n); ids = np.arange(n, dtype=np.uint64)
    base = O.Hnsw(d, O.L2, O.default_cfg(), canonical_build=True); base.insert_many(ids, X, lv)
    div = O.Hnsw(d, O.L2, O.default_cfg(algo=2, keepPruned=0)); div.insert_many(ids, X, lv)
    gb, gd = base.export(with_vectors=False), div.export(with_vectors=False)
    assert np.diff(gd["row_offsets"]).max() <= 32
    assert np.diff(gd["row_offsets"]).mean() < np.diff(gb["row_offsets"]).mean()      # fewer, more spread-out edges
