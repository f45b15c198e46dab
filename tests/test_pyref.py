"""Two independent restatements of the reference's HNSW must agree bit for bit: oracle/coltt_oracle.cpp (C++, line-by-line)
and oracle/pyref.py (pure Python, written from the Go text).  A misreading of hnsw.go / priority_queue.go / avx.cpp in either
one shows up here.  Also pins both to tests/golden/hnsw_pyref.npz (generated from pyref)."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref as P
from util import bits

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hnsw_pyref.npz")


def _replay_cpp(c, mode):
    d, metric, m, efc, algo, k, ef = (int(v) for v in c["cfg"])
    h = O.Hnsw(d, metric, O.default_cfg(m=m, efConstruction=efc, algo=algo), canonical_build=(mode == "canonical"))
    h.insert_many(c["ids"], c["X"], c["levels"])
    for i in c["removed"]:
        assert h.remove(int(i)) == 0
    h.insert_many(c["y_ids"], c["Y"], c["y_levels"])
    return h, (d, metric, m, efc, algo, k, ef)


def _cases():
    z = np.load(GOLD)
    for ci in range(int(z["n_cases"])):
        yield ci, {k[len(f"c{ci}_"):]: z[k] for k in z.files if k.startswith(f"c{ci}_")}


@pytest.mark.parametrize("mode", ["literal", "canonical"])
def test_cpp_oracle_reproduces_the_pyref_golden(mode):
    for ci, c in _cases():
        h, (d, metric, m, efc, algo, k, ef) = _replay_cpp(c, mode)
        g = h.export(with_vectors=False)
        assert np.array_equal(g["levels"], c["g_levels"]) and np.array_equal(g["deleted"], c["g_deleted"]), ci
        assert np.array_equal(g["row_offsets"], c["g_row_offsets"]) and np.array_equal(g["nbr"], c["g_nbr"]), ci
        assert np.array_equal(bits(g["nbr_dist"]), bits(c["g_nbr_dist"])) and g["entry"] == int(c["g_entry"]), ci
        for qi in range(len(c["Q"])):
            wi, ws = h.search(c["Q"][qi], k, mode=0 if mode == "literal" else 1, ef=ef)
            n = int(c["res_n"][qi])
            assert len(wi) == n and np.array_equal(wi, c["res_ids"][qi, :n]) and np.array_equal(bits(ws), bits(c["res_scores"][qi, :n])), (ci, qi)


def test_distance_and_normalize_restatements_agree():
    rng = np.random.default_rng(5)
    for d in (1, 7, 8, 9, 31, 32, 33, 128):
        for _ in range(20):
            a = rng.standard_normal(d).astype(np.float32); b = rng.standard_normal(d).astype(np.float32)
            assert bits(P.euclidean(a, b)) == bits(np.float32(O.l2(a, b)))
            assert bits(P.cosine(a, b)) == bits(np.float32(O.cosine(a, b)))
            assert np.array_equal(bits(P.normalize(a)), bits(O.normalize(a)))
    assert np.array_equal(P.normalize(np.zeros(5, np.float32)), np.zeros(5, np.float32))


def test_random_configurations_python_equals_cpp():
    """random sizes, dims, M, ef, metrics, both algorithms, removals interleaved with inserts: graph + answers + counters"""
    rng = np.random.default_rng(20260927)
    for trial in range(14):
        n = int(rng.integers(20, 110)); d = int(rng.integers(2, 20)); m = int(rng.choice([3, 4, 8])); metric = int(rng.integers(0, 2))
        efc = int(rng.integers(m, 40)); algo = int(rng.integers(0, 2)); ef = int(rng.integers(1, 40)); k = int(rng.integers(1, 12))
        X = rng.standard_normal((n, d)).astype(np.float32); ids = rng.permutation(n).astype(np.uint64) + np.uint64(100)
        lv = np.floor(-np.log(1.0 - rng.random(n)) / np.log(float(m))).astype(np.int32)
        hp = P.Hnsw(d, metric, m=m, ef=ef, ef_construction=efc, algo=algo)
        hc = O.Hnsw(d, metric, O.default_cfg(m=m, ef=ef, efConstruction=efc, algo=algo))
        live = []
        for i in range(n):
            assert hp.insert(int(ids[i]), X[i], int(lv[i])) is None and hc.insert(int(ids[i]), X[i], int(lv[i])) == 0
            live.append(int(ids[i]))
            if rng.random() < 0.15 and live:       # removals in between, the entrypoint among them now and then
                v = live.pop(int(rng.integers(0, len(live)))) if rng.random() < 0.7 or hp.entry is None else hp.entry.id
                if v in live: live.remove(v)
                assert hp.remove(v) is None and hc.remove(v) == 0
            if rng.random() < 0.03:
                assert hp.insert(int(ids[i]), X[i], 0) == "ItemAlreadyExistsError" and hc.insert(int(ids[i]), X[i], 0) == -2
        assert hp.remove(10**9) == "ItemNotFoundError" and hc.remove(10**9) == -3
        gp, gc = hp.export(), hc.export(with_vectors=False)
        for key in ("ids", "levels", "deleted", "row_offsets", "nbr"):
            assert np.array_equal(gp[key], gc[key]), (trial, key)
        assert np.array_equal(bits(gp["nbr_dist"]), bits(gc["nbr_dist"])) and gp["entry"] == gc["entry"], trial
        for q in rng.standard_normal((6, d)).astype(np.float32):
            hp.n_dist = 0
            rp = hp.search(q, k)
            wi, ws, st = hc.search(q, k, mode=0, with_stats=True)
            assert [i for i, _ in rp] == wi.tolist() and np.array_equal(bits(np.float32([s for _, s in rp])), bits(ws)), trial
            assert hp.n_dist == st["n_dist"], (trial, hp.n_dist, st)


def test_edge_queue_restatements_agree():
    """edge.PriorityQueue keeps the K LARGEST scores (min-heap + pop-min) and returns them ascending — both restatements"""
    rng = np.random.default_rng(9)
    n, d, k = 300, 12, 7
    X = rng.standard_normal((n, d)).astype(np.float32); ids = np.arange(n, dtype=np.uint64) * np.uint64(3)
    f = O.Flat(d, O.L2); f.upsert(ids, X)
    q = rng.standard_normal(d).astype(np.float32)
    wi, ws = f.search(q, k, nearest=False, mode=0)           # literal Go-heap queue, scan order = shard by shard, ascending id inside
    order = sorted(range(n), key=lambda i: (O.shard_vertex(int(ids[i]), 16), int(ids[i])))
    got = P.edge_queue([(P.euclidean(q, X[i]), int(ids[i])) for i in order], k)
    assert [i for _, i in got] == wi.tolist() and np.array_equal(bits(np.float32([s for s, _ in got])), bits(ws))
    far = sorted((float(P.euclidean(q, X[i])), int(ids[i])) for i in range(n))[-k:]
    assert [i for _, i in got] == [i for _, i in far]        # ... and they ARE the k farthest


def test_literal_restatements_agree_under_nan_distances():
    """Zero vectors under cosine give NaN distances; Go's heaps compare priorities with `<` (false both ways), so a NaN item never
    sifts and can sit at the root of the candidate heap — the reference's result is then an artefact of the heap array's layout.
    Both LITERAL restatements (C++ and Python, each with its own container/heap) follow that behaviour and must still agree bit for
    bit: graph (NaN edge distances included), answers, counters.  (The canonical closed form the GPU runs orders NaN after +Inf
    instead and is NOT equal to this — DESIGN.md §4; such collections are outside the parity contract.)"""
    rng = np.random.default_rng(77)
    saw_nan_edges = 0
    for trial in range(10):
        n = int(rng.integers(30, 90)); d = int(rng.integers(2, 10)); m = int(rng.choice([3, 4, 8]))
        efc = int(rng.integers(m, 30)); algo = int(rng.integers(0, 2)); ef = int(rng.integers(1, 30)); k = int(rng.integers(1, 10))
        X = rng.standard_normal((n, d)).astype(np.float32)
        for z in rng.choice(n, size=3, replace=False):
            X[z] = 0.0
        ids = np.arange(n, dtype=np.uint64) + np.uint64(5)
        lv = np.floor(-np.log(1.0 - rng.random(n)) / np.log(float(m))).astype(np.int32)
        hp = P.Hnsw(d, P.Hnsw.COSINE, m=m, ef=ef, ef_construction=efc, algo=algo)
        hc = O.Hnsw(d, O.COSINE, O.default_cfg(m=m, ef=ef, efConstruction=efc, algo=algo))
        for i in range(n):
            assert hp.insert(int(ids[i]), X[i], int(lv[i])) is None and hc.insert(int(ids[i]), X[i], int(lv[i])) == 0
        gp, gc = hp.export(), hc.export(with_vectors=False)
        for key in ("ids", "levels", "deleted", "row_offsets", "nbr"):
            assert np.array_equal(gp[key], gc[key]), (trial, key)
        assert np.array_equal(bits(gp["nbr_dist"]), bits(gc["nbr_dist"])) and gp["entry"] == gc["entry"], trial
        saw_nan_edges += int(np.isnan(gc["nbr_dist"]).sum())
        Q = rng.standard_normal((5, d)).astype(np.float32); Q[0] = 0.0
        for q in Q:
            rp = hp.search(q, k)
            wi, ws = hc.search(q, k, mode=0)
            assert [i for i, _ in rp] == wi.tolist() and np.array_equal(bits(np.float32([s for _, s in rp])), bits(ws)), trial
    assert saw_nan_edges > 0
