"""Product-quantised HNSW, the CPU side: the oracle's definition (coltt_oracle.cpp "Product-quantised HNSW": the canonical Hnsw.Search walk of
core/vectorindex/hnsw.go:243-278, 320-389 with the product quantiser's table distance, then an exact re-rank) against an independent
plain-Python restatement of the same definition (oracle/pyref.py: csr_search_pq).  The package the reference drives for this
(pkg/hnswpq, playground/hnswpq_verification.go:69-105) is absent from its tree: there is nothing of the reference's to pin this against."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import pyref as P


def _padded(g, w0, wu):
    """export() CSR -> the padded HBM-layout arrays (adj0 [n][w0], upper_off [n], adjU [rows][wu]), rows ascending by slot"""
    lv = g["levels"]; n = len(lv)
    adj0 = np.full((n, w0), 0xFFFFFFFF, np.uint32); upper_off = np.full(n, 0xFFFFFFFF, np.uint32)
    nu = int(lv.sum()); adjU = np.full((max(nu, 1), wu), 0xFFFFFFFF, np.uint32)
    row = 0; up = 0
    for i in range(n):
        if lv[i] > 0:
            upper_off[i] = up
        for l in range(lv[i] + 1):
            b, e = g["row_offsets"][row], g["row_offsets"][row + 1]
            nb = np.sort(g["nbr"][b:e].astype(np.uint32))
            if l == 0:
                adj0[i, :len(nb)] = nb
            else:
                adjU[up + l - 1, :len(nb)] = nb
            row += 1
        up += int(lv[i])
    return adj0, upper_off, adjU


@pytest.mark.parametrize("metric,quant,pqm,scale", [(O.COSINE, O.Q_NONE, O.PQ_COSINE, 1.0), (O.L2, O.Q_F16, O.PQ_EUCLIDEAN, 1.0), (O.COSINE, O.Q_F16, O.PQ_EUCLIDEAN, 1.0),
                                                         (O.L2, O.Q_NONE, O.PQ_EUCLIDEAN, 2.0 ** -9),    # table entries that are binary16 denormals
                                                         (O.L2, O.Q_NONE, O.PQ_EUCLIDEAN, 300.0)])       # entries beyond binary16's 65504: the table scale (round 6)
def test_oracle_pq_walk_equals_the_independent_python_restatement(metric, quant, pqm, scale):
    n, d, m, c, k = 260, 32, 8, 16, 5
    X = (O.fill_normal(4100 + metric + 3 * quant, (n, d)) * np.float32(scale)).astype(np.float32); lv = O.levels(4200, n)
    stored_f32 = np.array([O.normalize(x) for x in X]) if metric == O.COSINE else X
    rows = O.lower(quant, stored_f32) if quant != O.Q_NONE else stored_f32       # what the index stores
    seen = O.f16_decode(rows) if quant != O.Q_NONE else rows                      # ... as its distance sees it
    h = O.Hnsw(d, metric, cfg=O.default_cfg(m=6, ef=16, efConstruction=30))     # any valid graph will do: the topology is an input of the walk
    h.insert_many(np.arange(n, dtype=np.uint64), X, lv)
    g = h.export(with_vectors=False)
    adj0, upper_off, adjU = _padded(g, h.cfg.mMax0, h.cfg.mMax)
    entry = int(g["entry"]); entry_level = int(g["levels"][entry])
    cb = O.pq_train(seen[:120], m, c, iters=3)
    codes = O.pq_encode(cb, seen)
    Q = (O.fill_normal(4300, (5, d)) * np.float32(scale)).astype(np.float32)
    if scale < 1.0:
        h16 = np.concatenate([O.pq_lut(pqm, cb, Q[i]).ravel() for i in range(len(Q))]).astype(np.float16)
        assert (np.abs(h16[h16 != 0]) < 6.2e-5).mean() > 0.3
    if scale > 1.0:
        assert max(float(O.pq_lut(pqm, cb, Q[i]).max()) for i in range(len(Q))) > 65504.0, "the case no longer overflows binary16 without the scale"
    for ef, rr in ((12, 0), (40, 7), (5, 2)):
        sl, sc, cn, st, _ = O.csr_search_pq(rows, quant, adj0, upper_off, adjU, d, metric, entry, entry_level, codes, cb, pqm, Q, k, ef, rerank=rr)
        tot = {"n_dist": 0, "n_exp": 0, "n_hops": 0, "n_exact": 0}
        for qi in range(len(Q)):
            q = O.normalize(Q[qi]) if metric == O.COSINE else Q[qi]
            if quant != O.Q_NONE:
                q = O.f16_decode(O.lower(quant, q))
            ws, wsc, cnt = P.csr_search_pq(seen, adj0, upper_off, adjU, 0 if metric == O.COSINE else 1, entry, entry_level, codes, cb, pqm, q, k, ef, rr)
            assert list(sl[qi, :cn[qi]]) == ws, (ef, rr, qi)
            assert np.array_equal(sc[qi, :cn[qi]].view(np.uint32), np.array(wsc, np.float32).view(np.uint32)), (ef, rr, qi)
            for kk in tot: tot[kk] += cnt[kk]
        assert st == tot, (ef, rr, st, tot)


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_oracle_pq_walk_equals_the_python_restatement_on_random_shapes(seed):
    """the same double entry over random shapes: sub-vector counts that leave padding rows in the table (m not a multiple of 16), centroid counts that are
    not powers of two (the table's row is as long as the next power of two), tombstone-free graphs of random degree, ef below / at / above k"""
    rng = np.random.default_rng(seed)
    m = int(rng.choice([2, 4, 6, 8, 12])); dsub = int(rng.choice([2, 3, 4])); d = m * dsub
    c = int(rng.choice([5, 16, 17, 33, 64])); n = int(rng.integers(150, 320)); k = int(rng.choice([1, 3, 7]))
    metric = int(rng.choice([O.COSINE, O.L2])); quant = int(rng.choice([O.Q_NONE, O.Q_F16]))
    pqm = O.PQ_COSINE if (metric == O.COSINE and rng.random() < 0.5) else O.PQ_EUCLIDEAN
    X = O.fill_normal(7000 + seed, (n, d)); lv = O.levels(7100 + seed, n)
    stored_f32 = np.array([O.normalize(x) for x in X]) if metric == O.COSINE else X
    rows = O.lower(quant, stored_f32) if quant != O.Q_NONE else stored_f32
    seen = O.f16_decode(rows) if quant != O.Q_NONE else rows
    h = O.Hnsw(d, metric, cfg=O.default_cfg(m=int(rng.choice([4, 6, 8])), ef=16, efConstruction=24))
    h.insert_many(np.arange(n, dtype=np.uint64), X, lv)
    g = h.export(with_vectors=False)
    adj0, upper_off, adjU = _padded(g, h.cfg.mMax0, h.cfg.mMax)
    entry = int(g["entry"]); entry_level = int(g["levels"][entry])
    cb = O.pq_train(seen[: max(c, 100)], m, c, iters=2)
    codes = O.pq_encode(cb, seen)
    Q = O.fill_normal(7200 + seed, (4, d))
    for ef, rr in ((max(1, k - 1), 0), (k, k), (4 * k + 9, 0), (30, 5)):
        sl, sc, cn, st, _ = O.csr_search_pq(rows, quant, adj0, upper_off, adjU, d, metric, entry, entry_level, codes, cb, pqm, Q, k, ef, rerank=rr)
        tot = {"n_dist": 0, "n_exp": 0, "n_hops": 0, "n_exact": 0}
        for qi in range(len(Q)):
            q = O.normalize(Q[qi]) if metric == O.COSINE else Q[qi]
            if quant != O.Q_NONE:
                q = O.f16_decode(O.lower(quant, q))
            ws, wsc, cnt = P.csr_search_pq(seen, adj0, upper_off, adjU, 0 if metric == O.COSINE else 1, entry, entry_level, codes, cb, pqm, q, k, ef, rr)
            assert list(sl[qi, :cn[qi]]) == ws, (seed, ef, rr, qi)
            assert np.array_equal(sc[qi, :cn[qi]].view(np.uint32), np.array(wsc, np.float32).view(np.uint32)), (seed, ef, rr, qi)
            for kk in tot: tot[kk] += cnt[kk]
        assert st == tot, (seed, ef, rr, st, tot)
