"""tests/golden/round6_definitions.npz — vectors written by oracle/pyref.py (pure Python) for the two DEFINITIONS of round 6 (neither is reference
behaviour: committed vectors are what pins them): COLTT_HNSW_DIVERSE graphs, and the walk over product-quantiser codes (two half-row table sums, bounded
visiting).  The C++ oracle must reproduce them bit for bit here; the HIP path does in tests/test_gpu_diverse.py / test_gpu_round5.py."""
import os

import numpy as np
import pytest

from oracle import oracle as O

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "round6_definitions.npz"))


@pytest.mark.parametrize("ci", range(int(Z["n_diverse"])))
def test_cpp_oracle_builds_the_golden_diverse_graph(ci):
    g = lambda k: Z[f"d{ci}_{k}"]
    d, metric, m, mmax0, efc, keep = (int(v) for v in g("cfg"))
    h = O.Hnsw(d, O.COSINE if metric == 0 else O.L2, O.default_cfg(m=m, mMax0=mmax0, efConstruction=efc, algo=2, keepPruned=keep))
    X, ids, lv = g("X"), g("ids"), g("levels")
    rem = {int(a): int(b) for a, b in g("removes")}
    for i in range(len(X)):
        assert h.insert(ids[i], X[i], lv[i]) == 0
        if i in rem:
            assert h.remove(ids[rem[i]]) == 0
    e = h.export(with_vectors=False)
    for k in ("levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(e[k], g("g_" + k)), k
    assert np.array_equal(e["nbr_dist"].view(np.uint32), g("g_nbr_dist").view(np.uint32)) and e["entry"] == int(g("g_entry"))


@pytest.mark.parametrize("ci", range(int(Z["n_pq"])))
def test_cpp_oracle_walks_the_golden_pq_case(ci):
    g = lambda k: Z[f"p{ci}_{k}"]
    d, metric, m, c, ef, k, rr = (int(v) for v in g("cfg"))
    om = O.COSINE if metric == 0 else O.L2
    sl, sc, cn, st, _ = O.csr_search_pq(g("seen"), O.Q_NONE, g("adj0"), g("upper_off"), g("adjU"), d, om, int(g("entry")), int(g("entry_level")), g("codes"), g("cb"),
                                        O.PQ_EUCLIDEAN, g("Q"), k, ef, rerank=rr)
    assert np.array_equal(cn.astype(np.int64), g("counts").astype(np.int64))
    for qi in range(len(cn)):
        n_ = int(cn[qi])
        assert np.array_equal(sl[qi, :n_].astype(np.int64), g("slots")[qi, :n_]), qi
        assert np.array_equal(sc[qi, :n_].view(np.uint32), g("scores")[qi, :n_].view(np.uint32)), qi
    assert [st["n_dist"], st["n_exp"], st["n_hops"], st["n_exact"]] == [int(v) for v in g("counters")]
    assert np.array_equal(O.pq_encode(g("cb"), g("seen")), g("codes"))
