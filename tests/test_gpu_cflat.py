"""experimental CFLAT (multi-vector weighted FLAT scan) on the GPU vs the oracle restatement: bit-exact ids, ranks, scores."""
import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", [O.COSINE, O.L2])
def test_cflat_search_parity(gpu, metric):
    n, d, nf = 1500, 64, 3
    X = O.fill_normal(51, (n, nf, d)); ids = np.arange(n, dtype=np.uint64) * np.uint64(7) + np.uint64(2)
    oc = O.CFlat(d, nf, metric); oc.upsert(ids, X)
    gc = gpu.MultiVectorSpace(d, nf, metric); gc.ChangedVertex(ids, X)
    assert gc.Len() == n
    Q = O.fill_normal(52, (6, nf, d))
    for ratios, inc in (([50, 30, 20], [1, 1, 1]), ([100, 0, 40], [1, 0, 1]), ([33, 33, 34], [0, 1, 0])):
        for k in (1, 10, 50):
            gi, gs, gcnt = gc.MultiVertexSearch(k, Q, ratios, inc)
            for qi in range(len(Q)):
                wi, ws = oc.search(Q[qi], ratios, inc, k)
                assert_same_results(gi[qi, :gcnt[qi]], gs[qi, :gcnt[qi]], wi, ws, f"q{qi} r{ratios} k{k}")
    # overwrite + remove keep matching
    up = O.fill_normal(53, (40, nf, d)); oc.upsert(ids[100:140], up); gc.ChangedVertex(ids[100:140], up)
    rm = np.concatenate([ids[300:360], np.array([10**9], np.uint64)]); oc.remove(rm); gc.RemoveVertex(rm)
    assert gc.Len() == n - 60
    gi, gs, gcnt = gc.MultiVertexSearch(20, Q, [60, 25, 15])
    for qi in range(len(Q)):
        wi, ws = oc.search(Q[qi], [60, 25, 15], [1, 1, 1], 20)
        assert_same_results(gi[qi, :gcnt[qi]], gs[qi, :gcnt[qi]], wi, ws)
    e = gpu.MultiVectorSpace(d, nf, metric)
    assert e.MultiVertexSearch(5, Q[:1], [1, 1, 1])[2][0] == 0
