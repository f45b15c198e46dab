"""Round-2 GPU parity cases: BASELINE.json configs[4] shape ("bf16" HNSW at efSearch 256 on a GPU-built graph), the
Heuristic search algorithm, Insert with a nil entrypoint after Remove, transactional loads, odd FLAT dims, concurrent
searches beside inserts / removes (the reference's RWMutex discipline, hnsw.go:51, hnsw_vertex.go:39)."""
import threading
import time

import numpy as np
import pytest

from oracle import oracle as O
from util import assert_same_results, bits

pytestmark = pytest.mark.gpu


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


@pytest.mark.parametrize("mwq", ["0", "128"])
@pytest.mark.parametrize("n,d", [(5000, 128), (1800, 768)])
def test_c5_shape_bf16_hnsw_ef256_on_gpu_built_graph(gpu, monkeypatch, mwq, n, d):
    """configs[4]: cosine HNSW over the reference's "BF16" codes (= binary16, bf16.go:233-317), efSearch 256 (the HBM-visited
    kernel), graph built by the GPU's batched Insert.  Checker: the oracle's canonical Hnsw.Search over the very arrays
    copied out of HBM (stored codes, adjacency), query lowered as bf16_vectorstore.go:136 does: ids, score bits, counters."""
    monkeypatch.setenv("COLTT_MW_MAX_NQ", mwq)   # one wave per query / the opt-in 256-thread staged kernel (HBM-visited variants of both)
    X = O.fill_normal(500 + d, (n, d)); lv = O.levels(501 + d, n)
    gh = _gpu_build(gpu, X, lv, O.COSINE, O.Q_BF16, gpu.HnswCfg.default(ef_construction=100))
    g = gh.ExportRaw(); rows = gh.FetchRows()
    assert rows.dtype == np.uint16 and rows.shape == (n, d)
    # the stored bits are Normalize + Lower of the input (hnsw.go:105-107 + bf16_quantization.go Lower)
    want = np.stack([O.lower(O.Q_BF16, O.normalize(X[i])) for i in range(0, n, 97)])
    assert np.array_equal(rows[::97], want)
    Q = O.fill_normal(502 + d, (48, d))
    for ef in (256, 128):
        gi, gs, gc, st = gh.Search(Q, 10, ef=ef, with_stats=True)
        sl, sc, cn, ost, _ = O.csr_search(rows, O.Q_BF16, g["adj0"], g["upper_off"], g["adjU"], d, O.COSINE, g["entry"], g["entry_level"],
                                          Q, 10, ef, threads=4)
        assert st["n_visit_resets"] == 0
        for qi in range(len(Q)):
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"q{qi} ef{ef}")
        assert {k: st[k] for k in ost} == ost, (st, ost)


def test_unknown_quantization_is_rejected_not_mapped(gpu):
    for q in (4, 7, -1):
        with pytest.raises(gpu.ColttError) as e:
            gpu.Hnsw(16, O.L2, quantization=q)
        assert e.value.code == -4  # "not support quantization type" (edge/vectorstore.go:79)
        with pytest.raises(gpu.ColttError) as e:
            gpu.FlatSpace(16, O.L2, q)
        assert e.value.code == -4


def test_hnsw_heuristic_algorithm(gpu):
    """HnswSearchHeuristic with extendCandidates=false (selectNeighborsHeuristic, hnsw.go:399-447: the k nearest; the
    keepPruned loop is dead code): build (batch 1 == the reference's sequential Insert) and search equal the oracle's LITERAL
    restatement run with algo=1; extendCandidates=true is rejected."""
    import torch
    n, d = 600, 32
    X = O.fill_normal(601, (n, d)); lv = O.levels(602, n); ids = np.arange(n, dtype=np.uint64) + np.uint64(9)
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(algo=1)); oh.insert_many(ids, X, lv)
    gh = gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(algo=1))
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    gh.InsertBatchDevice(xd.data_ptr(), n, lv, batch=1, ids=ids)
    go, oo = gh.Export(), oh.export(with_vectors=False)
    for k in ("levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(go[k], oo[k]), k
    assert np.array_equal(bits(go["nbr_dist"]), bits(oo["nbr_dist"]))
    Q = O.fill_normal(603, (20, d))
    gi, gs, gc = gh.Search(Q, 10, ef=50)
    for qi in range(len(Q)):
        wi, ws = oh.search(Q[qi], 10, mode=0, ef=50)   # literal Go heaps + selectNeighborsHeuristic
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi}")
    with pytest.raises(gpu.ColttError) as e:
        gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(algo=1, extend_candidates=1))
    assert e.value.code == -4


def test_hnsw_insert_with_nil_entrypoint_after_remove(gpu):
    """Remove can leave the index without an entrypoint (hnsw.go:197-217 CASes it to the removed vertex's closest neighbour,
    nil when it has none); the next Insert stores its vertex at level 0 and makes it the entrypoint (hnsw.go:108-116).
    insert(1); remove(1); insert(2) must work, and so must remove-everything-then-rebuild."""
    d = 12
    X = O.fill_normal(701, (40, d)); lv = O.levels(702, 40); lv[:3] = [2, 1, 3]
    oh = O.Hnsw(d, O.L2); gh = gpu.Hnsw(d, O.L2)

    def both(op, *a):
        if op == "ins":
            assert oh.insert(a[0], X[a[1]], int(lv[a[1]])) == 0; gh.Insert(a[0], X[a[1]], int(lv[a[1]]))
        else:
            assert oh.remove(a[0]) == 0; gh.Remove(a[0])

    def same():
        go, oo = gh.Export(), oh.export(with_vectors=False)
        for k in ("ids", "levels", "deleted", "row_offsets", "nbr"):
            assert np.array_equal(go[k], oo[k]), k
        assert go["entry"] == oo["entry"]
        gi, gs, gc = gh.Search(X[:6], 5, ef=16)
        for qi in range(6):
            wi, ws = oh.search(X[qi], 5, mode=1, ef=16)
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi}")

    both("ins", 1, 0); both("rem", 1); same()
    assert gh.Len() == 0 and gh.Export()["entry"] == -1
    both("ins", 2, 1); same()                      # level forced to 0 although lv[1] = 1
    assert gh.Export()["levels"][1] == 0
    for i in range(2, 20): both("ins", 100 + i, i)
    same()
    for i in range(2, 20): both("rem", 100 + i)     # removing everything (the entrypoint among them, repeatedly)
    both("rem", 2); same()
    assert gh.Len() == 0
    for i in range(20, 40): both("ins", 200 + i, i)
    same()


def test_failed_loads_leave_the_index_untouched(gpu):
    """A truncated / corrupt stream is rejected BEFORE the object changes: config, graph and answers stay what they were
    (Hnsw.Load hnsw_commit.go:164-278; LoadVertex none_vectorstore.go:425-516)."""
    n, d = 400, 24
    X = O.fill_normal(801, (n, d)); lv = O.levels(802, n); ids = np.arange(n, dtype=np.uint64) * np.uint64(5)
    oh = O.Hnsw(d, O.COSINE, O.default_cfg(ef=31)); oh.insert_many(ids, X, lv)
    gh = gpu.Hnsw(d, O.COSINE, gpu.HnswCfg.default(ef=31)); gh.BulkLoad(oh.export(with_vectors=False), X)
    Q = O.fill_normal(803, (10, d))
    before = gh.Search(Q, 10, ef=40)
    other = O.Hnsw(d, O.COSINE, O.default_cfg(m=8, ef=77)); other.insert_many(ids[:200], X[:200], lv[:200])
    stream = other.commit(header=True)
    for cut in (len(stream) // 3, len(stream) - 7, 30):
        with pytest.raises(gpu.ColttError):
            gh.Load(stream[:cut])
        c = gh.Config()
        assert c.ef == 31 and c.m == 16 and gh.Len() == n                  # cfg was NOT taken from the bad stream
        after = gh.Search(Q, 10, ef=40)
        assert np.array_equal(before[0], after[0]) and np.array_equal(bits(before[1]), bits(after[1]))
    assert gh.Load(stream) == 200 and gh.Config().m == 8 and gh.Config().ef == 77  # and a good stream still loads
    # FLAT
    of = O.Flat(d, O.L2, O.Q_F16); of.upsert(ids, X)
    gf = gpu.FlatSpace(d, O.L2, O.Q_F16); gf.ChangedVertex(ids, X)
    fb = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST)
    blob = of.save_vertex()
    for bad in (blob[:len(blob) // 2], blob[:11]):
        with pytest.raises(gpu.ColttError):
            gf.LoadVertex(bad)
        fa = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST)
        assert gf.LoadSize() == n and np.array_equal(fb[0], fa[0]) and np.array_equal(bits(fb[1]), bits(fa[1]))


@pytest.mark.parametrize("d", [7, 9, 10, 50, 1])
@pytest.mark.parametrize("quant", [O.Q_NONE, O.Q_F16])
def test_flat_dims_not_multiple_of_four(gpu, d, quant):
    """dim % 4 != 0: the query tile in LDS is padded to 16-byte rows; the scalar tail follows the AVX kernel's (avx.cpp:27-31)."""
    n = 700
    X = O.fill_normal(900 + d, (n, d)); ids = np.arange(n, dtype=np.uint64) + np.uint64(3)
    Q = O.fill_normal(901 + d, (19, d))        # more than one 16-query tile, last one ragged
    for metric in (O.COSINE, O.L2):
        of = O.Flat(d, metric, quant); of.upsert(ids, X)
        gf = gpu.FlatSpace(d, metric, quant); gf.ChangedVertex(ids, X)
        for nearest in (True, False):
            gi, gs, gc = gf.VertexSearch(Q, 7, gpu.SELECT_NEAREST if nearest else gpu.SELECT_REFERENCE)
            for qi in range(len(Q)):
                wi, ws = of.search(Q[qi], 7, nearest=nearest, mode=2)
                assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"d{d} q{qi} near{nearest}")


def test_flat_upsert_overwrite_remove_rounds(gpu):
    """several rounds of upsert (new + overwritten + repeated ids in one batch), removal and search against the oracle: the id
    table on the device is patched incrementally (only the appended range / the moved slot is uploaded)."""
    d = 16
    rng = np.random.default_rng(11)
    of = O.Flat(d, O.COSINE); gf = gpu.FlatSpace(d, O.COSINE)
    Q = O.fill_normal(1000, (8, d))
    live = set()
    for rnd in range(6):
        ids = rng.integers(0, 300, 120).astype(np.uint64)          # collisions with earlier rounds and inside the batch
        V = O.fill_normal(1001 + rnd, (len(ids), d))
        gf.ChangedVertex(ids, V)
        for i in range(len(ids)): of.upsert(ids[i:i + 1], V[i:i + 1])  # sequential semantics: last one wins
        live |= set(ids.tolist())
        rem = rng.choice(sorted(live), 25, replace=False).astype(np.uint64)
        gf.RemoveVertex(np.concatenate([rem, np.uint64([999999])])); of.remove(rem)
        live -= set(rem.tolist())
        assert gf.LoadSize() == len(of) == len(live)
        gi, gs, gc = gf.VertexSearch(Q, 12, gpu.SELECT_NEAREST)
        for qi in range(len(Q)):
            wi, ws = of.search(Q[qi], 12, nearest=True, mode=2)
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"round{rnd} q{qi}")


def test_concurrent_searches_beside_inserts_and_removes(gpu, capsys):
    """Searches hold the index lock shared (each on its own stream / workspaces), Insert and Remove hold it exclusive.
    (1) 64 threads issuing single-query calls concurrently get exactly the batch answers (LDS- and HBM-visited kernels);
    (2) while one thread applies a fixed sequence of inserts / removes, searcher threads keep running and only ever see
        well-formed answers; once quiesced, the graph and the answers equal the oracle's serial application of the sequence."""
    from concurrent.futures import ThreadPoolExecutor
    n, d = 4000, 48
    X = O.fill_normal(1101, (n + 300, d)); lv = O.levels(1102, n + 300); ids = np.arange(n + 300, dtype=np.uint64)
    oh = O.Hnsw(d, O.L2, O.default_cfg(efConstruction=60)); oh.insert_many(ids[:n], X[:n], lv[:n])
    gh = gpu.Hnsw(d, O.L2, gpu.HnswCfg.default(ef_construction=60)); gh.BulkLoad(oh.export(with_vectors=False), X[:n])
    Q = O.fill_normal(1103, (256, d))
    for ef in (64, 200):
        want = gh.Search(Q, 10, ef=ef)

        def one(i):
            a = gh.Search(Q[i:i + 1], 10, ef=ef)
            return np.array_equal(a[0][0], want[0][i]) and np.array_equal(bits(a[1][0]), bits(want[1][i]))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(64) as ex:
            ok = list(ex.map(one, list(range(256)) * 4))
        dt = time.perf_counter() - t0
        assert all(ok)
        with capsys.disabled():
            print(f"\n[concurrency] 64 callers x 1-query HNSW calls, ef={ef}, {n}x{d}: {len(ok) / dt:.0f} queries/s "
                  f"({dt / len(ok) * 64 * 1e3:.2f} ms per call at 64 in flight)")
    # (2)
    stop = threading.Event(); errors = []
    valid_ids = set(range(n + 300))

    def searcher(t):
        rng = np.random.default_rng(t)
        while not stop.is_set():
            i = int(rng.integers(0, 250))
            gi, gs, gc = gh.Search(Q[i:i + 4], 10, ef=int(rng.choice([48, 160])))
            for r in range(4):
                c = int(gc[r])
                if c > 10 or not set(gi[r, :c].tolist()) <= valid_ids or np.any(np.diff(gs[r, :c]) < 0) or len(set(gi[r, :c].tolist())) != c:
                    errors.append((t, i, gi[r], gs[r], c))
    th = [threading.Thread(target=searcher, args=(t,)) for t in range(8)]
    for t in th: t.start()
    rng = np.random.default_rng(99)
    victims = rng.choice(n, 150, replace=False)
    try:
        for j in range(300):
            gh.Insert(int(ids[n + j]), X[n + j], int(lv[n + j])); assert oh.insert(ids[n + j], X[n + j], int(lv[n + j])) == 0
            if j % 2 == 0:
                v = int(victims[j // 2]); gh.Remove(v); assert oh.remove(v) == 0
    finally:
        stop.set()
        for t in th: t.join()
    assert not errors, errors[:2]
    go, oo = gh.Export(), oh.export(with_vectors=False)
    for k in ("ids", "levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(go[k], oo[k]), k
    assert np.array_equal(bits(go["nbr_dist"]), bits(oo["nbr_dist"])) and go["entry"] == oo["entry"]
    gi, gs, gc = gh.Search(Q[:40], 10, ef=80)
    for qi in range(40):
        wi, ws = oh.search(Q[qi], 10, mode=1, ef=80)
        assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"q{qi}")


def test_concurrent_flat_searches_and_upserts(gpu):
    """FLAT: concurrent callers (exact and MFMA modes) get the single-threaded answers; upserts interleave safely."""
    from concurrent.futures import ThreadPoolExecutor
    n, d = 6000, 64
    X = O.fill_normal(1201, (n, d)); ids = np.arange(n, dtype=np.uint64)
    gf = gpu.FlatSpace(d, O.COSINE, O.Q_F16); gf.ChangedVertex(ids, X)
    Q = O.fill_normal(1202, (64, d))
    want = gf.VertexSearch(Q, 10, gpu.SELECT_NEAREST)

    def one(i):
        mode = gpu.MODE_MFMA if i % 2 else gpu.MODE_EXACT
        a = gf.VertexSearch(Q[(i * 4) % 64:(i * 4) % 64 + 4], 10, gpu.SELECT_NEAREST, mode=mode)
        lo = (i * 4) % 64
        return np.array_equal(a[0], want[0][lo:lo + 4]) and np.array_equal(bits(a[1]), bits(want[1][lo:lo + 4]))
    with ThreadPoolExecutor(16) as ex:
        assert all(ex.map(one, range(96)))
    # writers beside readers: re-upserting the SAME vectors never changes an answer
    def writer(_):
        for r in range(10):
            gf.ChangedVertex(ids[r * 100:(r + 1) * 100], X[r * 100:(r + 1) * 100])
        return True
    with ThreadPoolExecutor(10) as ex:
        res = list(ex.map(lambda i: writer(i) if i == 0 else one(i), range(40)))
    assert all(res)


def test_gpu_reproduces_the_independent_python_golden(gpu):
    """tests/golden/hnsw_pyref.npz comes from oracle/pyref.py — the second restatement, written from the Go text.  The HIP path
    (sequential Insert, Remove, Insert after removals, Search; Simple and Heuristic) must reproduce graph, edge distances, ids and
    score bits."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hnsw_pyref.npz"))
    for ci in range(int(z["n_cases"])):
        c = {k[len(f"c{ci}_"):]: z[k] for k in z.files if k.startswith(f"c{ci}_")}
        d, metric, m, efc, algo, k, ef = (int(v) for v in c["cfg"])
        gh = gpu.Hnsw(d, metric, gpu.HnswCfg.default(m=m, ef_construction=efc, algo=algo))
        for i in range(len(c["ids"])):
            gh.Insert(int(c["ids"][i]), c["X"][i], int(c["levels"][i]))
        for i in c["removed"]:
            gh.Remove(int(i))
        for i in range(len(c["y_ids"])):
            gh.Insert(int(c["y_ids"][i]), c["Y"][i], int(c["y_levels"][i]))
        g = gh.Export()
        assert np.array_equal(g["levels"], c["g_levels"]) and np.array_equal(g["deleted"], c["g_deleted"]), ci
        assert np.array_equal(g["row_offsets"], c["g_row_offsets"]) and np.array_equal(g["nbr"], c["g_nbr"]), ci
        assert np.array_equal(bits(g["nbr_dist"]), bits(c["g_nbr_dist"])) and g["entry"] == int(c["g_entry"]), ci
        gi, gs, gc = gh.Search(c["Q"], k, ef=ef)
        for qi in range(len(c["Q"])):
            n = int(c["res_n"][qi])
            assert gc[qi] == n and np.array_equal(gi[qi, :n], c["res_ids"][qi, :n]) and np.array_equal(bits(gs[qi, :n]), bits(c["res_scores"][qi, :n])), (ci, qi)


@pytest.mark.parametrize("quant", [O.Q_NONE, O.Q_F16, O.Q_BF16])
@pytest.mark.parametrize("shape", ["plain", "norms", "dups"])
def test_flat_mfma_euclidean_equals_exact_mode(gpu, quant, shape):
    """Euclidean collections on the matrix cores (pkg/distance/space.go:61-63 is a selectable edge metric): candidates from
    s~^2 = ||q||^2 + ||r||^2 - 2 dot with a norm-scaled margin, exact re-score — ids, ranks and score bits must equal the exact
    mode's in both select directions; rows of very different norms, and stored duplicates of the query (distance 0, where
    s~^2 can come out negative), included."""
    n, d, k = 20000, 128, 10
    X = O.fill_normal(1600, (n, d))
    if shape == "norms":
        X = X * np.exp(O.fill_normal(1601, (n, 1)) * 1.5).astype(np.float32)       # norms spread over ~3 orders of magnitude
    Q = O.fill_normal(1602, (70, d))                                                # 70: one 64-query tile + a ragged one
    if shape == "dups":
        X[100:170] = Q; X[300:370] = Q * np.float32(1.0 + 2 ** -12)                # exact and near duplicates
    ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(1)
    gf = gpu.FlatSpace(d, O.L2, quant); gf.ChangedVertex(ids, X)
    for sel in (gpu.SELECT_NEAREST, gpu.SELECT_REFERENCE):
        e = gf.VertexSearch(Q, k, sel, mode=gpu.MODE_EXACT)
        m = gf.VertexSearch(Q, k, sel, mode=gpu.MODE_MFMA)
        assert np.array_equal(e[0], m[0]) and np.array_equal(bits(e[1]), bits(m[1])) and np.array_equal(e[2], m[2]), (quant, shape, sel)
    st = gf.Stats()
    assert st["mfma_groups"] == 2 and st["mfma_fallbacks"] == 0, st     # the matrix-core path really served both searches
    of = O.Flat(d, O.L2, quant); of.upsert(ids, X)
    m = gf.VertexSearch(Q[:6], k, gpu.SELECT_NEAREST, mode=gpu.MODE_MFMA)
    for qi in range(6):
        wi, ws = of.search(Q[qi], k, nearest=True, mode=2)
        assert_same_results(m[0][qi, :m[2][qi]], m[1][qi, :m[2][qi]], wi, ws, f"q{qi}")


def test_flat_mfma_euclidean_768_and_nonfinite_rows(gpu):
    n, d, k = 30000, 768, 10
    X = O.fill_normal(1700, (n, d)); ids = np.arange(n, dtype=np.uint64)
    Q = O.fill_normal(1701, (33, d))
    gf = gpu.FlatSpace(d, O.L2, O.Q_F16); gf.ChangedVertex(ids, X)
    e = gf.VertexSearch(Q, k, gpu.SELECT_NEAREST, mode=gpu.MODE_EXACT)
    m = gf.VertexSearch(Q, k, gpu.SELECT_NEAREST, mode=gpu.MODE_MFMA)
    assert np.array_equal(e[0], m[0]) and np.array_equal(bits(e[1]), bits(m[1])) and gf.Stats() == {"mfma_groups": 1, "mfma_fallbacks": 0}
    # a row with an infinite norm switches the Euclidean matrix-core path off for the store (the margin is scaled by the largest
    # norm): answers still equal the exact mode's
    Y = X[:5].copy(); Y[2, 7] = np.float32(1e30)
    g2 = gpu.FlatSpace(d, O.L2); g2.ChangedVertex(ids, X); g2.ChangedVertex(ids[:5] + np.uint64(10**6), Y)
    e = g2.VertexSearch(Q, k, gpu.SELECT_NEAREST, mode=gpu.MODE_EXACT)
    m = g2.VertexSearch(Q, k, gpu.SELECT_NEAREST, mode=gpu.MODE_MFMA)
    assert np.array_equal(e[0], m[0]) and np.array_equal(bits(e[1]), bits(m[1])) and g2.Stats()["mfma_groups"] == 0


def test_multi_wave_latency_path_falls_back_when_its_visited_table_is_too_small(gpu, monkeypatch):
    """ef far beyond what the multi-wave kernel's LDS visited table can hold without its reset path: the call is served by the
    single-wave kernel instead (host-side guard, or err 8 from the kernel) and the answers stay exact."""
    monkeypatch.setenv("COLTT_MW_MAX_NQ", "128"); monkeypatch.setenv("COLTT_VISG", "0")
    n, d = 20000, 8
    X = O.fill_normal(1800, (n, d)); lv = O.levels(1801, n)
    gh = _gpu_build(gpu, X, lv, O.L2, O.Q_NONE, gpu.HnswCfg.default(ef_construction=40), batch=1024)
    g = gh.ExportRaw(); rows = gh.FetchRows()
    Q = O.fill_normal(1802, (5, d))
    for ef in (600, 2500):
        gi, gs, gc, st = gh.Search(Q, 10, ef=ef, with_stats=True)
        sl, sc, cn, ost, _ = O.csr_search(rows, O.Q_NONE, g["adj0"], g["upper_off"], g["adjU"], d, O.L2, g["entry"], g["entry_level"], Q, 10, ef)
        for qi in range(len(Q)):
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"ef{ef} q{qi}")
        assert st["n_dist"] == ost["n_dist"] and st["n_exp"] == ost["n_exp"]


@pytest.mark.parametrize("budget_mb,expect", [("4", "capped at"), ("0", "using the LDS hash")])
def test_hbm_visited_workspace_cap_transitions(gpu, monkeypatch, capfd, budget_mb, expect):
    """The HBM visited set takes cap bytes x <= 2048 regions, bounded by a quarter of device memory (20 GB at 10 M; at 40 M on
    one GPU the bound bites).  COLTT_VISG_BUDGET_MB shrinks the bound so both transitions run here: fewer regions than resident
    workgroups (the launch shrinks to the lease), and fewer than 256 (the LDS hash serves ef > 128 too).  Either way a line
    on stderr says so and the answers and counters stay the oracle's."""
    monkeypatch.setenv("COLTT_VISG_BUDGET_MB", budget_mb); monkeypatch.setenv("COLTT_MW_MAX_NQ", "0")
    n, d = 4000, 32
    X = O.fill_normal(1900, (n, d)); lv = O.levels(1901, n)
    gh = _gpu_build(gpu, X, lv, O.L2, O.Q_NONE, gpu.HnswCfg.default(ef_construction=40), batch=512)   # efConstruction <= 128: no byte map yet
    g = gh.ExportRaw(); rows = gh.FetchRows()
    Q = O.fill_normal(1902, (700, d))    # more queries than the capped launch has workgroups
    capfd.readouterr()
    for ef in (300, 200):
        gi, gs, gc, st = gh.Search(Q, 10, ef=ef, with_stats=True)
        sl, sc, cn, ost, _ = O.csr_search(rows, O.Q_NONE, g["adj0"], g["upper_off"], g["adjU"], d, O.L2, g["entry"], g["entry_level"], Q, 10, ef, threads=4)
        for qi in range(len(Q)):
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"ef{ef} q{qi}")
        assert st["n_dist"] == ost["n_dist"] and st["n_exp"] == ost["n_exp"]
    err = capfd.readouterr().err
    assert expect in err, err


def test_nan_distances_have_one_total_order_on_both_sides(gpu):
    """Found by tools/fuzz_parity.py: stored vectors whose norm underflows to zero under cosine (f8 codes decode to denormals; a
    zero vector) give NaN distances.  The reference's heaps compare NaN priorities with `<` (false both ways), so their order is
    not defined there; the canonical order used here is the order of the score's IEEE bits (NaN after +Inf) on the GPU AND in the
    oracle — FLAT farthest-k / nearest-k and an HNSW walk over such data agree bit for bit."""
    n, d = 3000, 4
    X = O.fill_normal(2100, (n, d)); X[5] = 0.0; X[77] = 0.0
    ids = np.arange(n, dtype=np.uint64)
    Q = O.fill_normal(2101, (12, d))
    for quant in (O.Q_F8, O.Q_NONE):
        gf = gpu.FlatSpace(d, O.COSINE, quant); gf.ChangedVertex(ids, X)
        of = O.Flat(d, O.COSINE, quant); of.upsert(ids, X)
        saw_nan = False
        for sel in (gpu.SELECT_REFERENCE, gpu.SELECT_NEAREST):
            gi, gs, gc = gf.VertexSearch(Q, 40, sel)
            saw_nan |= bool(np.isnan(gs).any())
            for qi in range(len(Q)):
                wi, ws = of.search(Q[qi], 40, nearest=bool(sel), mode=2)
                assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws, f"flat quant{quant} sel{sel} q{qi}")
        assert saw_nan, "the case must really contain NaN scores"
    lv = O.levels(2102, n, 8)
    gh = _gpu_build(gpu, X, lv, O.COSINE, O.Q_F8, gpu.HnswCfg.default(m=8, ef_construction=16), batch=256)
    g = gh.ExportRaw(); rows = gh.FetchRows()
    for ef in (64, 300):
        gi, gs, gc, st = gh.Search(Q, 10, ef=ef, with_stats=True)
        sl, sc, cn, ost, _ = O.csr_search(rows, O.Q_F8, g["adj0"], g["upper_off"], g["adjU"], d, O.COSINE, g["entry"], g["entry_level"], Q, 10, ef)
        for qi in range(len(Q)):
            assert_same_results(gi[qi, :gc[qi]], gs[qi, :gc[qi]], sl[qi, :cn[qi]].astype(np.uint64), sc[qi, :cn[qi]], f"hnsw ef{ef} q{qi}")
        assert st["n_dist"] == ost["n_dist"] and st["n_exp"] == ost["n_exp"]
