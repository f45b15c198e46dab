"""Product-quantised store on the GPU (pq.hip, coltt_pq_*) against the oracle's definition of the scan:
codes, table bits, ids, ranks and score bits must be EQUAL — the scan is integer / table work plus a fixed-order f32 sum."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import oracle as O  # noqa: E402
import make_golden_pq as MG  # noqa: E402
from util import bits  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "pq.npz"))


def _ids(n, seed=3):
    return ((np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(2654435761)) % np.uint64(1 << 40)


def _check_search(pq, metric, cb, codes, ids, Q, k):
    gi, gs, gc = pq.Search(Q, k)
    oi, os_, oc, _ = O.pq_search(metric, cb, codes, Q, k, ids=ids)
    assert np.array_equal(gc.astype(np.int64), oc.astype(np.int64)), (gc, oc)
    for q in range(len(Q)):
        c = int(oc[q])
        assert np.array_equal(gi[q, :c], oi[q, :c]), (q, gi[q, :c], oi[q, :c])
        assert np.array_equal(bits(gs[q, :c]), bits(os_[q, :c])), (q, gs[q, :c], os_[q, :c])


@pytest.mark.parametrize("metric", [0, 1, 2])
@pytest.mark.parametrize("dim,m,c", [(64, 8, 256), (768, 96, 256), (24, 6, 17), (80, 2, 33), (96, 16, 256), (44, 11, 5)])
def test_codes_tables_and_search_equal_the_oracle(gpu, metric, dim, m, c):
    """dsub 8 / 8 / 4 / 40 (a 32-block + tail) / 6 / 4; rows of 8, 96, 6 -> 8, 2 -> 4, 16, 11 -> 12 codes: every piece width"""
    n = 3000
    X = O.fill_normal(500 + dim, (n, dim)); Q = O.fill_normal(501 + dim, (5, dim)); T = O.fill_normal(502 + dim, (max(c, 64), dim))
    ids = _ids(n)
    cb = O.pq_train(T, m, c, 1)
    pq = gpu.PQSpace(dim, metric, m, c)
    pq.SetCodebooks(cb)
    assert np.array_equal(pq.Codebooks().view(np.uint32), cb.view(np.uint32))
    want_codes = O.pq_encode(cb, X)
    assert np.array_equal(pq.Encode(X[:700]), want_codes[:700])
    for qi in range(2):
        assert np.array_equal(bits(pq.Lut(Q[qi])), bits(O.pq_lut(metric, cb, Q[qi]))), qi
    pq.Insert(ids, X)
    assert pq.Len() == n
    fc, fi = pq.FetchCodes()
    assert np.array_equal(fc, want_codes) and np.array_equal(fi, ids)
    for k in (1, 10, 100):
        _check_search(pq, metric, cb, want_codes, ids, Q, k)          # 5 queries: the four-queries-per-pass path (when it fits)
    _check_search(pq, metric, cb, want_codes, ids, Q[:1], 10)        # one query: the single-table path
    pq.close()


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 513, 4095, 4096, 4097, 70001, 262145])
def test_sizes_around_tiles_and_segments(gpu, n):
    dim, m, c = 32, 8, 64
    T = O.fill_normal(600, (200, dim)); cb = O.pq_train(T, m, c, 1)
    X = O.fill_normal(601 + n, (max(n, 1), dim))[:n]; Q = O.fill_normal(602, (3, dim)); ids = _ids(n, 9)
    pq = gpu.PQSpace(dim, gpu.PQ_EUCLIDEAN, m, c); pq.SetCodebooks(cb)
    if n:
        pq.Insert(ids, X)
    codes = O.pq_encode(cb, X) if n else np.zeros((0, m), np.uint8)
    for k in (1, 10):
        _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q, k)
        _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q[:1], k)
    pq.close()


def test_the_golden_fixture(gpu):
    """part A was produced by the independent pure-Python restatement, part B by the C++ oracle (tests/golden/make_golden_pq.py)"""
    for cfg, tag in ((MG.A, "a"), (MG.B, "b")):
        X, Q, T, ids = MG.inputs(cfg)
        for metric in (0, 1, 2):
            pq = gpu.PQSpace(cfg["dim"], metric, cfg["m"], cfg["c"])
            pq.Fit(T, 2 if tag == "a" else 1)
            if tag == "a":
                assert np.array_equal(pq.Codebooks().view(np.uint32), GOLD["a_codebooks_bits"])
            pq.Insert(ids, X)
            assert np.array_equal(pq.FetchCodes()[0], GOLD[f"{tag}_codes"])
            gi, gs, gc = pq.Search(Q, cfg["k"])
            for qi in range(cfg["nq"]):
                if tag == "a":
                    assert np.array_equal(bits(pq.Lut(Q[qi])), GOLD[f"a_lut_{metric}_{qi}"])
                    assert np.array_equal(gi[qi], GOLD[f"a_ids_{metric}_{qi}"]) and np.array_equal(bits(gs[qi]), GOLD[f"a_scores_{metric}_{qi}"])
                else:
                    assert np.array_equal(gi[qi], GOLD[f"b_ids_{metric}"][qi]) and np.array_equal(bits(gs[qi]), GOLD[f"b_scores_{metric}"][qi])
            pq.close()


@pytest.mark.parametrize("dim,m,c,iters", [(16, 4, 8, 3), (64, 8, 32, 2), (80, 2, 16, 2), (48, 12, 256, 1)])
def test_train_equals_the_oracle(gpu, dim, m, c, iters):
    T = O.fill_normal(700 + dim, (600, dim))
    pq = gpu.PQSpace(dim, gpu.PQ_EUCLIDEAN, m, c)
    with pytest.raises(gpu.ColttError):
        pq.Search(T[:1], 1)                               # no codebooks yet
    with pytest.raises(gpu.ColttError):
        pq.Fit(T[:c - 1], 1)                              # fewer sample vectors than centroids
    pq.Fit(T, iters)
    assert np.array_equal(pq.Codebooks().view(np.uint32), O.pq_train(T, m, c, iters).view(np.uint32))
    pq.close()


def test_mass_ties_fall_back_to_segments_that_cannot_overflow(gpu):
    """200 000 rows with ONE code pattern (every score equal: the whole store sits on the threshold) plus a few better rows: the
    candidate list overflows, the search is re-run in bounded segments, and the (score, id) winners are exact"""
    dim, m, c = 16, 4, 4
    cb = np.zeros((m, c, 4), np.float32)
    for cc in range(c):
        cb[:, cc, :] = cc
    n = 200_000
    codes = np.full((n, m), 2, np.uint8)
    codes[[5, 77_777, 150_000, 199_999]] = 1
    ids = _ids(n, 11)
    pq = gpu.PQSpace(dim, gpu.PQ_EUCLIDEAN, m, c); pq.SetCodebooks(cb)
    pq.InsertCodes(ids, codes)
    Q = np.ones((2, dim), np.float32)
    _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q, 64)
    _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q[:1], 3)
    pq.close()


def test_overwrite_remove_and_ready_codes(gpu):
    dim, m, c = 48, 12, 100
    T = O.fill_normal(800, (300, dim)); cb = O.pq_train(T, m, c, 1)
    X = O.fill_normal(801, (1000, dim)); Q = O.fill_normal(802, (4, dim)); ids = _ids(1000, 5)
    pq = gpu.PQSpace(dim, gpu.PQ_COSINE, m, c); pq.SetCodebooks(cb)
    with pytest.raises(gpu.ColttError):
        pq.InsertCodes(ids[:1], np.full((1, m), c, np.uint8))      # a code >= numCentroids
    assert pq.Len() == 0
    pq.Insert(ids, X)
    with pytest.raises(gpu.ColttError):
        pq.SetCodebooks(cb)                                         # the stored codes belong to the current codebooks
    # model of the store: id -> codes
    model = {int(i): r for i, r in zip(ids, O.pq_encode(cb, X))}
    X2 = O.fill_normal(803, (50, dim))
    up = np.concatenate([ids[100:140], np.array([1 << 41, (1 << 41) + 1, 1 << 41], np.uint64), ids[7:14]])   # overwrite, new ids, one repeated (last wins)
    pq.Insert(up, X2)
    for i, r in zip(up, O.pq_encode(cb, X2)):
        model[int(i)] = r
    rm = np.concatenate([ids[::7], np.array([12345678901234], np.uint64)])                                     # incl. an unknown id
    pq.Remove(rm)
    for i in rm:
        model.pop(int(i), None)
    ready = np.random.default_rng(3).integers(0, c, (30, m), dtype=np.uint8)
    rid = np.concatenate([ids[1:16], np.arange(15, dtype=np.uint64) + np.uint64(1 << 42)])
    rid = np.array([i for i in rid], np.uint64)
    pq.InsertCodes(rid, ready)
    for i, r in zip(rid, ready):
        model[int(i)] = r
    assert pq.Len() == len(model)
    fc, fi = pq.FetchCodes()
    assert sorted(int(i) for i in fi) == sorted(model)
    for r, i in zip(fc, fi):
        assert np.array_equal(r, model[int(i)]), int(i)
    _check_search(pq, O.PQ_COSINE, cb, fc, fi, Q, 20)
    pq.close()


def test_dense_device_ingest_and_device_search(gpu):
    import torch
    dim, m, c, n = 256, 32, 256, 50_000
    T = O.fill_normal(900, (400, dim)); cb = O.pq_train(T, m, c, 1)
    X = O.fill_normal(901, (n, dim)); Q = O.fill_normal(902, (9, dim))
    pq = gpu.PQSpace(dim, gpu.PQ_EUCLIDEAN, m, c); pq.SetCodebooks(cb)
    xd = torch.from_numpy(X).to("cuda:0"); qd = torch.from_numpy(Q).to("cuda:0")
    torch.cuda.synchronize()
    pq.InsertDevice(xd.data_ptr(), 30_000, first_id=1000)
    pq.InsertDevice(xd.data_ptr() + 30_000 * dim * 4, 20_000, first_id=31_000)          # continues the dense id range
    oi = torch.empty((9, 10), dtype=torch.int64, device="cuda:0"); osc = torch.empty((9, 10), dtype=torch.float32, device="cuda:0")
    oc = torch.empty(9, dtype=torch.int32, device="cuda:0")
    pq.SearchDevice(qd.data_ptr(), 9, 10, oi.data_ptr(), osc.data_ptr(), oc.data_ptr())
    codes = O.pq_encode(cb, X)
    wi, ws, wc, _ = O.pq_search(O.PQ_EUCLIDEAN, cb, codes, Q, 10, ids=np.arange(n, dtype=np.uint64) + np.uint64(1000), threads=4)
    assert np.array_equal(oi.cpu().numpy().astype(np.uint64), wi) and np.array_equal(bits(osc.cpu().numpy()), bits(ws))
    ms, scan_ms = pq.last_kernel_ms()
    assert ms > 0 and 0 < scan_ms <= ms
    pq.close()


def test_tables_too_large_for_lds_take_the_global_path(gpu):
    dim, m, c = 320, 160, 16            # 160 KiB of table per query: beyond a workgroup's LDS
    T = O.fill_normal(950, (100, dim)); cb = O.pq_train(T, m, c, 1)
    X = O.fill_normal(951, (2000, dim)); Q = O.fill_normal(952, (3, dim)); ids = _ids(2000, 2)
    pq = gpu.PQSpace(dim, gpu.PQ_DOT, m, c); pq.SetCodebooks(cb)
    pq.Insert(ids, X)
    _check_search(pq, O.PQ_DOT, cb, O.pq_encode(cb, X), ids, Q, 10)
    pq.close()


def test_parameters_are_validated(gpu):
    for args in ((64, 0, 8, 1), (64, 0, 8, 257), (64, 0, 1, 16), (64, 0, 7, 16), (64, 5, 8, 16)):
        with pytest.raises(gpu.ColttError):
            gpu.PQSpace(*args)
    pq = gpu.PQSpace(64, 0, 8, 16)
    with pytest.raises(ValueError):
        pq.SetCodebooks(np.zeros((8, 16, 7), np.float32))
    pq.close()


@pytest.mark.parametrize("dim,m,c", [(768, 96, 256), (96, 16, 256), (384, 48, 256), (64, 8, 32)])
def test_one_query_takes_one_scan_launch_and_equals_the_oracle_and_the_segment_chain(gpu, monkeypatch, dim, m, c):
    """Round 6: ONE query over a store of more than 65 536 rows runs table + ONE scan launch (per-wave self-tightening lists,
    pq_scan1_kernel: 1 024 / 256 / 512 / 256 threads per workgroup for these table sizes) + one selection instead of seven launches.
    Same ids, same score bits as the oracle — and as the segment chain (COLTT_PQ_ONE=0) in the same process.  k = 65 exceeds what a
    wave's list keeps: that call takes the chain by itself."""
    n = 300_000
    T = O.fill_normal(700 + m, (max(c, 64), dim)); cb = O.pq_train(T, m, c, 1)
    rng = np.random.default_rng(701 + m)
    codes = rng.integers(0, c, (n, m), dtype=np.uint8)
    ids = _ids(n, 13)
    pq = gpu.PQSpace(dim, gpu.PQ_EUCLIDEAN, m, c); pq.SetCodebooks(cb)
    pq.InsertCodes(ids, codes)
    Q = O.fill_normal(702 + m, (3, dim))
    for k in (1, 10, 64, 65):
        for qi in range(len(Q)):
            monkeypatch.setenv("COLTT_PQ_ONE", "1")
            gi, gs, gc = pq.Search(Q[qi:qi + 1], k)
            monkeypatch.setenv("COLTT_PQ_ONE", "0")
            ci, cs, cc = pq.Search(Q[qi:qi + 1], k)
            assert np.array_equal(gc, cc) and np.array_equal(gi, ci) and np.array_equal(bits(gs), bits(cs)), (k, qi)
        monkeypatch.setenv("COLTT_PQ_ONE", "1")
        _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q[:1], k)
    pq.close()


def test_one_scan_launch_with_heavy_ties(gpu, monkeypatch):
    """4 codes x 4 centroids: 256 distinct scores over 150 000 rows — every wave's list fills with ties of its threshold, the overflow
    flag sends the call to the bounded segments; a milder case (8 x 16) stays on the one-launch path.  Exact (score, id) winners both ways."""
    for dim, m, c, n in ((16, 4, 4, 150_000), (32, 8, 16, 150_000)):
        T = O.fill_normal(710 + m, (64, dim)); cb = O.pq_train(T, m, c, 1)
        rng = np.random.default_rng(711 + m)
        codes = rng.integers(0, c, (n, m), dtype=np.uint8)
        ids = _ids(n, 17)
        pq = gpu.PQSpace(dim, gpu.PQ_EUCLIDEAN, m, c); pq.SetCodebooks(cb)
        pq.InsertCodes(ids, codes)
        Q = O.fill_normal(712 + m, (2, dim))
        for k in (1, 10, 64):
            _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q[:1], k)
            _check_search(pq, O.PQ_EUCLIDEAN, cb, codes, ids, Q[1:], k)
        pq.close()
