"""Collection groups, the parts that need no device: the routing rule (sharding.ShardVertex, pkg/sharding/shard.go:34-41)
and the host-side final merge (local queues -> global queue, edge/none_vectorstore.go:148-178) of libcoltt_gpu.so,
checked against the oracle's FNV and a plain sort of the union."""
import numpy as np

from oracle import oracle as O
import coltt_amd as G
from coltt_amd import group as GG


def test_shard_vertex_host_equals_reference_fnv():
    rng = np.random.default_rng(1)
    ids = np.concatenate([np.arange(300, dtype=np.uint64), rng.integers(0, 2**63, 700).astype(np.uint64) * np.uint64(2) + np.uint64(1)])
    for c in (1, 2, 4, 8, 16, 7):
        for i in ids:
            assert GG.shard_vertex_host(int(i), c) == O.shard_vertex(int(i), c)


def _packed(world, nq, k, rng, ties):
    recs = np.zeros((world, nq, k), GG.REC_DTYPE)
    for s in range(world):
        for q in range(nq):
            c = int(rng.integers(0, k + 1))
            sc = np.sort(rng.integers(0, 12, c).astype(np.float32) if ties else rng.random(c).astype(np.float32))
            ids = rng.integers(0, 10**6, c).astype(np.uint64) * np.uint64(world) + np.uint64(s)   # disjoint across shards
            order = np.lexsort((ids, sc))
            recs[s, q, :c]["id"] = ids[order]; recs[s, q, :c]["score"] = sc[order]; recs[s, q, :c]["valid"] = 1
    return recs


def test_group_merge_host_equals_sorted_union():
    rng = np.random.default_rng(7)
    for world, nq, k, ties in ((1, 5, 10, False), (4, 40, 10, False), (8, 64, 10, True), (8, 9, 1, True), (3, 17, 33, True)):
        recs = _packed(world, nq, k, rng, ties)
        for nearest in (True, False):
            ids, sc, cnt = GG.merge_host(recs, world, nq, k, nearest)
            for q in range(nq):
                u = [(float(r["score"]), int(r["id"])) for s in range(world) for r in recs[s, q] if r["valid"]]
                u.sort()
                want = u[:k] if nearest else u[-k:]
                assert cnt[q] == len(want)
                assert [(float(sc[q, j]), int(ids[q, j])) for j in range(cnt[q])] == want, (world, q, nearest)


def test_group_merge_host_split_over_threads_equals_sorted_union():
    """batches above 1024 queries are merged by several host threads (query ranges are independent): same answers, every row filled"""
    rng = np.random.default_rng(11)
    world, nq, k = 8, 5000, 10
    sc = np.sort(rng.random((world, nq, k)).astype(np.float32), axis=2)
    recs = np.zeros((world, nq, k), GG.REC_DTYPE)
    recs["score"] = sc
    recs["id"] = (rng.integers(0, 10**6, (world, nq, k)).astype(np.uint64) * np.uint64(world) + np.arange(world, dtype=np.uint64)[:, None, None])
    cnt_in = rng.integers(0, k + 1, (world, nq))
    recs["valid"] = (np.arange(k)[None, None, :] < cnt_in[:, :, None]).astype(np.uint32)
    for nearest in (True, False):
        ids, s, cnt = GG.merge_host(recs, world, nq, k, nearest)
        for q in list(range(0, nq, 97)) + [nq - 1, nq // 8, nq // 8 - 1, nq // 8 + 1]:
            u = sorted((float(r["score"]), int(r["id"])) for w in range(world) for r in recs[w, q] if r["valid"])
            want = u[:k] if nearest else u[-k:]
            assert cnt[q] == len(want)
            assert [(float(s[q, j]), int(ids[q, j])) for j in range(cnt[q])] == want, (q, nearest)
        assert np.array_equal(cnt, np.minimum(k, cnt_in.sum(0)))


def test_normalize_host_is_the_reference_arithmetic_without_a_device():
    """coltt_normalize_host (what the Go layer's Normalize calls once per RPC): bit-identical to the oracle's restatement of
    edge.Normalize (edge/vectorstore.go:173-189), zero vector -> zeros, no HIP device needed"""
    import ctypes as C
    from coltt_amd import _lib as L
    f = L.lib().coltt_normalize_host
    rng = np.random.default_rng(3)
    for d in (1, 7, 8, 9, 128, 768, 1536):
        for scale in (1.0, 1e-20, 1e18):
            v = (rng.standard_normal(d) * scale).astype(np.float32)
            out = np.empty(d, np.float32)
            assert f(L.vp(v), C.c_uint32(d), L.vp(out)) == 0
            assert np.array_equal(out.view(np.uint32), O.normalize(v).view(np.uint32)), (d, scale)
    z = np.zeros(33, np.float32); out = np.ones(33, np.float32)
    assert f(L.vp(z), C.c_uint32(33), L.vp(out)) == 0 and not out.any()


def test_first_failed_rank_reads_the_status_every_rank_packs_into_its_records():
    """Round 6: a rank whose shard search fails still takes part in the exchange with a block of STATUS records (bits 8..31 of `valid`);
    every rank finds the same failed rank in the gathered block and returns the same error — nobody waits for an all-gather that never comes.  The merge
    itself only ever looks at bit 0."""
    import ctypes as C
    rng = np.random.default_rng(5)
    world, nq, k = 4, 6, 10
    recs = _packed(world, nq, k, rng, False)
    st = C.c_uint32(123)
    f = G.lib().coltt_group_first_failed_rank_host
    assert f(recs.ctypes.data_as(C.c_void_p), world, C.c_size_t(nq * k), C.byref(st)) == -1 and st.value == 0
    bad = recs.copy()
    bad[2]["id"] = 0; bad[2]["score"] = 0; bad[2]["valid"] = np.uint32(4 << 8)            # rank 2: no answers, status 4 in every record
    assert f(bad.ctypes.data_as(C.c_void_p), world, C.c_size_t(nq * k), C.byref(st)) == 2 and st.value == 4
    bad[1]["valid"] = np.uint32(9 << 8)
    assert f(bad.ctypes.data_as(C.c_void_p), world, C.c_size_t(nq * k), C.byref(st)) == 1 and st.value == 9   # the FIRST failed rank: the same on every rank
    ok = recs.copy(); ok["valid"] |= np.uint32(0)                                          # status 0 everywhere: the merge is what it was
    a = GG.merge_host(ok, world, nq, k, True); b = GG.merge_host(recs, world, nq, k, True)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
