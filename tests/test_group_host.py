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
