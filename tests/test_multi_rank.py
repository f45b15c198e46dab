"""N > 1 on CPU: two (and three) processes, a gloo rendezvous, each process owning collection shards — through the PRODUCT's
multi-process pieces: the routing rule (coltt_shard_vertex_host = sharding.ShardVertex, pkg/sharding/shard.go:34-41), the
shared-memory all-gather of packed per-shard top-k (coltt_shm_*, the transport of COLTT_EXCHANGE_SHM groups: same records and
same chunking as group.hip's search) and the host-side final merge (coltt_group_merge_host: local queues -> global queue,
edge/none_vectorstore.go:148-178).  Only the per-shard search is stood in for by the oracle (there is no GPU on this box); the GPU
twin of this test runs real coltt_group_* members in two processes on one device (tests/test_gpu_group.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K = 3000, 32, 10


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data():
    from oracle import oracle as O
    X = O.fill_normal(5, (N, D)); ids = (np.arange(N, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 33)
    Q = O.fill_normal(6, (12, D)); lv = O.levels(7, N)
    return X, ids, Q, lv


def _pack(GG, search, nq):
    recs = np.zeros((nq, K), GG.REC_DTYPE)
    for q in range(nq):
        i, s = search(q)
        recs[q, :len(i)]["id"] = i; recs[q, :len(i)]["score"] = s; recs[q, :len(i)]["valid"] = 1
    return recs


def _worker(proc, ranks_of_proc, world, port, out, bytes_per_rank):
    sys.path.insert(0, ROOT)
    from coltt_amd import group as GG
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=proc, world_size=len(ranks_of_proc))
    # the id of the group is made by ONE process and handed to the others (what bench.py --mode shard does over torch.distributed)
    uid = torch.zeros(GG.UNIQUE_ID_BYTES, dtype=torch.uint8)
    if proc == 0:
        uid = torch.frombuffer(bytearray(GG.unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(uid, 0)
    my_ranks = ranks_of_proc[proc]
    ex = GG.ShmExchange(bytes(uid.numpy().tobytes()), world, len(my_ranks), my_ranks[0], bytes_per_rank)
    X, ids, Q, lv = _data()
    shard = np.array([GG.shard_vertex_host(int(i), world) for i in ids])
    res = {}
    per_q = max(1, bytes_per_rank // (K * GG.REC_DTYPE.itemsize))       # queries per chunk: group.hip splits a batch the same way
    for tag, nearest in (("ref", False), ("near", True), ("hnsw", True)):
        local = []
        for r in my_ranks:
            mine = shard == r
            if tag == "hnsw":
                h = O.Hnsw(D, O.COSINE); h.insert_many(ids[mine], X[mine], lv[mine])
                local.append(_pack(GG, lambda q: h.search(Q[q], K, mode=1, ef=64), len(Q)))
            else:
                f = O.Flat(D, O.COSINE, O.Q_F16); f.upsert(ids[mine], X[mine])
                local.append(_pack(GG, lambda q: f.search(Q[q], K, nearest, 2), len(Q)))
        local = np.stack(local)                                          # [n_local, nq, K]
        oi = np.zeros((len(Q), K), np.uint64); os_ = np.zeros((len(Q), K), np.float32); oc = np.zeros(len(Q), np.uint32)
        for q0 in range(0, len(Q), per_q):
            q1 = min(len(Q), q0 + per_q)
            allr = ex.allgather(np.ascontiguousarray(local[:, q0:q1]))   # [world, q1-q0, K], rank-major
            i, s, c = GG.merge_host(allr, world, q1 - q0, K, nearest)
            oi[q0:q1], os_[q0:q1], oc[q0:q1] = i, s, c
        res[tag] = (oi, os_, oc)
    np.savez(f"{out}.{proc}.npz", **{f"{t}_{j}": v for t, r in res.items() for j, v in enumerate(r)})
    ex.close()
    dist.barrier(); dist.destroy_process_group()


def _check(out, n_procs):
    from oracle import oracle as O
    X, ids, Q, lv = _data()
    f = O.Flat(D, O.COSINE, O.Q_F16); f.upsert(ids, X)   # the unsharded collection
    g = O.Flat(D, O.COSINE); g.upsert(ids, X)
    first = None
    for p in range(n_procs):
        r = np.load(f"{out}.{p}.npz")
        for tag, nearest in (("ref", False), ("near", True)):
            for q in range(len(Q)):
                i, s = f.search(Q[q], K, nearest, 2)
                assert r[f"{tag}_2"][q] == len(i)
                assert np.array_equal(r[f"{tag}_0"][q], i) and np.array_equal(r[f"{tag}_1"][q].view(np.uint32), s.view(np.uint32)), (p, tag, q)
        # sharded HNSW is approximate per shard; the merge itself is exact: results ascending, ids unique, recall sane
        hi, hs = r["hnsw_0"], r["hnsw_1"]
        assert all(np.all(np.diff(hs[q]) >= 0) and len(set(hi[q].tolist())) == K for q in range(len(Q)))
        rec = np.mean([len(set(g.search(Q[q], K, True, 2)[0].tolist()) & set(hi[q].tolist())) / K for q in range(len(Q))])
        assert rec > 0.9, rec
        if first is None:
            first = r
        else:   # an all-gather: every process ends up with the same merged answers
            assert all(np.array_equal(first[k_], r[k_]) for k_ in first.files)


@pytest.mark.parametrize("ranks_of_proc,bytes_per_rank", [([[0], [1]], 1 << 20),        # two processes, one shard each, one gather per batch
                                                          ([[0], [1]], 3 * K * 16),      # slots hold 3 queries: the batch travels in 4 chunks
                                                          ([[0, 1], [2]], 5 * K * 16)])  # three shards: two in one process, one in the other
def test_shards_in_several_processes_shm_allgather_and_merge(tmp_path, ranks_of_proc, bytes_per_rank):
    out = str(tmp_path / "merged")
    world = sum(len(r) for r in ranks_of_proc)
    mp.spawn(_worker, args=(ranks_of_proc, world, _free_port(), out, bytes_per_rank), nprocs=len(ranks_of_proc), join=True)
    _check(out, len(ranks_of_proc))


def test_shm_exchange_refuses_mismatched_geometry_and_times_out(monkeypatch):
    from coltt_amd import group as GG
    import coltt_amd as G
    monkeypatch.setenv("COLTT_SHM_TIMEOUT_S", "0.3")
    uid = GG.unique_id()
    with pytest.raises(G.ColttError) as e:       # a peer that never shows up: the rendezvous gives up instead of hanging
        GG.ShmExchange(uid, 2, 1, 0, 4096)
    assert "attached" in str(e.value)
    with pytest.raises(G.ColttError):
        GG.ShmExchange(uid, 2, 2, 1, 4096)       # ranks [1,3) outside a world of 2
    one = GG.ShmExchange(GG.unique_id(), 1, 1, 0, 64)   # a world of one rank: the gather is a copy
    a = np.arange(16, dtype=np.uint32).reshape(1, 16)
    assert np.array_equal(one.allgather(a), a)
    with pytest.raises(G.ColttError):
        one.allgather(np.zeros((1, 100), np.uint32))     # more bytes than the slot holds
    one.close()


def _failing_peer(proc, port, out):
    sys.path.insert(0, ROOT)
    import time
    from coltt_amd import group as GG
    import coltt_amd as G
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=proc, world_size=2)
    uid = torch.zeros(GG.UNIQUE_ID_BYTES, dtype=torch.uint8)
    if proc == 0:
        uid = torch.frombuffer(bytearray(GG.unique_id()), dtype=torch.uint8).clone()
    dist.broadcast(uid, 0)
    ex = GG.ShmExchange(bytes(uid.numpy().tobytes()), 2, 1, proc, 256)      # default timeout: 120 s
    good = np.arange(16, dtype=np.uint32).reshape(1, 16)
    assert np.array_equal(ex.allgather(good)[proc], good[0])                # generation 0 works
    dist.barrier()
    t0 = time.time(); err = ""
    try:
        if proc == 0:
            ex.allgather(np.zeros((1, 200), np.uint32))                     # 800 bytes into a 256-byte slot: refused, and the segment is marked failed
        else:
            ex.allgather(good)                                              # the peer is waiting for rank 0's contribution
    except G.ColttError as e:
        err = str(e)
    with open(f"{out}.{proc}.txt", "w") as f:
        f.write(f"{time.time() - t0:.3f}\n{err}\n")
    ex.close()
    dist.barrier(); dist.destroy_process_group()


def test_a_failing_rank_releases_its_peers_at_once(tmp_path):
    """a process that fails inside the exchange (here: a contribution larger than its slot) marks the shared segment failed, so
    the peer that is waiting for it returns an error immediately instead of spinning for COLTT_SHM_TIMEOUT_S (120 s by default)."""
    out = str(tmp_path / "fail")
    mp.spawn(_failing_peer, args=(_free_port(), out), nprocs=2, join=True)
    for p in (0, 1):
        secs, err = open(f"{out}.{p}.txt").read().split("\n")[:2]
        assert err, p
        assert float(secs) < 20.0, (p, secs)
