"""N>1 path on CPU: world_size-2 gloo processes, each owning a collection shard (FNV ShardVertex rule), one all-gather
of per-shard top-k, host merge on rank 0.  The per-shard search is the oracle here (no GPU on this box); on the GPU box
the same plumbing carries HBM tensors over RCCL (bench.py --mode shard)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from coltt_amd import dist as D
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, d, k = 3000, 32, 10
    X = O.fill_normal(5, (n, d)); ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 33)
    Q = O.fill_normal(6, (12, d))
    mine = D.shard_mask(ids, rank, world, "fnv")
    res = {}
    for tag, nearest in (("ref", False), ("near", True)):
        f = O.Flat(d, O.COSINE, O.Q_F16); f.upsert(ids[mine], X[mine])
        li = np.zeros((len(Q), k), np.int64); ls = np.zeros((len(Q), k), np.float32); lc = np.zeros(len(Q), np.int32)
        for q in range(len(Q)):
            i, s = f.search(Q[q], k, nearest, 2)
            li[q, :len(i)] = i.astype(np.int64); ls[q, :len(i)] = s; lc[q] = len(i)
        gi, gs, gc = D.allgather_topk(torch.from_numpy(li), torch.from_numpy(ls), torch.from_numpy(lc))
        if rank == 0:
            res[tag] = D.merge_topk(gi.numpy().astype(np.uint64), gs.numpy(), gc.numpy(), k, nearest)
    # sharded HNSW: one independent graph per shard, same exchange
    lv = O.levels(7, n)
    h = O.Hnsw(d, O.COSINE); h.insert_many(ids[mine], X[mine], lv[mine])
    li = np.zeros((len(Q), k), np.int64); ls = np.zeros((len(Q), k), np.float32); lc = np.zeros(len(Q), np.int32)
    for q in range(len(Q)):
        i, s = h.search(Q[q], k, mode=1, ef=64)
        li[q, :len(i)] = i.astype(np.int64); ls[q, :len(i)] = s; lc[q] = len(i)
    gi, gs, gc = D.allgather_topk(torch.from_numpy(li), torch.from_numpy(ls), torch.from_numpy(lc))
    if rank == 0:
        res["hnsw"] = D.merge_topk(gi.numpy().astype(np.uint64), gs.numpy(), gc.numpy(), k, True)
        np.savez(out, **{f"{t}_{j}": v for t, r in res.items() for j, v in enumerate(r)})
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_shard_allgather_merge(tmp_path):
    from coltt_amd import dist as D
    from oracle import oracle as O
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = np.load(out)
    n, d, k = 3000, 32, 10
    X = O.fill_normal(5, (n, d)); ids = (np.arange(n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 33)
    Q = O.fill_normal(6, (12, d))
    assert set(np.unique(D.fnv1a_shard(ids, 2)).tolist()) == {0, 1}
    assert all(int(D.fnv1a_shard(ids[i:i + 1], 16)[0]) == O.shard_vertex(int(ids[i]), 16) for i in range(0, n, 37))
    f = O.Flat(d, O.COSINE, O.Q_F16); f.upsert(ids, X)   # the unsharded collection
    for tag, nearest in (("ref", False), ("near", True)):
        for q in range(len(Q)):
            i, s = f.search(Q[q], k, nearest, 2)
            assert np.array_equal(r[f"{tag}_0"][q], i) and np.array_equal(r[f"{tag}_1"][q].view(np.uint32), s.view(np.uint32))
    # sharded HNSW is approximate per shard; the merge itself is exact: results ascending, ids unique, recall sane
    truth = [set(f.search(Q[q], k, True, 2)[0].tolist()) for q in range(len(Q))]
    hi, hs = r["hnsw_0"], r["hnsw_1"]
    assert all(np.all(np.diff(hs[q]) >= 0) and len(set(hi[q].tolist())) == k for q in range(len(Q)))
    # scores of the f32 HNSW vs the f16 FLAT differ slightly; compare id sets only
    g = O.Flat(d, O.COSINE); g.upsert(ids, X)
    rec = np.mean([len(set(g.search(Q[q], k, True, 2)[0].tolist()) & set(hi[q].tolist())) / k for q in range(len(Q))])
    assert rec > 0.9, rec


def test_merge_topk_ragged_counts():
    from coltt_amd import dist as D
    ids = np.array([[[5, 9, 0]], [[7, 2, 1]]], np.uint64); sc = np.array([[[0.1, 0.5, 0]], [[0.1, 0.2, 0.9]]], np.float32)
    cnt = np.array([[2], [3]])
    i, s, c = D.merge_topk(ids, sc, cnt, 4, nearest=True)
    assert i[0].tolist() == [5, 7, 2, 9] and c[0] == 4          # tie 0.1 broken by id
    i, s, c = D.merge_topk(ids, sc, cnt, 2, nearest=False)
    assert i[0].tolist() == [9, 1] and np.allclose(s[0], [0.5, 0.9])
    i, s, c = D.merge_topk(ids, sc, np.array([[0], [1]]), 3, nearest=True)
    assert c[0] == 1 and i[0, 0] == 7
