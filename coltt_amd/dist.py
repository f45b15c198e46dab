"""Multi-GPU layout of the search path: collection shards across ranks, ONE all-gather of per-shard
top-k (RCCL over xGMI when the tensors live in HBM, gloo on CPU in tests) and a host-side k-way merge — the shape of the
reference's own `highCpu` path (16 local queues, then one global queue: edge/none_vectorstore.go:148-178).

Nothing here touches the kernels; it is rank arithmetic + torch.distributed plumbing, testable on CPU with gloo.
"""
import numpy as np


def fnv1a_shard(ids, shard_count):
    """sharding.ShardVertex (pkg/sharding/shard.go:34-41) vectorised on the host: FNV-1a-64 of the 8 LE bytes, mod c."""
    x = np.ascontiguousarray(ids, np.uint64)
    h = np.full(x.shape, 14695981039346656037, np.uint64)
    with np.errstate(over="ignore"):
        for i in range(8):
            h ^= (x >> np.uint64(8 * i)) & np.uint64(0xFF)
            h *= np.uint64(1099511628211)
    return h % np.uint64(shard_count)


def shard_mask(ids, rank, world, how="fnv"):
    """which of `ids` live on `rank`.  'fnv' = the reference's id->shard rule, 'mod' = id % world, 'range' = contiguous."""
    ids = np.ascontiguousarray(ids, np.uint64)
    if how == "fnv":
        return fnv1a_shard(ids, world) == np.uint64(rank)
    if how == "mod":
        return (ids % np.uint64(world)) == np.uint64(rank)
    n = len(ids); per = (n + world - 1) // world
    m = np.zeros(n, bool); m[rank * per:(rank + 1) * per] = True
    return m


def merge_topk(ids, scores, counts, k, nearest=True):
    """k-way merge of per-shard results.  ids/scores: [world, nq, k]; counts: [world, nq].  Canonical (score, id) order;
    nearest=False keeps the K LARGEST (the reference FLAT direction), output ascending either way."""
    ids = np.asarray(ids); scores = np.asarray(scores); counts = np.asarray(counts)
    world, nq, kk = ids.shape
    out_i = np.zeros((nq, k), np.uint64); out_s = np.zeros((nq, k), np.float32); out_c = np.zeros(nq, np.uint32)
    for q in range(nq):
        ci = np.concatenate([ids[r, q, :counts[r, q]] for r in range(world)]).astype(np.uint64)
        cs = np.concatenate([scores[r, q, :counts[r, q]] for r in range(world)]).astype(np.float32)
        order = np.lexsort((ci, cs))            # ascending by (score, id)
        sel = order[:k] if nearest else order[max(0, len(order) - k):]
        n = len(sel)
        out_i[q, :n] = ci[sel]; out_s[q, :n] = cs[sel]; out_c[q] = n
    return out_i, out_s, out_c


def allgather_topk(ids_t, scores_t, counts_t):
    """one collective per tensor; tensors may be CUDA (RCCL) or CPU (gloo).  Returns [world, ...] stacked tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    outs = []
    for t in (ids_t, scores_t, counts_t):
        t = t.contiguous()
        buf = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, t)          # concatenated along dim 0 (the form both gloo and RCCL accept)
        outs.append(buf.view((world,) + tuple(t.shape)))
    return outs
