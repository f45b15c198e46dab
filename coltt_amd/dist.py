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
    nearest=False keeps the K LARGEST (the reference FLAT direction), output ascending either way.  Vectorised over queries."""
    ids = np.asarray(ids).astype(np.uint64); scores = np.asarray(scores, dtype=np.float32); counts = np.asarray(counts)
    world, nq, kk = ids.shape
    ci = np.transpose(ids, (1, 0, 2)).reshape(nq, world * kk)
    cs = np.transpose(scores, (1, 0, 2)).reshape(nq, world * kk).astype(np.float64)
    valid = (np.arange(kk)[None, None, :] < counts[:, :, None])
    valid = np.transpose(valid, (1, 0, 2)).reshape(nq, world * kk)
    # invalid slots sort to the far end of whichever side is NOT selected
    cs = np.where(valid, cs, np.inf if nearest else -np.inf)
    order = np.lexsort((ci, cs), axis=1)            # ascending by (score, id) per row
    n_valid = valid.sum(axis=1)
    out_c = np.minimum(n_valid, k).astype(np.uint32)
    if nearest:
        sel = order[:, :k]
    else:
        sel = order[:, max(0, world * kk - k):]
    out_i = np.take_along_axis(ci, sel, 1); out_s = np.take_along_axis(cs, sel, 1).astype(np.float32)
    if sel.shape[1] < k:
        pad = k - sel.shape[1]
        out_i = np.pad(out_i, ((0, 0), (0, pad))); out_s = np.pad(out_s, ((0, 0), (0, pad)))
    if not nearest:  # rows with fewer than k valid entries: shift the valid tail to the front
        short = np.nonzero(n_valid < k)[0]
        for q in short:
            c = int(n_valid[q]); w = out_i.shape[1]
            vi = out_i[q, w - c:].copy() if c else out_i[q, :0]; vs = out_s[q, w - c:].copy() if c else out_s[q, :0]
            out_i[q, :] = 0; out_s[q, :] = 0; out_i[q, :c] = vi; out_s[q, :c] = vs
    else:
        mask = np.arange(out_i.shape[1])[None, :] >= out_c[:, None]
        out_i[mask] = 0; out_s[mask] = 0
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
