"""Round 5 on the GPU: graph quality of the batched builder against the reference's sequential Insert."""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _lowrank(seed, n, d, r, sigma):
    rng = np.random.default_rng(seed)
    basis = rng.standard_normal((r, d)).astype(np.float32)
    return (rng.standard_normal((n, r)).astype(np.float32) @ basis + sigma * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)


def _recall(h, fl, Q, k, ef, gpu):
    ti, _, _ = fl.VertexSearch(Q, k, gpu.SELECT_NEAREST)
    gi, _, gc, st = h.Search(Q, k, ef=ef, with_stats=True)
    return np.mean([len(set(gi[q, :gc[q]].tolist()) & set(ti[q].tolist())) / k for q in range(len(Q))]), st["n_dist"] / len(Q)


def test_batched_build_matches_sequential_recall(gpu):
    """The batched builder (every vertex of a batch searches the graph as it was before the batch, bench.py's schedule: <= 1/32 of the graph)
    must not cost recall against ONE Insert at a time — the reference's sequential Insert (core/vectorindex/hnsw.go:104-167, 449-474), which
    `batch=1` reproduces bit for bit.  20 000 x 128 f16 rows of a rank-16 mixture (the full-size record, 300 000 x 768 against a 543 s
    sequential build: profiles/r05_build_quality.md — equal to three digits at every ef)."""
    import torch
    n, d, k = 20000, 128, 10
    X = _lowrank(50, n, d, 16, 1.0)
    # in-distribution queries: stored rows plus noise
    Q = X[np.random.default_rng(52).choice(n, 300, replace=False)] + 0.3 * np.random.default_rng(53).standard_normal((300, d)).astype(np.float32)
    lv = O.levels(54, n)
    xd = torch.from_numpy(X).to("cuda:0"); torch.cuda.synchronize()
    fl = gpu.FlatSpace(d, gpu.COSINE, gpu.Q_F16); fl.ChangedVertex(np.arange(n, dtype=np.uint64), X)
    cfg = gpu.HnswCfg.default(m=16, ef=64, ef_construction=100)
    seq = gpu.Hnsw(d, gpu.COSINE, cfg, quantization=gpu.Q_F16)
    seq.InsertBatchDevice(xd.data_ptr(), n, lv, batch=1)
    bat = gpu.Hnsw(d, gpu.COSINE, cfg, quantization=gpu.Q_F16)
    i = 0
    while i < n:
        b = int(min(n - i, max(1, min(16384, i // 32))))
        bat.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, first_id=i)
        i += b
    assert seq.Len() == bat.Len() == n
    for ef in (16, 32, 64, 128, 256):
        rs, ns = _recall(seq, fl, Q, k, ef, gpu)
        rb, nb = _recall(bat, fl, Q, k, ef, gpu)
        print(f"\n[build quality] ef {ef}: sequential recall {rs:.4f} n_dist {ns:.0f} | batched recall {rb:.4f} n_dist {nb:.0f}")
        assert rb >= rs - 0.01, (ef, rb, rs)
        assert nb <= 1.10 * ns, (ef, nb, ns)
