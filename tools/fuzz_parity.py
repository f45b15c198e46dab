#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity: random operation sequences on FLAT stores and HNSW indexes (upsert / overwrite / remove /
search in every mode / filtered scan / save-load round trips; sequential and batched inserts / removes / searches at random ef, k)
— every answer compared bit for bit with the CPU oracle driven through the same sequence.
`python tools/fuzz_parity.py [seconds] [seed]`; exits non-zero at the first mismatch and prints the seed and the step."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def same(a_ids, a_sc, b_ids, b_sc):
    return (len(a_ids) == len(b_ids) and np.array_equal(np.asarray(a_ids, np.uint64), np.asarray(b_ids, np.uint64))
            and np.array_equal(np.ascontiguousarray(a_sc, np.float32).view(np.uint32), np.ascontiguousarray(b_sc, np.float32).view(np.uint32)))


def fuzz_flat(G, O, rng, log):
    d = int(rng.choice([7, 32, 100, 128, 130, 200, 256, 768]))
    metric = int(rng.integers(0, 2)); quant = int(rng.choice([O.Q_NONE, O.Q_F16, O.Q_F8, O.Q_BF16]))
    gf = G.FlatSpace(d, metric, quant); of = O.Flat(d, metric, quant)
    live = {}
    nxt = 1
    log(f"flat d={d} metric={metric} quant={quant}")
    for step in range(int(rng.integers(4, 10))):
        op = rng.choice(["add", "add", "add", "overwrite", "remove", "search", "search", "filter", "saveload"])
        if op == "add" or not live:
            n = int(rng.choice([1, 37, 900, 5000, 40000]))
            ids = (np.arange(nxt, nxt + n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 44); nxt += n
            X = O.fill_normal(int(rng.integers(1, 1 << 30)), (n, d))
            if rng.random() < 0.1: X[int(rng.integers(0, n))] = 0.0          # a zero vector (cosine: NaN norm path)
            if n > 10 and rng.random() < 0.3: X[1:4] = X[0]                   # exact duplicates: ties
            gf.ChangedVertex(ids, X); of.upsert(ids, X)
            for i in ids: live[int(i)] = 1
        elif op == "overwrite":
            ks = np.array(list(live)[:: max(1, len(live) // 50)], dtype=np.uint64)
            X = O.fill_normal(int(rng.integers(1, 1 << 30)), (len(ks), d))
            gf.ChangedVertex(ks, X); of.upsert(ks, X)
        elif op == "remove":
            ks = np.array(list(live)[int(rng.integers(0, 7)):: max(2, len(live) // 40)], dtype=np.uint64)
            if len(ks) == len(live): ks = ks[:-1]
            if len(ks):
                gf.RemoveVertex(ks); of.remove(ks)
                for i in ks: live.pop(int(i))
        elif op == "saveload":
            blob = gf.SaveVertex()
            g2 = G.FlatSpace(d, metric, quant); g2.LoadVertex(blob)
            assert g2.LoadSize() == gf.LoadSize() == len(of), ("saveload size", step)
            gf = g2
        else:
            nq = int(rng.choice([1, 3, 17, 70, 200])); k = int(rng.choice([1, 5, 10, 64, 300]))
            Q = O.fill_normal(int(rng.integers(1, 1 << 30)), (nq, d))
            if op == "filter":
                cand = np.array(list(live)[:: max(1, len(live) // int(rng.integers(1, 400) + 1))], dtype=np.uint64)
                sel = int(rng.integers(0, 2))
                gi, gs, gc = gf.FilterableVertexSearch(cand, Q, k, sel)
                for qi in range(min(nq, 6)):
                    wi, ws = of.search(Q[qi], k, nearest=bool(sel), mode=2, cand=cand)
                    assert same(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws), ("filter", step, qi, d, metric, quant, k, sel)
            else:
                for sel in (0, 1):
                    for mode in (G.MODE_EXACT, G.MODE_MFMA):
                        gi, gs, gc = gf.VertexSearch(Q, k, sel, mode)
                        for qi in range(min(nq, 5)):
                            wi, ws = of.search(Q[qi], k, nearest=bool(sel), mode=2)
                            assert same(gi[qi, :gc[qi]], gs[qi, :gc[qi]], wi, ws), ("search", step, qi, d, metric, quant, k, sel, mode, len(live))
        assert gf.LoadSize() == len(of) == len(live), ("size", step)
    gf.close()


def fuzz_hnsw(G, O, rng, log):
    import torch
    d = int(rng.choice([4, 24, 64, 128, 768]))
    metric = int(rng.integers(0, 2)); quant = int(rng.choice([O.Q_NONE, O.Q_NONE, O.Q_F16, O.Q_BF16, O.Q_F8]))
    m = int(rng.choice([4, 8, 16])); efc = int(rng.choice([16, 40, 100, 200])); algo = 0   # (the CSR oracle driver runs the Simple select; Heuristic is covered by tests/)
    n = int(rng.choice([60, 400, 3000] if d >= 128 else [60, 400, 3000, 12000]))
    cfg_g = G.HnswCfg.default(m=m, ef_construction=efc, algo=algo, ef=int(rng.choice([8, 20, 64])))
    X = O.fill_normal(int(rng.integers(1, 1 << 30)), (n, d)); lv = O.levels(int(rng.integers(1, 1 << 30)), n, m)
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(11400714819323198485 % (1 << 63))) % np.uint64(1 << 50)
    log(f"hnsw n={n} d={d} metric={metric} quant={quant} m={m} efc={efc} algo={algo}")
    # the oracle decodes what the GPU stores: build the oracle over the lowered vectors with the matching metric (as the tests do)
    sequential = n <= 400 and rng.random() < 0.5
    gh = G.Hnsw(d, metric, cfg_g, quantization=quant)
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    if sequential:
        gh.InsertBatchDevice(xd.data_ptr(), n, lv, batch=1, ids=ids)
    else:
        i = 0
        while i < n:
            b = int(min(n - i, max(1, min(int(rng.choice([16, 256, 4096])), i // 16))))
            gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, ids=ids[i:i + b]); i += b
    removed = []
    if rng.random() < 0.5:
        for j in rng.choice(n, size=min(n // 3, int(rng.integers(1, 60))), replace=False):
            gh.Remove(int(ids[j])); removed.append(int(j))
    g = gh.ExportRaw(); rows = gh.FetchRows()
    del_bits = None
    if removed:
        del_bits = np.zeros((n + 31) // 32, np.uint32)
        for j in removed: del_bits[j >> 5] |= np.uint32(1 << (j & 31))
    if g["entry"] < 0:      # everything reachable was removed: nothing to search
        gh.close(); return
    Q = O.fill_normal(int(rng.integers(1, 1 << 30)), (int(rng.choice([1, 9, 40])), d))
    for ef in (int(rng.choice([1, 7, 20])), int(rng.choice([64, 128])), int(rng.choice([129, 300, 1000, 4096]))):
        k = int(rng.choice([1, 10, min(ef, 50)])); k = min(k, max(ef, 1))
        for visg in ("0", "1", None):
            if visg is None: os.environ.pop("COLTT_VISG", None)
            else: os.environ["COLTT_VISG"] = visg
            gi, gs, gc, st = gh.Search(Q, k, ef=ef, with_stats=True)
            sl, sc, cn, ost, _ = O.csr_search(rows, quant, g["adj0"], g["upper_off"], g["adjU"], d, metric, g["entry"], g["entry_level"], Q, k, max(ef, k),
                                              del_bits=del_bits, threads=2)
            for qi in range(len(Q)):
                want = ids[sl[qi, :cn[qi]].astype(np.int64)]          # slot = insertion index
                assert same(gi[qi, :gc[qi]], gs[qi, :gc[qi]], want, sc[qi, :cn[qi]]), ("hnsw search", qi, ef, k, visg, n, d, metric, quant, m, efc, algo, len(removed))
            assert st["n_dist"] == ost["n_dist"] and st["n_exp"] == ost["n_exp"], ("counters", ef, k, visg, st, ost)
    os.environ.pop("COLTT_VISG", None)
    # the product-quantised walk over the same graph (hnsw_pq.hpp; oracle definition coltt_oracle.cpp "Product-quantised HNSW"): tombstones, both
    # visited sets, partial re-ranks, every supported table metric
    if quant != O.Q_F8 and d % 4 == 0 and d >= 8 and n >= 60:
        seen = rows if quant == O.Q_NONE else O.f16_decode(rows)
        pm = int(rng.choice([v for v in (2, 4, 8, 16, 32, 64, 96, 128) if d % v == 0 and v <= d]))
        pc = int(rng.choice([c for c in (5, 16, 64, 256) if c <= n]))
        pqm = int(rng.choice([G.PQ_EUCLIDEAN, G.PQ_COSINE])) if metric == O.COSINE else G.PQ_EUCLIDEAN
        pq = G.PQSpace(d, pqm, pm, pc); pq.Fit(seen[: max(pc, min(n, 1500))], iterations=int(rng.integers(1, 4)))
        gh.PqAttach(pq); cb = pq.Codebooks(); codes = gh.PqCodes()
        assert np.array_equal(codes, O.pq_encode(cb, seen)), ("pq codes", n, d, pm, pc)
        for ef in (int(rng.choice([3, 20, 100])), int(rng.choice([129, 400, 1500]))):
            k = int(rng.choice([1, 10, min(ef, 40)])); rr = int(rng.choice([0, 0, k, 2 * k + 1, ef]))
            gi, gs, gc, st = gh.PqSearch(Q, k, ef=ef, rerank=rr, with_stats=True)
            sl, sc, cn, ost, _ = O.csr_search_pq(rows, quant, g["adj0"], g["upper_off"], g["adjU"], d, metric, g["entry"], g["entry_level"], codes, cb, pqm, Q, k, max(ef, k),
                                                 rerank=rr, del_bits=del_bits, threads=2)
            for qi in range(len(Q)):
                want = ids[sl[qi, :cn[qi]].astype(np.int64)]
                assert same(gi[qi, :gc[qi]], gs[qi, :gc[qi]], want, sc[qi, :cn[qi]]), ("pq walk", qi, ef, k, rr, n, d, metric, quant, pm, pc, pqm, len(removed))
            assert st == ost, ("pq counters", ef, k, rr, st, ost)
        pq.close()
    # Commit -> Load round trip keeps answers (the reference stream stores f32 vectors: f32 indexes only)
    if quant == O.Q_NONE:
        # (Load renumbers the slots in stream order — tombstoned vertices are not stored — so the canonical neighbour order, and with
        #  it an approximate answer, may differ from the index that was committed: the loaded index is compared with the ORACLE
        #  loaded from the same bytes, which is what "drop-in" means for a stream written by either side)
        blob = gh.Commit()
        g2 = G.Hnsw(d, metric, cfg_g, quantization=quant); g2.Load(blob)
        assert g2.Len() == gh.Len(), "commit/load size"
        oh = O.Hnsw(d, metric, O.default_cfg(m=m, efConstruction=efc, ef=cfg_g.ef)); oh.load_stream(blob)
        b = g2.Search(Q, 5, ef=50)
        for qi in range(len(Q)):
            wi, ws = oh.search(Q[qi], 5, mode=1, ef=50)
            assert same(b[0][qi, :b[2][qi]], b[1][qi, :b[2][qi]], wi, ws), ("commit/load vs oracle", qi)
        g2.close()
    gh.close()


def fuzz_hnsw_build(G, O, rng, log):
    """GRAPH parity: the GPU builder driven through a random sequence of batched inserts (batch 1 = the reference's sequential
    Insert) and removes — the oracle goes through the same sequence — then every level, edge list, stored edge distance, tombstone
    and the entrypoint must be equal (f32 rows; quantised rows are covered by the search rounds)."""
    import torch
    d = int(rng.choice([3, 16, 48, 128])); metric = int(rng.integers(0, 2))
    m = int(rng.choice([4, 8, 16])); efc = int(rng.choice([8, 32, 64, 200]))
    n = int(rng.choice([40, 300, 2500]))
    X = O.fill_normal(int(rng.integers(1, 1 << 30)), (n, d)); lv = O.levels(int(rng.integers(1, 1 << 30)), n, m)
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(7919) + np.uint64(17)) % np.uint64(1 << 40)
    log(f"build n={n} d={d} metric={metric} m={m} efc={efc}")
    gh = G.Hnsw(d, metric, G.HnswCfg.default(m=m, ef_construction=efc))
    oh = O.Hnsw(d, metric, O.default_cfg(m=m, efConstruction=efc))
    xd = torch.from_numpy(X).cuda(); torch.cuda.synchronize()
    i = 0; live = []
    while i < n:
        if live and rng.random() < 0.25:     # remove a few (possibly the entrypoint, possibly everything that is left)
            for _ in range(int(rng.integers(1, 6))):
                if not live: break
                j = live.pop(int(rng.integers(0, len(live))))
                gh.Remove(int(ids[j])); assert oh.remove(int(ids[j])) == 0
        b = int(min(n - i, rng.choice([1, 1, 2, 7, 64, 512])))
        if b > 1: b = int(max(1, min(b, max(1, len(live) // 8))))      # a batch is linked against the graph before it: keep it a fraction of it
        gh.InsertBatchDevice(xd.data_ptr() + i * d * 4, b, lv[i:i + b], batch=b, ids=ids[i:i + b])
        oh.insert_batched(ids[i:i + b], X[i:i + b], lv[i:i + b], b)
        live.extend(range(i, i + b)); i += b
    go = gh.Export(); oo = oh.export(with_vectors=False)
    for key in ("ids", "levels", "deleted", "row_offsets", "nbr"):
        assert np.array_equal(go[key], oo[key]), ("graph", key)
    assert np.array_equal(go["nbr_dist"].view(np.uint32), oo["nbr_dist"].view(np.uint32)), "edge distances"
    assert go["entry"] == oo["entry"] and gh.Len() == len(oh) == len(live), "entry / len"
    gh.close()


def fuzz_group(G, O, rng, log):
    """a FLAT collection group of 2-5 members on this one GPU (ShardVertex routing, host exchange, k-way merge) against the ORACLE
    store holding the same vertices, through random upserts / overwrites / removes"""
    from coltt_amd import group as GG
    d = int(rng.choice([8, 96, 128, 256])); metric = int(rng.integers(0, 2)); quant = int(rng.choice([O.Q_NONE, O.Q_F16, O.Q_BF16]))
    nm = int(rng.integers(2, 6)); layout = GG.LAYOUT_SHARD if rng.random() < 0.7 else GG.LAYOUT_REPLICA
    log(f"group members={nm} layout={layout} d={d} metric={metric} quant={quant}")
    grp = G.Group([0] * nm, d, metric, quant, kind=GG.GROUP_FLAT, layout=layout)
    of = O.Flat(d, metric, quant)
    live = {}; nxt = 1
    for step in range(int(rng.integers(3, 8))):
        op = rng.choice(["add", "add", "overwrite", "remove", "search", "search"])
        if op == "add" or not live:
            n = int(rng.choice([1, 50, 3000, 20000]))
            ids = (np.arange(nxt, nxt + n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15 % (1 << 62))) % np.uint64(1 << 48); nxt += n
            X = O.fill_normal(int(rng.integers(1, 1 << 30)), (n, d))
            grp.ChangedVertex(ids, X); of.upsert(ids, X)
            for i in ids: live[int(i)] = 1
        elif op == "overwrite":
            ks = np.array(list(live)[:: max(1, len(live) // 30)], dtype=np.uint64)
            X = O.fill_normal(int(rng.integers(1, 1 << 30)), (len(ks), d))
            grp.ChangedVertex(ks, X); of.upsert(ks, X)
        elif op == "remove":
            ks = np.array(list(live)[1:: max(2, len(live) // 25)], dtype=np.uint64)
            if len(ks) and len(ks) < len(live):
                grp.RemoveVertex(ks) if hasattr(grp, "RemoveVertex") else grp.Remove(ks); of.remove(ks)
                for i in ks: live.pop(int(i))
        else:
            nq = int(rng.choice([1, 9, 80])); k = int(rng.choice([1, 10, 100])); Q = O.fill_normal(int(rng.integers(1, 1 << 30)), (nq, d))
            for sel in (0, 1):
                for mode in (G.MODE_EXACT, G.MODE_MFMA):
                    a = grp.Search(Q, k, select=sel, mode=mode)
                    for qi in range(min(nq, 5)):
                        wi, ws = of.search(Q[qi], k, nearest=bool(sel), mode=2)
                        assert same(a[0][qi, :a[2][qi]], a[1][qi, :a[2][qi]], wi, ws), ("group search", step, qi, k, sel, mode, len(live))
        assert grp.Len() == len(live), ("group size", step, grp.Len(), len(live))
    grp.close()


def fuzz_pq(G, O, rng, log):
    """product-quantised store (pq.hip): random shape, trained or installed codebooks, vectors / ready codes in, overwrites, removals,
    searches at random nq / k — codes, ids and score bits against the oracle's definition of the scan"""
    m = int(rng.choice([2, 4, 6, 8, 12, 16, 32, 48]))
    dsub = int(rng.choice([1, 2, 4, 5, 8, 16, 32, 40]))
    c = int(rng.choice([2, 5, 16, 17, 100, 256]))
    d = m * dsub
    metric = int(rng.integers(0, 3))
    pq = G.PQSpace(d, metric, m, c)
    T = O.fill_normal(int(rng.integers(1, 1 << 30)), (max(c, 64) + int(rng.integers(0, 200)), d))
    iters = int(rng.integers(0, 4))
    log(f"pq d={d} m={m} c={c} dsub={dsub} metric={metric} iters={iters}")
    if rng.random() < 0.5:
        pq.Fit(T, iters); cb = O.pq_train(T, m, c, iters)
        assert np.array_equal(pq.Codebooks().view(np.uint32), cb.view(np.uint32)), "trained codebooks"
    else:
        cb = O.pq_train(T, m, c, iters)
        if rng.random() < 0.3: cb[:, 1] = cb[:, 0]          # duplicate centroids: Encode ties
        pq.SetCodebooks(cb)
    model = {}
    nxt = 1
    for step in range(int(rng.integers(3, 9))):
        op = rng.choice(["add", "add", "codes", "overwrite", "remove", "search", "search"])
        if op in ("add", "codes") or not model:
            n = int(rng.choice([1, 63, 65, 700, 5000, 70000]))
            ids = (np.arange(nxt, nxt + n, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(1 << 44); nxt += n
            if op == "codes":
                codes = rng.integers(0, c, (n, m), dtype=np.uint8)
                if n > 5 and rng.random() < 0.3: codes[1:5] = codes[0]
                pq.InsertCodes(ids, codes)
            else:
                X = O.fill_normal(int(rng.integers(1, 1 << 30)), (n, d))
                if n > 5 and rng.random() < 0.3: X[1:5] = X[0]
                pq.Insert(ids, X); codes = O.pq_encode(cb, X)
            for i, r in zip(ids, codes): model[int(i)] = r
        elif op == "overwrite":
            ks = np.array(list(model)[:: max(1, len(model) // 40)], dtype=np.uint64)
            X = O.fill_normal(int(rng.integers(1, 1 << 30)), (len(ks), d))
            pq.Insert(ks, X)
            for i, r in zip(ks, O.pq_encode(cb, X)): model[int(i)] = r
        elif op == "remove":
            ks = np.array(list(model)[int(rng.integers(0, 5)):: max(2, len(model) // 30)], dtype=np.uint64)
            if len(ks) == len(model): ks = ks[:-1]
            if len(ks):
                pq.Remove(ks)
                for i in ks: model.pop(int(i))
        else:
            fc, fi = pq.FetchCodes()
            assert len(fi) == len(model) and all(np.array_equal(r, model[int(i)]) for r, i in zip(fc[:: max(1, len(fi) // 200)], fi[:: max(1, len(fi) // 200)])), ("codes", step)
            nq = int(rng.choice([1, 2, 5, 33])); k = int(rng.choice([1, 10, 64, 200]))
            Q = O.fill_normal(int(rng.integers(1, 1 << 30)), (nq, d))
            gi, gs, gc = pq.Search(Q, k)
            wi, ws, wc, _ = O.pq_search(metric, cb, fc, Q, k, ids=fi, threads=4)
            for qi in range(nq):
                n_ = int(wc[qi])
                assert gc[qi] == n_ and same(gi[qi, :n_], gs[qi, :n_], wi[qi, :n_], ws[qi, :n_]), ("pq search", step, qi, nq, k)
        assert pq.Len() == len(model), ("pq size", step)
    pq.close()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time()) & 0xffffff
    import coltt_amd as G
    from oracle import oracle as O
    assert G.lib().coltt_init(0) == 0
    t0 = time.time(); rounds = 0
    while time.time() - t0 < budget:
        rs = seed + rounds
        rng = np.random.default_rng(rs)
        msgs = []
        try:
            (fuzz_flat, fuzz_hnsw, fuzz_group, fuzz_hnsw_build, fuzz_pq)[rounds % 5](G, O, rng, msgs.append)
        except Exception as e:
            print(f"FUZZ FAILURE seed={rs} {' | '.join(msgs)}: {type(e).__name__}: {e}", flush=True)
            raise
        rounds += 1
    print(f"fuzz ok: {rounds} rounds in {time.time() - t0:.0f} s, seeds {seed}..{seed + rounds - 1}")


if __name__ == "__main__":
    main()
