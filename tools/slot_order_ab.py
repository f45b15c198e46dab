#!/usr/bin/env python3
"""Slot order by graph locality, one A/B (DESIGN_r01-r05_history §11.1 of round 4): the operating-point index (10 M x 768 f16, lowrank:32:1.0) as
built, against THE SAME graph and rows with the slots renumbered in breadth-first order from the entrypoint over level 0 (ids travel with their vertices, so
answers are comparable by id; adjacency rows are re-sorted ascending by the new slots — the canonical neighbour order is by slot, so the permuted walk is
another legal order of the reference's map iteration, not the same traversal).  Measured per arm: the plain walk (ef 1024) and the walk over 64 x 32
product-quantiser codes (ef 1344, re-rank 768; ONE set of codebooks for both arms) — queries/s, kernel ms, recall@10 against the exact scan.
`python tools/slot_order_ab.py [n]`; one JSON line per measurement, appended to $PROBE_OUT."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402


def bfs_order(adj0, entry):
    """order[i] = the old slot visited i-th by a level-synchronous BFS over the level-0 rows (unreached vertices keep their relative order at the end)"""
    n = len(adj0); seen = np.zeros(n, bool); order = np.empty(n, np.int64); k = 0
    front = np.array([entry], np.int64); seen[entry] = True
    while len(front):
        order[k:k + len(front)] = front; k += len(front)
        nb = adj0[front].ravel()
        nb = nb[nb != 0xFFFFFFFF].astype(np.int64)
        nb = nb[~seen[nb]]
        nb = np.unique(nb)
        seen[nb] = True
        front = nb
    rest = np.nonzero(~seen)[0]
    order[k:k + len(rest)] = rest
    return order


def permute_graph(g, order):
    """the export() graph with slot i of the result = slot order[i] of g: levels / ids / tombstones travel, every adjacency row is renumbered and re-sorted
    ascending by the new slots (the canonical neighbour order), stored edge distances travel with their edges"""
    n = len(order)
    new_of = np.empty(n, np.int64); new_of[order] = np.arange(n)
    lv = g["levels"].astype(np.int64); off = np.asarray(g["row_offsets"], np.int64)
    rows_of = np.concatenate([[0], np.cumsum(lv + 1)])            # first CSR row of every old slot (a vertex of level L owns L + 1 rows)
    deg = np.diff(off)
    starts = rows_of[order]; counts = lv[order] + 1
    idx = np.repeat(starts - np.concatenate([[0], np.cumsum(counts)[:-1]]), counts) + np.arange(int(counts.sum()))   # old CSR row of every new CSR row
    deg2 = deg[idx]; off2 = np.concatenate([[0], np.cumsum(deg2)]).astype(np.int64)
    eidx = np.repeat(off[idx] - off2[:-1], deg2) + np.arange(int(deg2.sum()))                                       # old edge of every new edge
    nb2 = new_of[np.asarray(g["nbr"])[eidx].astype(np.int64)].astype(np.int32)
    nd2 = np.asarray(g["nbr_dist"])[eidx]
    rowid = np.repeat(np.arange(len(deg2)), deg2)
    p = np.lexsort((nb2, rowid))
    return {"ids": np.asarray(g["ids"])[order], "levels": np.asarray(g["levels"])[order], "deleted": np.asarray(g["deleted"])[order], "row_offsets": off2,
            "nbr": nb2[p], "nbr_dist": nd2[p], "entry": int(new_of[int(g["entry"])])}


def main():
    import torch
    import coltt_amd as G
    assert G.lib().coltt_init(0) == 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dim, k, rq, nq, seed, quant = 768, 10, 1000, 10000, 0xC0177, 1
    dev = torch.device("cuda", 0)
    out_path = os.environ.get("PROBE_OUT")

    def emit(rec):
        print(json.dumps(rec), flush=True)
        if out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    class A:
        m = 16; ef = 128; efc = 200; build_batch = 16384; reserve = True
    ds = B.Dataset(torch, dev, dim, "lowrank:32:1.0")
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, seed, quant)
    gq = torch.Generator(device=dev); gq.manual_seed(0x5EED5)
    q = ds.rows(nq, gq)
    fl = B.fill_flat(G, torch, dev, ds, n, dim, quant, seed)
    t = B.Out(torch, dev, rq, k)
    fl.VertexSearchDevice(q.data_ptr(), rq, k, *t.ptrs(), select=G.SELECT_NEAREST)
    truth = t.ids.cpu().numpy()
    del fl
    o = B.Out(torch, dev, nq, k)

    def recall(ids):
        return sum(len(set(truth[i].tolist()) & set(ids[i].tolist())) for i in range(rq)) / (rq * k)

    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(n, size=min(n, 65536), replace=False))
    rows16 = h.FetchRows()                                        # [n][dim] binary16 codes, natural element order
    sample = rows16[pick].view(np.float16).astype(np.float32)
    pq = G.PQSpace(dim, G.PQ_EUCLIDEAN, 64, 32); pq.Fit(sample, iterations=6)

    def measure(hx, arm):
        for ef in (1024,):
            hx.SearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef)
            ms = []
            for _ in range(3):
                st = hx.SearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef); ms.append(hx.last_kernel_ms())
            emit({"arm": arm, "kind": "plain", "n": n, "ef": ef, "recall": round(recall(o.ids.cpu().numpy()), 4), "qps": round(nq / (min(ms) / 1e3)), "kernel_ms": round(min(ms), 3),
                  "n_dist": round(st["n_dist"] / nq, 1)})
        hx.PqAttach(pq)
        for ef, rr in ((1344, 768), (1344, 0)):
            hx.PqSearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef, rerank=rr)
            ms = []
            for _ in range(3):
                st = hx.PqSearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef, rerank=rr); ms.append(hx.last_kernel_ms())
            emit({"arm": arm, "kind": "pq 64x32", "n": n, "ef": ef, "rerank": rr, "recall": round(recall(o.ids.cpu().numpy()), 4), "qps": round(nq / (min(ms) / 1e3)),
                  "kernel_ms": round(min(ms), 3), "n_dist": round(st["n_dist"] / nq, 1)})

    measure(h, "as built")
    # ---- the permuted twins: breadth-first order from the entrypoint, and order by coarse cluster (4096 centroids drawn from the rows, clusters
    # laid out along their similarity to the first one): a kNN graph's edges stay inside a cluster or go to a similar one
    g = h.Export(); raw = h.ExportRaw()
    h.close()

    def cluster_order(nc=4096):
        cpick = np.sort(rng.choice(n, size=nc, replace=False))
        cent = torch.from_numpy(rows16[cpick].view(np.float16).astype(np.float32)).to(dev)
        cent = torch.nn.functional.normalize(cent, dim=1).half()
        cid = np.empty(n, np.int32)
        for b0 in range(0, n, 262144):
            x = torch.from_numpy(rows16[b0:b0 + 262144].view(np.float16)).to(dev)
            cid[b0:b0 + 262144] = (x @ cent.T).argmax(dim=1).int().cpu().numpy()
        crank = np.empty(nc, np.int64)
        crank[torch.argsort((cent.float() @ cent[0].float()), descending=True).cpu().numpy()] = np.arange(nc)
        return np.argsort(crank[cid], kind="stable")

    # "same order" = the control: the graph as built, exported and bulk-loaded like the permuted twins (another set of allocations, nothing renumbered)
    arms = {"same order": lambda: np.arange(n, dtype=np.int64), "cluster order": cluster_order, "bfs order": lambda: bfs_order(raw["adj0"], int(raw["entry"]))}
    want = [a for a in os.environ.get("SLOT_ARMS", "same order,cluster order,bfs order").split(",") if a in arms]
    for arm, make in [(a, arms[a]) for a in want]:
        t0 = time.time()
        order = make()                                            # new slot i <- old slot order[i]
        g2 = permute_graph(g, order)
        vec = rows16[order].view(np.float16).astype(np.float32)   # decoded codes: BulkLoad normalises (cosine) and lowers them again
        prep_s = time.time() - t0
        h2 = G.Hnsw(dim, G.COSINE, G.HnswCfg.default(m=A.m, ef=A.ef, ef_construction=A.efc), quantization=quant)
        t0 = time.time(); h2.BulkLoad(g2, vec); load_s = time.time() - t0
        del vec, g2
        emit({"arm": arm, "kind": "prepare", "permute_s": round(prep_s, 1), "bulk_load_s": round(load_s, 1)})
        measure(h2, arm)
        h2.close()
    pq.close()


if __name__ == "__main__":
    main()
