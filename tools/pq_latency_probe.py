#!/usr/bin/env python3
"""One query at a time on the operating-point collection (10 M x 768 f16, lowrank:32:1.0): latency and recall@10 of the plain walk
(coltt_hnsw_search_device, the 256-thread latency kernel) and of the product-quantised walk + exact re-rank (coltt_hnsw_pq_search_device)
per ef ($PROBE_PLAIN_EFS / $PROBE_PQ_EFS: comma-separated lists).  DESIGN §11.2 asks whether the table walk is the shorter chain for a single query.  `python tools/pq_latency_probe.py [n]`;
one JSON line per configuration, appended to $PROBE_OUT."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402


def main():
    import torch
    import coltt_amd as G
    assert G.lib().coltt_init(0) == 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dim, k, rq, seed, quant = 768, 10, 200, 0xC0177, 1
    dev = torch.device("cuda", 0)
    out_path = os.environ.get("PROBE_OUT")

    def emit(rec):
        print(json.dumps(rec), flush=True)
        if out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    class A:
        m = 16; ef = 128; efc = 200; build_batch = 16384; reserve = True
    ds = B.Dataset(torch, dev, dim, os.environ.get("PROBE_DATASET", "lowrank:32:1.0"))
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, seed, quant)
    gq = torch.Generator(device=dev); gq.manual_seed(0x5EED5)
    q = ds.rows(rq, gq)
    fl = B.fill_flat(G, torch, dev, ds, n, dim, quant, seed)
    t = B.Out(torch, dev, rq, k)
    fl.VertexSearchDevice(q.data_ptr(), rq, k, *t.ptrs(), select=G.SELECT_NEAREST)
    truth = t.ids.cpu().numpy()
    del fl
    o = B.Out(torch, dev, 1, k)
    row_bytes = dim * 4   # queries are f32

    def run(fn):
        ids = np.zeros((rq, k), np.uint64); wall = []; kern = []
        for i in range(rq):
            t0 = time.perf_counter(); fn(q.data_ptr() + i * row_bytes); wall.append((time.perf_counter() - t0) * 1e3)
            kern.append(h.last_kernel_ms()); ids[i] = o.ids.cpu().numpy()[0]
        rec = sum(len(set(truth[i].tolist()) & set(ids[i].tolist())) for i in range(rq)) / (rq * k)
        return round(rec, 4), round(float(np.median(wall)), 4), round(float(np.median(kern)), 4)

    for ef in [int(v) for v in os.environ.get("PROBE_PLAIN_EFS", "128,256,512").split(",") if v]:
        r, w, km = run(lambda p: h.SearchDevice(p, 1, k, *o.ptrs(), ef=ef))
        emit({"kind": "plain", "n": n, "ef": ef, "recall": r, "call_wall_ms_median": w, "kernel_ms_median": km})
    sample = h.FetchRows(0, min(n, 65536)).view(np.float16).astype(np.float32)
    pq = G.PQSpace(dim, G.PQ_EUCLIDEAN, 64, 32); pq.Fit(sample, iterations=6); h.PqAttach(pq)
    for ef in [int(v) for v in os.environ.get("PROBE_PQ_EFS", "128,192,256,384,512,768").split(",") if v]:
        r, w, km = run(lambda p: h.PqSearchDevice(p, 1, k, *o.ptrs(), ef=ef, rerank=0))
        emit({"kind": "pq 64x32", "n": n, "ef": ef, "recall": r, "call_wall_ms_median": w, "kernel_ms_median": km})
    pq.close()


if __name__ == "__main__":
    main()
