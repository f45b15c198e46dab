#!/usr/bin/env python3
"""Product-quantised HNSW at the operating-point shape: recall@10 and queries/s of coltt_hnsw_pq_search (table-distance walk + exact
re-rank) against the plain walk on the same index.  `python tools/hnswpq_probe.py [n] [m,m,..] [ef,ef,..] [rerank,..]`; one JSON line per
configuration, appended to $PROBE_OUT.  $PROBE_KNOBS = "K=V;K=V|K=V|" runs every product-quantised search once per knob setting (an empty
setting = the defaults) on the same index; $PROBE_PLAIN=0 skips the plain walk."""
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
    ms = [(int(v.split(":")[0]), int(v.split(":")[1]) if ":" in v else 256) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,96").split(",")]   # m or m:centroids
    efs = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "512,1024,2048").split(",")]
    rrs = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "0,128").split(",")]
    spec = os.environ.get("PROBE_DATASET", "lowrank:32:1.0")
    dim, k, rq, nq, seed, quant = 768, 10, 1000, 10000, 0xC0177, 1
    dev = torch.device("cuda", 0)
    out_path = os.environ.get("PROBE_OUT")

    def emit(rec):
        print(json.dumps(rec), flush=True)
        if out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    class A:  # what build_index reads
        m = 16; ef = 128; efc = 200; build_batch = 16384; reserve = True
    ds = B.Dataset(torch, dev, dim, spec)
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

    knob_sets = [dict(kv.split("=") for kv in ks.split(";") if kv) for ks in os.environ.get("PROBE_KNOBS", "").split("|")]
    for ef in (efs if os.environ.get("PROBE_PLAIN", "1") != "0" else []):   # the plain walk on this box, for the ratio
        h.SearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef)
        t0 = time.time(); st = h.SearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef); dt = time.time() - t0
        emit({"kind": "plain", "n": n, "ef": ef, "recall": round(recall(o.ids.cpu().numpy()), 4), "qps": round(nq / dt), "kernel_ms": round(h.last_kernel_ms(), 3),
              "n_dist": round(st["n_dist"] / nq, 1), "build_s": round(build_s, 1)})
    # training sample: the first 65 536 stored rows as the index's distance sees them
    ns = min(n, 65536)
    sample = h.FetchRows(0, ns).view(np.float16).astype(np.float32)
    for m, nc in ms:
        for pqm, pname in ((G.PQ_EUCLIDEAN, "l2"),):
            pq = G.PQSpace(dim, pqm, m, nc)
            t0 = time.time(); pq.Fit(sample, iterations=6); fit_s = time.time() - t0
            t0 = time.time(); h.PqAttach(pq); attach_s = time.time() - t0
            for ef, rr, knobs in [(e_, r_, k_) for e_ in efs for r_ in rrs for k_ in knob_sets]:
                    try:
                        for kk in [kk for ks in knob_sets for kk in ks]:
                            os.environ.pop(kk, None)
                        os.environ.update(knobs)
                        h.PqSearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef, rerank=rr)
                        t0 = time.time(); st = h.PqSearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef, rerank=rr); dt = time.time() - t0
                        nd, nx = st["n_dist"] / nq, st["n_exact"] / nq
                        bytes_q = st["n_exp"] / nq * (32 * ((m + 15) // 16 * 16) + 32 * 4 + 32) + nd + nx * dim * 2   # block + adjacency row + probes per expansion, marks, exact rows
                        emit({"kind": "pq", "knobs": knobs, "n": n, "m": m, "C": nc, "pq_metric": pname, "ef": ef, "rerank": rr, "recall": round(recall(o.ids.cpu().numpy()), 4),
                              "qps": round(nq / dt), "kernel_ms": round(h.last_kernel_ms(), 3), "n_dist": round(nd, 1), "n_exact": round(nx, 1),
                              "MB_per_query": round(bytes_q / 1e6, 3), "GBps": round(bytes_q * nq / max(h.last_kernel_ms(), 1e-9) / 1e6, 1),
                              "fit_s": round(fit_s, 2), "attach_s": round(attach_s, 2)})
                    except Exception as e:  # noqa
                        emit({"kind": "pq", "m": m, "ef": ef, "rerank": rr, "error": str(e)[:300]})
            pq.close()


if __name__ == "__main__":
    main()
