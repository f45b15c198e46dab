#!/usr/bin/env python3
"""COLTT_HNSW_DIVERSE against the reference's k-nearest selection at the operating-point shape (768-d binary16 rows, `lowrank:32:1.0`):
the SAME vectors and level draws built once per arm, then recall@10 / queries/s / evaluations per query of the plain walk over an ef sweep,
and of the walk over 64 x 32 product-quantiser codes.  `python tools/diverse_probe.py [n] [ef,ef,..] [arm,arm,..] [pq ef,..]`;
arms: `default` (algo 0), `diverse` (algo 2, keepPruned 0), `diverse_keep` (algo 2, keepPruned 1).  One JSON line per measurement,
appended to $PROBE_OUT."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

ARMS = {"default": (0, 1), "diverse": (2, 0), "diverse_keep": (2, 1)}


def main():
    import torch
    import coltt_amd as G
    assert G.lib().coltt_init(0) == 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    efs = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "128,256,384,512,768,1024").split(",")]
    arms = (sys.argv[3] if len(sys.argv) > 3 else "default,diverse").split(",")
    pq_efs = [int(v) for v in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if v]
    spec = os.environ.get("PROBE_DATASET", "lowrank:32:1.0")
    dim, k, rq, nq, seed, quant = int(os.environ.get("PROBE_DIM", "768")), 10, 1000, 10000, 0xC0177, 1
    dev = torch.device("cuda", 0)
    out_path = os.environ.get("PROBE_OUT")

    def emit(rec):
        print(json.dumps(rec), flush=True)
        if out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")

    class A:
        m = 16; ef = 128; efc = 200; build_batch = int(os.environ.get("PROBE_BATCH", "16384")); reserve = True
    ds = B.Dataset(torch, dev, dim, spec)
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

    for arm in arms:
        algo, keep = ARMS[arm]
        h = G.Hnsw(dim, G.COSINE, G.HnswCfg.default(m=A.m, ef=A.ef, ef_construction=A.efc, algo=algo, keep_pruned=keep), quantization=quant)
        h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, seed, quant, h=h)
        g = h.ExportRaw()
        deg = (g["adj0"] != 0xffffffff).sum(axis=1)
        emit({"kind": "build", "arm": arm, "n": n, "build_s": round(build_s, 1), "mean_degree0": round(float(deg.mean()), 2), "full_rows": round(float((deg == 2 * A.m).mean()), 3)})
        del g, deg
        for ef in efs:
            h.SearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef)
            t0 = time.time(); st = h.SearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef); dt = time.time() - t0
            emit({"kind": "plain", "arm": arm, "n": n, "ef": ef, "recall": round(recall(o.ids.cpu().numpy()), 4), "qps": round(nq / dt), "kernel_ms": round(h.last_kernel_ms(), 3),
                  "n_dist": round(st["n_dist"] / nq, 1), "n_exp": round(st["n_exp"] / nq, 1)})
        for ef in (16, 32, 64, 128, 256):     # one query per call (the reference's RPC shape)
            if ef > max(efs): break
            lat = []
            for i in range(40):
                t0 = time.time(); h.SearchDevice(q.data_ptr() + i * dim * 4, 1, k, *o.ptrs(), ef=ef); lat.append(time.time() - t0)
            h.SearchDevice(q.data_ptr(), rq, k, *o.ptrs(), ef=ef)
            emit({"kind": "single", "arm": arm, "ef": ef, "recall": round(recall(o.ids.cpu().numpy()), 4), "call_ms_median": round(float(np.median(lat[8:])) * 1e3, 4)})
        if pq_efs:
            ns = min(n, 65536)
            sample = h.FetchRows(0, ns).view(np.float16).astype(np.float32)
            pq = G.PQSpace(dim, G.PQ_EUCLIDEAN, 64, 32)
            pq.Fit(sample, iterations=6); h.PqAttach(pq)
            for ef in pq_efs:
                for rr in (0, 768):
                    if rr and rr >= ef: continue
                    h.PqSearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef, rerank=rr)
                    t0 = time.time(); st = h.PqSearchDevice(q.data_ptr(), nq, k, *o.ptrs(), ef=ef, rerank=rr); dt = time.time() - t0
                    emit({"kind": "pq", "arm": arm, "n": n, "ef": ef, "rerank": rr, "recall": round(recall(o.ids.cpu().numpy()), 4), "qps": round(nq / dt),
                          "kernel_ms": round(h.last_kernel_ms(), 3), "n_dist": round(st["n_dist"] / nq, 1)})
            pq.close()
        del h
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
