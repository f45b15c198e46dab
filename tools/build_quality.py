#!/usr/bin/env python3
"""What does the batched GPU builder cost in recall against the reference's sequential Insert (hnsw.go:104-167, 449-474)?

Builds the SAME collection (same vectors, same levels) with several batch schedules — `1` is the reference's sequential
Insert (GPU graph == the oracle's literal Insert, tests/test_gpu_hnsw.py), `bench` is bench.py's schedule (<= 1/32 of the graph,
<= 16 384) — and records recall@10 against the exact nearest-10 (FLAT nearest mode) and n_dist per query at each ef.
`python tools/build_quality.py [n] [quant] [dataset] [schedules] [efs]`; one JSON object per schedule on stdout, appended to
$BQ_OUT as they finish (a sequential build of 300 k rows takes minutes: partial results survive a timeout)."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    quant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    spec = sys.argv[3] if len(sys.argv) > 3 else "lowrank:32:1.0"
    scheds = (sys.argv[4] if len(sys.argv) > 4 else "bench,1024,1").split(",")
    efs = [int(v) for v in (sys.argv[5] if len(sys.argv) > 5 else "64,128,256,512,1024,2048").split(",")]
    dim, k, rq, seed, m = 768, 10, 1000, 0xC0177, 16
    dev = torch.device("cuda", 0)
    ds = B.Dataset(torch, dev, dim, spec)
    gq = torch.Generator(device=dev); gq.manual_seed(0x5EED5)
    q = ds.rows(rq, gq)
    fl = B.fill_flat(G, torch, dev, ds, n, dim, quant, seed)
    t = B.Out(torch, dev, rq, k)
    fl.VertexSearchDevice(q.data_ptr(), rq, k, *t.ptrs(), select=G.SELECT_NEAREST)
    truth = t.ids.cpu().numpy()
    del fl
    levels = B.draw_levels(n, m, seed)
    out_path = os.environ.get("BQ_OUT")
    for sc in scheds:
        h = G.Hnsw(dim, G.COSINE, G.HnswCfg.default(m=m, ef=128, ef_construction=200), quantization=quant)
        h.Reserve(n)
        gen = torch.Generator(device=dev); gen.manual_seed(seed)
        chunk = min(n, 1 << 20); done = 0
        t0 = time.time()
        while done < n:
            c = min(chunk, n - done)
            x = ds.rows(c, gen)
            if sc == "1":   # one library call: the library loops one Insert at a time
                h.InsertBatchDevice(x.data_ptr(), c, levels[done:done + c], batch=1, first_id=done)
            else:
                cap, frac = (16384, 32) if sc == "bench" else (int(sc.split("/")[0]), int(sc.split("/")[1]) if "/" in sc else 32)
                i = 0
                while i < c:
                    cur = done + i
                    b = int(min(c - i, max(1, min(cap, cur // frac))))
                    h.InsertBatchDevice(x.data_ptr() + i * dim * 4, b, levels[cur:cur + b], batch=b, first_id=cur)
                    i += b
            done += c
            del x
        torch.cuda.synchronize()
        build_s = time.time() - t0
        o = B.Out(torch, dev, rq, k)
        rec = {"schedule": sc, "n": n, "quant": quant, "dataset": spec, "build_s": round(build_s, 2), "curve": {}}
        for ef in efs:
            st = h.SearchDevice(q.data_ptr(), rq, k, *o.ptrs(), ef=ef)
            ids = o.ids.cpu().numpy()
            r = sum(len(set(truth[i].tolist()) & set(ids[i].tolist())) for i in range(rq)) / (rq * k)
            rec["curve"][str(ef)] = {"recall": round(r, 4), "n_dist": round(st["n_dist"] / rq, 1), "n_exp": round(st["n_exp"] / rq, 1)}
        g = h.ExportRaw()
        deg = (g["adj0"] != 0xFFFFFFFF).sum(1)
        rec["adj0_mean_degree"] = float(deg.mean()); rec["adj0_full_rows_frac"] = float((deg == g["adj0"].shape[1]).mean())
        print(json.dumps(rec), flush=True)
        if out_path:
            with open(out_path, "a") as f:
                f.write(json.dumps(rec) + "\n")
        h.close(); del h


if __name__ == "__main__":
    main()
