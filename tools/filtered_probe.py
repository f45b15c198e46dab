#!/usr/bin/env python3
"""FilterableVertexSearch (edge/none_vectorstore.go:182-253) on the GPU: kernel time per call and gathered bytes/s for several
candidate-list shapes and batch sizes — what secondary.f3 of bench.py reports, plus a random candidate list, a 1 M-candidate list
and exact-vs-MFMA mode.  Run it under `rocprofv3 --kernel-trace --stats` for the per-kernel split (profiles/r03_f3_*).

    python tools/filtered_probe.py [--n 1000000] [--dim 768] [--quant 0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--quant", type=int, default=0)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--lists", default="every_10th,random_10pct,all")
    ap.add_argument("--nqs", default="1,16,64")
    ap.add_argument("--modes", default="exact,mfma")
    a = ap.parse_args()
    import torch
    import coltt_amd as G
    import bench as B
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)
    ds = B.Dataset(torch, dev, a.dim, "normal")
    fl = B.fill_flat(G, torch, dev, ds, a.n, a.dim, a.quant, 0xC0177 + 505)
    qgen = torch.Generator(device=dev); qgen.manual_seed(0x5EED5 + 17)
    q = ds.rows(64, qgen).cpu().numpy()
    rng = np.random.default_rng(1)
    lists = {"every_10th": np.arange(0, a.n, 10, dtype=np.uint64),
             "random_10pct": np.sort(rng.choice(a.n, a.n // 10, replace=False)).astype(np.uint64),
             "all": np.arange(a.n, dtype=np.uint64)}
    rows = []
    sb = B.QBYTES[a.quant]
    for name, cand in lists.items():
        if name not in a.lists.split(","):
            continue
        for nq in [int(x) for x in a.nqs.split(",")]:
            ref = None
            for mode, mname in ((G.MODE_EXACT, "exact"), (G.MODE_MFMA, "mfma")):
                if mname not in a.modes.split(","):
                    continue
                try:
                    r = fl.FilterableVertexSearch(cand, q[:nq], a.k, G.SELECT_NEAREST, mode)
                    ms, wall = [], []
                    for _ in range(a.reps):
                        t0 = time.perf_counter(); r2 = fl.FilterableVertexSearch(cand, q[:nq], a.k, G.SELECT_NEAREST, mode)
                        wall.append(time.perf_counter() - t0); ms.append(fl.last_kernel_ms())
                    if mode == G.MODE_EXACT:
                        ref = r
                    same = None if ref is None else bool(np.array_equal(r[0], ref[0]) and np.array_equal(r[1].view(np.uint32), ref[1].view(np.uint32)))
                    km = float(np.median(ms))
                    row = {"list": name, "candidates": int(len(cand)), "nq": nq, "mode": mname, "kernels_ms": km, "call_ms": float(np.median(wall)) * 1e3,
                           "gathered_GBps": len(cand) * a.dim * sb / (km / 1e3) / 1e9, "equals_exact_mode": same, "stats": fl.Stats()}
                except Exception as e:
                    row = {"list": name, "nq": nq, "mode": mname, "error": str(e)}
                rows.append(row)
                print(json.dumps(row), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"n": a.n, "dim": a.dim, "quant": a.quant, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
