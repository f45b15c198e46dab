#!/usr/bin/env python3
"""The product-quantiser scan by itself: `python tools/pq_probe.py [n] [dim] [m]` fills a store with n random-normal
vectors encoded on the GPU (default 10 M x 768 as 96 one-byte codes), runs single-query searches and one 64-query batch, prints one JSON
line.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel record (profiles/r04_pq_kernel_stats.csv)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import coltt_amd as G
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 96
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    pq = G.PQSpace(dim, G.PQ_EUCLIDEAN, m, 256)
    pq.Fit(torch.randn((10_000, dim), device=dev, generator=gen).cpu().numpy(), 4)
    done = 0
    while done < n:
        c = min(1 << 20, n - done)
        x = torch.randn((c, dim), device=dev, dtype=torch.float32, generator=gen); torch.cuda.synchronize()
        pq.InsertDevice(x.data_ptr(), c, first_id=done); done += c; del x
    q = torch.randn((64, dim), device=dev, dtype=torch.float32, generator=gen); torch.cuda.synchronize()
    k = 10
    oi = torch.empty((64, k), device=dev, dtype=torch.int64); osc = torch.empty((64, k), device=dev, dtype=torch.float32)
    oc = torch.empty((64,), device=dev, dtype=torch.int32)
    res = {"n": n, "dim": dim, "m": m, "rows": {}}
    for nq in (1, 4, 64):
        ms, scan = [], []
        for r in range(12):
            pq.SearchDevice(q.data_ptr(), nq, k, oi.data_ptr(), osc.data_ptr(), oc.data_ptr())
            a, b = pq.last_kernel_ms()
            if r >= 2:
                ms.append(a); scan.append(b)
        rows = pq.last_scan_rows
        mp = (m + 3) & ~3
        qb = 4 if (nq >= 2 and mp * 1024 * 4 <= 152 * 1024) else 1          # queries per pass over the code stream (pq.hip: launch_scan)
        passes = (nq + qb - 1) // qb
        t = float(np.median(scan)) / 1e3
        res["rows"][str(nq)] = {"search_ms": float(np.median(ms)), "dominant_scan_launch_ms": t * 1e3, "scan_rows": int(rows), "passes_over_the_codes": passes,
                                "streamed_TBps": rows * mp * passes / t / 1e12, "frac_of_hbm_peak": rows * mp * passes / t / 8e12,
                                "table_lookups_per_s_T": rows * m * nq / t / 1e12}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
