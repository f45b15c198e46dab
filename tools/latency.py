#!/usr/bin/env python3
"""Single-query / small-batch latency of Hnsw.Search on the GPU: one wave per query vs the 256-thread latency kernel
(coltt_amd/csrc/hnsw_lat.hpp, COLTT_LAT_MAX_NQ).  `python tools/latency.py [n] [quant]` — builds n x 768 (default 10 M f32) with the batched builder, then times
nq in {1, 4, 16, 64, 128} queries per call (kernel time from the hipEvent pair on the search stream, and wall time of the call)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import coltt_amd as G
    import bench as B
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    quant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dim, k, ef = 768, 10, 128
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)

    class A: m = 16; ef = 128; efc = 200; build_batch = 16384
    ds = B.Dataset(torch, dev, dim, "normal")
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, 0xC0177, quant)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    q = ds.rows(4096, gen)
    out = B.Out(torch, dev, 4096, k)
    res = {"n": n, "dim": dim, "quant": quant, "ef": ef, "build_s": build_s, "rows": []}
    ref = {}
    for mw in ("0", "128"):
        os.environ["COLTT_LAT_MAX_NQ"] = mw
        for nq in (1, 4, 16, 64, 128):
            ms, wall = [], []
            for r in range(40):
                off = (r * nq) % (4096 - nq)
                t0 = time.perf_counter()
                h.SearchDevice(q.data_ptr() + off * dim * 4, nq, k, *out.ptrs(), ef=ef)
                wall.append((time.perf_counter() - t0) * 1e3); ms.append(h.last_kernel_ms())
            h.SearchDevice(q.data_ptr(), nq, k, *out.ptrs(), ef=ef)
            ids = out.ids[:nq].cpu().numpy().copy(); sc = out.sc[:nq].cpu().numpy().copy()
            if mw == "0": ref[nq] = (ids, sc)
            same = bool(np.array_equal(ids, ref[nq][0]) and np.array_equal(sc.view(np.uint32), ref[nq][1].view(np.uint32)))
            res["rows"].append({"kernel": "256-thread latency kernel (hnsw_lat.hpp)" if mw != "0" else "one wave per query", "nq": nq, "kernel_ms_median": float(np.median(ms[5:])),
                                "call_wall_ms_median": float(np.median(wall[5:])), "equals_one_wave_answers": same})
            print(res["rows"][-1], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
