#!/usr/bin/env python3
"""Single-query latency of the 256-thread latency kernel with 0 .. 3 cache-warming helper workgroups per walking workgroup (hnsw_lat.hpp; COLTT_LAT_HELPERS), one
index, one process: `python tools/lat_helpers_ab.py [n] [quant] [dataset]`.  Answers must be identical whatever the helpers do."""
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
    spec = sys.argv[3] if len(sys.argv) > 3 else "normal"
    dim, k = 768, 10
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)

    class A: m = 16; ef = 128; efc = 200; build_batch = 16384
    ds = B.Dataset(torch, dev, dim, spec)
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, 0xC0177, quant)
    gen = torch.Generator(device=dev); gen.manual_seed(5)
    q = ds.rows(1024, gen)
    out = B.Out(torch, dev, 8, k)
    ref = {}
    for ef in (128, 512):
        for nq in (1, 4, 8):
            for helpers in ("0", "1", "2", "3", "0"):
                os.environ["COLTT_LAT_HELPERS"] = helpers
                ms, wall, same = [], [], True
                for r in range(120):
                    off = (r * nq) % (1024 - nq)
                    t0 = time.perf_counter()
                    h.SearchDevice(q.data_ptr() + off * dim * 4, nq, k, *out.ptrs(), ef=ef)
                    wall.append((time.perf_counter() - t0) * 1e3); ms.append(h.last_kernel_ms())
                    if r < 40:
                        key = (ef, nq, r); cur = (out.ids[:nq].cpu().numpy().copy(), out.sc[:nq].cpu().numpy().view(np.uint32).copy())
                        if key in ref: same = same and np.array_equal(cur[0], ref[key][0]) and np.array_equal(cur[1], ref[key][1])
                        else: ref[key] = cur
                print(json.dumps({"n": n, "quant": quant, "dataset": spec, "ef": ef, "nq": nq, "helpers": int(helpers), "kernel_ms_median": round(float(np.median(ms[10:])), 4),
                                  "kernel_ms_p90": round(float(np.percentile(ms[10:], 90)), 4), "call_wall_ms_median": round(float(np.median(wall[10:])), 4), "answers_identical": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
