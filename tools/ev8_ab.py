#!/usr/bin/env python3
"""A/B of the level-0 distance core on ONE index in ONE process: eight lanes per row over the line-transposed copy (rows8.hpp, default)
against the pair-owned rows (COLTT_EV8=0).  `python tools/ev8_ab.py [n] [quant] [dataset] [ef,ef,...]` builds n x 768 with the batched
builder, then per ef runs 10 000 queries three times per variant (kernel time from the hipEvent pair on the search stream), checks that
ids, score bits and the traversal counters are equal, and prints one JSON line with ms per launch and the fraction of the HBM peak."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import coltt_amd as G
    import bench as B
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    quant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dataset = sys.argv[3] if len(sys.argv) > 3 else "normal"
    efs = [int(e) for e in (sys.argv[4] if len(sys.argv) > 4 else "128,256,1024").split(",")]
    dim = int(os.environ.get("EV8_AB_DIM", "768")); k, nq = 10, 10_000
    build_ef = int(os.environ.get("EV8_AB_CFG_EF", "128"))
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)

    class A: m = 16; ef = build_ef; efc = 200; build_batch = 16384; reserve = not (len(sys.argv) > 5 and sys.argv[5] == "noreserve")
    ds = B.Dataset(torch, dev, dim, dataset)
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, 0xC0177, quant)
    gen = torch.Generator(device=dev); gen.manual_seed(0x5EED5)
    q = ds.rows(nq, gen)
    out = B.Out(torch, dev, nq, k)
    res = {"n": n, "dim": dim, "quant": quant, "dataset": dataset, "build_s": build_s, "reserved": A.reserve, "rows8": h.Rows8(), "ef": {}}
    for ef in efs:
        row = {}
        keep = {}
        for name, env in (("eight_lanes", None), ("lane_pairs", "0")):
            if env is None:
                os.environ.pop("COLTT_EV8", None)
            else:
                os.environ["COLTT_EV8"] = env
            ms = []
            for r in range(4):
                st = h.SearchDevice(q.data_ptr(), nq, k, *out.ptrs(), ef=ef)
                if r:
                    ms.append(h.last_kernel_ms())
            keep[name] = (out.ids.cpu().numpy().copy(), out.sc.cpu().numpy().copy(), {kk: st[kk] for kk in ("n_dist", "n_exp", "n_hops")})
            bpq = B.hnsw_bytes_per_query(st["n_dist"] / nq, st["n_exp"] / nq, dim, quant, 16)
            t = float(np.median(ms)) / 1e3
            row[name] = {"ms_per_launch": t * 1e3, "min_ms": float(min(ms)), "queries_per_s": nq / t, "frac_of_hbm_peak": bpq * nq / t / 8e12}
        for name, env in (("eight_lanes", None), ("lane_pairs", "0")):   # ONE query per call (kernel time, median of 60)
            if env is None:
                os.environ.pop("COLTT_EV8", None)
            else:
                os.environ["COLTT_EV8"] = env
            one = []
            for r in range(70):
                h.SearchDevice(q.data_ptr() + (r % 64) * dim * 4, 1, k, *out.ptrs(), ef=ef)
                if r >= 10:
                    one.append(h.last_kernel_ms())
            row[name]["one_query_kernel_ms"] = float(np.median(one))
        os.environ.pop("COLTT_EV8", None)
        a, b = keep["eight_lanes"], keep["lane_pairs"]
        row["identical"] = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and a[2] == b[2])
        row["speedup"] = row["lane_pairs"]["ms_per_launch"] / row["eight_lanes"]["ms_per_launch"]
        res["ef"][str(ef)] = row
        print(json.dumps({str(ef): row}), file=sys.stderr, flush=True)
    res["rows8_launches"] = h.Rows8()[0]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
