#!/usr/bin/env python3
"""One index, one process: ms per 10 000 queries of Hnsw.Search at the given ef under a list of knob settings (the policy snapshot is
reloaded by the binding when os.environ changes).  `python tools/knob_sweep.py <n> <quant> <dataset> <ef[,ef..]> "K=V,K2=V2" "K=V" ...`
("-" = defaults).  Answers and counters of every setting are compared with the first one's."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import coltt_amd as G
    import bench as B
    n, quant, dataset, efs = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], [int(e) for e in sys.argv[4].split(",")]   # one index, every ef in turn
    settings = sys.argv[5:] or ["-"]
    dim, k, nq = 768, 10, 10_000
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)

    class A: m = 16; ef = 128; efc = 200; build_batch = 16384
    ds = B.Dataset(torch, dev, dim, dataset)
    h, build_s = B.build_index(G, torch, dev, ds, n, dim, A, 0xC0177, quant)
    gen = torch.Generator(device=dev); gen.manual_seed(0x5EED5)
    q = ds.rows(nq, gen)
    out = B.Out(torch, dev, nq, k)
    for ef in efs:
        res = {"n": n, "quant": quant, "dataset": dataset, "ef": ef, "build_s": build_s, "rows": []}
        ref = None
        touched = set()
        for s in settings + settings[:1]:          # the first setting again at the end: drift check
            for kk in touched:
                os.environ.pop(kk, None)
            if s != "-":
                for kv in s.split(","):
                    a, b = kv.split("=")
                    os.environ[a] = b; touched.add(a)
            ms = []
            for r in range(4):
                st = h.SearchDevice(q.data_ptr(), nq, k, *out.ptrs(), ef=ef)
                if r:
                    ms.append(h.last_kernel_ms())
            cur = (out.ids.cpu().numpy().copy(), out.sc.cpu().numpy().copy(), {x: st[x] for x in ("n_dist", "n_exp", "n_hops")})
            if ref is None:
                ref = cur
            same = bool(np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1].view(np.uint32), ref[1].view(np.uint32)) and cur[2] == ref[2])
            bpq = B.hnsw_bytes_per_query(st["n_dist"] / nq, st["n_exp"] / nq, dim, quant, 16)
            t = float(np.median(ms)) / 1e3
            import hashlib
            sha = hashlib.sha256(cur[0].tobytes() + cur[1].tobytes() + json.dumps(cur[2], sort_keys=True).encode()).hexdigest()[:16]   # compare across libraries (COLTT_LIB)
            row = {"setting": s, "ef": ef, "ms": t * 1e3, "min_ms": float(min(ms)), "qps": nq / t, "frac_of_hbm_peak": bpq * nq / t / 8e12, "identical_to_first": same, "answers_sha16": sha,
                   "lib": os.path.basename(os.environ.get("COLTT_LIB", "libcoltt_gpu.so"))}
            res["rows"].append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
