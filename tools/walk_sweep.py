#!/usr/bin/env python3
"""A/B of the large-ef level-0 walk variants (coltt_amd/csrc/hnsw_walk2.hpp) on ONE index: COLTT_WALK2 = off (round-2 kernel) and
every OPT x profile combination compiled into the library (all 16 with a -DCOLTT_WALK_EXPERIMENTS build:
`COLTT_OUT=coltt_amd/variants/libcoltt_exp.so COLTT_OBJ=coltt_amd/variants/obj_exp COLTT_EXTRA_FLAGS=-DCOLTT_WALK_EXPERIMENTS
python -m coltt_amd.build`, then COLTT_LIB=<that .so>).

    python tools/walk_sweep.py [--n 10000000] [--quant 1] [--dataset lowrank:32:1.0] [--efs 256,1024] [--variants off,0,2,...]

Every variant must return the SAME ids, score bits and traversal counters as the round-2 kernel (checked here, per ef); the
table is kernel ms per launch of --queries queries (hipEvent pair on the search stream, best and median of --reps).
Prints one JSON object per (variant, ef) and a final summary line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--quant", type=int, default=1)
    ap.add_argument("--dataset", default="lowrank:32:1.0")
    ap.add_argument("--queries", type=int, default=10_000)
    ap.add_argument("--efs", default="256,1024")
    ap.add_argument("--variants", default="off,0,1,2,3,4,5,6,7,8,10,14,15")
    ap.add_argument("--bloom-kb", default="", help="comma list of COLTT_BLOOM_KB values tried for variants with the Bloom bit ('' = the library's choice)")
    ap.add_argument("--waves", default="", help="comma list of COLTT_WAVES_PER_CU values ('' = the library's choice)")
    ap.add_argument("--env-var", default="COLTT_WALK2", help="COLTT_WALK2 (ef > 128: HBM visited map) or COLTT_WALK2_LDS (ef <= 128: off,2,4,6)")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    import coltt_amd as G
    import bench as B
    assert G.lib().coltt_init(0) == 0
    dev = torch.device("cuda", 0)

    class A: m = 16; ef = 128; efc = 200; build_batch = 16384
    ds = B.Dataset(torch, dev, a.dim, a.dataset)
    t0 = time.time()
    h, build_s = B.build_index(G, torch, dev, ds, a.n, a.dim, A, 0xC0177 + 101, a.quant)
    gen = torch.Generator(device=dev); gen.manual_seed(0x5EED5 + 7)
    q = ds.rows(a.queries, gen)
    out = B.Out(torch, dev, a.queries, 10)
    print(json.dumps({"built": a.n, "dim": a.dim, "quant": a.quant, "dataset": a.dataset, "build_s": build_s, "lib": G.lib_path() if hasattr(G, "lib_path") else os.environ.get("COLTT_LIB", "default")}), flush=True)
    efs = [int(e) for e in a.efs.split(",") if e]
    ref = {}
    rows = []
    combos = []
    for v in [x for x in a.variants.split(",") if x]:
        bl = [""]
        if v != "off" and (int(v) & 1) and a.bloom_kb and a.env_var == "COLTT_WALK2":
            bl = a.bloom_kb.split(",")
        for b in bl:
            for wv in (a.waves.split(",") if a.waves else [""]):
                combos.append((v, b, wv))
    for v, b, wv in combos:
        os.environ[a.env_var] = v
        for key, val in (("COLTT_BLOOM_KB", b), ("COLTT_WAVES_PER_CU", wv)):
            if val: os.environ[key] = val
            else: os.environ.pop(key, None)
        for ef in efs:
            try:
                st = h.SearchDevice(q.data_ptr(), a.queries, 10, *out.ptrs(), ef=ef)   # warm-up
                ms = []
                for _ in range(a.reps):
                    st = h.SearchDevice(q.data_ptr(), a.queries, 10, *out.ptrs(), ef=ef)
                    ms.append(h.last_kernel_ms())
                ids = out.ids.cpu().numpy().copy(); sc = out.sc.cpu().numpy().view(np.uint32).copy(); cn = out.cnt.cpu().numpy().copy()
                if ef not in ref:
                    ref[ef] = (ids, sc, cn, st)
                r = ref[ef]
                same = bool(np.array_equal(ids, r[0]) and np.array_equal(sc, r[1]) and np.array_equal(cn, r[2]))
                same_ctr = all(st[k] == r[3][k] for k in ("n_dist", "n_exp", "n_hops"))
                nd = st["n_dist"] / a.queries; ne = st["n_exp"] / a.queries
                bpq = B.hnsw_bytes_per_query(nd, ne, a.dim, a.quant, A.m)
                best = min(ms)
                row = {"variant": v, "bloom_kb": b, "waves": wv, "ef": ef, "ms_best": best, "ms_median": float(np.median(ms)), "qps": a.queries / best * 1e3,
                       "frac_of_8TBs": bpq * a.queries / (best / 1e3) / 8e12, "n_dist": nd, "n_exp": ne, "same_answers": same, "same_counters": same_ctr}
            except Exception as e:   # a variant that is not compiled in, or a watchdog trip
                row = {"variant": v, "bloom_kb": b, "waves": wv, "ef": ef, "error": str(e)}
            rows.append(row)
            print(json.dumps(row), flush=True)
    bad = [r for r in rows if not r.get("error") and not (r["same_answers"] and r["same_counters"])]
    summary = {"n": a.n, "dim": a.dim, "quant": a.quant, "dataset": a.dataset, "build_s": build_s, "queries": a.queries, "rows": rows,
               "all_variants_identical_to_first": not bad, "wall_s": time.time() - t0}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(summary, f, indent=1)
    print(json.dumps({"summary": {"identical": not bad, "wall_s": summary["wall_s"]}}), flush=True)


if __name__ == "__main__":
    main()
