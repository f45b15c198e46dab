#!/usr/bin/env python3
"""Generates tests/golden/hnsw_pyref.npz FROM oracle/pyref.py — the independent pure-Python restatement written from the Go
text (NOT from the C++ oracle).  Inputs and expected outputs only; the C++ oracle (CPU suite) and the HIP path (GPU suite)
must both reproduce them bit for bit.  Run: python tests/golden/make_golden_pyref.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402


def case(seed, n, d, metric, m, efc, algo, n_remove, nq, k, ef):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d)).astype(np.float32)
    ids = (rng.permutation(n).astype(np.uint64) * np.uint64(13) + np.uint64(7))
    mult = 1.0 / np.log(float(m))
    lv = np.floor(-np.log(1.0 - rng.random(n)) * mult).astype(np.int32)
    h = P.Hnsw(d, metric, m=m, ef_construction=efc, algo=algo)
    for i in range(n):
        assert h.insert(int(ids[i]), X[i], int(lv[i])) is None
    rem = rng.choice(n, n_remove, replace=False)
    for i in rem:
        assert h.remove(int(ids[i])) is None
    # inserts after removals (tombstones are skipped by every traversal)
    Y = rng.standard_normal((10, d)).astype(np.float32); yl = np.floor(-np.log(1.0 - rng.random(10)) * mult).astype(np.int32)
    yid = np.arange(10, dtype=np.uint64) + np.uint64(10**6)
    for i in range(10):
        assert h.insert(int(yid[i]), Y[i], int(yl[i])) is None
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    res_ids = np.zeros((nq, k), np.uint64); res_sc = np.zeros((nq, k), np.float32); res_n = np.zeros(nq, np.int32)
    for qi in range(nq):
        r = h.search(Q[qi], k, ef)
        res_n[qi] = len(r)
        for j, (i, s) in enumerate(r):
            res_ids[qi, j] = i; res_sc[qi, j] = s
    g = h.export()
    return dict(X=X, ids=ids, levels=lv, removed=ids[rem], Y=Y, y_ids=yid, y_levels=yl, Q=Q, res_ids=res_ids, res_scores=res_sc, res_n=res_n,
                cfg=np.array([d, metric, m, efc, algo, k, ef], np.int32), g_levels=g["levels"], g_deleted=g["deleted"],
                g_row_offsets=g["row_offsets"], g_nbr=g["nbr"], g_nbr_dist=g["nbr_dist"], g_entry=np.int32(g["entry"]))


def main():
    cases = [case(11, 220, 24, 0, 8, 40, 0, 30, 12, 10, 32),     # cosine, Simple, dim % 8 == 0
             case(12, 180, 19, 1, 6, 32, 1, 25, 12, 5, 20),      # l2, Heuristic(extend=false), ragged dim (scalar tail)
             case(13, 160, 33, 0, 16, 48, 0, 0, 8, 10, 64)]      # cosine, M=16, no removals
    out = {}
    for ci, c in enumerate(cases):
        for k, v in c.items():
            out[f"c{ci}_{k}"] = v
    out["n_cases"] = np.int32(len(cases))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "hnsw_pyref.npz"), **out)
    print("wrote hnsw_pyref.npz", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and k.startswith("c0_")})


if __name__ == "__main__":
    main()
